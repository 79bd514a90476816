#!/bin/bash
# Sweep of the persistent kernel's refill threshold under the ordered walk; lines into gpurun_out/r3/fast_refill_sweep.txt.
mkdir -p gpurun_out/r3
OUT=gpurun_out/r3/fast_refill_sweep.txt
rm -f $OUT
run() { timeout -k 10 200 python bench.py "$@" --no-config5 --no-workloads --no-cpu-baseline 2>>gpurun_out/r3/sweep4_err.txt | python -c "
import sys, json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], d.get('ms_per_frame'), d['roofline'].get('lane_utilisation'))
" >> $OUT; }
for w in instanced heightfield soup; do
  for r in 8 16 24 32 48; do
    echo "== $w fast persistent refill $r" >> $OUT
    RTHIP_CAST=persistent RTHIP_REFILL=$r run --workload $w --traversal fast || exit 1
  done
done
for r in 8 16 24 32; do
  echo "== instanced exact persistent refill $r batch 4" >> $OUT
  RTHIP_CAST=persistent RTHIP_REFILL=$r RTHIP_INST_BATCH=4 run --workload instanced || exit 1
done
cat $OUT
