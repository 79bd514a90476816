#!/bin/bash
# Sweep of deferred instance entries in the EXACT walk's resident kernels: the ray-stream kernel (RTHIP_RAYS_INST_BATCH)
# and the persistent renderer selected by name (RTHIP_INST_BATCH); one line per run into gpurun_out/r3/exact_batch_sweep.txt.
mkdir -p gpurun_out/r3
OUT=gpurun_out/r3/exact_batch_sweep.txt
rm -f $OUT
run() { timeout -k 10 200 python bench.py "$@" --no-config5 --no-workloads --no-cpu-baseline 2>>gpurun_out/r3/sweep2_err.txt | python -c "
import sys, json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], d.get('ms_per_frame'))
" >> $OUT; }
for b in 0 4 8 16; do
  echo "== rays batch $b instanced-rays exact" >> $OUT
  RTHIP_RAYS_INST_BATCH=$b run --workload instanced-rays || exit 1
done
for b in 0 4 8 16; do
  echo "== frame: persistent refill 16 batch $b instanced exact" >> $OUT
  RTHIP_CAST=persistent RTHIP_REFILL=16 RTHIP_INST_BATCH=$b run --workload instanced || exit 1
done
for b in 0 8; do
  echo "== frame: persistent refill 16 batch $b cornell-mesh exact" >> $OUT
  RTHIP_CAST=persistent RTHIP_REFILL=16 RTHIP_INST_BATCH=$b run --workload cornell-mesh || exit 1
done
cat $OUT
