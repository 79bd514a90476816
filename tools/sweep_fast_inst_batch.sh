#!/bin/bash
# Sweep of the batch size of deferred instance entries in the ordered walk (RTHIP_FAST_INST_BATCH) on the instanced
# workloads; one line per run into gpurun_out/r3/fib_sweep.txt.
mkdir -p gpurun_out/r3
for b in 0 4 8 16 32; do
  for w in instanced instanced-rays; do
    echo "== batch $b workload $w" >> gpurun_out/r3/fib_sweep.txt
    RTHIP_FAST_INST_BATCH=$b timeout -k 10 120 python bench.py --workload $w --traversal fast --no-config5 --no-workloads --no-cpu-baseline 2>>gpurun_out/r3/fib_err.txt | python -c "
import sys, json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], d['unit'], d.get('ms_per_frame'), d.get('roofline',{}).get('achieved'))
" >> gpurun_out/r3/fib_sweep.txt || exit 1
  done
done
