mkdir -p gpurun_out/r3
OUT=gpurun_out/r3/exact_refill_sweep.txt
rm -f $OUT
run() { timeout -k 10 300 python bench.py "$@" --no-config5 --no-workloads --no-cpu-baseline 2>>gpurun_out/r3/sweep5_err.txt | python -c "
import sys, json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], d.get('ms_per_frame'), d['roofline'].get('lane_utilisation'), d['roofline'].get('kernel'))
" >> $OUT; }
for w in heightfield soup; do
  echo "== $w exact auto" >> $OUT; run --workload $w || exit 1
  for r in 16 24 32 48; do
    echo "== $w exact persistent refill $r" >> $OUT
    RTHIP_CAST=persistent RTHIP_REFILL=$r run --workload $w || exit 1
  done
done
echo "== cornell-mesh exact auto" >> $OUT; RTHIP_TUNE_LOG=1 run --workload cornell-mesh || exit 1
for r in 16 32 48; do
  echo "== cornell-mesh exact persistent refill $r batch 8" >> $OUT
  RTHIP_CAST=persistent RTHIP_REFILL=$r RTHIP_INST_BATCH=8 run --workload cornell-mesh || exit 1
done
cat $OUT; grep "rthip tune" gpurun_out/r3/sweep5_err.txt | tail -3
