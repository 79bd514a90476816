mkdir -p gpurun_out/r3
OUT=gpurun_out/r3/kernel_choice_4streams.txt
rm -f $OUT
run() { timeout -k 10 300 python bench.py "$@" --no-config5 --no-workloads --no-cpu-baseline --build-sizes "" 2>>gpurun_out/r3/sweep7_err.txt | python -c "
import sys, json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], d.get('ms_per_frame'), d['roofline'].get('lane_utilisation'), d['roofline'].get('kernel'))
" >> $OUT; }
for t in exact fast; do
  echo "== bunny $t auto" >> $OUT; run --workload bunny --traversal $t || exit 1
  for r in 16 32 48; do
    echo "== bunny $t persistent refill $r" >> $OUT; RTHIP_CAST=persistent RTHIP_REFILL=$r run --workload bunny --traversal $t || exit 1
  done
done
for t in exact fast; do
  echo "== cornell $t auto" >> $OUT; run --workload cornell --traversal $t || exit 1
  for r in 16 32; do
    echo "== cornell $t persistent refill $r" >> $OUT; RTHIP_CAST=persistent RTHIP_REFILL=$r run --workload cornell --traversal $t || exit 1
  done
  echo "== cornell $t static" >> $OUT; RTHIP_CAST=static run --workload cornell --traversal $t || exit 1
done
cat $OUT
