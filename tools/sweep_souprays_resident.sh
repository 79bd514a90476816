mkdir -p gpurun_out/r3
OUT=gpurun_out/r3/souprays_resident.txt
rm -f $OUT
run() { timeout -k 10 300 python bench.py "$@" --no-config5 --no-workloads --no-cpu-baseline 2>>gpurun_out/r3/sweep6_err.txt | python -c "
import sys, json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], d.get('ms_per_frame'), d['roofline'].get('lane_utilisation'), d['roofline'].get('kernel'))
" >> $OUT; }
echo "== soup-rays fast auto" >> $OUT; run --workload soup-rays --traversal fast || exit 1
for r in 8 16 32; do
echo "== soup-rays fast resident refill $r" >> $OUT; RTHIP_CAST=persistent RTHIP_REFILL=$r run --workload soup-rays --traversal fast || exit 1
done
echo "== soup-rays exact auto" >> $OUT; run --workload soup-rays || exit 1
echo "== soup-rays exact resident refill 16" >> $OUT; RTHIP_CAST=persistent RTHIP_REFILL=16 run --workload soup-rays || exit 1
cat $OUT
