#!/bin/bash
# HBM traffic and VALU instruction counts of the bench kernels: FETCH_SIZE, WRITE_SIZE and SQ_INSTS_VALU in rocprofv3 runs of their own
# (run on the GPU box from the repository root; results land under gpurun_out/pmc_traffic/).
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_traffic
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for W in ${@:-bunny instanced soup cornell soup-rays}; do
  SIZES=""
  [ "$W" = "bunny" ] && SIZES="10000000"
  for C in FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU; do
    timeout -s KILL 240 rocprofv3 --pmc $C --output-format csv -d $OUT/${W}_${C} -- \
      python3 $ROOT/bench.py --workload $W --streams 1 --steps 2 --warmup 0 --no-cpu-baseline --build-sizes "$SIZES" \
      > $OUT/${W}_${C}.log 2>&1 || { echo "pass $W $C failed"; tail -5 $OUT/${W}_${C}.log; exit 1; }
    echo "pass $W $C done"
  done
done
python3 $ROOT/tools/pmc_traffic.py $OUT $ROOT/gpurun_out/traffic.json
