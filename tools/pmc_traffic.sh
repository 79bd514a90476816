#!/bin/bash
# HBM traffic and VALU instruction counts of the bench kernels: FETCH_SIZE, WRITE_SIZE and SQ_INSTS_VALU in rocprofv3 runs of their own
# (run on the GPU box from the repository root; results land under gpurun_out/pmc_traffic/).
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_traffic
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
# rthip_cast into page-locked memory (the bench's host_cast figure) renders a frame as four launches of a quarter each:
# one launch per frame here, so that the per-launch averages of the cast kernels are whole frames
export RTHIP_CAST_BANDS=1
# a workload named <name>+fast runs with --traversal fast
for W in ${@:-bunny instanced soup cornell soup-rays instanced-rays cornell-mesh heightfield bunny+fast instanced+fast soup+fast soup-rays+fast instanced-rays+fast heightfield+fast}; do
  SIZES=""
  [ "$W" = "bunny" ] && SIZES="10000000"
  BASE=${W%@samples=*}
  SAMPLES=0   # <name>@samples=K: K x K samples per pixel instead of the workload's own (the table keeps the suffix)
  [ "$BASE" != "$W" ] && SAMPLES=${W#*@samples=}
  NAME=${BASE%+fast}
  TRAV=exact
  [ "$NAME" != "$BASE" ] && TRAV=fast
  for C in FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU; do
    timeout -s KILL 240 rocprofv3 --pmc $C --output-format csv -d $OUT/${W}_${C} -- \
      python3 $ROOT/bench.py --workload $NAME --traversal $TRAV --samples $SAMPLES --streams 1 --steps 2 --warmup 0 --min-region-s 0 --no-cpu-baseline --no-workloads --build-sizes "$SIZES" \
      > $OUT/${W}_${C}.log 2>&1 || { echo "pass $W $C failed"; tail -5 $OUT/${W}_${C}.log; exit 1; }
    echo "pass $W $C done"
  done
done
python3 $ROOT/tools/pmc_traffic.py $OUT $ROOT/gpurun_out/traffic.json
