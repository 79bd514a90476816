#!/bin/bash
# HBM traffic + L2 hit rate of the incoherent-ray workload, static vs resident-wave kernel
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_souprays
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for mode in static auto; do
  for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" ; do
    tag=${mode}_$(echo $C | cut -d' ' -f1)
    RTHIP_CAST=$mode timeout -s KILL 300 rocprofv3 --pmc $C --output-format csv -d $OUT/$tag -- \
      python3 $ROOT/bench.py --workload soup-rays --streams 1 --steps 1 --warmup 0 --no-cpu-baseline --build-sizes "" \
      > $OUT/$tag.log 2>&1 || { echo "pass $tag failed"; tail -5 $OUT/$tag.log; exit 1; }
    echo "pass $tag done"
  done
done
