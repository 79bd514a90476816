#!/bin/bash
# Per-kernel times of one LBVH build (default 10^7 triangles): rocprofv3 kernel trace of tools/probe_build.py
N=${1:-10000000}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_build_$N
rm -rf $OUT; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
timeout -s KILL 180 rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $ROOT/tools/probe_build.py $N $2 > $OUT/run.log 2>&1 || { tail -5 $OUT/run.log; exit 1; }
python3 $ROOT/tools/build_breakdown.py $(find $OUT -name '*kernel_trace.csv' | head -1)
