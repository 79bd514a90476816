#!/bin/bash
# The round's build and headline artefacts in one GPU call (run from the repository root on the GPU box):
# counter pass of the default workload (its build kernels at 10^7 included), per-kernel build breakdowns at the
# benchmark's sizes, the hand-off modes, the default bench line and its kernel-trace summary.  Results under gpurun_out/final/.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/final
mkdir -p $OUT
cd $ROOT
bash tools/pmc_traffic.sh bunny > $OUT/pmc_build_10M.txt 2>&1 || { tail -5 $OUT/pmc_build_10M.txt; exit 1; }
cp gpurun_out/traffic.json $OUT/traffic_bunny.json
timeout -k 10 200 bash tools/prof_build.sh 81920 > $OUT/build_81920_breakdown.txt 2>&1 || exit 1
timeout -k 10 200 bash tools/prof_build.sh 1000000 > $OUT/build_1M_breakdown.txt 2>&1 || exit 1
timeout -k 10 200 bash tools/prof_build.sh 10000000 > $OUT/build_10M_breakdown.txt 2>&1 || exit 1
timeout -k 10 200 bash tools/prof_build.sh 10000000 heightfield > $OUT/build_10M_heightfield_breakdown.txt 2>&1 || exit 1
bash tools/run_build_modes.sh > $OUT/build_handoff_modes.txt 2>&1 || exit 1
timeout -k 10 500 python3 -u bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err || { tail -5 $OUT/bench_default.err; exit 1; }
cd /tmp; export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_default -- python3 $ROOT/bench.py --no-cpu-baseline > $OUT/prof_default.log 2>&1 || { tail -5 $OUT/prof_default.log; exit 1; }
cd $ROOT
cp $(find $OUT/prof_default -name '*kernel_stats.csv' | head -1) $OUT/bench_default_kernel_stats.csv
echo done
