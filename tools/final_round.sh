#!/bin/bash
# The artefacts the round's documents quote, from one tree in one GPU call (run from the repository root on the GPU box):
# counter passes of every workload in both walks (-> profiles/traffic.json, which the bench lines quote), the default
# bench line, its kernel-trace summary, the ordered walk's record conversion.  Results under gpurun_out/final/.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/final
mkdir -p $OUT
cd $ROOT
bash tools/pmc_traffic.sh > $OUT/pmc_traffic.log 2>&1 || { tail -5 $OUT/pmc_traffic.log; exit 1; }
cp gpurun_out/traffic.json $OUT/traffic.json
cp gpurun_out/traffic.json profiles/traffic.json
python3 tools/pmc_traffic.py gpurun_out/pmc_traffic $OUT/traffic_check.json > $OUT/pmc_traffic.txt
timeout -k 10 500 python3 -u bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err || { tail -5 $OUT/bench_default.err; exit 1; }
cd /tmp; export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_default -- python3 $ROOT/bench.py --no-cpu-baseline > $OUT/prof_default.log 2>&1 || { tail -5 $OUT/prof_default.log; exit 1; }
cp $(find $OUT/prof_default -name '*kernel_stats.csv' | head -1) $OUT/bench_default_kernel_stats.csv
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_wide -- python3 $ROOT/tools/probe_wide_convert.py > $OUT/wide_convert.txt 2>&1 || { tail -5 $OUT/wide_convert.txt; exit 1; }
grep -E 'wide_|scan_' $(find $OUT/prof_wide -name '*kernel_stats.csv' | head -1) >> $OUT/wide_convert.txt
cd $ROOT
timeout -k 10 300 python3 -u tools/probe_fast.py heightfield3163 > $OUT/probe_fast_3163.txt 2>&1 || { tail -5 $OUT/probe_fast_3163.txt; exit 1; }
echo done
