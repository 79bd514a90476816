#!/bin/bash
# Every measurement and profile of a round, in one GPU call (run from the repository root on the GPU box).
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/round
mkdir -p $OUT
cd $ROOT
# the counter passes first: the bench lines quote the traffic and instruction counts of THIS tree from profiles/traffic.json
bash tools/pmc_traffic.sh bunny instanced soup cornell soup-rays instanced-rays cornell-mesh > $OUT/pmc_traffic.log 2>&1 || exit 1
cp gpurun_out/traffic.json $OUT/traffic.json
python3 tools/pmc_traffic.py gpurun_out/pmc_traffic $OUT/traffic.json > $OUT/pmc_traffic.txt
cp $OUT/traffic.json profiles/traffic.json
timeout -k 10 400 python3 -u bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err || exit 1
timeout -k 10 400 python3 -u bench.py --streams 1 --no-cpu-baseline --build-sizes "" > $OUT/bench_default_streams1.json 2> $OUT/bench_default_streams1.err || exit 1
for w in cornell instanced soup soup-rays instanced-rays cornell-mesh; do
  timeout -k 10 400 python3 -u bench.py --workload $w --steps 40 --warmup 8 --build-sizes "" > $OUT/bench_$w.json 2> $OUT/bench_$w.err || exit 1
done
WORKLOADS="bunny cornell instanced soup cornell-mesh" bash tools/sweep_cast_kernels.sh RTHIP_CAST=static > $OUT/sweep.txt 2>&1 || exit 1
WORKLOADS="bunny cornell instanced soup cornell-mesh" bash tools/sweep_cast_kernels.sh RTHIP_CAST=persistent RTHIP_REFILL=16 >> $OUT/sweep.txt 2>&1 || exit 1
WORKLOADS="bunny cornell instanced soup cornell-mesh" bash tools/sweep_cast_kernels.sh RTHIP_CAST=persistent RTHIP_REFILL=32 >> $OUT/sweep.txt 2>&1 || exit 1
WORKLOADS="bunny cornell instanced soup cornell-mesh" bash tools/sweep_cast_kernels.sh RTHIP_CAST=wavefront >> $OUT/sweep.txt 2>&1 || exit 1
cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_default -- python3 $ROOT/bench.py --no-cpu-baseline > $OUT/prof_default.log 2>&1 || exit 1
cd $ROOT
python3 tools/build_breakdown.py $OUT/prof_default/*/*kernel_trace.csv > $OUT/build_10M_breakdown.txt
echo done
