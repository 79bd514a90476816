#!/bin/bash
# L1 (TCP) hit rate of the cast kernels of one bench workload.  usage: pmc_tcp.sh WORKLOAD [ENV=VALUE ...]
W=${1:-bunny}; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp; export TMPDIR=/tmp
out=$ROOT/gpurun_out/pmc_tcp_$W
env "$@" timeout -s KILL 200 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum --output-format csv -d $out -- python3 $ROOT/bench.py --workload $W --streams 1 --steps 2 --warmup 0 --no-cpu-baseline --build-sizes "" > $out.log 2>&1 || { tail -3 $out.log; exit 1; }
python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(list)
for p in glob.glob("$out/*/*counter_collection.csv"):
    for r in csv.DictReader(open(p)):
        n=r['Kernel_Name']
        if ('cast_' in n or 'trace_rays' in n) and '<true' not in n:
            acc[(n.split('(')[0].split('::')[-1], r['Counter_Name'])].append(float(r['Counter_Value']))
for (k,c),v in sorted(acc.items()):
    print(f"$W {k:30s} {c:32s} {sum(v)/len(v):16.0f}")
PY
