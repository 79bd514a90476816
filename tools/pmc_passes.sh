#!/bin/bash
# SQ counter passes for one bench workload (8 SQ counters per pass).  usage: pmc_passes.sh WORKLOAD [ENV=VALUE ...]
W=${1:-bunny}; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp; export TMPDIR=/tmp
k=0
for pass in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VALU" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_INSTS_BRANCH" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  k=$((k+1))
  out=$ROOT/gpurun_out/pmc_${W}_pass$k
  env "$@" timeout -s KILL 200 rocprofv3 --pmc $pass --output-format csv -d $out -- python3 $ROOT/bench.py --workload $W --streams 1 --steps 2 --warmup 0 --no-cpu-baseline --build-sizes "" > $out.log 2>&1 || { tail -3 $out.log; exit 1; }
done
python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(list)
for p in glob.glob("$ROOT/gpurun_out/pmc_${W}_pass*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(p)):
        n=r['Kernel_Name']
        if ('cast_' in n or 'wf_trace' in n) and '<true' not in n:
            acc[(n.split('(')[0].split('::')[-1], r['Counter_Name'])].append((float(r['Counter_Value']), int(r['End_Timestamp'])-int(r['Start_Timestamp'])))
for (k,c),v in sorted(acc.items()):
    print(f"{k:30s} {c:26s} {sum(a for a,_ in v)/len(v):16.0f}   kernel {sum(b for _,b in v)/len(v)/1e6:9.3f} ms")
PY
