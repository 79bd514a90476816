#!/bin/bash
# dev helper: PMC passes for the cast kernel, static vs persistent, one workload
W=${1:-bunny}
cd /tmp; export TMPDIR=/tmp
for mode in static persistent; do
 for pass in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VALU" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_INSTS_BRANCH"; do
  out=$GRAFT_REPO_ROOT/gpurun_out/pmc_${W}_${mode}_$(echo $pass | cut -c1-12 | tr ' ' '_')
  RTHIP_CAST=$mode RTHIP_REFILL=32 timeout -s KILL 120 rocprofv3 --pmc $pass --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/bench.py --workload $W --steps 2 --warmup 0 --no-cpu-baseline --build-sizes "" > $out.log 2>&1 || exit 1
 done
done
