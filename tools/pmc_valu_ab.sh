#!/bin/bash
# SQ_INSTS_VALU per launch of the cast kernels of one workload under two settings of an environment variable.
# Usage: bash tools/pmc_valu_ab.sh <workload> <VAR> <value_a> <value_b>      (GPU box, repository root)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
W=$1; VAR=$2
cd /tmp; export TMPDIR=/tmp
for V in $3 $4; do
  export $VAR=$V
  mkdir -p $ROOT/gpurun_out/pmc_ab
  OUT=$ROOT/gpurun_out/pmc_ab/${W}_${VAR}_$V
  timeout -s KILL 240 rocprofv3 --pmc SQ_INSTS_VALU --output-format csv -d $OUT -- python3 $ROOT/bench.py --workload $W --streams 1 --steps 2 --warmup 0 --no-cpu-baseline --no-workloads --build-sizes "" > $OUT.log 2>&1 || { tail -5 $OUT.log; exit 1; }
  python3 - $OUT "$VAR=$V" <<'PY'
import csv, glob, sys, collections
rows = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "cast" in r["Kernel_Name"]:
            rows[r["Kernel_Name"].split("(")[0][:90]].append(float(r["Counter_Value"]))
for k, v in rows.items():
    print(sys.argv[2], k, "launches", len(v), "VALU insts per launch %.0f" % (sum(v) / len(v)))
PY
done
