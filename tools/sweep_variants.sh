#!/bin/bash
# Mrays/s per workload for the product library ("default") and for development builds under tools/variants/
# (librthip_<name>.so).  Usage: TRAVERSAL=exact|fast WORKLOADS="..." bash tools/sweep_variants.sh <name>...   (GPU box)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
TRAVERSAL=${TRAVERSAL:-exact}
WORKLOADS=${WORKLOADS:-"bunny instanced soup heightfield instanced-rays cornell-mesh"}
for lib in default "$@"; do
  if [ "$lib" = default ]; then unset RTHIP_LIBRARY; else export RTHIP_LIBRARY=$ROOT/tools/variants/librthip_$lib.so; fi
  for w in $WORKLOADS; do
    for streams in 4 1; do
      line=$(timeout -k 10 300 python3 -u bench.py --traversal $TRAVERSAL --workload $w --streams $streams --steps 30 --warmup 6 --build-sizes "" --no-cpu-baseline --no-workloads 2>/dev/null | grep '^{' | tail -1) || exit 1
      python3 - "$lib" "$w" "$streams" "$line" <<'PY'
import json, sys
lib, w, streams, line = sys.argv[1:5]
d = json.loads(line)
r = d.get("roofline", {})
print(f"{lib:8s} {w:15s} streams {streams}: {d['value']:10.1f} Mrays/s  {d['ms_per_step']:9.4f} ms  kernel {r.get('kernel')}", flush=True)
PY
    done
  done
done
