#!/bin/bash
# The ordered walk's figures per workload for library builds with other stack depths (tools/variants/librthip_fs<N>.so,
# made with -DRT_FAST_STACK=<N>); "default" is the product library.  Run from the repository root on the GPU box.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
for lib in default "$@"; do
  if [ "$lib" = default ]; then unset RTHIP_LIBRARY; else export RTHIP_LIBRARY=$ROOT/tools/variants/librthip_$lib.so; fi
  for w in bunny instanced soup soup-rays heightfield instanced-rays; do
    for streams in 4 1; do
      line=$(timeout -k 10 300 python3 -u bench.py --traversal fast --workload $w --streams $streams --steps 30 --warmup 6 --build-sizes "" --no-cpu-baseline --no-workloads 2>/dev/null | grep '^{' | tail -1) || exit 1
      python3 - "$lib" "$w" "$streams" "$line" <<'PY'
import json, sys
lib, w, streams, line = sys.argv[1:5]
d = json.loads(line)
r = d.get("roofline", {})
print(f"{lib:8s} {w:15s} streams {streams}: {d['value']:10.1f} Mrays/s  {d['ms_per_step']:9.4f} ms  kernel {r.get('kernel')}  per ray {r.get('per_ray')}", flush=True)
PY
    done
  done
done
