#!/bin/bash
# dev helper: compare cast implementations / knobs, one launch after the other (--streams 1).  usage: sweep_cast_kernels.sh [ENV=VALUE ...] -- per run set
run() {
  for w in ${WORKLOADS:-bunny cornell instanced soup}; do
    env "$@" timeout -k 10 300 python -u bench.py --workload $w --streams 1 --steps 10 --warmup 2 --no-cpu-baseline --build-sizes "" 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$* $w', d['value'], 'Mrays/s  ms_per_step', d['ms_per_step'], 'lane_util', d['roofline']['lane_utilisation'])" || return 1
  done
}
if [ $# -eq 0 ]; then
  run RTHIP_CAST=auto && run RTHIP_CAST=static && run RTHIP_CAST=persistent
else
  run "$@"
fi
