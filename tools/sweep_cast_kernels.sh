#!/bin/bash
# dev helper: compare cast implementations / knobs
run() {
  for w in bunny cornell instanced soup; do
    env "$@" timeout -k 10 200 python -u bench.py --workload $w --steps 10 --warmup 2 --no-cpu-baseline --build-sizes "" 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$* $w', d['value'], 'Mrays/s  ms_per_step', d['ms_per_step'], 'lane_util', d['roofline']['lane_utilisation'])" || return 1
  done
}
run A=1 &&
run RTHIP_CAST=static &&
run RTHIP_CAST=persistent
