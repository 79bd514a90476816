#!/bin/bash
# dev helper: the cast implementations with frames overlapped on 4 streams (the bench's default), per workload.
for w in ${WORKLOADS:-bunny cornell instanced soup}; do
  for k in auto static static-samples persistent; do
    RTHIP_CAST=$k timeout -k 10 300 python -u bench.py --workload $w --streams 4 --steps 40 --warmup 8 --no-cpu-baseline --build-sizes "" 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$k $w streams=4', d['value'], 'Mrays/s  ms_per_step', d['ms_per_step'], d['roofline']['kernel'], 'lane_util', d['roofline']['lane_utilisation'])" || echo "$k $w failed"
  done
done
