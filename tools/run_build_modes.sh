#!/bin/bash
# LBVH build time at the benchmark's sizes under the three hand-off modes of the upper hierarchy kernel
# (RTHIP_REFIT_FENCE: 0 acquire-release counter [default], 1 __threadfence, 2 relaxed + s_waitcnt).
set -e
for N in 81920 1000000 10000000; do
  for M in 2 0 1; do
    echo -n "mode $M: "
    RTHIP_REFIT_FENCE=$M timeout -k 10 120 python3 tools/probe_build.py $N
  done
done
