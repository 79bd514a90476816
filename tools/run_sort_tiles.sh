#!/bin/bash
# LBVH build time at middle sizes with the radix passes in 1024-key tiles (RTHIP_SORT_SMALL_TILE_MAX >= n) and in 4096-key tiles (= 0)
set -e
for N in 200000 500000 1000000 2000000 4000000 10000000; do
  for M in 0 100000000; do
    echo -n "small-tile max $M: "
    RTHIP_SORT_SMALL_TILE_MAX=$M timeout -k 10 120 python3 tools/probe_build.py $N
  done
done
