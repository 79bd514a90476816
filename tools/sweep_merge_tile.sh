#!/bin/bash
# Tile size of the sort-in-LDS-then-merge path and upper-bytes-first against all eight digits: per-kernel breakdown of
# one build of N boxes (default 81 920) for each setting.  Run from the repository root on the GPU box.
N=${1:-81920}
for tile in 1024 2048 4096; do
  for top in 4 8; do
    echo "== RTHIP_SORT_MERGE_TILE=$tile RTHIP_SORT_TOP_BYTES=$top N=$N"
    RTHIP_SORT_MERGE_TILE=$tile RTHIP_SORT_TOP_BYTES=$top RTHIP_SORT_MERGE_MAX=${MERGE_MAX:-131072} timeout -k 10 200 bash tools/prof_build.sh $N || exit 1
  done
done
