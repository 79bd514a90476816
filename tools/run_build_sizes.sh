#!/bin/bash
# LBVH build time of random boxes at small and middle sizes for each tile size of the sort in tiles, and with all eight
# digits sorted from the start (tools/probe_build_sizes.py).  Run from the repository root on the GPU box.
echo -n "tiles of 2048, upper bytes first:  "; timeout -k 10 100 python3 tools/probe_build_sizes.py 2>/dev/null || exit 1
echo -n "tiles of 1024, upper bytes first:  "; RTHIP_SORT_MERGE_TILE=1024 timeout -k 10 100 python3 tools/probe_build_sizes.py 2>/dev/null || exit 1
echo -n "tiles of 1024, all eight digits:   "; RTHIP_SORT_MERGE_TILE=1024 RTHIP_SORT_TOP_BYTES=8 timeout -k 10 100 python3 tools/probe_build_sizes.py 2>/dev/null || exit 1
echo -n "tiles of 4096, upper bytes first:  "; RTHIP_SORT_MERGE_TILE=4096 timeout -k 10 100 python3 tools/probe_build_sizes.py 2>/dev/null || exit 1
