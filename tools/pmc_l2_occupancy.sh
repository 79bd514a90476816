#!/bin/bash
# L2 / L1 hit rates and wave occupancy of the renderer kernels (rocprofv3 --pmc, one pass per counter block), for the
# workloads named on the command line (default: the headline frame, the instanced frame, the HBM-bound ray stream).
# Run on the GPU box from the repository root; prints one line per kernel and workload.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_l2occ
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
export RTHIP_CAST_BANDS=1
# a workload named <name>+fast runs with --traversal fast
for W in ${@:-bunny instanced soup-rays}; do
  NAME=${W%+fast}
  TRAV=exact
  [ "$NAME" != "$W" ] && TRAV=fast
  k=0
  for C in "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU"; do
    k=$((k+1))
    timeout -s KILL 240 rocprofv3 --pmc $C --output-format csv -d $OUT/${W}_$k -- \
      python3 $ROOT/bench.py --workload $NAME --traversal $TRAV --streams 1 --steps 2 --warmup 0 --min-region-s 0 --no-cpu-baseline --no-workloads --build-sizes "" \
      > $OUT/${W}_$k.log 2>&1 || { echo "pass $W $k failed"; tail -5 $OUT/${W}_$k.log; exit 1; }
  done
done
python3 - <<PY
import csv, glob, collections, os
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for p in glob.glob("$OUT/*/*/*counter_collection.csv"):
    w = os.path.basename(os.path.dirname(os.path.dirname(p))).rsplit("_", 1)[0]
    for r in csv.DictReader(open(p)):
        n = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "")
        if not any(t in n for t in ("cast_", "trace_rays", "wf_trace")) or "<true" in n:
            continue  # (the counting launches have COUNT = true as their first template argument)
        acc[(w, n.split("(")[0].split("<")[0])][r["Counter_Name"]].append(float(r["Counter_Value"]))
for (w, k), c in sorted(acc.items()):
    m = {name: sum(v) / len(v) for name, v in c.items()}
    g = lambda name: m.get(name, float("nan"))
    l2 = g("TCC_HIT_sum") / max(g("TCC_HIT_sum") + g("TCC_MISS_sum"), 1)
    l1 = 1.0 - g("TCP_TCC_READ_REQ_sum") / max(g("TCP_TOTAL_CACHE_ACCESSES_sum"), 1)
    # SQ_WAVE_CYCLES / SQ_BUSY_CYCLES: waves resident per shader engine while it is busy (both are summed over the chip's
    # engines by the same rule); per SIMD: / (CUs x 4 SIMDs per engine-instance counted) is left to the reader, the ratio
    # to the same figure of another workload is what matters
    print(f"{w:16s} {k:30s} L2 hit {l2:6.3f}  L1 hit {l1:6.3f}  waves {g('SQ_WAVES'):12.0f}  wave-cycles / busy-cycles {g('SQ_WAVE_CYCLES') / max(g('SQ_BUSY_CYCLES'), 1):8.2f}"
          f"  VALU insts {g('SQ_INSTS_VALU'):14.0f}  lanes per VALU inst {g('SQ_THREAD_CYCLES_VALU') / max(g('SQ_ACTIVE_INST_VALU'), 1) / 4:6.1f}"
          f"  waiting share {g('SQ_WAIT_ANY') / max(g('SQ_WAVE_CYCLES'), 1):6.3f}  waiting for an instruction {g('SQ_WAIT_INST_ANY') / max(g('SQ_WAVE_CYCLES'), 1):6.3f}")
PY
