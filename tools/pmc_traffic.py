"""Turns the FETCH_SIZE / WRITE_SIZE / SQ_INSTS_VALU passes of tools/pmc_traffic.sh into profiles/traffic.json.

Per workload and kernel: mean counter value per launch and
    hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024
FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE tallies the 128-byte read requests of
the L2's memory side at 64 bytes each, hence the factor 2 on the read side.
valu_insts_per_launch = SQ_INSTS_VALU: vector-ALU instructions issued, counted per wave (one wave64 instruction
occupies its SIMD for 4 cycles).
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def short(name: str) -> str:
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    name = name.split("(")[0]
    return name.split("::")[-1].strip()


def main(src: str, dst: str) -> None:
    table = defaultdict(lambda: defaultdict(lambda: defaultdict(list)))
    for d in sorted(glob.glob(os.path.join(src, "*_SIZE")) + glob.glob(os.path.join(src, "*_SQ_INSTS_VALU"))):
        base = os.path.basename(d)
        counter = "SQ_INSTS_VALU" if base.endswith("_SQ_INSTS_VALU") else "_".join(base.rsplit("_", 2)[1:])
        workload = base[: -len(counter) - 1]
        for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(path) as f:
                for row in csv.DictReader(f):
                    if row["Counter_Name"] != counter:
                        continue
                    k = short(row["Kernel_Name"])
                    if k.startswith("__amd") or "at::" in row["Kernel_Name"]:
                        continue  # runtime fills / copies, torch kernels
                    renderer = k.startswith(("cast_", "trace_rays", "occlusion_", "primary_hits", "wf_"))
                    if renderer and "<true" in k:
                        continue  # the counting launches (first template argument of the renderer kernels)
                    if not k.startswith("climb_"):  # the two climbs of the builder are told apart by their arguments
                        k = k.split("<")[0]
                    table[workload][k][counter].append((int(row["Grid_Size"]), float(row["Counter_Value"])))
    out = {}
    for workload, kernels in table.items():
        out[workload] = {}
        for k, c in kernels.items():
            f = c.get("FETCH_SIZE", [])
            w = c.get("WRITE_SIZE", [])
            if not f or not w:
                continue
            # a kernel launched at several sizes (LBVH builds): keep the largest grid, i.e. the biggest input
            f = [v for g, v in f if g == max(g for g, _ in f)]
            w = [v for g, v in w if g == max(g for g, _ in w)]
            fm, wm = sum(f) / len(f), sum(w) / len(w)
            out[workload][k] = dict(launches=len(f), fetch_size_kib=round(fm, 1), write_size_kib=round(wm, 1),
                                    hbm_bytes_per_launch=int((2.0 * fm + wm) * 1024))
            v = c.get("SQ_INSTS_VALU", [])
            if v:
                v = [x for g, x in v if g == max(g for g, _ in v)]
                out[workload][k]["valu_insts_per_launch"] = int(sum(v) / len(v))
    # the revision the counters belong to: bench.py hands a counter out only while the kernel's sources still have these
    # hashes (`git hash-object` of every file under raytracer_amd/csrc)
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    import datetime

    from bench import source_hashes

    out["_recorded_on"] = {"sources": source_hashes(), "when": datetime.date.today().isoformat()}
    with open(dst, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    del out["_recorded_on"]
    for workload in out:
        for k, r in sorted(out[workload].items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"]):
            print(f"{workload:10s} {k:28s} x{r['launches']:<3d} fetch {r['fetch_size_kib']:>12.1f} KiB  write {r['write_size_kib']:>12.1f} KiB"
                  f"  -> {r['hbm_bytes_per_launch'] / 1e6:10.2f} MB/launch")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
