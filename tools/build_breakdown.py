"""Per-kernel time of the LAST LBVH build in a rocprofv3 --kernel-trace CSV (bench.py's largest build)."""
import csv
import sys
from collections import OrderedDict

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
short = lambda n: n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
last = max(i for i, r in enumerate(rows) if short(r["Kernel_Name"]).startswith("tri_prepare_kernel"))
acc = OrderedDict()
for r in rows[last:]:
    k = short(r["Kernel_Name"])
    if k.startswith("at::") or "elementwise" in k:
        break
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    c, t = acc.get(k, (0, 0.0))
    acc[k] = (c + 1, t + d)
    if k.startswith("emit_kernel"):
        break
total = 0.0
for k, (c, t) in acc.items():
    print(f"{k:34s} calls {c:3d} total {t:9.1f} us")
    total += t
print(f"sum {total:.1f} us (kernel time only; gaps between launches not included)")
