"""dev tool: best duration of every hdr_* kernel per grid size, from a rocprofv3 kernel trace of tools/probe_hdr.py."""
import csv, glob, sys
rows = list(csv.DictReader(open(glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0])))
best = {}
for r in rows:
    if "hdr_" in r["Kernel_Name"]:
        k = (r["Kernel_Name"].split("(")[0].split("::")[-1], r.get("Grid_Size") or r.get("Grid_Size_X"))
        t = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        best[k] = min(best.get(k, 1e9), t)
for k, v in sorted(best.items(), key=lambda kv: (int(kv[0][1]), kv[0][0])):
    print(f"{k[0]:22s} grid {k[1]:>9s}  {v:8.1f} us")
