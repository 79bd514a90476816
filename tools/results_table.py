"""Prints the results table of the design notes (markdown) from a default bench line: python3 tools/results_table.py <json>"""
import json
import sys

line = [x for x in open(sys.argv[1]) if x.startswith("{")][-1]
b = json.loads(line)


def row(label, v, base=None):
    c = v.get("ceilings") or v.get("roofline", {}).get("ceilings") or {}
    pr = v.get("per_ray") or v.get("roofline", {}).get("per_ray") or {}
    cpu = v.get("cpu_baseline") or {}
    ratio = f" (×{v['value'] / base:.2f})" if base else ""
    bound = c.get("bounding")
    frac = c.get("frac_of_bounding")
    if bound == "hbm":
        btxt = f"HBM ({c.get('hbm_measured_frac'):.2f} of peak by measured bytes; {c.get('hbm_algorithmic_frac'):.2f} by algorithmic bytes)"
    elif bound == "valu-issue":
        btxt = f"VALU issue ({frac:.2f})"
    else:
        btxt = str(bound)
    toa = c.get("traffic_over_algorithmic")
    print(f"| {label} | {v['value']:,.0f}{ratio} | {v['ms_per_step']:.3g} | {pr.get('nodes')} | {btxt} | {toa} | {cpu.get('value', '')} |".replace(",", " "))


print("| workload | Mrays/s | ms per step | boxes / ray | bounding ceiling (fraction of it) | measured HBM bytes ÷ algorithmic | CPU reference, 1 core |")
print("|---|---|---|---|---|---|---|")
head = dict(b)
head["per_ray"] = b["roofline"]["per_ray"]
head["ceilings"] = b["roofline"]["ceilings"]
row("bunny-class 1080p, 1 spp, 2 bounces (headline)", head)
w = b["workloads"]
names = [("bunny_one_stream", "the same on ONE stream (one `rt_tracer_cast`)", None), ("instanced", "10 000 instances 1080p, 4 spp, 8 bounces", None),
         ("instanced_one_stream", "the same on one stream", None), ("soup", "10 M-triangle soup 1080p, 1 spp, 4 bounces", None),
         ("soup_rays", "10 M-triangle soup, 2²⁴ incoherent rays", None), ("heightfield", "10 M-triangle height field 1080p, 1 spp, 4 bounces", None),
         ("instanced_rays", "10 000 instances, 2²² incoherent rays", None), ("cornell", "Cornell 512², 4 spp, 8 bounces", None),
         ("cornell_mesh_4k", "Cornell + mesh 3840×2160, 16 spp, 8 bounces (one stream)", None),
         ("soup_rays_fast", "**ordered walk**: soup incoherent rays", "soup_rays"), ("soup_fast", "ordered walk: soup 1080p", "soup"),
         ("heightfield_fast", "ordered walk: height field 1080p", "heightfield"), ("instanced_rays_fast", "ordered walk: instances, incoherent rays", "instanced_rays"),
         ("instanced_fast", "ordered walk: 10 000 instances 1080p", "instanced"), ("bunny_fast", "ordered walk: bunny-class 1080p", None)]
for key, label, base in names:
    if key in w:
        row(label, w[key], w[base]["value"] if base else (b["value"] if key == "bunny_fast" else None))
hc = b.get("host_cast", {})
print("host_cast", hc.get("ms_per_call"), hc.get("value"), hc.get("pinned", {}).get("ms_per_call"), hc.get("pinned", {}).get("value"))
print("lbvh_build", b.get("lbvh_build"))
print("scene_build_ms", b.get("scene_build_ms"))
print("roofline", {k: b["roofline"][k] for k in ("bound", "achieved", "frac", "traffic", "kernel_ms")})
