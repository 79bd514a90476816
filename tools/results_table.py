"""Writes the results table of the design document (between its RESULTS markers) from a default `python bench.py` line."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def bound(v):
    b, f = v["bounding"], v["frac_of_bounding"]
    if b == "hbm":
        return f"HBM {f:.2f} (measured bytes)"
    if b == "valu-issue":
        return f"VALU {f:.2f}"
    return "not recorded"


def ms(x):
    return f"{x:.0f}" if x >= 100 else f"{x:.3g}"


def row(label, v, cpu=None):
    c = v["ceilings"]
    cpu = cpu if cpu is not None else v.get("cpu_baseline", {}).get("value", "")
    return (f"| {label} | {v['value']:,.0f} | {ms(v['ms_per_frame'])} | {v['per_ray']['nodes']:.1f} | {v['lane_utilisation']:.2f} | "
            f"{bound(v)} | {c.get('traffic_over_algorithmic', '')} | {cpu} |")


def main(path):
    d = json.load(open(path))
    w = d["workloads"]
    head = dict(value=d["value"], ms_per_frame=d["ms_per_frame"], per_ray=d["roofline"]["per_ray"], lane_utilisation=d["roofline"]["lane_utilisation"],
                bounding=d["roofline"]["bounding"], frac_of_bounding=d["roofline"]["frac_of_bounding"], ceilings=d["roofline"]["ceilings"])
    rows = [
        "| workload (exact walk, 4 streams unless said) | Mrays/s | ms / frame | boxes / ray | lanes at work | ceiling reached | HBM bytes ÷ algorithmic | reference, 1 CPU core |",
        "|---|---|---|---|---|---|---|---|",
        row("**bunny-class 1080p, 1 spp, 2 bounces (headline)**", head, d["cpu_baseline"]["value"]),
        row("same, one stream", w["bunny_one_stream"]),
        row("10 000 instances 1080p, 4 spp, 8 bounces", w["instanced"]),
        row("10 M soup 1080p, 1 spp, 4 bounces", w["soup"]),
        row("10 M soup, 2²⁴ incoherent rays", w["soup_rays"]),
        row("10 M height field 1080p, 1 spp, 4 bounces", w["heightfield"]),
        row("10 000 instances, 2²² incoherent rays", w["instanced_rays"]),
        row("Cornell 512², 4 spp, 8 bounces", w["cornell"]),
        row("Cornell + mesh 4K, 16 spp, 8 bounces (one stream)", w["cornell_mesh_4k"]),
        row("same at 1024 spp (config 5's frame)", w["cornell_mesh_4k_1024spp"]),
    ]
    for key, label in (("bunny_fast", "bunny-class 1080p"), ("instanced_fast", "10 000 instances 1080p"), ("instanced_rays_fast", "instances, incoherent rays"),
                       ("soup_fast", "soup 1080p"), ("soup_rays_fast", "soup, incoherent rays"), ("heightfield_fast", "height field 1080p")):
        base = {"bunny_fast": d["value"], "instanced_fast": w["instanced"]["value"], "instanced_rays_fast": w["instanced_rays"]["value"],
                "soup_fast": w["soup"]["value"], "soup_rays_fast": w["soup_rays"]["value"], "heightfield_fast": w["heightfield"]["value"]}[key]
        rows.append(row(f"ordered: {label} (×{w[key]['value'] / base:.2f})", w[key]))
    b = d["lbvh_build"]
    r = d["regions"]
    text = "\n".join(rows) + "\n\n"
    text += (f"Headline: median of {r['count']} regions of {d['steps']} steps ({r['ms_per_step_min']}–{r['ms_per_step_max']} ms per step of {d['config']['frames_per_step']} frames); "
             f"one launch {d['roofline']['kernel_ms']} ms with three beside it = algorithmic {d['roofline']['achieved']:.0f} GB/s (`frac` {d['roofline']['frac']:.2f}; "
             f"cache-resident: measured HBM traffic {d['roofline']['traffic'] / 1e6:.0f} MB per launch).  "
             f"`rthip_cast` into pageable / page-locked host memory: {d['host_cast']['value']:,.0f} / {d['host_cast']['pinned']['value']:,.0f} Mrays/s; "
             f"shadow rays {d['shadow_rays']['value']:,.0f} Mrays/s; CPU restatement on {d['cpu_port_all_cores']['cores']} threads {d['cpu_port_all_cores']['value']:.1f} Mrays/s.\n\n")
    text += "LBVH build, vertices resident (random soup; 10⁷: " + f"{b['10000000']['ms_per_1M']:.4f} ms per 1 M" + "): " + "; ".join(
        f"{int(n):,} triangles **{v['ms']:.3f} ms** ({v['hbm_roofline_frac']:.3f} of the 444 B/triangle roofline)" for n, v in sorted(b.items(), key=lambda kv: int(kv[0])))
    text += (f".  `.hdr` file of the 1080p frame {d['hdr_output']['ms']} ms ({d['hdr_output']['hbm_roofline_frac']:.3f} of its roofline; reference writer {d['hdr_output'].get('cpu_ms', '?')} ms); "
             f"`rthip_obj_load` of the mesh {d['obj_load']['ms']} ms (reference loader {d['obj_load'].get('cpu_ms', '?')} ms).\n")
    design = os.path.join(ROOT, "DESIGN.md")
    s = open(design).read()
    a, z = "<!-- RESULTS -->\n", "<!-- /RESULTS -->\n"
    if "RESULTS_TABLE\n" in s:
        s = s.replace("RESULTS_TABLE\n", a + text + z)
    else:
        s = s[: s.index(a) + len(a)] + text + s[s.index(z):]
    open(design, "w").write(s)
    print(text)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r03_bench_default.json"))
