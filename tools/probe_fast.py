"""RTHIP_TRAV_FAST against RTHIP_TRAV_EXACT on the benchmark's scenes: how many rays report another primitive / another
distance, box tests per ray, and Mrays/s of both walks.  Run on the GPU box: python3 tools/probe_fast.py [names...]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from raytracer_amd import abi, scenes, tracer as T


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


def rays_case(tr, o, d, label, out):
    n = o.shape[0]
    stream = torch.cuda.current_stream().cuda_stream
    res = {}
    for mode in ("exact", "fast"):
        T.set_traversal(tr, mode)
        hits = torch.zeros((n, 4), dtype=torch.int32, device="cuda")
        run = lambda: tr.check(tr.lib.rthip_trace_rays_device(tr.h, o.data_ptr(), d.data_ptr(), n, hits.data_ptr(), stream), "trace")
        T.set_counting(tr, True)
        T.get_stats(tr)
        run()
        st = T.get_stats(tr)
        T.set_counting(tr, False)
        ms = timed(run)
        res[mode] = dict(hits=hits.clone(), ms=ms, nodes=st["nodes_tested"] / n, tris=st["tris_tested"] / n,
                         util=st["nodes_tested"] / max(1, 64 * st["wave_steps"]))
    a, b = res["exact"]["hits"], res["fast"]["hits"]
    prim = ((a[:, 0] != b[:, 0]) | (a[:, 1] != b[:, 1]))
    tbits = a[:, 2] != b[:, 2]
    hit_miss = (a[:, 0] != 0) != (b[:, 0] != 0)
    ta, tb = a[:, 2].view(torch.float32), b[:, 2].view(torch.float32)
    both = (a[:, 0] != 0) & (b[:, 0] != 0) & tbits
    rel = ((ta - tb).abs() / ta.abs().clamp_min(1e-30))[both]
    row = dict(case=label, rays=n, differ_primitive=int(prim.sum()), differ_t_bits=int(tbits.sum()), hit_vs_miss=int(hit_miss.sum()),
               same_t_other_primitive=int((prim & ~tbits).sum()), max_rel_t_diff=float(rel.max()) if rel.numel() else 0.0,
               fast_closer=int((both & (tb < ta)).sum()), exact_closer=int((both & (ta < tb)).sum()),
               exact=dict(ms=round(res["exact"]["ms"], 4), mrays=round(n / res["exact"]["ms"] / 1e3, 1), boxes_per_ray=round(res["exact"]["nodes"], 2),
                          tris_per_ray=round(res["exact"]["tris"], 3), lane_util=round(res["exact"]["util"], 3)),
               fast=dict(ms=round(res["fast"]["ms"], 4), mrays=round(n / res["fast"]["ms"] / 1e3, 1), boxes_per_ray=round(res["fast"]["nodes"], 2),
                         tris_per_ray=round(res["fast"]["tris"], 3), lane_util=round(res["fast"]["util"], 3)))
    row["speedup"] = round(res["exact"]["ms"] / res["fast"]["ms"], 2)
    print(json.dumps(row), flush=True)
    out.append(row)


def frame_case(tr, cam, W, H, samples, label, out, kernels=("auto",)):
    stream = torch.cuda.current_stream().cuda_stream
    c = cam(W, H, samples)
    for kernel in kernels:
        T.set_cast_kernel(tr, kernel)
        res = {}
        for mode in ("exact", "fast"):
            T.set_traversal(tr, mode)
            fb = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
            run = lambda: tr.check(tr.lib.rthip_cast_device(tr.h, abi.ptr(c), 1, fb.data_ptr(), W, H, None, stream), "cast")
            T.set_counting(tr, True)
            T.get_stats(tr)
            run()
            st = T.get_stats(tr)
            T.set_counting(tr, False)
            ms = timed(run)
            res[mode] = dict(fb=fb.clone(), ms=ms, rays=st["rays"], nodes=st["nodes_tested"] / max(st["rays"], 1), kernel=T.get_stats(tr)["last_cast_kernel"])
        a, b = res["exact"]["fb"], res["fast"]["fb"]
        diff = (a.view(torch.int32) != b.view(torch.int32)).any(dim=2)
        rmse = float(((a.double() - b.double()) ** 2).mean().sqrt())
        row = dict(case=label, kernel=kernel, pixels=W * H, pixels_differ=int(diff.sum()), rmse=rmse, mean=float(a.mean()), mean_fast=float(b.mean()),
                   exact=dict(ms=round(res["exact"]["ms"], 4), mrays=round(res["exact"]["rays"] / res["exact"]["ms"] / 1e3, 1), boxes_per_ray=round(res["exact"]["nodes"], 2), kernel=res["exact"]["kernel"]),
                   fast=dict(ms=round(res["fast"]["ms"], 4), mrays=round(res["fast"]["rays"] / res["fast"]["ms"] / 1e3, 1), boxes_per_ray=round(res["fast"]["nodes"], 2), kernel=res["fast"]["kernel"]))
        row["speedup"] = round(res["exact"]["ms"] / res["fast"]["ms"], 2)
        print(json.dumps(row), flush=True)
        out.append(row)
    T.set_cast_kernel(tr, "auto")


def main():
    names = sys.argv[1:] or ["cornell", "bunny", "instanced", "soup", "heightfield"]
    out = []
    for name in names:
        if name == "cornell":
            w, cam, sky = scenes.cornell()
            bounces = 8
        elif name == "bunny":
            w, cam, sky = scenes.bunny(*scenes.bunny_class_mesh())
            bounces = 2
        elif name == "instanced":
            v, i = scenes.bunny_class_mesh()
            w, cam, sky = scenes.instanced_grid(100, seed=1, verts=v, indices=i)
            bounces = 8
        elif name == "soup":
            w, cam, sky = scenes.soup_world(10_000_000, seed=1)
            bounces = 4
        elif name == "heightfield":
            w, cam, sky = scenes.heightfield_world(2236, seed=1)
            bounces = 4
        elif name == "heightfield3163":  # the same surface with twice the triangles (not in the default list)
            w, cam, sky = scenes.heightfield_world(3163, seed=1)
            bounces = 4
        tr = T.rt_make_tracer(T.RT_TracerSettings(max_bounces=bounces, sky=sky))
        T.rt_tracer_build_blas(tr, w)
        T.rt_tracer_build_tlas(tr, w)
        from oracle import oracle as O
        if name == "cornell":
            o, d = O.primary_rays(cam(512, 512, 1), 512, 512)
            rays_case(tr, torch.from_numpy(o).cuda(), torch.from_numpy(d).cuda(), "cornell 512x512 primary rays", out)
            frame_case(tr, cam, 512, 512, 2, "cornell 512x512 4 spp 8 bounces", out, ("auto", "static", "persistent"))
        elif name == "bunny":
            o, d = O.primary_rays(cam(1920, 1080, 1), 1920, 1080)
            rays_case(tr, torch.from_numpy(o).cuda(), torch.from_numpy(d).cuda(), "bunny-class 1920x1080 primary rays", out)
            frame_case(tr, cam, 1920, 1080, 1, "bunny-class 1080p 1 spp 2 bounces", out, ("auto", "static", "persistent"))
        elif name == "instanced":
            o, d = scenes.incoherent_rays_device(torch, 1 << 22, 2, (-125.0, -1.0, -125.0), (125.0, 3.0, 125.0))
            rays_case(tr, o, d, "10k instances, 2^22 incoherent rays", out)
            frame_case(tr, cam, 1920, 1080, 2, "10k instances 1080p 4 spp 8 bounces", out, ("auto", "static", "persistent"))
        elif name == "soup":
            o, d = scenes.incoherent_rays_device(torch, 1 << 24, 2)
            rays_case(tr, o[:100_000].contiguous(), d[:100_000].contiguous(), "10M soup, first 10^5 incoherent rays", out)
            rays_case(tr, o, d, "10M soup, 2^24 incoherent rays", out)
            frame_case(tr, cam, 1920, 1080, 1, "10M soup 1080p 1 spp 4 bounces", out, ("auto", "static", "persistent"))
        elif name == "heightfield":
            o, d = scenes.incoherent_rays_device(torch, 1 << 22, 2, (0.0, 0.0, 0.0), (1.0, 0.6, 1.0))
            rays_case(tr, o, d, "10M height field, 2^22 incoherent rays", out)
            frame_case(tr, cam, 1920, 1080, 1, "10M height field 1080p 1 spp 4 bounces", out, ("auto", "static", "persistent"))
        elif name == "heightfield3163":
            o, d = scenes.incoherent_rays_device(torch, 1 << 22, 2, (0.0, 0.0, 0.0), (1.0, 0.6, 1.0))
            rays_case(tr, o, d, "20M height field (3163^2 cells), 2^22 incoherent rays", out)
        T.rt_tracer_cleanup(tr)
    with open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "probe_fast.json"), "w") as f:
        json.dump(out, f, indent=1)


main()
