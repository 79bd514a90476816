#!/usr/bin/env python3
"""Headline benchmark: Mrays/s of the HIP rendering backend (whole job) + LBVH build time.

    python bench.py [--gpus N --steps K --warmup W --workload bunny|instanced|soup|cornell]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One step = one frame of the workload on EACH of the `--streams` HIP streams (default 4: a batch of four frames in flight
together), rendered through rthip_cast_accumulate_device with everything (BVHs, triangles, framebuffers) resident in HBM;
`--streams 1` makes a step one frame.  The timed region of exactly `--steps` steps is repeated until about half a second
of timed work has been done and the MEDIAN region is reported (`regions`).  With N > 1 every rank renders the same frame
with its own sample set (seed + rank) and the framebuffers are summed onto rank 0 with RCCL
(`torch.distributed.reduce`), i.e. N times the samples in the same wall time: weak scaling.  By
default the image is accumulated progressively: every step of every rank renders its own sample set
into the rank's accumulator and the ranks exchange ONE framebuffer sum per image, at the end of the
timed steps and inside the timed region (`--combine every` delivers a combined frame per step).
A ray is one TLAS closest-hit query (primary and bounce rays alike).

Rank 0 prints ONE JSON line.  `roofline` prices the cast kernel against HBM bandwidth using
the algorithmic byte model 48 + 32*nodes + 48*triangles + 64*instances per ray, with the
node / triangle / instance counts taken from the kernel's own traversal counters (a separate,
untimed launch).  `cpu_baseline` times the CPU reference on the host cores of the same box.
"""
from __future__ import annotations

import argparse
import collections
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak


def make_workload(name: str):
    from raytracer_amd import scenes

    if name == "bunny":
        verts, idx = scenes.bunny_class_mesh()
        w, cam, sky = scenes.bunny(verts, idx)
        return dict(world=w, cam=cam, sky=sky, width=1920, height=1080, samples=1, bounces=2,
                    label="bunny-class stand-in mesh (81920 triangles, 1 instance), 1920x1080, samples=1, bounces=2, sky on")
    if name == "instanced":
        verts, idx = scenes.bunny_class_mesh()
        w, cam, sky = scenes.instanced_grid(100, seed=1, verts=verts, indices=idx)
        return dict(world=w, cam=cam, sky=sky, width=1920, height=1080, samples=2, bounces=8,
                    label="bunny-class mesh x 10000 instances under a TLAS, 1920x1080, samples=2, bounces=8, sky on")
    if name == "soup":
        w, cam, sky = scenes.soup_world(10_000_000, seed=1)
        return dict(world=w, cam=cam, sky=sky, width=1920, height=1080, samples=1, bounces=4,
                    label="10M-triangle soup (1 BLAS), 1920x1080, samples=1, bounces=4, sky on")
    if name == "heightfield":
        w, cam, sky = scenes.heightfield_world(2236, seed=1)
        return dict(world=w, cam=cam, sky=sky, width=1920, height=1080, samples=1, bounces=4,
                    label="10M-triangle displaced height field (2236^2 cells x 2 triangles, 1 indexed BLAS: a coherent surface), "
                          "1920x1080, samples=1, bounces=4, sky on")
    if name == "soup-rays":
        w, cam, sky = scenes.soup_world(10_000_000, seed=1)
        return dict(world=w, cam=cam, sky=sky, width=4096, height=4096, samples=1, bounces=1, rays=1 << 24,
                    label="10M-triangle soup (1 BLAS), 2^24 incoherent rays: origins uniform in the scene box, directions "
                          "uniform on the sphere (closest-hit queries through rthip_trace_rays_device)")
    if name == "instanced-rays":
        verts, idx = scenes.bunny_class_mesh()
        w, cam, sky = scenes.instanced_grid(100, seed=1, verts=verts, indices=idx)
        return dict(world=w, cam=cam, sky=sky, width=4096, height=1024, samples=1, bounces=1, rays=1 << 22,
                    ray_box=((-125.0, -1.0, -125.0), (125.0, 3.0, 125.0)),
                    label="bunny-class mesh x 10000 instances, 2^22 incoherent rays through the grid's box")
    if name == "cornell-mesh":
        verts, idx = scenes.bunny_class_mesh()
        w, cam, sky = scenes.cornell(extra_mesh=(verts, idx))
        return dict(world=w, cam=cam, sky=sky, width=3840, height=2160, samples=4, bounces=8,
                    label="Cornell box + bunny-class mesh (metal, x80) on the short box, 3840x2160, bounces=8")
    if name == "cornell":
        w, cam, sky = scenes.cornell()
        return dict(world=w, cam=cam, sky=sky, width=512, height=512, samples=2, bounces=8,
                    label="Cornell box demo, 512x512, samples=2, bounces=8")
    raise SystemExit(f"unknown workload {name}")


def build_times(tracer, torch, sizes):
    """LBVH build (triangle boxes -> Morton -> sort -> hierarchy+refit -> pre-order layout) with
    the vertices already in HBM, timed with events on the launch stream."""
    from raytracer_amd import abi

    out = {}
    stream = torch.cuda.current_stream()
    for n in sizes:
        g = torch.Generator(device="cuda")
        g.manual_seed(1)
        cen = torch.rand((n, 1, 3), device="cuda", generator=g)
        off = (torch.rand((n, 3, 3), device="cuda", generator=g) * 2 - 1) * 0.001
        p = (cen + off).reshape(3 * n, 3).float()
        px, py, pz = (p[:, k].contiguous() for k in range(3))
        nodes = torch.empty((2 * n - 1, 8), dtype=torch.float32, device="cuda")
        best = None
        for it in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            rc = tracer.lib.rthip_mesh_build_device(tracer.h, px.data_ptr(), py.data_ptr(), pz.data_ptr(), 3 * n, None,
                                                    n, nodes.data_ptr(), stream.cuda_stream)
            tracer.check(rc, "rthip_mesh_build_device")
            e1.record(stream)
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1)
            if it > 0:
                best = ms if best is None else min(best, ms)
        # size-independent validity: pre-order root covers everything, skip of the root == node count
        root = nodes[0].cpu().numpy().view(np.uint32)
        assert int(root[3]) == 2 * n - 1 and int(root[7]) == 0, "root record is malformed"
        out[str(n)] = dict(ms=round(best, 4), ms_per_1M=round(best * 1e6 / n, 4),
                           hbm_roofline_frac=round((444.0 * n / (best * 1e-3)) / (HBM_PEAK_GBS * 1e9), 4))
        del cen, off, p, px, py, pz, nodes
    return out


def camera_rays_torch(torch, cast, W, H):
    """Pixel-centre camera rays as device tensors (same construction as rt_cpu_raygen with samples = 1)."""
    c = cast[0]
    f = lambda k: torch.tensor(np.asarray(c[k], dtype=np.float32), device="cuda")
    xs = (torch.arange(W, device="cuda", dtype=torch.float32) + 0.5) / W
    ys = (torch.arange(H, device="cuda", dtype=torch.float32) + 0.5) / H
    ndc_x = (2 * xs - 1)[None, :].expand(H, W)
    ndc_y = (1 - 2 * ys)[:, None].expand(H, W)
    vp = f("viewport")
    sample = (f("eye")[None, None, :] + f("right")[None, None, :] * (ndc_x * vp[0])[..., None]
              + f("up")[None, None, :] * (ndc_y * vp[1])[..., None] + f("forward")[None, None, :] * vp[2])
    origin = f("eye")[None, None, :].expand(H, W, 3)
    d = sample - origin
    d = d / d.norm(dim=2, keepdim=True)
    return origin.reshape(-1, 3).contiguous(), d.reshape(-1, 3).contiguous()


def shadow_ray_rate(tr, torch, cast, W, H, steps):
    """Shadow rays of the bunny workload: from every primary hit point towards a directional light, answered by the
    any-hit kernel (rthip_occluded_device).  Reported beside the headline, never mixed into it."""
    stream = torch.cuda.current_stream()
    o, d = camera_rays_torch(torch, cast, W, H)
    hits = torch.zeros((W * H, 4), dtype=torch.int32, device="cuda")
    tr.check(tr.lib.rthip_trace_rays_device(tr.h, o.data_ptr(), d.data_ptr(), W * H, hits.data_ptr(), stream.cuda_stream), "trace")
    torch.cuda.synchronize()
    hit = hits[:, 0] != 0
    t = hits[:, 2].view(torch.float32)
    so = (o + d * (t[:, None] - 1e-3))[hit].contiguous()
    light = torch.tensor([0.5, 1.0, 0.3], device="cuda")
    sd = (light / light.norm())[None, :].expand(so.shape[0], 3).contiguous()
    n = so.shape[0]
    out = torch.zeros(n, dtype=torch.uint8, device="cuda")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for it in range(steps + 2):
        if it == 2:
            e0.record(stream)
        tr.check(tr.lib.rthip_occluded_device(tr.h, so.data_ptr(), sd.data_ptr(), n, 3.402823466e+38, out.data_ptr(),
                                              stream.cuda_stream), "rthip_occluded_device")
    e1.record(stream)
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / steps
    return dict(rays=int(n), kernel_ms=round(ms, 4), value=round(n / ms / 1e3, 2), unit="Mrays/s",
                occluded_fraction=round(float(out.float().mean().item()), 4),
                what="any-hit occlusion rays from the primary hit points towards a directional light, occlusion_kernel")


def hdr_output(tr, torch, cast, W, H, steps, want_cpu):
    """The image output step that follows every cast in the reference's programs (stbi_write_hdr,
    src/demos/cornell/main.c:222): rthip_hdr_encode_device on the frame in HBM, HIP-event time of its three kernels,
    against the reference's writer (or its CPU restatement) on the same frame on this host."""
    from raytracer_amd import abi

    stream = torch.cuda.current_stream()
    fb = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
    tr.check(tr.lib.rthip_cast_device(tr.h, abi.ptr(cast), 1, fb.data_ptr(), W, H, None, stream.cuda_stream), "rthip_cast_device")
    cap = int(tr.lib.rthip_hdr_bound(W, H))
    out = torch.zeros(cap, dtype=torch.uint8, device="cuda")
    size = torch.zeros(1, dtype=torch.int64, device="cuda")

    def encode():
        tr.check(tr.lib.rthip_hdr_encode_device(tr.h, fb.data_ptr(), W, H, out.data_ptr(), cap, size.data_ptr(),
                                                stream.cuda_stream), "rthip_hdr_encode_device")

    encode()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(stream)
    for _ in range(steps):
        encode()
    b.record(stream)
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / steps
    n = int(size.item())
    algo = 12 * W * H + n  # the float frame read once, the file written once
    res = {"what": "Radiance .hdr file bytes of the frame (RGBE + run-length coding), hdr_rows_kernel + hdr_offsets_kernel + "
                   "hdr_gather_kernel", "ms": round(ms, 4), "file_bytes": n, "bytes_per_pixel": round(n / (W * H), 3),
           "hbm_roofline_frac": round(algo / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
    if want_cpu:
        from oracle import oracle as O

        host = fb.cpu().numpy()
        use_ref = O.have_ref("hdr")
        t0 = time.perf_counter()
        want = O.ref_hdr_encode(host) if use_ref else O.hdr_encode(host)
        res["cpu_ms"] = round((time.perf_counter() - t0) * 1e3, 2)
        res["cpu_kind"] = "reference" if use_ref else "port"
        res["identical_to_cpu"] = bool(out[:n].cpu().numpy().tobytes() == want)
    return res


def obj_load_figure(tr, want_cpu):
    """Loading the workload's mesh from a Wavefront OBJ file (ms_load_obj, src/mesh/mesh.c:73-219; the bunny demo's first
    step, src/demos/bunny/main.c:41-49): rthip_obj_load (host threads parse, the GPU deduplicates) against the reference's
    loader (or its CPU restatement) on the same file on this host."""
    import tempfile

    from raytracer_amd import scenes, tracer as T

    verts, idx = scenes.bunny_class_mesh()
    path = os.path.join(tempfile.gettempdir(), f"rthip_bench_{os.getpid()}.obj")
    scenes.write_obj(path, verts, idx)
    try:
        best = None
        for _ in range(3):
            _, _, info = T.ms_load_obj(tr, path, 3, read_back=False)
            if best is None or info["total_ms"] < best["total_ms"]:
                best = info
        res = {"what": "rthip_obj_load of the workload's mesh as a PN .obj file (v / vn / f a//a)", "file_bytes": os.path.getsize(path),
               "triangles": best["index_count"] // 3, "vertices": best["vertex_count"], "ms": round(best["total_ms"], 3),
               "parse_ms": round(best["parse_ms"], 3), "gpu_dedup_ms": round(best["dedup_ms"], 4), "host_threads": best["host_threads"]}
        if want_cpu:
            from oracle import oracle as O

            gv, gi, _ = T.ms_load_obj(tr, path, 3)
            use_ref = O.have_ref("core")
            t0 = time.perf_counter()
            cv, ci = O.ref_load_obj(path, 3, gv.shape[0] + 8, gi.size + 8) if use_ref else O.load_obj(path, 3)
            res["cpu_ms"] = round((time.perf_counter() - t0) * 1e3, 2)
            res["cpu_kind"] = "reference" if use_ref else "port"
            cols = [0, 1, 2, 4, 5, 6]
            res["identical_to_cpu"] = bool((gi == ci).all() and gv[:, cols].tobytes() == cv[:, cols].tobytes())
        return res
    finally:
        os.unlink(path)


CSRC = os.path.join(ROOT, "raytracer_amd", "csrc")
# the sources a kernel family is compiled from: a recorded counter is only valid for the revision it was recorded on
RENDER_SOURCES = ("rt_device.h", "trace.hip", "wavefront.hip", "rthip_internal.h")
BUILD_SOURCES = ("lbvh_build.hip", "rthip_internal.h")


def blob_hash(path: str) -> str:
    """`git hash-object` of a file (sha1 over 'blob <size>\\0' + contents), without git."""
    import hashlib

    with open(path, "rb") as f:
        data = f.read()
    return hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()


def source_hashes() -> dict:
    return {f: blob_hash(os.path.join(CSRC, f)) for f in sorted(os.listdir(CSRC)) if f.endswith((".hip", ".h"))}


_RECORDED = {}


def recorded_table():
    """profiles/traffic.json (tools/pmc_traffic.py writes it from rocprofv3 --pmc passes of this same command) and the
    list of kernel sources that have changed since it was recorded."""
    if "table" not in _RECORDED:
        path = os.path.join(ROOT, "profiles", "traffic.json")
        table, changed = {}, None
        if os.path.exists(path):
            with open(path) as f:
                table = json.load(f)
            stamp = table.get("_recorded_on", {}).get("sources")
            now = source_hashes()
            changed = sorted(now) if not stamp else sorted(f for f in now if stamp.get(f) != now[f])
        _RECORDED["table"], _RECORDED["changed"] = table, changed
    return _RECORDED["table"], _RECORDED["changed"]


def recorded_counters(workload: str, kernel_name: str, sources=RENDER_SOURCES):
    """(HBM bytes per launch, VALU instructions per launch, provenance) of a kernel from the recorded PMC passes
    (FETCH_SIZE, WRITE_SIZE, SQ_INSTS_VALU: separate rocprofv3 runs).  The counters are constants of a revision, not
    measurements of this run: they are handed out only while the sources the kernel is compiled from have the hashes
    they were recorded on, otherwise (None, None, reason)."""
    table, changed = recorded_table()
    if changed is None:
        return None, None, "no counter pass recorded (profiles/traffic.json is absent)"
    rec = table.get(workload, {}).get(kernel_name.split(" ")[0])
    if rec is None:
        return None, None, f"no counter pass recorded for {workload} / {kernel_name.split(' ')[0]}"
    stale = [f for f in changed if f in sources]
    if stale:
        return None, None, "stale: recorded on another revision of " + ", ".join(stale)
    on = table.get("_recorded_on", {})
    return rec["hbm_bytes_per_launch"], rec.get("valu_insts_per_launch"), \
        f"recorded rocprofv3 --pmc passes of this command ({on.get('when', 'undated')}), sources unchanged since"


SIMDS = 256 * 4           # compute units x SIMDs
ENGINE_CLOCK_GHZ = 2.4    # peak engine clock of the MI355X


def valu_issue(insts, kernel_ms, ms_per_step, n_streams):
    """Share of the chip's vector-ALU issue cycles one launch's instructions take: a wave64 instruction holds its SIMD for
    4 cycles.  `issue_frac` over the launch's own duration when launches run one after the other, `issue_frac_overlapped`
    over the time a launch costs (timed region / steps) when several overlap."""
    if not insts:
        return None
    busy_ms = insts * 4.0 / (SIMDS * ENGINE_CLOCK_GHZ * 1e9) * 1e3
    out = {"insts_per_launch": int(insts), "cycles_per_inst": 4, "simds": SIMDS, "clock_ghz": ENGINE_CLOCK_GHZ}
    if n_streams > 1:
        out["issue_frac_overlapped"] = round(busy_ms / ms_per_step, 4)
    else:
        out["issue_frac"] = round(busy_ms / kernel_ms, 4)
    return out


def cpu_sample_frame(wl, max_paths=1 << 21):
    """The frame the CPU renders: the workload's own when it has at most ~2M paths, otherwise the same scene,
    camera and samples per pixel at a resolution reduced to that many paths (a bounded sample, tens of seconds)."""
    W, H, smp = wl["width"], wl["height"], wl["samples"]
    scale = 1
    while (W // scale) * (H // scale) * smp * smp > max_paths:
        scale += 1
    Ws, Hs = max(W // scale, 16), max(H // scale, 16)
    what = f"the full {W}x{H} frame of the same workload" if scale == 1 else \
        f"the same workload at {Ws}x{Hs} (1/{scale} of the resolution, same camera and samples per pixel)"
    return Ws, Hs, wl["cam"](Ws, Hs, smp), what


def cpu_baseline(wl, cast, flat, budget_s=20.0, max_paths=1 << 21):
    """The CPU path on this box's host cores: the unmodified reference when oracle/_ref travelled
    with the tree (kind 'reference'), otherwise the repo's CPU restatement (kind 'port')."""
    from oracle import oracle as O

    W, H, cast, what = cpu_sample_frame(wl, max_paths)
    ora = O.Scene(flat, wl["bounces"], wl["sky"])
    # the oracle in glibc-rand mode reproduces the reference's paths exactly, so its ray count is the reference's
    t0 = time.time()
    _, cnt = ora.cast(cast, 1, W, H, rng_mode=O.RNG_GLIBC)
    port_s = time.time() - t0
    rays = cnt["rays"]
    if O.have_ref("core"):
        rs = O.RefScene(flat, wl["bounces"], wl["sky"])
        reps, spent, best = 0, 0.0, None
        min_reps = 3 if budget_s >= 20.0 else 1
        while reps < min_reps or (spent < budget_s and reps < 10):
            _, sec = rs.cast(cast, 1, W, H)
            spent += sec
            best = sec if best is None else min(best, sec)
            reps += 1
            if spent > budget_s:
                break
        return dict(value=round(rays / best / 1e6, 3), unit="Mrays/s", cores=1, kind="reference",
                    sample=f"{what}, {reps} repeats, best of them; rays counted by the oracle run with the same glibc "
                           "rand() stream")
    return dict(value=round(rays / port_s / 1e6, 3), unit="Mrays/s", cores=1, kind="port",
                sample=f"{what}, one pass of the CPU restatement")


def cpu_baseline_rays(wl, flat, o, d, sample=1 << 16):
    """Ray-stream workloads: the reference's TLAS query on the first `sample` rays, one core."""
    from oracle import oracle as O

    if O.have_ref("core"):
        rs = O.RefScene(flat, wl["bounces"], wl["sky"])
        sec = rs.trace(o[:sample], d[:sample], want_normal=False)[5]
        kind = "reference"
    else:
        ora = O.Scene(flat, wl["bounces"], wl["sky"])
        t0 = time.time()
        ora.trace(o[:sample], d[:sample], want_normal=False)
        sec = time.time() - t0
        kind = "port"
    return dict(value=round(sample / sec / 1e6, 4), unit="Mrays/s", cores=1, kind=kind,
                sample=f"the first {sample} rays of the same ray stream, one pass")


def cpu_port_parallel(wl, cast, flat, threads):
    """The CPU restatement row-parallel over the host cores (the reference itself is single-threaded)."""
    from oracle import oracle as O

    W, H, cast, what = cpu_sample_frame(wl)
    ora = O.Scene(flat, wl["bounces"], wl["sky"])
    t0 = time.time()
    _, cnt = ora.cast(cast, 1, W, H, rng_mode=O.RNG_COUNTER, threads=threads)
    sec = time.time() - t0
    return dict(value=round(cnt["rays"] / sec / 1e6, 3), unit="Mrays/s", cores=threads, kind="port",
                sample=f"{what}, one pass, rows dealt to {threads} threads")


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="bunny")
    ap.add_argument("--samples", type=int, default=0, help="samples per pixel axis (spp = samples^2); 0 = the workload's own")
    ap.add_argument("--shard", default="frames", choices=["frames", "tiles", "samples"],
                    help="N > 1: every rank renders its own frame (weak scaling), or the ranks split one frame by "
                         "64x64 tiles or by sample index (strong scaling)")
    ap.add_argument("--combine", default="auto", choices=["auto", "end", "every"],
                    help="when the ranks' framebuffers are combined: 'every' step (a frame is delivered per step; the reduce "
                         "of frame k overlaps the render of frame k + 1) or once at the 'end' of the timed steps (progressive "
                         "accumulation: every step adds its own sample set, one exchange per image).  auto: end for "
                         "--shard frames, every for tiles / samples")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend for N > 1: nccl = RCCL over xGMI (the product path); gloo = rehearsal "
                         "of the N > 1 code on a box whose ranks share one GPU (framebuffers staged through the host)")
    ap.add_argument("--streams", type=int, default=4,
                    help="HIP streams the steps are dealt to in turn (each with its own framebuffer and accumulator, summed "
                         "inside the timed region): a frame's first waves run while the previous frames' last waves drain.  "
                         "1 = strictly one launch after the other.  Only with --combine end (progressive accumulation) and "
                         "for ray streams; --combine every delivers frames in order on one stream")
    ap.add_argument("--traversal", default="exact", choices=["exact", "fast"],
                    help="exact: the reference's fixed left-before-right walk (bit-identical results, the headline); fast: "
                         "the opt-in ordered walk (rthip_set_traversal)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-workloads", action="store_true",
                    help="default workload at N = 1 only: skip the `workloads` object (the other BASELINE configurations, the "
                         "one-stream figure and the host-buffer rthip_cast figure, each a short run of its own)")
    ap.add_argument("--width", type=int, default=0, help="frame width instead of the workload's own (rehearsals and tests)")
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--launch-check", action="store_true",
                    help="N ranks only join the process group, add up their ranks and rank 0 prints one JSON line: a test of "
                         "the launch path that needs no GPU with --backend gloo")
    ap.add_argument("--build-sizes", default="81920,1000000,10000000")
    ap.add_argument("--min-region-s", type=float, default=0.5,
                    help="the timed region of --steps steps is repeated until this many seconds of timed work have been done and "
                         "the median region is reported; 0 = one region (counter passes under rocprofv3)")
    ap.add_argument("--multi", default="ranks", choices=["ranks", "group"],
                    help="N > 1: 'ranks' = one process per GPU, torch.distributed over RCCL (what a launcher starts; the default); "
                         "'group' = ONE process drives all N GPUs through rthip_group_cast_device (one context and one host "
                         "thread per GPU, the shares exchanged over RCCL inside the library, peer copies without it): needs "
                         "no launcher and no torch.distributed")
    ap.add_argument("--group-devices", default="",
                    help="--multi group: the device ids, comma separated (default 0..N-1; an id may repeat, which is how the "
                         "path is rehearsed on a one-GPU box)")
    ap.add_argument("--c5-samples", type=int, default=32,
                    help="samples per pixel axis of the `cornell_mesh_4k_*` strong-scaling entries (BASELINE config 5: 32 = 1024 spp)")
    ap.add_argument("--no-config5", action="store_true", help="skip the config-5 strong-scaling entries of a multi-GPU run")
    ap.add_argument("--config5-limit-s", type=float, default=300.0,
                    help="N > 1: if the config-5 entries have not finished after this many seconds, rank 0 prints the line without them")
    return ap.parse_args(argv)


def self_launch(args, argv):
    """`bench.py --gpus N` started without a launcher: this process starts the N ranks itself (one process per GPU,
    torch.distributed.run on 127.0.0.1) before it has touched the GPU, lets rank 0's JSON line through and returns the
    launcher's exit code."""
    import socket
    import subprocess

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *argv]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # the host driver only supports dmabuf IPC (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", "4")
    return subprocess.run(cmd, env=env, cwd=ROOT).returncode


def launch_check(args, rank, world):
    import torch
    import torch.distributed as dist

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if world > 1:
        if args.backend == "nccl":
            device_index = int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1)
            torch.cuda.set_device(device_index)
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", device_index))
            t = torch.tensor([float(rank)], device="cuda")
        else:
            dist.init_process_group(backend="gloo")
            t = torch.tensor([float(rank)])
        dist.all_reduce(t)
        dist.barrier()
        dist.destroy_process_group()
    else:
        t = torch.tensor([0.0])
    if rank == 0:
        print(json.dumps({"launch_check": True, "n_gpus": world, "backend": args.backend, "rank_sum": float(t.item())}), flush=True)


def ceilings(algo_bytes, traffic, valu_insts, kernel_ms, ms_per_step, n_streams, why_no_counters=None):
    """What bounds a launch.  `cost_ms` is the time one launch costs on the chip: its own duration when launches run one
    after the other, the timed region / steps when several overlap.  HBM is the bounding ceiling only when the measured
    traffic is of the order of the algorithmic bytes (the scene does not fit the caches); cache-resident scenes are priced
    against vector-ALU issue instead, and an HBM fraction is not printed for more bytes than HBM moved."""
    cost_ms = ms_per_step if n_streams > 1 else kernel_ms
    algo_frac = algo_bytes / (cost_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
    out = {"cost_ms_per_launch": round(cost_ms, 4)}
    if algo_frac <= 1.0:
        out["hbm_algorithmic_frac"] = round(algo_frac, 4)
    else:  # more algorithmic bytes per second than HBM can move: the scene is served by the caches
        out["hbm_algorithmic_frac"] = None
        out["hbm_algorithmic_note"] = "cache-resident: the algorithmic bytes per second exceed the HBM peak, so this is no HBM statement"
    measured = None
    if traffic:
        measured = traffic / (cost_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
        if measured > 1.0:  # FETCH_SIZE / WRITE_SIZE count at the L2's far side: bytes served by the Infinity Cache are among them
            out["hbm_measured_note"] = f"{measured:.2f} by bytes counted behind the L2, Infinity Cache hits included: capped at 1"
            measured = 1.0
        out["hbm_measured_frac"] = round(measured, 4)
        out["traffic_over_algorithmic"] = round(traffic / max(algo_bytes, 1), 3)
    issue = None
    if valu_insts:
        # 4 cycles per wave64 instruction is the most an instruction takes: one whose active lanes all lie in one half of
        # the wave is done in 2, so with few lanes at work the estimate overshoots -- it is capped at 1 and says so
        issue = valu_insts * 4.0 / (SIMDS * ENGINE_CLOCK_GHZ * 1e9) * 1e3 / cost_ms
        if issue > 1.0:
            out["valu_issue_note"] = (f"{issue:.2f} by instructions x 4 cycles: an upper estimate (an instruction with one "
                                      "half of the wave idle takes 2 cycles), capped at 1")
            issue = 1.0
        out["valu_issue_frac"] = round(issue, 4)
    if measured is not None and traffic >= 0.7 * algo_bytes:
        out["bounding"] = "hbm"
        out["frac_of_bounding"] = round(measured, 4)
    elif issue is not None:
        out["bounding"] = "valu-issue"
        out["frac_of_bounding"] = round(issue, 4)
    elif measured is not None:
        out["bounding"] = "caches (L2 / Infinity Cache) and valu issue: HBM moved a fraction of the algorithmic bytes; no SQ_INSTS_VALU pass recorded"
        out["frac_of_bounding"] = None
    else:
        out["bounding"] = "unknown: " + (why_no_counters or "no counter pass recorded for this workload (profiles/traffic.json)")
        out["frac_of_bounding"] = None
    return out


_WORKLOAD_CACHE = {}


def cached_workload(name):
    if name not in _WORKLOAD_CACHE:
        key = {"soup-rays": "soup", "instanced-rays": "instanced"}.get(name)
        wl = make_workload(name)
        if key and key in _WORKLOAD_CACHE:  # the same scene: keep one copy of the world
            wl["world"] = _WORKLOAD_CACHE[key]["world"]
        _WORKLOAD_CACHE[name] = wl
    return dict(_WORKLOAD_CACHE[name])


def measure(args, name, steps, warmup, n_streams_arg, rank, world, device_index, want_cpu, cpu_budget_s=20.0, cpu_max_paths=1 << 21,
            extras=True, traversal=None, min_region_s=0.5, max_regions=64):
    """One workload: scene build, counted launches, warm-up, the timed steps, and the figures derived from them."""
    import torch
    import torch.distributed as dist

    from raytracer_amd import abi, dist as rdist, scenes, tracer as T, world as Wd

    wl = cached_workload(name)
    if args.samples > 0:
        wl["samples"] = args.samples
    if args.width > 0 and args.height > 0:
        wl["width"], wl["height"] = args.width, args.height
        wl["label"] += f" [frame overridden: {args.width}x{args.height}]"
    wl["label"] += f", samples={wl['samples']}" if "samples=" not in wl["label"] else ""
    W, H = wl["width"], wl["height"]
    tr = T.rt_make_tracer(T.RT_TracerSettings(max_bounces=wl["bounces"], sky=wl["sky"]), device_id=device_index)
    traversal = traversal or args.traversal
    if traversal == "fast":
        T.set_traversal(tr, "fast")
        wl["label"] += " [RTHIP_TRAV_FAST: ordered walk, results equal up to ties and seam rays]"
    T.rt_tracer_build_blas(tr, wl["world"])
    blas_first_ms = T.get_stats(tr)["last_build_ms"]  # into freshly allocated memory: first-touch cost included
    T.rt_tracer_build_blas(tr, wl["world"])
    blas_ms = T.get_stats(tr)["last_build_ms"]
    T.rt_tracer_build_tlas(tr, wl["world"])
    cast = wl["cam"](W, H, wl["samples"])
    flat = Wd.flatten_world(wl["world"])
    # animation = rt_tracer_build_tlas again (raytracer_cpu.c:32-37): the whole call from flat host records (upload of
    # instances and materials, instance boxes, LBVH), and the device time of the LBVH inside it
    tlas_call_ms, tlas_ms = 1e9, 0.0
    for _ in range(5):
        t0 = time.perf_counter()
        T.build_tlas_flat(tr, flat.instances, flat.materials)
        tlas_call_ms = min(tlas_call_ms, (time.perf_counter() - t0) * 1e3)
        tlas_ms = T.get_stats(tr)["last_build_ms"]

    combine = args.combine if args.combine != "auto" else ("end" if args.shard == "frames" else "every")
    if combine == "end" and args.shard != "frames":
        raise SystemExit("--combine end accumulates independent sample sets: it needs --shard frames")
    is_rays = "rays" in wl
    n_streams = max(1, n_streams_arg) if (combine == "end" or is_rays) else 1
    main_stream = torch.cuda.current_stream()
    if n_streams > 1:
        # HIP binds a stream to one of its hardware queues (4 on this box, GPU_MAX_HW_QUEUES) when the stream is first
        # used, and the first streams a process uses land beside the default stream and the library's own: measured on
        # this workload with 4 working streams, 0.215 ms per frame when they are the first streams used, 0.135 ms once
        # two or more other streams have been used before them.  So four streams are used once and set aside, and the
        # working streams (a multiple of 4 spreads evenly; 6 measured worse than 4) come after them.
        aside = [torch.cuda.Stream() for _ in range(4)]
        for u in aside:
            with torch.cuda.stream(u):
                torch.zeros(16, device="cuda").add_(1.0)
        torch.cuda.synchronize()
    streams = [main_stream] if n_streams == 1 else [torch.cuda.Stream() for _ in range(n_streams)]
    accum_all = torch.zeros((len(streams), H, W, 3), dtype=torch.float32, device="cuda") if combine == "end" else None
    accums = [accum_all[k] for k in range(len(streams))] if combine == "end" else None
    accum_sum = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda") if (combine == "end" and n_streams > 1) else None
    accum = (accum_sum if n_streams > 1 else accums[0]) if accums else None
    step_no = [0]
    # --combine every: two framebuffers, the RCCL reduce of frame k overlaps the render of frame k + 1
    pipe = rdist.FramePipeline([torch.zeros((H, W, 3), dtype=torch.float32, device="cuda") for _ in range(2)],
                               args.shard, wl["samples"], world, dst=0, tracer=tr)
    seed = 1 + (rank if args.shard == "frames" else 0)
    shard = rdist.shard_record(args.shard, rank, world)

    ray_o = ray_d = ray_hits = None
    if is_rays:  # a ray-stream workload: closest-hit queries for explicit rays instead of a frame
        lo, hi = wl.get("ray_box", ((0.0, 0.0, 0.0), (1.0, 1.0, 1.0)))
        ray_o, ray_d = scenes.incoherent_rays_device(torch, wl["rays"], 2 + rank, lo, hi)
        ray_hits = [torch.zeros((wl["rays"], 4), dtype=torch.int32, device="cuda") for _ in streams]
    torch.cuda.synchronize()

    # A step = one launch on EACH of the streams (a batch of `n_streams` frames in flight together; one frame with
    # --streams 1): the pipeline's fill and drain -- about one frame's latency per timed region, whatever its length --
    # then weigh the same few per cent at the driver's 20 steps as in a long run.
    frames_per_step = n_streams

    # The library refuses a 65th unfinished launch of its persistent kernels: the host keeps at most `depth` steps
    # enqueued ahead of the device (it waits for the step that far back, while the device still has the later ones).
    depth = max(2, 40 // frames_per_step)
    enqueued = collections.deque()

    def step(events=None):
        for j in range(frames_per_step):
            launch(events[j] if events else None)
        marks = []
        for s in streams:
            marks.append(torch.cuda.Event())
            marks[-1].record(s)
        enqueued.append(marks)
        if len(enqueued) > depth:
            for m in enqueued.popleft():
                m.synchronize()

    def launch(events=None):
        k = step_no[0] % n_streams
        stream = streams[k]
        if ray_o is not None:
            step_no[0] += 1
            if events:
                events[0].record(stream)
            rc = tr.lib.rthip_trace_rays_device(tr.h, ray_o.data_ptr(), ray_d.data_ptr(), wl["rays"], ray_hits[k].data_ptr(),
                                                stream.cuda_stream)
            if events:
                events[1].record(stream)
            tr.check(rc, "rthip_trace_rays_device")
            return
        # progressive accumulation: every step of every rank renders its own sample set and adds it to the stream's
        # accumulator (rthip_cast_accumulate_device: no pass of its own for the addition)
        step_seed = seed if combine == "every" else 1 + step_no[0] * world + rank
        step_no[0] += 1
        if combine == "end":
            if events:
                events[0].record(stream)
            rc = tr.lib.rthip_cast_accumulate_device(tr.h, abi.ptr(cast), step_seed, accums[k].data_ptr(), W, H, abi.ptr(shard),
                                                     stream.cuda_stream)
            if events:
                events[1].record(stream)
            tr.check(rc, "rthip_cast_accumulate_device")
            return
        fb = pipe.acquire()
        if events:
            events[0].record(stream)
        rc = tr.lib.rthip_cast_device(tr.h, abi.ptr(cast), step_seed, fb.data_ptr(), W, H, abi.ptr(shard), stream.cuda_stream)
        if events:
            events[1].record(stream)
        tr.check(rc, "rthip_cast_device")
        pipe.submit(fb)

    def finish_image(n_steps):
        """--combine end: the streams' accumulators are added up and the image is exchanged once, inside the timed region."""
        for s in streams[0 if n_streams > 1 else 1:]:
            main_stream.wait_stream(s)
        if combine != "end" or ray_o is not None:
            return
        if n_streams > 1:
            torch.sum(accum_all, dim=0, out=accum_sum)  # one pass over the streams' accumulators
        if world > 1:
            if dist.get_backend() == "gloo":  # rehearsal: gloo has no GPU reduce
                host = accum.cpu()
                dist.reduce(host, dst=0, op=dist.ReduceOp.SUM)
                accum.copy_(host)
            else:
                dist.reduce(accum, dst=0, op=dist.ReduceOp.SUM)
        if rank == 0:
            accum.mul_(1.0 / float(n_steps * frames_per_step * world))

    # Traversal counters of this rank's frames: untimed launches with counting on, one per seed of the first timed steps
    # (at most 4; the steps of a progressive image differ only in their sample sets, so their ray counts differ by a
    # fraction of a per cent -- the mean over the counted seeds stands for every step; a 1-bounce-deep frame and a ray
    # stream have the same count for every seed).
    counted = 1 if (is_rays or combine == "every") else max(1, min(4, steps))
    T.set_counting(tr, True)
    T.get_stats(tr)
    for _ in range(counted):
        launch()
    pipe.drain()
    torch.cuda.synchronize()
    st = T.get_stats(tr)
    T.set_counting(tr, False)
    step_no[0] = 0
    st = {k: (v / counted if k in ("rays", "nodes_tested", "tris_tested", "inst_entered", "wave_steps", "lane_steps") else v) for k, v in st.items()}
    rays = st["rays"]
    algo_bytes = 48 * st["rays"] + 32 * st["nodes_tested"] + 48 * st["tris_tested"] + 64 * st["inst_entered"]

    for _ in range(warmup):
        step()
    pipe.drain()
    finish_image(warmup + 1)

    def all_ranks(value, op):
        """One number per rank -> the MAX / SUM over ranks (the same on every rank)."""
        if world == 1:
            return float(value)
        t = torch.tensor([float(value)], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else "cuda")
        dist.all_reduce(t, op=op)
        return float(t.item())

    def timed_region():
        """EXACTLY `steps` steps, bracketed by a barrier and a device synchronisation on both sides; the image's exchange
        (--combine end) lies inside.  Returns (seconds: MAX over ranks, mean launch duration in ms, host enqueue ms)."""
        if accums is not None:
            accum_all.zero_()
        step_no[0] = 0
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps * frames_per_step)]
        t0 = time.perf_counter()
        for k in range(steps):
            step(ev[k * frames_per_step:(k + 1) * frames_per_step])
        enqueue_ms = (time.perf_counter() - t0) * 1e3 / steps
        pipe.drain()
        finish_image(steps)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        seconds = time.perf_counter() - t0
        launch_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
        return all_ranks(seconds, dist.ReduceOp.MAX if world > 1 else None), launch_ms, enqueue_ms

    # A region of `steps` steps can be a few milliseconds (the default frame: 20 steps = 3 ms), which is shorter than the
    # clocks take to settle and than one scheduling hiccup of the host.  The region is therefore run again and again, each
    # time bracketed as above, until about `min_region_s` of timed work has been done, and the MEDIAN region is the one
    # reported (the first region decides how many: the same number on every rank, since its time is the MAX over ranks).
    regions = [timed_region()]
    repeats = int(min(max_regions, max(1, np.ceil(min_region_s / max(regions[0][0], 1e-6))))) if min_region_s > 0 else 1
    for _ in range(repeats - 1):
        regions.append(timed_region())
    order = sorted(range(len(regions)), key=lambda k: regions[k][0])
    elapsed, kernel_ms, host_enqueue_ms = regions[order[(len(order) - 1) // 2]]  # the median region (the lower one of two)
    region_ms = [r[0] * 1e3 / steps for r in regions]
    total_rays = all_ranks(rays, dist.ReduceOp.SUM if world > 1 else None)

    last = T.get_stats(tr)
    kernel_name = {0: f"cast_kernel (persistent, refill {last['last_cast_refill']})", 1: "cast_static_kernel", 4: "cast_samples_kernel",
                   2: "wf_trace_kernel + wf_shade_kernel (wavefront)"}[last["last_cast_kernel"]]
    if ray_o is not None:
        kernel_name = "trace_rays_kernel"
    # (the recorded passes ran the workloads at their own samples per pixel and frame size: nothing is quoted for others)
    table_key = name + ("+fast" if traversal == "fast" else "") + (f"@samples={args.samples}" if args.samples > 0 else "") + \
        (f"@{args.width}x{args.height}" if args.width > 0 and args.height > 0 else "")
    traffic, valu_insts, counters_from = recorded_counters(table_key, kernel_name)
    if ray_o is not None and traffic is None:  # deep scenes: the library takes the ray stream's persistent kernel
        t2, v2, from2 = recorded_counters(table_key, "trace_rays_persistent_kernel")
        if t2 is not None:
            traffic, valu_insts, counters_from = t2, v2, from2
            kernel_name = "trace_rays_persistent_kernel"
    line = None
    if rank == 0:
        value = total_rays * steps * frames_per_step / elapsed / 1e6
        achieved = algo_bytes / (kernel_ms * 1e-3) / 1e9
        ms_per_step = elapsed / steps * 1e3
        ms_per_launch = ms_per_step / frames_per_step  # what one frame (or ray stream) costs with the others beside it
        overlapped = algo_bytes * steps * frames_per_step / elapsed / 1e9
        bound = ceilings(algo_bytes, traffic, valu_insts, kernel_ms, ms_per_launch, n_streams, counters_from)
        line = {
            "metric": "Mrays/sec (whole node); LBVH build ms per 1M triangles",
            "value": round(value, 2),
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": steps,
            "warmup": warmup,
            "ms_per_step": round(ms_per_step, 4),
            # `value` / `ms_per_step` are the MEDIAN of `count` timed regions of exactly `steps` steps each (see timed_region)
            "regions": {"count": len(regions), "ms_per_step_min": round(min(region_ms), 4), "ms_per_step_median": round(ms_per_step, 4),
                        "ms_per_step_max": round(max(region_ms), 4), "first_region_ms_per_step": round(region_ms[0], 4)},
            "host_enqueue_ms_per_step": round(host_enqueue_ms, 4),
            "higher_is_better": True,
            "scaling": "weak" if args.shard == "frames" else "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "ms_per_frame": round(ms_per_launch, 4),
            "config": {"workload": wl["label"], "rays_per_frame_per_gpu": int(round(rays)), "streams": n_streams,
                       # a step = one launch on each stream: `frames_per_step` frames (ray streams) in flight together
                       "frames_per_step": frames_per_step,
                       "rays_counted_over_seeds": counted,
                       "sharding": "none" if world == 1 else {
                           "frames": "every rank renders the frame with its own sample set; " + (
                               "framebuffers summed onto rank 0 with RCCL every step" if combine == "every" else
                               "every step adds its sample set to the rank's accumulator, ONE RCCL sum onto rank 0 at the end "
                               "of the timed steps (inside the timed region)"),
                           "tiles": "ranks render interleaved 64x64 tiles of one frame; every rank's own tiles (1/N of the frame) gathered onto rank 0 with RCCL, nothing is added",
                           "samples": "ranks render sample indices s % N == rank of one frame; sums added onto rank 0 with RCCL"}[args.shard]},
            # `achieved` / `frac`: ALGORITHMIC bytes of one launch over the mean duration of one launch (HIP events on its
            # stream; with --streams > 1 other launches run beside it, as rocprofv3 reports it).  Whether HBM is the ceiling
            # that binds is said by `bounding` (see ceilings()): `traffic` far below the algorithmic bytes = cache-resident.
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": counters_from, "kernel": kernel_name,
                         "kernel_ms": round(kernel_ms, 4), "algorithmic_bytes_per_launch": int(algo_bytes),
                         "per_ray": {"nodes": round(st["nodes_tested"] / max(rays, 1), 2),
                                     "triangles": round(st["tris_tested"] / max(rays, 1), 3),
                                     "instances": round(st["inst_entered"] / max(rays, 1), 3)},
                         # share of the lanes at work in a step of the walk (lane steps / 64 x wave steps, from the kernel's
                         # own counters; the same meaning in both traversal modes, never above 1)
                         "lane_utilisation": round(st["lane_steps"] / max(64 * st["wave_steps"], 1), 4),
                         "box_tests_per_lane_step": round(st["nodes_tested"] / max(st["lane_steps"], 1), 3),
                         "concurrent_launches": n_streams,
                         "algorithmic_rate_overlapped_gbs": round(overlapped, 2),
                         # a fraction of the HBM peak only where that is a statement about HBM: never above 1
                         "frac_overlapped": round(overlapped / HBM_PEAK_GBS, 5) if overlapped <= HBM_PEAK_GBS else None,
                         "ceilings": bound, "bounding": bound["bounding"], "frac_of_bounding": bound["frac_of_bounding"],
                         "valu": valu_issue(valu_insts, kernel_ms, ms_per_launch, n_streams)},
            "scene_build_ms": {"blas": round(blas_ms, 4), "blas_first_call": round(blas_first_ms, 4), "tlas": round(tlas_ms, 4), "tlas_rebuild_call": round(tlas_call_ms, 4),
                               "instances": int(flat.instances.size)},
        }
        if extras:
            if args.build_sizes:
                line["lbvh_build"] = build_times(tr, torch, [int(s) for s in args.build_sizes.split(",") if s])
                big = max(line["lbvh_build"], key=lambda k: int(k))
                line["lbvh_build_ms_per_1M_triangles"] = line["lbvh_build"][big]["ms_per_1M"]
            if name == "bunny":
                line["shadow_rays"] = shadow_ray_rate(tr, torch, cast, W, H, steps)
                line["obj_load"] = obj_load_figure(tr, want_cpu)
            if ray_o is None:
                line["hdr_output"] = hdr_output(tr, torch, cast, W, H, steps, want_cpu)
                line["host_cast"] = host_cast_figure(tr, cast, W, H, rays)
        if want_cpu and ray_o is not None:
            line["cpu_baseline"] = cpu_baseline_rays(wl, flat, ray_o[: 1 << 16].cpu().numpy(), ray_d[: 1 << 16].cpu().numpy(),
                                                     sample=(1 << 16) if cpu_budget_s >= 20 else (1 << 14))
        if want_cpu and ray_o is None:
            line["cpu_baseline"] = cpu_baseline(wl, cast, flat, budget_s=cpu_budget_s, max_paths=cpu_max_paths)
            if extras:
                threads = min(16, os.cpu_count() or 1)  # the box's CPU share for one GPU
                line["cpu_port_all_cores"] = cpu_port_parallel(wl, cast, flat, threads)
    T.rt_tracer_cleanup(tr)
    del accums, accum, accum_all, accum_sum, pipe, ray_o, ray_d, ray_hits
    torch.cuda.empty_cache()
    return line


def host_cast_figure(tr, cast, W, H, rays, reps=8):
    """rt_tracer_cast as the reference's callers use it: rthip_cast renders and copies the frame into the caller's
    ordinary (pageable) host buffer, one frame after the other.  The PCIe-inclusive figure; never the headline."""
    from raytracer_amd import tracer as T

    img = np.zeros((H, W, 3), dtype=np.float32)
    times = []
    for k in range(reps + 2):
        t0 = time.perf_counter()
        T.rt_tracer_cast(tr, cast, img, W, H, seed=1 + k)
        times.append((time.perf_counter() - t0) * 1e3)
    ms = float(np.median(times[2:]))
    pimg = T.host_alloc(tr, (H, W, 3))  # the same into page-locked memory from rthip_host_alloc: DMA at the link's rate
    pinned = []
    for k in range(reps + 2):
        t0 = time.perf_counter()
        T.rt_tracer_cast(tr, cast, pimg, W, H, seed=1 + k)
        pinned.append((time.perf_counter() - t0) * 1e3)
    same = bool(pimg.tobytes() == img.tobytes())
    T.host_free(tr, pimg)
    del pimg
    pms = float(np.median(pinned[2:]))
    return {"what": "rthip_cast into the caller's pageable host buffer (render + device-to-host copy of the frame), one call "
                    "after the other: the rate a drop-in rt_tracer_cast delivers", "ms_per_call": round(ms, 4),
            "value": round(rays / ms / 1e3, 2), "unit": "Mrays/s", "frame_bytes": int(W) * int(H) * 12,
            "pinned": {"what": "the same into a page-locked buffer from rthip_host_alloc", "ms_per_call": round(pms, 4),
                       "value": round(rays / pms / 1e3, 2), "unit": "Mrays/s", "same_bits": same}}


# The other BASELINE configurations and views of the default one, each a short run of its own inside the default
# `python bench.py` so that the driver's record holds them: (key, workload, streams, steps, warm-up, walk); a step is one
# launch on each of the streams
EXTRA_RUNS = [
    ("bunny_one_stream", "bunny", 1, 40, 5, "exact"),
    ("instanced", "instanced", 4, 3, 1, "exact"),
    ("instanced_one_stream", "instanced", 1, 6, 1, "exact"),
    ("soup", "soup", 4, 3, 1, "exact"),
    ("soup_rays", "soup-rays", 1, 3, 1, "exact"),
    ("heightfield", "heightfield", 4, 3, 1, "exact"),
    ("instanced_rays", "instanced-rays", 4, 2, 1, "exact"),
    ("cornell", "cornell", 4, 10, 2, "exact"),
    ("cornell_mesh_4k", "cornell-mesh", 1, 4, 1, "exact"),
    # BASELINE config 5's frame itself (1024 spp: --c5-samples 32), one step: the N = 1 end of its strong-scaling curve
    ("cornell_mesh_4k_1024spp", "cornell-mesh", 1, 1, 0, "exact"),
    # the opt-in ordered walk (rthip_set_traversal(RTHIP_TRAV_FAST)) on the deep scenes
    ("bunny_fast", "bunny", 4, 10, 2, "fast"),
    ("instanced_fast", "instanced", 4, 3, 1, "fast"),
    ("instanced_rays_fast", "instanced-rays", 4, 2, 1, "fast"),
    ("soup_fast", "soup", 4, 3, 1, "fast"),
    ("soup_rays_fast", "soup-rays", 1, 3, 1, "fast"),
    ("heightfield_fast", "heightfield", 4, 3, 1, "fast"),
]


def config5_entries(args, rank, world, device_index):
    """One frame of the Cornell + mesh scene at --c5-samples^2 spp, sharded by tiles and by samples over the ranks."""
    import copy

    out = {}
    for shard in ("tiles", "samples"):
        a = copy.copy(args)
        a.shard, a.combine, a.samples = shard, "every", args.c5_samples
        sub = measure(a, "cornell-mesh", 1, 0, 1, rank, world, device_index, False, extras=False, min_region_s=0.0)
        if rank == 0:
            c = compact(sub)
            c["scaling"], c["n_gpus"], c["sharding"] = "strong", world, sub["config"]["sharding"]
            out[f"cornell_mesh_4k_{args.c5_samples * args.c5_samples}spp_{shard}"] = c
    return out


def group_main(args):
    """`--multi group`: the whole node from ONE process.  A group of contexts (rthip_group_*, one per GPU, one host thread
    each) renders ONE frame per step, split by 64x64 tiles (--shard tiles, also what `frames` means here) or by sample
    index (--shard samples), and the shares meet on the first device inside the library -- over RCCL when the devices are
    distinct and the RCCL library loads, by peer copies otherwise.  No launcher, no torch.distributed: a box on which
    those fail still yields a number.  Strong scaling; `value` = rays of the frame / time of a call."""
    import torch

    from raytracer_amd import tracer as T

    ids = [int(x) for x in args.group_devices.split(",") if x] or list(range(args.gpus))
    if len(ids) != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but --group-devices names {len(ids)} devices")
    torch.cuda.set_device(ids[0])

    def run(name, shard, samples, steps, warmup):
        wl = cached_workload(name)
        if samples > 0:
            wl["samples"] = samples
        if args.width > 0 and args.height > 0:
            wl["width"], wl["height"] = args.width, args.height
            wl["label"] += f" [frame overridden: {args.width}x{args.height}]"
        W, H = wl["width"], wl["height"]
        if shard == "samples" and wl["samples"] < 2:
            raise SystemExit("--shard samples needs a workload with samples >= 2")
        g = T.RT_TracerGroup(ids, T.RT_TracerSettings(max_bounces=wl["bounces"], sky=wl["sky"]), shard=shard)
        if args.traversal == "fast":
            g.check(g.lib.rthip_group_set_traversal(g.g, T.TRAVERSALS["fast"]), "rthip_group_set_traversal")
        t0 = time.perf_counter()
        g.build_blas(wl["world"])
        g.build_tlas(wl["world"])
        build_ms = (time.perf_counter() - t0) * 1e3
        cast = wl["cam"](W, H, wl["samples"])
        fb = torch.zeros((H, W, 3), dtype=torch.float32, device=f"cuda:{ids[0]}")
        ctxs = [g.ctx(k) for k in range(len(ids))]
        for c in ctxs:
            T.set_counting(c, True)
            T.get_stats(c)
        g.cast_device(cast, fb.data_ptr(), W, H, seed=1)
        st = [T.get_stats(c) for c in ctxs]
        for c in ctxs:
            T.set_counting(c, False)
        tot = {k: sum(s[k] for s in st) for k in ("rays", "nodes_tested", "tris_tested", "inst_entered", "wave_steps", "lane_steps")}
        algo = 48 * tot["rays"] + 32 * tot["nodes_tested"] + 48 * tot["tris_tested"] + 64 * tot["inst_entered"]
        for k in range(warmup):
            g.cast_device(cast, fb.data_ptr(), W, H, seed=1)

        def region():
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for k in range(steps):
                g.cast_device(cast, fb.data_ptr(), W, H, seed=1)  # returns when the frame is whole on the first device
            torch.cuda.synchronize()
            return time.perf_counter() - t0

        regions = [region()]
        repeats = int(min(16, max(1, np.ceil(0.5 / max(regions[0], 1e-6)))))
        regions += [region() for _ in range(repeats - 1)]
        elapsed = sorted(regions)[(len(regions) - 1) // 2]
        ms = elapsed / steps * 1e3
        transport = g.transport()
        g.close()
        per_gpu = algo / len(ids) / (ms * 1e-3) / 1e9  # every device walks its share while the others walk theirs
        return {"workload": wl["label"] + f", samples={wl['samples']}", "value": round(tot["rays"] / ms / 1e3, 2), "unit": "Mrays/s",
                "ms_per_step": round(ms, 4), "steps": steps,
                "regions": {"count": len(regions), "ms_per_step_min": round(min(regions) / steps * 1e3, 4),
                            "ms_per_step_max": round(max(regions) / steps * 1e3, 4)},
                "rays_per_frame": int(tot["rays"]), "shard": shard, "transport": transport, "devices": ids,
                "scene_build_ms_all_devices": round(build_ms, 2), "algorithmic_bytes_per_frame": int(algo),
                "per_ray": {"nodes": round(tot["nodes_tested"] / max(tot["rays"], 1), 2), "triangles": round(tot["tris_tested"] / max(tot["rays"], 1), 3),
                            "instances": round(tot["inst_entered"] / max(tot["rays"], 1), 3)},
                "lane_utilisation": round(tot["lane_steps"] / max(64 * tot["wave_steps"], 1), 4),
                "algorithmic_gbs_per_gpu": round(per_gpu, 2)}

    shard = "tiles" if args.shard == "frames" else args.shard
    main_run = run(args.workload, shard, args.samples, args.steps, args.warmup)
    line = {
        "metric": "Mrays/sec (whole node); LBVH build ms per 1M triangles", "value": main_run["value"], "unit": "Mrays/s",
        "n_gpus": len(ids), "steps": args.steps, "warmup": args.warmup, "ms_per_step": main_run["ms_per_step"],
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": main_run["workload"], "multi": "group: one process, rthip_group_cast_device", "shard": shard,
                   "transport": main_run["transport"], "devices": ids, "rays_per_frame": main_run["rays_per_frame"]},
        # whole-call time (render on every device + exchange + placement), algorithmic bytes of the frame / devices
        "roofline": {"bound": "hbm", "achieved": main_run["algorithmic_gbs_per_gpu"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(main_run["algorithmic_gbs_per_gpu"] / HBM_PEAK_GBS, 5), "traffic": None,
                     "traffic_source": "not recorded for the group path", "per_ray": main_run["per_ray"],
                     "lane_utilisation": main_run["lane_utilisation"], "kernel": "the cast kernels of every device + exchange"},
        "regions": main_run["regions"],
    }
    if not args.no_config5 and args.workload == "bunny" and args.samples == 0:
        line["workloads"] = {f"cornell_mesh_4k_{args.c5_samples * args.c5_samples}spp_{m}": run("cornell-mesh", m, args.c5_samples, 1, 0)
                             for m in ("tiles", "samples")}
    print(json.dumps(line), flush=True)


def compact(line):
    r = line["roofline"]
    out = {"workload": line["config"]["workload"], "value": line["value"], "unit": line["unit"], "ms_per_step": line["ms_per_step"],
           "ms_per_frame": line["ms_per_frame"], "frames_per_step": line["config"]["frames_per_step"],
           "steps": line["steps"], "regions": line["regions"], "streams": line["config"]["streams"],
           "rays_per_frame": line["config"]["rays_per_frame_per_gpu"],
           "kernel": r["kernel"], "kernel_ms": r["kernel_ms"], "per_ray": r["per_ray"], "lane_utilisation": r["lane_utilisation"],
           "box_tests_per_lane_step": r["box_tests_per_lane_step"], "traffic_source": r["traffic_source"],
           "algorithmic_bytes_per_launch": r["algorithmic_bytes_per_launch"], "traffic": r["traffic"], "ceilings": r["ceilings"],
           "bounding": r["bounding"], "frac_of_bounding": r["frac_of_bounding"], "scene_build_ms": line["scene_build_ms"]}
    if "cpu_baseline" in line:
        out["cpu_baseline"] = line["cpu_baseline"]
    return out


def main():
    argv = sys.argv[1:]
    args = parse_args(argv)
    if args.multi == "group":
        group_main(args)
        return
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args, argv))  # nothing in this process has touched the GPU yet

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started {world} ranks")
    if args.launch_check:
        launch_check(args, rank, world)
        return

    import torch
    import torch.distributed as dist

    device_index = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(device_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", device_index))
        else:
            dist.init_process_group(backend="gloo")

    want_cpu = not args.no_cpu_baseline and world == 1  # the CPU baseline is reported at N = 1 only
    line = measure(args, args.workload, args.steps, args.warmup, args.streams, rank, world, device_index, want_cpu,
                   min_region_s=args.min_region_s)
    default_run = (args.workload == "bunny" and world == 1 and not args.no_workloads and args.samples == 0 and args.width == 0
                   and args.traversal == "exact")
    if rank == 0 and default_run:
        others = {}
        import copy

        for key, name, n_streams, steps, warmup, traversal in EXTRA_RUNS:
            a, c5 = args, key.endswith("spp")
            if c5:
                a = copy.copy(args)
                a.samples = args.c5_samples
                key = f"cornell_mesh_4k_{args.c5_samples * args.c5_samples}spp"
            try:  # a side figure that fails must not take the headline with it
                sub = measure(a, name, steps, warmup, n_streams, rank, world, device_index, want_cpu and traversal == "exact" and not c5,
                              cpu_budget_s=4.0, cpu_max_paths=1 << 17, extras=False, traversal=traversal, min_region_s=0.0 if c5 else 0.3)
                others[key] = compact(sub)
            except Exception as e:  # noqa: BLE001
                others[key] = {"error": f"{type(e).__name__}: {e}"}
        line["workloads"] = others
    if world > 1 and not args.no_config5 and args.workload == "bunny" and args.shard == "frames":
        # BASELINE config 5 proper beside the weak-scaling headline: ONE Cornell + mesh frame at 3840x2160 and 1024 spp
        # split between the ranks by tiles (an RCCL gather of disjoint tiles) and by sample index (an RCCL sum), one
        # exchange per frame: strong scaling.  Every rank takes part; rank 0 keeps the figures.
        # Their collectives are the first of their kind in the run (a gather of tiles, a per-frame reduce): should one of
        # them never return, a timer lets every rank leave with the headline instead of hanging the launcher.
        import threading

        def give_up():
            if rank == 0:
                line["workloads"] = {"error": f"the config-5 entries did not finish within {args.config5_limit_s:.0f} s"}
                print(json.dumps(line), flush=True)
            os._exit(0)

        timer = threading.Timer(args.config5_limit_s, give_up)
        timer.daemon = True
        timer.start()
        try:
            others = config5_entries(args, rank, world, device_index)
        except Exception as e:  # noqa: BLE001
            others = {"error": f"{type(e).__name__}: {e}"}
        timer.cancel()
        if rank == 0:
            line["workloads"] = others
    if rank == 0:
        print(json.dumps(line), flush=True)
    if world > 1:
        # the line is out: a rank that never reaches this barrier (it failed or gave up above) must not hold the others
        import threading

        last = threading.Timer(60.0, lambda: os._exit(0))
        last.daemon = True
        last.start()
        dist.barrier()
        dist.destroy_process_group()
        last.cancel()


if __name__ == "__main__":
    main()
