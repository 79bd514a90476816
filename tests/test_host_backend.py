"""The drop-in host backend (raytracer_amd/host/raytracer_hip.{h,c}).

CPU: where the reference checkout exists, the reference's own demo programs must compile and link
against the backend unchanged (apart from the two-line backend switch), and must refuse to run
without a GPU.  GPU: the demo binaries built that way (they travel with the tree) must render the
same floats as the Python mirror of the hooks."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT

HOST = os.path.join(ROOT, "raytracer_amd", "host")
BUILD = os.path.join(HOST, "_build")


@pytest.mark.skipif(not os.path.isdir("/root/reference/src"), reason="needs the reference checkout")
def test_reference_demos_build_against_the_hip_backend():
    subprocess.check_call(["make", "-s", "-C", HOST, "demos"])
    for demo in ("cornell", "tri", "spheres", "bunny"):
        assert os.path.exists(os.path.join(BUILD, demo + "_hip"))
    import torch

    if not torch.cuda.is_available():
        r = subprocess.run([os.path.join(BUILD, "tri_hip"), "/tmp/unused.hdr", "--width=4", "--height=4"],
                           capture_output=True, cwd="/tmp")
        assert r.returncode != 0 and b"no usable MI355X" in r.stderr, "the backend must not fall back to a CPU path"


@pytest.mark.gpu
@pytest.mark.parametrize("demo,samples", [("cornell", 2), ("tri", 1), ("spheres", 2)])
def test_reference_demo_binary_renders_through_the_backend(gpu_tracer, tmp_path, demo, samples):
    exe = os.path.join(BUILD, demo + "_hip")
    if not os.path.exists(exe):
        pytest.skip("demo binaries are only built where the reference checkout exists")
    from raytracer_amd import scenes, tracer as T

    wd, ht, seed = 72, 56, 4
    raw = tmp_path / "frame.f32"
    env = dict(os.environ, RT_HIP_DUMP_F32=str(raw), RT_HIP_HDR=str(tmp_path / "gpu.hdr"))
    subprocess.check_call([exe, str(tmp_path / "out.hdr"), f"--width={wd}", f"--height={ht}", f"--samples={samples}",
                           "--bounces=8", f"--seed={seed}"], cwd=str(tmp_path), env=env, timeout=120)
    got = np.fromfile(str(raw), dtype=np.float32).reshape(ht, wd, 3)
    assert os.path.getsize(tmp_path / "out.hdr") > 0
    # the hook seeds the GPU generator with the first rand() after the demo's srand(seed)
    libc = ctypes.CDLL("libc.so.6")
    libc.srand(seed)
    first = libc.rand()
    w, cam, sky = {"cornell": scenes.cornell, "tri": scenes.tri, "spheres": scenes.spheres}[demo]()
    gpu_tracer.check(gpu_tracer.lib.rthip_set_tracer(gpu_tracer.h, 8, 1 if sky else 0), "set_tracer")
    T.rt_tracer_build_blas(gpu_tracer, w)
    T.rt_tracer_build_tlas(gpu_tracer, w)
    want = np.zeros((ht, wd, 3), dtype=np.float32)
    T.rt_tracer_cast(gpu_tracer, cam(wd, ht, samples), want, wd, ht, seed=first)
    assert (got.view(np.uint32) == want.view(np.uint32)).all()
    # like the CPU backend, the hook leaves the TLAS in the reference's dump format in ./out.bvh
    from conftest import golden

    dump = (tmp_path / "out.bvh").read_bytes()
    assert dump == golden("demo_" + demo)["ref_tlas_dump"].tobytes(), "out.bvh differs from the reference's"
    # the demo wrote out.hdr with the reference's own stbi_write_hdr; the backend's GPU encoder (RT_HIP_HDR) wrote
    # the frame still in device memory: the two files must be the same bytes
    assert (tmp_path / "gpu.hdr").read_bytes() == (tmp_path / "out.hdr").read_bytes()


@pytest.mark.gpu
def test_reference_demo_binary_on_a_group_of_contexts(tmp_path):
    """RT_HIP_DEVICES puts several GPUs behind the same five hooks (here three contexts on the one GPU of the box): the
    reference's cornell program must render the bits it renders on one context, write the same out.bvh, and the GPU
    .hdr writer the same file."""
    exe = os.path.join(BUILD, "cornell_hip")
    if not os.path.exists(exe):
        pytest.skip("demo binaries are only built where the reference checkout exists")
    args = ["--width=200", "--height=150", "--samples=2", "--bounces=8", "--seed=9"]
    frames = {}
    for name, devices in (("one", None), ("group", "0,0,0")):
        d = tmp_path / name
        d.mkdir()
        env = dict(os.environ, RT_HIP_DUMP_F32=str(d / "frame.f32"), RT_HIP_HDR=str(d / "gpu.hdr"))
        env.pop("RT_HIP_DEVICES", None)
        if devices:
            env["RT_HIP_DEVICES"] = devices
        subprocess.check_call([exe, str(d / "out.hdr")] + args, cwd=str(d), env=env, timeout=120)
        frames[name] = (d / "frame.f32").read_bytes()
        assert (d / "gpu.hdr").read_bytes() == (d / "out.hdr").read_bytes()
    assert frames["one"] == frames["group"] and len(frames["one"]) == 200 * 150 * 12
    assert (tmp_path / "one" / "out.bvh").read_bytes() == (tmp_path / "group" / "out.bvh").read_bytes()


@pytest.mark.gpu
def test_reference_demo_binaries_take_the_ordered_walk_from_the_environment(gpu_tracer, tmp_path):
    """RT_HIP_TRAVERSAL=fast selects the ordered walk behind the five hooks, for one context and for a group.  On the
    Cornell box the two walks answer every ray alike (no ties between different primitives along its paths at this
    size), so the reference's cornell program must write the same frame; the frame is also the one the Python mirror
    renders with rthip_set_traversal(RTHIP_TRAV_FAST), which shows that the switch was taken."""
    exe = os.path.join(BUILD, "cornell_hip")
    if not os.path.exists(exe):
        pytest.skip("demo binaries are only built where the reference checkout exists")
    from raytracer_amd import scenes, tracer as T

    wd, ht, seed = 160, 120, 5
    args = [f"--width={wd}", f"--height={ht}", "--samples=2", "--bounces=8", f"--seed={seed}"]
    frames = {}
    for name, trav, devices in (("exact", None, None), ("fast", "fast", None), ("fast group", "fast", "0,0"), ("typo", "quick", None)):
        d = tmp_path / name.replace(" ", "_")
        d.mkdir()
        env = dict(os.environ, RT_HIP_DUMP_F32=str(d / "frame.f32"), RTHIP_TUNE_LOG="1")
        for k in ("RT_HIP_DEVICES", "RT_HIP_TRAVERSAL"):
            env.pop(k, None)
        if trav:
            env["RT_HIP_TRAVERSAL"] = trav
        if devices:
            env["RT_HIP_DEVICES"] = devices
        r = subprocess.run([exe, str(d / "out.hdr")] + args, cwd=str(d), env=env, timeout=120, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-1500:]
        frames[name] = np.fromfile(str(d / "frame.f32"), dtype=np.float32).reshape(ht, wd, 3)
        if name == "typo":
            assert "neither" in r.stderr
    libc = ctypes.CDLL("libc.so.6")
    libc.srand(seed)
    first = libc.rand()
    w, cam, sky = scenes.cornell()
    gpu_tracer.check(gpu_tracer.lib.rthip_set_tracer(gpu_tracer.h, 8, 1 if sky else 0), "set_tracer")
    try:
        T.set_traversal(gpu_tracer, "fast")
        T.rt_tracer_build_blas(gpu_tracer, w)
        T.rt_tracer_build_tlas(gpu_tracer, w)
        T.set_counting(gpu_tracer, True)
        T.get_stats(gpu_tracer)
        want = np.zeros((ht, wd, 3), dtype=np.float32)
        T.rt_tracer_cast(gpu_tracer, cam(wd, ht, 2), want, wd, ht, seed=first)
        fast_nodes = T.get_stats(gpu_tracer)["nodes_tested"]
    finally:
        T.set_counting(gpu_tracer, False)
        T.set_traversal(gpu_tracer, "exact")
    assert fast_nodes > 0
    assert (frames["fast"].view(np.uint32) == want.view(np.uint32)).all()
    assert (frames["fast group"].view(np.uint32) == want.view(np.uint32)).all()
    assert (frames["typo"].view(np.uint32) == frames["exact"].view(np.uint32)).all()
    # the two walks: the same frame up to paths that meet a tie or a seam ray (none or a handful of pixels here)
    differ = int((frames["fast"].view(np.uint32) != frames["exact"].view(np.uint32)).any(axis=2).sum())
    assert differ <= wd * ht // 1000, differ


@pytest.mark.gpu
def test_bunny_demo_with_gpu_loader_and_gpu_hdr_writer(gpu_tracer, tmp_path):
    """The reference's bunny demo with its two host-side I/O calls swapped for the backend's (ms_load_obj ->
    rt_hip_load_obj, stbi_write_hdr -> rt_hip_write_hdr; raytracer_amd/host/Makefile): data/bunny.obj is the stand-in
    mesh, the image must equal a render of the same mesh through the Python mirror, the .hdr the oracle's encoding."""
    exe = os.path.join(BUILD, "bunny_gpuio_hip")
    plain = os.path.join(BUILD, "bunny_hip")
    if not os.path.exists(exe):
        pytest.skip("demo binaries are only built where the reference checkout exists")
    from oracle import oracle as O
    from raytracer_amd import scenes, tracer as T

    verts, idx = scenes.bunny_class_mesh(3)
    (tmp_path / "data").mkdir()
    scenes.write_obj(str(tmp_path / "data" / "bunny.obj"), verts, idx)
    wd, ht, seed = 96, 54, 6
    args = [f"--width={wd}", f"--height={ht}", "--samples=2", "--bounces=2", f"--seed={seed}"]
    raw = tmp_path / "frame.f32"
    subprocess.check_call([exe, str(tmp_path / "gpuio.hdr")] + args, cwd=str(tmp_path),
                          env=dict(os.environ, RT_HIP_DUMP_F32=str(raw)), timeout=120)
    got = np.fromfile(str(raw), dtype=np.float32).reshape(ht, wd, 3)
    assert (tmp_path / "gpuio.hdr").read_bytes() == O.hdr_encode(got)
    # the loader restarts rand() from its own next value after the GPU runtime came up, and rt_make_tracer takes the
    # next one: seed -> srand(seed); a = rand(); srand(a); rand()
    libc = ctypes.CDLL("libc.so.6")
    libc.srand(seed)
    libc.srand(libc.rand())
    first = libc.rand()
    w, cam, sky = scenes.bunny(*T.ms_load_obj(gpu_tracer, str(tmp_path / "data" / "bunny.obj"), 3)[:2])
    gpu_tracer.check(gpu_tracer.lib.rthip_set_tracer(gpu_tracer.h, 2, 1 if sky else 0), "set_tracer")
    T.rt_tracer_build_blas(gpu_tracer, w)
    T.rt_tracer_build_tlas(gpu_tracer, w)
    want = np.zeros((ht, wd, 3), dtype=np.float32)
    T.rt_tracer_cast(gpu_tracer, cam(wd, ht, 2), want, wd, ht, seed=first)
    assert (got.view(np.uint32) == want.view(np.uint32)).all()
    # and the unpatched demo (the reference's own loader and writer on top of the hooks) sees the same mesh: same hits,
    # another seed, so compare the primary-ray silhouette through the sky colour of the first sample instead of bits
    subprocess.check_call([plain, str(tmp_path / "plain.hdr")] + args, cwd=str(tmp_path),
                          env=dict(os.environ, RT_HIP_DUMP_F32=str(tmp_path / "plain.f32")), timeout=120)
    plain_img = np.fromfile(str(tmp_path / "plain.f32"), dtype=np.float32).reshape(ht, wd, 3)
    assert abs(float(plain_img.mean()) - float(got.mean())) < 0.02 * float(got.mean())


@pytest.mark.gpu
def test_debug_build_of_a_demo_writes_the_reference_ray_dump(gpu_tracer, tmp_path):
    """The Cornell demo in the reference's debug flavour on top of the backend: ./out.rays must hold, for a one-bounce
    frame, exactly the records the CPU backend's debug build wrote (tests/golden/ray_records.npz; the material
    handles are addresses, so they are compared as a grouping), and for a multi-bounce frame the records of
    rthip_cast_ray_records with the seed the hook derives."""
    exe = os.path.join(BUILD, "cornell_debug_hip")
    if not os.path.exists(exe):
        pytest.skip("demo binaries are only built where the reference checkout exists")
    from conftest import golden
    from raytracer_amd import abi, scenes, tracer as T
    from test_oracle_golden import same_ray_records

    file_t = np.dtype([("origin", "<f4", 3), ("direction", "<f4", 3), ("p", "<f4", 3), ("n", "<f4", 3), ("t", "<f4"),
                       ("handle", "<u8")])
    env = dict(os.environ, RT_HIP_NO_DUMP="1")
    subprocess.check_call([exe, str(tmp_path / "a.hdr"), "--width=16", "--height=16", "--samples=1", "--bounces=1", "--seed=1"],
                          cwd=str(tmp_path), env=dict(os.environ), timeout=120)
    # a debug build's trees are not a release build's (lbvh.c:97-104): out.bvh must be the debug flavour's
    with open(str(tmp_path / "out.bvh"), "rb") as f:
        assert f.read() == golden("debug_flavour")["cornell_demo_tlas_dump"].tobytes()
    rec = np.fromfile(str(tmp_path / "out.rays"), dtype=file_t)
    mine = np.zeros(rec.size, dtype=abi.RAY_RECORD)
    for f in ("origin", "direction", "p", "n", "t"):
        mine[f] = rec[f]
    handles = {h: k for k, h in enumerate(sorted(set(rec["handle"].tolist()) - {0}))}
    mine["material"] = [handles.get(h, -1) for h in rec["handle"].tolist()]
    same_ray_records(mine, golden("ray_records"), "cornell_16x16_s1_b1_seed1")

    wd, ht, seed = 24, 18, 5
    subprocess.check_call([exe, str(tmp_path / "b.hdr"), f"--width={wd}", f"--height={ht}", "--samples=2", "--bounces=5",
                           f"--seed={seed}"], cwd=str(tmp_path), env=env, timeout=120)
    rec = np.fromfile(str(tmp_path / "out.rays"), dtype=file_t)
    libc = ctypes.CDLL("libc.so.6")
    libc.srand(seed)
    first = libc.rand()
    w, cam, sky = scenes.cornell()
    gpu_tracer.check(gpu_tracer.lib.rthip_set_tracer(gpu_tracer.h, 5, 1 if sky else 0), "set_tracer")
    T.rt_tracer_build_blas(gpu_tracer, w)
    T.rt_tracer_build_tlas(gpu_tracer, w)
    want = T.cast_ray_records(gpu_tracer, cam(wd, ht, 2), wd, ht, seed=first)
    assert rec.size == want.size
    for f in ("origin", "direction", "p", "n", "t"):
        assert np.ascontiguousarray(rec[f]).tobytes() == np.ascontiguousarray(want[f]).tobytes(), f
    assert ((rec["handle"] == 0) == (want["material"] < 0)).all()
    assert len(set(zip(rec["handle"].tolist(), want["material"].tolist()))) == len(set(want["material"].tolist()))
