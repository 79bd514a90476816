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
