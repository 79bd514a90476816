"""rthip_lbvh_build_device_noalloc: the LBVH build whose workspace is a fixed array inside the library (no device
allocation in the call) must produce the tree of rthip_lbvh_build_device bit for bit -- on calls in a row that share
the array, each smaller than the one before, through every size class of the builder."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

MAX_BOXES = 1 << 17  # RTHIP_NOALLOC_MAX_BOXES, include/rthip.h


def boxes_on_device(torch, n, seed):
    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    cen = torch.rand((n, 3), device="cuda", generator=g)
    half = torch.rand((n, 3), device="cuda", generator=g) * 0.01
    return torch.cat([cen - half, cen + half], dim=1).contiguous()


def test_noalloc_build_equals_the_allocating_build(gpu_tracer):
    import torch

    lib, h = gpu_tracer.lib, gpu_tracer.h
    stream = torch.cuda.current_stream().cuda_stream
    # largest first: every later call finds the array full of the previous call's contents
    sizes = [MAX_BOXES, 100_000, 20_000, 3_000, 600, 300, 2, 1]
    inputs = [boxes_on_device(torch, n, 100 + k) for k, n in enumerate(sizes)]
    want = []
    for n, b in zip(sizes, inputs):
        nodes = torch.zeros((2 * n - 1, 8), dtype=torch.int32, device="cuda")
        gpu_tracer.check(lib.rthip_lbvh_build_device(h, b.data_ptr(), n, nodes.data_ptr(), stream), "rthip_lbvh_build_device")
        want.append(nodes)
    torch.cuda.synchronize()
    got = []
    for n, b in zip(sizes, inputs):  # back to back on one stream, no synchronisation in between
        nodes = torch.full((2 * n - 1, 8), -1, dtype=torch.int32, device="cuda")
        gpu_tracer.check(lib.rthip_lbvh_build_device_noalloc(h, b.data_ptr(), n, nodes.data_ptr(), stream),
                         "rthip_lbvh_build_device_noalloc")
        got.append(nodes)
    torch.cuda.synchronize()
    for n, a, b in zip(sizes, got, want):
        assert torch.equal(a, b), f"n = {n}: {int((a != b).any(dim=1).sum())} of {2 * n - 1} node records differ"
    # and once more, smaller after larger, with the results of the first round as the reference
    for k in (1, 3):
        n, b = sizes[k], inputs[k]
        nodes = torch.full((2 * n - 1, 8), -1, dtype=torch.int32, device="cuda")
        gpu_tracer.check(lib.rthip_lbvh_build_device_noalloc(h, b.data_ptr(), n, nodes.data_ptr(), stream),
                         "rthip_lbvh_build_device_noalloc")
        torch.cuda.synchronize()
        assert torch.equal(nodes, want[k]), f"second round, n = {n}"


def test_noalloc_build_rejects_what_does_not_fit(gpu_tracer):
    import torch

    from raytracer_amd import tracer as T

    lib, h = gpu_tracer.lib, gpu_tracer.h
    b = boxes_on_device(torch, 16, 1)
    nodes = torch.zeros((31, 8), dtype=torch.int32, device="cuda")
    with pytest.raises(T.RTHipError, match="at most"):
        gpu_tracer.check(lib.rthip_lbvh_build_device_noalloc(h, b.data_ptr(), MAX_BOXES + 1, nodes.data_ptr(), None), "noalloc")
    with pytest.raises(T.RTHipError):
        gpu_tracer.check(lib.rthip_lbvh_build_device_noalloc(h, b.data_ptr(), 0, nodes.data_ptr(), None), "noalloc")
    with pytest.raises(T.RTHipError):
        gpu_tracer.check(lib.rthip_lbvh_build_device_noalloc(h, None, 16, nodes.data_ptr(), None), "noalloc")


def test_noalloc_build_can_be_captured_and_replayed(gpu_tracer):
    """What the entry point is for: under stream capture no allocation is allowed.  The captured build is replayed on
    other boxes in the same buffer and gives the allocating build's tree each time."""
    import torch

    lib, h = gpu_tracer.lib, gpu_tracer.h
    n = 50_000
    boxes = boxes_on_device(torch, n, 7)
    nodes = torch.zeros((2 * n - 1, 8), dtype=torch.int32, device="cuda")
    s = torch.cuda.Stream()
    graph = torch.cuda.CUDAGraph()
    torch.cuda.synchronize()
    with torch.cuda.graph(graph, stream=s):
        gpu_tracer.check(lib.rthip_lbvh_build_device_noalloc(h, boxes.data_ptr(), n, nodes.data_ptr(), s.cuda_stream), "noalloc (capture)")
    for seed in (7, 8):
        boxes.copy_(boxes_on_device(torch, n, seed))
        nodes.fill_(-1)
        graph.replay()
        torch.cuda.synchronize()
        ref = torch.zeros((2 * n - 1, 8), dtype=torch.int32, device="cuda")
        gpu_tracer.check(lib.rthip_lbvh_build_device(h, boxes.data_ptr(), n, ref.data_ptr(), torch.cuda.current_stream().cuda_stream), "build")
        torch.cuda.synchronize()
        assert torch.equal(nodes, ref), f"replay with seed {seed}"
