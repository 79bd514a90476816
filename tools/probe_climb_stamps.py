"""dev probe: where a workgroup of the LBVH's chunk climb spends its time (library built with -DRT_CLIMB_STAMP,
loaded through RTHIP_LIBRARY): wall-clock stamps at the phases of the first 32768 workgroups of the last build."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from raytracer_amd import tracer as T
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
tr = T.rt_make_tracer(T.RT_TracerSettings(max_bounces=1, sky=True))
g = torch.Generator(device="cuda"); g.manual_seed(1)
if len(sys.argv) > 2 and sys.argv[2] == "coherent":  # centroids along a space-filling order: a coherent gather
    k = torch.arange(n, device="cuda")
    side = int(round(n ** (1 / 3))) + 1
    cen = torch.stack([(k % side), (k // side) % side, k // (side * side)], 1).float().reshape(n, 1, 3) / side
else:
    cen = torch.rand((n, 1, 3), device="cuda", generator=g)
off = (torch.rand((n, 3, 3), device="cuda", generator=g) * 2 - 1) * 0.001
p = (cen + off).reshape(3 * n, 3).float()
px, py, pz = (p[:, k].contiguous() for k in range(3))
nodes = torch.empty((2 * n - 1, 8), dtype=torch.float32, device="cuda")
s = torch.cuda.current_stream()
for it in range(3):
    tr.check(tr.lib.rthip_mesh_build_device(tr.h, px.data_ptr(), py.data_ptr(), pz.data_ptr(), px.numel(), None, n, nodes.data_ptr(), s.cuda_stream), "build")
torch.cuda.synchronize()
groups = min(32768, (n + 511) // 512)
st = np.zeros(8 * 32768, dtype=np.uint64)
tr.lib.rthip_debug_climb_stamps.argtypes = [C.c_void_p, C.c_size_t]
assert tr.lib.rthip_debug_climb_stamps(st.ctypes.data, st.size) == 0
st = st.reshape(-1, 8)[:groups].astype(np.int64)
t = (st[:, :6] - st[:, :1]) / 100.0  # us (100 MHz clock)
names = ["start", "window done (ids, priorities read)", "boxes arrived", "climb done (all waves)", "list flushed", "stores drained"]
if n <= (1 << 17) and os.environ.get("RTHIP_UPPER_RANGES", "2") == "2":  # the range form: other phases behind the same stamps
    names = ["start", "shape scan and window done", "boxes arrived", "(nothing)", "all nodes of the chunk, its tables and its list", "stores drained"]
    for k in range(1, 6):
        d = t[:, k] - t[:, k - 1]
        print(f"{names[k]:40s} +{d.mean():6.2f} us (median {np.median(d):6.2f}, p95 {np.percentile(d, 95):6.2f})")
    print(f"of that: block scans, tables and nearest entries in the wave +{((st[:, 6] - st[:, 3]) / 100.0).mean():.2f} us, block tables, chunk tables and entries beyond the wave +{((st[:, 7] - st[:, 6]) / 100.0).mean():.2f}, leaves' boxes to the chunk's ends, nodes' boxes, records and the list +{((st[:, 4] - st[:, 7]) / 100.0).mean():.2f}")
    print(f"workgroup lifetime {t[:, 5].mean():.2f} us; kernel span {(st[:, 5].max() - st[:, 0].min()) / 100.0:.1f} us for {groups} workgroups")
    sys.exit(0)
for k in range(1, 6):
    d = t[:, k] - t[:, k - 1]
    print(f"{names[k]:40s} +{d.mean():6.2f} us (median {np.median(d):6.2f}, p95 {np.percentile(d, 95):6.2f})")
print(f"generations per chunk: mean {st[:, 6].mean():.1f}, max {st[:, 6].max()}; leaves' records stored and handed up: {((st[:, 7] - st[:, 2]) / 100.0).mean():.2f} us; per generation after that {((st[:, 3] - st[:, 7]) / 100.0 / st[:, 6]).mean():.2f} us")
print(f"workgroup lifetime {t[:, 5].mean():.2f} us; kernel span {(st[:, 5].max() - st[:, 0].min()) / 100.0:.1f} us for {groups} workgroups")
start = np.sort(st[:, 0] - st[:, 0].min()) / 100.0
print("workgroups started within 5 / 20 / 100 us:", int((start < 5).sum()), int((start < 20).sum()), int((start < 100).sum()))

