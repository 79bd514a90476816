"""dev probe: rthip_hdr_encode_device on device-resident noise / flat frames, several times (for kernel traces; no checks)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracer_amd import tracer as T
tr = T.rt_make_tracer(T.RT_TracerSettings(max_bounces=1, sky=True))
s = torch.cuda.current_stream().cuda_stream
for (W, H) in ((1920, 1080), (3840, 2160)):
    fb = torch.rand((H, W, 3), device="cuda")
    fb[H // 3: H // 2] = 0.5
    cap = int(tr.lib.rthip_hdr_bound(W, H))
    out = torch.zeros(cap, dtype=torch.uint8, device="cuda")
    size = torch.zeros(1, dtype=torch.int64, device="cuda")
    for _ in range(5):
        tr.check(tr.lib.rthip_hdr_encode_device(tr.h, fb.data_ptr(), W, H, out.data_ptr(), cap, size.data_ptr(), s), "encode")
    torch.cuda.synchronize()
    print(W, H, int(size.item()))
