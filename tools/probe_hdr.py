"""dev probe: device time of the .hdr encoder against the reference's writer / the CPU restatement on this host."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from raytracer_amd import scenes, tracer as T
from oracle import oracle as O

tr = T.rt_make_tracer(T.RT_TracerSettings(max_bounces=4, sky=True))
w, cam, sky = scenes.cornell()
T.rt_tracer_build_blas(tr, w); T.rt_tracer_build_tlas(tr, w)
for (W, H) in ((512, 512), (1920, 1080), (3840, 2160)):
    out = np.zeros((H, W, 3), np.float32)
    T.rt_tracer_cast(tr, cam(W, H, 1), out, W, H, seed=1)
    for src, label in ((None, "resident frame"), (out, "host frame")):
        best = 1e9; wall = 1e9
        for _ in range(5):
            t0 = time.perf_counter(); data = T.hdr_encode(tr, src, W, H); wall = min(wall, time.perf_counter() - t0)
            best = min(best, T.get_stats(tr)["last_encode_ms"])
        print(f"{W}x{H} {label:14s}: kernels {best:.3f} ms, call {wall*1e3:.2f} ms, {len(data)} bytes ({len(data)/(W*H):.2f} B/pixel)", flush=True)
    t0 = time.perf_counter(); want = O.hdr_encode(out); t_port = time.perf_counter() - t0
    assert want == data
    if O.have_ref("hdr"):
        t0 = time.perf_counter(); O.ref_hdr_encode(out); t_ref = time.perf_counter() - t0
        print(f"   CPU: reference writer {t_ref*1e3:.1f} ms, restatement {t_port*1e3:.1f} ms")
    else:
        print(f"   CPU: restatement {t_port*1e3:.1f} ms")
T.rt_tracer_cleanup(tr)
