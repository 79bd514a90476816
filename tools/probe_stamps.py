"""dev probe (needs the -DRT_STAMP build of the library, RTHIP_LIBRARY=tools/variants/librthip_stamp.so): start / end
wall-clock stamps of every wave of the static kernel on one frame -> where the frame's time goes."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from raytracer_amd import tracer as T
import bench

name = sys.argv[1] if len(sys.argv) > 1 else "bunny"
wl = bench.make_workload(name)
tr = T.rt_make_tracer(T.RT_TracerSettings(max_bounces=wl["bounces"], sky=wl["sky"]))
T.rt_tracer_build_blas(tr, wl["world"]); T.rt_tracer_build_tlas(tr, wl["world"])
T.set_cast_kernel(tr, "static")
W, H = wl["width"], wl["height"]
cast = wl["cam"](W, H, wl["samples"])
out = np.zeros((H, W, 3), np.float32)
T.set_counting(tr, True)
for _ in range(3):
    T.rt_tracer_cast(tr, cast, out, W, H, seed=1)
print("kernel ms", T.get_stats(tr)["last_cast_ms"])
n_waves = ((W + 15) // 16) * ((H + 15) // 16) * 4
n_waves = min(n_waves, 1 << 16)
st = np.zeros(4 * n_waves, np.uint64)
tr.lib.rthip_debug_stamps.argtypes = [C.c_void_p, C.c_size_t]
assert tr.lib.rthip_debug_stamps(st.ctypes.data, st.size) == 0
t0 = st[0::4].astype(np.int64); t1 = st[1::4].astype(np.int64); steps = st[2::4].astype(np.int64); nodes = st[3::4].astype(np.int64)
base = t0.min()
s = (t0 - base) / 100.0; e = (t1 - base) / 100.0   # 100 MHz -> us
d = e - s
print(f"waves {n_waves}; frame span {e.max():.1f} us; last start {s.max():.1f} us")
print("duration percentiles us (50/90/99/99.9/max):", [round(float(np.percentile(d, p)), 1) for p in (50, 90, 99, 99.9, 100)])
print("sum of wave durations / (span * 1024 SIMDs) = avg waves per SIMD:", round(float(d.sum() / (e.max() * 1024)), 2))
# how many waves are alive over time
grid = np.linspace(0, e.max(), 21)
alive = [(int(((s <= t) & (e > t)).sum())) for t in grid]
print("alive waves at 5% steps:", alive)
# when do the long waves start?
long = d > np.percentile(d, 99)
print(f"top 1% waves: start {s[long].min():.1f}..{s[long].max():.1f} us, end {e[long].min():.1f}..{e[long].max():.1f}")
order = np.argsort(s)
print("start time of wave index quantiles (dispatch order vs time):", [round(float(s[int(q * (n_waves - 1))]), 1) for q in (0, .25, .5, .75, 1)])
top = np.argsort(-d)[:8]
for w in top:
    print(f"wave {w}: {d[w]:.1f} us, {steps[w]} iterations ({d[w]*1e3/max(steps[w],1):.0f} ns each), {nodes[w]} lane-steps, util {nodes[w]/max(64*steps[w],1):.2f}")
heavy = steps > 50
print(f"all waves with >50 iterations: {heavy.sum()} waves, mean {d[heavy].mean():.1f} us, mean iterations {steps[heavy].mean():.0f}, ns/iteration {1e3*d[heavy].sum()/steps[heavy].sum():.0f}")
# ns per iteration as a function of the number of waves alive
for lo_t, hi_t in ((0, 40), (40, 100), (100, 200), (200, 400)):
    sel = heavy & (s >= 0) & (e > lo_t) & (e <= hi_t)
    if sel.any(): print(f"  waves ending in {lo_t}-{hi_t} us: {sel.sum()}, ns/iteration {1e3*d[sel].sum()/steps[sel].sum():.0f}")
T.rt_tracer_cleanup(tr)
