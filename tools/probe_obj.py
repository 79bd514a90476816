"""dev probe: rthip_obj_load against the reference's ms_load_obj (oracle/_ref) and the CPU restatement on big files."""
import sys, os, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from raytracer_amd import tracer as T
from oracle import oracle as O

tr = T.rt_make_tracer(T.RT_TracerSettings(max_bounces=2, sky=True))
for n_tris in [int(a) for a in sys.argv[1:]] or [100_000, 1_000_000]:
    rng = np.random.default_rng(1)
    cen = rng.random((n_tris, 1, 3), dtype=np.float32)
    v = (cen + (rng.random((n_tris, 3, 3), dtype=np.float32) - 0.5) * 0.002).reshape(-1, 3)
    path = os.path.join(tempfile.gettempdir(), f"soup_{n_tris}.obj")
    t0 = time.perf_counter()
    with open(path, "w") as f:
        f.write("\n".join("v %.9g %.9g %.9g" % (a, b, c) for a, b, c in v.tolist()))
        f.write("\n")
        f.write("\n".join("f %d %d %d" % (3 * k + 1, 3 * k + 2, 3 * k + 3) for k in range(n_tris)))
        f.write("\n")
    print(f"{n_tris} triangles: wrote {os.path.getsize(path) / 1e6:.0f} MB in {time.perf_counter() - t0:.1f} s", flush=True)
    best = None
    for _ in range(3):
        _, _, info = T.ms_load_obj(tr, path, 1, read_back=False)
        if best is None or info["total_ms"] < best["total_ms"]:
            best = info
    print(f"  rthip_obj_load: total {best['total_ms']:.1f} ms (parse on {best['host_threads']} threads {best['parse_ms']:.1f} ms, "
          f"GPU dedup {best['dedup_ms']:.2f} ms), {best['vertex_count']} vertices, {best['index_count']} indices", flush=True)
    verts, idx, _ = T.ms_load_obj(tr, path, 1)
    t0 = time.perf_counter(); ov, oi = O.load_obj(path, 1); t_port = time.perf_counter() - t0
    assert (idx == oi).all() and verts.tobytes() == ov.tobytes()
    line = f"  CPU restatement {t_port * 1e3:.0f} ms"
    if O.have_ref("core"):
        t0 = time.perf_counter(); rv, ri = O.ref_load_obj(path, 1, 3 * n_tris + 8, 3 * n_tris + 8); t_ref = time.perf_counter() - t0
        assert (idx == ri).all()
        line += f", reference ms_load_obj {t_ref * 1e3:.0f} ms"
    print(line, flush=True)
    os.unlink(path)
T.rt_tracer_cleanup(tr)
