"""dev probe: BLAS builds far above the benchmark's 10^7 triangles, checked through the size-independent tree
invariants of the 10^7 test (tests/helpers.py) and timed.  Usage: probe_build_big.py [n ...]"""
import sys, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
sys.path.insert(0, os.path.join(root, "tests"))
import torch
from raytracer_amd import tracer as T
from helpers import build_and_check_invariants

tr = T.rt_make_tracer(T.RT_TracerSettings(max_bounces=1, sky=True))
for n in [int(a) for a in sys.argv[1:]] or [50_000_000, 100_000_000]:
    ms = build_and_check_invariants(tr, n, seed=5, builds=3)
    print(f"n={n}: build {ms:.2f} ms = {ms * 1e6 / n:.3f} ms per 1 M, invariants hold; "
          f"peak device memory seen by torch {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB", flush=True)
    torch.cuda.empty_cache()
