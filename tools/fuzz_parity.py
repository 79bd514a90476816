"""dev probe: seeded random worlds (scenes.random_world) rendered by every cast kernel on the GPU and by the CPU
restatement with the same counter-based random numbers; trees, closest hits of random rays and whole frames must
agree bit for bit.  Usage: fuzz_parity.py [first_seed [count]]"""
import sys, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
from raytracer_amd import scenes, world, tracer as T
from oracle import oracle
from helpers import check_random_world

oracle.build_oracle()
first = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 200
tr = T.rt_make_tracer(T.RT_TracerSettings(max_bounces=8, sky=True))
bad = 0
for seed in range(first, first + count):
    try:
        check_random_world(oracle, tr, seed)
    except AssertionError as e:
        bad += 1
        print(f"seed {seed}: {e}", flush=True)
    if (seed - first) % 20 == 19:
        print(f"... {seed - first + 1} worlds, {bad} with differences", flush=True)
print(f"{count} worlds from seed {first}: {bad} with differences", flush=True)
sys.exit(1 if bad else 0)
