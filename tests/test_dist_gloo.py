"""The N > 1 path on CPU: two gloo ranks, each holding the share of a frame that its GPU would have
rendered (here: rendered by the oracle with the same shard record), combined with the product's
`combine_framebuffer`.  Tiles must reproduce the single-rank frame bit for bit."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from raytracer_amd import dist as rdist, scenes, world


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world_size, port, mode, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    from oracle import oracle as O

    w, cam, sky = scenes.cornell()
    flat = world.flatten_world(w)
    wd, ht, smp = 96, 80, 2
    c = cam(wd, ht, smp)
    sc = O.Scene(flat, 8, sky)
    sh = rdist.shard_record(mode, rank, world_size)
    seed = 5 + (rank if mode == "frames" else 0)
    share, _ = sc.cast(c, seed, wd, ht, rng_mode=O.RNG_COUNTER, shard=tuple(int(x) for x in sh[0]), threads=2)
    fb = torch.from_numpy(share.copy())
    rdist.combine_framebuffer(fb, mode, smp, world_size, dst=0)
    if rank == 0:
        np.save(os.path.join(out_dir, f"{mode}.npy"), fb.numpy())
    dist.barrier()
    dist.destroy_process_group()


def _run(mode, tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, mode, str(tmp_path)), nprocs=2, join=True)
    return np.load(os.path.join(str(tmp_path), f"{mode}.npy"))


def _full(seed):
    from oracle import oracle as O

    w, cam, sky = scenes.cornell()
    sc = O.Scene(world.flatten_world(w), 8, sky)
    img, _ = sc.cast(cam(96, 80, 2), seed, 96, 80, rng_mode=O.RNG_COUNTER, threads=4)
    return img


def test_tile_shards_reproduce_the_frame(tmp_path):
    """Tiles are gathered (every rank sends the pixels of its own tiles only), not summed: the frame's own bits."""
    got = _run("tiles", tmp_path)
    want = _full(5)
    assert (got.view(np.uint32) == want.view(np.uint32)).all()


def test_tile_pixels_partition_the_frame():
    for (w, h, world) in ((96, 80, 2), (200, 130, 3), (64, 64, 4), (1, 1, 2), (129, 65, 5)):
        seen = np.zeros(w * h, dtype=np.int32)
        for r in range(world):
            idx = rdist.tile_pixels(w, h, r, world).numpy()
            seen[idx] += 1
            ys, xs = idx // w, idx % w
            assert ((((ys // 64) * (-(-w // 64)) + xs // 64) % world) == r).all()
        assert (seen == 1).all()


def test_sample_shards_sum_to_the_frame(tmp_path):
    got = _run("samples", tmp_path)
    want = _full(5)
    assert np.allclose(got, want, rtol=2e-6, atol=1e-7)


def test_frame_replicas_average(tmp_path):
    got = _run("frames", tmp_path)
    want = (_full(5).astype(np.float32) + _full(6).astype(np.float32)) * np.float32(0.5)
    assert np.allclose(got, want, rtol=1e-6, atol=1e-7)


def _pipeline_worker(rank, world_size, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    bufs = [torch.zeros((8, 8, 3)), torch.zeros((8, 8, 3))]
    pipe = rdist.FramePipeline(bufs, "frames", 1, world_size, dst=0)
    seen = []
    for frame in range(5):
        fb = pipe.acquire()
        if rank == 0 and frame >= 2:  # the buffer comes back holding the combined frame of two frames ago
            seen.append(float(fb[0, 0, 0]))
        fb.fill_(float(10 * frame + rank))  # what a render of this frame on this rank would have left
        pipe.submit(fb)
    last = pipe.drain()
    if rank == 0:
        seen.append(float(last[0, 0, 0]))
        np.save(os.path.join(out_dir, "pipe.npy"), np.array(seen, dtype=np.float64))
    dist.barrier()
    dist.destroy_process_group()


def test_frame_pipeline_keeps_frames_apart(tmp_path):
    port = _free_port()
    mp.spawn(_pipeline_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    got = np.load(os.path.join(str(tmp_path), "pipe.npy"))
    # mean over the two ranks of 10 * frame + rank = 10 * frame + 0.5, for frames 0, 1, 2 and the last one (4)
    assert np.allclose(got, [0.5, 10.5, 20.5, 40.5])


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` with WORLD_SIZE unset must launch the two ranks itself and pass rank 0's JSON line and
    the exit code on (checked here without a GPU: the ranks only join the gloo group and add up their ranks)."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--launch-check"],
                       capture_output=True, text=True, timeout=300, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["rank_sum"] == 1.0
    # a launcher that started another number of ranks than --gpus says is an error, not a silent run
    env2 = dict(env, WORLD_SIZE="1", RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--launch-check"], capture_output=True,
                       text=True, timeout=120, env=env2, cwd=root)
    assert r.returncode != 0
