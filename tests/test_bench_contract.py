"""bench.py prints ONE JSON line with the agreed keys; N = 1, short runs (needs the GPU)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def run_bench(*flags):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags], cwd=ROOT, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def check_common(d, steps, warmup):
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == steps and d["warmup"] == warmup
    assert d["unit"] == "Mrays/s" and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["dtype"] == "f32" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "kernel", "kernel_ms"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert d["value"] > 0 and d["ms_per_step"] > 0
    # the reported region is the median of `count` regions of exactly `steps` steps
    g = d["regions"]
    assert g["count"] >= 1 and g["ms_per_step_min"] <= d["ms_per_step"] <= g["ms_per_step_max"]
    assert d["ms_per_step"] == g["ms_per_step_median"]
    check_derived(r)
    # a counter that was not recorded on these sources is not handed out, and nothing is derived from it
    if r["traffic"] is None:
        assert "hbm_measured_frac" not in r["ceilings"] and r["valu"] is None and r["traffic_source"]
        assert r["bounding"].startswith("unknown") and r["frac_of_bounding"] is None
    # value = rays of all steps / time of the timed region
    rays = d["config"]["rays_per_frame_per_gpu"]
    frames = d["config"]["frames_per_step"]
    assert frames == d["config"]["streams"]
    assert abs(d["value"] - frames * rays / (d["ms_per_step"] * 1e-3) / 1e6) < 0.02 * d["value"]
    assert abs(d["ms_per_frame"] * frames - d["ms_per_step"]) < 0.01 * d["ms_per_step"]


def check_derived(r):
    """Derived figures that must never contradict themselves: no share of the lanes above 1, no HBM fraction above 1."""
    assert 0.0 < r["lane_utilisation"] <= 1.0, r["lane_utilisation"]
    c = r["ceilings"]
    assert c["hbm_algorithmic_frac"] is None or c["hbm_algorithmic_frac"] <= 1.0
    if c["hbm_algorithmic_frac"] is None:
        assert "cache-resident" in c["hbm_algorithmic_note"]
    assert c.get("hbm_measured_frac", 0.0) <= 1.0
    assert c.get("valu_issue_frac", 0.0) <= 1.0 and (c.get("frac_of_bounding") or 0.0) <= 1.0
    if "frac_overlapped" in r:
        assert r["frac_overlapped"] is None or r["frac_overlapped"] <= 1.0


def test_default_workload_line():
    d = run_bench("--steps", "24", "--warmup", "4", "--build-sizes", "81920", "--no-cpu-baseline", "--no-workloads")
    check_common(d, 24, 4)
    assert "workloads" not in d and d["host_cast"]["value"] > 0
    r = d["roofline"]
    assert r["frac_overlapped"] is None or r["frac_overlapped"] <= 1.0
    assert r["bounding"] and "hbm_algorithmic_frac" in r["ceilings"]
    assert "1920x1080" in d["config"]["workload"] and "samples=1" in d["config"]["workload"]
    assert d["config"]["streams"] == 4
    assert "81920" in d["lbvh_build"] and d["lbvh_build_ms_per_1M_triangles"] > 0
    assert d["shadow_rays"]["value"] > 0 and d["hdr_output"]["file_bytes"] > 0 and d["obj_load"]["triangles"] == 81921
    assert "cpu_baseline" not in d


def test_cpu_baseline_and_single_stream():
    d = run_bench("--workload", "cornell", "--streams", "1", "--steps", "5", "--warmup", "1", "--build-sizes", "")
    check_common(d, 5, 1)
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("reference", "port") and c["cores"] == 1 and c["value"] > 0
    assert d["hdr_output"]["identical_to_cpu"] is True
    assert d["roofline"]["concurrent_launches"] == 1


def test_default_run_carries_every_configuration():
    """The driver's plain `python bench.py`: the line also holds the other BASELINE configurations, each with its
    counters per ray and the ceiling that bounds it."""
    d = run_bench("--steps", "8", "--warmup", "2", "--build-sizes", "", "--no-cpu-baseline")
    check_common(d, 8, 2)
    w = d["workloads"]
    for key in ("bunny_one_stream", "instanced", "instanced_one_stream", "soup", "soup_rays", "instanced_rays", "cornell", "cornell_mesh_4k"):
        assert key in w, key
        for k in ("workload", "value", "ms_per_step", "kernel", "kernel_ms", "per_ray", "traffic", "bounding", "ceilings"):
            assert k in w[key], (key, k)
        assert w[key]["value"] > 0
    assert w["bunny_one_stream"]["streams"] == 1 and w["soup_rays"]["rays_per_frame"] == 1 << 24
    assert w["soup_rays"]["per_ray"]["nodes"] > 100
    for key, sub in w.items():
        check_derived(sub)
        assert sub["regions"]["count"] >= 1
    # both walks enter an instance only behind its box: the same count on the same frame
    assert w["bunny_fast"]["per_ray"]["instances"] == w["bunny_one_stream"]["per_ray"]["instances"]
    assert "cornell_mesh_4k_1024spp" in w and w["cornell_mesh_4k_1024spp"]["value"] > 0


def test_twenty_and_two_hundred_steps_agree():
    """The driver runs a short region: the median over repeated regions must make its figure the long run's (measured
    2.6 % apart; one region of 20 single-frame steps used to be 16 % under the long run).  A timing test on a shared
    box: the bound is twice what is measured, and a miss is tried once more before it counts."""
    for attempt in range(2):
        a = run_bench("--steps", "20", "--warmup", "5", "--build-sizes", "", "--no-cpu-baseline", "--no-workloads")
        b = run_bench("--steps", "200", "--warmup", "20", "--build-sizes", "", "--no-cpu-baseline", "--no-workloads")
        assert a["regions"]["count"] > b["regions"]["count"] >= 1
        if abs(a["value"] - b["value"]) <= 0.06 * b["value"]:
            return
    raise AssertionError((a["value"], b["value"], a["regions"], b["regions"]))


def test_two_ranks_start_themselves():
    """`python bench.py --gpus 2` without a launcher starts its own ranks (here both on the one GPU, gloo) and prints
    rank 0's line."""
    d = run_bench("--gpus", "2", "--backend", "gloo", "--steps", "3", "--warmup", "1", "--width", "256", "--height", "144",
                  "--no-cpu-baseline", "--build-sizes", "", "--c5-samples", "4")
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["value"] > 0 and d["scaling"] == "weak"
    # beside the weak-scaling headline: config 5's scene as ONE frame split by tiles and by samples (strong scaling)
    for shard in ("tiles", "samples"):
        sub = d["workloads"][f"cornell_mesh_4k_16spp_{shard}"]
        assert sub["scaling"] == "strong" and sub["n_gpus"] == 2 and sub["value"] > 0 and "Cornell" in sub["workload"]
    assert "gathered" in d["workloads"]["cornell_mesh_4k_16spp_tiles"]["sharding"]
    for shard in ("tiles", "samples"):  # one frame split over the ranks, a combined frame delivered per step
        d = run_bench("--gpus", "2", "--backend", "gloo", "--shard", shard, "--workload", "cornell", "--steps", "3", "--warmup", "1",
                      "--width", "192", "--height", "128", "--no-cpu-baseline", "--build-sizes", "")
        assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["value"] > 0


def test_one_process_drives_the_group():
    """`--multi group`: no launcher, no torch.distributed; here both contexts on the one GPU (peer-copy transport) and a
    group of one device whose exchange is forced through RCCL."""
    d = run_bench("--gpus", "2", "--multi", "group", "--group-devices", "0,0", "--steps", "3", "--warmup", "1", "--width", "256",
                  "--height", "144", "--c5-samples", "4")
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["value"] > 0 and "peer" in d["config"]["transport"]
    for shard in ("tiles", "samples"):
        assert d["workloads"][f"cornell_mesh_4k_16spp_{shard}"]["value"] > 0
    env = dict(os.environ, RTHIP_GROUP_TRANSPORT="rccl")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--multi", "group", "--workload", "cornell", "--shard",
                        "samples", "--steps", "2", "--warmup", "1", "--width", "192", "--height", "128"], cwd=ROOT, capture_output=True,
                       text=True, timeout=240, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert "RCCL" in d["config"]["transport"] and d["value"] > 0
