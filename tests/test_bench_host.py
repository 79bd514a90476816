"""Host-side logic of bench.py that needs no GPU: the recorded counter table is only quoted for the revision of the kernel
sources it was recorded on."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import bench  # noqa: E402


def _with_table(monkeypatch, table):
    now = bench.source_hashes()
    stamp = table.get("_recorded_on", {}).get("sources")
    changed = sorted(now) if not stamp else sorted(f for f in now if stamp.get(f) != now[f])
    monkeypatch.setattr(bench, "_RECORDED", {"table": table, "changed": changed})


def test_blob_hash_is_git_hash_object(tmp_path):
    p = tmp_path / "x.txt"
    p.write_bytes(b"hello\n")
    assert bench.blob_hash(str(p)) == "ce013625030ba8dba906f756967f9e9ca394464a"  # `git hash-object` of "hello\n"


def test_counters_are_quoted_only_for_the_sources_they_were_recorded_on(monkeypatch):
    rec = {"hbm_bytes_per_launch": 1000, "valu_insts_per_launch": 77}
    fresh = {"_recorded_on": {"sources": bench.source_hashes(), "when": "today"}, "bunny": {"cast_static_kernel": rec}}
    _with_table(monkeypatch, fresh)
    t, v, why = bench.recorded_counters("bunny", "cast_static_kernel")
    assert (t, v) == (1000, 77) and "sources unchanged" in why
    assert bench.recorded_counters("bunny", "cast_kernel (persistent, refill 16)")[0] is None
    # a renderer source changed since: the renderer's counters are withheld, a builder kernel's are not
    stale = json.loads(json.dumps(fresh))
    stale["_recorded_on"]["sources"]["trace.hip"] = "0" * 40
    _with_table(monkeypatch, stale)
    t, v, why = bench.recorded_counters("bunny", "cast_static_kernel")
    assert t is None and v is None and "trace.hip" in why
    assert bench.recorded_counters("bunny", "cast_static_kernel", bench.BUILD_SOURCES)[0] == 1000
    # a table without a stamp is stale as a whole
    _with_table(monkeypatch, {"bunny": {"cast_static_kernel": rec}})
    assert bench.recorded_counters("bunny", "cast_static_kernel")[0] is None


def test_no_hbm_fraction_above_one():
    c = bench.ceilings(algo_bytes=2e9, traffic=None, valu_insts=None, kernel_ms=0.1, ms_per_step=0.1, n_streams=1, why_no_counters="stale")
    assert c["hbm_algorithmic_frac"] is None and "cache-resident" in c["hbm_algorithmic_note"] and c["bounding"] == "unknown: stale"
    c = bench.ceilings(algo_bytes=4e8, traffic=5e8, valu_insts=1e6, kernel_ms=0.1, ms_per_step=0.1, n_streams=1)
    assert c["hbm_algorithmic_frac"] == 0.5 and c["bounding"] == "hbm" and c["frac_of_bounding"] <= 1.0
