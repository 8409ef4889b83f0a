"""bench.py::pmc_record — `roofline.traffic` comes from a committed PMC summary, but only while the kernel sources it was
taken on are unchanged (sha256 in the summary); a stale constant must turn into null, not into a wrong number."""
import hashlib
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_summary_is_dropped_when_the_sources_change(tmp_path, monkeypatch):
    b = load_bench()
    (tmp_path / "profiles").mkdir()
    src = tmp_path / "k.hip"
    src.write_text("kernel v1\n")
    rec = {"shape": [1, 2, 3], "algorithmic_bytes_per_launch": 10, "hbm_bytes_per_launch": 25.0, "sources": ["k.hip"],
           "sources_sha256": hashlib.sha256(b"kernel v1\n").hexdigest()}
    (tmp_path / "profiles" / "x.json").write_text(json.dumps(rec))
    monkeypatch.setattr(b, "ROOT", str(tmp_path))
    monkeypatch.setattr(b, "PMC_FILES", {2590: "x.json"})
    assert b.pmc_traffic(2590) == 25.0 and b.pmc_note(2590)["shape_MNK"] == [1, 2, 3]
    assert b.pmc_traffic(256) is None                      # no summary for that kernel
    src.write_text("kernel v2\n")
    assert b.pmc_traffic(2590) is None and b.pmc_note(2590) is None
    src.unlink()
    assert b.pmc_traffic(2590) is None


def test_committed_summary_matches_the_committed_sources():
    b = load_bench()
    import pytest
    rec = b.pmc_record(2590)
    if rec is None:   # not a failure (bench.py then reports traffic: null), but say so
        pytest.skip("profiles/r02_pmc_gemm_p5.json is stale: re-run tools/pmc_traffic.sh + tools/pmc_summarise.py on a GPU box")
    assert rec["write_bytes"] == 2 * rec["shape"][0] * rec["shape"][1] or abs(rec["write_bytes"] / (2 * rec["shape"][0] * rec["shape"][1]) - 1) < 0.01
