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
        pytest.skip("profiles/" + b.PMC_FILES[2590] + " is stale: re-run tools/pmc_traffic.sh + tools/pmc_summarise.py on a GPU box")
    assert rec["write_bytes"] == 2 * rec["shape"][0] * rec["shape"][1] or abs(rec["write_bytes"] / (2 * rec["shape"][0] * rec["shape"][1]) - 1) < 0.01


def test_committed_driver_lines_keep_the_bench_contract():
    """The lines `python3 bench.py --gpus 1 --steps 20 --warmup 5` printed on the GPU (committed under profiles/): every key of
    the bench contract, `roofline` and `cpu_baseline` objects complete, fractions consistent with their own numerators."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r03c_bench_driver_cmd_run*.json")) +
                   glob.glob(os.path.join(ROOT, "profiles", "r04*_bench_driver_cmd_run*.json")))
    assert len(files) >= 5
    for f in files:
        with open(f) as fh:
            d = json.load(fh)
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                  "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
            assert k in d, (f, k)
        assert d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 5 and d["higher_is_better"] is True
        assert d["vs_baseline"] is None and d["dtype"] == "bf16" and d["scaling"] == "weak" and "workload" in d["config"]
        B = d["config"]["global_batch"]
        assert abs(d["value"] - B / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6          # images/s = batch / step time
        r = d["roofline"]
        assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 2500.0
        assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and r["traffic"] is not None and r["traffic"] > 1e9
        reg = r["region"]
        assert abs(reg["frac"] - reg["algorithmic_tflop_per_step"] / (reg["ms_per_step"] * 1e-3) / 2500.0) < 1e-9
        assert abs(reg["algorithmic_tflop_per_step"] - 12.0 * B) / (12.0 * B) < 0.01         # the reference's 12.0 TFLOP per image
        assert reg["ms_per_step"] < d["ms_per_step"] and reg.get("absorbed_kv") is True
        c = d["cpu_baseline"]
        assert c["kind"] in ("reference", "port") and c["value"] > 0 and c["cores"] >= 1 and c["sample"]
