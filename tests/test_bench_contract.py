"""The bench line the driver parses: the committed default line of this round carries every field of the contract
(metric / value / unit / n_gpus / steps / warmup / ms_per_step / higher_is_better / scaling / vs_baseline / dtype / data /
config.workload, roofline{bound, achieved, peak, unit, frac, traffic}, cpu_baseline{value, unit, cores, kind, sample}),
names BASELINE.json's metric, and `bench.py` parses its arguments without a GPU."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_default_line_has_the_contract_fields():
    d = json.load(open(os.path.join(ROOT, "profiles", "r03_bench_default_10m.json")))
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["metric"].replace(" x ", "×").replace("x", "×").split(",")[0].startswith("queries/sec")
    assert "10M" in base["metric"] and d["config"]["docs_total"] == 10_000_000 and d["n_gpus"] == 1
    assert d["unit"] == "queries/s" and d["higher_is_better"] is True and d["scaling"] == "strong" and d["vs_baseline"] is None
    assert d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - d["config"]["batch"] / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-3
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4
    assert r["traffic"] is None or (r["traffic"] > 0 and r["traffic_source"]["kernels_sha"])   # counters name the build they belong to
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0
    assert d["parity_vs_oracle"]["topk_ids_identical"] == d["parity_vs_oracle"]["queries"]


def test_traffic_file_is_keyed_by_workload():
    t = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    assert t["docs_per_gpu"] == 10_000_000 and len(t["kernels_sha"]) == 16     # bench.py drops the counters of another build
    for k in ("qc_gemm(S1)", "probe(S2)", "candidates(S3)", "approx(S4)", "select(S5)", "exact(S6)"):
        assert t[k] > 0, k


def test_bench_argument_parsing_needs_no_gpu():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "--gpus" in out.stdout and "--steps" in out.stdout and "--warmup" in out.stdout
