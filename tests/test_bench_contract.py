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
    d = json.load(open(os.path.join(ROOT, "profiles", "r04_bench_default_10m.json")))
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
    if r["traffic"] is not None:   # the second fraction prices the bytes the stage really moved (PMC), same time, same peak
        t = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
        assert r["traffic_source"]["kernels_sha"] == t["kernels_sha"] and r["traffic"] == t[r["kernel"]]
        assert abs(r["frac_physical"] - r["traffic"] / (r["ms_per_launch"] * 1e-3) / (r["peak"] * 1e9)) < 1e-3
    assert "2^16" in d["config"]["centroids_note"] and "2^19" in d["config"]["centroids_note"]   # K is this repository's choice
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0
    assert d["parity_vs_oracle"]["topk_ids_identical"] == d["parity_vs_oracle"]["queries"]


def test_every_regime_line_carries_parity_and_a_cpu_baseline():
    """every regime / variant line is a full line -- its own CPU baseline, parity at full size, the call it
    came from -- and only the default workload carries PMC traffic."""
    n = 0
    for f in ("r04_bench_regimes.jsonl", "r04_bench_variants_10m.jsonl"):
        for l in open(os.path.join(ROOT, "profiles", f)):
            d = json.loads(l)
            n += 1
            assert d["cpu_baseline"] and d["cpu_baseline"]["value"] > 0, d["name"]
            pv = d["parity_vs_oracle"]
            assert pv["queries"] >= 16 and pv["top1_identical"] == pv["queries"], d["name"]
            if d["config"].get("s1_split") or "prec1" in d["name"] or "prec3" in d["name"]:
                assert pv["topk_ids_identical"] >= 0.8 * pv["queries"], d["name"]      # reduced-precision modes: near-ties may swap
            else:
                assert pv["topk_ids_identical"] == pv["queries"], d["name"]
            assert d["evidence_call"] and d["roofline"]["traffic"] is None, d["name"]
    assert n >= 15


def test_traffic_file_is_keyed_by_workload():
    t = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    assert t["docs_per_gpu"] == 10_000_000 and len(t["kernels_sha"]) == 16     # bench.py drops the counters of another build
    for k in ("qc_gemm(S1)", "probe(S2)", "candidates(S3)", "approx(S4)", "select(S5)", "exact(S6)"):
        assert t[k] > 0, k


def test_bench_argument_parsing_needs_no_gpu():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "--gpus" in out.stdout and "--steps" in out.stdout and "--warmup" in out.stdout
