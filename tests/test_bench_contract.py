"""The bench line the driver parses: the committed default line of this round carries every field of the contract
(metric / value / unit / n_gpus / steps / warmup / ms_per_step / higher_is_better / scaling / vs_baseline / dtype / data /
config.workload, roofline{bound, achieved, peak, unit, frac, traffic}, cpu_baseline{value, unit, cores, kind, sample}),
names BASELINE.json's metric, and `bench.py` parses its arguments without a GPU."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _latest(name):
    """profiles/rNN_<name> of the newest round that has one."""
    for r in range(9, 0, -1):
        p = os.path.join(ROOT, "profiles", f"r{r:02d}_{name}")
        if os.path.exists(p):
            return p, r
    raise FileNotFoundError(name)


def test_default_line_has_the_contract_fields():
    path, rnd = _latest("bench_default_10m.json")
    d = json.load(open(path))
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
        assert r["traffic_source"]["kernels_sha"] == t["kernels_sha"]
        if rnd >= 5:   # round 4's line was printed from a traffic file whose S4 lacked approx_hotp_kernel (4.28 of 8.56 GB): VERDICT r4
            assert r["traffic"] == t[r["kernel"]]
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
    for f in ("bench_regimes.jsonl", "bench_variants_10m.jsonl"):
        for l in open(_latest(f)[0]):
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


def _stage_map():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import stage_map
    return stage_map


def test_stage_map_covers_every_launch_site():
    """the S4 kernel was renamed and dropped out of the traffic table.  Every kernel np_search.hip launches
    has a stage (or is listed as outside the batch pass), and every listed name is still a kernel of np_kernels.h."""
    sm = _stage_map()
    launched, defined = sm.launched_kernels(), sm.defined_kernels()
    assert len(launched) > 40 and launched <= defined
    assert launched - set(sm.STAGE) - set(sm.OTHER) == set()
    assert (set(sm.STAGE) | set(sm.OTHER)) - defined == set()
    assert set(sm.STAGE.values()) == set(sm.STAGES)
    assert sm.kernel_of("void np::approx_hotp_kernel<32, unsigned short, 2, 2, 4, 1>(unsigned int const*)") == "approx_hotp_kernel"


def test_stage_traffic_is_the_sum_over_all_kernels_of_the_stage():
    """traffic[stage] == sum of per_kernel bytes over EVERY kernel the stage map assigns to the stage, and every kernel of the
    file that moved bytes inside a batch is either mapped or a one-off of the index build."""
    sm = _stage_map()
    t = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    pk = t["per_kernel"]
    for st in sm.STAGES:
        ks = [k for k in pk if sm.STAGE.get(k) == st]
        assert t[st] == sum(pk[k]["bytes_per_batch"] for k in ks), st
        for k in ks:
            e = pk[k]
            want = t["fetch_factor"] * e.get("FETCH_SIZE_KiB_per_batch", 0) * 1024 + e.get("WRITE_SIZE_KiB_per_batch", 0) * 1024
            assert abs(e["bytes_per_batch"] - want) <= 0.001 * want + 2048, k
    s4 = {k for k in pk if sm.STAGE.get(k) == sm.S4}
    assert {"approx_hotp_kernel", "approx_ub_kernel", "approx_xcd_kernel", "ub_cut_kernel"} <= s4   # the default line's S4 kernels
    assert pk["approx_hotp_kernel"]["bytes_per_batch"] > 0.3 * t[sm.S4]                              # ... dominant one included
    batch = sum(t[st] for st in sm.STAGES)
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_traffic
    build = make_traffic.build_kernels()
    for k, e in pk.items():
        if k not in sm.STAGE and k not in sm.OTHER and k not in build:
            assert e["bytes_per_batch"] <= 0.01 * batch, k


def test_bench_argument_parsing_needs_no_gpu():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "--gpus" in out.stdout and "--steps" in out.stdout and "--warmup" in out.stdout


def test_make_traffic_refuses_a_kernel_without_a_stage(tmp_path):
    """The failure of round 4, as a test: a library kernel that moves more than 1 % of a batch's bytes and has no stage in
    tools/stage_map.py makes tools/make_traffic.py fail instead of silently dropping out of the roofline."""
    def write(pass_dir, counter, rows):
        os.makedirs(pass_dir)
        with open(os.path.join(pass_dir, "p_counter_collection.csv"), "w") as f:
            f.write("Kernel_Name,Counter_Name,Counter_Value\n")
            for k, v in rows:
                f.write(f'"void np::{k}<32>(int)",{counter},{v}\n')
    rows = [("prep_queries_kernel", 100), ("qc_gemm_kernel", 1000), ("approx_hotp_kernel", 5000), ("approx_hotq_kernel", 4000)]
    write(str(tmp_path / "bad" / "p1"), "FETCH_SIZE", rows)
    write(str(tmp_path / "bad" / "p2"), "WRITE_SIZE", rows)
    tool = os.path.join(ROOT, "tools", "make_traffic.py")
    out = subprocess.run([sys.executable, tool, str(tmp_path / "bad"), "10000000", str(tmp_path / "t.json")], capture_output=True, text=True)
    assert out.returncode == 2 and "approx_hotq_kernel" in out.stderr, (out.returncode, out.stderr[-400:])
    write(str(tmp_path / "good" / "p1"), "FETCH_SIZE", rows[:3])
    write(str(tmp_path / "good" / "p2"), "WRITE_SIZE", rows[:3])
    out = subprocess.run([sys.executable, tool, str(tmp_path / "good"), "10000000", str(tmp_path / "t.json")], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-400:]
    t = json.load(open(tmp_path / "t.json"))
    assert t["approx(S4)"] == (2 * 5000 + 5000) * 1024 and t["qc_gemm(S1)"] == (2 * 1000 + 1000 + 2 * 100 + 100) * 1024


def test_gpus_n_without_a_launcher_never_runs_one_rank_silently():
    """`python bench.py --gpus N` outside torch.distributed.run re-executes itself under the launcher, and where the box
    has fewer than N devices (this container has none) it FAILS, naming the count -- it used to run one rank labelled n_gpus: 1."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    import next_plaid_amd as npa
    if npa.device_count() < 2:
        assert out.returncode != 0 and "--gpus 2" in out.stderr and "gfx950 device" in out.stderr, out.stderr[-2000:]
        assert not [l for l in out.stdout.splitlines() if l.startswith("{")]
    # a launcher whose world size disagrees with --gpus is an error too, whatever the sizes
    env["WORLD_SIZE"], env["RANK"], env["LOCAL_RANK"] = "1", "0", "0"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=300,
                         cwd=ROOT, env=env)
    assert out.returncode != 0 and "WORLD_SIZE=1" in out.stderr
