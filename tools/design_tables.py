#!/usr/bin/env python3
"""Installs the evidence of a final GPU call under profiles/ and regenerates the measured tables of DESIGN.md
(between <!-- BEGIN:x --> / <!-- END:x --> markers) from profiles/r06_*.

  design_tables.py install <gpurun_out tag> [--merge]   copy gpurun_out/<tag>/... to profiles/r06_* (names below); --merge keeps
                                                the committed lines the call did not re-measure
  design_tables.py                              regenerate the tables from profiles/r06_*
"""
import json
import os
import re
import shutil
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
P = os.path.join(ROOT, "profiles") + "/"
R = "r06"

GROUPS = {
    f"{R}_bench_shard_sizes.jsonl": ["shard_5m", "shard_2500k", "shard_1250k", "shard_1250k_rccl"],
    f"{R}_bench_variants_10m.jsonl": ["prec0", "prec1", "prec3", "single_level_10m"],
    f"{R}_bench_regimes.jsonl": ["api_default", "tcs_none", "dist05", "dist08", "lq48_10m", "nfs8192_10m", "colgrep_10m", "k19_10m",
                                 "k19_split_10m", "k19_rest", "k19_rest_lq48_nfs8192", "c3_np32", "c4_k18_12500k", "k20_2500k", "dist05_gain", "dist08_gain"],
}
SINGLES = {f"{R}_bench_default_10m.json": "default_10m", f"{R}_bench_1m.json": "1m", f"{R}_bench_c4_shard_12500k.json": "c4_shard_12500k",
           f"{R}_bench_disk_1m.json": "disk1m"}


def install(tag, merge=False):
    """merge: a later, smaller call re-measured some lines -- replace those, keep the rest of the committed evidence."""
    src = os.path.join(ROOT, "gpurun_out", tag)
    for dst, name in SINGLES.items():
        pth = os.path.join(src, f"b_{name}.json")
        if merge and (not os.path.exists(pth) or os.path.getsize(pth) == 0):   # (an empty file: the run failed, e.g. out of memory)
            continue
        d = json.load(open(os.path.join(src, f"b_{name}.json")))
        if name != "default_10m" and d.get("roofline", {}).get("traffic") is not None:
            d["roofline"]["traffic"] = d["roofline"]["traffic_source"] = d["roofline"]["frac_physical"] = None
        json.dump(d, open(P + dst, "w"))
        open(P + dst, "a").write("\n")
    for dst, names in GROUPS.items():
        old = {}
        if merge and os.path.exists(P + dst):
            old = {json.loads(l)["name"]: l for l in open(P + dst) if l.strip()}
        with open(P + dst, "w") as f:
            for n in names:
                p = os.path.join(src, f"b_{n}.json")
                if not os.path.exists(p) or os.path.getsize(p) == 0:
                    if n in old:
                        f.write(old[n] if old[n].endswith("\n") else old[n] + "\n")
                    else:
                        print("missing", n)
                    continue
                d = json.load(open(p))
                d["name"] = n
                d["evidence_call"] = tag
                # the PMC traffic figure belongs to the DEFAULT workload only (bench.py now checks that itself)
                if d.get("roofline", {}).get("traffic") is not None:
                    d["roofline"]["traffic"] = d["roofline"]["traffic_source"] = d["roofline"]["frac_physical"] = None
                f.write(json.dumps(d) + "\n")
    for a, b in (("stats_d10m.md", f"{R}_kernel_stats_10m.md"), ("stats_d10m.csv", f"{R}_kernel_stats_10m.csv"),
                 ("stats_d1m.md", f"{R}_kernel_stats_1m.md"), ("stats_d1m.csv", f"{R}_kernel_stats_1m.csv"),
                 ("pmc_d10m.md", f"{R}_pmc_10m.md"), ("traffic_d10m.json", "traffic.json"), ("host.txt", f"{R}_host.txt")):
        if os.path.exists(os.path.join(src, a)):
            shutil.copy(os.path.join(src, a), P + b)
    tp = P + "traffic.json"
    if os.path.exists(os.path.join(src, "traffic_d10m.json")):   # the GPU box has no .git: name the commit the call ran at here
        import subprocess
        t = json.load(open(tp))
        if not t.get("commit"):
            t["commit"] = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
            json.dump(t, open(tp, "w"), indent=1)
    if os.path.exists(os.path.join(src, "pmcd", "derived.txt")):
        shutil.copy(os.path.join(src, "pmcd", "derived.txt"), P + f"{R}_pmc_derived_10m.txt")
    if not os.path.exists(os.path.join(src, "test_all.log")) or "-k" in open(os.path.join(src, "test_all.log")).read()[:0]:
        return
    if " deselected" in open(os.path.join(src, "test_all.log")).read() and "passed" not in open(os.path.join(src, "test_all.log")).read() \
            and os.path.exists(P + f"{R}_test_gpu.log"):
        return   # a partial / superseded test run never replaces the full one
    with open(P + f"{R}_test_gpu.log", "w") as f:
        for a in ("test_all.log", "smoke.log"):
            if os.path.exists(os.path.join(src, a)):
                f.write(open(os.path.join(src, a)).read())


def load(n):
    return json.load(open(P + n))


def lines(n):
    return {json.loads(l)["name"]: json.loads(l) for l in open(P + n) if l.strip()}


def stage_cells(s):
    return (f"{s['ms_centroid']:.2f} | {s['ms_probe']:.2f} | {s['ms_candidates']:.2f} | {s['ms_approx']:.2f} | {s['ms_select']:.2f} | "
            f"{s['ms_exact']:.2f}")


def tables():
    d10, d1, tr = load(f"{R}_bench_default_10m.json"), load(f"{R}_bench_1m.json"), load("traffic.json")
    rows = [("S1 `qc_gemm` (+ hot prep)", "qc_gemm(S1)", "ms_centroid"), ("S2 probe", "probe(S2)", "ms_probe"),
            ("S3 candidates (+ hot levels / plane rows)", "candidates(S3)", "ms_candidates"), ("S4 two-level filter + exact survivors", "approx(S4)", "ms_approx"),
            ("S5 select", "select(S5)", "ms_select"), ("S6 exact (prec 2)", "exact(S6)", "ms_exact")]
    t = ["| stage | 10 M docs: ms / batch | achieved vs §8(d) algorithmic roofline | 1 M docs: ms / batch | achieved |", "|---|---:|---|---:|---|"]
    for name, k, ms in rows:
        a, b = d10["roofline"]["all"][k], d1["roofline"]["all"][k]
        t.append(f"| {name} | {d10['stages'][ms]:.3f} | {a['achieved']:.0f} {a['unit']} = {100*a['frac']:.1f} % of {a['bound'].upper()} peak | "
                 f"{d1['stages'][ms]:.3f} | {b['achieved']:.0f} {b['unit']} = {100*b['frac']:.1f} % |")
    t.append(f"| S7 top-k | {d10['stages']['ms_topk']:.3f} | | {d1['stages']['ms_topk']:.3f} | |")
    t.append(f"| **one batch alone (p50)** | **{d10['p50_batch_latency_ms']:.2f}** | | **{d1['p50_batch_latency_ms']:.2f}** | |")
    t.append(f"| **sustained, 3 streams** | **{d10['ms_per_step']:.2f} → {d10['value']:.0f} queries/s** | host buffers in / out, 3 host threads: {d10['value_pcie_inclusive']:.0f} "
             f"({100*d10['pcie_inclusive']['ratio_to_value']:.1f} %) | "
             f"**{d1['ms_per_step']:.2f} → {d1['value']:.0f} queries/s** | {d1['value_pcie_inclusive']:.0f} ({100*d1['pcie_inclusive']['ratio_to_value']:.1f} %) |")
    s10, s1 = d10["stages"], d1["stages"]
    cb10, cb1 = d10["cpu_baseline"], d1["cpu_baseline"]
    t += ["", f"Per batch of 64 queries at 10 M documents: {s10['n_cells']:.0f} probed cells, {s10['n_ivf_ids']/1e6:.1f} M posting entries, "
              f"{s10['n_candidates']/1e6:.1f} M candidates ({s10['n_candidates']/64/1e3:.0f} k / query), {s10['n_cand_tokens']/1e9:.2f} G candidate tokens = "
              f"{s10['n_cand_dcodes']/1e6:.0f} M distinct (document, code) pairs scanned by the hot level → {s10['n_cand_codes']/1e6:.0f} M u8 table rows gathered "
              f"(both levels), {s10['n_level2']/1e6:.2f} M documents at the exact level, {s10['n_survivors']/1e3:.0f} k survivors, "
              f"{s10['n_exact_docs']:.0f} docs / {s10['n_exact_tokens']/1e6:.2f} M tokens exact-scored. At 1 M: {s1['n_candidates']/1e6:.2f} M candidates, "
              f"{s1['n_cand_codes']/1e6:.1f} M rows, {s1['n_survivors']/1e3:.0f} k survivors. Index build in HBM {d10['index_build_s']:.1f} s (10 M), "
              f"{d1['index_build_s']:.2f} s (1 M); {d10['hbm_bytes_per_token']:.1f} B per token. CPU baseline (oracle C restatement, {cb10['cores']} threads, "
              f"{cb10['cpu_model']}): {cb10['value']:.1f} queries/s at 10 M, {cb1['value']:.1f} at 1 M. `roofline.traffic` (PMC, 10 M): "
              f"S4 {tr['approx(S4)']/1e9:.2f} GB, S6 {tr['exact(S6)']/1e9:.2f} GB, S3 {tr['candidates(S3)']/1e9:.2f} GB per batch.", "",
          f"Round 5 → round 6 (this path is untouched by the round's work, which went into the regimes without a threshold): 10 M documents 20.2 k → "
          f"{d10['value']/1e3:.1f} k queries/s (p50 3.62 → {d10['p50_batch_latency_ms']:.2f} ms, S4 2.19 → "
          f"{s10['ms_approx']:.2f} ms); 1 M documents 41.4 k → {d1['value']/1e3:.1f} k queries/s (S4 0.72 → {s1['ms_approx']:.2f} ms).  "
          f"The bench line's `roofline.frac` prices the contract's algorithmic bytes ({d10['roofline']['frac']:.2f} for S4); the bytes the stage "
          f"really moved (PMC, every kernel of the stage) give `roofline.frac_physical` = {d10['roofline'].get('frac_physical')}; the dominant kernel "
          f"alone (`approx_hotp_kernel`, {d10['roofline']['dominant_kernel']['ms_per_launch']:.3f} ms by its own HIP events): "
          f"{d10['roofline']['dominant_kernel']['traffic']/1e9:.2f} GB → `frac_physical` {d10['roofline']['dominant_kernel']['frac_physical']}."]
    measured = "\n".join(t)

    reg = lines(f"{R}_bench_regimes.jsonl")
    var = lines(f"{R}_bench_variants_10m.jsonl")

    def par(d):
        pv = d.get("parity_vs_oracle")
        return "—" if not pv else f"{pv['topk_ids_identical']}/{pv['queries']}"

    dist = ["| distinct codes per token (`--rand256`) | candidates / query | table rows / batch | queries/s | S3 ms | S4 ms | CPU oracle q/s | top-10 = oracle |", "|---|---:|---:|---:|---:|---:|---:|---:|"]
    for lab, d in (("0.23 (51, the default corpus)", d10), ("0.50 (121)", reg.get("dist05")),
                   ("0.50 with the zeroth level on (`NP_S3_GAIN=2`: where the run / skip rule arrives after its first trial)", reg.get("dist05_gain")),
                   ("0.80 (200)", reg.get("dist08")), ("0.80 with the zeroth level on", reg.get("dist08_gain"))):
        if not d:
            continue
        s = d["stages"]
        cpu = (d.get("cpu_baseline") or {}).get("value")
        dist.append(f"| {lab} | {s['n_candidates']/64/1e3:.0f} k | {s['n_cand_codes']/1e6:.0f} M | {d['value']:.0f} | {s['ms_candidates']:.2f} | {s['ms_approx']:.2f} | "
                    f"{'—' if not cpu else f'{cpu:.1f}'} | {par(d)} |")
    dist = "\n".join(dist)

    rg = ["| line (10 M docs × 300 tok unless stated) | queries/s | p50 ms | S1 | S2 | S3 | S4 | S5 | S6 | CPU oracle q/s | top-10 = oracle |",
          "|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|"]
    labs = [("api_default", "**the REST API's default**: `centroid_score_threshold = None`, `n_ivf_probe = 8` (`next-plaid-api/src/models.rs:281-284`, `handlers/search.rs:211-217`)"),
            ("tcs_none", "`centroid_score_threshold = None`, nprobe 32"),
            ("lq48_10m", "48-token queries (ONNX encoder default, `next-plaid-onnx/src/lib.rs:628-630`)"),
            ("nfs8192_10m", "`n_full_scores = 8192` (ColGREP, `colgrep/src/index/mod.rs:771-777`)"),
            ("colgrep_10m", "**ColGREP's defaults together**: 48-token queries + `n_full_scores = 8192` + nprobe 8 (`colgrep/src/index/mod.rs:777-819`)"),
            ("k19_10m", "K = 2¹⁹ (the crate's k-means heuristic at this size, `kmeans.rs:303-309`): batched path, bit-exact S1-S5"),
            ("k19_split_10m", "K = 2¹⁹ with the opt-in split-bf16 S1 (`s1_split`, no mat-vec re-scoring)"),
            ("k19_rest", "**the crate-natural regime**: K = 2¹⁹ (`kmeans.rs:303-309`) × `centroid_score_threshold = None` × `n_ivf_probe = 8` (`models.rs:271-284`): batched probe + mat-vec re-scoring + no threshold, bit-exact S1-S5"),
            ("k19_rest_lq48_nfs8192", "the same with 48-token queries and `n_full_scores = 8192`"),
            ("c3_np32", "config 3 shape: 8 841 823 docs, clipped LogNormal lengths (mean 73, max 180), K = 2¹⁸, nbits 2, nprobe 32")]
    for k, lab in labs:
        d = reg.get(k)
        if not d:
            continue
        cpu = (d.get("cpu_baseline") or {}).get("value")
        rg.append(f"| {lab} | {d['value']:.0f} | {d['p50_batch_latency_ms']:.2f} | {stage_cells(d['stages'])} | {'—' if not cpu else f'{cpu:.1f}'} | {par(d)} |")
    c4 = load(f"{R}_bench_c4_shard_12500k.json")
    cpu = (c4.get("cpu_baseline") or {}).get("value")
    rg.append(f"| config 4's shard: 12.5 M docs on one GPU ({c4['hbm_index_bytes']/1e9:.1f} GB, {c4['hbm_bytes_per_token']:.1f} B/token) | {c4['value']:.0f} | "
              f"{c4['p50_batch_latency_ms']:.2f} | {stage_cells(c4['stages'])} | {'—' if not cpu else f'{cpu:.1f}'} | {par(c4)} |")
    d = reg.get("c4_k18_12500k")
    if d:
        cpu = (d.get("cpu_baseline") or {}).get("value")
        rg.append(f"| config 4's shard at K = 2¹⁸ (SURVEY §8(d) C4: batched probe, u32 code lists, {d['hbm_index_bytes']/1e9:.1f} GB) | {d['value']:.0f} | "
                  f"{d['p50_batch_latency_ms']:.2f} | {stage_cells(d['stages'])} | {'—' if not cpu else f'{cpu:.1f}'} | {par(d)} |")
    d = reg.get("k20_2500k")
    if d:
        cpu = (d.get("cpu_baseline") or {}).get("value")
        rg.append(f"| K = 2²⁰ on 2.5 M docs (config 4's K range on a corpus that leaves room for the 8.6 GB of f32 scores per batch; single-level filter: "
                  f"the hot bitmap of K > 2¹⁹ does not fit LDS) | {d['value']:.0f} | {d['p50_batch_latency_ms']:.2f} | {stage_cells(d['stages'])} | "
                  f"{'—' if not cpu else f'{cpu:.1f}'} | {par(d)} |")
    dk = load(f"{R}_bench_disk_1m.json")
    do = dk["disk_open"]
    cpu = (dk.get("cpu_baseline") or {}).get("value")
    rg.append(f"| 1 M docs opened from its index DIRECTORY ({do['bytes']/1e9:.1f} GB in {do['files']} files: open {do['open_warm_s']:.2f} s warm = "
              f"{do['open_warm_gbs']:.0f} GB/s, {do['open_cold_s']} s cold) | {dk['value']:.0f} | {dk['p50_batch_latency_ms']:.2f} | {stage_cells(dk['stages'])} | "
              f"{'—' if not cpu else f'{cpu:.1f}'} | {par(dk)} |")
    rg = "\n".join(rg)

    c5 = ["| 10 M docs, B = 64 | queries/s | p50 ms | S4 ms | S6 ms | max rel. score error vs oracle | top-10 ids identical |", "|---|---:|---:|---:|---:|---:|---:|"]
    for lab, d in (("precision 2 (default; split-bf16 QC-reuse)", d10), ("precision 0 (exact-f32 MFMA everywhere)", var.get("prec0")),
                   ("precision 1 (bf16 QC-reuse)", var.get("prec1")), ("precision 3 (plain bf16 MaxSim)", var.get("prec3")),
                   ("precision 2, single-level filter (`--hot 0`)", var.get("single_level_10m"))):
        if not d:
            continue
        pv = d["parity_vs_oracle"]
        c5.append(f"| {lab} | {d['value']:.0f} | {d['p50_batch_latency_ms']:.2f} | {d['stages']['ms_approx']:.2f} | {d['stages']['ms_exact']:.2f} | "
                  f"{pv['max_rel_score_err']:.1e} | {pv['topk_ids_identical']}/{pv['queries']} |")
    c5 = "\n".join(c5)

    sh = lines(f"{R}_bench_shard_sizes.jsonl")
    st = ["| documents on the GPU (= one rank of) | queries/s | p50 ms | S1 | S2 | S3 | S4 | S5 | S6 | CPU oracle q/s | top-10 = oracle |",
          "|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|"]
    for lab, d in (("10 M (1 GPU, the N=1 line)", d10), ("5 M (2-way)", sh.get("shard_5m")), ("2.5 M (4-way)", sh.get("shard_2500k")),
                   ("1.25 M (8-way)", sh.get("shard_1250k")), ("1.25 M, through `np_hip_search_batch_sharded` (RCCL, world 1)", sh.get("shard_1250k_rccl"))):
        if d:
            cpu = (d.get("cpu_baseline") or {}).get("value")
            st.append(f"| {lab} | {d['value']:.0f} | {d['p50_batch_latency_ms']:.2f} | {stage_cells(d['stages'])} | {'—' if not cpu else f'{cpu:.1f}'} | {par(d)} |")
    st += ["", "(Single-GPU runs of a corpus of that size: S6 is the full `n_sel` here, whereas a rank of the real split exact-scores only its share of the global cut.)"]
    st = "\n".join(st)

    hs = ["| S4 filter, 10 M documents | table rows gathered / batch | documents at the exact level | S4 ms | queries/s |", "|---|---:|---:|---:|---:|"]
    r4 = {}
    try:
        r4 = {json.loads(l)["name"]: json.loads(l) for l in open(P + "r04_bench_variants_10m.jsonl") if l.strip()}
        r4["default"] = json.load(open(P + "r04_bench_default_10m.json"))
    except OSError:
        pass
    for lab, d in (("round 4 (table over [-s, s]): single level, `NP_S4_HOT=0`", r4.get("single_level_10m")),
                   ("round 4: two levels, first level as byte maxima (round 3's kernel, `NP_S4_PLANES=0`; 10 % hot)", r4.get("planes0_10m")),
                   ("round 4: two levels, bit planes (6 % hot), exact level gathers every row (`NP_S4_WARM=1000`)", r4.get("warm1000_10m")),
                   ("round 4: two levels, bit planes + floored exact level (its default)", r4.get("default")),
                   ("**round 5** (table over the positive scores): single level (`--hot 0`)", var.get("single_level_10m")),
                   ("**round 5: two levels, bit planes (6 % hot) + floored exact level: default**", d10)):
        if d:
            x = d["stages"]
            hs.append(f"| {lab} | {x['n_cand_codes']/1e6:.0f} M | {(x['n_level2'] or x['n_candidates'])/1e6:.2f} M | {x['ms_approx']:.2f} | {d['value']:.0f} |")
    hs = "\n".join(hs)

    hy = ["| 8 GPUs as | per-GPU line | projected queries/s | batch latency |", "|---|---|---:|---:|"]
    for S, k, lab in ((8, "shard_1250k", "1.25 M"), (4, "shard_2500k", "2.5 M"), (2, "shard_5m", "5 M"), (1, None, "10 M")):
        d = d10 if k is None else sh.get(k)
        if d:
            hy.append(f"| {S} shard(s) × {8 // S} replica group(s) | {lab} docs: {d['value']:.0f} q/s | {8 // S * d['value']:.0f} | ≈ {d['p50_batch_latency_ms']:.1f} ms + collectives |")
    hy += ["", "(Upper bounds: the collectives of S > 1 are not in the one-GPU lines; S = 1 has none. Pure sharding minimises latency, pure replication maximises throughput.)"]
    hy = "\n".join(hy)

    path = os.path.join(ROOT, "DESIGN.md")
    x = open(path).read()
    for tag, body in (("measured", measured), ("hotsweep", hs), ("distinct", dist), ("regimes", rg), ("c5", c5), ("shards", st), ("hybrid", hy)):
        x, n = re.subn(rf"<!-- BEGIN:{tag} -->\n.*?<!-- END:{tag} -->", lambda m: f"<!-- BEGIN:{tag} -->\n{body}\n<!-- END:{tag} -->", x, flags=re.S)
        assert n == 1, tag
    open(path, "w").write(x)
    print("tables written")


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "install":
        install(sys.argv[2], merge="--merge" in sys.argv)
    tables()
