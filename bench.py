#!/usr/bin/env python3
"""bench.py -- queries/sec of the MI355X PLAID search path on BASELINE.json's config 2.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is ONE pass of the hot path (S1 centroid scoring -> S7 top-k) over one batch of 64
queries whose embeddings are already resident in HBM; the index is resident too (built in HBM
by the seeded generator of next_plaid_amd/synth.py).  value = queries / second, whole job.

Workload (config.workload): 1M docs x 300 tokens x d=128, nbits=4, K=2^16 centroids, nprobe=32,
batch 64 x 32-token queries, n_full_scores=4096 (1024 exact re-ranks), t_cs=0.4, top_k=10.
With N GPUs every rank holds its own 1M-document shard (weak scaling: the corpus is N x 1M docs),
all ranks answer the same query batch and exchange rank keys / top-k over RCCL (dist.py).

Rank 0 prints ONE JSON line.  Extra objects: "roofline" (dominant kernel: achieved algorithmic
bytes or flops per launch / measured launch duration vs the chip peak), "cpu_baseline" (the
oracle restatement of the reference CPU path timed on this box's host cores on a bounded
sample of the same workload), "stages" (per-stage ms and work counters per batch).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "next-plaid_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
MFMA_F32_PEAK_TF = 157.3     # exact-f32 MFMA (v_mfma_f32_32x32x2_f32)
MFMA_BF16_PEAK_TF = 2500.0   # dense bf16 MFMA


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--docs-per-gpu", type=int, default=1_000_000)
    ap.add_argument("--doc-len", type=int, default=300)
    ap.add_argument("--centroids", type=int, default=65536)
    ap.add_argument("--nprobe", type=int, default=32)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--query-tokens", type=int, default=32)
    ap.add_argument("--n-full-scores", type=int, default=4096)
    ap.add_argument("--top-k", type=int, default=10)
    ap.add_argument("--threshold", type=float, default=0.4, help="centroid_score_threshold; <0 = None")
    ap.add_argument("--precision", type=int, default=2,
                    help="exact MaxSim arithmetic: 0 exact-f32 MFMA, 1 QC-reuse bf16, 2 QC-reuse split-bf16 (f32-class), 3 plain bf16")
    ap.add_argument("--cpu-queries", type=int, default=64, help="queries of the CPU-oracle leg (0 = skip)")
    ap.add_argument("--query-batches", type=int, default=4)
    ap.add_argument("--force-dist", action="store_true",
                    help="run the sharded RCCL protocol even with one rank (exercises the N > 1 code path on a 1-GPU box)")
    ap.add_argument("--streams", type=int, default=3,
                    help="HIP streams the steps are issued on round-robin (each step = one full batch pass; 2 lets the "
                         "small launch-bound kernels of one batch overlap the memory-bound ones of the next)")
    return ap.parse_args()


def main():
    a = parse()
    import torch
    import next_plaid_amd as npa
    from next_plaid_amd import api, synth

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.gpus != world and world > 1:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    if npa.device_count() < 1:
        raise SystemExit("bench.py needs a gfx950 GPU (the HIP path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or a.force_dist
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    # ---- corpus: generated in HBM, one shard per rank -----------------------------------------
    spec = synth.SynthSpec(num_docs=a.docs_per_gpu * world, num_centroids=a.centroids, dim=128, nbits=4,
                           doc_len_min=a.doc_len, doc_len_max=a.doc_len, seed=1236)
    cen = synth.centroids(spec)
    t0 = time.time()
    ix = npa.MmapIndex.synth(spec, centroids=cen, device=local_rank, shard_rank=rank, shard_count=world,
                             max_batch=a.batch, n_contexts=max(1, a.streams))
    t_build = time.time() - t0
    thr = None if a.threshold < 0 else a.threshold
    prm = npa.SearchParameters(n_full_scores=a.n_full_scores, top_k=a.top_k, n_ivf_probe=a.nprobe,
                               centroid_score_threshold=thr, precision=a.precision)

    # ---- queries: resident in HBM before the timed region ----------------------------------------
    nq = a.batch * a.query_batches
    qs, src = synth.make_queries(spec, nq, n_tokens=a.query_tokens, cen=cen)
    off = np.arange(a.batch + 1, dtype=np.int32) * a.query_tokens
    nstr = max(1, a.streams)
    streams = [torch.cuda.Stream(dev) for _ in range(nstr)]
    stream = streams[0]
    with torch.cuda.stream(stream):
        dq = [torch.from_numpy(np.concatenate(qs[i * a.batch:(i + 1) * a.batch], 0)).to(dev) for i in range(a.query_batches)]
        doff = torch.from_numpy(off).to(dev)
        o_ids = [torch.zeros((a.batch, max(a.top_k, 1)), dtype=torch.int64, device=dev) for _ in range(nstr)]
        o_sc = [torch.zeros((a.batch, max(a.top_k, 1)), dtype=torch.float32, device=dev) for _ in range(nstr)]
        o_cnt = [torch.zeros(a.batch, dtype=torch.int32, device=dev) for _ in range(nstr)]
    torch.cuda.synchronize(dev)
    L = api.lib()
    cp = prm._c()

    if use_dist:
        # one searcher per stream, each with its own process group (= its own RCCL communicator), so the two
        # small all-gathers of batch i overlap the kernels of batch i+1; every rank issues the same round-robin order
        from next_plaid_amd.dist import HipShardBackend, ShardedSearcher
        groups = [dist.new_group(ranks=list(range(world))) for _ in range(nstr)]
        sss = [ShardedSearcher([HipShardBackend(ix, stream=streams[s])], use_dist=True, group=groups[s])
               for s in range(nstr)]
        ss = sss[0]

        def step(i):
            return sss[i % nstr].search_batch_device(dq[i % a.query_batches], doff, off, prm)
    else:
        def step(i):
            s = i % nstr
            api._check(L.np_hip_search_batch_device(
                ix._h, C.c_void_p(dq[i % a.query_batches].data_ptr()), C.c_void_p(doff.data_ptr()),
                off.ctypes.data_as(C.c_void_p), a.batch, 128, C.byref(cp), None, -1, C.c_void_p(o_ids[s].data_ptr()),
                C.c_void_p(o_sc[s].data_ptr()), C.c_void_p(o_cnt[s].data_ptr()), C.c_void_p(streams[s].cuda_stream)))
            return o_ids[s], o_sc[s], o_cnt[s]

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # initialisation, not warmup: every stream's workspace (and, sharded, its RCCL communicator) is created on first
    # use; touch each once so that a small --warmup cannot push that one-time cost into the timed region
    for i in range(nstr):
        step(i)
    barrier()
    for i in range(a.warmup):
        step(i)
    barrier()
    t0 = time.perf_counter()
    for i in range(a.steps):
        step(i)
    barrier()
    dt = time.perf_counter() - t0
    if use_dist:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    qps = a.batch * a.steps / dt

    # ---- p50 latency of one batch (per-step synchronisation; not part of `value`) ---------------------
    lat = []
    for i in range(min(a.steps, 20)):
        barrier()
        t1 = time.perf_counter()
        step(i)
        torch.cuda.synchronize(dev)
        lat.append((time.perf_counter() - t1) * 1e3)
    p50 = float(np.median(lat)) if lat else None

    # ---- per-stage durations (HIP events on the call's stream) + work counters ------------------------------
    stages = None
    nprof = max(2, min(a.steps, 6))
    acc = {}
    for i in range(nprof):
        b = i % a.query_batches
        ix.search_batch(qs[b * a.batch:(b + 1) * a.batch], prm)
        for k, v in ix.last_stats.items():
            acc[k] = acc.get(k, 0) + v
    stages = {k: v / nprof for k, v in acc.items()}

    if rank != 0:
        if use_dist:
            dist.barrier()
            dist.destroy_process_group()
        return

    # ---- roofline of the dominant kernel ---------------------------------------------------------------------
    Lq, d, pd = a.query_tokens, 128, 64
    cand_tokens, exact_tokens = stages["n_cand_tokens"], stages["n_exact_tokens"]
    per_stage = {
        # name: (ms, bound, algorithmic units per launch, unit, peak)
        "qc_gemm(S1)": (stages["ms_centroid"], "mfma", 2.0 * a.batch * Lq * d * a.centroids / 1e12, "TFLOP/s", MFMA_F32_PEAK_TF),
        "probe(S2)": (stages["ms_probe"], "hbm", (a.batch * (a.centroids / 32) * Lq * 4) / 1e9, "GB/s", HBM_PEAK_GBS),
        "candidates(S3)": (stages["ms_candidates"], "hbm", (stages["n_ivf_ids"] * 4 + stages["n_candidates"] * 4) / 1e9, "GB/s", HBM_PEAK_GBS),
        "approx(S4)": (stages["ms_approx"], "hbm", (cand_tokens * 4 + stages["n_candidates"] * 8) / 1e9, "GB/s", HBM_PEAK_GBS),
        "select(S5)": (stages["ms_select"], "hbm", (stages["n_candidates"] * 8) / 1e9, "GB/s", HBM_PEAK_GBS),
        "exact(S6)": ((stages["ms_exact"], "mfma", 2.0 * Lq * d * exact_tokens / 1e12, "TFLOP/s",
                       MFMA_F32_PEAK_TF if a.precision == 0 else MFMA_BF16_PEAK_TF)),
        "exact-hbm(S6)": (stages["ms_exact"], "hbm", (exact_tokens * (pd + 8) + exact_tokens * 128) / 1e9, "GB/s", HBM_PEAK_GBS),
    }
    dom = max(per_stage, key=lambda k: per_stage[k][0])
    ms, bound, units, unit, peak = per_stage[dom]
    achieved = units / (ms * 1e-3) if ms > 0 else 0.0
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")     # PMC-derived HBM bytes per launch, if collected
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get(dom)
        except Exception:
            traffic = None
    roofline = dict(kernel=dom, bound=bound, achieved=round(achieved, 3), peak=peak, unit=unit,
                    frac=round(achieved / peak, 5), traffic=traffic, ms_per_launch=round(ms, 4),
                    hbm_bytes_s6=round(exact_tokens * (pd + 4) / 1e9 / (stages["ms_exact"] * 1e-3), 2) if stages["ms_exact"] > 0 else None)

    # ---- CPU baseline: the oracle restatement on this box's host cores, bounded sample ------------------------
    cpu = None
    parity = None
    if a.cpu_queries > 0 and world == 1:
        try:
            if use_dist:   # sharded protocol on one rank: results come from the merged path
                ss_check = ss
            from oracle import oracle as O
            e = ix.export()
            ox = O.OracleIndex(cen, synth.bucket_tables(spec)[1], e["ivf"], e["ivf_lengths"], e["doc_lengths"],
                               e["codes"], e["residuals"], 4)
            po = O.SearchParameters(n_full_scores=a.n_full_scores, top_k=a.top_k, n_ivf_probe=a.nprobe,
                                    centroid_score_threshold=thr)
            nc = min(a.cpu_queries, nq)
            ox.search_batch(qs[:min(8, nc)], po)            # warm page cache / threads
            t1 = time.perf_counter()
            ref = ox.search_batch(qs[:nc], po)
            tc = time.perf_counter() - t1
            cpu = dict(value=round(nc / tc, 3), unit="queries/s", cores=O.num_threads(), kind="port",
                       sample=f"{nc} queries (one batch) of the same 1M-doc index and parameters, oracle C restatement "
                              f"of next-plaid 1.6.1 search.rs, OpenMP over queries/candidates, {tc:.1f} s")
            got = ss.search_batch(qs[:nc], prm) if use_dist else ix.search_batch(qs[:nc], prm)
            agree = sum(int(np.array_equal(g.passage_ids, r.passage_ids)) for g, r in zip(got, ref))
            top1 = sum(int(g.passage_ids[:1].tolist() == r.passage_ids[:1].tolist()) for g, r in zip(got, ref))
            rel = max((float(np.max(np.abs(g.scores - r.scores) / np.maximum(np.abs(r.scores), 1e-6)))
                       for g, r in zip(got, ref) if g.scores.size and g.scores.size == r.scores.size), default=0.0)
            parity = dict(queries=nc, topk_ids_identical=agree, top1_identical=top1, max_rel_score_err=rel,
                          source_doc_rank1=sum(int(g.passage_ids[0] == s) for g, s in zip(got, src[:nc]) if g.passage_ids.size))
            del ox, e
        except Exception as ex:  # e.g. host too small for the 21.6 GB export: report, do not fail the bench
            cpu = dict(value=None, unit="queries/s", cores=None, kind="port", sample=f"skipped: {type(ex).__name__}: {ex}")

    out = {
        "metric": "queries/sec, k=10, PLAID candidate-gen -> residual-decompress -> MaxSim",
        "value": round(qps, 2), "unit": "queries/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(dt / a.steps * 1e3, 4), "p50_batch_latency_ms": None if p50 is None else round(p50, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": {0: "f32", 1: "f32 (bf16 MFMA on the residual term of MaxSim)",
                  2: "f32 (split-bf16 hi/lo MFMA on the residual term of MaxSim, f32-class accuracy)",
                  3: "f32 + bf16 MaxSim"}[a.precision], "data": "synthetic",
        "streams": nstr,
        # weak scaling here = the corpus grows with N (one 1M-doc shard per GPU) while every rank answers the same
        # queries, so queries/s is expected to stay flat; the work rate that grows with N is documents searched per s
        "docs_searched_per_s": round(qps * a.docs_per_gpu * world, 1),
        "config": {"workload": f"{a.docs_per_gpu * world} docs x {a.doc_len} tok x d128 (nbits=4), 2^{int(np.log2(a.centroids))} centroids, "
                               f"nprobe={a.nprobe}, batch={a.batch}x{a.query_tokens} tok, n_full_scores={a.n_full_scores}, "
                               f"t_cs={thr}, top_k={a.top_k}; {a.docs_per_gpu} docs per GPU shard",
                   "docs_total": a.docs_per_gpu * world, "docs_per_gpu": a.docs_per_gpu, "batch": a.batch,
                   "parallelism": f"doc-shard x{world} + RCCL all-gather" if use_dist else "single GPU"},
        "roofline": roofline, "cpu_baseline": cpu, "parity_vs_oracle": parity,
        "stages": {k: (round(v, 4) if isinstance(v, float) else v) for k, v in stages.items()},
        "index_build_s": round(t_build, 2), "hbm_index_bytes": int(ix.info.device_bytes),
    }
    print(json.dumps(out))
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
