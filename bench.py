#!/usr/bin/env python3
"""bench.py -- queries/sec + p50 latency of the MI355X PLAID search path on BASELINE.json's metric configuration.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Metric (BASELINE.json): queries/sec + p50 latency, k=10, 10M-doc x 300-tok x d128 PLAID index, 1/2/4/8 GPU.
Workload (config.workload): ONE fixed corpus of 10 000 000 docs x 300 tokens x d=128, nbits=4, K=2^16 centroids,
nprobe=32, batch 64 x 32-token queries, n_full_scores=4096 (1024 exact re-ranks), t_cs=0.4, top_k=10 -- generated
in HBM by the seeded generator of next_plaid_amd/synth.py.  With N GPUs the SAME corpus is document-sharded N ways
(10M/N docs per GPU: strong scaling); every rank answers the same query batch and the shards exchange rank keys /
top-k over RCCL (dist.py), so the merged result is the unsharded result.

A "step" is ONE pass of the hot path (S1 centroid scoring -> S7 top-k) over one batch of 64 queries whose
embeddings are already resident in HBM; value = queries / second, whole job (`value_pcie_inclusive` = the same batches through
np_hip_search_batch -- host query buffers in, host results out -- from as many host threads as `value` uses streams).

Rank 0 prints ONE JSON line.  Extra objects: "roofline" (dominant kernel: algorithmic bytes or flops per launch /
its HIP-event duration on the call's stream vs the chip peak), "cpu_baseline" (the oracle restatement of the
reference CPU path on this box's host cores, bounded sample, median of 3), "parity_vs_oracle", "stages".
Other BASELINE.json configs are reachable with flags (e.g. config 2: --docs 1000000), they are not the default line.
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
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--docs", type=int, default=10_000_000, help="documents of the WHOLE corpus (sharded over --gpus)")
    ap.add_argument("--doc-len", type=int, default=300)
    ap.add_argument("--doc-len-min", type=int, default=0, help="> 0: ragged documents, uniform in [doc-len-min, doc-len]")
    ap.add_argument("--len-dist", choices=["uniform", "lognormal"], default="uniform",
                    help="lognormal: clipped LogNormal(mean ~73, max 180) document lengths, the MS MARCO passage shape (config 3)")
    ap.add_argument("--topics", type=int, default=8, help="topic centroids per document (generator)")
    ap.add_argument("--rand256", type=int, default=51,
                    help="of 256: share of tokens with a uniformly random code (51 = 20 %%: ~68 distinct codes per 300-token "
                         "document = 0.23 per token; 121 -> 0.5; 200 -> 0.8)")
    ap.add_argument("--hot", type=int, default=-1, help="s4_hot per-mille (first filter level); -1 = library default")
    ap.add_argument("--s1-split", action="store_true",
                    help="opt into the split-bf16 S1 (qc_gemm_b3_kernel) where K > centroid_batch_size and precision >= 1: S1-S5 are "
                         "then no longer bit-equal to the f32 chain (the crate-heuristic K = 2^19 line runs with it)")
    ap.add_argument("--centroids", type=int, default=65536)
    ap.add_argument("--nbits", type=int, default=4)
    ap.add_argument("--nprobe", type=int, default=32)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--query-tokens", type=int, default=32)
    ap.add_argument("--n-full-scores", type=int, default=4096)
    ap.add_argument("--top-k", type=int, default=10)
    ap.add_argument("--threshold", type=float, default=0.4, help="centroid_score_threshold; <0 = None")
    ap.add_argument("--centroid-batch-size", type=int, default=100_000, help="K above this takes the batched-probe path")
    ap.add_argument("--precision", type=int, default=2,
                    help="exact MaxSim arithmetic: 2 QC-reuse split-bf16 (f32-class, the library default), 0 exact-f32 MFMA, "
                         "1 QC-reuse bf16, 3 plain bf16")
    ap.add_argument("--cpu-queries", type=int, default=64,
                    help="queries per repeat of the CPU-oracle leg (0 = skip).  64 = one batch, BASELINE.md section 3: the oracle's outer "
                         "OpenMP level runs min(queries, threads) queries at once like the reference's rayon par_iter (search.rs:650-664), "
                         "so a batch of 64 fills a 128-thread box with 2 inner threads each; 16 queries left the inner level to do it")
    ap.add_argument("--cpu-repeats", type=int, default=3)
    ap.add_argument("--cpu-outer", type=int, default=16,
                    help="queries the CPU oracle takes at once (its outer OpenMP width; the inner level gets threads / this).  The batch of "
                         "--cpu-queries goes through in pieces of this size; the whole batch at once is timed too and the faster split reported")
    ap.add_argument("--parity-queries", type=int, default=64, help="queries compared with the oracle at full size")
    ap.add_argument("--cpu-docs", type=int, default=0,
                    help="CPU leg on the first N docs only (0 = the whole corpus; used automatically if the export fails)")
    ap.add_argument("--query-batches", type=int, default=4)
    ap.add_argument("--force-dist", action="store_true",
                    help="run the sharded RCCL protocol even with one rank (exercises the N > 1 code path on a 1-GPU box)")
    ap.add_argument("--dist-impl", choices=["c", "torch"], default="c",
                    help="sharded protocol through np_hip_search_batch_sharded (RCCL below the C ABI) or the torch.distributed harness")
    ap.add_argument("--shards", type=int, default=0,
                    help="document shards of the corpus (0 = one per rank, the north_star layout).  With S < N ranks the job runs "
                         "N/S REPLICAS of an S-way sharded index: rank r holds shard r %% S in replica group r // S, the groups "
                         "answer DIFFERENT batches concurrently (S = 1: every GPU holds the whole index, no collective at all)")
    ap.add_argument("--hosted", action="store_true",
                    help="N > 1 ranks that SHARE GPU 0, the collectives over the hosted transport (np_hip_comm_create_hosted + gloo): "
                         "executes this script's whole multi-rank control flow (process group, replica groups, one communicator per "
                         "stream, status polling, max-over-ranks timing) on a 1-GPU box.  RCCL refuses two ranks on one device, so this "
                         "is NOT a scaling measurement and the line says so")
    ap.add_argument("--streams", type=int, default=3,
                    help="HIP streams the steps are issued on round-robin (each step = one full batch pass; the "
                         "small launch-bound kernels of one batch overlap the memory-bound ones of the next)")
    ap.add_argument("--workspace-gib", type=float, default=0.0, help="per-context scratch budget (0 = library default)")
    ap.add_argument("--from-disk", default="",
                    help="directory: the generated corpus is exported, WRITTEN there as an index directory in the crate's on-disk "
                         "format (np_hip_index_write_dir), the HBM copy dropped, and the bench runs on np_hip_index_open of that "
                         "directory (= MmapIndex::load).  The line gains `disk_open` (seconds, bytes, GB/s: warm page cache and, "
                         "when the box lets us drop caches, cold)")
    return ap.parse_args()


def kernels_sha():
    """Identity of the kernel sources the loaded library was built from (PMC traffic figures are only valid for them)."""
    import hashlib
    h = hashlib.sha256()
    for f in ("np_kernels.h", "np_search.hip", "np_index.hip", "np_internal.h"):
        with open(os.path.join(ROOT, "next-plaid_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def spawn_ranks(a):
    """`python bench.py --gpus N` without a launcher: start the N ranks here (torch.distributed.run, one rank per GPU, rendezvous on
    127.0.0.1) and pass rank 0's JSON line through.  Fails loudly when the box has fewer than N GPUs (unless --hosted: N ranks
    that share GPU 0) -- a silent one-rank run labelled n_gpus: 1 is what this replaces."""
    import socket
    import subprocess
    if not a.hosted:
        import next_plaid_amd as npa
        have = npa.device_count()
        if have < a.gpus:
            raise SystemExit(f"bench.py: --gpus {a.gpus} but this box has {have} gfx950 device(s) "
                             f"(--hosted runs {a.gpus} ranks on GPU 0 over the hosted transport: a control-flow run, not a scaling one)")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["MASTER_PORT"] = str(port)          # main() only setdefault()s these: torchrun's values win inside the ranks
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print("bench.py: no launcher in the environment (WORLD_SIZE unset): starting the ranks with\n  " + " ".join(cmd), file=sys.stderr)
    raise SystemExit(subprocess.call(cmd, env=env))


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(a)
    # stdout carries exactly ONE line, the JSON: everything else this process (or a C library inside it -- RCCL prints
    # a version banner through C stdio) writes to fd 1 goes to stderr
    json_fd = os.dup(1)
    os.dup2(2, 1)
    import torch
    import next_plaid_amd as npa
    from next_plaid_amd import api, synth

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.gpus != world:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    if npa.device_count() < 1:
        raise SystemExit("bench.py needs a gfx950 GPU (the HIP path has no CPU fallback)")
    if a.hosted:
        local_rank = 0                                   # every rank on GPU 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    cdev = torch.device("cpu") if a.hosted else dev      # where the harness's own small collectives live (gloo: host tensors)
    n_shards = a.shards if a.shards > 0 else world
    if world % n_shards:
        raise SystemExit(f"--shards {n_shards} does not divide the {world} ranks")
    n_repl = world // n_shards
    shard, repl = rank % n_shards, rank // n_shards      # replica group `repl` = ranks [repl * S, (repl + 1) * S)
    use_dist = world > 1 or a.force_dist                 # process group: timing barrier / max-over-ranks, id exchange
    use_shards = n_shards > 1 or a.force_dist            # the sharded search protocol (RCCL all-gathers inside a replica group)
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if a.hosted:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    # ---- corpus: ONE fixed corpus generated in HBM, document-sharded over the ranks -------------------------------
    dim = 128
    pd = dim * a.nbits // 8
    len_min = a.doc_len_min if a.doc_len_min > 0 else a.doc_len
    gen = dict(n_topics=a.topics, rand256=a.rand256)
    if a.len_dist == "lognormal":
        gen["len_table"] = synth.lognormal_len_table()
        len_min, a.doc_len = 1, 180
    spec = synth.SynthSpec(num_docs=a.docs, num_centroids=a.centroids, dim=dim, nbits=a.nbits,
                           doc_len_min=len_min, doc_len_max=a.doc_len, seed=1236, **gen)
    cen = synth.centroids(spec)
    opts = dict(device=local_rank, shard_rank=shard, shard_count=n_shards, max_batch=a.batch, n_contexts=max(1, a.streams))
    if a.workspace_gib > 0:
        opts["workspace_bytes"] = int(a.workspace_gib * (1 << 30))
    t0 = time.time()
    ix = npa.MmapIndex.synth(spec, centroids=cen, **opts)
    t_build = time.time() - t0
    disk_open = None
    if a.from_disk:
        if world > 1:
            raise SystemExit("--from-disk is a one-GPU regime line")
        import shutil
        e = ix.export()
        t_exp = time.time() - t0 - t_build
        ix.close()
        shutil.rmtree(a.from_disk, ignore_errors=True)
        t1 = time.time()
        npa.write_index_dir(a.from_disk, cen, synth.bucket_tables(spec)[1], e["doc_lengths"], e["codes"], e["residuals"], a.nbits,
                            ivf=e["ivf"], ivf_lengths=e["ivf_lengths"])
        t_write = time.time() - t1
        nbytes = sum(os.path.getsize(os.path.join(a.from_disk, f)) for f in os.listdir(a.from_disk))
        del e
        t1 = time.time()
        ix = npa.MmapIndex.load(a.from_disk, **opts)
        t_warm = time.time() - t1
        t_cold = None
        try:   # cold: the index files' pages are dropped from the page cache file by file (posix_fadvise DONTNEED after a sync:
            # no machine-wide setting is touched); where the kernel keeps them anyway the "cold" time simply equals the warm one
            ix.close()
            os.sync()
            for f in os.listdir(a.from_disk):
                fd = os.open(os.path.join(a.from_disk, f), os.O_RDONLY)
                try:
                    os.posix_fadvise(fd, 0, 0, os.POSIX_FADV_DONTNEED)
                finally:
                    os.close(fd)
            t1 = time.time()
            ix = npa.MmapIndex.load(a.from_disk, **opts)
            t_cold = time.time() - t1
        except OSError:
            ix = npa.MmapIndex.load(a.from_disk, **opts)
        disk_open = dict(dir=a.from_disk, bytes=nbytes, files=len(os.listdir(a.from_disk)), export_s=round(t_exp, 2),
                         write_s=round(t_write, 2), open_warm_s=round(t_warm, 2), open_warm_gbs=round(nbytes / 1e9 / t_warm, 2),
                         open_cold_s=None if t_cold is None else round(t_cold, 2),
                         open_cold_gbs=None if t_cold is None else round(nbytes / 1e9 / t_cold, 2),
                         note="np_hip_index_open = MmapIndex::load of the crate's own file set (chunked i64 codes, packed residuals, "
                              "JSON doclens, ivf.npy): seconds until the handle is searchable, derived structures included")
    if a.hot >= 0:
        ix.tune("s4_hot", a.hot)
        ix.tune("s4_hot_auto", 0)      # exactly this share, whatever the candidate count
    if a.s1_split:
        ix.tune("s1_split", 1)
    s1_split = (a.s1_split or os.environ.get("NP_S1_SPLIT", "0") not in ("", "0")) and a.precision >= 1 and \
        0 < a.centroid_batch_size < a.centroids
    docs_local = int(ix.info.shard_doc_end - ix.info.shard_doc_begin)
    thr = None if a.threshold < 0 else a.threshold
    prm = npa.SearchParameters(n_full_scores=a.n_full_scores, top_k=a.top_k, n_ivf_probe=a.nprobe,
                               centroid_batch_size=a.centroid_batch_size, centroid_score_threshold=thr,
                               precision=a.precision)

    # ---- queries: resident in HBM before the timed region ------------------------------------------------------------
    nq = a.batch * a.query_batches
    qs, src = synth.make_queries(spec, nq, n_tokens=a.query_tokens, cen=cen)
    off = np.arange(a.batch + 1, dtype=np.int32) * a.query_tokens
    nstr = max(1, a.streams)
    streams = [torch.cuda.Stream(dev) for _ in range(nstr)]
    with torch.cuda.stream(streams[0]):
        dq = [torch.from_numpy(np.concatenate(qs[i * a.batch:(i + 1) * a.batch], 0)).to(dev) for i in range(a.query_batches)]
        doff = torch.from_numpy(off).to(dev)
        o_ids = [torch.zeros((a.batch, max(a.top_k, 1)), dtype=torch.int64, device=dev) for _ in range(nstr)]
        o_sc = [torch.zeros((a.batch, max(a.top_k, 1)), dtype=torch.float32, device=dev) for _ in range(nstr)]
        o_cnt = [torch.zeros(a.batch, dtype=torch.int32, device=dev) for _ in range(nstr)]
    torch.cuda.synchronize(dev)
    L = api.lib()
    cp = prm._c()

    def qbatch(i):   # replica groups walk the query batches at different offsets: different batches at the same time
        return (i + repl) % a.query_batches

    dist_impl, dist_note = a.dist_impl, ""
    if use_shards and dist_impl == "c":
        # the whole protocol below the C ABI (np_hip_search_batch_sharded, RCCL all-gathers issued by the library on
        # the call's stream): one communicator per stream so the collectives of batch i overlap the kernels of batch
        # i+1; rank 0's ncclUniqueId reaches the other ranks through a torch.distributed broadcast
        from next_plaid_amd.dist import CShardedSearcher, ShardComm, gloo_all_gather

        grp = None
        if use_dist and n_repl > 1:   # new_group is collective over ALL ranks: every rank creates every replica group
            grps = [dist.new_group(ranks=list(range(g * n_shards, (g + 1) * n_shards))) for g in range(n_repl)]
            grp = grps[repl]

        def exchange(b):              # rank 0 of the replica group draws the id, the group's other ranks receive it
            obj = [b]
            dist.broadcast_object_list(obj, src=repl * n_shards, group=grp)
            return obj[0]

        # np_hip_comm_create is collective (ncclCommInitRank): if it fails anywhere (librccl missing, an id that could not
        # be drawn) EVERY rank falls back to the torch.distributed harness of the same protocol, and the line says so.
        comms, err = [], ""
        try:
            try:
                if shard == 0 and n_shards > 1 and not a.hosted:
                    buf = C.create_string_buffer(128)
                    api._check(L.np_hip_comm_unique_id(buf))   # librccl loads here, before any collective is entered
            except Exception as ex:   # noqa: BLE001
                err = f"{type(ex).__name__}: {ex}"
                print(f"bench.py: rank {rank}: np_hip_comm_unique_id failed: {err}", file=sys.stderr)
            if use_dist:
                flag = torch.tensor([1 if err else 0], dtype=torch.int32, device=cdev)
                dist.all_reduce(flag, op=dist.ReduceOp.MAX)
                if int(flag.item()) and not err:
                    err = "np_hip_comm_unique_id failed on another rank"
            if not err and a.hosted:
                comms = [ShardComm(ix, shard, n_shards, all_gather=gloo_all_gather(grp)) for _ in range(nstr)]
            elif not err:
                comms = [ShardComm(ix, shard, n_shards, exchange=exchange if n_shards > 1 else None) for _ in range(nstr)]
        except Exception as ex:       # noqa: BLE001 -- a failure inside ncclCommInitRank itself
            err = err or f"{type(ex).__name__}: {ex}"
            print(f"bench.py: rank {rank} (shard {shard}, replica group {repl}): communicator creation failed: {err}", file=sys.stderr)
        if use_dist:
            flag = torch.tensor([1 if err else 0], dtype=torch.int32, device=cdev)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            if int(flag.item()) and not err:
                err = "np_hip_comm_create failed on another rank"
        if err and n_repl == 1 and use_dist and not a.hosted:
            for cm in comms:
                cm.close()
            dist_impl = "torch"
            dist_note = f" [np_hip_comm_create failed ({err[:160]}): fell back to the torch.distributed harness]"
            print("bench.py:" + dist_note, file=sys.stderr)
        elif err:
            raise SystemExit(f"bench.py: the RCCL communicator could not be created: {err}")
    comm_info = None
    if use_shards and dist_impl == "c":
        comm_info = comms[0].info()
        sss = [CShardedSearcher(ix, comms[s], stream=streams[s]) for s in range(nstr)]
        ss = sss[0]

        def step(i):
            s = i % nstr
            return sss[s].search_batch_device(dq[qbatch(i)], doff, off, prm, out=(o_ids[s], o_sc[s], o_cnt[s]))
    elif use_shards:
        # torch.distributed harness of the same protocol (dist.py): one searcher per stream, each with its own process
        # group (= its own RCCL communicator); every rank issues the same round-robin order
        from next_plaid_amd.dist import HipShardBackend, ShardedSearcher
        if n_repl > 1:
            raise SystemExit("--dist-impl torch runs one replica group only (use the default C implementation with --shards)")
        groups = [dist.new_group(ranks=list(range(world))) for _ in range(nstr)]
        sss = [ShardedSearcher([HipShardBackend(ix, stream=streams[s])], use_dist=True, group=groups[s])
               for s in range(nstr)]
        ss = sss[0]

        def step(i):
            return sss[i % nstr].search_batch_device(dq[qbatch(i)], doff, off, prm)
    else:
        def step(i):
            s = i % nstr
            api._check(L.np_hip_search_batch_device(
                ix._h, C.c_void_p(dq[qbatch(i)].data_ptr()), C.c_void_p(doff.data_ptr()),
                off.ctypes.data_as(C.c_void_p), a.batch, dim, C.byref(cp), None, -1, C.c_void_p(o_ids[s].data_ptr()),
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
    if use_shards and dist_impl == "c":
        for cm in comms:   # a rank that failed locally empties the batch everywhere and leaves its status here (np_dist.hip)
            fr, code = cm.status()
            if code:
                raise SystemExit(f"bench.py: shard {fr} failed with np_status {code} inside the timed region")
    if use_dist:
        tmax = torch.tensor([dt], dtype=torch.float64, device=cdev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    qps = a.batch * a.steps * n_repl / dt      # every replica group answered `steps` batches of its own in that time

    # ---- p50 latency of one batch (per-step synchronisation; not part of `value`) ---------------------------------------
    lat = []
    for i in range(min(a.steps, 20)):
        barrier()
        t1 = time.perf_counter()
        step(i)
        torch.cuda.synchronize(dev)
        lat.append((time.perf_counter() - t1) * 1e3)
        if use_shards and dist_impl == "c":   # per batch here (the batch is synchronised anyway): a peer's failure shows at once
            fr, code = comms[i % nstr].status()
            if code:
                raise SystemExit(f"bench.py: shard {fr} failed with np_status {code} in a latency batch")
    p50 = float(np.median(lat)) if lat else None

    # ---- per-stage durations (HIP events on the call's stream) + work counters; the same calls, timed on the host
    # clock, are the PCIe-inclusive rate: np_hip_search_batch takes host query buffers and returns host results ---------
    nprof = max(2, min(a.steps, 6))
    acc = {}
    host_t = []
    for i in range(nprof):
        b = i % a.query_batches
        t1 = time.perf_counter()
        ix.search_batch(qs[b * a.batch:(b + 1) * a.batch], prm)
        host_t.append(time.perf_counter() - t1)
        for k, v in ix.last_stats.items():
            acc[k] = acc.get(k, 0) + v
    stages = {k: v / nprof for k, v in acc.items()}
    qps_pcie_1 = a.batch / float(np.median(host_t)) if world == 1 else None   # ONE call in flight, host buffers

    # ---- SURVEY 8(d)'s protocol at the SAME concurrency as `value`: np_hip_search_batch (host query buffers in, host results
    # out: H2D + S1..S7 + D2H inside the call) from as many host threads as `value` uses streams -- the way the API's tokio
    # workers call MmapIndex::search_batch (next-plaid-api/src/handlers/search.rs:219-229).  Each thread has its own host
    # buffers; the library hands each concurrent call its own context (stream + workspace + pinned result staging).
    qps_pcie, pcie_calls = None, 0
    if world == 1:
        import threading
        hq = [np.ascontiguousarray(np.concatenate(qs[i * a.batch:(i + 1) * a.batch], 0), np.float32) for i in range(a.query_batches)]
        hoff = np.ascontiguousarray(off, np.int32)
        k1 = max(a.top_k, 1)
        per_thread = max(4, min(a.steps, 240) // nstr)
        outs = [(np.zeros(a.batch * k1, np.int64), np.zeros(a.batch * k1, np.float32), np.zeros(a.batch, np.int32)) for _ in range(nstr)]
        errs = []
        gate = threading.Barrier(nstr + 1)

        def host_worker(t):
            oi, osc, oc = outs[t]
            try:
                for j in range(2):          # untimed: the thread's first calls
                    api._check(L.np_hip_search_batch(ix._h, hq[(t + j) % a.query_batches].ctypes.data, hoff.ctypes.data, a.batch, dim,
                                                     C.byref(cp), None, -1, oi.ctypes.data, osc.ctypes.data, oc.ctypes.data, None))
                gate.wait()
                for j in range(per_thread):
                    api._check(L.np_hip_search_batch(ix._h, hq[(t + j) % a.query_batches].ctypes.data, hoff.ctypes.data, a.batch, dim,
                                                     C.byref(cp), None, -1, oi.ctypes.data, osc.ctypes.data, oc.ctypes.data, None))
            except Exception as ex:   # noqa: BLE001
                errs.append(f"{type(ex).__name__}: {ex}")
                try:
                    gate.abort()
                except Exception:   # noqa: BLE001
                    pass

        ths = [threading.Thread(target=host_worker, args=(t,)) for t in range(nstr)]
        for th in ths:
            th.start()
        try:
            gate.wait()
            t1 = time.perf_counter()
            for th in ths:
                th.join()
            dt_h = time.perf_counter() - t1
            if not errs:
                pcie_calls = per_thread * nstr
                qps_pcie = a.batch * pcie_calls / dt_h
        except threading.BrokenBarrierError:
            for th in ths:
                th.join()
        if errs:
            print("bench.py: PCIe-inclusive leg failed: " + "; ".join(errs[:2]), file=sys.stderr)

    # ---- N > 1: the merged result of the sharded protocol against the oracle on the WHOLE corpus (small corpora only: rank 0
    # builds the unsharded corpus next to its shard to export it for the CPU oracle).  Collective: every rank takes part.
    got_sharded = None
    if use_shards and world > 1 and a.parity_queries > 0 and a.docs <= 1_000_000:
        got_sharded = ss.search_batch(qs[:min(a.parity_queries, nq)], prm)

    if rank != 0:
        if use_dist:
            dist.barrier()
            dist.destroy_process_group()
        return

    # ---- roofline of the dominant kernel (rank 0's shard) ------------------------------------------------------------------
    Lq, d = a.query_tokens, dim
    cand_tokens, exact_tokens = stages["n_cand_tokens"], stages["n_exact_tokens"]
    lvl0 = stages.get("n_level0", 0)
    if lvl0 and lvl0 < stages["n_candidates"]:
        # the zeroth filter level ran: the hot level -- which counts the candidates' tokens -- saw only what it handed over; the
        # contract's Tc is the token count of EVERY candidate (exact for fixed-length corpora, scaled by the counts otherwise)
        cand_tokens = stages["n_candidates"] * a.doc_len if len_min == a.doc_len else cand_tokens * stages["n_candidates"] / lvl0
    per_stage = {
        # name: (ms, bound, algorithmic units per launch, unit, peak)
        # split-bf16 S1: three bf16 MFMAs per product, priced against the dense bf16 peak
        "qc_gemm(S1)": ((stages["ms_centroid"], "mfma", 3 * 2.0 * a.batch * Lq * d * a.centroids / 1e12, "TFLOP/s", MFMA_BF16_PEAK_TF)
                        if s1_split else
                        (stages["ms_centroid"], "mfma", 2.0 * a.batch * Lq * d * a.centroids / 1e12, "TFLOP/s", MFMA_F32_PEAK_TF)),
        "probe(S2)": (stages["ms_probe"], "hbm", (a.batch * (a.centroids / 32) * Lq * 4) / 1e9, "GB/s", HBM_PEAK_GBS),
        "candidates(S3)": (stages["ms_candidates"], "hbm", (stages["n_ivf_ids"] * 4 + stages["n_candidates"] * 4) / 1e9, "GB/s", HBM_PEAK_GBS),
        "approx(S4)": (stages["ms_approx"], "hbm", (cand_tokens * 4 + stages["n_candidates"] * 8) / 1e9, "GB/s", HBM_PEAK_GBS),
        "select(S5)": (stages["ms_select"], "hbm", (stages["n_candidates"] * 8) / 1e9, "GB/s", HBM_PEAK_GBS),
        "exact(S6)": (stages["ms_exact"], "hbm", exact_tokens * (pd + 4) / 1e9, "GB/s", HBM_PEAK_GBS),
    }
    dom = max(per_stage, key=lambda k: per_stage[k][0])
    ms, bound, units, unit, peak = per_stage[dom]
    achieved = units / (ms * 1e-3) if ms > 0 else 0.0
    traffic, traffic_src, tj_ok = None, None, None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")     # PMC-derived HBM bytes per launch, if collected
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            # counters belong to one workload AND one build of the kernels: never carried over to another
            default_workload = (a.rand256 == 51 and a.centroids == 65536 and a.nbits == 4 and a.nprobe == 32 and a.batch == 64 and
                                a.query_tokens == 32 and a.n_full_scores == 4096 and a.threshold == 0.4 and a.precision == 2 and
                                a.hot < 0 and a.doc_len == 300 and a.doc_len_min == 0 and a.len_dist == "uniform" and not s1_split and
                                not any(k.startswith("NP_S") for k in os.environ))
            if default_workload and tj.get("docs_per_gpu") == docs_local and tj.get("kernels_sha") == kernels_sha():
                traffic = tj.get(dom)
                traffic_src = {"commit": tj.get("commit"), "kernels_sha": tj.get("kernels_sha")}
                tj_ok = tj
        except Exception:
            traffic = None
    # `frac` prices the contract's ALGORITHMIC bytes (Tc*4 + C*8 for S4) against the stage's duration; the filter
    # never reads most of them, so the bytes the stage really moved (PMC `traffic`) give the PHYSICAL fraction beside it
    frac_phys = None
    if traffic and ms > 0 and bound == "hbm":
        frac_phys = round(traffic / 1e9 / (ms * 1e-3) / peak, 5)
    # The dominant KERNEL of that stage by itself: the first filter level (one launch per batch, timed by its own pair of HIP events
    # on the call's stream, np_stats.ms_hot_level).  Its algorithmic bytes are what ITS algorithm has to move per launch -- per
    # candidate the id (4 B in), the list block's header (16 B) and distinct codes, the 16-B record and the u16 bound it writes,
    # plus one plane row per hot (document, code) pair it gathers -- not the contract's per-token figure, which this kernel exists
    # to avoid; `traffic` = what the PMC counters saw for this kernel alone.
    dom_kernel = None
    ms_hot = stages.get("ms_hot_level", 0.0)
    if dom == "approx(S4)" and ms_hot > 0:
        code_b = 2 if a.centroids <= 65536 else 4
        row_b = 32 if Lq <= 32 else 64
        # rows gathered by the hot level alone are not counted separately (n_cand_codes = both levels): the upper bound of its
        # share is all of them
        alg = stages["n_candidates"] * (4 + 16 + 16 + 2) + stages.get("n_cand_dcodes", 0) * code_b
        kname = "approx_hotp_kernel" if os.environ.get("NP_S4_PLANES", "1") != "0" else "approx_hot_kernel"
        ktr = None
        if tj_ok is not None:
            ktr = (tj_ok.get("per_kernel", {}).get(kname) or {}).get("bytes_per_batch")
        dom_kernel = dict(kernel=kname, ms_per_launch=round(ms_hot, 4), bound="hbm", algorithmic_bytes=int(alg),
                          achieved=round(alg / 1e9 / (ms_hot * 1e-3), 2), unit="GB/s", peak=HBM_PEAK_GBS,
                          frac=round(alg / 1e9 / (ms_hot * 1e-3) / HBM_PEAK_GBS, 5), traffic=ktr,
                          frac_physical=None if not ktr else round(ktr / 1e9 / (ms_hot * 1e-3) / HBM_PEAK_GBS, 5),
                          note="algorithmic bytes = candidates x (4 B id + 16 B block header + 16 B record + 2 B bound) + distinct "
                               "(document, code) pairs x code bytes; plane rows (L2-resident) not counted")
    roofline = dict(kernel=dom, bound=bound, achieved=round(achieved, 3), peak=peak, unit=unit,
                    frac=round(achieved / peak, 5), frac_physical=frac_phys, traffic=traffic, traffic_source=traffic_src,
                    ms_per_launch=round(ms, 4), dominant_kernel=dom_kernel,
                    all={k: dict(ms=round(v[0], 4), bound=v[1], achieved=round(v[2] / (v[0] * 1e-3), 2) if v[0] > 0 else None,
                                 unit=v[3], frac=round(v[2] / (v[0] * 1e-3) / v[4], 4) if v[0] > 0 else None)
                         for k, v in per_stage.items()},
                    s6_mfma_tflops=round(2.0 * Lq * d * exact_tokens / 1e12 / (stages["ms_exact"] * 1e-3), 2) if stages["ms_exact"] > 0 else None)

    # ---- CPU baseline + parity at full size: the oracle restatement on this box's host cores ----------------------------
    cpu = None
    parity = None
    if ((a.cpu_queries > 0 or a.parity_queries > 0) and world == 1) or got_sharded is not None:
        try:
            from oracle import oracle as O
            cix, cdocs, e = ix, a.docs, None
            if not (0 < a.cpu_docs < a.docs) and world == 1:
                try:
                    e = ix.export()                     # whole corpus back to host arrays in the on-disk dtypes
                except (MemoryError, npa.NextPlaidError):
                    e = None
            if e is None:
                # bounded sample: the first cdocs documents of the same corpus (same generator, same seed)
                cdocs = a.cpu_docs if 0 < a.cpu_docs < a.docs and world == 1 else min(a.docs, 1_000_000)
                sspec = synth.SynthSpec(num_docs=cdocs, num_centroids=a.centroids, dim=dim, nbits=a.nbits,
                                        doc_len_min=len_min, doc_len_max=a.doc_len, seed=1236, **gen)
                cix = npa.MmapIndex.synth(sspec, centroids=cen, device=local_rank, max_batch=a.batch, n_contexts=1)
                e = cix.export()
            ox = O.OracleIndex(cen, synth.bucket_tables(spec)[1], e["ivf"], e["ivf_lengths"], e["doc_lengths"],
                               e["codes"], e["residuals"], a.nbits)
            po = O.SearchParameters(n_full_scores=a.n_full_scores, top_k=a.top_k, n_ivf_probe=a.nprobe,
                                    centroid_batch_size=a.centroid_batch_size, centroid_score_threshold=thr)
            if a.cpu_queries > 0 and world == 1:
                nc = min(a.cpu_queries, nq)
                ox.search_batch(qs[:min(4, nc)], po)            # warm page cache / threads
                # The batch of `nc` queries goes through the oracle `outer` queries at a time: its outer OpenMP level takes min(queries,
                # threads) queries at once and leaves threads / that many to the inner (per-candidate) level, like the reference's nested
                # rayon pools.  On the 128-thread box 16 x 8 measured TWICE as fast as 64 x 2 (12.1 vs 6.0 queries/s at 10 M documents:
                # 64 concurrent 8 MB score tables thrash the caches), so the reported value is the faster split and the other one is
                # in `alt` -- the baseline is the CPU path at its best, not at its most convenient.
                outer = max(1, min(a.cpu_outer, nc))

                def cpu_pass(width):
                    t1 = time.perf_counter()
                    for i in range(0, nc, width):
                        ox.search_batch(qs[i:min(i + width, nc)], po)
                    return time.perf_counter() - t1

                reps = [cpu_pass(outer) for _ in range(max(1, a.cpu_repeats))]
                tc = float(np.median(reps))
                alt = None
                if nc > outer and a.cpu_repeats > 1:
                    t_alt = cpu_pass(nc)
                    alt = dict(queries_at_once=nc, value=round(nc / t_alt, 3), seconds=round(t_alt, 2))
                    if t_alt < tc:   # whichever split is faster is the baseline
                        alt, tc, outer, reps = dict(queries_at_once=outer, value=round(nc / tc, 3), seconds=round(tc, 2)), t_alt, nc, [t_alt]
                cpu = dict(value=round(nc / tc, 3), unit="queries/s", cores=O.num_threads(), kind="port",
                           cpu_model=cpu_model(), repeats=len(reps), seconds=[round(x, 2) for x in reps],
                           threads_outer=min(outer, O.num_threads()), threads_inner=max(1, O.num_threads() // max(1, min(outer, O.num_threads()))),
                           alt=alt,
                           sample=f"{nc} queries (one batch, {outer} at a time) x {len(reps)} repeats (median) on {cdocs} of the {a.docs} docs, same parameters; "
                                  f"oracle C restatement of next-plaid 1.6.1 search.rs, OpenMP over queries (outer) / candidates (inner) "
                                  f"like the reference's nested rayon structure")
            if a.parity_queries > 0:
                npq = min(a.parity_queries, nq)
                ref = ox.search_batch(qs[:npq], po)
                if got_sharded is not None:
                    got = got_sharded
                else:
                    got = ss.search_batch(qs[:npq], prm) if (use_shards and cix is ix) else cix.search_batch(qs[:npq], prm)
                agree = sum(int(np.array_equal(g.passage_ids, r.passage_ids)) for g, r in zip(got, ref))
                top1 = sum(int(g.passage_ids[:1].tolist() == r.passage_ids[:1].tolist()) for g, r in zip(got, ref))
                rel = max((float(np.max(np.abs(g.scores - r.scores) / np.maximum(np.abs(r.scores), 1e-6)))
                           for g, r in zip(got, ref) if g.scores.size and g.scores.size == r.scores.size), default=0.0)
                parity = dict(queries=npq, docs=cdocs, through="np_hip_search_batch_sharded over %d ranks" % world if got_sharded is not None else "np_hip_search_batch",
                              topk_ids_identical=agree, top1_identical=top1, max_rel_score_err=rel,
                              source_doc_rank1=sum(int(g.passage_ids[0] == s) for g, s in zip(got, src[:npq])
                                                   if g.passage_ids.size and s < cdocs))
            del ox, e
        except Exception as ex:  # report, do not fail the bench
            print(f"bench.py: CPU baseline / parity leg failed: {type(ex).__name__}: {ex}", file=sys.stderr)
            cpu = cpu or dict(value=None, unit="queries/s", cores=None, kind="port", sample=f"skipped: {type(ex).__name__}: {ex}")

    k2 = int(round(np.log2(a.centroids)))
    out = {
        "metric": "queries/sec + p50 latency, k=10, 10M-doc x 300-tok x d128 PLAID index, 1/2/4/8 GPU",
        "value": round(qps, 2), "unit": "queries/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(dt / a.steps * 1e3, 4), "p50_batch_latency_ms": None if p50 is None else round(p50, 4),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "hosted": bool(a.hosted),
        # what the communicator of rank 0's replica group really is (np_hip_comm_info): transport "rccl" with rccl_ranks_seen =
        # ncclCommCount == shards on a healthy multi-GPU run; "hosted" = gloo on one GPU; None = no sharded protocol in this run
        "transport": (comm_info or {}).get("transport") if use_shards and dist_impl == "c" else ("torch.distributed" if use_shards else None),
        "rccl_ranks_seen": (comm_info or {}).get("rccl_ranks"),
        "hosted_note": ("all %d ranks share GPU 0 and the collectives run over the hosted (gloo) transport: a run of the multi-rank "
                        "control flow, NOT a scaling measurement (RCCL refuses two ranks on one device)" % world) if a.hosted else None,
        "dtype": {0: "f32", 1: "f32 (bf16 MFMA on the residual term of MaxSim)",
                  2: "f32 (split-bf16 hi/lo MFMA on the residual term of MaxSim, f32-class accuracy)",
                  3: "f32 + bf16 MaxSim"}[a.precision], "data": "synthetic",
        "streams": nstr,
        "value_note": "value = device-resident I/O (queries and results stay in HBM, np_hip_search_batch_device on `streams` HIP streams); "
                      "value_pcie_inclusive = SURVEY 8(d)'s protocol, host buffers in / host results out through np_hip_search_batch from "
                      "the same number of host threads",
        "value_pcie_inclusive": None if qps_pcie is None else round(qps_pcie, 2),
        "pcie_inclusive": None if qps_pcie is None else dict(host_threads=nstr, calls=pcie_calls, one_call_in_flight=round(qps_pcie_1, 2),
                                                              ratio_to_value=round(qps_pcie / qps, 4)),
        "config": {"workload": f"{a.docs} docs x {a.doc_len if len_min == a.doc_len else f'{len_min}-{a.doc_len}'} tok x d128 "
                               f"(nbits={a.nbits}), 2^{k2} centroids, nprobe={a.nprobe}, batch={a.batch}x{a.query_tokens} tok, "
                               f"n_full_scores={a.n_full_scores}, t_cs={thr}, top_k={a.top_k}; one fixed corpus sharded "
                               f"{n_shards} way(s) x {n_repl} replica group(s): {docs_local} docs on rank 0's GPU",
                   "centroids_note": (f"K = 2^{k2} is an explicit choice (BASELINE config 2's K carried to this corpus; the metric "
                                      f"names no K).  The crate's k-means heuristic (kmeans.rs:303-309) would pick 2^19 at 10 M x 300 "
                                      f"tokens, the batched-probe regime: run --centroids 524288 for that line") if a.centroids == 65536 and a.docs >= 5_000_000 else None,
                   "s1_split": bool(s1_split),
                   "docs_total": a.docs, "docs_per_gpu": docs_local, "batch": a.batch, "shards": n_shards, "replicas": n_repl,
                   "parallelism": ((f"doc-shard x{n_shards} + {'hosted gloo' if a.hosted else 'RCCL'} all-gather ({'np_hip_search_batch_sharded' if dist_impl == 'c' else 'torch.distributed harness'}){dist_note}"
                                    if use_shards else "whole index per GPU")
                                   + (f", x{n_repl} replica groups on different batches" if n_repl > 1 else ""))
                                  if use_dist else "single GPU"},
        "roofline": roofline, "cpu_baseline": cpu, "parity_vs_oracle": parity,
        "stages": {k: (round(v, 4) if isinstance(v, float) else v) for k, v in stages.items()},
        "disk_open": disk_open,
        "index_build_s": round(t_build, 2), "hbm_index_bytes": int(ix.info.device_bytes),
        "hbm_bytes_per_token": round(ix.info.device_bytes / max(int(ix.info.shard_embeddings), 1), 2),
    }
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    sys.stdout.flush()
    os.write(json_fd, (json.dumps(out) + "\n").encode())


if __name__ == "__main__":
    main()
