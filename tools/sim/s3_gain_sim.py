#!/usr/bin/env python3
"""CPU simulation of the ZEROTH filter level (design aid, not product code): an upper bound of a candidate's approximate score
(search.rs:305-324) from the probed cells alone, i.e. from what S3 sees while it unions the posting lists -- no code list read.

  top(q)   = token q's top-nprobe centroids (search.rs:388-414), theta_q = the smallest score among them
  P        = union of top(q);  P' = cells the threshold keeps (search.rs:417-425)
  theta'_q = max(theta_q, max_{c in P \\ P'} QC[q,c])       -- a centroid outside P' scores <= theta'_q for token q
  gain(c)  = sum_q max(0, QC[q,c] - theta'_q)                for c in P'
  UB0(d)   = sum_q theta'_q + sum_{c in P', d in list(c)} gain(c)   >=   sum_q max_{c in codes(d)} QC[q,c] = approx(d)

Reports, per query: candidates, the true n_sel-th best approximate score tau, how many candidates have UB0 >= tau (the ideal
zeroth-level survivors), and the survivors of the realisable scheme -- S0 = the ~n_sel candidates with the largest UB0, tau0 =
the n_sel-th best exact score inside S0, survivors = {UB0 >= tau0} -- next to the per-token-max variant of the bound
(UBmax, which a single accumulator per document cannot hold) to show what the sum over cells costs.

  SIM_DOCS=10000000 SIM_Q=4 SIM_NPROBE=32 SIM_TCS=0.4 python tools/sim/s3_gain_sim.py
  SIM_NPROBE=8 SIM_TCS=none ...   (the REST API's default)
"""
import os, sys, time
import numpy as np
from multiprocessing import Pool
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "next-plaid_amd"))
from next_plaid_amd import synth

NDOCS = int(os.environ.get("SIM_DOCS", "10000000"))
NQ = int(os.environ.get("SIM_Q", "4"))
LQ = int(os.environ.get("SIM_LQ", "32"))
NPROBE = int(os.environ.get("SIM_NPROBE", "32"))
TCS = os.environ.get("SIM_TCS", "0.4")
TCS = None if TCS.lower() == "none" else float(TCS)
RAND = int(os.environ.get("SIM_RAND256", "51"))
NSEL = int(os.environ.get("SIM_NSEL", "1024"))
SWEEP_REMOVED = int(os.environ.get("SIM_SWEEP_REMOVED", "0"))   # 1: the cells a threshold removes are swept as BOUND-ONLY cells (their
                                                                # gains count, their documents are no candidates): theta stays theta_q
EXACT_SHARE = float(os.environ.get("SIM_EXACT_SHARE", "1.0"))   # share of a chunk's candidates (largest UB0) that get exact scores
K = 65536
spec = synth.SynthSpec(num_docs=NDOCS, num_centroids=K, dim=128, nbits=4, doc_len_min=300, doc_len_max=300,
                       n_topics=8, rand256=RAND, seed=1236)
CH = 25000
G = {}


def codes_only(d0, d1):
    L = 300
    docs = np.repeat(np.arange(d0, d1, dtype=np.uint64), L)
    t = np.tile(np.arange(L, dtype=np.uint64), d1 - d0)
    r = synth.rnd(spec.seed, synth.S_TOK, docs * np.uint64(65536) + t)
    is_rand = (r & np.uint64(0xFF)) < np.uint64(spec.rand256)
    rand_code = ((r >> np.uint64(8)) & np.uint64(0xFFFFFFFF)) % np.uint64(spec.num_centroids)
    top_code = synth._topic(spec, docs, (r >> np.uint64(40)) % np.uint64(spec.n_topics))
    return np.where(is_rand, rand_code, top_code).astype(np.uint16).reshape(d1 - d0, L)


def scan(args):
    d0, d1 = args
    c = codes_only(d0, d1)
    cs = np.sort(c, axis=1)
    first = np.ones(cs.shape, bool)
    first[:, 1:] = cs[:, 1:] != cs[:, :-1]          # one entry per DISTINCT code of a document
    out = []
    for qi in range(NQ):
        kept, gain, QC, thp = G["kept"][qi], G["gain"][qi], G["QC"][qi], G["thp"][qi]
        kh = kept[cs] & first
        hit = kh.any(axis=1)
        idx = np.nonzero(hit)[0]
        cc = cs[idx].astype(np.int64)
        kk = kh[idx]
        if SWEEP_REMOVED:      # gains of every probed cell the document holds (kept or removed)
            kk = (G["probed"][qi][cs] & first)[idx]
        ub0 = thp.sum() + np.where(kk, gain[cc], 0.0).sum(axis=1)
        ncell = kk.sum(axis=1)
        n_ex = idx.size if EXACT_SHARE >= 1.0 else max(int(idx.size * EXACT_SHARE), min(idx.size, 64))
        sel = np.argsort(-ub0)[:n_ex]
        approx = np.full(idx.size, -np.inf, np.float32)
        ubmax = np.full(idx.size, np.inf, np.float32)
        for i0 in range(0, sel.size, 1024):
            s = sel[i0:i0 + 1024]
            g = QC[:, cc[s]]                           # [Lq, m, 300]
            approx[s] = g.max(axis=2).sum(axis=0)
            gk = np.where(kk[s][None], g, -np.inf).max(axis=2)
            ubmax[s] = np.maximum(gk, thp[:, None]).sum(axis=0)
        cut_ub0 = ub0[sel[-1]] if n_ex < idx.size and sel.size else -np.inf   # largest UB0 left without an exact score <= this
        out.append((idx + d0, ub0.astype(np.float32), approx, ubmax, ncell.astype(np.int16), float(cut_ub0)))
    return out


def main():
    cen = synth.centroids(spec)
    qs, src = synth.make_queries(spec, NQ, n_tokens=LQ, cen=cen)
    G["kept"], G["gain"], G["QC"], G["thp"], G["probed"] = [], [], [], [], []
    for q in qs:
        QC = (q @ cen.T).astype(np.float32)              # [Lq, K]
        P = np.zeros(K, bool)
        theta = np.zeros(LQ, np.float32)
        for t, row in enumerate(QC):
            top = np.argpartition(-row, NPROBE - 1)[:NPROBE]
            P[top] = True
            theta[t] = row[top].min()
        kept = P.copy()
        if TCS is not None:
            kept &= QC.max(axis=0) >= TCS
        removed = P & ~kept
        thp = theta.copy()
        if removed.any() and not SWEEP_REMOVED:
            thp = np.maximum(thp, QC[:, removed].max(axis=1))
        gset = P if SWEEP_REMOVED else kept
        gain = np.where(gset, np.maximum(QC - thp[:, None], 0.0).sum(axis=0), 0.0).astype(np.float32)
        G["kept"].append(kept); G["gain"].append(gain); G["QC"].append(QC); G["thp"].append(thp); G["probed"].append(P)
        print(f"query: probed {int(P.sum())} kept {int(kept.sum())} sum theta {theta.sum():.2f} sum theta' {thp.sum():.2f} "
              f"gain of kept cells: max {gain.max():.2f} mean {gain[kept].mean():.2f} total {gain.sum():.1f}", flush=True)
    t0 = time.time()
    jobs = [(d, min(d + CH, NDOCS)) for d in range(0, NDOCS, CH)]
    with Pool(8) as p:
        res = p.map(scan, jobs, chunksize=1)
    print("scan", round(time.time() - t0, 1), "s", flush=True)
    for qi in range(NQ):
        ids = np.concatenate([r[qi][0] for r in res])
        ub0 = np.concatenate([r[qi][1] for r in res])
        approx = np.concatenate([r[qi][2] for r in res])
        ubmax = np.concatenate([r[qi][3] for r in res])
        ncell = np.concatenate([r[qi][4] for r in res])
        cutmax = max(r[qi][5] for r in res)
        n = ids.size
        have = np.isfinite(approx)
        tau = np.sort(approx[have])[-NSEL]
        ok = cutmax < tau            # every candidate without an exact score has UB0 < tau: it is not in the top n_sel
        ideal = int((ub0 >= tau).sum())
        ideal_max = int(((ubmax >= tau) & have).sum()) if EXACT_SHARE >= 1.0 else -1
        line = (f"q{qi}: cand {n} cells/doc {ncell.mean():.2f} tau {tau:.2f} (valid {ok}) median approx {np.median(approx[have]):.2f} "
                f"| UB0>=tau {ideal} ({100.0 * ideal / n:.1f} %)  UBmax>=tau {ideal_max}")
        for mult in (1, 2, 4):
            order = np.argsort(-ub0)[:NSEL * mult]
            a = approx[order]
            if not np.isfinite(a).all():
                line += f" | S0 x{mult}: n/a"
                continue
            tau0 = np.sort(a)[-NSEL]
            line += f" | S0 x{mult}: tau0 {tau0:.2f} surv {int((ub0 >= tau0).sum())}"
        print(line, flush=True)
        # quantised accumulators: unit such that no sum can overflow 16 bits; gains rounded up
        gain = G["gain"][qi]; kept = G["kept"][qi]
        for bits in (8, 16):
            unit = gain.sum() / (2 ** bits - 1 - kept.sum())
            extra = ncell.astype(np.float32) * unit            # worst-case rounding of a document's cells
            surv = int((ub0 + extra >= tau).sum())
            print(f"      {bits}-bit accumulators: unit {unit:.4f} score units, UB0+rounding >= tau {surv}")


if __name__ == "__main__":
    main()
