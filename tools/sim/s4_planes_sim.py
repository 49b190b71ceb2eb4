#!/usr/bin/env python3
# NOTE (round 5): this simulation models the u8 table of rounds 2-4, u = floor((x / s + 1) * 127.5) + 1 over [-s, s] -- the table the
# committed outputs under profiles/r03_sim_* / r04_sim_* were produced with.  The shipped table now spans the positive scores only
# (u = floor(max(x, 0) / s * 254) + 1, np_kernels.h): shares and level spacings carry over (they are ratios of the per-centroid
# maxima), absolute thresholds and survivor counts do not (the slack is half as wide in score units).
"""CPU simulation of a bit-PLANE form of the S4 hot bound (design aid, not product code).

The hot bound U'(d) = sum_q max(Lambda, max_{c in codes(d), c hot} u[q,c]) folds 32 byte maxima per table row.  If the
values above Lambda are rounded UP to one of P levels l_1 < ... < l_P, a row becomes P bit planes (plane i = the tokens whose
value reaches l_i), the max over rows is a bitwise OR and the bound is  Lq * Lambda + sum_i (l_i - l_{i-1}) * popcount(plane_i):
U''(d) >= U'(d) >= U(d).  This script measures what the rounding costs: |S1|, |S2| and the survivor set of the three-step
cut (np_search.hip) under U' and under U'' for several level schemes, on the metric corpus.
"""
import os, sys, time
import numpy as np
from multiprocessing import Pool
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import s4_hot_sim as H

LQ, NQ, NDOCS, spec = H.LQ, H.NQ, H.NDOCS, H.spec
HOT = [float(x) for x in os.environ.get("SIM_HOT", "0.10").split(",")]


def levels(kind, P, lam, hot_vals):
    top = 255
    if kind == "uniform":
        return np.unique(np.ceil(lam + (top - lam) * np.arange(1, P + 1) / P).astype(np.int64))
    if kind == "sqrt":     # finer near the top than near Lambda
        w = np.sqrt(np.arange(1, P + 1) / P)
        return np.unique(np.ceil(lam + (top - lam) * w).astype(np.int64))
    if kind == "sq":       # finer near Lambda, mildly
        w = (np.arange(1, P + 1) / P) ** 1.5
        return np.unique(np.maximum(np.ceil(lam + (top - lam) * w), lam + 1 + np.arange(P)).astype(np.int64))
    if kind.startswith("pw"):   # power spacing: level j at Lambda + span * (j / P) ** e  (e > 1: finer next to Lambda)
        e = float(kind[2:])
        w = (np.arange(1, P + 1) / P) ** e
        return np.unique(np.maximum(np.ceil(lam + (top - lam) * w), lam + 1 + np.arange(P)).astype(np.int64))
    if kind == "geo":      # fine next to Lambda, doubling steps
        w = 2.0 ** np.arange(P); w = np.cumsum(w) / w.sum()
        return np.unique(np.maximum(np.ceil(lam + (top - lam) * w), lam + 1 + np.arange(P)).astype(np.int64))
    if kind == "quant":    # quantiles of the table entries above Lambda (half, 3/4, 7/8 ... of them below each level)
        qs_ = 1.0 - 0.5 ** np.arange(1, P)
        l = np.quantile(hot_vals, qs_, method="higher").astype(np.int64)
        return np.unique(np.concatenate([l, [top]]))
    raise ValueError(kind)


def main():
    cen = H.synth.centroids(spec)
    qs, src = H.synth.make_queries(spec, NQ, n_tokens=LQ, cen=cen)
    nprobe, tcs = 32, 0.4
    cellmasks, tabs = [], []
    for q in qs:
        QC = q @ cen.T
        cells = set()
        for row in QC:
            cells |= set(np.argpartition(-row, nprobe)[:nprobe].tolist())
        cells = np.array(sorted(cells))
        cells = cells[QC[:, cells].max(axis=0) >= tcs]
        m = np.zeros(65536, bool); m[cells] = True
        cellmasks.append(m)
        s = 1.001 * np.linalg.norm(q, axis=1).max() * 1.0001
        tabs.append((np.floor((QC / s + 1.0) * 127.5) + 1).astype(np.int32))
    t0 = time.time()
    jobs = [(d, min(d + H.CH, NDOCS), cellmasks) for d in range(0, NDOCS, H.CH)]
    with Pool(8) as p:
        res = p.map(H.scan, jobs, chunksize=1)
    print("scan", round(time.time() - t0, 1), "s", flush=True)
    n_sel = 1024
    for qi in range(NQ):
        ids = np.concatenate([r[qi][0] for r in res]); C = np.concatenate([r[qi][1] for r in res])
        u = tabs[qi]; n = ids.size
        M = u.max(axis=0)
        U = np.zeros(n, np.int64)
        for i0 in range(0, n, 4096):
            cc = C[i0:i0 + 4096].astype(np.int64)
            U[i0:i0 + 4096] = u[:, cc].max(axis=2).sum(axis=0)
        thrU = np.sort(U)[-n_sel] - (LQ + 2)
        print(f"q{qi}: cand {n} exact-U survivors {int((U >= thrU).sum())} (U n_sel-th {np.sort(U)[-n_sel]}, median {np.median(U)})", flush=True)
        for f in HOT:
            lam = int(np.sort(M)[int((1 - f) * 65536) - 1])
            hot = M > lam
            hm = np.zeros((n, LQ), np.int32)                 # per (doc, token): max over the doc's hot codes, at least Lambda
            for i0 in range(0, n, 4096):
                cc = C[i0:i0 + 4096].astype(np.int64)
                g = np.where(hot[cc][None], u[:, cc], 0)     # [Lq, m, 300]
                hm[i0:i0 + 4096] = np.maximum(g.max(axis=2), lam).T
            hot_vals = u[:, hot][u[:, hot] > lam]

            def report(name, B):
                assert (B >= U).all()
                for mult in [int(x) for x in os.environ.get("SIM_S1MULT", "1").split(",")]:
                    # S1 = the mult * n_sel documents with the largest bound: a larger sample of exact bounds lifts tau
                    thr1 = np.sort(B)[-min(mult * n_sel, n)]
                    S1 = B >= thr1
                    tau = np.sort(U[S1])[-n_sel] - (LQ + 2)
                    S2 = B >= tau
                    S3 = S2 & (U >= np.sort(U[S2])[-n_sel] - (LQ + 2))
                    miss = int(((U >= thrU) & ~S3).sum())
                    print(f"   f={f:.2f} lam={lam} {name:28s} S1x{mult} slack mean {np.mean(B - U):6.1f} |S1| {int(S1.sum()):5d} |S2| {int(S2.sum()):6d} "
                          f"|S3| {int(S3.sum())} missed {miss}", flush=True)
            report("exact hot bound U'", hm.sum(axis=1).astype(np.int64))
            kinds = os.environ.get("SIM_KINDS", "uniform:8,geo:8,quant:8,uniform:16,geo:16,quant:16,quant:4")
            for kind, P in [(k.split(":")[0], int(k.split(":")[1])) for k in kinds.split(",")]:
                l = levels(kind, P, lam, hot_vals)
                idx = np.searchsorted(l, hm, side="left")            # smallest level >= value
                qz = np.where(hm <= lam, lam, l[np.minimum(idx, l.size - 1)])
                report(f"{kind}-{P} {l.tolist() if l.size <= 8 else ''}", qz.sum(axis=1).astype(np.int64))


if __name__ == "__main__":
    main()
