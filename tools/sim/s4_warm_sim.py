#!/usr/bin/env python3
# NOTE (round 5): this simulation models the u8 table of rounds 2-4, u = floor((x / s + 1) * 127.5) + 1 over [-s, s] -- the table the
# committed outputs under profiles/r03_sim_* / r04_sim_* were produced with.  The shipped table now spans the positive scores only
# (u = floor(max(x, 0) / s * 254) + 1, np_kernels.h): shares and level spacings carry over (they are ratios of the per-centroid
# maxima), absolute thresholds and survivor counts do not (the slack is half as wide in score units).
"""CPU simulation of a FLOORED exact level for S4 (design aid, not product code).

The exact level gathers one u8 table row per (document of S2, distinct code): ~70 rows per document, of which ~90 % belong
to centroids no query token is close to.  If the rows of the centroids with M[c] = max_q u[q, c] <= Lambda2 are skipped and
every token's maximum is floored at Lambda2,
    U2(d) = sum_q max(Lambda2, max_{c in codes(d), M[c] > Lambda2} u[q, c])  >=  U(d)     (still an upper bound)
    L2(d) = sum_q          max_{c in codes(d), M[c] > Lambda2} u[q, c]       <=  U(d)     (a lower bound: tau may use it)
This script measures, per share of centroids kept, the row requests left and the size of the survivor set that goes on to
the exact f32 scores, starting from the plane bound of the shipped first level (8 planes, power-law 1.5 spacing, 10 % hot).
"""
import os, sys, time
import numpy as np
from multiprocessing import Pool
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import s4_hot_sim as H
import s4_planes_sim as P

LQ, NQ, NDOCS = H.LQ, H.NQ, H.NDOCS
KEEP = [float(x) for x in os.environ.get("SIM_KEEP", "0.2,0.3,0.4,0.5,0.7").split(",")]


def main():
    cen = H.synth.centroids(H.spec)
    qs, src = H.synth.make_queries(H.spec, NQ, n_tokens=LQ, cen=cen)
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
    slack = LQ + 2
    for qi in range(NQ):
        C = np.concatenate([r[qi][1] for r in res])
        u = tabs[qi]; n = C.shape[0]
        M = u.max(axis=0)
        U = np.zeros(n, np.int64)
        for i0 in range(0, n, 4096):
            cc = C[i0:i0 + 4096].astype(np.int64)
            U[i0:i0 + 4096] = u[:, cc].max(axis=2).sum(axis=0)
        # shipped first level: 10 % hot, 8 planes, power-law 1.5
        lam = int(np.sort(M)[int(0.9 * 65536) - 1])
        hot = M > lam
        hm = np.zeros((n, LQ), np.int32)
        for i0 in range(0, n, 4096):
            cc = C[i0:i0 + 4096].astype(np.int64)
            g = np.where(hot[cc][None], u[:, cc], 0)
            hm[i0:i0 + 4096] = np.maximum(g.max(axis=2), lam).T
        l = P.levels("pw1.5", 8, lam, None)
        idx = np.searchsorted(l, hm, side="left")
        B = np.where(hm <= lam, lam, l[np.minimum(idx, l.size - 1)]).sum(axis=1).astype(np.int64)
        S1 = B >= np.sort(B)[-n_sel]
        tau = np.sort(U[S1])[-n_sel] - slack
        S2 = np.nonzero(B >= tau)[0]
        U_s2 = U[S2]
        tau3 = np.sort(U_s2)[-n_sel] - slack
        S3 = int((U_s2 >= tau3).sum())
        Cs = C[S2].astype(np.int64)
        # distinct codes per S2 document
        srt = np.sort(Cs, axis=1)
        first = np.concatenate([np.ones((srt.shape[0], 1), bool), srt[:, 1:] != srt[:, :-1]], axis=1)
        nd_all = first.sum()
        print(f"q{qi}: cand {n} |S2| {S2.size} exact survivors {S3} rows {int(nd_all)} ({nd_all / max(S2.size, 1):.1f} per doc) tau {tau} -> {tau3}", flush=True)
        for keep in KEEP:
            lam2 = int(np.sort(M)[int((1 - keep) * 65536) - 1])
            warm = M > lam2
            rows = int((first & warm[srt]).sum())
            lo = np.zeros(S2.size, np.int64); up = np.zeros(S2.size, np.int64)
            for i0 in range(0, S2.size, 4096):
                cc = Cs[i0:i0 + 4096]
                g = np.where(warm[cc][None], u[:, cc], 0).max(axis=2)     # [Lq, m]; 0 = no kept code
                lo[i0:i0 + 4096] = g.sum(axis=0)
                up[i0:i0 + 4096] = np.maximum(g, lam2).sum(axis=0)
            assert (up >= U_s2).all() and (lo <= U_s2).all()
            # tau from lower bounds over S1 + S2 (every S1 document is in S2: its bound B >= thr1 >= ... >= tau)
            tau2 = max(tau, np.sort(lo)[-n_sel] - slack)
            surv = int((up >= tau2).sum())
            miss = int(((U_s2 >= tau3) & ~(up >= tau2)).sum())
            print(f"   keep {keep:.2f} lam2 {lam2:3d} rows {rows:8d} ({100.0 * rows / nd_all:5.1f} %) slack up-U mean {np.mean(up - U_s2):6.1f} U-lo mean {np.mean(U_s2 - lo):6.1f} "
                  f"tau2 {tau2} survivors {surv} (x{surv / max(S3, 1):.2f}) missed {miss}", flush=True)


if __name__ == "__main__":
    main()
