"""Independent numpy restatement of the next-plaid search path -- TEST INFRASTRUCTURE ONLY.

Written separately from plaid_oracle.c (different data structures, numpy BLAS matmul instead of
the k-ordered FMA chain) so that agreement between the two is evidence, not tautology.  Values
therefore agree with the C oracle to fp32 round-off (~1e-6), not bitwise; tests compare with a
stated tolerance.  Citations are into /root/reference/next-plaid/src.
"""
from __future__ import annotations

import numpy as np

NEG_INF = np.float32(-np.inf)


# -- comparators (search.rs:110-133) ------------------------------------------

def _order_key(x: np.ndarray) -> np.ndarray:
    """Monotone uint32 key of f32::total_cmp for finite x; every non-finite maps to 0 (they all
    compare Equal to each other and below any finite value)."""
    x = np.ascontiguousarray(x, np.float32)
    b = x.view(np.uint32)
    key = np.where(b >> 31 == 1, ~b, b | np.uint32(0x80000000)).astype(np.uint32)
    # keep finite keys >= 1: the only finite pattern mapping to 0 would be ~0xFFFFFFFF (a NaN)
    return np.where(np.isfinite(x), key, np.uint32(0))


def finite_first_max(col: np.ndarray) -> np.float32:
    """iter.max_by(cmp_score_ascending) (search.rs:419-422): best finite value, or - if none is
    finite - the LAST element (max_by keeps the last of equal maxima)."""
    if col.size == 0:
        return NEG_INF
    f = np.isfinite(col)
    if f.any():
        return np.float32(col[f].max())
    return np.float32(col[-1])


# -- codec (codec.rs:168-214, 423-470) ----------------------------------------

def bucket_indices(packed: np.ndarray, nbits: int) -> np.ndarray:
    """[n, pd] bytes -> [n, pd*8/nbits] bucket ids.  Derived directly from the packing rule of
    quantize_residuals (codec.rs:384-396): bucket bits are emitted LSB-first into an MSB-first
    bit stream, so dim j's bucket = sum_b bit[j*nbits+b] << b."""
    bits = np.unpackbits(np.ascontiguousarray(packed, np.uint8), axis=1, bitorder="big")
    n = bits.shape[0]
    bits = bits.reshape(n, -1, nbits).astype(np.int64)
    return (bits << np.arange(nbits, dtype=np.int64)).sum(-1)


def decompress(packed, codes, centroids, bucket_weights, nbits):
    idx = bucket_indices(packed, nbits)[:, : centroids.shape[1]]
    out = centroids[np.asarray(codes, np.int64)].astype(np.float32) + bucket_weights.astype(np.float32)[idx]
    out = out.astype(np.float32)
    norm = np.maximum(np.sqrt((out * out).sum(1, dtype=np.float32)), np.float32(1e-12)).astype(np.float32)
    return (out / norm[:, None]).astype(np.float32)


# -- maxsim (maxsim.rs:270-315) -------------------------------------------------

def maxsim_score(q, d):
    s = (np.asarray(q, np.float32) @ np.asarray(d, np.float32).T).astype(np.float32)
    total = np.float32(0)
    for row in s:
        f = np.isfinite(row)
        if f.any():
            total = np.float32(total + row[f].max())
    return total


# -- search (search.rs:327-640) -------------------------------------------------

class NumpyIndex:
    def __init__(self, centroids, bucket_weights, ivf, ivf_lengths, doc_lengths, codes, residuals, nbits):
        self.centroids = np.asarray(centroids, np.float32)
        self.bucket_weights = np.asarray(bucket_weights, np.float32)
        self.ivf = np.asarray(ivf, np.int64)
        self.ivf_lengths = np.asarray(ivf_lengths, np.int64)
        self.ivf_offsets = np.concatenate([[0], np.cumsum(self.ivf_lengths)])
        self.doc_lengths = np.asarray(doc_lengths, np.int64)
        self.doc_offsets = np.concatenate([[0], np.cumsum(self.doc_lengths)])
        self.codes = np.asarray(codes, np.int64)
        self.residuals = np.asarray(residuals, np.uint8)
        self.nbits = nbits
        self.K, self.d = self.centroids.shape
        self.N = self.doc_lengths.size

    def doc_codes(self, doc):
        return self.codes[self.doc_offsets[doc]: self.doc_offsets[doc + 1]]

    def doc_embeddings(self, doc):
        s, e = self.doc_offsets[doc], self.doc_offsets[doc + 1]
        return decompress(self.residuals[s:e], self.codes[s:e], self.centroids, self.bucket_weights, self.nbits)

    # index.rs:1142-1156
    def get_candidates(self, cells):
        parts = [self.ivf[self.ivf_offsets[c]: self.ivf_offsets[c + 1]] for c in cells if 0 <= c < self.K]
        if not parts:
            return np.zeros(0, np.int64)
        return np.unique(np.concatenate(parts))

    def _top_n(self, scores, ids, n):
        """top-n by cmp_score_descending, ties -> lower id (reference: unspecified)."""
        order = np.lexsort((ids, -_order_key(scores).astype(np.int64)))
        return ids[order[:n]]

    def _probe_dense(self, qc, p, subset):
        K, N = self.K, self.N
        if subset is not None:
            elig = set()
            for doc in subset:
                if 0 <= doc < N:
                    elig.update(self.doc_codes(doc).tolist())
            elig = np.array(sorted(elig), np.int64)
        else:
            elig = None
        nprobe = p.n_ivf_probe
        if elig is not None and elig.size > 0:
            scaled = p.n_ivf_probe * N // len(subset) if len(subset) > 0 else p.n_ivf_probe
            nprobe = min(max(scaled, p.n_ivf_probe), elig.size)
        pool = np.arange(K, dtype=np.int64) if elig is None else elig
        cells = set()
        for q in range(qc.shape[0]):
            n = min(nprobe, pool.size)
            cells.update(self._top_n(qc[q, pool], pool, n).tolist())
        if p.centroid_score_threshold is not None:
            cells = {c for c in cells if finite_first_max(qc[:, c]) >= np.float32(p.centroid_score_threshold)}
        return sorted(cells)

    def _probe_batched(self, q, p):
        """search.rs:140-254 including the 'ever pushed into a slab-local heap' max_scores rule."""
        K, n_probe, bs = self.K, p.n_ivf_probe, p.centroid_batch_size
        Lq = q.shape[0]
        final = [[] for _ in range(Lq)]          # lists of (score, c)
        max_scores = {}
        better = lambda a, b: _order_key(np.float32([a]))[0] > _order_key(np.float32([b]))[0]
        for b0 in range(0, K, bs):
            b1 = min(b0 + bs, K)
            s = (q @ self.centroids[b0:b1].T).astype(np.float32)
            for qi in range(Lq):
                heap = []                         # (score, c)
                for lc in range(b1 - b0):
                    sc, c = s[qi, lc], b0 + lc
                    pushed = False
                    if len(heap) < n_probe:
                        heap.append((sc, c)); pushed = True
                    else:
                        # peek = lowest score; ties -> largest c
                        w = min(range(len(heap)), key=lambda i: (_order_key(np.float32([heap[i][0]]))[0], -heap[i][1]))
                        if better(sc, heap[w][0]):
                            heap[w] = (sc, c); pushed = True
                    if pushed:
                        if c in max_scores:
                            if better(sc, max_scores[c]):
                                max_scores[c] = sc
                        else:
                            max_scores[c] = sc
                heap.sort(key=lambda e: (-int(_order_key(np.float32([e[0]]))[0]), e[1]))
                for e in heap:
                    f = final[qi]
                    if len(f) < n_probe:
                        f.append(e)
                    else:
                        w = min(range(len(f)), key=lambda i: (_order_key(np.float32([f[i][0]]))[0], -f[i][1]))
                        if better(e[0], f[w][0]):
                            f[w] = e
        cells = {c for f in final for _, c in f}
        if p.centroid_score_threshold is not None:
            t = np.float32(p.centroid_score_threshold)
            cells = {c for c in cells if max_scores.get(c, NEG_INF) >= t}
        return sorted(cells)

    def search(self, query, p, subset=None, return_trace=False):
        q = np.asarray(query, np.float32)
        Lq = q.shape[0]
        batched = p.centroid_batch_size > 0 and self.K > p.centroid_batch_size
        qc = (q @ self.centroids.T).astype(np.float32)      # [Lq, K]
        cells = self._probe_batched(q, p) if batched else self._probe_dense(qc, p, subset)
        cand = self.get_candidates(cells)
        if subset is not None:
            cand = cand[np.isin(cand, np.asarray(subset, np.int64))]
        trace = dict(cells=np.asarray(cells, np.int64), cand=cand)
        if cand.size == 0:
            r = (np.zeros(0, np.int64), np.zeros(0, np.float32))
            return (r + (trace,)) if return_trace else r
        # search.rs:305-324 (dense) / :275-302 (sparse): plain '>' max, skip if still -inf
        approx = np.zeros(cand.size, np.float32)
        for i, doc in enumerate(cand):
            codes = self.doc_codes(doc)
            if codes.size == 0:
                continue
            sub = qc[:, codes]                              # [Lq, len]
            with np.errstate(invalid="ignore"):
                m = np.fmax.reduce(np.where(np.isnan(sub), NEG_INF, sub), axis=1)
            s = np.float32(0)
            for v in m:
                if v > NEG_INF:
                    s = np.float32(s + v)
            approx[i] = s
        trace["approx"] = approx
        order = np.lexsort((np.arange(cand.size), -_order_key(approx).astype(np.int64)))  # stable desc
        top = order[: p.n_full_scores]
        n_dec = max(p.n_full_scores // 4, p.top_k)
        sel = cand[top[:n_dec]]
        trace["sel"] = sel
        if sel.size == 0:
            r = (np.zeros(0, np.int64), np.zeros(0, np.float32))
            return (r + (trace,)) if return_trace else r
        exact = np.array([maxsim_score(q, self.doc_embeddings(d)) for d in sel], np.float32)
        trace["sel_exact"] = exact
        order2 = np.lexsort((np.arange(sel.size), -_order_key(exact).astype(np.int64)))
        k = min(p.top_k, sel.size)
        r = (sel[order2[:k]], exact[order2[:k]])
        return (r + (trace,)) if return_trace else r
