"""An oracle-backed stand-in for next_plaid_amd.dist.HipShardBackend (CPU tensors), so the
two-collective sharded protocol in dist.py can be exercised with world_size 2 over gloo.
Test infrastructure: mirrors what np_hip_search_phase_a / select_cut / phase_b / merge_topk compute."""
import contextlib

import numpy as np
import torch

from helpers import O, synth
from oracle.plaid_numpy import _order_key


def shard_arrays(a, b0, b1):
    off = np.concatenate([[0], np.cumsum(a["doc_lengths"])])
    codes = a["codes"][off[b0]:off[b1]]
    lens = a["doc_lengths"][b0:b1]
    ivf, ivl = synth.build_ivf(codes, lens, a["centroids"].shape[0])
    return dict(a, doc_lengths=lens, codes=codes, residuals=a["residuals"][off[b0]:off[b1]], ivf=ivf, ivf_lengths=ivl)


def n_sel_of(p):
    return min(max(p.n_full_scores // 4, p.top_k), p.n_full_scores)


def rank_keys(approx, gids):
    return (_order_key(approx).astype(np.uint64) << np.uint64(32)) | (np.uint64(0xFFFFFFFF) - gids.astype(np.uint64))


class OracleShardBackend:
    def __init__(self, a, b0, b1):
        s = shard_arrays(a, b0, b1)
        self.ix = O.OracleIndex(s["centroids"], s["bucket_weights"], s["ivf"], s["ivf_lengths"], s["doc_lengths"],
                                s["codes"], s["residuals"], s["nbits"])
        self.b0, self.b1 = b0, b1
        self.n_total = int(a["doc_lengths"].size)
        self.K = int(a["centroids"].shape[0])
        self.codes, self.off = s["codes"], np.concatenate([[0], np.cumsum(s["doc_lengths"])])
        self.device = torch.device("cpu")

    def num_partitions(self):
        return self.K

    def eligible(self, d_subset):
        """np_hip_subset_eligible: bitmap (u32 words as int32) of the codes of this shard's subset documents."""
        sub = d_subset.numpy()
        loc = sub[(sub >= self.b0) & (sub < self.b1)] - self.b0
        words = (self.K + 63) // 64 * 2
        bits = np.zeros(words, np.uint32)
        for d in loc:
            c = self.codes[self.off[d]:self.off[d + 1]].astype(np.int64)
            np.bitwise_or.at(bits, c >> 5, (np.uint32(1) << (c & 31).astype(np.uint32)))
        return torch.from_numpy(bits.view(np.int32).copy())

    def stream_ctx(self):
        return contextlib.nullcontext()

    def phase_a(self, d_q, d_qoff, h_qoff, params, d_subset=None, elig=None):
        q = d_q.numpy()
        sub = view = None
        if d_subset is not None:
            g = d_subset.numpy()
            sub = (g[(g >= self.b0) & (g < self.b1)] - self.b0).astype(np.int64)
            if elig is not None:
                w = elig.numpy().view(np.uint32)
                el = ((w[np.arange(self.K) >> 5] >> (np.arange(self.K) & 31).astype(np.uint32)) & 1).astype(np.uint8)
                view = (el, self.n_total, int(g.size))
        B, ns = len(h_qoff) - 1, n_sel_of(params)
        keys = np.zeros((B, max(ns, 1)), np.uint64)
        per_q = []
        # every shard candidate must be ranked, so ask the oracle for the whole approx list
        wide = O.SearchParameters(n_full_scores=params.n_full_scores, top_k=params.top_k,
                                  n_ivf_probe=params.n_ivf_probe, centroid_batch_size=params.centroid_batch_size,
                                  centroid_score_threshold=params.centroid_score_threshold)
        for b in range(B):
            qb = q[h_qoff[b]:h_qoff[b + 1]]
            t = self.ix.search(qb, wide, sub, trace=True, shard_view=view).trace
            k = np.sort(rank_keys(t.approx, t.cand + self.b0))[::-1][:ns]
            keys[b, :k.size] = k
            per_q.append((qb, k))
        return torch.from_numpy(keys[:, :ns].view(np.int64).copy()), (per_q, params)

    def select_cut(self, all_keys):
        k = all_keys.numpy().view(np.uint64)
        G, B, ns = k.shape
        cut = np.ones(B, np.uint64)
        for b in range(B):
            s = np.sort(k[:, b, :].ravel())[::-1]
            if ns > 0 and s[ns - 1] != 0:
                cut[b] = s[ns - 1]
        return torch.from_numpy(cut.view(np.int64).copy())

    def phase_b(self, state, cut):
        per_q, params = state
        cut = cut.numpy().view(np.uint64)
        k = max(params.top_k, 1)
        packed = np.zeros((len(per_q), 3 * k + 1), np.int64)
        for b, (qb, keys) in enumerate(per_q):
            keep = keys[keys >= cut[b]]
            gids = (np.uint64(0xFFFFFFFF) - (keep & np.uint64(0xFFFFFFFF))).astype(np.int64)
            exact = np.array([O.maxsim_score(qb, self.ix.get_document_embeddings(g - self.b0)) for g in gids], np.float32)
            order = np.lexsort((-(keep.astype(np.float64)), -_order_key(exact).astype(np.int64)))  # exact desc, key desc
            # float64 cannot order u64 keys exactly; do an exact two-level sort instead
            order = sorted(range(len(keep)), key=lambda i: (-int(_order_key(exact[i:i + 1])[0]), -int(keep[i])))
            order = order[: params.top_k]
            n = len(order)
            packed[b, :n] = gids[order]
            packed[b, k:k + n] = keep[order].view(np.int64)
            packed[b, 2 * k:2 * k + n] = exact[order].view(np.int32).astype(np.int64)
            packed[b, 3 * k] = n
        return torch.from_numpy(packed)

    def end(self, state):
        pass

    def merge(self, all_packed, top_k):
        p = all_packed.numpy()
        G, B, _ = p.shape
        k = max(top_k, 1)
        ids = np.zeros((B, k), np.int64)
        sc = np.zeros((B, k), np.float32)
        cnt = np.zeros(B, np.int32)
        for b in range(B):
            ent = []
            for g in range(G):
                n = int(p[g, b, 3 * k])
                for j in range(n):
                    s = np.int32(p[g, b, 2 * k + j]).view(np.float32)
                    ent.append((-int(_order_key(np.float32([s]))[0]), -int(np.int64(p[g, b, k + j]).view(np.uint64)),
                                int(p[g, b, j]), s))
            ent.sort()
            ent = ent[:top_k]
            cnt[b] = len(ent)
            for j, e in enumerate(ent):
                ids[b, j], sc[b, j] = e[2], e[3]
        return torch.from_numpy(ids), torch.from_numpy(sc), torch.from_numpy(cnt)
