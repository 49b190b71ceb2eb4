"""The arithmetic behind the S4 filter, restated in numpy and checked as properties (no GPU, no library call).

The HIP path never computes the approximate score of most candidates: it keeps a document only while an integer UPPER bound
of its score reaches a threshold that rests on LOWER bounds of the n_sel-th best score.  The GPU
tests check the kernels against the oracle; this file checks the inequalities and the cut rule themselves on random
instances, including the round-4 forms (bit planes, floored exact level), ragged query lengths and heavy ties:

    lo(d) <= L(d) <= U(d) <= up(d),   U(d) <= U'(d) <= U''(d),   L - Lq - 1 <= 254 approx / s <= U + 1,
    survivors of the three-step cut  >=  { d : approx(d) is among the n_sel largest (ties included) }.

Round 5: the u8 table spans the POSITIVE scores only (u = floor(max(x, 0) / s * 254) + 1: twice the resolution of rounds
2-4's table over [-s, s]).  The bottom entry u = 1 is then a clipped one -- the score may be anything down to -s -- so a
LOWER bound charges 254 units for every real token whose maximum is the bottom entry: L(d) = U(d) - 254 #{q: max u = 1}.
Thresholds rest on lower bounds (the histograms of approx_ub_kernel count them), cuts compare upper bounds; the instances
below include documents of one or two codes, whose maxima against a query token are negative about as often as not.
"""
import numpy as np
import pytest

PLANES = 8


SCALE = 254.0


def u8_table(QC, s):
    """np_kernels.h, qc_gemm epilogue: u = floor(max(x * (1 / s), 0) * 254) + 1 in [1, 254], monotone in x."""
    inv = np.float32(1.0) / np.float32(s)
    return (np.floor(np.maximum(QC.astype(np.float32) * inv, np.float32(0.0)) * np.float32(SCALE)) + 1).astype(np.int64)


def clipped_charge(m, real=None):
    """approx_ub_kernel: 254 units off the lower bound per real token whose maximum is the bottom entry (or, with the floor,
    none of whose rows was requested: 0)."""
    real = np.ones(m.shape, bool) if real is None else real
    return 254 * int(((m <= 1) & real).sum())


def lam_for_share(M, permille):
    """hot_levels_kernel: the smallest level with at most permille/1000 of the centroids above it."""
    K = M.size
    limit = K * permille // 1000
    for lam in range(0, 256):
        if int((M > lam).sum()) <= limit:
            return lam
    return 255


def plane_levels(lam, top, pexp10=15):
    """hot_levels_kernel: t_j = Lambda + ceil(span * (j / 8)^e), strictly increasing while there is room, t_8 = top."""
    top = max(top, lam + 1)
    span = top - lam
    t = []
    for v in range(PLANES + 1):
        if v == 0:
            x = lam
        elif v == PLANES:
            x = top
        else:
            x = lam + int(np.ceil(np.float32(span) * np.float32((v / PLANES) ** (0.1 * pexp10))))
        t.append(min(max(x, min(lam + v, top)), top))
    return t


def make_instance(rng, K=512, n_docs=600, Lq=13, dim=16, codes_per_doc=(3, 40), ties=False):
    cen = rng.standard_normal((K, dim)).astype(np.float32)
    cen /= np.linalg.norm(cen, axis=1, keepdims=True)
    q = rng.standard_normal((Lq, dim)).astype(np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    if ties:                       # few distinct centroids: many documents share a score exactly
        cen[K // 8:] = cen[rng.integers(0, K // 8, K - K // 8)]
    QC = q @ cen.T                                       # [Lq, K]
    s = 1.001 * np.linalg.norm(q, axis=1).max() * np.linalg.norm(cen, axis=1).max()
    docs = [np.unique(rng.integers(0, K, rng.integers(*codes_per_doc))) for _ in range(n_docs)]
    return QC, s, docs


def bounds(QC, s, docs, hot_permille, warm_permille):
    Lq = QC.shape[0]
    u = u8_table(QC, s)                                  # [Lq, K]
    M = u.max(axis=0)
    lam = lam_for_share(M, hot_permille)
    lam2 = lam_for_share(M, max(warm_permille, hot_permille))
    t = plane_levels(lam, int(M.max()))
    approx = np.array([QC[:, c].max(axis=1).astype(np.float32).sum(dtype=np.float32) for c in docs])
    U = np.array([u[:, c].max(axis=1).sum() for c in docs])
    Lb = np.array([u[:, c].max(axis=1).sum() - clipped_charge(u[:, c].max(axis=1)) for c in docs])   # signed: may be < 0
    Uh, Up, up, lo = [], [], [], []
    for c in docs:
        hot = c[M[c] > lam]
        hm = np.maximum(u[:, hot].max(axis=1), lam) if hot.size else np.full(Lq, lam)
        Uh.append(hm.sum())
        # planes: round every value above Lambda up to the next level
        b = Lq * lam
        for j in range(PLANES):
            bit = (u[:, hot] > t[j]).any(axis=1) if hot.size else np.zeros(Lq, bool)
            b += (t[j + 1] - t[j]) * int(bit.sum())
        Up.append(b)
        kept = c[M[c] > lam2]
        km = u[:, kept].max(axis=1) if kept.size else np.zeros(Lq, np.int64)
        up.append(np.maximum(km, lam2).sum())
        lo.append(km.sum() - clipped_charge(km))
    return approx, U, Lb, np.array(Uh), np.array(Up), np.array(up), np.array(lo), lam, lam2


def nth_largest(x, n):
    return np.sort(x)[-min(n, x.size)]


@pytest.mark.parametrize("seed", range(6))
@pytest.mark.parametrize("hot,warm", [(100, 500), (60, 500), (10, 110), (300, 1000), (500, 300)])
def test_bounds_are_ordered_and_the_cut_keeps_the_top(seed, hot, warm):
    rng = np.random.default_rng(1000 + seed)
    Lq = int(rng.integers(1, 33))
    # every other seed: documents of one or two codes, so that many per-token maxima are negative (clipped entries)
    QC, s, docs = make_instance(rng, Lq=Lq, ties=(seed % 3 == 2), codes_per_doc=(1, 3) if seed % 2 else (3, 40))
    approx, U, Lb, Uh, Up, up, lo, lam, lam2 = bounds(QC, s, docs, hot, warm)
    assert lam2 <= lam
    assert (U <= Uh).all() and (Uh <= Up).all(), "hot bound / plane bound must dominate the exact bound"
    assert (lo <= Lb).all() and (Lb <= U).all() and (U <= up).all(), "floored exact level: lower and upper bound"
    # the bracket of the f32 score by the integer bounds (np_kernels.h): z < 1 covers the rounding of u and of the f32 sum
    scaled = SCALE * approx.astype(np.float64) / s
    assert (Lb - Lq - 1 <= scaled).all() and (scaled <= U + 1).all()
    if seed % 2:
        assert (Lb < U).any(), "the instance was meant to hold clipped entries"
    slack = Lq + 2
    # the kernels clamp a negative lower bound into histogram bin 0, and a threshold that reaches bin 0 means "the filter does
    # not apply" (everything is kept): harmless, documents lumped at 0 only ever define a threshold of 0
    Lb, lo = np.maximum(Lb, 0), np.maximum(lo, 0)
    for n_sel in (1, 7, 64, len(docs) + 5):
        true_top = approx >= nth_largest(approx, n_sel)              # ties included
        # single-level filter (round 2): the threshold counts lower bounds, the cut compares upper bounds
        thr1 = nth_largest(Lb, n_sel) - slack
        keep1 = (U >= thr1) | (thr1 <= 0)
        assert (keep1 | ~true_top).all()
        # three-step cut with the plane bound in front and the floored exact level behind (np_search.hip)
        S1 = Up >= nth_largest(Up, n_sel)
        tau = nth_largest(Lb[S1], n_sel) - slack                     # S1 keeps its exact bound (the histogram: its lower form)
        S2 = (Up >= tau) & ~S1
        low = np.where(S1, Lb, lo)                                   # what the histogram counts
        upp = np.where(S1, U, up)                                    # what the cut compares
        L = S1 | S2
        tau2 = nth_largest(low[L], n_sel) - slack
        assert tau2 >= tau, "the threshold can only rise"
        surv = L & (upp >= tau2)
        assert (surv | ~true_top).all(), f"n_sel={n_sel}: a document of the true top was cut"
        # and the filter is not vacuous on the larger instances
        if n_sel == 7 and len(docs) >= 500 and not (seed % 3 == 2) and not (seed % 2):   # (one-code documents: thresholds reach 0)
            assert surv.sum() < len(docs)


def test_padding_tokens_are_neutral():
    """A query padded to the row width: padding columns hold 0 in the u8 table, are never floored, and leave every bound
    of the real tokens unchanged (the slack follows the query's own token count, not the row width)."""
    rng = np.random.default_rng(7)
    QC, s, docs = make_instance(rng, Lq=20)
    approx, U, Lb, Uh, Up, up, lo, lam, lam2 = bounds(QC, s, docs, 100, 500)
    u = u8_table(QC, s)
    upad = np.vstack([u, np.zeros((12, u.shape[1]), np.int64)])      # rows 20..31: padding
    M = upad.max(axis=0)
    assert (M == u.max(axis=0)).all()
    for c, U0, up0, lo0 in zip(docs[:50], U, up, lo):
        assert upad[:, c].max(axis=1).sum() == U0
        kept = c[M[c] > lam2]
        km = upad[:, kept].max(axis=1) if kept.size else np.zeros(32, np.int64)
        floor = np.where(np.arange(32) < 20, lam2, 0)                # approx_ub_kernel FLOOR: real tokens only
        assert np.maximum(km, floor).sum() == up0
        assert km.sum() - clipped_charge(km, np.arange(32) < 20) == lo0      # padding tokens are never charged


def test_plane_levels_are_increasing_and_end_at_the_top():
    for lam, top in ((0, 255), (158, 255), (250, 255), (254, 255), (255, 255), (100, 101), (10, 14)):
        for e in (10, 15, 30):
            t = plane_levels(lam, top, e)
            assert t[0] == lam and t[-1] == max(top, lam + 1)
            assert all(a <= b for a, b in zip(t, t[1:]))
            room = max(top, lam + 1) - lam
            if room >= PLANES:
                assert all(a < b for a, b in zip(t, t[1:])), (lam, top, e, t)


# ---- round 6: the zeroth level (np_kernels.h "S3, ZEROTH filter level": gain_prep_kernel / gain_sweep_kernel) -------------------------

def probe(QC, nprobe):
    """search.rs:388-414: per token the top-nprobe centroids; returns the marked cells and theta_q = the nprobe-th best score."""
    Lq, K = QC.shape
    marks = np.zeros(K, bool)
    theta = np.zeros(Lq, np.float32)
    for q in range(Lq):
        top = np.argsort(-QC[q], kind="stable")[:nprobe]
        marks[top] = True
        theta[q] = QC[q, top].min()
    return marks, theta


def gain_bound(QC, s, docs, nprobe, depth, acc_bits=15, thr=None):
    """The level's arithmetic: ut_q = u(theta_q at the DEPTH), G(c) = sum_q max(0, u[q,c] - ut_q) for the cells marked at that
    depth, scaled by 2^-sh (rounded up) so that the sum over all of them fits the accumulator; U0 = base + (sum of the document's
    marked cells' scaled gains << sh).  Candidates = documents holding a cell the SEARCH probes (depth nprobe)."""
    u = u8_table(QC, s)
    real, _ = probe(QC, nprobe)
    if thr is not None:        # search.rs:417-425: a probed cell is kept iff its best score over the tokens reaches the threshold;
        real = real & (QC.max(axis=0) >= thr)   # the removed ones stay in the sweep as bound-only cells
    deep, theta = probe(QC, max(nprobe, depth))
    ut = u8_table(theta[:, None], s)[:, 0]
    G = np.maximum(u - ut[:, None], 0).sum(axis=0)               # [K]
    cells = np.nonzero(deep)[0]
    sh = 0
    assert cells.size < (1 << acc_bits) - 1      # np_search.hip runs the level only while depth x query tokens stays far below 2^15
    while (int(G[cells].sum()) >> sh) + cells.size > (1 << acc_bits) - 1:
        sh += 1
    Gs = np.where(deep, (G + (1 << sh) - 1) >> sh, 0)
    base = int(ut.sum())
    U0 = np.array([base + (int(Gs[c].sum()) << sh) for c in docs])
    cand = np.array([bool(real[c].any()) for c in docs])
    return U0, cand, base, sh, real, deep


@pytest.mark.parametrize("seed", range(8))
@pytest.mark.parametrize("nprobe,depth,acc_bits,thr", [(4, 4, 15, None), (2, 8, 15, None), (8, 32, 15, None), (4, 16, 10, None),
                                                        (8, 8, 15, 0.45), (4, 16, 15, 0.3)])
def test_zeroth_level_bound_and_cut(seed, nprobe, depth, acc_bits, thr):
    """U0 >= U for every candidate, at the probe's own depth and deeper (bound-only cells), with the gains scaled into a narrow
    accumulator; and the cut the pipeline makes with it -- tau0 = the n_sel-th largest LOWER bound among ANY set S0 of candidates
    (the ones with the largest U0, a random set, a set missing the best documents) minus the slack, candidates with U0 below it
    dropped -- never removes a document of the true top n_sel of the candidates (ties included).  With a centroid_score_threshold
    the cells it removes stay in the sum as bound-only cells and only the kept cells' documents are candidates."""
    rng = np.random.default_rng(7000 + seed)
    Lq = int(rng.integers(1, 33))
    QC, s, docs = make_instance(rng, K=256, n_docs=500, Lq=Lq, ties=(seed % 3 == 2), codes_per_doc=(1, 4) if seed % 2 else (3, 30))
    U0, cand, base, sh, real, deep = gain_bound(QC, s, docs, nprobe, depth, acc_bits, thr)
    assert (deep | ~real).all()
    if not cand.any():
        return                 # the threshold removed every probed cell: no candidates, nothing to check
    if acc_bits == 10:
        assert sh > 0, "the narrow accumulator was meant to force a scaling"
    u = u8_table(QC, s)
    U = np.array([u[:, c].max(axis=1).sum() for c in docs])
    Lb = np.maximum(np.array([u[:, c].max(axis=1).sum() - clipped_charge(u[:, c].max(axis=1)) for c in docs]), 0)
    approx = np.array([QC[:, c].max(axis=1).astype(np.float32).sum(dtype=np.float32) for c in docs])
    assert (U0[cand] >= U[cand]).all(), "the gain sum must dominate the exact bound"
    # a centroid outside the marks scores <= theta_q for every token: the premise of the bound
    _, theta = probe(QC, max(nprobe, depth))
    assert (QC[:, ~deep] <= theta[:, None] + 0).all()
    slack = Lq + 2
    ci = np.nonzero(cand)[0]
    for n_sel in (1, 5, 40):
        if ci.size <= n_sel:
            continue
        true_top = np.zeros(len(docs), bool)
        true_top[ci] = approx[ci] >= nth_largest(approx[ci], n_sel)
        order = ci[np.argsort(-U0[ci], kind="stable")]
        for S0 in (order[:3 * n_sel], rng.permutation(ci)[:max(3 * n_sel, n_sel + 1)], order[n_sel // 2:n_sel // 2 + 2 * n_sel + 1]):
            if S0.size < n_sel:
                continue
            tau0 = nth_largest(Lb[S0], n_sel) - slack
            keep = cand & ((U0 >= tau0) | (tau0 <= 0))
            assert (keep | ~true_top).all(), f"n_sel={n_sel}: the zeroth level cut a document of the true top"
    # the level is not vacuous: with the best bounds as S0 something is dropped on the instances with real lists
    if not seed % 2 and ci.size > 200 and depth >= 8 and thr is None:
        tau0 = nth_largest(Lb[order[:15]], 5) - slack
        assert (U0[ci] < tau0).any()
