"""Pins the oracle against every known-answer test the reference holds for the hot path
(SURVEY.md section 8c).  Each test names the reference test it restates.  CPU only."""
import numpy as np
import pytest

from helpers import O
from oracle import plaid_numpy as PN

NAN, INF = float("nan"), float("inf")


def test_colbert_score_1_7():
    # search.rs:684-705 test_colbert_score, maxsim.rs:392-413 test_maxsim_score_basic
    q = [[1, 0, 0, 0], [0, 1, 0, 0]]
    d = [[.5, .5, 0, 0], [.8, .2, 0, 0], [0, .9, .1, 0]]
    assert abs(O.maxsim_score(q, d) - 1.7) < 1e-5
    assert abs(PN.maxsim_score(q, d) - 1.7) < 1e-5


def test_maxsim_ignores_non_finite_row_entries_8_0():
    # maxsim.rs:497-507 (16x2 ones-query vs 15 x [0.5,0] + [NaN,0]; Lq*len = 256 -> GEMM path)
    q = np.tile([1.0, 0.0], (16, 1))
    d = np.tile([0.5, 0.0], (16, 1))
    d[15, 0] = NAN
    assert abs(O.maxsim_score(q, d) - 8.0) < 1e-5
    assert abs(PN.maxsim_score(q, d) - 8.0) < 1e-5


def test_simd_max():
    # maxsim.rs:415-430 test_simd_max
    assert abs(O.simd_max(np.arange(100)) - 99.0) < 1e-5
    assert abs(O.simd_max(np.arange(-50, 50)) - 49.0) < 1e-5
    assert abs(O.simd_max([1.0, 5.0, 3.0]) - 5.0) < 1e-5


def test_simd_max_ignores_non_finite_when_finite_values_exist():
    # maxsim.rs:432-441
    assert O.simd_max([1, 2, 3, NAN, 4, 5, 6, 7]) == 7.0
    assert O.simd_max([1, 2, 3, INF, 4, 5, 6, 7]) == 7.0


def test_cmp_score_descending_places_non_finite_scores_last():
    # search.rs:717-726
    import functools
    s = sorted([1.0, INF, 0.5, NAN], key=functools.cmp_to_key(lambda a, b: O.cmp_score_ascending(b, a)))
    assert s[0] == 1.0 and s[1] == 0.5 and not np.isfinite(s[2]) and not np.isfinite(s[3])


def test_score_replacement_and_max_score():
    # search.rs:728-742
    assert O.is_score_better(1.0, NAN) and O.is_score_better(1.0, INF)
    assert not O.is_score_better(NAN, 1.0) and not O.is_score_better(INF, 1.0)
    for a, b in [(NAN, 1.0), (1.0, NAN), (INF, 1.0), (1.0, INF)]:
        assert O.max_score(a, b) == 1.0


def test_search_params_default():
    # search.rs:707-715
    p = O.SearchParameters()
    assert (p.batch_size, p.n_full_scores, p.top_k, p.n_ivf_probe, p.centroid_score_threshold) == (2000, 4096, 10, 8, 0.4)
    assert p.centroid_batch_size == 100_000


def test_packbits_msb_first():
    # utils.rs:296-304 test_packbits_unpackbits
    bits = [1, 0, 1, 0, 1, 0, 1, 0, 1, 1, 1, 1, 0, 0, 0, 0]
    assert list(O.packbits(bits)) == [0b10101010, 0b11110000]


def test_bit_layout_known_answers():
    # nbits=4 byte 0xA3 -> buckets (5, 12); nbits=2 byte 0b10011100 -> (1, 2, 3, 0)
    t4 = O.bucket_weight_indices_lookup(4)[O.byte_reversed_bits_map(4)[0xA3]]
    t2 = O.bucket_weight_indices_lookup(2)[O.byte_reversed_bits_map(2)[0b10011100]]
    assert list(t4) == [5, 12] and list(t2) == [1, 2, 3, 0]


def test_quantize_decompress_roundtrip_4bit():
    # codec.rs:665-730: packed width dim*4/8, and sign agreement after the round trip
    dim = 8
    cutoffs = np.array([(i / 16.0 - 0.5) * 2.0 for i in range(1, 16)], np.float32)
    weights = np.array([((i + 0.5) / 16.0 - 0.5) * 2.0 for i in range(16)], np.float32)
    res = np.array([[-0.9, -0.7, -0.5, -0.3, 0.0, 0.3, 0.5, 0.9],
                    [-0.8, -0.4, 0.0, 0.4, 0.8, -0.6, 0.2, 0.6]], np.float32)
    packed = O.quantize_residuals(res, 4, cutoffs)
    assert packed.shape == (2, dim * 4 // 8)
    out = O.decompress(packed, [0, 0], np.zeros((4, dim), np.float32), weights, 4)
    for i in range(2):
        for j in range(dim):
            if abs(res[i, j]) > 0.2:
                assert (res[i, j] > 0) == (out[i, j] > 0) or abs(out[i, j]) < 0.1
    # and the packing is what an independent unpack of the bit stream sees
    b = PN.bucket_indices(packed, 4)
    expect = (res[:, :, None] > cutoffs[None, None, :]).sum(-1)
    assert np.array_equal(b, expect)


def test_codec_rejects_bad_nbits_convention():
    # codec.rs:161-166: nbits must divide 8 -- the LUT builders are only defined for those
    for nbits in (1, 2, 4, 8):
        assert O.bucket_weight_indices_lookup(nbits).shape == (256, 8 // nbits)


def test_rerank_known_answers():
    # next-plaid-api/tests/integration_tests.rs:2301-2376: 2.0 / 1.0 / 0.0 (+-0.01)
    q = np.array([[1, 0, 0, 0], [0, 1, 0, 0]], np.float32)
    d_both = np.array([[1, 0, 0, 0], [0, 1, 0, 0]], np.float32)
    d_one = np.array([[1, 0, 0, 0], [0, 0, 1, 0]], np.float32)
    d_none = np.array([[0, 0, 1, 0], [0, 0, 0, 1]], np.float32)
    assert abs(O.rerank_maxsim(q, d_both) - 2.0) < 0.01
    assert abs(O.rerank_maxsim(q, d_one) - 1.0) < 0.01
    assert abs(O.rerank_maxsim(q, d_none) - 0.0) < 0.01
    with pytest.raises(ValueError):
        O.rerank_maxsim(np.array([[NAN, 0, 0, 0]], np.float32), d_both)


def test_unrolled_dot_matches_sum():
    g = np.random.default_rng(0)
    for n in (1, 7, 8, 9, 31, 128):
        x, y = g.standard_normal(n).astype(np.float32), g.standard_normal(n).astype(np.float32)
        from oracle.oracle import lib, _ptr
        v = lib().po_unrolled_dot(_ptr(x), _ptr(y), n)
        assert abs(v - float(np.dot(x.astype(np.float64), y.astype(np.float64)))) < 1e-4


# ---- N3: index-time encode (codec.rs:297-411, index.rs:17-40) --------------------------------------------------
def _unit(x):
    return (x / np.linalg.norm(x, axis=1, keepdims=True)).astype(np.float32)


def test_compress_into_codes_last_of_equal_maxima_and_non_finite():
    rng = np.random.default_rng(5)
    cen = _unit(rng.standard_normal((64, 16)))
    cen[9] = cen[3]                                   # two identical rows: Iterator::max_by keeps the LAST
    z = rng.integers(0, 64, 500)
    x = _unit(cen[z] + 0.05 * rng.standard_normal((500, 16)))
    codes = O.compress_into_codes(x, cen)
    ref = np.argmax(x.astype(np.float64) @ cen.T.astype(np.float64), 1)
    ref = np.where(ref == 3, 9, ref)
    assert np.array_equal(codes, ref)
    assert not np.any(codes == 3) and np.any(codes == 9)
    # a NaN embedding makes every score non-finite: all equal, the last centroid wins (cmp_f32_for_max)
    bad = x[:1].copy()
    bad[0, 0] = np.nan
    assert O.compress_into_codes(bad, cen)[0] == 63
    # a finite score always beats a non-finite one
    cen2 = cen.copy()
    cen2[50:] = np.inf
    assert np.all(O.compress_into_codes(x, cen2) < 50)


def test_encode_tokens_matches_numpy_bit_layout_and_roundtrips():
    from next_plaid_amd import synth
    rng = np.random.default_rng(6)
    for nbits in (2, 4):
        dim = 32
        cen = _unit(rng.standard_normal((40, dim)))
        x = _unit(cen[rng.integers(0, 40, 300)] + 0.08 * rng.standard_normal((300, dim)))
        n = 1 << nbits
        cut = np.quantile((x - cen[np.argmax(x @ cen.T, 1)]).ravel(), [i / n for i in range(1, n)]).astype(np.float32)
        codes, packed = O.encode_tokens(x, cen, nbits, cut)
        assert packed.shape == (300, dim * nbits // 8)           # codec.rs:722-729
        res = x - cen[codes]
        buckets = (res[:, :, None] > cut[None, None, :]).sum(-1)  # strictly-below count (codec.rs:386)
        assert np.array_equal(synth.unpack_buckets(packed, nbits), buckets)
        # independent packing: bucket bits LSB-first, written MSB-first
        bits = ((buckets[:, :, None] >> np.arange(nbits)) & 1).astype(np.uint8).reshape(300, -1)
        assert np.array_equal(np.packbits(bits, axis=1, bitorder="big"), packed)
        # decompress(encode(x)) points the same way as x (codec.rs:700-713 sign agreement, cosine)
        wts = np.array([res[buckets == b].mean() if np.any(buckets == b) else 0.0 for b in range(n)], np.float32)
        rec = O.decompress(packed, codes, cen, wts, nbits)
        cos = (rec * x).sum(1)
        assert cos.min() > 0.9
