"""CPU statement of the storage geometry the device index uses for shapes without a kernel instantiation
(np_internal.h storage_dim / storage_nbits, np_index.hip repack_rows_kernel), checked against the oracle's codec:
a 1-bit row widened to 2-bit segments with weights {w0, w1, 0, 0}, zero-padded to the storage width, reconstructs the
same values as the file row (codec.rs:168-214 LUTs, 423-470 decompress)."""
import numpy as np
import pytest

from helpers import O


def widen_1_to_2(rows: np.ndarray) -> np.ndarray:
    """repack_rows_kernel, widen = 1: file byte (8 dims, first dim in bit 7) -> two bytes of four 2-bit segments, each
    holding bucket << 1 (= bitrev2(bucket))."""
    n, lpd = rows.shape
    out = np.zeros((n, 2 * lpd), np.uint8)
    for half in (0, 1):
        nib = (rows >> (0 if half else 4)) & 15
        out[:, half::2] = ((nib & 8) << 4) | ((nib & 4) << 3) | ((nib & 2) << 2) | ((nib & 1) << 1)
    return out


def narrow_2_to_1(rows: np.ndarray) -> np.ndarray:
    """unpack_rows_kernel, widen = 1"""
    hi, lo = rows[:, 0::2].astype(np.uint32), rows[:, 1::2].astype(np.uint32)
    f = lambda b: ((b >> 4) & 8) | ((b >> 3) & 4) | ((b >> 2) & 2) | ((b >> 1) & 1)
    return ((f(hi) << 4) | f(lo)).astype(np.uint8)


@pytest.mark.parametrize("dim", [8, 40, 64, 128])
def test_one_bit_rows_as_two_bit_rows(dim):
    rng = np.random.default_rng(dim)
    n, K = 200, 32
    cen = rng.standard_normal((K, dim)).astype(np.float32)
    codes = rng.integers(0, K, n).astype(np.int64)
    rows = rng.integers(0, 256, (n, dim // 8), dtype=np.uint8)
    w1 = np.array([-0.03, 0.05], np.float32)
    ref = O.decompress(rows, codes, cen, w1, 1)
    wide = widen_1_to_2(rows)
    assert wide.shape == (n, dim // 4)
    got = O.decompress(wide, codes, cen, np.array([w1[0], w1[1], 0, 0], np.float32), 2)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
    assert np.array_equal(narrow_2_to_1(wide), rows)


@pytest.mark.parametrize("dim,nbits", [(48, 4), (50, 4), (24, 2), (72, 8)])
def test_zero_padded_rows_score_the_same(dim, nbits):
    """Padding centroids / queries with zeros and residual rows with zero BYTES leaves every q . (c + w) unchanged as long
    as the norm stops at the file dim: the padded dims multiply a zero query value."""
    rng = np.random.default_rng(dim * 10 + nbits)
    n, K, sdim = 64, 16, (dim + 31) // 32 * 32
    cen = rng.standard_normal((K, dim)).astype(np.float32)
    codes = rng.integers(0, K, n).astype(np.int64)
    pd, spd = dim * nbits // 8, sdim * nbits // 8
    rows = rng.integers(0, 256, (n, pd), dtype=np.uint8)
    w = np.sort(rng.standard_normal(1 << nbits)).astype(np.float32) * 0.05
    q = rng.standard_normal((5, dim)).astype(np.float32)
    ref = O.decompress(rows, codes, cen, w, nbits)          # normalised rows, file geometry
    cen_p = np.zeros((K, sdim), np.float32); cen_p[:, :dim] = cen
    rows_p = np.zeros((n, spd), np.uint8); rows_p[:, :pd] = rows
    q_p = np.zeros((5, sdim), np.float32); q_p[:, :dim] = q
    # un-normalised storage rows (what the S6 kernels multiply), scaled by the FILE-dim inverse norm
    lut = O.bucket_weight_indices_lookup(nbits)
    rev = O.byte_reversed_bits_map(nbits)
    raw = cen_p[codes] + w[lut[rev[rows_p]]].reshape(n, -1)[:, :sdim]
    inv = 1.0 / np.maximum(np.linalg.norm(raw[:, :dim].astype(np.float64), axis=1), 1e-12)
    got = (raw.astype(np.float64) @ q_p.T.astype(np.float64)) * inv[:, None]
    want = ref.astype(np.float64) @ q.T.astype(np.float64)
    assert np.allclose(got, want, rtol=0, atol=2e-6)
