"""The numerical argument behind the bf16-pipe products of csrc/cloud_kernels.hip (cloud_bwd1_paper_kernel, cloud_rbwd_kernel),
restated in numpy so that it is checked without a GPU:

  * an fp32 value splits EXACTLY into three bf16 pieces, x = h + m + l with h = bf16(x), m = bf16(x - h), l = bf16(x - h - m)
    (round to nearest even, the differences exact in fp32) -- `split3` in the kernels;
  * with every bf16 x bf16 product exact in the fp32 accumulator, the six partial products l h' + h l' + m m' + m h' + h m' + h h'
    reproduce an fp32 dot product at least as accurately as the sequential fp32 fma chain the f32 MFMA instruction stands for
    (the three dropped terms are below 2^-24 of |x y|), while the three-term variant (h h' + h m' + m h') does not.
"""
import numpy as np


def bf16_rne(x):
    """float32 -> bfloat16 (round to nearest even) -> float32, like v_cvt_pk_bf16_f32."""
    u = np.asarray(x, np.float32).view(np.uint32).astype(np.uint64)
    u = ((u + (((u >> 16) & 1) + 0x7FFF)) >> 16) << 16
    return u.astype(np.uint32).view(np.float32)


def split3(x):
    x = np.asarray(x, np.float32)
    h = bf16_rne(x)
    r1 = (x - h).astype(np.float32)
    m = bf16_rne(r1)
    r2 = (r1 - m).astype(np.float32)
    return h, m, bf16_rne(r2)


def test_three_bf16_pieces_reproduce_fp32_exactly():
    rs = np.random.RandomState(0)
    x = np.concatenate([
        (rs.randn(200000) * np.exp(rs.uniform(-20, 20, 200000))).astype(np.float32),      # 17 decades
        np.float32([0.0, -0.0, 1.0, -1.0, 1.0 + 2.0 ** -23, 1.0 - 2.0 ** -24, 3.38e38, -3.38e38, 1e-30, 255.99998, 0.1, 1 / 3]),
        np.nextafter(np.float32(2.0) ** np.arange(-20, 20), np.float32(0)).astype(np.float32),   # all-ones mantissas
    ])
    h, m, l = split3(x)
    total = h.astype(np.float64) + m.astype(np.float64) + l.astype(np.float64)
    assert np.array_equal(total, x.astype(np.float64))
    # each piece really is a bf16 value (low 16 bits of the fp32 pattern clear)
    for p in (h, m, l):
        assert np.all((p.view(np.uint32) & 0xFFFF) == 0)
    # the one limit: magnitudes within half a bf16 ulp of FLT_MAX (>= 3.3895e38) round to infinity in bf16 -- outside anything a
    # feature, a Laplacian entry or a gradient of this path can be
    with np.errstate(invalid="ignore"):
        assert np.isinf(split3(np.float32([3.4e38]))[0][0])


def _chain_fp32(a, b):
    """Sequential fp32 accumulation along k (one rounding per product and per addition: an upper bound of the fma chain's error)."""
    acc = np.zeros((a.shape[0], b.shape[1]), np.float32)
    for k in range(a.shape[1]):
        acc = (acc + (a[:, k:k + 1] * b[k:k + 1, :]).astype(np.float32)).astype(np.float32)
    return acc


def _blocks(x, y, acc, K):
    """One v_mfma_f32_16x16x32_bf16 per 32 k: exact products, fp32 accumulator (modelled as: the block's exact sum, rounded once)."""
    for k0 in range(0, K, 32):
        acc = (acc.astype(np.float64) + x[:, k0:k0 + 32].astype(np.float64) @ y[k0:k0 + 32, :].astype(np.float64)).astype(np.float32)
    return acc


def test_six_partial_products_are_as_accurate_as_the_fp32_chain():
    rs = np.random.RandomState(1)
    K = 128                                               # the depth of P = [g_1 | X_0] . [X_0 | g_1]^T at 64 features
    A = (rs.randn(64, K) * rs.rand(64, K)).astype(np.float32)
    B = rs.randn(K, 64).astype(np.float32)
    ref = A.astype(np.float64) @ B.astype(np.float64)
    scale = np.abs(A).astype(np.float64) @ np.abs(B).astype(np.float64)
    e_chain = (np.abs(_chain_fp32(A, B) - ref) / scale).max()
    (ah, am, al), (bh, bm, bl) = split3(A), split3(B)
    six = [(al, bh), (ah, bl), (am, bm), (am, bh), (ah, bm), (ah, bh)]          # small terms first, as mfma_x6 issues them
    acc = np.zeros((64, 64), np.float32)
    for x, y in six:
        acc = _blocks(x, y, acc, K)
    e_six = (np.abs(acc - ref) / scale).max()
    acc3 = np.zeros((64, 64), np.float32)
    for x, y in [(ah, bm), (am, bh), (ah, bh)]:
        acc3 = _blocks(x, y, acc3, K)
    e_three = (np.abs(acc3 - ref) / scale).max()
    assert e_six <= e_chain, (e_six, e_chain)
    assert e_six < 2.0 ** -22                              # a few ulp of the terms' magnitude
    assert e_three > 4 * e_chain                           # the cheaper split is NOT fp32-accurate: why six products
