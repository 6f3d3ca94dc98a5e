"""The algebra behind the split-bf16 coarse pass of the descriptor kNN (gloc3d_amd/csrc/knn_kernels.hpp, dist_bf16x3_kernel;
DESIGN.md section 2), restated in numpy: x = h + m + r with h = bf16(x), m = bf16(x - h), both conversions rounding to
nearest even as v_cvt_pk_bf16_f32 does, and  q d ~ qh dh + qh dm + qm dh.  What the device's completeness proof
relies on (knn.hip: 776 u of its bound) is checked here on vectors of every scale; the kernel itself is checked on the GPU
by results (tests/test_knn_gpu.py)."""
import numpy as np
import pytest


def bf16_rn(x):
    u = np.asarray(x, np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32)


@pytest.mark.parametrize("seed", range(6))
def test_split_is_exact_where_the_bound_says_so_and_small_where_it_says_so(seed):
    rng = np.random.default_rng(seed)
    x = (rng.standard_normal(1 << 16) * 10.0 ** rng.uniform(-8, 8, 1 << 16)).astype(np.float32)
    h = bf16_rn(x)
    r1 = x - h                                     # fp32 subtraction, as the kernel does it
    assert (r1.astype(np.float64) == x.astype(np.float64) - h.astype(np.float64)).all()      # exact
    assert (np.abs(r1) <= np.abs(x) * 2.0 ** -8).all()
    m = bf16_rn(r1)
    r2 = r1.astype(np.float64) - m.astype(np.float64)
    assert (np.abs(r2) <= np.abs(x).astype(np.float64) * 2.0 ** -16).all()


@pytest.mark.parametrize("seed", range(6))
def test_dropped_product_terms_stay_below_the_bound(seed):
    rng = np.random.default_rng(100 + seed)
    D = 4096
    q = (rng.standard_normal(D) * 10.0 ** rng.uniform(-4, 4)).astype(np.float32)
    d = (rng.standard_normal(D) * 10.0 ** rng.uniform(-4, 4)).astype(np.float32)
    if seed % 2:
        q, d = np.abs(q), np.abs(d)                # no cancellation in the sum: the worst case of the summed bound
    f = np.float64
    qh, dh = bf16_rn(q), bf16_rn(d)
    qm, dm = bf16_rn(q - qh), bf16_rn(d - dh)
    kept = qh.astype(f) * dh.astype(f) + qh.astype(f) * dm.astype(f) + qm.astype(f) * dh.astype(f)
    true = q.astype(f) * d.astype(f)
    per_term = np.abs(true - kept)
    assert (per_term <= 3.03 * 2.0 ** -16 * np.abs(true)).all()
    # summed, and doubled into the distance  |q|^2 + |d|^2 - 2 q.d :  3.03 * 2^-16 (|q|^2 + |d|^2)  =  776 u
    bound = 3.03 * 2.0 ** -16 * ((q.astype(f) ** 2).sum() + (d.astype(f) ** 2).sum())
    assert 2.0 * abs(true.sum() - kept.sum()) <= bound
    assert abs(3.03 * 2.0 ** -16 / 2.0 ** -24 - 776) < 1.0
