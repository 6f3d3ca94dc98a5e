"""GPU: NetVLAD-FC pooling head through the C ABI -- against the reference module's golden outputs
(small configurations) and against the pinned numpy oracle at the reference's full size
(64 clusters x 512 channels x 48x48 positions -> 512-D, model/netvlad_fc.py, loop_detector.h:97)."""
import numpy as np
import pytest

from test_oracle_vlad import CASES, load, load_gating

pytestmark = pytest.mark.gpu
TOL = 2e-5  # absolute, on descriptors of norm ~1 (fp32 everywhere; summation orders differ)


@pytest.mark.parametrize("case", CASES)
def test_matches_reference_goldens(capi, case):
    x, w, b, c, fc, y = load(case)
    m = capi.NetVladFC(w, c, fc, conv_b=b)
    gate = load_gating(case)
    if gate is not None:          # GatingContext after the FC (model/netvlad_fc.py:106-107)
        m.set_gating(*gate)
    got = m.forward(x)
    if gate is not None:          # and off again: the un-gated descriptor differs
        m.set_gating(None)
        assert np.abs(m.forward(x) - y).max() > 1e-3
    m.close()
    assert got.shape == y.shape
    assert np.abs(got - y).max() < TOL * max(1.0, np.abs(y).max())


@pytest.mark.parametrize("n,K,C,H,W,out", [(1, 64, 512, 48, 48, 512), (3, 64, 512, 7, 5, 512),
                                           (9, 24, 100, 10, 13, 40), (2, 1, 8, 1, 1, 3)])
def test_full_size_and_ragged_vs_oracle(capi, n, K, C, H, W, out):
    from oracle import vlad_oracle
    rng = np.random.default_rng(K * 1000 + C)
    x = np.maximum(rng.standard_normal((n, C, H, W)), 0).astype(np.float32)
    w = (rng.standard_normal((K, C)) * 4.0 / np.sqrt(C)).astype(np.float32)
    b = (rng.standard_normal(K) * 0.1).astype(np.float32) if K % 2 == 0 else None
    c = rng.random((K, C)).astype(np.float32)
    fc = (rng.standard_normal((K * C, out)) / np.sqrt(C)).astype(np.float32)
    m = capi.NetVladFC(w, c, fc, conv_b=b)
    got = m.forward(x)
    m.close()
    ref = vlad_oracle.netvlad_fc_forward(x, w, b, c, fc)
    assert np.abs(got - ref).max() < TOL * max(1.0, np.abs(ref).max())


def test_zero_feature_positions_and_errors(capi):
    from oracle import vlad_oracle
    rng = np.random.default_rng(1)
    x = np.maximum(rng.standard_normal((1, 32, 5, 5)), 0).astype(np.float32)
    x[0, :, 2, 3] = 0.0  # an all-zero position: F.normalize leaves it at zero (eps 1e-12)
    w = rng.standard_normal((8, 32)).astype(np.float32)
    c = rng.random((8, 32)).astype(np.float32)
    fc = rng.standard_normal((256, 16)).astype(np.float32)
    m = capi.NetVladFC(w, c, fc)
    got = m.forward(x)
    m.close()
    assert np.abs(got - vlad_oracle.netvlad_fc_forward(x, w, None, c, fc)).max() < TOL * 4
    with pytest.raises(capi.GlocError):
        capi.NetVladFC(np.zeros((65, 8), np.float32), np.zeros((65, 8), np.float32), np.zeros((520, 4), np.float32))
