"""CPU: the NetVLAD-FC oracle (numpy) against the reference module's golden outputs."""
import glob
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "vlad_*.npz")))


def load(case):
    g = np.load(os.path.join(GOLDEN, case + ".npz"))
    b = g["conv_b"] if g["conv_b"].size else None
    return g["x"], g["conv_w"], b, g["centroids"], g["fc_w"], g["y"]


def load_gating(case):
    """(gating_w, scale, shift) of a fixture made with GatingContext (BatchNorm1d in eval mode), else None."""
    from oracle import vlad_oracle
    g = np.load(os.path.join(GOLDEN, case + ".npz"))
    if "gating_w" not in g:
        return None
    scale, shift = vlad_oracle.fold_batch_norm(g["bn_weight"], g["bn_bias"], g["bn_mean"], g["bn_var"], float(g["bn_eps"]))
    return g["gating_w"], scale, shift


@pytest.mark.parametrize("case", CASES)
def test_vlad_oracle_matches_reference_goldens(case):
    from oracle import vlad_oracle
    x, w, b, c, fc, y = load(case)
    got = vlad_oracle.netvlad_fc_forward(x, w, b, c, fc)
    gate = load_gating(case)
    if gate is not None:
        got = vlad_oracle.gating_forward(got, *gate)
    assert got.shape == y.shape
    assert np.abs(got - y).max() < 2e-6 * max(1.0, np.abs(y).max())
