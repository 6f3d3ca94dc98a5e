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


@pytest.mark.parametrize("case", CASES)
def test_vlad_oracle_matches_reference_goldens(case):
    from oracle import vlad_oracle
    x, w, b, c, fc, y = load(case)
    got = vlad_oracle.netvlad_fc_forward(x, w, b, c, fc)
    assert got.shape == y.shape
    assert np.abs(got - y).max() < 2e-6 * max(1.0, np.abs(y).max())
