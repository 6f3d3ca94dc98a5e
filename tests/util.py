"""Shared helpers for the tests: golden loading and input regeneration."""
import importlib.util
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")

_spec = importlib.util.spec_from_file_location("make_goldens", os.path.join(GOLDEN, "make_goldens.py"))
make_goldens = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(make_goldens)

KNN_CASES = list(make_goldens.KNN_CASES)


def load_knn_case(case):
    """Regenerate the inputs from their seeds and load the committed reference outputs."""
    db, q, k = make_goldens.knn_inputs(case)
    g = np.load(os.path.join(GOLDEN, case + ".npz"))
    # the inputs are regenerated, not stored: make sure they are the ones the goldens were made on
    assert int(q.view(np.uint32).sum(dtype=np.uint64)) == int(g["q_crc"])
    assert int(db.view(np.uint32).sum(dtype=np.uint64)) == int(g["db_crc"])
    return db, q, k, g["idx"], g["d2_bits"]


def load_nn3_case():
    g = np.load(os.path.join(GOLDEN, "nn3_scanpair.npz"))
    return g["src"], g["tgt"], g["idx"], g["d2_bits"]


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def random_se3(rng, max_deg=10.0, max_t=2.0):
    from gloc3d_amd import synth
    return synth.se3(rng.uniform(-max_deg, max_deg), (rng.uniform(-max_t, max_t), rng.uniform(-max_t, max_t),
                                                      rng.uniform(-0.2, 0.2)))
