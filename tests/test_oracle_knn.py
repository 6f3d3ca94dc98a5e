"""CPU: the kNN oracle against the reference's golden vectors and (where built) the reference itself."""
import numpy as np
import pytest

from util import KNN_CASES, bits, load_knn_case


@pytest.mark.parametrize("case", KNN_CASES)
def test_oracle_matches_reference_goldens(oracle_mod, case):
    db, q, k, g_idx, g_bits = load_knn_case(case)
    idx, d2 = oracle_mod.knn_search(db, q, k, threads=4)
    assert (idx == g_idx).all()
    assert (bits(d2) == g_bits).all()


def test_oracle_matches_reference_live(oracle_mod):
    if not oracle_mod.have_ref():
        pytest.skip("oracle/_ref not built (no /root/reference here)")
    from gloc3d_amd import synth
    db = synth.descriptors_iid(91, 0, 2000, 128)
    q = synth.descriptors_iid(92, 0, 16, 128)
    a = oracle_mod.knn_search(db, q, 20)
    b = oracle_mod.ref_knn_search(db, q, 20)
    assert (a[0] == b[0]).all() and (bits(a[1]) == bits(b[1])).all()


def test_l2_eval_order_and_tail(oracle_mod):
    # groups of four then a scalar tail (nanoflann.hpp:463-485): reproduce by hand in float32
    rng = np.random.default_rng(0)
    for dim in (1, 3, 4, 7, 510, 512):
        a = rng.standard_normal(dim).astype(np.float32)
        b = rng.standard_normal(dim).astype(np.float32)
        r = np.float32(0)
        d = 0
        while d + 3 < dim:
            e = a[d:d + 4] - b[d:d + 4]
            s = e * e
            r = np.float32(r + np.float32(np.float32(np.float32(s[0] + s[1]) + s[2]) + s[3]))
            d += 4
        while d < dim:
            e = np.float32(a[d] - b[d])
            r = np.float32(r + np.float32(e * e))
            d += 1
        got = np.float32(oracle_mod.lib().oracle_l2_eval(a, b, dim))
        assert bits(got) == bits(r), dim


def test_ragged_and_window(oracle_mod):
    from gloc3d_amd import synth
    db = synth.descriptors_iid(5, 0, 50, 16)
    q = synth.descriptors_iid(6, 0, 3, 16)
    idx, d2 = oracle_mod.knn_search(db, q, 20, first_row=40, last_row=45)  # 5 rows < k
    assert (idx[:, 5:] == np.iinfo(np.uint64).max).all()
    assert (d2[:, 5:] == np.finfo(np.float32).max).all()
    assert ((idx[:, :5] >= 40) & (idx[:, :5] < 45)).all()
    assert (np.diff(d2[:, :5], axis=1) >= 0).all()
    # empty window
    idx, d2 = oracle_mod.knn_search(db, q, 4, first_row=10, last_row=10)
    assert (idx == np.iinfo(np.uint64).max).all()


def test_duplicates_come_out_in_index_order(oracle_mod):
    from gloc3d_amd import synth
    db = synth.descriptors_iid(7, 0, 40, 32)
    db[10] = db[3]
    db[25] = db[3]
    q = db[3:4].copy()
    idx, d2 = oracle_mod.knn_search(db, q, 5)
    assert list(idx[0, :3]) == [3, 10, 25] and (d2[0, :3] == 0).all()


def test_recall_first_hit_semantics(oracle_mod):
    idx = np.array([[5, 9, 1, 2], [7, 7, 7, 7], [3, 4, 0, 8]], np.uint64)
    # q0: {1,9}; q1: no positives (skipped, global_localization.cpp:226); q2: {100, 8}
    valid, rec = oracle_mod.recall_at(idx, [[1, 9], [], [100, 8]], k_values=(1, 2, 4))
    assert valid == 2
    assert rec == [0.0, 0.5, 1.0]
