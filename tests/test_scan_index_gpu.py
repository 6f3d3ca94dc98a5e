"""GPU: the scan index itself (scan_store.hip + seg_sort.hpp).  The registration tests cannot see a wrong ORDER --
the 1-NN search is exact whatever order the points are in, a bad sort only makes it slow -- so the orders are
checked here directly: the curve order (hand-written segmented radix sort: sorted, a permutation, stable), the
launch order of the source groups (widest first), the kd order of a target index against a numpy statement of
the same rule, and a batch of scans against the same scans added one at a time."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def f2ord(x):
    u = np.ascontiguousarray(x, np.float32).view(np.uint32)
    return np.where(u & 0x80000000, ~u, u | 0x80000000).astype(np.uint32)


def kd_order_numpy(pts_curve):
    """scan_index.hpp's rule on points given in curve order: P = 16 * 2^L >= n positions, at every level every
    aligned node is sorted (stable) along the widest axis of its bounding box.  Returns the kd position -> curve
    position table."""
    n = len(pts_curve)
    L = 0
    while (16 << L) < n:
        L += 1
    cur = np.arange(n)
    for l in range(L):
        shift = 4 + L - l
        p = pts_curve[cur]
        node = np.arange(n) >> shift
        o = np.stack([f2ord(p[:, a]) for a in range(3)], 1)
        lo = np.full((1 << l, 3), 0xFFFFFFFF, np.uint32)
        hi = np.zeros((1 << l, 3), np.uint32)
        np.minimum.at(lo, node, o)
        np.maximum.at(hi, node, o)
        def ord2f(v):
            u = np.where(v & 0x80000000, v & 0x7FFFFFFF, ~v).astype(np.uint32)
            return u.view(np.float32)
        ext = ord2f(hi) - ord2f(lo)                      # fp32, as the kernel
        axis = np.zeros(1 << l, np.int64)
        e = ext[:, 0].copy()
        m = ext[:, 1] > e
        axis[m] = 1
        e[m] = ext[m, 1]
        axis[ext[:, 2] > e] = 2
        key = (node.astype(np.uint64) << np.uint64(32)) | o[np.arange(n), axis[node]].astype(np.uint64)
        cur = cur[np.argsort(key, kind="stable")]
    return cur


@pytest.fixture(scope="module")
def clouds():
    from gloc3d_amd import synth
    rng = np.random.default_rng(9)
    w = synth.make_world(1001)
    full = np.ascontiguousarray(synth.lidar_scan(w, synth.se3(3.0, (0.4, -0.2, 0.0)), seed=4)[:, :3])
    dup = rng.uniform(-20, 20, (3000, 3)).astype(np.float32)
    dup[1000:2000] = dup[:1000]                           # exact duplicates: equal keys, equal coordinates
    dup[2500:] = np.float32(1.25)                         # 500 identical points
    big = np.concatenate([full, full[:20000] + np.float32(0.013)])   # 143 929 points: 71 sort tiles (> 64: the scan
                                                                      # kernel's carry over tile blocks), 14 kd levels
    return [full, np.ascontiguousarray(full[::7]), dup, rng.normal(0, 8, (2049, 3)).astype(np.float32),
            rng.uniform(-5, 5, (17, 3)).astype(np.float32), rng.uniform(-5, 5, (1, 3)).astype(np.float32), big]


def test_curve_order_is_sorted_stable_and_a_permutation(capi, clouds):
    st = capi.ScanStore()
    for c in clouds:
        d = st.debug_index(st.add(c))
        n = len(c)
        assert not d["kd"] and (np.sort(d["perm"]) == np.arange(n)).all()
        assert (np.diff(d["keys"].astype(np.int64)) >= 0).all() and d["keys"].max() < (1 << 30)
        same = np.diff(d["keys"].astype(np.int64)) == 0   # equal keys keep ascending original index (stable sort)
        assert (np.diff(d["perm"].astype(np.int64))[same] > 0).all()
        # launch order: groups of 128 sorted points, widest bounding box first, equal extents in ascending id
        p = c[d["perm"]]
        ng = (n + 127) // 128
        ext = np.empty(ng, np.float32)
        for g in range(ng):
            q = p[g * 128:(g + 1) * 128]
            dd = (q.max(0) - q.min(0)).astype(np.float32)
            ext[g] = (dd[0] * dd[0] + dd[1] * dd[1]) + dd[2] * dd[2]
        assert (np.sort(d["order2"]) == np.arange(ng)).all()
        eo = ext[d["order2"]]
        assert (np.diff(eo) <= 0).all()
        assert (np.diff(d["order2"].astype(np.int64))[np.diff(eo) == 0] > 0).all()
    st.close()


def test_a_batch_of_scans_is_indexed_like_the_scans_one_by_one(capi, clouds):
    st = capi.ScanStore()
    one = [st.debug_index(st.add(c)) for c in clouds]
    ids = st.add_batch(clouds + [np.zeros((0, 3), np.float32)])
    assert st.points(ids[-1]) == 0
    for c, i, a in zip(clouds, ids, one):
        b = st.debug_index(i)
        assert (st.download(i) == c).all()
        for k in ("perm", "keys", "kpos", "order2"):
            assert (a[k] == b[k]).all(), k
    kitti = [np.concatenate([c, np.ones((len(c), 1), np.float32)], 1) for c in clouds[:3]]   # x y z i: stride 4
    for i, a in zip(st.add_batch(kitti), one):
        assert (st.debug_index(i)["perm"] == a["perm"]).all()
    st.close()


def test_kd_order_is_the_stated_rule(capi, clouds):
    st = capi.ScanStore()
    ids = st.add_batch(clouds)
    before = [st.debug_index(i) for i in ids]
    st.build_target_index_batch(ids)                      # one batch: scans of 13, 11, 8, 8, 1 and 0 levels together
    for c, i, b in zip(clouds, ids, before):
        d = st.debug_index(i)
        assert d["kd"] and (d["keys"] == b["keys"]).all()  # the curve keys stay the cold-start lookup
        want = kd_order_numpy(c[b["perm"]])               # kd position -> curve position
        assert (d["perm"] == b["perm"][want]).all()
        inv = np.empty(len(c), np.int64)
        inv[want] = np.arange(len(c))
        assert (d["kpos"] == inv).all()                   # curve position -> kd position
        assert (np.sort(d["order2"]) == np.arange((len(c) + 127) // 128)).all()
    one = capi.ScanStore()                                # the same scans re-sorted one at a time
    for c, i in zip(clouds, ids):
        j = one.build_target_index(one.add(c))
        assert (one.debug_index(j)["perm"] == st.debug_index(i)["perm"]).all()
    one.close()
    st.close()


def test_kd_cells_are_disjoint_along_the_split_axis(capi, clouds):
    """The property the search gains from: the two halves of every kd node are separated along the node's widest
    axis (lower half <= upper half), so sibling boxes overlap in at most a plane."""
    st = capi.ScanStore()
    c = clouds[0]
    i = st.build_target_index(st.add(c))
    p = c[st.debug_index(i)["perm"]]
    n = len(p)
    L = 0
    while (16 << L) < n:
        L += 1
    for l in (0, 3, 7, L - 1):
        size = 16 << (L - l)
        for node in range(0, min(1 << l, 40)):
            a, b = node * size, min((node + 1) * size, n)
            if b - a <= size // 2:
                continue
            q = p[a:b]
            ax = int(np.argmax(q.max(0) - q.min(0)))
            assert q[:size // 2, ax].max() <= q[size // 2:, ax].min()
    st.close()
