"""GPU: descriptor kNN through the C ABI -- bit-exact against the reference goldens and the oracle."""
import numpy as np
import pytest

from util import KNN_CASES, bits, load_knn_case

pytestmark = pytest.mark.gpu
U64MAX = np.iinfo(np.uint64).max


def _index(capi, db, algo=0, **opts):
    ix = capi.KnnIndex(db.shape[1])
    ix.set_option(capi.KNN_OPT_ALGO, algo)
    for k_, v in opts.items():
        ix.set_option(k_, v)
    ix.add(db)
    return ix


@pytest.mark.parametrize("algo", [1, 2, 3, 0])   # exact, split-bf16 coarse, fp32-MFMA coarse, auto
@pytest.mark.parametrize("case", KNN_CASES)
def test_matches_reference_goldens(capi, case, algo):
    db, q, k, g_idx, g_bits = load_knn_case(case)
    ix = _index(capi, db, algo)
    idx, d2 = ix.search(q, k)
    assert (idx == g_idx).all()
    assert (bits(d2) == g_bits).all()
    st = ix.stats()
    assert st["queries_total"] == q.shape[0]
    ix.close()


@pytest.mark.parametrize("algo", [1, 2, 3])
@pytest.mark.parametrize("N,D,Q,k", [(1, 8, 1, 1), (63, 4, 2, 5), (100, 7, 3, 20), (257, 510, 9, 20),
                                     (1000, 96, 17, 20), (2049, 33, 33, 1), (3000, 256, 70, 50),
                                     (5000, 1024, 130, 20)])
def test_ragged_shapes_vs_oracle(capi, oracle_mod, algo, N, D, Q, k):
    from gloc3d_amd import synth
    db = synth.descriptors_iid(100 + N, 0, N, D)
    q = synth.descriptors_iid(200 + N, 0, Q, D)
    ix = _index(capi, db, algo)
    idx, d2 = ix.search(q, k)
    oi, od = oracle_mod.knn_search(db, q, k, threads=4)
    assert (idx == oi).all()
    assert (bits(d2) == bits(od)).all()
    ix.close()


@pytest.mark.parametrize("algo", [1, 2, 3])
def test_window_empty_and_short(capi, oracle_mod, algo):
    from gloc3d_amd import synth
    db = synth.descriptors_traj(7, 0, 400, 128)
    q = synth.queries_near(7, [3, 77, 390], 128)
    ix = _index(capi, db, algo)
    for first, last in [(0, 370), (100, 105), (50, 50), (399, 10 ** 9), (0, None)]:  # SLAM window etc.
        idx, d2 = ix.search(q, 20, first, last)
        oi, od = oracle_mod.knn_search(db, q, 20, first, 400 if last is None else min(last, 400))
        assert (idx == oi).all() and (bits(d2) == bits(od)).all(), (first, last)
    idx, d2 = ix.search(q, 20, 100, 105)
    assert (idx[:, 5:] == U64MAX).all() and (d2[:, 5:] == np.finfo(np.float32).max).all()
    ix.close()


@pytest.mark.parametrize("algo", [1, 2, 3])
def test_duplicates_in_index_order_and_incremental_add(capi, oracle_mod, algo):
    from gloc3d_amd import synth
    db = synth.descriptors_iid(8, 0, 900, 64)
    db[500] = db[20]
    db[700] = db[20]
    ix = capi.KnnIndex(64)
    ix.set_option(capi.KNN_OPT_ALGO, algo)
    for a in range(0, 900, 123):  # add_keyframe-style growth; rows searchable immediately
        ix.add(db[a:a + 123])
        assert len(ix) == min(900, a + 123)
    q = np.concatenate([db[20:21], synth.descriptors_iid(9, 0, 12, 64)])
    idx, d2 = ix.search(q, 10)
    oi, od = oracle_mod.knn_search(db, q, 10)
    assert list(idx[0, :3]) == [20, 500, 700]
    assert (idx == oi).all() and (bits(d2) == bits(od)).all()
    ix.clear()
    assert len(ix) == 0
    ix.close()


def test_forced_fallback_still_exact(capi, oracle_mod):
    # 20 candidates for k = 20 can never be proven complete -> every query takes the exact fallback
    from gloc3d_amd import synth
    db = synth.descriptors_iid(31, 0, 3000, 256)
    q = synth.descriptors_iid(32, 0, 24, 256)
    ix = _index(capi, db, 2)
    ix.set_option(capi.KNN_OPT_CANDIDATES, 1)
    idx, d2 = ix.search(q, 52)  # k' = min(64, k + 12) = 64 here
    oi, od = oracle_mod.knn_search(db, q, 52)
    assert (idx == oi).all() and (bits(d2) == bits(od)).all()
    ix.close()
    # rows that differ by less than the coarse form can resolve: the candidate set cannot be proven complete, the
    # queries are redone on the exact path -- on the device, without a flag read-back (<= 16384 rows)
    base = synth.descriptors_iid(35, 0, 1, 256)
    db = (base + np.float32(2e-4) * synth.descriptors_iid(36, 0, 3000, 256)).astype(np.float32)
    q = (base + np.float32(2e-4) * synth.descriptors_iid(37, 0, 24, 256)).astype(np.float32)
    ix = _index(capi, db, 2)
    idx, d2 = ix.search(q, 20)
    oi, od = oracle_mod.knn_search(db, q, 20)
    assert (bits(d2) == bits(od)).all() and (idx == oi).all()
    assert ix.stats()["queries_fallback"] > 0
    ix.close()
    # the same above the device-side limit (flags read back, exact path per flagged query)
    db = synth.descriptors_iid(33, 0, 17000, 64)
    q = synth.descriptors_iid(34, 0, 12, 64)
    ix = _index(capi, db, 2)
    idx, d2 = ix.search(q, 52)
    oi, od = oracle_mod.knn_search(db, q, 52)
    assert (idx == oi).all() and (bits(d2) == bits(od)).all()
    ix.close()


def test_invalid_arguments(capi):
    ix = capi.KnnIndex(16)
    with pytest.raises(capi.GlocError):
        ix.search(np.zeros((1, 16), np.float32), 0)
    with pytest.raises(capi.GlocError):
        ix.search(np.zeros((1, 16), np.float32), 1000)
    with pytest.raises(capi.GlocError):
        ix.set_option(99, 1)
    idx, d2 = ix.search(np.zeros((2, 16), np.float32), 3)  # empty database: all sentinels
    assert (idx == U64MAX).all()
    ix.close()


def test_device_api_offset_and_merge(capi, oracle_mod):
    """Row-sharded search on one GPU: two shards with global offsets, merged by K3."""
    import torch
    from gloc3d_amd import synth
    N, D, Q, k = 3000, 128, 19, 20
    db = synth.descriptors_iid(41, 0, N, D)
    q = synth.descriptors_iid(42, 0, Q, D)
    dq = torch.from_numpy(q).cuda()
    outs_i, outs_d = [], []
    for lo, hi in [(0, 1700), (1700, N)]:
        ix = _index(capi, db[lo:hi], 0)
        di = torch.empty((Q, k), dtype=torch.int64, device="cuda")
        dd = torch.empty((Q, k), dtype=torch.float32, device="cuda")
        ix.search_device(dq.data_ptr(), Q, k, di.data_ptr(), dd.data_ptr(), index_offset=lo)
        ix.synchronize()
        outs_i.append(di)
        outs_d.append(dd)
        ix.close()
    gi = torch.stack(outs_i).contiguous()
    gd = torch.stack(outs_d).contiguous()
    mi = torch.empty((Q, k), dtype=torch.int64, device="cuda")
    md = torch.empty((Q, k), dtype=torch.float32, device="cuda")
    capi.topk_merge_device(0, torch.cuda.current_stream().cuda_stream, gi.data_ptr(), gd.data_ptr(), 2,
                           Q, k, mi.data_ptr(), md.data_ptr())
    torch.cuda.synchronize()
    oi, od = oracle_mod.knn_search(db, q, k)
    assert (mi.cpu().numpy().astype(np.uint64) == oi).all()
    assert (bits(md.cpu().numpy()) == bits(od)).all()


def test_merge_eight_lists(capi, oracle_mod):
    """K3 at the node's width: eight row shards with global offsets (ragged sizes, one smaller than k),
    merged on the device = the single-database top-k, bit for bit."""
    import torch
    from gloc3d_amd import synth
    N, D, Q, k = 4000, 96, 33, 20
    db = synth.descriptors_iid(43, 0, N, D)
    q = synth.descriptors_iid(44, 0, Q, D)
    db[1234] = db[77]                       # equal rows in different shards: ties ordered by global row
    dq = torch.from_numpy(q).cuda()
    cuts = [0, 700, 712, 1500, 1501, 2300, 3000, 3900, N]
    outs_i, outs_d = [], []
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        ix = _index(capi, db[lo:hi], 0)
        di = torch.empty((Q, k), dtype=torch.int64, device="cuda")
        dd = torch.empty((Q, k), dtype=torch.float32, device="cuda")
        ix.search_device(dq.data_ptr(), Q, k, di.data_ptr(), dd.data_ptr(), index_offset=lo)
        ix.synchronize()
        outs_i.append(di)
        outs_d.append(dd)
        ix.close()
    gi = torch.stack(outs_i).contiguous()
    gd = torch.stack(outs_d).contiguous()
    mi = torch.empty((Q, k), dtype=torch.int64, device="cuda")
    md = torch.empty((Q, k), dtype=torch.float32, device="cuda")
    capi.topk_merge_device(0, torch.cuda.current_stream().cuda_stream, gi.data_ptr(), gd.data_ptr(), 8,
                           Q, k, mi.data_ptr(), md.data_ptr())
    torch.cuda.synchronize()
    oi, od = oracle_mod.knn_search(db, q, k)
    assert (mi.cpu().numpy().astype(np.uint64) == oi).all()
    assert (bits(md.cpu().numpy()) == bits(od)).all()


def test_sharded_search_through_the_c_abi_world_1(capi, oracle_mod):
    """gloc_comm_* + gloc_knn_search_sharded on the one GPU of this box (an RCCL communicator of one rank):
    the collective path runs end to end and equals the plain search; interleaved global indices."""
    import torch
    from gloc3d_amd import sharded, synth
    N, D, Q, k = 3000, 128, 19, 20
    db = synth.descriptors_iid(41, 0, N, D)
    q = synth.descriptors_iid(42, 0, Q, D)
    comm = capi.Comm(0, 0, 1, lambda data: data)
    ix = _index(capi, db, 0)
    knn = sharded.CapiShardedKnn(ix, comm)
    idx, d2 = knn.search(torch.from_numpy(q).cuda(), k)
    torch.cuda.synchronize()
    oi, od = oracle_mod.knn_search(db, q, k)
    assert (idx.cpu().numpy().astype(np.uint64) == oi).all() and (bits(d2.cpu().numpy()) == bits(od)).all()
    # stride / offset: what rank 3 of 8 would report for its local rows
    di = torch.empty((Q, k), dtype=torch.int64, device="cuda")
    dd = torch.empty((Q, k), dtype=torch.float32, device="cuda")
    ix.search_sharded(comm, torch.from_numpy(q).cuda().data_ptr(), Q, k, di.data_ptr(), dd.data_ptr(), 8, 3)
    ix.synchronize()
    assert (di.cpu().numpy().astype(np.uint64) == oi * 8 + 3).all()
    t = torch.arange(2 * 5 * 19, dtype=torch.float32, device="cuda").view(2, 5, 19)
    g = knn.all_gather_tables(t)
    torch.cuda.synchronize()
    assert (g == t).all()
    ix.close()
    comm.close()


def test_on_device_generator_matches_numpy(capi):
    import torch
    from gloc3d_amd import synth
    for kind, gen in [(0, synth.descriptors_iid), (1, synth.descriptors_traj)]:
        out = torch.empty((37, 96), dtype=torch.float32, device="cuda")
        capi.synth_fill_device(0, torch.cuda.current_stream().cuda_stream, kind, 5001, 1000, 37, 96,
                               out.data_ptr())
        torch.cuda.synchronize()
        assert (bits(out.cpu().numpy()) == bits(gen(5001, 1000, 37, 96))).all()


def test_full_size_properties(capi):
    """cfg E shard size (125k x 4096 on-device rows): size-independent properties."""
    N, D, Q, k = 125_000, 4096, 64, 20
    ix = capi.KnnIndex(D)
    ix.add_synthetic(1, 5001, 0, N)
    from gloc3d_amd import synth
    rows = (np.arange(Q, dtype=np.uint64) * 1931 + 17) % N
    q = np.concatenate([synth.descriptors_traj(5001, int(r), 1, D) for r in rows])  # exact db rows
    idx, d2 = ix.search(q, k)
    assert (idx[:, 0] == rows).all() and (d2[:, 0] == 0).all()       # self match, distance 0
    assert (np.diff(d2.astype(np.float64), axis=1) >= 0).all()         # ascending
    assert all(len(set(r)) == k for r in idx.tolist())                 # distinct rows
    ix.set_option(capi.KNN_OPT_ALGO, 1)                                # exact path agrees bit for bit
    idx2, d22 = ix.search(q[:4], k)
    assert (idx2 == idx[:4]).all() and (bits(d22) == bits(d2[:4])).all()
    # sharding invariance: top-k of the union == merge of the halves' top-k (checked on the host)
    a = ix.search(q[:4], k, 0, N // 2)
    b = ix.search(q[:4], k, N // 2, N)
    for r in range(4):
        cat = sorted(zip(np.concatenate([a[1][r], b[1][r]]).tolist(), np.concatenate([a[0][r], b[0][r]]).tolist()))
        assert [c[1] for c in cat[:k]] == idx[r].tolist()
    ix.close()


def test_many_queries_and_empty_add(capi, oracle_mod):
    """More queries than one workspace block (1024) and add() of zero rows."""
    from gloc3d_amd import synth
    db = synth.descriptors_iid(61, 0, 700, 32)
    q = synth.descriptors_iid(62, 0, 1500, 32)
    ix = capi.KnnIndex(32)
    ix.add(db[:0])
    ix.add(db)
    for algo in (1, 2):
        ix.set_option(capi.KNN_OPT_ALGO, algo)
        idx, d2 = ix.search(q, 9)
        oi, od = oracle_mod.knn_search(db, q, 9, threads=4)
        assert (idx == oi).all() and (bits(d2) == bits(od)).all()
    ix.close()


def test_save_load_round_trip(capi, oracle_mod, tmp_path):
    from gloc3d_amd import gloc_io, synth
    db = synth.descriptors_traj(71, 0, 900, 48)
    q = synth.queries_near(71, [5, 444, 899], 48)
    a = capi.KnnIndex(48)
    a.add(db)
    a.save(tmp_path / "db.desc")
    a.close()
    assert (gloc_io.read_descriptors(tmp_path / "db.desc") == db).all()  # the CLI's descriptor format
    b = capi.KnnIndex(48)
    b.load(tmp_path / "db.desc")
    assert len(b) == 900
    idx, d2 = b.search(q, 20)
    oi, od = oracle_mod.knn_search(db, q, 20)
    assert (idx == oi).all() and (bits(d2) == bits(od)).all()
    c = capi.KnnIndex(32)
    with pytest.raises(capi.GlocError):
        c.load(tmp_path / "db.desc")       # dimension mismatch
    with pytest.raises(capi.GlocError):
        c.load(tmp_path / "missing.desc")
    # a truncated file is refused before a single row is added: the index keeps its rows
    raw = (tmp_path / "db.desc").read_bytes()
    (tmp_path / "cut.desc").write_bytes(raw[: len(raw) // 2])
    with pytest.raises(capi.GlocError):
        b.load(tmp_path / "cut.desc")
    assert len(b) == 900
    idx, d2 = b.search(q, 20)
    assert (idx == oi).all()
    b.load(tmp_path / "db.desc")           # load appends
    assert len(b) == 1800
    b.close()
    c.close()


@pytest.mark.parametrize("algo", [1, 2, 3])
@pytest.mark.parametrize("N,D,Q,k", [(16384, 64, 12, 20), (16385, 64, 12, 20), (12000, 100, 20, 52), (40, 4096, 10, 20)])
def test_one_work_group_per_query_select_limits(capi, oracle_mod, algo, N, D, Q, k):
    """The one-launch selection (<= 16384 rows, <= 64 keys) at its limits, one row past them (chunked form),
    dim % 256 != 0 (tail groups of the re-rank), k = 52 (64 candidates: selection only), fewer rows than candidates."""
    from gloc3d_amd import synth
    db = synth.descriptors_iid(300 + N, 0, N, D)
    q = synth.descriptors_iid(400 + N, 0, Q, D)
    ix = _index(capi, db, algo)
    idx, d2 = ix.search(q, k)
    oi, od = oracle_mod.knn_search(db, q, k, threads=4)
    assert (idx == oi).all() and (bits(d2) == bits(od)).all()
    ix.close()


@pytest.mark.parametrize("algo", [1, 2, 3])
def test_select_with_near_rows_on_few_threads(capi, oracle_mod, algo):
    """Rows j with j % 1024 < 4 lie near the query: four of the selecting work-group's 1024 threads hold all the
    small keys, so more than 64 keys pass the threshold of thread minima (the list is sorted through LDS then);
    with a row window that does not start at 0, and duplicated rows among the near ones (ties by row index)."""
    from gloc3d_amd import synth
    N, D, Q = 9000, 128, 11
    db = synth.descriptors_iid(51, 0, N, D)
    q = synth.descriptors_iid(52, 0, Q, D)
    first = 37
    near = np.array([j for j in range(first, N) if (j - first) % 1024 < 4])
    noise = synth.descriptors_iid(53, 0, len(near), D)
    db[near] = (q[0][None, :] + np.float32(0.02) * noise).astype(np.float32)
    db[near[5]] = db[near[1]]
    db[near[9]] = db[near[1]]
    ix = _index(capi, db, algo)
    for k in (20, 33):
        idx, d2 = ix.search(q, k, first, N)
        oi, od = oracle_mod.knn_search(db, q, k, first, N)
        assert (idx == oi).all() and (bits(d2) == bits(od)).all()
    ix.close()


@pytest.mark.parametrize("algo", [1, 2, 3])
@pytest.mark.parametrize("N,first,Q", [(40001, 13, 9), (70000, 0, 1), (33000, 5000, 40)])
def test_sliced_select_above_16384_rows(capi, oracle_mod, algo, N, first, Q):
    """Windows above 16 384 rows (slices of <= 16 384 rows, then the selection over the slices' lists): a window that
    does not start at 0, a ragged last slice, the near rows of query 0 crowded into ONE slice (all its k results come
    from one list) and duplicated rows on both sides of a slice boundary (ties by row index across lists)."""
    from gloc3d_amd import synth
    D = 64
    db = synth.descriptors_iid(61, 0, N, D)
    q = synth.descriptors_iid(62, 0, Q, D)
    near = first + 100 + np.arange(40)                      # 40 near rows, all in the first slice
    db[near] = (q[0][None, :] + np.float32(0.02) * synth.descriptors_iid(63, 0, 40, D)).astype(np.float32)
    for j in (first + 4000, first + 9000, first + 17000, N - 2):         # the same row in several slices
        db[j] = db[near[3]]
    ix = _index(capi, db, algo)
    for k in (1, 20, 52):
        idx, d2 = ix.search(q, k, first, N)
        oi, od = oracle_mod.knn_search(db, q, k, first, N, threads=4)
        assert (idx == oi).all() and (bits(d2) == bits(od)).all()
    ix.close()


@pytest.mark.parametrize("algo", [2, 3])
def test_unproven_queries_above_16384_rows_are_redone_on_the_device(capi, oracle_mod, algo):
    """Rows closer to each other than the coarse form can resolve, in a window above 16 384 rows: the fused selection +
    re-rank over the slices' lists flags the queries, and the exact pass (distances, slices, list selection -- launched
    for flagged queries only) replaces their results, with no read-back of the flags."""
    from gloc3d_amd import synth
    N, D, Q = 20000, 256, 9
    base = synth.descriptors_iid(71, 0, 1, D)
    db = (base + np.float32(2e-4) * synth.descriptors_iid(72, 0, N, D)).astype(np.float32)
    q = (base + np.float32(2e-4) * synth.descriptors_iid(73, 0, Q, D)).astype(np.float32)
    ix = _index(capi, db, algo)
    idx, d2 = ix.search(q, 20)
    oi, od = oracle_mod.knn_search(db, q, 20, threads=4)
    assert (idx == oi).all() and (bits(d2) == bits(od)).all()
    assert ix.stats()["queries_fallback"] > 0
    ix.close()


@pytest.mark.parametrize("D", [256, 4096])
def test_split_bf16_bound_across_the_resolution_of_the_coarse_form(capi, oracle_mod, D):
    """The split-bf16 coarse pass drops product terms of relative size 2^-16: a database whose rows sit at a graded
    distance from the queries (spread 1 ... 3e-4 of the norm) walks the gaps between neighbours from far above to
    far below that resolution.  Whatever the coarse order is, the result has to be the reference's bits -- by the
    proof where the gaps are wide, by the exact pass where they are not (a bound set too tight would show here as a
    neighbour missing from a result)."""
    from gloc3d_amd import synth
    N, Q = 3000, 16
    base = synth.descriptors_iid(81, 0, 1, D)
    fallbacks = []
    for i, spread in enumerate((1.0, 0.3, 0.1, 3e-2, 1e-2, 3e-3, 3e-4)):
        db = (base + np.float32(spread) * synth.descriptors_iid(82 + i, 0, N, D)).astype(np.float32)
        q = (base + np.float32(spread) * synth.descriptors_iid(92 + i, 0, Q, D)).astype(np.float32)
        ix = _index(capi, db, 2)
        idx, d2 = ix.search(q, 20)
        oi, od = oracle_mod.knn_search(db, q, 20, threads=4)
        assert (idx == oi).all() and (bits(d2) == bits(od)).all(), spread
        fallbacks.append(ix.stats()["queries_fallback"])
        ix.close()
    assert fallbacks[0] == 0 and fallbacks[-1] == Q, fallbacks   # proven at the wide end, redone at the narrow end


def test_handle_leaves_the_split_form_when_its_proofs_keep_failing(capi, oracle_mod):
    """Rows clustered tightly relative to their norms: the split-bf16 coarse pass (bound 8 x the fp32 form's) cannot prove
    its candidate sets and every query goes to the exact pass, while the fp32 coarse pass proves them all.  The handle
    notices (the fallback counter of a search is looked at by a later one, without a wait) and takes the fp32 form for
    its next searches: the same bits throughout, and the fallbacks stop."""
    from gloc3d_amd import synth
    N, D, Q = 5000, 2048, 64
    cen = synth.descriptors_iid(111, 0, 7, D)
    rng = np.random.default_rng(5)
    db = (cen[rng.integers(0, 7, N)] + np.float32(0.05) * synth.descriptors_iid(112, 0, N, D)).astype(np.float32)
    q = (db[rng.integers(0, N, Q)] + np.float32(0.02) * synth.descriptors_iid(113, 0, Q, D)).astype(np.float32)
    oi, od = oracle_mod.knn_search(db, q, 5, threads=4)
    ix = _index(capi, db, 2)
    seen = []
    for _ in range(6):
        idx, d2 = ix.search(q, 5)                       # (the host-buffer entry point waits for its result: the copy has landed)
        assert (idx == oi).all() and (bits(d2) == bits(od)).all()
        seen.append(ix.stats()["queries_fallback"])
    ix.close()
    assert seen[0] > Q // 4, seen                       # the split form fell back ...
    assert seen[-1] == seen[2], seen                    # ... and after the handle looked, nothing falls back any more
    ix = _index(capi, db, 3)                            # the fp32 form proves these queries
    ix.search(q, 5)
    assert ix.stats()["queries_fallback"] == 0
    ix.close()


def test_a_view_searches_its_parents_rows_beside_it(capi):
    """gloc_knn_create_view (round 6): a second search handle over the same resident rows, own stream and workspace.  Its
    results are the parent's bit for bit -- alone, interleaved with the parent's searches (the two run side by side on the
    device), after the parent has grown, on a row window -- and it cannot change the database."""
    import torch
    from gloc3d_amd import synth
    dim, n = 512, 6000
    db = synth.descriptors_traj(77, 0, n, dim)
    ix = capi.KnnIndex(dim)
    ix.add(db[:4000])
    v = ix.view()
    assert len(v) == 4000
    qa = synth.queries_near(77, (np.arange(64) * 53 + 3) % 4000, dim)
    qb = synth.queries_near(77, (np.arange(48) * 31 + 11) % 4000, dim)
    wa, wb = ix.search(qa, 20), ix.search(qb, 20)
    ga, gb = v.search(qa, 20), v.search(qb, 20)
    for (gi, gd), (wi, wd) in ((ga, wa), (gb, wb)):
        assert (gi == wi).all() and (bits(gd) == bits(wd)).all()
    # interleaved on the device: parent and view each on its own stream, 20 searches in flight
    dqa, dqb = torch.from_numpy(qa).cuda(), torch.from_numpy(qb).cuda()
    oa = [(torch.empty((64, 20), dtype=torch.int64, device="cuda"), torch.empty((64, 20), dtype=torch.float32, device="cuda")) for _ in range(10)]
    ob = [(torch.empty((48, 20), dtype=torch.int64, device="cuda"), torch.empty((48, 20), dtype=torch.float32, device="cuda")) for _ in range(10)]
    torch.cuda.synchronize()
    for i in range(10):
        ix.search_device(dqa.data_ptr(), 64, 20, oa[i][0].data_ptr(), oa[i][1].data_ptr())
        v.search_device(dqb.data_ptr(), 48, 20, ob[i][0].data_ptr(), ob[i][1].data_ptr())
    ix.synchronize()
    v.synchronize()
    for i in range(10):
        assert (oa[i][0].cpu().numpy().astype(np.uint64) == wa[0]).all() and (bits(oa[i][1].cpu().numpy()) == bits(wa[1])).all()
        assert (ob[i][0].cpu().numpy().astype(np.uint64) == wb[0]).all() and (bits(ob[i][1].cpu().numpy()) == bits(wb[1])).all()
    # the parent grows (its buffers may move): the view's next search sees all rows
    ix.add(db[4000:])
    ix.synchronize()
    qc = synth.queries_near(77, np.array([4100, 5999, 17, 5000]), dim)
    wi, wd = ix.search(qc, 20)
    gi, gd = v.search(qc, 20)
    assert len(v) == n and (gi == wi).all() and (bits(gd) == bits(wd)).all() and gi[1, 0] == 5999
    gi, gd = v.search(qc, 20, first_row=100, last_row=4500)            # the SLAM window through a view
    wi, wd = ix.search(qc, 20, first_row=100, last_row=4500)
    assert (gi == wi).all() and (bits(gd) == bits(wd)).all()
    for call in (lambda: v.add(db[:10]), lambda: v.reserve(10000), lambda: v.clear()):
        with pytest.raises(capi.GlocError) as ei:
            call()
        assert ei.value.code == 5                                     # GLOC_ERR_STATE: a view cannot change the database
    with pytest.raises(capi.GlocError):
        ix.close()                                                    # not while a view lives
    v.close()
    ix.close()


@pytest.mark.parametrize("D", [8, 16, 24, 40])
def test_coarse_mirror_with_a_dim_below_one_step(capi, oracle_mod, D):
    """Round 6: the split-bf16 coarse pass streams a tiled mirror in steps of 32 k.  A dim below one step (8, 16, 24) made its
    prefetch index planes BELOW the mirror (a GPU memory fault in the long fuzz run); 40 = one step and a quarter.  Windows
    that start inside a tile, a database that ends inside one."""
    from gloc3d_amd import synth
    N, Q = 1000 + D, 33
    db = synth.descriptors_iid(300 + D, 0, N, D)
    q = synth.descriptors_iid(400 + D, 0, Q, D)
    ix = _index(capi, db, 2)
    for first, last in ((0, N), (37, N - 5), (64, 65 + D)):
        idx, d2 = ix.search(q, 5, first_row=first, last_row=last)
        oi, od = oracle_mod.knn_search(db[first:last], q, 5)
        oi = np.where(oi == np.iinfo(np.uint64).max, oi, oi + np.uint64(first))
        assert (idx == oi).all() and (bits(d2) == bits(od)).all(), (first, last)
    ix.close()
