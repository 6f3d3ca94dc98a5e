"""GPU: BASELINE configs[4] at its FULL size on one GPU -- the synthetic 1 M-place x 4096-D descriptor database, searched
(a) as ONE index of 1 000 000 rows and (b) as the EIGHT interleaved shards of 125 000 rows a node of 8 GPUs would hold
(place g on shard g % 8 at local row g // 8), every shard searched by its own handle with global indices and the eight
top-k lists merged by K3 (gloc_topk_merge_device) -- everything of the sharded path except the RCCL transport of the
160-byte-per-query lists, which needs more than one GPU.  (a) == (b) bit for bit (indices and distance bits) for 64
queries and for one; and 8 of the queries against the CPU checker over the same 16.4 GB database (brute force,
evalMetric order).  288 GB of HBM hold both copies (33 GB) with room to spare."""
import numpy as np
import pytest

from util import bits

pytestmark = pytest.mark.gpu
N, D, G, K, SEED = 1_000_000, 4096, 8, 20, 5001


@pytest.fixture(scope="module")
def cfg_e(capi):
    import torch
    from gloc3d_amd import synth
    full = capi.KnnIndex(D)
    full.reserve(N)
    full.add_synthetic(1, SEED, 0, N)                       # the trajectory distribution of SURVEY 8d cfg E
    shards = []
    for r in range(G):
        ix = capi.KnnIndex(D)
        ix.reserve(N // G)
        ix.add_synthetic(1, SEED, r, N // G, row_stride=G)  # global rows r, r + 8, ...: rank r's shard
        shards.append(ix)
    rows = (np.arange(64, dtype=np.int64) * 15485 + 977) % N
    q = synth.queries_near(SEED, rows, D)
    yield dict(full=full, shards=shards, q=q, rows=rows, torch=torch)
    full.close()
    for ix in shards:
        ix.close()


def _sharded(capi, torch, shards, q):
    nq = q.shape[0]
    li = np.empty((G, nq, K), np.uint64)
    ld = np.empty((G, nq, K), np.float32)
    for r, ix in enumerate(shards):
        i, d = ix.search(q, K)
        li[r] = np.where(i == np.iinfo(np.uint64).max, i, i * np.uint64(G) + np.uint64(r))   # local row l -> global l * 8 + r
        ld[r] = d
    ti, td = torch.from_numpy(li.view(np.int64)).cuda(), torch.from_numpy(ld).cuda()
    oi = torch.empty((nq, K), dtype=torch.int64, device="cuda")
    od = torch.empty((nq, K), dtype=torch.float32, device="cuda")
    capi.topk_merge_device(0, torch.cuda.current_stream().cuda_stream, ti.data_ptr(), td.data_ptr(), G, nq, K,
                           oi.data_ptr(), od.data_ptr())
    torch.cuda.synchronize()
    return oi.cpu().numpy().astype(np.uint64), od.cpu().numpy()


def test_eight_shards_merged_equal_the_one_million_row_index(capi, cfg_e):
    c = cfg_e
    assert len(c["full"]) == N and all(len(s) == N // G for s in c["shards"])
    for nq in (64, 1):
        q = np.ascontiguousarray(c["q"][:nq])
        fi, fd = c["full"].search(q, K)
        si, sd = _sharded(capi, c["torch"], c["shards"], q)
        assert (fi == si).all() and (bits(fd) == bits(sd)).all(), nq
        assert (fi[:, 0] == c["rows"][:nq].astype(np.uint64)).all()      # a query sits next to its own row
        assert (np.diff(fd, axis=1) >= 0).all()
    st = c["full"].stats()
    assert st["queries_fallback"] <= 2, st                               # (the completeness proof holds at this size too)


def test_one_million_rows_against_the_cpu_checker(capi, oracle_mod, cfg_e):
    """8 queries, brute force over all 1 000 000 rows on the host (16.4 GB, generated on the device and copied)."""
    c, torch = cfg_e, cfg_e["torch"]
    db = np.empty((N, D), np.float32)
    chunk = 50_000
    buf = torch.empty((chunk, D), dtype=torch.float32, device="cuda")
    for a in range(0, N, chunk):
        capi.synth_fill_device(0, torch.cuda.current_stream().cuda_stream, 1, SEED, a, chunk, D, buf.data_ptr())
        torch.cuda.synchronize()
        db[a:a + chunk] = buf.cpu().numpy()
    q = np.ascontiguousarray(c["q"][:8])
    oi, od = oracle_mod.knn_search(db, q, K, threads=8)
    gi, gd = c["full"].search(q, K)
    assert (gi == oi).all() and (bits(gd) == bits(od)).all()
