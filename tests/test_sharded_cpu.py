"""CPU: the multi-GPU orchestration (gloc3d_amd/sharded.py) over `gloo`, world_size 2, 3 and 8.
The HIP calls are replaced by checker callables (the oracle) -- what is under test is the sharding
arithmetic, the fused all-gather, the replicated merge order and the candidate all-reduce."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


Q_ROWS = [5, 40, 202, 17, 63, 99, 0, 1, 150, 77, 201, 33, 120, 8, 190, 101]   # 16 queries: 2 per rank at world 8


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _checker_merge(all_i, all_d):
    G, Q, k = all_i.shape
    oi = torch.empty((Q, k), dtype=torch.int64)
    od = torch.empty((Q, k), dtype=torch.float32)
    for q in range(Q):
        d = all_d[:, q].reshape(-1).numpy()
        i = all_i[:, q].reshape(-1).numpy().astype(np.uint64)  # -1 -> UINT64_MAX sorts last
        order = np.lexsort((i, d))[:k]
        oi[q] = torch.from_numpy(i[order].astype(np.int64))
        od[q] = torch.from_numpy(d[order])
    return oi, od


def _worker(rank, world, port, n_places, dim, out_dir):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import oracle
    from gloc3d_amd import sharded, synth
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    rows = sharded.shard_rows(n_places, rank, world)
    shard = np.concatenate([synth.descriptors_traj(77, int(g), 1, dim) for g in rows])

    def local_search(q, k):
        idx, d2 = oracle.knn_search(shard, q.numpy(), k)
        return torch.from_numpy(idx.astype(np.int64)), torch.from_numpy(d2)

    knn = sharded.ShardedKnn(rank, world, local_search, _checker_merge)
    q_rows = np.array(Q_ROWS)
    q = torch.from_numpy(synth.queries_near(77, q_rows, dim))
    gi, gd = knn.search(q, 20)

    # candidate-sharded "registration": a deterministic stand-in keyed by (place, retrieval rank)
    calls = []

    def local_register(query, local_rows, ranks):
        calls.append((list(map(int, local_rows)), list(map(int, ranks))))
        out = np.zeros((len(local_rows), sharded.RESULT_COLS), np.float32)
        for r, (l, c) in enumerate(zip(local_rows, ranks)):
            g = int(l) * world + rank
            out[r, :16] = np.arange(16) + g
            out[r, 16] = 0.5 * c
            out[r, 17] = g
            out[r, 18] = 1.0 if (c >= 3 and g % 2 == 0) else 0.0
        return out

    sreg = sharded.ShardedRegistrar(rank, world, local_register)
    table = sreg.register(0, gi[0].numpy(), torch.device("cpu"))
    sel = sreg.select_first_ok(table)
    # the same exchange as an all-gather of the per-rank tables (the form below the C ABI,
    # gloc_comm_all_gather_device; here a gloo stand-in with the same [1, n, C] -> [world, n, C] contract)
    def gather(t):
        out = torch.empty((world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype)
        dist.all_gather_into_tensor(out, t.contiguous())
        return out
    sreg_g = sharded.ShardedRegistrar(rank, world, local_register, gather=gather)
    table_g = sreg_g.register(0, gi[0].numpy(), torch.device("cpu"))
    assert (table_g == table).all() and sreg_g.select_first_ok(table_g) == sel
    # throughput mode: query r registered entirely by rank r, tables all-gathered
    def register_all(query, places, ranks):
        out = np.zeros((len(places), sharded.RESULT_COLS), np.float32)
        out[:, 16] = query
        out[:, 17] = places
        out[:, 18] = (np.asarray(ranks) % 2).astype(np.float32)
        return out

    qreg = sharded.QueryParallelRegistrar(rank, world, register_all)
    tables = qreg.register(100 + rank, gi[:world].numpy(), torch.device("cpu"))
    # two queries in flight per rank, registered in ONE batch: row k is tagged with 1000 * k
    def register_multi(queries, places):
        out = np.stack([register_all(q, p, np.arange(len(p))) for q, p in zip(queries, places)])
        out[:, :, 0] = 1000 * np.arange(len(queries))[:, None]
        return out

    many = qreg.register_many([200 + 2 * rank, 201 + 2 * rank], gi[:2 * world].numpy(), torch.device("cpu"),
                              register_multi)
    np.savez(os.path.join(out_dir, f"r{rank}.npz"), gi=gi.numpy(), gd=gd.numpy(), table=table.numpy(),
             sel=sel, mine=np.array(calls[0][0] if calls else [], np.int64), tables=tables.numpy(),
             many=many.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 8])
def test_sharded_search_and_registration_over_gloo(oracle_mod, tmp_path, world):
    from gloc3d_amd import sharded, synth
    n_places, dim = 203, 64
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n_places, dim, str(tmp_path)), nprocs=world, join=True)
    outs = [np.load(tmp_path / f"r{r}.npz") for r in range(world)]
    # single-database truth
    db = synth.descriptors_traj(77, 0, n_places, dim)
    q = synth.queries_near(77, np.array(Q_ROWS), dim)
    oi, od = oracle_mod.knn_search(db, q, 20)
    for o in outs:                                   # replicated and equal to the 1-GPU result
        assert (o["gi"].astype(np.uint64) == oi).all()
        assert (o["gd"].view(np.uint32) == od.view(np.uint32)).all()
        assert (o["table"] == outs[0]["table"]).all() and o["sel"] == outs[0]["sel"]
    # every candidate registered exactly once, by its owner
    cand = oi[0].astype(np.int64)
    t = outs[0]["table"]
    assert (t[:, 17] == cand).all() and np.allclose(t[:, 16], 0.5 * np.arange(20))
    for r, o in enumerate(outs):
        assert sorted(o["mine"].tolist()) == sorted((cand[cand % world == r] // world).tolist())
    expect = [c for c in range(20) if c >= 3 and cand[c] % 2 == 0]
    assert outs[0]["sel"] == (expect[0] if expect else -1)
    # query-parallel mode: table r was produced by rank r for query r, replicated everywhere
    for o in outs:
        tb = o["tables"]
        assert tb.shape == (world, 20, sharded.RESULT_COLS) and (tb == outs[0]["tables"]).all()
        for r in range(world):
            assert (tb[r, :, 16] == 100 + r).all() and (tb[r, :, 17] == oi[r].astype(np.float32)).all()
        # two in flight: row r*2+k was produced by rank r's handle k for query 200 + 2r + k
        mn = o["many"]
        assert mn.shape == (2 * world, 20, sharded.RESULT_COLS) and (mn == outs[0]["many"]).all()
        for r in range(world):
            for k in range(2):
                row = mn[2 * r + k]
                assert (row[:, 16] == 200 + 2 * r + k).all() and (row[:, 0] == 1000 * k).all()
                assert (row[:, 17] == oi[2 * r + k].astype(np.float32)).all()


def test_sharding_arithmetic():
    from gloc3d_amd import sharded
    for world in (1, 2, 8):
        seen = np.concatenate([sharded.shard_rows(1000, r, world) for r in range(world)])
        assert sorted(seen.tolist()) == list(range(1000))
        for r in range(world):
            rows = sharded.shard_rows(1000, r, world)
            assert (sharded.owner_rank(rows, world) == r).all()
            assert (sharded.global_row(sharded.local_row(rows, world), r, world) == rows).all()
