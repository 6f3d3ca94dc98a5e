"""Multi-GPU hot path: one process per GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI).

Sharding:
  * descriptor database: place g lives on rank g % G at local row g // G (interleaved, so the
    consecutive places a query retrieves spread over all ranks); queries are replicated; each rank
    searches its shard, the per-shard top-k (d2, global index) lists are ALL-GATHERED (Q x k x 12 B
    per rank: latency-bound, one fused buffer) and every rank runs the same G*k -> k merge (K3), so
    the result is replicated and bit-equal to the single-GPU search.
  * registration: candidate c of a query is registered by the rank that owns its scan
    (g_c % G); results (pose, rmse, inliers, ok) are combined with one ALL-REDUCE of a small
    zero-initialised table; the reference's selection rule (lowest-rank successful candidate,
    registration/global_localization.cpp:519-572) is then evaluated identically on every rank.

The orchestration below is backend-agnostic: `local_search`, `merge` and `local_register` default to
the HIP C-ABI calls; the CPU `gloo` tests inject checker callables to exercise the collectives.
"""
import numpy as np
import torch
import torch.distributed as dist

I64_SENTINEL = -1  # UINT64_MAX reinterpreted: "no row"
RESULT_COLS = 19   # 16 pose + rmse + inliers + ok


def owner_rank(global_idx, world):
    return global_idx % world


def local_row(global_idx, world):
    return global_idx // world


def global_row(local_idx, rank, world):
    return local_idx * world + rank


def shard_rows(n_global, rank, world):
    """Global ids of the places this rank owns, ascending."""
    return np.arange(rank, n_global, world, dtype=np.int64)


class ShardedKnn:
    """Row-sharded exact top-k with an all-gather of per-shard lists."""

    def __init__(self, rank, world, local_search, merge, group=None, comm_device=None):
        self.rank, self.world, self.group = rank, world, group
        self.comm_device = comm_device    # stage collectives through this device (gloo rehearsal)
        self.local_search = local_search  # (q [Q,D] tensor, k) -> (local idx int64 [Q,k], d2 f32 [Q,k])
        self.merge = merge                # (idx [G,Q,k], d2 [G,Q,k]) -> (idx [Q,k], d2 [Q,k])

    def search(self, q, k):
        li, ld = self.local_search(q, k)
        gi = torch.where(li == I64_SENTINEL, li, li * self.world + self.rank)
        if self.world == 1:
            return gi, ld
        # one fused buffer per rank: [Q, k, 3] int32 words = (idx lo, idx hi, d2 bits)
        Q = gi.shape[0]
        packed = torch.empty((Q, k, 3), dtype=torch.int32, device=gi.device)
        packed[..., :2] = gi.contiguous().view(torch.int32).view(Q, k, 2)
        packed[..., 2] = ld.contiguous().view(torch.int32)
        # output laid out as the concatenation along dim 0 (the form every backend accepts)
        cdev = self.comm_device or gi.device
        packed = packed.to(cdev)
        gathered = torch.empty((self.world * Q, k, 3), dtype=torch.int32, device=cdev)
        dist.all_gather_into_tensor(gathered, packed, group=self.group)
        gathered = gathered.to(gi.device).view(self.world, Q, k, 3)
        all_i = gathered[..., :2].contiguous().view(torch.int64).view(self.world, Q, k)
        all_d = gathered[..., 2].contiguous().view(torch.float32)
        return self.merge(all_i, all_d)


class CapiShardedKnn:
    """The same search with the collective BELOW the C ABI (gloc_knn_search_sharded): local search ->
    one fused RCCL all-gather of the per-shard lists -> merge on the device, all on one stream, no host
    hop.  `comm` is a gloc3d_amd.capi.Comm; the index holds this rank's interleaved shard."""

    def __init__(self, index, comm):
        self.index, self.comm = index, comm
        self.side = torch.cuda.Stream()
        index.set_stream(self.side.cuda_stream)

    def search(self, q, k):
        Q = q.shape[0]
        idx = torch.empty((Q, k), dtype=torch.int64, device=q.device)
        d2 = torch.empty((Q, k), dtype=torch.float32, device=q.device)
        cur = torch.cuda.current_stream()
        self.side.wait_stream(cur)
        self.index.search_sharded(self.comm, q.data_ptr(), Q, k, idx.data_ptr(), d2.data_ptr(),
                                  index_stride=self.comm.world, index_offset=self.comm.rank)
        cur.wait_stream(self.side)
        return idx, d2

    def all_gather_tables(self, tables):
        """tables: this rank's [K, n, RESULT_COLS] float32 device tensor -> [world * K, n, RESULT_COLS]."""
        t = tables.contiguous()
        out = torch.empty((self.comm.world,) + tuple(t.shape), dtype=t.dtype, device=t.device)
        # ONE stream carries every collective of the communicator (the side stream the sharded search runs on):
        # RCCL orders a communicator's operations by issue order only if they share a stream
        cur = torch.cuda.current_stream()
        self.side.wait_stream(cur)
        self.comm.all_gather_device(t.data_ptr(), out.data_ptr(), t.numel() * t.element_size(), self.side.cuda_stream)
        cur.wait_stream(self.side)
        t.record_stream(self.side)
        out.record_stream(self.side)
        return out.view((self.comm.world * t.shape[0],) + tuple(t.shape[1:]))


def torch_exchange(device):
    """exchange() for capi.Comm over an initialised torch.distributed group: broadcast of rank 0's id."""
    def fn(data):
        t = torch.zeros(128, dtype=torch.uint8, device=device)
        if data is not None:
            t.copy_(torch.frombuffer(bytearray(data), dtype=torch.uint8))
        dist.broadcast(t, src=0)
        return bytes(t.cpu().numpy().tobytes())
    return fn


class ShardedRegistrar:
    """Candidate-sharded registration: each rank registers the candidates whose scans it owns."""

    def __init__(self, rank, world, local_register, group=None, comm_device=None, gather=None):
        self.rank, self.world, self.group = rank, world, group
        self.comm_device = comm_device
        # (query handle, local scan ids [m], retrieval ranks [m]) -> float32 [m, RESULT_COLS]
        self.local_register = local_register
        # [1, n, RESULT_COLS] device tensor -> [world, n, RESULT_COLS]: the exchange below the C ABI
        # (CapiShardedKnn.all_gather_tables = gloc_comm_all_gather_device); None: torch.distributed
        self.gather = gather

    def register(self, query, cand_global, device):
        cand_global = np.asarray(cand_global, dtype=np.int64)
        n = cand_global.shape[0]
        table = torch.zeros((n, RESULT_COLS), dtype=torch.float32, device=device)
        mine = np.nonzero((cand_global >= 0) & (cand_global % self.world == self.rank))[0]
        if mine.size:
            res = self.local_register(query, cand_global[mine] // self.world, mine.astype(np.uint32))
            table[torch.as_tensor(mine, device=device)] = torch.as_tensor(res, device=device)
        if self.world > 1 and self.gather is not None:
            # every rank's table holds its own rows and zeros elsewhere: the sum over ranks is exact
            table = self.gather(table[None]).sum(dim=0)
        elif self.world > 1:
            t = table.to(self.comm_device) if self.comm_device else table
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)  # disjoint rows: exact
            table = t.to(device)
        return table

    @staticmethod
    def select_first_ok(table):
        """Lowest retrieval rank whose registration succeeded, or -1 (global_localization.cpp:519).
        table: [n, RESULT_COLS], numpy or torch."""
        if isinstance(table, np.ndarray):
            ok = np.flatnonzero(table[:, 18] > 0.5)
            return int(ok[0]) if ok.size else -1
        ok = (table[:, 18] > 0.5).nonzero()
        return int(ok[0, 0]) if ok.numel() else -1


class QueryParallelRegistrar:
    """Throughput mode: a step handles G queries, one per rank.  Retrieval is still sharded (the
    database rows are interleaved over the ranks, per-shard top-k all-gathered and merged); query r's
    20 candidates are then registered entirely by rank r against its replica of the scan store, so
    every registration launch keeps its full batch of candidates, and the G result tables are
    all-gathered.  Per-GPU work is fixed as G grows (weak scaling)."""

    def __init__(self, rank, world, local_register_all, group=None, comm_device=None):
        self.rank, self.world, self.group, self.comm_device = rank, world, group, comm_device
        # (query handle, global place ids [n], retrieval ranks [n]) -> float32 [n, RESULT_COLS]
        self.local_register_all = local_register_all

    def register(self, my_query, cand_global_all, device):
        """cand_global_all: [G, n] global place ids (replicated).  Returns [G, n, RESULT_COLS]."""
        cand = np.asarray(cand_global_all, dtype=np.int64)
        n = cand.shape[1]
        mine = cand[self.rank]
        table = torch.zeros((n, RESULT_COLS), dtype=torch.float32, device=device)
        ok_rows = np.nonzero(mine >= 0)[0]
        if ok_rows.size:
            res = self.local_register_all(my_query, mine[ok_rows], ok_rows.astype(np.uint32))
            table[torch.as_tensor(ok_rows, device=device)] = torch.as_tensor(res, device=device)
        if self.world == 1:
            return table[None]
        t = table.to(self.comm_device) if self.comm_device else table
        out = torch.empty((self.world * n, RESULT_COLS), dtype=torch.float32, device=t.device)
        dist.all_gather_into_tensor(out, t, group=self.group)
        return out.to(device).view(self.world, n, RESULT_COLS)

    def register_many(self, my_queries, cand_global_all, device, register_multi, capi_knn=None):
        """K queries per rank in flight (K = len(my_queries)): a step handles G*K queries, rank r owning
        rows r*K .. r*K+K-1 of cand_global_all [G*K, n].  register_multi(query handles [K], global place
        ids [K, n] (-1 = none)) -> float32 [K, n, RESULT_COLS] registers them in ONE batch (every kernel
        launch covers the K x n candidates: gloc_reg_batch_multi).  One all-gather of the K result
        tables.  Returns [G*K, n, RESULT_COLS] (a numpy array at G = 1, else a tensor on `device`)."""
        cand = np.asarray(cand_global_all, dtype=np.int64)
        K, n = len(my_queries), cand.shape[1]
        assert cand.shape[0] == self.world * K
        tables = np.ascontiguousarray(register_multi(my_queries, cand[self.rank * K:(self.rank + 1) * K]),
                                      np.float32)
        return self.gather_tables(tables.reshape(K, n, RESULT_COLS), device, capi_knn)

    def gather_tables(self, tables, device, capi_knn=None):
        """This rank's [K, n, RESULT_COLS] result tables (numpy) -> all ranks' [G*K, n, RESULT_COLS], rank-major: ONE
        all-gather (through the C ABI's communicator when `capi_knn` is given).  Collective."""
        K, n = tables.shape[0], tables.shape[1]
        if self.world == 1:
            return tables   # (host memory: the results already are there, nothing to exchange)
        t = torch.from_numpy(np.ascontiguousarray(tables, np.float32).reshape(K * n, RESULT_COLS))
        if capi_knn is not None:   # the all-gather through the C ABI's communicator
            return capi_knn.all_gather_tables(t.to(device).view(K, n, RESULT_COLS))
        t = t.to(self.comm_device or device)
        out = torch.empty((self.world * K * n, RESULT_COLS), dtype=torch.float32, device=t.device)
        dist.all_gather_into_tensor(out, t, group=self.group)
        return out.to(device).view(self.world * K, n, RESULT_COLS)


# ---- HIP-backed defaults ------------------------------------------------------------------------

def hip_local_search(index):
    """local_search over a gloc3d_amd.capi.KnnIndex holding this rank's shard (device tensors).
    The search runs on a side stream that is ordered after torch's CURRENT stream (whatever produced
    `q`, and the allocator's reuse of the output blocks) and that the current stream then waits for:
    stream-ordered on both sides, no host synchronisation."""
    side = torch.cuda.Stream()
    index.set_stream(side.cuda_stream)

    def fn(q, k):
        Q = q.shape[0]
        idx = torch.empty((Q, k), dtype=torch.int64, device=q.device)
        d2 = torch.empty((Q, k), dtype=torch.float32, device=q.device)
        cur = torch.cuda.current_stream()
        side.wait_stream(cur)
        index.search_device(q.data_ptr(), Q, k, idx.data_ptr(), d2.data_ptr())
        cur.wait_stream(side)
        return idx, d2
    return fn


def hip_merge(device_ordinal):
    from . import capi

    def fn(all_i, all_d):   # on torch's current stream: ordered with the all-gather before it
        G, Q, k = all_i.shape
        oi = torch.empty((Q, k), dtype=torch.int64, device=all_i.device)
        od = torch.empty((Q, k), dtype=torch.float32, device=all_i.device)
        capi.topk_merge_device(device_ordinal, torch.cuda.current_stream().cuda_stream,
                               all_i.data_ptr(), all_d.data_ptr(), G, Q, k, oi.data_ptr(),
                               od.data_ptr())
        return oi, od
    return fn


def pack_results(r, shape):
    """capi result dict -> float32 [*shape, RESULT_COLS] (pose, rmse, inliers, ok)."""
    out = np.zeros(tuple(shape) + (RESULT_COLS,), np.float32)
    out[..., :16] = r["T"].reshape(tuple(shape) + (16,))
    out[..., 16] = r["rmse"].reshape(shape)
    out[..., 17] = r["inliers"].reshape(shape).astype(np.float32)  # < 2^24: exact
    out[..., 18] = r["ok"].reshape(shape).astype(np.float32)
    return out


def hip_local_register(registrar, params):
    """local_register over a gloc3d_amd.capi.Registrar whose scan store holds this rank's scans
    (local scan id = local row) plus the replicated query scans."""
    def fn(query_scan_id, local_scan_ids, retrieval_ranks):
        r = registrar.batch_ids(query_scan_id, np.asarray(local_scan_ids, np.uint32), params=params,
                                stream_ids=retrieval_ranks)
        m = len(local_scan_ids)
        out = np.zeros((m, RESULT_COLS), np.float32)
        out[:, :16] = r["T"].reshape(m, 16)
        out[:, 16] = r["rmse"]
        out[:, 17] = r["inliers"].astype(np.float32)  # < 2^24: exact
        out[:, 18] = r["ok"].astype(np.float32)
        return out
    return fn
