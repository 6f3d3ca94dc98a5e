"""ctypes binding of the C ABI in include/gloc3d.h (gloc3d_amd/lib/libgloc3d.so).

This is the same binding a reference-side maintainer would write.  It fails loudly
when the HIP extension is missing or no gfx950 device is usable: there is no CPU fallback.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GLOC3D_LIB_PATH") or os.path.join(HERE, "lib", "libgloc3d.so")  # (override: dev builds)

GLOC_OK = 0
ERR_NAMES = {1: "GLOC_ERR_INVALID", 2: "GLOC_ERR_HIP", 3: "GLOC_ERR_NOMEM", 4: "GLOC_ERR_NODEVICE",
             5: "GLOC_ERR_STATE"}
ALGO_AUTO, ALGO_EXACT, ALGO_MFMA, ALGO_MFMA_FP32 = 0, 1, 2, 3   # include/gloc3d.h GLOC_KNN_ALGO_*
KNN_OPT_ALGO, KNN_OPT_CANDIDATES, KNN_OPT_PROFILE = 1, 2, 3
REG_OPT_PROFILE, REG_OPT_NN_MODE, REG_OPT_NN_SRC_PER_LANE, REG_OPT_NN_JOB_GROUP, REG_OPT_TEMP_TARGET_INDEX = 1, 2, 3, 4, 5
REG_OPT_NN_SPLIT_HELPERS, REG_OPT_NN_SPLIT_THRESH, REG_OPT_NN_SUB_JOBS, REG_OPT_NN_HEAVY_THRESH, REG_OPT_SUB_BATCHES = 6, 7, 8, 9, 10
REG_OPT_NN_CHAIN = 11
REG_NN_CULLED, REG_NN_EXHAUSTIVE = 0, 1
NO_SCAN = 0xFFFFFFFF
SIZE_MAX = C.c_size_t(-1).value


class GlocError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"{ERR_NAMES.get(code, code)}: {msg}")
        self.code = code


class RaycastParams(C.Structure):
    _fields_ = [("n_beams", C.c_uint32), ("n_az", C.c_uint32), ("max_range", C.c_double), ("noise_sigma", C.c_double),
                ("fov_lo_deg", C.c_double), ("fov_hi_deg", C.c_double)]


class KnnStats(C.Structure):
    _fields_ = [("searches_exact", C.c_uint64), ("searches_mfma", C.c_uint64),
                ("queries_total", C.c_uint64), ("queries_fallback", C.c_uint64),
                ("last_n_tile", C.c_uint32), ("last_k_split", C.c_uint32),
                ("last_candidates", C.c_uint32)]


class RegParams(C.Structure):
    _fields_ = [("ransac_iters", C.c_uint32), ("inlier_thresh", C.c_float),
                ("min_inlier_ratio", C.c_float), ("icp_iters", C.c_uint32),
                ("max_corr_dist", C.c_float), ("seed", C.c_uint64),
                ("ransac_confidence", C.c_float), ("max_rmse", C.c_float), ("max_final_step", C.c_float)]


class BevParams(C.Structure):
    _fields_ = [("resolution", C.c_float), ("max_range", C.c_float), ("out_width", C.c_uint32),
                ("out_height", C.c_uint32), ("format", C.c_uint32), ("pad_bgr", C.c_uint8 * 3),
                ("reserved_", C.c_uint8)]


class BevInfo(C.Structure):
    _fields_ = [("min_ix", C.c_int32), ("min_iy", C.c_int32), ("max_ix", C.c_int32),
                ("max_iy", C.c_int32), ("width", C.c_uint32), ("height", C.c_uint32),
                ("n_returns", C.c_uint32), ("empty", C.c_uint32), ("ox", C.c_double),
                ("oy", C.c_double), ("resolution", C.c_double)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


BEV_U8_HWC3, BEV_F32_CHW = 0, 1
GROUND_OPT_KNN_EXHAUSTIVE = 1


class CoarseParams(C.Structure):
    _fields_ = [("resolution", C.c_float), ("cell_px", C.c_uint32), ("n_yaw", C.c_uint32), ("max_shift", C.c_uint32),
                ("top_yaw", C.c_uint32), ("refine", C.c_uint32), ("min_overlap", C.c_float), ("reserved_", C.c_uint32)]


class GroundParams(C.Structure):
    _fields_ = [("near_range2", C.c_float), ("knn", C.c_uint32), ("plane_thresh", C.c_float),
                ("ransac_iters", C.c_uint32), ("ransac_conf", C.c_float), ("reserved_", C.c_uint32),
                ("seed", C.c_uint64)]


class GroundInfo(C.Structure):
    _fields_ = [("n_near", C.c_uint32), ("hist", C.c_uint32 * 18), ("ground_bin", C.c_int32),
                ("n_ground", C.c_uint32), ("best_hyp", C.c_uint32), ("inliers", C.c_uint32),
                ("iters_used", C.c_uint32), ("plane", C.c_float * 4), ("found", C.c_int32)]

    def as_dict(self):
        d = {n: getattr(self, n) for n, _ in self._fields_}
        d["hist"] = np.array(list(self.hist), np.uint32)
        d["plane"] = np.array(list(self.plane), np.float32)
        return d


# every symbol include/gloc3d.h declares: (name, restype, argtypes)
_vp, _sz, _u64, _i, _u32 = C.c_void_p, C.c_size_t, C.c_uint64, C.c_int, C.c_uint32
_PROTOS = [
    ("gloc_last_error", C.c_char_p, []),
    ("gloc_abi_version", _i, []),
    ("gloc_device_count", _i, []),
    ("gloc_knn_create", _i, [_i, _sz, C.POINTER(_vp)]),
    ("gloc_knn_create_view", _i, [_vp, C.POINTER(_vp)]),
    ("gloc_knn_destroy", _i, [_vp]),
    ("gloc_knn_set_stream", _i, [_vp, _vp]),
    ("gloc_knn_synchronize", _i, [_vp]),
    ("gloc_knn_set_option", _i, [_vp, _i, C.c_int64]),
    ("gloc_knn_add", _i, [_vp, _vp, _sz]),
    ("gloc_knn_add_device", _i, [_vp, _vp, _sz]),
    ("gloc_knn_reserve", _i, [_vp, _sz]),
    ("gloc_knn_clear", _i, [_vp]),
    ("gloc_knn_size", _i, [_vp, C.POINTER(_sz)]),
    ("gloc_knn_dim", _i, [_vp, C.POINTER(_sz)]),
    ("gloc_knn_device_rows", _i, [_vp, C.POINTER(_vp)]),
    ("gloc_knn_save", _i, [_vp, C.c_char_p]),
    ("gloc_knn_load", _i, [_vp, C.c_char_p]),
    ("gloc_knn_search", _i, [_vp, _vp, _sz, _sz, _sz, _sz, _vp, _vp]),
    ("gloc_knn_search_device", _i, [_vp, _vp, _sz, _sz, _sz, _sz, _u64, _vp, _vp]),
    ("gloc_topk_merge_device", _i, [_i, _vp, _vp, _vp, _sz, _sz, _sz, _vp, _vp]),
    ("gloc_comm_unique_id", _i, [_vp]),
    ("gloc_comm_create", _i, [_i, _i, _i, _vp, C.POINTER(_vp)]),
    ("gloc_comm_destroy", _i, [_vp]),
    ("gloc_comm_rank", _i, [_vp, C.POINTER(_i), C.POINTER(_i)]),
    ("gloc_comm_all_gather_device", _i, [_vp, _vp, _vp, _sz, _vp]),
    ("gloc_knn_search_sharded", _i, [_vp, _vp, _vp, _sz, _sz, _u64, _u64, _vp, _vp]),
    ("gloc_knn_search_sharded_host", _i, [_vp, _vp, _vp, _sz, _sz, _u64, _u64, _vp, _vp]),
    ("gloc_comm_all_gather_host", _i, [_vp, _vp, _vp, _sz]),
    ("gloc_knn_get_stats", _i, [_vp, C.POINTER(KnnStats)]),
    ("gloc_knn_profile", _i, [_vp, C.c_char_p, C.POINTER(C.c_double), C.POINTER(_u64)]),
    ("gloc_knn_profile_reset", _i, [_vp]),
    ("gloc_reg_default_params", None, [C.POINTER(RegParams)]),
    ("gloc_reg_create", _i, [_i, C.POINTER(_vp)]),
    ("gloc_reg_destroy", _i, [_vp]),
    ("gloc_reg_set_stream", _i, [_vp, _vp]),
    ("gloc_reg_synchronize", _i, [_vp]),
    ("gloc_reg_set_option", _i, [_vp, _i, C.c_int64]),
    ("gloc_scan_store_create", _i, [_i, C.POINTER(_vp)]),
    ("gloc_scan_store_destroy", _i, [_vp]),
    ("gloc_scan_store_add", _i, [_vp, _vp, _sz, _sz, C.POINTER(_u32)]),
    ("gloc_scan_store_add_device", _i, [_vp, _vp, _sz, _sz, C.POINTER(_u32)]),
    ("gloc_scan_store_add_variant", _i, [_vp, _u32, _vp, C.c_float, _u64, C.POINTER(_u32)]),
    ("gloc_scan_store_add_raycast_batch", _i, [_vp, _sz, _vp, _vp, _vp, C.c_double, _vp, _vp, _vp, _vp]),
    ("gloc_scan_store_build_target_index", _i, [_vp, _u32]),
    ("gloc_scan_store_build_target_index_batch", _i, [_vp, _vp, _sz]),
    ("gloc_scan_store_add_batch", _i, [_vp, _vp, _vp, _sz, _sz, _vp]),
    ("gloc_scan_store_release", _i, [_vp, _u32]),
    ("gloc_scan_store_clear", _i, [_vp]),
    ("gloc_scan_store_count", _i, [_vp, C.POINTER(_sz)]),
    ("gloc_scan_store_bytes", _i, [_vp, C.POINTER(_sz), C.POINTER(_sz)]),
    ("gloc_scan_store_points", _i, [_vp, _u32, C.POINTER(_sz)]),
    ("gloc_scan_store_download", _i, [_vp, _u32, _vp, _sz]),
    ("gloc_reg_attach_store", _i, [_vp, _vp]),
    ("gloc_reg_scan_upload", _i, [_vp, _vp, _sz, _sz, C.POINTER(_u32)]),
    ("gloc_reg_scan_build_target_index", _i, [_vp, _u32]),
    ("gloc_reg_scan_release", _i, [_vp, _u32]),
    ("gloc_reg_scan_count", _i, [_vp, C.POINTER(_sz)]),
    ("gloc_reg_scan_clear", _i, [_vp]),
    ("gloc_reg_batch_multi", _i, [_vp, _sz, _vp, _vp, _sz, _vp, _vp, C.POINTER(RegParams), _vp, _vp, _vp,
                                  _vp]),
    ("gloc_reg_batch_multi_begin", _i, [_vp, _sz, _vp, _vp, _sz, _vp, _vp, C.POINTER(RegParams)]),
    ("gloc_reg_batch_multi_end", _i, [_vp, _vp, _vp, _vp, _vp]),
    ("gloc_reg_batch", _i, [_vp, _vp, _sz, C.POINTER(_vp), C.POINTER(_sz), _sz, _vp, _vp,
                            C.POINTER(RegParams), _vp, _vp, _vp, _vp]),
    ("gloc_reg_batch_ids", _i, [_vp, _u32, _vp, _sz, _vp, _vp, C.POINTER(RegParams), _vp, _vp, _vp,
                                _vp]),
    ("gloc_reg_first_success_multi", _i, [_vp, _sz, _vp, _vp, _sz, _vp, C.POINTER(RegParams), _vp, _vp, _vp, _vp,
                                          C.POINTER(_u64)]),
    ("gloc_reg_select_first_ok", _i, [_vp, _sz]),
    ("gloc_reg_final_steps", _i, [_vp, _vp, _sz]),
    ("gloc_reg_nn", _i, [_vp, _vp, _sz, _vp, _sz, _vp, _vp, _vp]),
    ("gloc_reg_ransac_hypotheses", _i, [_vp, _vp, _vp, _vp, _sz, _u64, _u32, _u32, _vp, _vp, _vp,
                                        C.c_float]),
    ("gloc_reg_profile", _i, [_vp, C.c_char_p, C.POINTER(C.c_double), C.POINTER(_u64)]),
    ("gloc_reg_profile_reset", _i, [_vp]),
    ("gloc_reg_nn_stats", _i, [_vp, C.POINTER(_u64), C.POINTER(_u64)]),
    ("gloc_vlad_create", _i, [_i, _sz, _sz, _sz, _vp, _vp, _vp, _vp, _i, C.POINTER(_vp)]),
    ("gloc_vlad_set_gating", _i, [_vp, _vp, _vp, _vp]),
    ("gloc_vlad_destroy", _i, [_vp]),
    ("gloc_vlad_set_stream", _i, [_vp, _vp]),
    ("gloc_vlad_forward", _i, [_vp, _vp, _sz, _sz, _vp]),
    ("gloc_vlad_forward_device", _i, [_vp, _vp, _sz, _sz, _vp]),
    ("gloc_vlad_set_profile", _i, [_vp, _i]),
    ("gloc_vlad_profile", _i, [_vp, C.c_char_p, C.POINTER(C.c_double), C.POINTER(_u64)]),
    ("gloc_bev_default_params", _i, [_vp]),
    ("gloc_bev_create", _i, [_i, C.POINTER(_vp)]),
    ("gloc_bev_destroy", _i, [_vp]),
    ("gloc_bev_set_stream", _i, [_vp, _vp]),
    ("gloc_bev_synchronize", _i, [_vp]),
    ("gloc_bev_project", _i, [_vp, _vp, _sz, _sz, _vp, _vp, _vp]),
    ("gloc_bev_project_batch_device", _i, [_vp, _vp, _vp, _sz, _sz, _vp, _vp, _vp]),
    ("gloc_bev_raw_image", _i, [_vp, _sz, _vp, _sz]),
    ("gloc_bev_set_profile", _i, [_vp, _i]),
    ("gloc_bev_profile", _i, [_vp, C.c_char_p, C.POINTER(C.c_double), C.POINTER(_u64)]),
    ("gloc_bev_device_flags", _i, [_vp, _sz, C.POINTER(_vp), C.POINTER(_i), C.POINTER(_i)]),
    ("gloc_coarse_default_params", _i, [_vp]),
    ("gloc_coarse_create", _i, [_i, C.POINTER(_vp)]),
    ("gloc_coarse_destroy", _i, [_vp]),
    ("gloc_coarse_add_image", _i, [_vp, _vp, _u32, _u32, C.c_float, C.c_float, C.c_float, _vp, C.POINTER(_u32)]),
    ("gloc_coarse_add_scan", _i, [_vp, _vp, _sz, _sz, _vp, C.POINTER(_u32)]),
    ("gloc_coarse_add_store_scan", _i, [_vp, _vp, _u32, _vp, C.POINTER(_u32)]),
    ("gloc_coarse_add_store_scans", _i, [_vp, _vp, _vp, _sz, _vp, _vp]),
    ("gloc_coarse_match_pairs", _i, [_vp, _vp, _vp, _sz, _vp, _vp, _vp, _vp, _vp]),
    ("gloc_coarse_release", _i, [_vp, _u32]),
    ("gloc_coarse_cells", _i, [_vp, _u32, C.POINTER(_u32), _vp, _sz]),
    ("gloc_coarse_match", _i, [_vp, _u32, _vp, _sz, _vp, _vp, _vp, _vp, _vp]),
    ("gloc_ground_default_params", _i, [_vp]),
    ("gloc_ground_create", _i, [_i, C.POINTER(_vp)]),
    ("gloc_ground_destroy", _i, [_vp]),
    ("gloc_ground_set_stream", _i, [_vp, _vp]),
    ("gloc_ground_set_option", _i, [_vp, _i, C.c_int64]),
    ("gloc_ground_estimate", _i, [_vp, _vp, _sz, _sz, _vp, _vp, _vp, _vp]),
    ("gloc_ground_estimate_device", _i, [_vp, _vp, _sz, _sz, _vp, _vp, _vp, _vp]),
    ("gloc_ground_knn", _i, [_vp, _vp, _sz, _u32, _vp, _vp]),
    ("gloc_ground_normals", _i, [_vp, _vp, _sz, _u32, _vp, _vp]),
    ("gloc_ground_transform_from_plane", _i, [_vp, _vp]),
    ("gloc_ground_set_profile", _i, [_vp, _i]),
    ("gloc_ground_profile", _i, [_vp, C.c_char_p, C.POINTER(C.c_double), C.POINTER(_u64)]),
    ("gloc_knn_add_synthetic", _i, [_vp, _i, _u64, _u64, _sz, _u64]),
    ("gloc_synth_fill_device", _i, [_i, _vp, _i, _u64, _u64, _sz, _sz, _u64, _vp]),
]
EXPORTED_SYMBOLS = [p[0] for p in _PROTOS]

_lib = None


def _preload_hip_runtime():
    """libgloc3d.so carries no NEEDED entry for the HIP runtime: exactly one must be in the process.
    PyTorch bundles its own libamdhip64; when torch is installed we run on that copy so that torch
    tensors, torch.distributed (RCCL) and our kernels share one runtime.  GLOC3D_HIP_RUNTIME
    overrides the choice."""
    cands = []
    if os.environ.get("GLOC3D_HIP_RUNTIME"):
        cands.append(os.environ["GLOC3D_HIP_RUNTIME"])
    try:
        import torch  # noqa: F401  (loads its bundled runtime)
        cands.append(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
    except ImportError:
        pass
    cands += ["/opt/rocm/lib/libamdhip64.so", "libamdhip64.so"]
    for p in cands:
        try:
            C.CDLL(p, mode=C.RTLD_GLOBAL)
            return p
        except OSError:
            continue
    raise RuntimeError("no HIP runtime (libamdhip64.so) found: " + ", ".join(cands))


def lib():
    """Load libgloc3d.so (no compute; safe without a GPU).  Raises if it was never built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build the HIP extension first "
                "(python -m gloc3d_amd.build or __graft_entry__.build()); there is no CPU fallback")
        _preload_hip_runtime()
        L = C.CDLL(LIB_PATH)
        for name, res, args in _PROTOS:
            fn = getattr(L, name)  # AttributeError if the library lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc):
    if rc != GLOC_OK:
        raise GlocError(rc, lib().gloc_last_error().decode("utf-8", "replace"))


def _np_ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class KnnIndex:
    """Resident descriptor database + exact L2 top-k (the reference's InvKeyTree / IndexFlatL2)."""

    def __init__(self, dim, device=0):
        self._h = C.c_void_p()
        self.dim = int(dim)
        self.device = device
        check(lib().gloc_knn_create(device, self.dim, C.byref(self._h)))

    def close(self):
        if self._h:
            check(lib().gloc_knn_destroy(self._h))
            self._h = C.c_void_p()
            self._parent = None

    def view(self):
        """A second search handle over this index's rows (gloc_knn_create_view): own stream and workspace, so that a search
        on it runs beside a search on this handle.  Close it before its parent."""
        v = KnnIndex.__new__(KnnIndex)
        v._h = C.c_void_p()
        v.dim, v.device = self.dim, self.device
        check(lib().gloc_knn_create_view(self._h, C.byref(v._h)))
        v._parent = self          # (keeps the parent alive as long as the view)
        return v

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __len__(self):
        n = C.c_size_t()
        check(lib().gloc_knn_size(self._h, C.byref(n)))
        return n.value

    def set_option(self, option, value):
        check(lib().gloc_knn_set_option(self._h, option, int(value)))

    def set_stream(self, hip_stream):
        check(lib().gloc_knn_set_stream(self._h, C.c_void_p(hip_stream or 0)))

    def synchronize(self):
        check(lib().gloc_knn_synchronize(self._h))

    def reserve(self, n):
        check(lib().gloc_knn_reserve(self._h, n))

    def clear(self):
        check(lib().gloc_knn_clear(self._h))

    def add(self, rows):
        rows = np.ascontiguousarray(rows, np.float32).reshape(-1, self.dim)
        check(lib().gloc_knn_add(self._h, _np_ptr(rows), rows.shape[0]))

    def add_device(self, dev_ptr, n):
        check(lib().gloc_knn_add_device(self._h, C.c_void_p(dev_ptr), n))

    def add_synthetic(self, kind, seed, first_row, n, row_stride=1):
        check(lib().gloc_knn_add_synthetic(self._h, kind, seed, first_row, n, row_stride))

    def save(self, path):
        check(lib().gloc_knn_save(self._h, str(path).encode()))

    def load(self, path):
        check(lib().gloc_knn_load(self._h, str(path).encode()))

    def device_rows(self):
        p = C.c_void_p()
        check(lib().gloc_knn_device_rows(self._h, C.byref(p)))
        return p.value

    def search(self, queries, k, first_row=0, last_row=None):
        q = np.ascontiguousarray(queries, np.float32).reshape(-1, self.dim)
        nq = q.shape[0]
        idx = np.empty((nq, k), np.uint64)
        d2 = np.empty((nq, k), np.float32)
        last = SIZE_MAX if last_row is None else last_row
        check(lib().gloc_knn_search(self._h, _np_ptr(q), nq, k, first_row, last, _np_ptr(idx),
                                    _np_ptr(d2)))
        return idx, d2

    def search_device(self, q_ptr, nq, k, idx_ptr, d2_ptr, first_row=0, last_row=None,
                      index_offset=0):
        last = SIZE_MAX if last_row is None else last_row
        check(lib().gloc_knn_search_device(self._h, C.c_void_p(q_ptr), nq, k, first_row, last,
                                           index_offset, C.c_void_p(idx_ptr), C.c_void_p(d2_ptr)))

    def search_sharded(self, comm, q_ptr, nq, k, idx_ptr, d2_ptr, index_stride=1, index_offset=0):
        """Collective: top-k over the whole row-sharded database (RCCL all-gather + merge on the device)."""
        check(lib().gloc_knn_search_sharded(self._h, comm._h, C.c_void_p(q_ptr), nq, k, index_stride, index_offset,
                                            C.c_void_p(idx_ptr), C.c_void_p(d2_ptr)))

    def stats(self):
        s = KnnStats()
        check(lib().gloc_knn_get_stats(self._h, C.byref(s)))
        return {f[0]: getattr(s, f[0]) for f in KnnStats._fields_}

    def profile(self, kernel):
        ms, n = C.c_double(), C.c_uint64()
        check(lib().gloc_knn_profile(self._h, kernel.encode(), C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def profile_reset(self):
        check(lib().gloc_knn_profile_reset(self._h))


class Comm:
    """RCCL communicator of this process's GPU (gloc_comm_*).  `exchange(id_bytes or None) -> id_bytes`
    broadcasts rank 0's 128-byte id to every rank (e.g. over torch.distributed or a shared file)."""

    def __init__(self, device, rank, world, exchange):
        uid = (C.c_uint8 * 128)()
        err = None
        if rank == 0:
            try:
                check(lib().gloc_comm_unique_id(uid))
            except Exception as e:      # the other ranks are waiting in exchange(): hand them a null id, then fail
                err, uid = e, (C.c_uint8 * 128)()
        data = exchange(bytes(uid) if rank == 0 else None)
        if err is not None:
            raise err
        assert len(data) == 128
        if not any(data):
            raise GlocError(5, "rank 0 could not create the RCCL id")
        uid = (C.c_uint8 * 128).from_buffer_copy(data)
        self._h = C.c_void_p()
        self.rank, self.world = rank, world
        check(lib().gloc_comm_create(device, rank, world, uid, C.byref(self._h)))

    def close(self):
        if self._h:
            lib().gloc_comm_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def rank_world(self):
        """(rank, world) as the communicator itself reports them (gloc_comm_rank)."""
        r, w = C.c_int(), C.c_int()
        check(lib().gloc_comm_rank(self._h, C.byref(r), C.byref(w)))
        return r.value, w.value

    def all_gather_device(self, send_ptr, recv_ptr, bytes_per_rank, stream=0):
        check(lib().gloc_comm_all_gather_device(self._h, C.c_void_p(send_ptr), C.c_void_p(recv_ptr), bytes_per_rank,
                                                C.c_void_p(stream or 0)))


def topk_merge_device(device, stream, idx_ptr, d2_ptr, n_lists, nq, k, out_idx_ptr, out_d2_ptr):
    check(lib().gloc_topk_merge_device(device, C.c_void_p(stream or 0), C.c_void_p(idx_ptr),
                                       C.c_void_p(d2_ptr), n_lists, nq, k,
                                       C.c_void_p(out_idx_ptr), C.c_void_p(out_d2_ptr)))


def synth_fill_device(device, stream, kind, seed, first_row, n, dim, out_ptr, row_stride=1):
    check(lib().gloc_synth_fill_device(device, C.c_void_p(stream or 0), kind, seed, first_row, n,
                                       dim, row_stride, C.c_void_p(out_ptr)))


def default_reg_params(**over):
    p = RegParams()
    lib().gloc_reg_default_params(C.byref(p))
    if os.environ.get("GLOC3D_MAX_FINAL_STEP"):          # developer override of the convergence check's threshold
        p.max_final_step = float(os.environ["GLOC3D_MAX_FINAL_STEP"])
    for k_, v in over.items():
        setattr(p, k_, v)
    return p


class ScanStore:
    """Resident scans + their search index, shared by any number of Registrars."""

    def __init__(self, device=0):
        self._h = C.c_void_p()
        self.device = device
        check(lib().gloc_scan_store_create(device, C.byref(self._h)))

    def close(self):
        if self._h:
            check(lib().gloc_scan_store_destroy(self._h))
            self._h = C.c_void_p()

    def __del__(self):
        try:
            if self._h:
                lib().gloc_scan_store_destroy(self._h)
        except Exception:
            pass

    def add(self, pts):
        pts = np.ascontiguousarray(pts, np.float32)
        assert pts.ndim == 2 and 3 <= pts.shape[1] <= 16
        sid = C.c_uint32()
        check(lib().gloc_scan_store_add(self._h, _np_ptr(pts), pts.shape[0], pts.shape[1], C.byref(sid)))
        return sid.value

    def add_device(self, dev_ptr, n, stride_floats=3):
        sid = C.c_uint32()
        check(lib().gloc_scan_store_add_device(self._h, C.c_void_p(dev_ptr), n, stride_floats, C.byref(sid)))
        return sid.value

    def add_variant(self, base_id, T=None, noise_sigma=0.0, seed=0):
        Tp = None if T is None else np.ascontiguousarray(T, np.float32).reshape(16)
        sid = C.c_uint32()
        check(lib().gloc_scan_store_add_variant(self._h, int(base_id), None if Tp is None else _np_ptr(Tp),
                                                float(noise_sigma), int(seed), C.byref(sid)))
        return sid.value

    def add_raycast(self, world, poses, seeds, n_beams=64, n_az=2000, max_range=80.0, noise=0.02, fov=(-24.8, 2.0),
                    reach_margin=1.0):
        """Ray-cast len(poses) scans on the device (gloc_scan_store_add_raycast_batch; the device twin of
        synth.lidar_scan): world = dict(lo [n,3], hi [n,3], ground), poses = world <- sensor 4x4 matrices.  Every pose is
        given the boxes whose footprint comes within max_range (+ margin) of it.  Returns the scan ids."""
        poses = np.ascontiguousarray(np.asarray(poses, np.float64).reshape(-1, 4, 4))
        lo, hi = np.asarray(world["lo"], np.float64).reshape(-1, 3), np.asarray(world["hi"], np.float64).reshape(-1, 3)
        ids = []
        for a in range(0, len(poses), 64):
            P = poses[a:a + 64]
            o = P[:, :2, 3]                                              # [k, 2]
            gap = np.maximum(np.maximum(lo[None, :, :2] - o[:, None, :], o[:, None, :] - hi[None, :, :2]), 0.0)
            near = np.hypot(gap[..., 0], gap[..., 1]) <= max_range + reach_margin        # [k, n_boxes]
            first = np.zeros(len(P) + 1, np.uint32)
            first[1:] = np.cumsum(near.sum(axis=1))
            sel = [np.nonzero(near[i])[0] for i in range(len(P))]
            cat = np.concatenate(sel) if sel else np.zeros(0, np.int64)
            blo, bhi = np.ascontiguousarray(lo[cat]), np.ascontiguousarray(hi[cat])
            prm = RaycastParams(int(n_beams), int(n_az), float(max_range), float(noise), float(fov[0]), float(fov[1]))
            sd = np.ascontiguousarray(np.asarray(seeds[a:a + 64], np.uint64))
            out = np.empty(len(P), np.uint32)
            check(lib().gloc_scan_store_add_raycast_batch(self._h, len(P), _np_ptr(blo) if len(cat) else None,
                                                          _np_ptr(bhi) if len(cat) else None, _np_ptr(first), float(world["ground"]),
                                                          _np_ptr(P), _np_ptr(sd), C.byref(prm), _np_ptr(out)))
            ids.extend(int(i) for i in out)
        return ids

    def add_batch(self, scans):
        """Several scans ([n_i, c] float32 arrays with the same number of columns) in one launch sequence."""
        arrs = [np.ascontiguousarray(p, np.float32) for p in scans]
        cols = arrs[0].shape[1]
        assert all(a.ndim == 2 and a.shape[1] == cols for a in arrs) and 3 <= cols <= 16
        k = len(arrs)
        ptrs = (C.c_void_p * k)(*[a.ctypes.data for a in arrs])
        cnts = (C.c_size_t * k)(*[a.shape[0] for a in arrs])
        ids = np.empty(k, np.uint32)
        check(lib().gloc_scan_store_add_batch(self._h, ptrs, cnts, k, cols, _np_ptr(ids)))
        return [int(i) for i in ids]

    def build_target_index(self, scan_id):
        """Re-sort the scan's index into kd order: for scans that serve as registration targets (database places)."""
        check(lib().gloc_scan_store_build_target_index(self._h, int(scan_id)))
        return scan_id

    def debug_index(self, scan_id):
        """Test aid: the scan's index as the search sees it -- dict(perm, keys, kpos, order2, kd)."""
        f = lib().gloc_scan_store_debug_index
        f.restype = C.c_int
        f.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]
        n = self.points(scan_id)
        perm, keys, kpos = np.empty(n, np.uint32), np.empty(n, np.uint32), np.empty(n, np.uint32)
        order2 = np.empty((n + 127) // 128, np.uint32)
        kd = C.c_int()
        check(f(self._h, int(scan_id), _np_ptr(perm), _np_ptr(keys), _np_ptr(kpos), _np_ptr(order2), C.byref(kd)))
        return dict(perm=perm, keys=keys, kpos=kpos, order2=order2, kd=bool(kd.value))

    def build_target_index_batch(self, scan_ids):
        ids = np.ascontiguousarray(scan_ids, np.uint32).reshape(-1)
        check(lib().gloc_scan_store_build_target_index_batch(self._h, _np_ptr(ids), ids.shape[0]))

    def release(self, scan_id):
        check(lib().gloc_scan_store_release(self._h, int(scan_id)))

    def clear(self):
        check(lib().gloc_scan_store_clear(self._h))

    def __len__(self):
        n = C.c_size_t()
        check(lib().gloc_scan_store_count(self._h, C.byref(n)))
        return n.value

    def bytes(self):
        a, b = C.c_size_t(), C.c_size_t()
        check(lib().gloc_scan_store_bytes(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def points(self, scan_id):
        n = C.c_size_t()
        check(lib().gloc_scan_store_points(self._h, int(scan_id), C.byref(n)))
        return n.value

    def download(self, scan_id):
        n = self.points(scan_id)
        out = np.empty((n, 3), np.float32)
        check(lib().gloc_scan_store_download(self._h, int(scan_id), _np_ptr(out), n))
        return out


class Registrar:
    """Batched candidate registration (RANSAC-SVD + ICP) over a resident scan store (its own, or a
    shared ScanStore passed in / attached later)."""

    def __init__(self, device=0, store=None):
        self._h = C.c_void_p()
        self.device = device
        self._store = None
        check(lib().gloc_reg_create(device, C.byref(self._h)))
        if store is not None:
            self.attach_store(store)

    def close(self):
        if self._h:
            lib().gloc_reg_destroy(self._h)
            self._h = C.c_void_p()
            self._store = None

    def attach_store(self, store):
        check(lib().gloc_reg_attach_store(self._h, store._h if store is not None else None))
        self._store = store  # keeps the store alive as long as this handle uses it

    def scan_release(self, scan_id):
        check(lib().gloc_reg_scan_release(self._h, int(scan_id)))

    def batch_multi(self, q_ids, cand_ids, init_T=None, params=None, stream_ids=None):
        """q_ids [Q]; cand_ids [Q, n] (NO_SCAN = no candidate).  Returns arrays with leading [Q, n]."""
        q = np.ascontiguousarray(q_ids, np.uint32).reshape(-1)
        ids = np.ascontiguousarray(cand_ids, np.uint32).reshape(q.shape[0], -1)
        Q, n = ids.shape
        prm = params or default_reg_params()
        it = None if init_T is None else np.ascontiguousarray(init_T, np.float32).reshape(Q * n, 16)
        T, rmse, inl, ok = self._outs(Q * n)
        sid = None if stream_ids is None else np.ascontiguousarray(stream_ids, np.uint32).reshape(Q * n)
        check(lib().gloc_reg_batch_multi(self._h, Q, _np_ptr(q), _np_ptr(ids), n,
                                         None if sid is None else _np_ptr(sid),
                                         None if it is None else _np_ptr(it), C.byref(prm),
                                         _np_ptr(T), _np_ptr(rmse), _np_ptr(inl), _np_ptr(ok)))
        return dict(T=T.reshape(Q, n, 4, 4), rmse=rmse.reshape(Q, n), inliers=inl.reshape(Q, n),
                    ok=ok.astype(bool).reshape(Q, n))

    def batch_multi_begin(self, q_ids, cand_ids, init_T=None, params=None, stream_ids=None):
        """Enqueue a batch and return at once; batch_multi_end() waits for it and returns what batch_multi returns."""
        q = np.ascontiguousarray(q_ids, np.uint32).reshape(-1)
        ids = np.ascontiguousarray(cand_ids, np.uint32).reshape(q.shape[0], -1)
        Q, n = ids.shape
        prm = params or default_reg_params()
        it = None if init_T is None else np.ascontiguousarray(init_T, np.float32).reshape(Q * n, 16)
        sid = None if stream_ids is None else np.ascontiguousarray(stream_ids, np.uint32).reshape(Q * n)
        check(lib().gloc_reg_batch_multi_begin(self._h, Q, _np_ptr(q), _np_ptr(ids), n,
                                               None if sid is None else _np_ptr(sid),
                                               None if it is None else _np_ptr(it), C.byref(prm)))
        self._pending_shape = (Q, n)

    def batch_multi_end(self):
        Q, n = getattr(self, "_pending_shape", None) or (1, 1)     # (end without a begin: the library says so, GLOC_ERR_STATE)
        self._pending_shape = None
        T, rmse, inl, ok = self._outs(Q * n)
        check(lib().gloc_reg_batch_multi_end(self._h, _np_ptr(T), _np_ptr(rmse), _np_ptr(inl), _np_ptr(ok)))
        return dict(T=T.reshape(Q, n, 4, 4), rmse=rmse.reshape(Q, n), inliers=inl.reshape(Q, n),
                    ok=ok.astype(bool).reshape(Q, n))

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_option(self, option, value):
        check(lib().gloc_reg_set_option(self._h, option, int(value)))

    def set_stream(self, hip_stream):
        check(lib().gloc_reg_set_stream(self._h, C.c_void_p(hip_stream or 0)))

    def synchronize(self):
        check(lib().gloc_reg_synchronize(self._h))

    def scan_upload(self, pts):
        pts = np.ascontiguousarray(pts, np.float32)
        assert pts.ndim == 2 and 3 <= pts.shape[1] <= 16
        sid = C.c_uint32()
        check(lib().gloc_reg_scan_upload(self._h, _np_ptr(pts), pts.shape[0], pts.shape[1],
                                         C.byref(sid)))
        return sid.value

    def scan_build_target_index(self, scan_id):
        check(lib().gloc_reg_scan_build_target_index(self._h, int(scan_id)))
        return scan_id

    def scan_count(self):
        n = C.c_size_t()
        check(lib().gloc_reg_scan_count(self._h, C.byref(n)))
        return n.value

    def scan_clear(self):
        check(lib().gloc_reg_scan_clear(self._h))

    @staticmethod
    def _outs(n):
        return (np.empty((n, 4, 4), np.float32), np.empty(n, np.float32), np.empty(n, np.uint32),
                np.empty(n, np.int32))

    def batch(self, q_xyz, cands, init_T=None, params=None, stream_ids=None):
        q = np.ascontiguousarray(q_xyz, np.float32).reshape(-1, 3)
        cs = [np.ascontiguousarray(c, np.float32).reshape(-1, 3) for c in cands]
        n = len(cs)
        ptrs = (C.c_void_p * n)(*[c.ctypes.data for c in cs])
        cnts = (C.c_size_t * n)(*[c.shape[0] for c in cs])
        prm = params or default_reg_params()
        it = None if init_T is None else np.ascontiguousarray(init_T, np.float32).reshape(n, 16)
        T, rmse, inl, ok = self._outs(n)
        sid = None if stream_ids is None else np.ascontiguousarray(stream_ids, np.uint32)
        check(lib().gloc_reg_batch(self._h, _np_ptr(q), q.shape[0], ptrs, cnts, n,
                                   None if sid is None else _np_ptr(sid),
                                   None if it is None else _np_ptr(it), C.byref(prm), _np_ptr(T),
                                   _np_ptr(rmse), _np_ptr(inl), _np_ptr(ok)))
        return dict(T=T, rmse=rmse, inliers=inl, ok=ok.astype(bool))

    def batch_ids(self, q_id, cand_ids, init_T=None, params=None, stream_ids=None):
        ids = np.ascontiguousarray(cand_ids, np.uint32)
        n = ids.shape[0]
        prm = params or default_reg_params()
        it = None if init_T is None else np.ascontiguousarray(init_T, np.float32).reshape(n, 16)
        T, rmse, inl, ok = self._outs(n)
        sid = None if stream_ids is None else np.ascontiguousarray(stream_ids, np.uint32)
        check(lib().gloc_reg_batch_ids(self._h, int(q_id), _np_ptr(ids), n,
                                       None if sid is None else _np_ptr(sid),
                                       None if it is None else _np_ptr(it), C.byref(prm),
                                       _np_ptr(T), _np_ptr(rmse), _np_ptr(inl), _np_ptr(ok)))
        return dict(T=T, rmse=rmse, inliers=inl, ok=ok.astype(bool))

    def first_success_multi(self, q_ids, cand_ids, init_T=None, params=None):
        """The reference's stop-at-the-first-success loop for several queries: returns rank [Q] (-1: none),
        T [Q, 4, 4], rmse [Q], inliers [Q] and the number of registrations actually run."""
        q = np.ascontiguousarray(q_ids, np.uint32).reshape(-1)
        ids = np.ascontiguousarray(cand_ids, np.uint32).reshape(q.shape[0], -1)
        Q, n = ids.shape
        prm = params or default_reg_params()
        it = None if init_T is None else np.ascontiguousarray(init_T, np.float32).reshape(Q * n, 16)
        rank = np.empty(Q, np.int32)
        T, rmse, inl = np.empty((Q, 4, 4), np.float32), np.empty(Q, np.float32), np.empty(Q, np.uint32)
        jobs = C.c_uint64()
        check(lib().gloc_reg_first_success_multi(self._h, Q, _np_ptr(q), _np_ptr(ids), n,
                                                 None if it is None else _np_ptr(it), C.byref(prm), _np_ptr(rank),
                                                 _np_ptr(T), _np_ptr(rmse), _np_ptr(inl), C.byref(jobs)))
        return dict(rank=rank, T=T, rmse=rmse, inliers=inl, jobs_run=jobs.value)

    def nn(self, src, tgt, T=None):
        s = np.ascontiguousarray(src, np.float32).reshape(-1, 3)
        t = np.ascontiguousarray(tgt, np.float32).reshape(-1, 3)
        idx = np.empty(s.shape[0], np.uint32)
        d2 = np.empty(s.shape[0], np.float32)
        Tp = None if T is None else np.ascontiguousarray(T, np.float32).reshape(16)
        check(lib().gloc_reg_nn(self._h, _np_ptr(s), s.shape[0], _np_ptr(t), t.shape[0],
                                None if Tp is None else _np_ptr(Tp), _np_ptr(idx), _np_ptr(d2)))
        return idx, d2

    def ransac_hypotheses(self, src, tgt, corr, seed, cand, n_hyp, inlier_thresh):
        s = np.ascontiguousarray(src, np.float32).reshape(-1, 3)
        t = np.ascontiguousarray(tgt, np.float32).reshape(-1, 3)
        c = np.ascontiguousarray(corr, np.uint32)
        Rt = np.empty((n_hyp, 12), np.float32)
        valid = np.empty(n_hyp, np.uint32)
        inl = np.empty(n_hyp, np.uint32)
        check(lib().gloc_reg_ransac_hypotheses(self._h, _np_ptr(s), _np_ptr(t), _np_ptr(c),
                                               s.shape[0], seed, cand, n_hyp, _np_ptr(Rt),
                                               _np_ptr(valid), _np_ptr(inl), inlier_thresh))
        return Rt, valid, inl

    def final_steps(self, n_jobs):
        """Per job of the last batch: RMS displacement of the last ICP update (what max_final_step gates)."""
        out = np.empty(n_jobs, np.float32)
        check(lib().gloc_reg_final_steps(self._h, _np_ptr(out), n_jobs))
        return out

    def profile(self, kernel):
        ms, n = C.c_double(), C.c_uint64()
        check(lib().gloc_reg_profile(self._h, kernel.encode(), C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def profile_reset(self):
        check(lib().gloc_reg_profile_reset(self._h))

    def nn_stats(self):
        c, n = C.c_uint64(), C.c_uint64()
        check(lib().gloc_reg_nn_stats(self._h, C.byref(c), C.byref(n)))
        return c.value, n.value

    def debug_chain(self):
        """Test aid: (chained launches enqueued, chained launches that timed out) -- GLOC_REG_OPT_NN_CHAIN."""
        f = lib().gloc_reg_debug_chain
        f.restype = C.c_int
        f.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        n, t = C.c_uint64(), C.c_uint64()
        check(f(self._h, C.byref(n), C.byref(t)))
        return n.value, t.value

    def debug_needed_iters(self, inl, n, conf, max_iters):
        """Test aid: the adaptive RANSAC stop's iteration count as the device computes it."""
        f = lib().gloc_reg_debug_needed_iters
        f.restype = C.c_int
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_float, C.c_uint32, C.c_void_p]
        a, b = np.ascontiguousarray(inl, np.uint32), np.ascontiguousarray(n, np.uint32)
        out = np.empty(a.shape[0], np.uint32)
        check(f(self._h, _np_ptr(a), _np_ptr(b), a.shape[0], conf, max_iters, _np_ptr(out)))
        return out

    def debug_chain_stall(self, on):
        """Test aid: make the chained launch's solvers wait for a wave that never comes (the bounded waits)."""
        f = lib().gloc_reg_debug_chain_stall
        f.restype = C.c_int
        f.argtypes = [C.c_void_p, C.c_int]
        check(f(self._h, 1 if on else 0))

    def debug_corr(self, job, n_src):
        """Test aid: correspondences of the last 1-NN pass of the last batch (caller's index space)."""
        f = lib().gloc_reg_debug_corr
        f.restype = C.c_int
        f.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
        idx, d2 = np.empty(n_src, np.uint32), np.empty(n_src, np.float32)
        check(f(self._h, job, n_src, _np_ptr(idx), _np_ptr(d2)))
        return idx, d2


def reg_select_first_ok(ok):
    a = np.ascontiguousarray(ok, np.int32)
    return lib().gloc_reg_select_first_ok(_np_ptr(a), a.shape[0])


class NetVladFC:
    """NetVLAD-FC pooling head (model/netvlad_fc.py NetVLAD.forward without gating)."""

    def __init__(self, conv_w, centroids, fc_w, conv_b=None, normalize_input=True, device=0):
        cw = np.ascontiguousarray(conv_w, np.float32)
        ce = np.ascontiguousarray(centroids, np.float32)
        fw = np.ascontiguousarray(fc_w, np.float32)
        cb = None if conv_b is None else np.ascontiguousarray(conv_b, np.float32)
        self.K, self.C = cw.shape
        self.out_dim = fw.shape[1]
        assert ce.shape == (self.K, self.C) and fw.shape[0] == self.K * self.C
        self._h = C.c_void_p()
        check(lib().gloc_vlad_create(device, self.C, self.K, self.out_dim, _np_ptr(cw),
                                     None if cb is None else _np_ptr(cb), _np_ptr(ce), _np_ptr(fw),
                                     1 if normalize_input else 0, C.byref(self._h)))

    def close(self):
        if self._h:
            lib().gloc_vlad_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_gating(self, gating_w=None, scale=None, shift=None):
        """GatingContext after the FC: y * sigmoid((y W) * scale + shift); None switches it off."""
        if gating_w is None:
            check(lib().gloc_vlad_set_gating(self._h, None, None, None))
            return
        gw = np.ascontiguousarray(gating_w, np.float32)
        sc = np.ascontiguousarray(scale, np.float32)
        sh = np.ascontiguousarray(shift, np.float32)
        assert gw.shape == (self.out_dim, self.out_dim) and sc.shape == (self.out_dim,) == sh.shape
        check(lib().gloc_vlad_set_gating(self._h, _np_ptr(gw), _np_ptr(sc), _np_ptr(sh)))

    def forward(self, feat):
        x = np.ascontiguousarray(feat, np.float32)
        n = x.shape[0]
        x = x.reshape(n, self.C, -1)
        out = np.empty((n, self.out_dim), np.float32)
        check(lib().gloc_vlad_forward(self._h, _np_ptr(x), n, x.shape[2], _np_ptr(out)))
        return out

    def forward_device(self, feat_ptr, n, hw, out_ptr):
        check(lib().gloc_vlad_forward_device(self._h, C.c_void_p(feat_ptr), n, hw, C.c_void_p(out_ptr)))

    def set_profile(self, on=True):
        check(lib().gloc_vlad_set_profile(self._h, 1 if on else 0))

    def profile(self, kernel):
        ms, n = C.c_double(), C.c_uint64()
        check(lib().gloc_vlad_profile(self._h, kernel.encode(), C.byref(ms), C.byref(n)))
        return ms.value, n.value


def default_bev_params(**over):
    p = BevParams()
    check(lib().gloc_bev_default_params(C.byref(p)))
    for k, v in over.items():
        if k == "pad_bgr":
            for i in range(3):
                p.pad_bgr[i] = v[i]
        else:
            setattr(p, k, v)
    return p


class BevProjector:
    """BEV occupancy projection (RpyPCLoopDetector::get_projected_grid + crop_pad_occupancy,
    registration/loop_detector.cpp:83-106,122-151)."""

    def __init__(self, device=0):
        self._h = C.c_void_p()
        check(lib().gloc_bev_create(device, C.byref(self._h)))

    def close(self):
        if self._h:
            lib().gloc_bev_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def _out_array(p, n_scans=None):
        shape = (p.out_height, p.out_width, 3) if p.format == BEV_U8_HWC3 else (3, p.out_height, p.out_width)
        if n_scans is not None:
            shape = (n_scans,) + shape
        return np.empty(shape, np.uint8 if p.format == BEV_U8_HWC3 else np.float32)

    def project(self, points, params=None):
        """points [n, 3 or more] float32 -> (image, info dict)."""
        p = params or default_bev_params()
        pts = np.ascontiguousarray(points, np.float32)
        if pts.ndim != 2:
            pts = pts.reshape(-1, 3)
        out, info = self._out_array(p), BevInfo()
        check(lib().gloc_bev_project(self._h, _np_ptr(pts), pts.shape[0], pts.shape[1], C.byref(p),
                                     _np_ptr(out), C.byref(info)))
        return out, info.as_dict()

    def project_batch_device(self, xyz_ptr, offsets, stride_floats, out_ptr, params=None, want_info=True):
        """Device buffers: scans back to back at xyz_ptr, host `offsets` (n_scans + 1, in points)."""
        p = params or default_bev_params()
        off = np.ascontiguousarray(offsets, np.uint64)
        n = off.shape[0] - 1
        infos = (BevInfo * n)() if want_info else None
        check(lib().gloc_bev_project_batch_device(self._h, C.c_void_p(xyz_ptr), _np_ptr(off), n, stride_floats,
                                                  C.byref(p), C.c_void_p(out_ptr),
                                                  C.cast(infos, C.c_void_p) if want_info else None))
        return [i.as_dict() for i in infos] if want_info else None

    def raw_image(self, info, scan=0):
        """The uncropped [height, width] u8 image (occupancy_grid) of a scan of the last projection."""
        out = np.empty((info["height"], info["width"]), np.uint8)
        check(lib().gloc_bev_raw_image(self._h, scan, _np_ptr(out), out.size))
        return out

    def set_stream(self, stream_ptr):
        check(lib().gloc_bev_set_stream(self._h, C.c_void_p(stream_ptr)))

    def synchronize(self):
        check(lib().gloc_bev_synchronize(self._h))

    def set_profile(self, on=True):
        check(lib().gloc_bev_set_profile(self._h, 1 if on else 0))

    def profile(self, kernel):
        ms, n = C.c_double(), C.c_uint64()
        check(lib().gloc_bev_profile(self._h, kernel.encode(), C.byref(ms), C.byref(n)))
        return ms.value, n.value


def default_coarse_params(**over):
    p = CoarseParams()
    check(lib().gloc_coarse_default_params(C.byref(p)))
    for k, v in over.items():
        setattr(p, k, v)
    return p


class CoarseMatcher:
    """Coarse global (x, y, yaw) match on BEV occupancy grids (RpyPCLoopDetector::match on two
    OccupancyGrids, registration/loop_detector.cpp:186-288)."""

    def __init__(self, device=0, params=None):
        self._h = C.c_void_p()
        check(lib().gloc_coarse_create(device, C.byref(self._h)))
        self.params = params or default_coarse_params()

    def close(self):
        if self._h:
            lib().gloc_coarse_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def add_image(self, occupancy, ox, oy, resolution):
        img = np.ascontiguousarray(occupancy, np.uint8)
        gid = C.c_uint32()
        check(lib().gloc_coarse_add_image(self._h, _np_ptr(img), img.shape[1], img.shape[0], ox, oy, resolution,
                                          C.byref(self.params), C.byref(gid)))
        return gid.value

    def add_scan(self, pts):
        pts = np.ascontiguousarray(pts, np.float32)
        gid = C.c_uint32()
        check(lib().gloc_coarse_add_scan(self._h, _np_ptr(pts), pts.shape[0], pts.shape[1], C.byref(self.params),
                                         C.byref(gid)))
        return gid.value

    def add_store_scan(self, store, scan_id):
        gid = C.c_uint32()
        check(lib().gloc_coarse_add_store_scan(self._h, store._h, int(scan_id), C.byref(self.params), C.byref(gid)))
        return gid.value

    def add_store_scans(self, store, scan_ids):
        ids = np.ascontiguousarray(scan_ids, np.uint32).reshape(-1)
        out = np.empty(ids.shape[0], np.uint32)
        check(lib().gloc_coarse_add_store_scans(self._h, store._h, _np_ptr(ids), ids.shape[0], C.byref(self.params), _np_ptr(out)))
        return out

    def match_pairs(self, q_grids, db_grids):
        qs = np.ascontiguousarray(q_grids, np.uint32).reshape(-1)
        ds = np.ascontiguousarray(db_grids, np.uint32).reshape(-1)
        n = qs.shape[0]
        assert ds.shape[0] == n
        xyyaw, ratio, ok = np.empty((n, 3), np.float32), np.empty(n, np.float32), np.empty(n, np.int32)
        self.last_scale = np.empty(n, np.float32)     # the reference's `scale` output of the same match
        check(lib().gloc_coarse_match_pairs(self._h, _np_ptr(qs), _np_ptr(ds), n, C.byref(self.params), _np_ptr(xyyaw),
                                            _np_ptr(ratio), _np_ptr(ok), _np_ptr(self.last_scale)))
        return xyyaw, ratio, ok.astype(bool)

    def release(self, grid_id):
        check(lib().gloc_coarse_release(self._h, int(grid_id)))

    def cells(self, grid_id):
        n = C.c_uint32()
        check(lib().gloc_coarse_cells(self._h, int(grid_id), C.byref(n), None, 0))
        out = np.empty(n.value, np.uint32)
        check(lib().gloc_coarse_cells(self._h, int(grid_id), C.byref(n), _np_ptr(out), n.value))
        return out

    def match(self, q_grid, db_grids):
        ids = np.ascontiguousarray(db_grids, np.uint32)
        n = ids.shape[0]
        xyyaw, ratio, ok = np.empty((n, 3), np.float32), np.empty(n, np.float32), np.empty(n, np.int32)
        self.last_scale = np.empty(n, np.float32)     # the reference's `scale` output of the same match
        check(lib().gloc_coarse_match(self._h, int(q_grid), _np_ptr(ids), n, C.byref(self.params), _np_ptr(xyyaw),
                                      _np_ptr(ratio), _np_ptr(ok), _np_ptr(self.last_scale)))
        return xyyaw, ratio, ok.astype(bool)


def default_ground_params(**over):
    p = GroundParams()
    check(lib().gloc_ground_default_params(C.byref(p)))
    for k, v in over.items():
        setattr(p, k, v)
    return p


def ground_transform_from_plane(plane):
    T = np.empty(16, np.float32)
    check(lib().gloc_ground_transform_from_plane(_np_ptr(np.ascontiguousarray(plane, np.float32)), _np_ptr(T)))
    return T.reshape(4, 4)


class GroundEstimator:
    """Ground pre-alignment (GroundEstimator::EsitmateGroundAndTransform,
    registration/ground_estimator.cpp:196-228)."""

    def __init__(self, device=0):
        self._h = C.c_void_p()
        check(lib().gloc_ground_create(device, C.byref(self._h)))

    def close(self):
        if self._h:
            lib().gloc_ground_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_option(self, option, value):
        check(lib().gloc_ground_set_option(self._h, option, value))

    def estimate(self, points, params=None, want_cloud=False):
        """points [n, 3 or more] float32 -> (T_l2g 4x4, info dict[, transformed cloud])."""
        p = params or default_ground_params()
        pts = np.ascontiguousarray(points, np.float32)
        T, info = np.empty(16, np.float32), GroundInfo()
        out = np.empty_like(pts) if want_cloud else None
        check(lib().gloc_ground_estimate(self._h, _np_ptr(pts), pts.shape[0], pts.shape[1], C.byref(p), _np_ptr(T),
                                         C.byref(info), _np_ptr(out) if want_cloud else None))
        return (T.reshape(4, 4), info.as_dict(), out) if want_cloud else (T.reshape(4, 4), info.as_dict())

    def estimate_device(self, xyz_ptr, n, stride_floats, out_ptr=None, params=None):
        p = params or default_ground_params()
        T, info = np.empty(16, np.float32), GroundInfo()
        check(lib().gloc_ground_estimate_device(self._h, C.c_void_p(xyz_ptr), n, stride_floats, C.byref(p), _np_ptr(T),
                                                C.byref(info), C.c_void_p(out_ptr) if out_ptr else None))
        return T.reshape(4, 4), info.as_dict()

    def knn(self, xyz, k=10):
        p = np.ascontiguousarray(xyz, np.float32)
        idx = np.empty((p.shape[0], k), np.uint32)
        d2 = np.empty((p.shape[0], k), np.float32)
        check(lib().gloc_ground_knn(self._h, _np_ptr(p), p.shape[0], k, _np_ptr(idx), _np_ptr(d2)))
        return idx, d2

    def normals(self, xyz, k=10):
        p = np.ascontiguousarray(xyz, np.float32)
        nrm = np.empty((p.shape[0], 3), np.float32)
        bins = np.empty(p.shape[0], np.uint8)
        check(lib().gloc_ground_normals(self._h, _np_ptr(p), p.shape[0], k, _np_ptr(nrm), _np_ptr(bins)))
        return nrm, bins

    def set_profile(self, on=True):
        check(lib().gloc_ground_set_profile(self._h, 1 if on else 0))

    def profile(self, kernel):
        ms, n = C.c_double(), C.c_uint64()
        check(lib().gloc_ground_profile(self._h, kernel.encode(), C.byref(ms), C.byref(n)))
        return ms.value, n.value
