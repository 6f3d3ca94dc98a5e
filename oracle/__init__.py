"""ctypes loader for the CPU oracle -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
package, and only as the checker.  The product package ``gloc3d_amd`` never imports it.

``libgloc_oracle.so``  -- this repo's C restatement (oracle/knn_oracle.c, oracle/reg_oracle.c)
``_ref/libgloc_ref.so`` -- the reference's own vendored nanoflann behind a C harness
                           (oracle/ref_harness.cpp); built only where /root/reference exists.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libgloc_oracle.so")
REF_PATH = os.path.join(HERE, "_ref", "libgloc_ref.so")

_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_u64p = np.ctypeslib.ndpointer(np.uint64, flags="C_CONTIGUOUS")
_u32p = np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")


class RegParams(C.Structure):
    _fields_ = [
        ("ransac_iters", C.c_uint32),
        ("inlier_thresh", C.c_float),
        ("min_inlier_ratio", C.c_float),
        ("icp_iters", C.c_uint32),
        ("max_corr_dist", C.c_float),
        ("seed", C.c_uint64),
        ("ransac_confidence", C.c_float),
        ("max_rmse", C.c_float),
        ("max_final_step", C.c_float),
    ]


class NnBackend(C.Structure):
    _fields_ = [("build", C.c_void_p), ("query", C.c_void_p), ("free_", C.c_void_p)]


class BevInfo(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("min_ix", "min_iy", "min_iz", "max_ix", "max_iy", "max_iz")] + [
        ("width", C.c_uint32), ("height", C.c_uint32),
        ("ox", C.c_double), ("oy", C.c_double), ("resolution", C.c_double),
        ("n_returns", C.c_uint32), ("n_cells_known", C.c_uint32),
        ("n_cells_obstructed", C.c_uint32), ("reserved_", C.c_uint32)]


class GroundParams(C.Structure):
    _fields_ = [("near_range2", C.c_float), ("knn", C.c_uint32), ("plane_thresh", C.c_float),
                ("ransac_iters", C.c_uint32), ("ransac_conf", C.c_float), ("reserved_", C.c_uint32),
                ("seed", C.c_uint64)]


class GroundInfo(C.Structure):
    _fields_ = [("n_near", C.c_uint32), ("hist", C.c_uint32 * 18), ("ground_bin", C.c_int32),
                ("n_ground", C.c_uint32), ("best_hyp", C.c_uint32), ("inliers", C.c_uint32),
                ("iters_used", C.c_uint32), ("plane", C.c_float * 4), ("found", C.c_int32)]


def build(ref=True, quiet=True):
    """Compile the oracle (and, where /root/reference exists, oracle/_ref)."""
    out = subprocess.DEVNULL if quiet else None
    subprocess.check_call(["make", "-s", "-C", HERE, "all"], stdout=out)
    if ref and os.path.exists("/root/reference/registration/nanoflann.hpp"):
        subprocess.check_call(["make", "-s", "-C", HERE, "ref"], stdout=out)


_lib = None
_ref = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build(ref=False)
        L = C.CDLL(LIB_PATH)
        L.oracle_l2_eval.restype = C.c_float
        L.oracle_l2_eval.argtypes = [_f32p, _f32p, C.c_size_t]
        L.oracle_knn_search.argtypes = [_f32p, C.c_size_t, C.c_size_t, _f32p, C.c_size_t,
                                        C.c_size_t, C.c_size_t, C.c_size_t, _u64p, _f32p]
        L.oracle_knn_search_mt.argtypes = L.oracle_knn_search.argtypes + [C.c_int]
        L.oracle_recall_at.restype = C.c_size_t
        L.oracle_recall_at.argtypes = [_u64p, C.c_size_t, C.c_size_t, _u64p, _u64p,
                                       C.POINTER(C.c_int), C.c_size_t, C.POINTER(C.c_float)]
        L.oracle_transform_points.argtypes = [_f32p, _f32p, C.c_size_t, _f32p]
        L.oracle_nn3.argtypes = [_f32p, C.c_size_t, _f32p, C.c_size_t, _u32p, _f32p]
        L.oracle_nn3_grid.argtypes = L.oracle_nn3.argtypes
        L.oracle_kabsch_from_cov.argtypes = [_f64p, _f64p, _f64p, _f64p, _f64p]
        L.oracle_ransac_sample.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, _u32p]
        L.oracle_ransac_hypothesis.restype = C.c_int
        L.oracle_ransac_hypothesis.argtypes = [_f32p, _f32p, _u32p, C.c_uint32, C.c_uint64,
                                               C.c_uint32, C.c_uint32, _f32p, _f32p]
        L.oracle_count_inliers.restype = C.c_uint32
        L.oracle_count_inliers.argtypes = [_f32p, _f32p, _u32p, C.c_uint32, _f32p, _f32p, C.c_float]
        L.oracle_reg_one.argtypes = [_f32p, C.c_size_t, _f32p, C.c_size_t, C.c_void_p,
                                     C.POINTER(RegParams), C.c_uint32, _f32p,
                                     C.POINTER(C.c_float), C.POINTER(C.c_uint32),
                                     C.POINTER(C.c_uint32), C.POINTER(C.c_int)]
        L.oracle_reg_one_nn.argtypes = [_f32p, C.c_size_t, _f32p, C.c_size_t, C.c_void_p,
                                        C.POINTER(RegParams), C.c_uint32, C.c_void_p, _f32p,
                                        C.POINTER(C.c_float), C.POINTER(C.c_uint32),
                                        C.POINTER(C.c_uint32), C.POINTER(C.c_int)]
        L.oracle_reg_one_nn_step.restype = None
        L.oracle_reg_one_nn_step.argtypes = L.oracle_reg_one_nn.argtypes + [C.POINTER(C.c_float)]
        L.oracle_reg_many_mt.argtypes = [_f32p, C.c_size_t, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t),
                                         C.c_size_t, C.POINTER(RegParams), C.c_void_p, C.c_void_p, C.c_int,
                                         _f32p, _f32p, _u32p, np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")]
        L.oracle_pose_error.argtypes = [_f32p, _f32p, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.oracle_rng_key.restype = C.c_uint64
        L.oracle_rng_key.argtypes = [C.c_uint64, C.c_uint64]
        L.oracle_rng_draw.restype = C.c_uint64
        L.oracle_rng_draw.argtypes = [C.c_uint64, C.c_uint64]
        L.oracle_rng_gauss.restype = C.c_float
        L.oracle_rng_gauss.argtypes = [C.c_uint64, C.c_uint64]
        L.oracle_synth_iid.argtypes = [C.c_uint64, C.c_size_t, C.c_size_t, C.c_size_t, _f32p]
        L.oracle_bev_project.restype = C.c_int
        L.oracle_bev_project.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_float, C.c_float,
                                         C.POINTER(C.c_void_p), C.POINTER(BevInfo)]
        L.oracle_bev_crop_pad.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32,
                                          C.c_uint32, C.c_void_p]
        L.oracle_bev_to_chw_f32.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
        L.oracle_free.argtypes = [C.c_void_p]
        L.oracle_ground_knn.argtypes = [_f32p, C.c_size_t, C.c_uint32, _u32p, _f32p]
        L.oracle_ground_normals.argtypes = [_f32p, C.c_size_t, _u32p, C.c_uint32, _f32p, C.c_void_p]
        L.oracle_ground_estimate.restype = C.c_int
        L.oracle_ground_estimate.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.POINTER(GroundParams),
                                             _f32p, C.POINTER(GroundInfo)]
        L.oracle_ground_transform_from_plane.argtypes = [_f32p, _f32p]
        L.oracle_jacobi_eig3.argtypes = [_f64p, _f64p]
        L.oracle_coarse_grid_from_image.restype = C.c_void_p
        L.oracle_coarse_grid_from_image.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_float, C.c_float,
                                                    C.c_float, C.c_uint32]
        L.oracle_coarse_grid_free.argtypes = [C.c_void_p]
        L.oracle_coarse_grid_cells.restype = C.c_uint32
        L.oracle_coarse_grid_cells.argtypes = [C.c_void_p, C.c_void_p]
        L.oracle_coarse_match.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_uint32, C.c_uint32, C.c_uint32,
                                          C.c_uint32, C.c_uint32, C.c_float, _f32p, C.POINTER(C.c_float), C.POINTER(C.c_int),
                                          C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.oracle_coarse_match_scale.argtypes = L.oracle_coarse_match.argtypes + [C.POINTER(C.c_float), C.POINTER(C.c_uint32)]
        _lib = L
    return _lib


def have_ref():
    return os.path.exists(REF_PATH)


def ref():
    """The reference's vendored nanoflann (oracle/_ref); raises if it was never built."""
    global _ref
    if _ref is None:
        if not have_ref():
            raise RuntimeError("oracle/_ref/libgloc_ref.so not built (needs /root/reference)")
        R = C.CDLL(REF_PATH)
        R.ref_knn_build.restype = C.c_void_p
        R.ref_knn_build.argtypes = [_f32p, C.c_size_t, C.c_size_t]
        R.ref_knn_query.argtypes = [C.c_void_p, _f32p, C.c_size_t, C.c_size_t, _u64p, _f32p]
        R.ref_knn_free.argtypes = [C.c_void_p]
        R.ref_nn3_build.restype = C.c_void_p
        R.ref_nn3_build.argtypes = [_f32p, C.c_size_t]
        R.ref_nn3_query.argtypes = [C.c_void_p, _f32p, C.c_size_t, _u32p, _f32p]
        R.ref_nn3_free.argtypes = [C.c_void_p]
        _ref = R
    return _ref


# ---- convenience wrappers ------------------------------------------------------------------

def knn_search(db, queries, k, first_row=0, last_row=None, threads=1):
    db = np.ascontiguousarray(db, np.float32)
    queries = np.ascontiguousarray(queries, np.float32).reshape(-1, db.shape[1])
    n, dim = db.shape
    nq = queries.shape[0]
    if last_row is None:
        last_row = n
    idx = np.empty((nq, k), np.uint64)
    d2 = np.empty((nq, k), np.float32)
    if threads > 1:
        lib().oracle_knn_search_mt(db, n, dim, queries, nq, k, first_row, last_row, idx, d2, threads)
    else:
        lib().oracle_knn_search(db, n, dim, queries, nq, k, first_row, last_row, idx, d2)
    return idx, d2


def recall_at(idx, positives, k_values=(1, 5, 10, 20)):
    """recall@N with the reference's first-hit semantics; positives = list of index lists."""
    idx = np.ascontiguousarray(idx, np.uint64)
    nq, k = idx.shape
    off = np.zeros(nq + 1, np.uint64)
    off[1:] = np.cumsum([len(p) for p in positives])
    pos = np.ascontiguousarray(np.concatenate([np.asarray(p, np.uint64) for p in positives] +
                                              [np.zeros(0, np.uint64)]), np.uint64)
    if pos.size == 0:
        pos = np.zeros(1, np.uint64)
    kv = (C.c_int * len(k_values))(*k_values)
    rec = (C.c_float * len(k_values))()
    valid = lib().oracle_recall_at(idx, nq, k, pos, off, kv, len(k_values), rec)
    return valid, list(rec)


def ref_knn_search(db, queries, k):
    db = np.ascontiguousarray(db, np.float32)
    queries = np.ascontiguousarray(queries, np.float32).reshape(-1, db.shape[1])
    n, dim = db.shape
    nq = queries.shape[0]
    R = ref()
    h = R.ref_knn_build(db, n, dim)
    idx = np.empty((nq, k), np.uint64)
    d2 = np.empty((nq, k), np.float32)
    R.ref_knn_query(h, queries, nq, k, idx, d2)
    R.ref_knn_free(h)
    return idx, d2


def nn3(src, tgt, grid=False):
    src = np.ascontiguousarray(src, np.float32).reshape(-1, 3)
    tgt = np.ascontiguousarray(tgt, np.float32).reshape(-1, 3)
    idx = np.empty(src.shape[0], np.uint32)
    d2 = np.empty(src.shape[0], np.float32)
    f = lib().oracle_nn3_grid if grid else lib().oracle_nn3
    f(src, src.shape[0], tgt, tgt.shape[0], idx, d2)
    return idx, d2


def ref_nn3(src, tgt):
    src = np.ascontiguousarray(src, np.float32).reshape(-1, 3)
    tgt = np.ascontiguousarray(tgt, np.float32).reshape(-1, 3)
    R = ref()
    h = R.ref_nn3_build(tgt, tgt.shape[0])
    idx = np.empty(src.shape[0], np.uint32)
    d2 = np.empty(src.shape[0], np.float32)
    R.ref_nn3_query(h, src, src.shape[0], idx, d2)
    R.ref_nn3_free(h)
    return idx, d2


def transform_points(T, xyz):
    T = np.ascontiguousarray(T, np.float32).reshape(16)
    xyz = np.ascontiguousarray(xyz, np.float32).reshape(-1, 3)
    out = np.empty_like(xyz)
    lib().oracle_transform_points(T, xyz, xyz.shape[0], out)
    return out


def _ref_nn_backend():
    """The reference's vendored nanoflann kd-tree (oracle/_ref) as the 1-NN search of oracle_reg_one_nn."""
    R = ref()
    cast = lambda f: C.cast(f, C.c_void_p).value
    return NnBackend(cast(R.ref_nn3_build), cast(R.ref_nn3_query), cast(R.ref_nn3_free))


def reg_many_mt(src, tgts, threads, ref_nn=False, cand_ids=None, ransac_iters=3000, inlier_thresh=0.6,
                min_inlier_ratio=0.3, icp_iters=30, max_corr_dist=0.0, seed=1234, ransac_confidence=0.99,
                max_rmse=0.0, max_final_step=0.0):
    """Candidates of one query registered on `threads` host threads (bench.py's all-cores leg)."""
    src = np.ascontiguousarray(src, np.float32).reshape(-1, 3)
    ts = [np.ascontiguousarray(t, np.float32).reshape(-1, 3) for t in tgts]
    n = len(ts)
    prm = RegParams(ransac_iters, inlier_thresh, min_inlier_ratio, icp_iters, max_corr_dist, seed,
                    ransac_confidence, max_rmse, max_final_step)
    ptrs = (C.c_void_p * n)(*[t.ctypes.data for t in ts])
    cnts = (C.c_size_t * n)(*[t.shape[0] for t in ts])
    T = np.empty((n, 16), np.float32)
    rmse, inl, ok = np.empty(n, np.float32), np.empty(n, np.uint32), np.empty(n, np.int32)
    be = _ref_nn_backend() if ref_nn else None
    cid = None if cand_ids is None else np.ascontiguousarray(cand_ids, np.uint32)
    lib().oracle_reg_many_mt(src, src.shape[0], ptrs, cnts, n, C.byref(prm),
                             None if cid is None else cid.ctypes.data_as(C.c_void_p),
                             C.cast(C.byref(be), C.c_void_p) if be is not None else None, threads, T, rmse, inl, ok)
    return dict(T=T.reshape(n, 4, 4), rmse=rmse, inliers=inl, ok=ok.astype(bool))


def reg_one(src, tgt, init_T=None, cand_id=0, ransac_iters=3000, inlier_thresh=0.6,
            min_inlier_ratio=0.3, icp_iters=30, max_corr_dist=0.0, seed=1234, ransac_confidence=0.99,
            ref_nn=False, max_rmse=0.0, max_final_step=0.0):
    src = np.ascontiguousarray(src, np.float32).reshape(-1, 3)
    tgt = np.ascontiguousarray(tgt, np.float32).reshape(-1, 3)
    prm = RegParams(ransac_iters, inlier_thresh, min_inlier_ratio, icp_iters, max_corr_dist, seed,
                    ransac_confidence, max_rmse, max_final_step)
    T = np.empty(16, np.float32)
    rmse, inl, hyp, ok, step = C.c_float(), C.c_uint32(), C.c_uint32(), C.c_int(), C.c_float()
    if init_T is not None:
        it = np.ascontiguousarray(init_T, np.float32).reshape(16)
        itp = it.ctypes.data_as(C.c_void_p)
    else:
        itp = None
    be = _ref_nn_backend() if ref_nn else None
    lib().oracle_reg_one_nn_step(src, src.shape[0], tgt, tgt.shape[0], itp, C.byref(prm), cand_id,
                                 C.cast(C.byref(be), C.c_void_p) if be is not None else None, T,
                                 C.byref(rmse), C.byref(inl), C.byref(hyp), C.byref(ok), C.byref(step))
    return dict(T=T.reshape(4, 4), rmse=rmse.value, inliers=inl.value, best_hyp=hyp.value,
                ok=bool(ok.value), final_step=step.value)


def pose_error(T_gt, T_est):
    a = np.ascontiguousarray(T_gt, np.float32).reshape(16)
    b = np.ascontiguousarray(T_est, np.float32).reshape(16)
    er, ep = C.c_float(), C.c_float()
    lib().oracle_pose_error(a, b, C.byref(er), C.byref(ep))
    return er.value, ep.value


def bev_project(points, resolution=0.2, max_range=100.0):
    """get_projected_grid: returns (raw image [H,W] u8 or None when empty, info dict)."""
    pts = np.ascontiguousarray(points, np.float32)
    stride = pts.shape[1] if pts.ndim == 2 else 3
    img_p, info = C.c_void_p(), BevInfo()
    rc = lib().oracle_bev_project(pts.ctypes.data, pts.shape[0], stride, resolution, max_range,
                                  C.byref(img_p), C.byref(info))
    d = {n: getattr(info, n) for n, _ in BevInfo._fields_ if n != "reserved_"}
    if rc:
        return None, d
    img = np.ctypeslib.as_array(C.cast(img_p, C.POINTER(C.c_uint8)), (info.height, info.width)).copy()
    lib().oracle_free(img_p)
    return img, d


def bev_crop_pad(img, out_w=768, out_h=768):
    src = np.ascontiguousarray(img, np.uint8)
    dst = np.empty((out_h, out_w, 3), np.uint8)
    lib().oracle_bev_crop_pad(src.ctypes.data, src.shape[1], src.shape[0], out_w, out_h, dst.ctypes.data)
    return dst


def bev_to_chw_f32(hwc3):
    src = np.ascontiguousarray(hwc3, np.uint8)
    out = np.empty((3, src.shape[0], src.shape[1]), np.float32)
    lib().oracle_bev_to_chw_f32(src.ctypes.data, src.shape[1], src.shape[0], out.ctypes.data)
    return out


def ground_params(near_range2=400.0, knn=10, plane_thresh=0.1, ransac_iters=1000, ransac_conf=0.99, seed=0):
    return GroundParams(near_range2, knn, plane_thresh, ransac_iters, ransac_conf, 0, seed)


def ground_knn(xyz, k=10):
    p = np.ascontiguousarray(xyz, np.float32)
    idx = np.empty((p.shape[0], k), np.uint32)
    d2 = np.empty((p.shape[0], k), np.float32)
    lib().oracle_ground_knn(p, p.shape[0], k, idx, d2)
    return idx, d2


def ground_normals(xyz, knn_idx):
    p = np.ascontiguousarray(xyz, np.float32)
    nb = np.ascontiguousarray(knn_idx, np.uint32)
    nrm = np.empty((p.shape[0], 3), np.float32)
    bins = np.empty(p.shape[0], np.uint8)
    lib().oracle_ground_normals(p, p.shape[0], nb, nb.shape[1], nrm, bins.ctypes.data)
    return nrm, bins


def ground_estimate(points, **kw):
    """EsitmateGroundAndTransform: (T_l2g 4x4 f32, info dict)."""
    pts = np.ascontiguousarray(points, np.float32)
    prm, info, T = ground_params(**kw), GroundInfo(), np.empty(16, np.float32)
    lib().oracle_ground_estimate(pts.ctypes.data, pts.shape[0], pts.shape[1], C.byref(prm), T, C.byref(info))
    d = {n: getattr(info, n) for n, _ in GroundInfo._fields_}
    d["hist"] = np.array(list(info.hist), np.uint32)
    d["plane"] = np.array(list(info.plane), np.float32)
    return T.reshape(4, 4), d


def ground_transform_from_plane(plane):
    T = np.empty(16, np.float32)
    lib().oracle_ground_transform_from_plane(np.ascontiguousarray(plane, np.float32), T)
    return T.reshape(4, 4)


# ---- coarse global (x, y, yaw) match (row a-12) ---------------------------------------------------

class CoarseGrid:
    """A scan's coarse search grid from its BEV occupancy image (0 = occupied) and (ox, oy, res)."""

    def __init__(self, img, ox, oy, res, cell_px=2):
        img = np.ascontiguousarray(img, np.uint8)
        self._g = lib().oracle_coarse_grid_from_image(img.ctypes.data_as(C.c_void_p), img.shape[1], img.shape[0],
                                                      ox, oy, res, cell_px)
        self.res, self.cell_px = res, cell_px

    def __del__(self):
        try:
            lib().oracle_coarse_grid_free(self._g)
        except Exception:
            pass

    def cells(self):
        n = lib().oracle_coarse_grid_cells(self._g, None)
        out = np.empty(n, np.uint32)
        lib().oracle_coarse_grid_cells(self._g, out.ctypes.data_as(C.c_void_p))
        return out


def coarse_match(q, d, n_yaw=360, max_shift=64, top_yaw=12, refine=4, min_overlap=0.25):
    xyyaw = np.empty(3, np.float32)
    ratio, ok, over, k, scale, nm = C.c_float(), C.c_int(), C.c_uint32(), C.c_uint32(), C.c_float(), C.c_uint32()
    lib().oracle_coarse_match_scale(q._g, d._g, q.res, q.cell_px, n_yaw, max_shift, top_yaw, refine, min_overlap, xyyaw,
                                    C.byref(ratio), C.byref(ok), C.byref(over), C.byref(k), C.byref(scale), C.byref(nm))
    return dict(xy_yaw=xyyaw, ratio=ratio.value, ok=bool(ok.value), overlap=over.value, k=k.value, scale=scale.value,
                matched=nm.value)
