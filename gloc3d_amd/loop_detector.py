"""Python mirror of the reference's RpyPCLoopDetector + GlocEvaluator for the hot path
(registration/loop_detector.h:41-119, registration/global_localization.cpp:202-574): same method
names, guards and constants, over the C ABI.  Descriptors come from the caller (the CNN backbone is
upstream of the hot path); get_projected_grid / get_place_input are the BEV projection in front of
it and match() is the 3-D RANSAC-SVD + ICP registration.
"""
import numpy as np

from . import capi


class RpyPCLoopDetector:
    NUM_EXCLUDE_RECENT = 30            # loop_detector.h:77

    def __init__(self, k_dim=512, device=0, top_k=20):
        self.k_dim_ = k_dim            # loop_detector.h:97
        self.top_k_ = top_k            # :98
        self.num_exclude_recent_ = 30  # :99
        self.tree_making_period_ = 30  # :100
        self.tree_making_period_counter_ = 0
        self.loop_metric_dist_th_ = 0.8  # :103, compared with the SQUARED distance (:54)
        self._searchable_end = 0
        self._index = capi.KnnIndex(k_dim, device)
        self._reg = capi.Registrar(device)
        self._db_scan_ids = []
        self._last_descriptor = None
        self.reg_params = capi.default_reg_params()
        self.high_resolution_max_range_ = 100.0  # loop_detector.h:115
        self.high_resolution_ = 0.2              # :116
        self._bev = None                         # created on first use
        self._ground = None
        self._device = device
        self._coarse = capi.CoarseMatcher(device)  # db_grids_ (loop_detector.h:108) as search grids
        self._db_grid_ids = []
        self.use_coarse_match = True             # match(): seed the 3-D registration with the 2-D match

    def close(self):
        self._index.close()
        self._reg.close()
        self._coarse.close()
        if self._bev is not None:
            self._bev.close()
        if self._ground is not None:
            self._ground.close()

    def _projector(self):
        if self._bev is None:
            self._bev = capi.BevProjector(self._device)
        return self._bev

    def _bev_params(self, **over):
        return capi.default_bev_params(resolution=self.high_resolution_,
                                       max_range=self.high_resolution_max_range_, **over)

    def align_to_ground(self, cloud):
        """GroundEstimator::EsitmateGroundAndTransform (registration/ground_estimator.cpp:196-228), the
        pre-step of the evaluator's 4th-argument mode (global_localization.cpp:431-436, 495-499):
        returns (T_l2g 4x4, ground-aligned cloud); T is the identity when no ground is found."""
        if self._ground is None:
            self._ground = capi.GroundEstimator(self._device)
        T, _, moved = self._ground.estimate(cloud, want_cloud=True)
        return T, moved

    def get_projected_grid(self, q_pc):
        """loop_detector.cpp:122-135: (occupancy image [H,W] u8, xy_res = (ox, oy, resolution))."""
        bev = self._projector()
        _, info = bev.project(q_pc, self._bev_params())
        if info["empty"]:
            raise ValueError("no point of the scan lies within range")  # the reference aborts in cv::Mat
        return bev.raw_image(info), (info["ox"], info["oy"], info["resolution"])

    def get_place_input(self, q_pc, width=768, height=768):
        """The tensor get_place_feature feeds the descriptor network (loop_detector.cpp:137-151):
        crop_pad_occupancy to width x height, /255, NHWC -> NCHW.  Returns ([1,3,H,W] f32, xy_res)."""
        chw, info = self._projector().project(
            q_pc, self._bev_params(out_width=width, out_height=height, format=capi.BEV_F32_CHW))
        return chw[None], (info["ox"], info["oy"], info["resolution"])

    def __len__(self):
        return len(self._db_scan_ids)

    def add_keyframe(self, descriptor, scan):
        """loop_detector.cpp:10-20: append one place (descriptor + its scan [n,3|4])."""
        d = np.ascontiguousarray(descriptor, np.float32).reshape(1, self.k_dim_)
        self._index.add(d)
        # a database place is a registration TARGET for the rest of the run: kd-ordered index (once, ~1 ms)
        self._db_scan_ids.append(self._reg.scan_build_target_index(self._reg.scan_upload(scan)))
        self._db_grid_ids.append(self._coarse.add_scan(scan))   # loop_detector.cpp:16-19: the place's grid
        self._last_descriptor = d

    def detect(self, q_descriptor):
        """Global localization (loop_detector.cpp:22-46): (indices, squared distances), or two
        empty arrays when the database is too small (:27-30)."""
        if len(self) <= self.num_exclude_recent_ + self.top_k_:
            print("Not enough keyframes in database.")
            return np.zeros(0, np.uint64), np.zeros(0, np.float32)
        idx, d2 = self._index.search(np.asarray(q_descriptor, np.float32).reshape(1, -1), self.top_k_)
        return idx[0], d2[0]

    def detect_slam(self):
        """SLAM mode (loop_detector.cpp:48-81): the newest keyframe is the query; every 30th call
        the searchable window is refreshed to db[0 : end-30]; accept iff best d2 < 0.8.
        Returns (found, q_idx, loop_idx)."""
        n = len(self)
        if n <= self.num_exclude_recent_ + self.top_k_:
            return False, None, None
        if self.tree_making_period_counter_ % self.tree_making_period_ == 0:
            self._searchable_end = n - self.num_exclude_recent_
        self.tree_making_period_counter_ += 1
        idx, d2 = self._index.search(self._last_descriptor, self.top_k_, 0, self._searchable_end)
        if d2[0, 0] < self.loop_metric_dist_th_:
            return True, n - 1, int(idx[0, 0])
        return False, None, None

    def match_2d(self, q_scan, db_indices):
        """RpyPCLoopDetector::match(q_grid, db_idx, xy_yaw, scale) (loop_detector.cpp:186-288) for several
        places at once: the coarse pose of the query in each place's frame, p_db = R(yaw) p_q + (x, y).
        Returns (xy_yaw [n, 3], overlap ratio [n], ok [n]); ok includes the reference's |1 - scale| < 0.1, the
        estimated scales are left in self.last_match_scale."""
        qg = self._coarse.add_scan(np.ascontiguousarray(q_scan, np.float32))
        try:
            out = self._coarse.match(qg, [self._db_grid_ids[int(i)] for i in db_indices])
            self.last_match_scale = self._coarse.last_scale
            return out
        finally:
            self._coarse.release(qg)

    @staticmethod
    def embed_3d(xy_yaw):
        """(x, y, yaw) -> 4x4: R = RollPitchYaw(0, 0, yaw), t = (x, y, 0) (global_localization.cpp:526-530)."""
        x, y, yaw = (float(v) for v in xy_yaw)
        T = np.eye(4, dtype=np.float32)
        c, s_ = np.cos(yaw), np.sin(yaw)
        T[:2, :2] = [[c, -s_], [s_, c]]
        T[0, 3], T[1, 3] = x, y
        return T

    def match(self, q_scan, db_indices, init_T=None):
        """Register the query scan against the retrieved places in one batch; returns
        (rank of the first successful candidate or -1, its 4x4 pose query->db, full result).
        Unless the caller gives initial poses, every candidate whose coarse 2-D match succeeds starts
        from it (the reference composes its pose from that match, global_localization.cpp:519-572)."""
        ids = [self._db_scan_ids[int(i)] for i in db_indices]
        q = np.ascontiguousarray(q_scan, np.float32)
        if init_T is None and self.use_coarse_match and len(ids):
            xy_yaw, _, ok2d = self.match_2d(q, db_indices)
            init_T = np.stack([self.embed_3d(v) if o else np.eye(4, dtype=np.float32) for v, o in zip(xy_yaw, ok2d)])
        qid = self._reg.scan_upload(q)
        try:
            res = self._reg.batch_ids(qid, ids, params=self.reg_params, init_T=init_T)
        finally:
            self._reg.scan_release(qid)  # the query scan is transient: HBM stays flat over a run
        r = capi.reg_select_first_ok(res["ok"].astype(np.int32))
        return r, (res["T"][r] if r >= 0 else np.eye(4, dtype=np.float32)), res


def recognition_recalls(queried_idx, gt_pos, k_values=(1, 5, 10, 20)):
    """recall@N with the reference's first-hit semantics (global_localization.cpp:221-268,
    main.py:336-348).  Returns (recalls, failed query indices)."""
    rec = np.zeros(len(k_values))
    valid, failed = 0, []
    for i, pos in enumerate(gt_pos):
        if len(pos) == 0:
            continue
        valid += 1
        cand = list(queried_idx[i])
        if len(cand) == 0:
            failed.append(i)
            continue
        detected = False
        for ki, k in enumerate(k_values):
            if any(int(c) in set(int(p) for p in pos) for c in cand[:k]):
                rec[ki] += 1
                detected = True
        if not detected:
            failed.append(i)
    return (rec / valid if valid else rec), failed


def pose_error(q2db_gt, est):
    """err_rot (deg, ~180-degree flips forgiven) and err_pos (m): global_localization.cpp:288-306."""
    g, e = np.asarray(q2db_gt, np.float32), np.asarray(est, np.float32)
    off = np.float32(0.5) * (np.trace(g[:3, :3].T @ e[:3, :3]) - np.float32(1))
    off = min(max(off, -0.999999), 0.999999)
    er = abs(float(np.arccos(off))) * 180.0 / np.pi
    ep = float(np.linalg.norm(g[:3, 3] - e[:3, 3]))
    if abs(er - 180.0) < 5.0:
        er = abs(er - 180.0)
    return er, ep


def registration_recalls(located, poses_db_q, n_db):
    """Success = err_pos < 1 m and err_rot < 5 deg (global_localization.cpp:270-335).
    located: list of (db_idx or >= n_db for failure, 4x4 pose)."""
    ok_rot, ok_pos, failed = [], [], []
    for i, (db_idx, pose) in enumerate(located):
        if db_idx >= n_db:
            failed.append(i)
            continue
        q2db = np.linalg.inv(poses_db_q[db_idx].astype(np.float64)) @ poses_db_q[n_db + i]
        er, ep = pose_error(q2db, pose)
        if ep < 1.0 and er < 5.0:
            ok_rot.append(er)
            ok_pos.append(ep)
    n = len(located)
    return dict(success_rate=(len(ok_pos) / n if n else 0.0), rot_mean=float(np.mean(ok_rot)) if ok_rot else 0.0,
                pos_mean=float(np.mean(ok_pos)) if ok_pos else 0.0, failed=failed, succeeded=len(ok_pos))
