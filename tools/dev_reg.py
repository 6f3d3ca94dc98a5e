"""Developer probe: registration parity + timing on the GPU box (not part of the test suite)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle
from gloc3d_amd import capi, synth

w = synth.make_world(1001)
A = synth.lidar_scan(w, None, seed=1001)[:, :3]
Tgt = synth.se3(5.0, (0.5, -0.3, 0.1))
B = synth.lidar_scan(w, Tgt, seed=1002)[:, :3]
reg = capi.Registrar()
reg.set_option(capi.REG_OPT_PROFILE, 1)

# 1. NN parity (small + medium), with transform
for (ss, ts) in [(50, 20), (8, 4), (1, 1)]:
    s = np.ascontiguousarray(B[::ss]); t = np.ascontiguousarray(A[::ts])
    T = synth.se3(3.0, (0.2, 0.1, 0.0)).astype(np.float32)
    gi, gd = reg.nn(s, t, T)
    sm = oracle.transform_points(T, s)
    oi, od = oracle.nn3(sm, t, grid=True)
    print(f"nn {s.shape[0]}x{t.shape[0]}: idx_eq={(gi==oi).all()} d2_biteq={(gd.view(np.uint32)==od.view(np.uint32)).all()} mism={(gi!=oi).sum()}", flush=True)

# 2. hypothesis parity
s = np.ascontiguousarray(B[::16]); t = np.ascontiguousarray(A[::4])
oi, od = oracle.nn3(s, t)
Rt, valid, inl = reg.ransac_hypotheses(s, t, oi, 1234, 3, 512, 0.6)
L = oracle.lib()
bad_rt = bad_inl = bad_valid = 0
for h in range(512):
    R = np.zeros(9, np.float32); tt = np.zeros(3, np.float32)
    v = L.oracle_ransac_hypothesis(s, t, oi, s.shape[0], 1234, 3, h, R, tt)
    if v != valid[h]: bad_valid += 1; continue
    if not v: continue
    if not (np.concatenate([R, tt]).view(np.uint32) == Rt[h].view(np.uint32)).all(): bad_rt += 1
    ci = L.oracle_count_inliers(s, t, oi, s.shape[0], np.ascontiguousarray(Rt[h][:9]), np.ascontiguousarray(Rt[h][9:]), 0.6)
    if ci != inl[h]: bad_inl += 1
print(f"ransac hyp: valid_mismatch={bad_valid} Rt_bit_mismatch={bad_rt} inlier_mismatch={bad_inl} of 512 (valid {valid.sum()})", flush=True)

# 3. full registration parity on subsampled clouds
q = np.ascontiguousarray(B[::16]); cands = [np.ascontiguousarray(A[::4]), np.ascontiguousarray(A[1::5]), np.ascontiguousarray(synth.lidar_scan(synth.make_world(77), None, seed=5)[::4, :3])]
prm = capi.default_reg_params(ransac_iters=500, icp_iters=10)
t0 = time.time(); g = reg.batch(q, cands, params=prm); t1 = time.time()
for c, cd in enumerate(cands):
    o = oracle.reg_one(q, cd, cand_id=c, ransac_iters=500, icp_iters=10)
    dT = np.abs(g["T"][c] - o["T"]).max()
    er, ep = oracle.pose_error(o["T"], g["T"][c])
    print(f"cand {c}: max|dT|={dT:.2e} rot_err={er:.2e}deg pos_err={ep:.2e}m inliers gpu={g['inliers'][c]} cpu={o['inliers']} besth={o['best_hyp']} ok gpu={g['ok'][c]} cpu={o['ok']} rmse gpu={g['rmse'][c]:.6f} cpu={o['rmse']:.6f}  gt_err={oracle.pose_error(Tgt, g['T'][c])}", flush=True)
print(f"batch wall {t1-t0:.3f}s")

# 4. full-size timing: 1 query x 20 candidates, full clouds
ids = [reg.scan_upload(A)]
for c in range(5):
    Tc = synth.se3(2.0 * c - 4.0, (0.3 * c, -0.2 * c, 0.0))
    ids.append(reg.scan_upload(synth.lidar_scan(w, Tc, seed=3000 + c)[:, :3]))
qid = reg.scan_upload(B)
cand_ids = [ids[i % len(ids)] for i in range(20)]
prm = capi.default_reg_params(ransac_iters=3000, icp_iters=20)
import sys as _s
mode = int(_s.argv[1]) if len(_s.argv) > 1 else 0
reg.set_option(capi.REG_OPT_NN_MODE, mode)
if len(_s.argv) > 2: reg.set_option(capi.REG_OPT_NN_SRC_PER_LANE, int(_s.argv[2]))
reg.batch_ids(qid, cand_ids[:2], params=capi.default_reg_params(ransac_iters=100, icp_iters=1))
reg.profile_reset()
t0 = time.time(); g = reg.batch_ids(qid, cand_ids, params=prm); t1 = time.time()
ch, nl = reg.nn_stats()
print(f"nn mode {mode}: {ch/max(nl,1):.3e} pairs/launch evaluated over {nl} launches")
print(f"full-size 1x20 cands ({B.shape[0]} src pts, ~{A.shape[0]} tgt): wall {t1-t0:.3f}s -> {1/(t1-t0):.2f} q/s")
for n in ["nn", "transform", "ransac_hyp", "ransac_score", "accum", "solve"]:
    ms, cnt = reg.profile(n)
    print(f"   {n}: total {ms:.2f} ms over {cnt} launches = {ms/max(cnt,1):.3f} ms each")
ms, cnt = reg.profile("nn")
pairs = float(B.shape[0]) * sum(reg_n for reg_n in [A.shape[0]] * 20)
print(f"   nn pairs/launch ~{pairs:.3e} -> {pairs/(ms/cnt*1e-3)/1e12:.2f} Tpairs/s; 8 flop/pair -> {8*pairs/(ms/cnt*1e-3)/1e12:.1f} TFLOP/s ({8*pairs/(ms/cnt*1e-3)/157.3e12*100:.1f}% of 157.3)")
print("ok:", g["ok"].astype(int), "inl:", g["inliers"][:6], "gt err cand0:", oracle.pose_error(Tgt, g["T"][0]))
