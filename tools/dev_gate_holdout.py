"""dev: the held-out check of gloc_reg_params.max_final_step (tools/gate_holdout.py): prints the table behind
DESIGN.md section 4 / legs.gate_holdout.  GATE_PRIOR=identity: the 3-D stage unseeded (which the reference never does)."""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gate_holdout as gh
views = gh.build_views("/tmp/gate_holdout_views.npz", workers=min(16, os.cpu_count() or 1))     # (before the GPU is touched)
res = gh.run(views, coarse=os.environ.get("GATE_PRIOR", "coarse") != "identity")
rows = res.pop("rows")
print("query rank kind         dist_m coarse_ok inlier_ok right  err_m err_deg final_step")
for r in rows:
    print(f"{r[0]:5d} {r[1]:4d} {r[2]:12s} {r[3]:6.2f} {int(r[4]):9d} {int(r[5]):9d} {int(r[6]):5d} {r[7]:6.2f} {r[8]:7.2f} {r[9]:10.4f}")
print(json.dumps(res, indent=1))
