"""The convergence check gloc_reg_params.max_final_step on data its suggested value (0.03 m) was NOT chosen on
(tools/gate_holdout.py: worlds 3003 / 4004, views and perturbation streams no other test, tool or bench leg uses).
What is pinned here is what the held-out run shows, with slack -- not a tuned pass mark: the first-success rule finds the
right place for every query with the check at 0.03 and with it off; the check removes most of the wrong poses the inlier
test accepts and rejects a few right ones just above it; it cannot remove a pose that CONVERGED a metre off."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))


def test_suggested_final_step_on_unseen_worlds():
    import gate_holdout as gh
    views = gh.build_views("/tmp/gloc3d_gate_holdout_views.npz", workers=0)     # (no fork: the GPU may be in use already)
    res = gh.run(views)
    off, at = res["thresholds"]["off"], res["thresholds"]["0.03"]
    assert res["queries"] == 8 and res["candidates_per_query"] == 20
    assert off["success"] == 8 and at["success"] == 8                       # every query located at its right place
    assert at["located_but_wrong"] == 0 and at["not_located"] == 0
    assert res["right_poses"] >= 50 and res["wrong_poses"] >= 10             # the inlier test alone accepts many wrong poses
    assert at["wrong_poses_accepted"] <= off["wrong_poses_accepted"] // 2    # the check removes most of them ...
    assert at["right_poses_rejected"] <= 4                                   # ... at the price of a few right ones just above it
    assert 0.025 < res["right_pose_final_step_max"] < 0.05
    assert res["wrong_pose_final_step_min"] < 0.03                           # a pose that converged ~1 m off passes: documented limit
