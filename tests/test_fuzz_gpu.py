"""Short runs of the randomised comparisons under tools/ (round 6: + the chained launch against launch by launch) (the long runs' logs: profiles/r04_knn_fuzz.txt,
profiles/r04_reg_fuzz.txt): matrix-core kNN paths == exact path, culled 1-NN == exhaustive 1-NN, bit for bit."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("tool,cases,seed", [("fuzz_knn.py", 40, 5), ("fuzz_reg.py", 200, 5), ("fuzz_chain.py", 30, 5)])
def test_randomised_comparison(tool, cases, seed):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool), str(cases), str(seed)], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "mismatches: 0" in r.stdout
