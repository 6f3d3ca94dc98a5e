"""GPU: the culled 1-NN kernel evaluating its work queue before every test step (lib/libgloc3d_smallq.so, -DGLOC_NN_EAGER).

The shipped kernel evaluates the FULL rounds that are waiting between two test steps of a chunk and everything at the
chunk's end; the variant evaluates everything before every step, so bounds tighten between all steps and tail rounds
of every size occur.  The tests that pin the search to the brute-force kernel, the oracle and the reference's kd-tree
golden are run again through that library (a child pytest with GLOC3D_LIB_PATH set): the results must not change in a
bit."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VARIANT = os.path.join(ROOT, "gloc3d_amd", "lib", "libgloc3d_smallq.so")
SELECT = ("every_pass_bit_identical or lattice or contested or culled_equals_exhaustive or golden or full_size_properties "
          "or nan_points or batch_matches_oracle or split")


def test_bit_identity_tests_through_the_small_queue_library():
    if not os.path.exists(VARIANT):
        pytest.fail("gloc3d_amd/lib/libgloc3d_smallq.so is missing: run __graft_entry__.build() (gloc3d_amd.build.build_test_variant)")
    env = dict(os.environ, GLOC3D_LIB_PATH=VARIANT)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_reg_gpu.py"), "-x", "-q", "-m", "gpu",
                        "-k", SELECT, "-p", "no:cacheprovider"], env=env, cwd=ROOT, capture_output=True, text=True, timeout=1500)
    tail = (r.stdout or "")[-3000:] + (r.stderr or "")[-1500:]
    assert r.returncode == 0, tail
    assert " passed" in r.stdout and "no tests ran" not in r.stdout, tail
