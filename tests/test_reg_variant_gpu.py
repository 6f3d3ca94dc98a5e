"""GPU: the culled 1-NN kernel with a 128-entry work queue (lib/libgloc3d_smallq.so, -DGLOC_NN_QCAP=128).

At the shipped queue size (512 entries) the early evaluation -- the queue flushed in the middle of a chunk's test
steps because the next step's items might not fit -- runs for a handful of chunks per launch; with 128 entries it
runs on nearly every chunk, bounds tighten between the steps of one chunk, tail rounds of every size occur.  The
tests that pin the search to the brute-force kernel, the oracle and the reference's kd-tree golden are run again
through that library (a child pytest with GLOC3D_LIB_PATH set): the results must not change in a bit."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VARIANT = os.path.join(ROOT, "gloc3d_amd", "lib", "libgloc3d_smallq.so")
SELECT = ("every_pass_bit_identical or lattice or contested or culled_equals_exhaustive or golden or full_size_properties "
          "or nan_points or batch_matches_oracle")


def test_bit_identity_tests_through_the_small_queue_library():
    if not os.path.exists(VARIANT):
        pytest.fail("gloc3d_amd/lib/libgloc3d_smallq.so is missing: run __graft_entry__.build() (gloc3d_amd.build.build_test_variant)")
    env = dict(os.environ, GLOC3D_LIB_PATH=VARIANT)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_reg_gpu.py"), "-x", "-q", "-m", "gpu",
                        "-k", SELECT, "-p", "no:cacheprovider"], env=env, cwd=ROOT, capture_output=True, text=True, timeout=1500)
    tail = (r.stdout or "")[-3000:] + (r.stderr or "")[-1500:]
    assert r.returncode == 0, tail
    assert " passed" in r.stdout and "no tests ran" not in r.stdout, tail
