"""Device unit test of the four-lane point addition (xyzz_lazy_add_quad, bp_curve.cuh) that the tree sums of k_small_msm, the bucket
reduce and the heavy-bucket combine run on: compiled here with hipcc from tests/cpp/quad_add_test.hip, it compares the quad form with
the one-lane xyzz_lazy_add as POINTS for generic operands, an identity on either side, doubling and cancellation, on both curves.
(The first version of the kernel failed exactly this test: the compiler's DPP combiner mis-folds two quad-permute moves into one
subtraction; the moves are pinned since.)"""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_quad_add_equals_one_lane_add(tmp_path):
    out = str(tmp_path / "quad_add_test")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-std=c++17", "-w", "-I", os.path.join(ROOT, "bulletproofs-amcl_amd", "csrc"),
                           os.path.join(ROOT, "tests", "cpp", "quad_add_test.hip"), "-o", out], timeout=900)
    p = subprocess.run([out], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0 and p.stdout.count("0 of 64 quads differ") == 2, p.stdout + p.stderr
