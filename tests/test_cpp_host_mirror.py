"""The C++ host mirror (include/bpmsm.hpp) compiled with g++ against libbpmsm.so: a CPU part (Merlin conformance vector,
error mapping) and a GPU part that reruns the reference's own n = 4 unit test shape (src/ipp.rs:325-390)."""
import os
import subprocess

import pytest

import __graft_entry__ as G

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def binary(tmp_path_factory):
    G.build()
    out = str(tmp_path_factory.mktemp("cpp") / "host_mirror_test")
    pkg = os.path.join(ROOT, "bulletproofs-amcl_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "host_mirror_test.cpp"),
                           "-L", pkg, "-lbpmsm", "-Wl,-rpath," + pkg, "-Wl,-rpath,/opt/rocm/lib", "-o", out])
    return out


def test_cpp_mirror_cpu(binary):
    p = subprocess.run([binary, "cpu"], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0 and "cpp cpu ok" in p.stdout, p.stdout + p.stderr


@pytest.mark.gpu
def test_cpp_mirror_gpu(binary):
    p = subprocess.run([binary, "gpu"], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and p.stdout.count("cpp gpu ok") == 2, p.stdout + p.stderr
