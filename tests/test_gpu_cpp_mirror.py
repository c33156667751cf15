"""Runs the C++ host-mirror property tests (tests/cpp/test_projective.cpp — the reference's
projective proptests re-stated over elliptic-curves_amd/host/ecgpu.hpp) on the GPU."""
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def test_cpp_mirror_builds():
    """CPU: the mirror compiles and links against the C ABI."""
    subprocess.check_call(["make", "-s", "-C", os.path.join(HERE, "cpp")])
    assert os.path.exists(os.path.join(HERE, "cpp", "test_projective"))


@pytest.mark.gpu
def test_cpp_mirror_proptests():
    subprocess.check_call(["make", "-s", "-C", os.path.join(HERE, "cpp")])
    out = subprocess.run([os.path.join(HERE, "cpp", "test_projective"), "6"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "all ok" in out.stdout
