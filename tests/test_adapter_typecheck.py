"""The ORB_SLAM3::ORBextractor / ORBmatcher adapter classes (the -DORBX_WITH_OPENCV halves of include/orbx/*.hpp) must
type-check against the reference's signatures.  OpenCV, Eigen and Sophus are absent here, so the check compiles them
with -fsyntax-only against declaration-only stand-ins (tests/cpp/typecheck/reference_decls.hpp); nothing is run."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not available")
def test_adapter_classes_type_check():
    out = subprocess.run(["g++", "-std=c++17", "-Wall", "-Werror", "-fsyntax-only", "-I", os.path.join(ROOT, "tests/cpp/typecheck"),
                          "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests/cpp/typecheck/check.cpp")],
                         capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
