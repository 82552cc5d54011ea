"""Sanitizer runs of the CPU-side code (never on the GPU box's device code):
* the saved-map record parser (orb_slam3_amd/csrc/orbx_archive.hip is host code that reads untrusted file bytes)
  compiled with g++ -fsanitize=address,undefined behind tests/cpp/archive_fuzz.cpp: every truncation and 10^4
  seeded mutations of a valid text and a valid binary image, each in an exact-size heap block;
* the CPU oracle built with -fsanitize=address,undefined (make -C oracle asan), with the oracle's own test files
  run on it in a child process that has the sanitizer runtime preloaded."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN = ["-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer"]


def _have_sanitizers():
    if shutil.which("g++") is None:
        return False
    r = subprocess.run(["g++", "-print-file-name=libasan.so"], capture_output=True, text=True)
    return os.path.isabs(r.stdout.strip())


pytestmark = pytest.mark.skipif(not _have_sanitizers(), reason="g++ with libasan is not on this machine")


def test_archive_parser_under_asan_and_ubsan(tmp_path):
    exe = str(tmp_path / "archive_fuzz")
    subprocess.check_call(["g++", *SAN, "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-x", "c++",
                           os.path.join(ROOT, "orb_slam3_amd/csrc/orbx_archive.hip"),
                           os.path.join(ROOT, "tests/cpp/archive_fuzz.cpp"), "-o", exe], cwd=ROOT)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    r = subprocess.run([exe, "10000"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-4000:]
    assert "archive_fuzz ok" in r.stdout and "runtime error" not in r.stderr


def test_oracle_tests_pass_on_the_sanitizer_build():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "asan"])
    libasan = subprocess.run(["g++", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    preload = ":".join(p for p in (libasan, os.environ.get("LD_PRELOAD", "")) if p)
    env = dict(os.environ, LD_PRELOAD=preload, ORB_ORACLE_BUILD="asan",
               ASAN_OPTIONS="detect_leaks=0:verify_asan_link_order=0:abort_on_error=1",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    files = ["tests/test_oracle_known_answers.py", "tests/test_oracle_matchers.py", "tests/test_oracle_cameras.py",
             "tests/test_oracle_golden.py", "tests/test_oracle_bow.py", "tests/test_oracle_stage.py", "tests/test_introsort.py"]
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", *files], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, (r.stdout[-3000:] + r.stderr[-3000:])
    assert "passed" in r.stdout and "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr
