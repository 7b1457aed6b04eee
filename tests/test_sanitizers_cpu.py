"""SURVEY §5 row 2 / VERDICT r2: an ASan + UBSan build of the host-side code — the product's OBJ loader (bvh_amd/csrc/obj.cpp) and
the C oracle — run through the host-side suites (tests/san_driver.py, in a subprocess with libasan preloaded)."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_code_under_asan_ubsan():
    if not shutil.which("gcc") or not shutil.which("g++"):
        pytest.skip("no host compiler")
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(asan) or not os.path.exists(asan):
        pytest.skip("this gcc ships no libasan")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "asan"], stdout=subprocess.DEVNULL)
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:abort_on_error=1",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1", BVH_AMD_NO_TORCH="1", OMP_NUM_THREADS="4")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "san_driver.py")], env=env, cwd=ROOT, capture_output=True, text=True,
                       timeout=1500)
    assert p.returncode == 0, (p.stdout[-3000:], p.stderr[-6000:])
    assert "obj loader under ASan/UBSan: ok" in p.stdout and "oracle under ASan/UBSan: ok" in p.stdout
    assert "ERROR: AddressSanitizer" not in p.stderr and "runtime error:" not in p.stderr
