"""The CPU side under AddressSanitizer + UndefinedBehaviorSanitizer (the sanitised build).

`oracle/` is ~5 k lines of malloc-heavy C and the only checker of every bit-exact claim, so a stray read in it could make a
wrong device result look right.  `make -C oracle asan` builds the same sources with -fsanitize=address,undefined (no OpenMP);
this test runs the oracle's own test modules against that library in a child process with libasan preloaded and requires a
clean exit and no sanitizer report.  (GPU AddressSanitizer is not available on this pool: the host half of libaps_hip.so is not
covered here.)"""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MODULES = ["test_match_oracle.py", "test_ransac_oracle.py", "test_ransac_types_oracle.py", "test_mlesac_types_oracle.py",
           "test_render_oracle.py", "test_sift_oracle.py", "test_crop_oracle.py", "test_bounds_oracle.py", "test_ba_oracle.py",
           "test_golden.py", "test_oracle_crosscheck.py", "test_sift_independent.py", "test_preprocess.py"]


def _runtime(name):
    p = subprocess.run(["gcc", "-print-file-name=" + name], capture_output=True, text=True).stdout.strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


@pytest.mark.skipif(shutil.which("gcc") is None or shutil.which("make") is None, reason="needs gcc and make")
def test_oracle_test_modules_pass_under_asan_and_ubsan():
    asan = _runtime("libasan.so")
    if asan is None:
        pytest.skip("this gcc has no libasan")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "asan"])
    lib = os.path.join(ROOT, "oracle", "lib", "libaps_oracle_asan.so")
    env = dict(os.environ)
    env.update(LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:halt_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1",
               APS_ORACLE_LIB=lib)
    env.pop("PYTEST_ADDOPTS", None)
    cmd = [sys.executable, "-m", "pytest", "-x", "-q", "-m", "not gpu", "-p", "no:cacheprovider", "-p", "no:xdist"] + \
          [os.path.join(ROOT, "tests", m) for m in MODULES]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    out = r.stdout + r.stderr
    assert "AddressSanitizer" not in out and "runtime error:" not in out, out[-4000:]
    assert r.returncode == 0, out[-4000:]
    assert " passed" in out and "libaps_oracle_asan.so" not in out  # (the library loaded: an OSError would name it)
