"""AddressSanitizer + UndefinedBehaviorSanitizer runs of the CPU side (the oracle, which every parity claim rests on):
the reference's known-answer tests again on a sanitizer build of the oracle, and a driver program that owns the heap
from its start and feeds every decoder damaged and truncated input from heap blocks of exactly the input's size."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN = os.path.join(ROOT, "oracle", "_san")


@pytest.fixture(scope="module")
def sanitizer_build():
    if shutil.which("gcc") is None:
        pytest.skip("no gcc here")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "sanitize"], stdout=subprocess.DEVNULL)
    return SAN


def test_oracle_driver_under_asan_and_ubsan(sanitizer_build):
    r = subprocess.run([os.path.join(sanitizer_build, "sanitize_driver"), "20261112", "300"], capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "no report" in r.stdout


def test_reference_kats_on_the_sanitizer_build_of_the_oracle(sanitizer_build):
    env = dict(os.environ)
    env["CVEC_ORACLE_LIB"] = os.path.join(sanitizer_build, "libcvec_oracle_san.so")
    # the interpreter is not instrumented, so the sanitizer runtime is not first in its library list
    env["ASAN_OPTIONS"] = "verify_asan_link_order=0:detect_leaks=0"
    env["UBSAN_OPTIONS"] = "print_stacktrace=1:halt_on_error=1"
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_oracle_kat.py"), "-x", "-q",
                        "-p", "no:cacheprovider"], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr
