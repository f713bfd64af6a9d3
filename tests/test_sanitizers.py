"""The host code that needs no GPU, under AddressSanitizer and UndefinedBehaviorSanitizer (CPU only; nothing here
runs on the GPU box's card): the oracle with its self-test as the driver (exact-size heap buffers around every
internal boundary of the CRC, SHA-256 and MD5 restatements and of the carry-less-multiply engine), the ChunkBuffer
mirror's test and the GF(2) operator test.  A sanitizer report makes the binary exit non-zero."""
import os
import shutil
import subprocess

import pytest

from conftest import ROOT

CASES = [
    (os.path.join(ROOT, "oracle"), "oracle_selftest_asan", "oracle selftest ok"),
    (os.path.join(ROOT, "tests", "cpp"), "chunk_buffer_test_asan", None),
    (os.path.join(ROOT, "tests", "cpp"), "gf2_test_asan", None),
]


@pytest.mark.parametrize("directory,binary,marker", CASES, ids=[c[1] for c in CASES])
def test_clean_under_asan_and_ubsan(directory, binary, marker):
    if not shutil.which("gcc") or not shutil.which("g++"):
        pytest.skip("no host compiler")
    subprocess.check_call(["make", "-s", "-C", directory, binary])
    env = dict(os.environ)
    env["ASAN_OPTIONS"] = "detect_leaks=1:abort_on_error=0:halt_on_error=1"
    env["UBSAN_OPTIONS"] = "print_stacktrace=1:halt_on_error=1"
    r = subprocess.run([os.path.join(directory, binary)], capture_output=True, text=True, timeout=300, env=env)
    out = r.stdout + r.stderr
    assert r.returncode == 0, out[-4000:]
    assert "ERROR: AddressSanitizer" not in out and "runtime error:" not in out, out[-4000:]
    if marker:
        assert marker in out
