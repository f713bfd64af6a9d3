"""The library's HOST code under AddressSanitizer + UndefinedBehaviorSanitizer while the real GPU path runs.

tests/cpp/host_asan_driver.cpp drives the C ABI from several threads (lone calls, batches of every layout class, verify,
the pipe per ticket beside a draining thread and in order, shutdown and re-init) and compares every digest with the
oracle; tests/cpp/Makefile builds it plain and against a copy of the library whose host side is instrumented (the device
code is not: GPU sanitizers are not available here).  Both are built by __graft_entry__.build()."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(name, env=None):
    exe = os.path.join(ROOT, "tests", "cpp", name)
    assert os.path.exists(exe), "run __graft_entry__.build() first"
    return subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, env=env)


@pytest.mark.gpu
def test_c_abi_driver_plain():
    r = run("host_driver")
    assert r.returncode == 0 and b"host driver ok" in r.stdout, r.stderr[-2000:]


@pytest.mark.gpu
def test_c_abi_driver_with_host_code_under_asan_and_ubsan():
    # leak detection off: the HIP runtime keeps its own allocations until the process ends
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0")
    r = run("host_driver_asan", env)
    err = r.stderr.decode("utf-8", "replace")
    assert "AddressSanitizer" not in err and "runtime error:" not in err, err[-4000:]
    assert r.returncode == 0 and b"host driver ok" in r.stdout, err[-2000:]
