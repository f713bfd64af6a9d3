"""The C-ABI library loads and exports exactly what include/ozone_checksum.h declares (no GPU needed)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT
from hadoop_ozone_amd import native

HEADER = os.path.join(ROOT, "include", "ozone_checksum.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"OZCS_API\s+[\w\s\*]+?\b(ozcs_\w+)\s*\(", text)))


def test_header_declares_something():
    syms = declared_symbols()
    assert "ozcs_compute" in syms and "ozcs_verify_device" in syms and len(syms) >= 14


def test_every_declared_symbol_is_exported_and_bound():
    L = native.lib()
    for name in declared_symbols():
        assert hasattr(L, name), "libozcs.so does not export %s" % name
        assert name in native.SIGNATURES, "native.py does not bind %s" % name
    for name in native.SIGNATURES:
        assert name in declared_symbols(), "%s is bound but not declared in the header" % name


def test_exports_only_the_abi():
    import subprocess
    out = subprocess.check_output(["nm", "-D", "--defined-only", native.LIB_PATH], text=True)
    exported = sorted(l.split()[-1] for l in out.splitlines() if " T " in l)
    assert exported == declared_symbols()


def test_pure_queries():
    L = native.lib()
    assert L.ozcs_abi_version() == 2
    assert [L.ozcs_digest_size(t) for t in (0, 1, 2, 3, 4, 5, 6)] == [0, 0, 4, 4, 32, 16, 0]
    assert L.ozcs_num_checksums(55, 10) == 6          # TestChecksum.java:47-62
    assert L.ozcs_num_checksums(100, 20) == 5
    assert L.ozcs_num_checksums(3072, 2048) == 2      # TestKeyValueContainerCheck.java:211-217
    assert L.ozcs_num_checksums(0, 10) == 0
    assert L.ozcs_num_checksums(10, 0) == 0
    assert L.ozcs_num_checksums((1 << 40) + 1, 1 << 20) == (1 << 20) + 1


def test_argument_errors_need_no_device():
    L = native.lib()
    out = ctypes.create_string_buffer(64)
    n = ctypes.c_uint64(7)
    assert L.ozcs_compute(9, 16, b"abc", 3, out, 64, ctypes.byref(n)) == native.EINVAL
    assert b"Unrecognized ChecksumType" in L.ozcs_last_error()   # OzoneChecksumException.java:43-47
    assert n.value == 0
    assert L.ozcs_compute(native.CRC32, 0, b"abc", 3, out, 64, ctypes.byref(n)) == native.EINVAL
    # NONE and empty input produce no checksums and never touch a device (Checksum.java:157-160)
    assert L.ozcs_compute(native.NONE, 16, b"abc", 3, out, 64, ctypes.byref(n)) == native.OK and n.value == 0
    assert L.ozcs_compute(native.CRC32, 16, b"", 0, out, 64, ctypes.byref(n)) == native.OK and n.value == 0
    fm = ctypes.c_int64(5)
    assert L.ozcs_verify(native.NONE, 16, b"abc", 3, b"", 0, 0, ctypes.byref(fm)) == native.OK and fm.value == -1
    assert L.ozcs_verify(native.CRC32, 16, b"abc", 3, b"", 0, 0, ctypes.byref(fm)) == native.EEMPTY
    assert b"Original checksumData has no checksums" in L.ozcs_last_error()
    assert L.ozcs_verify(native.CRC32, 16, b"", 0, b"abcd", 1, 0, ctypes.byref(fm)) == native.EEMPTY
    assert b"Computed checksumData has no checksums" in L.ozcs_last_error()
    assert L.ozcs_compute(native.CRC32, 16, b"abc", 3, out, 3, ctypes.byref(n)) == native.EINVAL  # short sums_cap


def test_no_gpu_fails_loudly_instead_of_falling_back():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    L = native.lib()
    assert L.ozcs_available() == 0
    out = ctypes.create_string_buffer(64)
    n = ctypes.c_uint64(0)
    assert L.ozcs_compute(native.CRC32, 16, b"abc", 3, out, 64, ctypes.byref(n)) == native.ENODEV
    with pytest.raises(native.OzcsError):
        native.compute(native.CRC32, 16, b"abc")


def test_stats_count_errors_and_unavailable_calls():
    # ABI 2: every entry point's negative outcome is counted; ENODEV / EHIP / ENOMEM also as "unavailable" (the calls a
    # binding answers with its own CPU path).  OZCS_EBUSY is flow control and not counted.
    import torch
    L = native.lib()
    before = native.stats()
    out = ctypes.create_string_buffer(64)
    n = ctypes.c_uint64(0)
    assert L.ozcs_compute(9, 16, b"abc", 3, out, 64, ctypes.byref(n)) == native.EINVAL
    assert L.ozcs_pipe_release(None, 0) == native.EINVAL
    fm = ctypes.c_int64(0)
    assert L.ozcs_verify(native.CRC32, 16, b"abc", 3, b"", 0, 0, ctypes.byref(fm)) == native.EEMPTY
    after = native.stats()
    assert after["errors"] - before["errors"] == 3
    assert after["unavailable"] == before["unavailable"]
    if not torch.cuda.is_available():
        assert L.ozcs_compute(native.CRC32, 16, b"abc", 3, out, 64, ctypes.byref(n)) == native.ENODEV
        last = native.stats()
        assert last["errors"] - after["errors"] == 1 and last["unavailable"] - after["unavailable"] == 1
        assert last["timed_calls"] == 0 and last["kernel_us"] == 0


def test_test_hooks_are_not_in_the_product_library():
    # the two environment hooks exist only in libozcs_testhooks.so (-DOZCS_TEST_HOOKS)
    product = open(native.LIB_PATH, "rb").read()
    hooks = open(os.path.join(native.CSRC, "libozcs_testhooks.so"), "rb").read()
    for name in (b"OZCS_TEST_REPLICATE_DEVICES", b"OZCS_FAULT_PIPE_DIGEST_LAUNCH"):
        assert name not in product and name in hooks
