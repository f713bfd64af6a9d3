"""CPU oracle (test infrastructure only) -- ctypes view of oracle/libozcs_oracle.so.

Only tests, the smoke check and the benchmark's cpu_baseline leg may import this.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libozcs_oracle.so")

NONE, CRC32, CRC32C, SHA256, MD5 = 1, 2, 3, 4, 5
DIGEST_SIZE = {NONE: 0, CRC32: 4, CRC32C: 4, SHA256: 32, MD5: 16}

VERIFY_OK, VERIFY_MISMATCH, VERIFY_ORIG_EMPTY, VERIFY_COMPUTED_EMPTY, VERIFY_OUT_OF_RANGE = 0, 1, 2, 3, 4


def build(force=False):
    src = [os.path.join(_HERE, f) for f in ("ozcs_oracle.c", "ozcs_oracle.h")]
    if force or not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "libozcs_oracle.so"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = ctypes.CDLL(_SO)
        u8p = ctypes.c_char_p
        L.oracle_crc_make_table.argtypes = [ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]
        L.oracle_crc_make_table.restype = None
        L.oracle_crc_update.argtypes = [ctypes.c_uint32, u8p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_uint32)]
        L.oracle_crc_update.restype = ctypes.c_uint32
        for f in (L.oracle_crc32, L.oracle_crc32c):
            f.argtypes = [u8p, ctypes.c_size_t]
            f.restype = ctypes.c_uint32
        for f in (L.oracle_sha256, L.oracle_md5):
            f.argtypes = [u8p, ctypes.c_size_t, ctypes.c_void_p]
            f.restype = None
        L.oracle_compute.argtypes = [ctypes.c_int, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_uint64,
                                     ctypes.c_void_p, ctypes.c_uint64]
        L.oracle_compute.restype = ctypes.c_int64
        L.oracle_verify.argtypes = [ctypes.c_int, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_uint64,
                                    ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64,
                                    ctypes.POINTER(ctypes.c_int64)]
        L.oracle_verify.restype = ctypes.c_int
        L.oracle_fill_splitmix64.argtypes = [ctypes.c_uint64, ctypes.c_void_p, ctypes.c_uint64]
        L.oracle_fill_splitmix64.restype = None
        L.oracle_bench.argtypes = [ctypes.c_int, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_uint64,
                                   ctypes.c_int, ctypes.c_int]
        L.oracle_bench.restype = ctypes.c_double
        L.oracle_bench_clmul.argtypes = L.oracle_bench.argtypes
        L.oracle_bench_clmul.restype = ctypes.c_double
        L.oracle_has_clmul.argtypes = []
        L.oracle_has_clmul.restype = ctypes.c_int
        L.oracle_compute_clmul.argtypes = [ctypes.c_int, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_uint64,
                                           ctypes.c_void_p, ctypes.c_uint64]
        L.oracle_compute_clmul.restype = ctypes.c_int64
        _lib = L
    return _lib


def _buf(data):
    """bytes / bytearray / numpy uint8 array -> (address, length, keepalive)."""
    import numpy as np
    a = np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else np.ascontiguousarray(data).view(np.uint8).reshape(-1)
    return a.ctypes.data, a.size, a


def crc_table(poly):
    t = (ctypes.c_uint32 * 2048)()
    lib().oracle_crc_make_table(poly, t)
    return list(t)


def crc32(data):
    return lib().oracle_crc32(bytes(data), len(data))


def crc32c(data):
    return lib().oracle_crc32c(bytes(data), len(data))


def sha256(data):
    out = ctypes.create_string_buffer(32)
    lib().oracle_sha256(bytes(data), len(data), out)
    return out.raw


def md5(data):
    out = ctypes.create_string_buffer(16)
    lib().oracle_md5(bytes(data), len(data), out)
    return out.raw


def splitmix64(seed, n):
    import numpy as np
    out = np.empty(n, dtype=np.uint8)
    if n:
        lib().oracle_fill_splitmix64(seed, out.ctypes.data, n)
    return out


def compute(type_, bpc, data):
    """Checksum.computeChecksum over a contiguous buffer -> concatenated digest bytes."""
    addr, n, keep = _buf(data)
    ds = DIGEST_SIZE.get(type_, 0)
    nwin = (n + bpc - 1) // bpc if bpc else 0
    out = ctypes.create_string_buffer(max(1, nwin * ds))
    rc = lib().oracle_compute(type_, bpc, addr, n, out, nwin * ds)
    if rc < 0:
        raise ValueError("oracle_compute rejected type=%r bpc=%r" % (type_, bpc))
    return out.raw[: rc * ds]


def verify(type_, bpc, data, stored, start_index=0):
    """-> (rc, first_mismatch) as ChecksumData.verifyChecksumDataMatches reports them."""
    addr, n, keep = _buf(data)
    ds = DIGEST_SIZE.get(type_, 0)
    fm = ctypes.c_int64(-1)
    sb = bytes(stored)
    rc = lib().oracle_verify(type_, bpc, addr, n, sb, (len(sb) // ds) if ds else 0, start_index, ctypes.byref(fm))
    return rc, fm.value


def bench(type_, bpc, data, threads=1, reps=3):
    addr, n, keep = _buf(data)
    return lib().oracle_bench(type_, bpc, addr, n, threads, reps)


def has_clmul():
    """True when the CPU-baseline CRC runs on PCLMULQDQ (else it falls back to the table method)."""
    return bool(lib().oracle_has_clmul())


def compute_clmul(type_, bpc, data):
    """compute() for CRC32 / CRC32C through the carry-less-multiply folding of crc_clmul.c (the CPU baseline)."""
    addr, n, keep = _buf(data)
    nwin = (n + bpc - 1) // bpc if bpc else 0
    out = ctypes.create_string_buffer(max(1, nwin * 4))
    rc = lib().oracle_compute_clmul(type_, bpc, addr, n, out, nwin * 4)
    if rc < 0:
        raise ValueError("oracle_compute_clmul rejected type=%r bpc=%r" % (type_, bpc))
    return out.raw[: rc * 4]


def bench_clmul(type_, bpc, data, threads=1, reps=3):
    addr, n, keep = _buf(data)
    return lib().oracle_bench_clmul(type_, bpc, addr, n, threads, reps)
