"""ctypes binding of libozcs.so -- exactly the entry points include/ozone_checksum.h declares.

There is no fallback: if the library is missing or no gfx950 GPU is usable, calls raise.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.environ.get("OZCS_LIB") or os.path.join(CSRC, "libozcs.so")

NONE, CRC32, CRC32C, SHA256, MD5 = 1, 2, 3, 4, 5
OK, EINVAL, ENODEV, EHIP, ENOMEM, ERANGE, EEMPTY, EBUSY = 0, -1, -2, -3, -4, -5, -6, -7
NOALLOC_MAX_CHUNKS = 65536
DIGEST_SIZE = {NONE: 0, CRC32: 4, CRC32C: 4, SHA256: 32, MD5: 16}


class OzcsChunk(ctypes.Structure):
    _fields_ = [("data", ctypes.c_void_p), ("len", ctypes.c_uint64), ("sums", ctypes.c_void_p),
                ("sums_cap", ctypes.c_uint64), ("bpc", ctypes.c_uint32), ("type", ctypes.c_int32),
                ("n_sums", ctypes.c_uint64), ("status", ctypes.c_int32), ("reserved", ctypes.c_int32)]


class OzcsVChunk(ctypes.Structure):
    _fields_ = [("data", ctypes.c_void_p), ("len", ctypes.c_uint64), ("expected", ctypes.c_void_p),
                ("n_expected", ctypes.c_uint64), ("start_index", ctypes.c_uint64), ("bpc", ctypes.c_uint32),
                ("type", ctypes.c_int32), ("first_mismatch", ctypes.c_int64), ("status", ctypes.c_int32),
                ("reserved", ctypes.c_int32)]


class OzcsStats(ctypes.Structure):
    _fields_ = [(n, ctypes.c_uint64) for n in
                ("calls", "bytes_hashed", "windows", "h2d_bytes", "d2h_bytes", "kernel_launches",
                 # OZCS_ABI_VERSION 2
                 "errors", "unavailable", "timed_calls", "timed_bytes", "h2d_us", "kernel_us", "d2h_us")]


# name -> (restype, argtypes); kept in step with include/ozone_checksum.h (tests check both ways)
SIGNATURES = {
    "ozcs_init": (ctypes.c_int, [ctypes.c_uint32]),
    "ozcs_shutdown": (None, []),
    "ozcs_available": (ctypes.c_int, []),
    "ozcs_device_count": (ctypes.c_int, []),
    "ozcs_abi_version": (ctypes.c_int, []),
    "ozcs_last_error": (ctypes.c_char_p, []),
    "ozcs_digest_size": (ctypes.c_size_t, [ctypes.c_int]),
    "ozcs_num_checksums": (ctypes.c_uint64, [ctypes.c_uint64, ctypes.c_uint32]),
    "ozcs_compute": (ctypes.c_int, [ctypes.c_int, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_uint64,
                                    ctypes.c_void_p, ctypes.c_uint64, ctypes.POINTER(ctypes.c_uint64)]),
    "ozcs_verify": (ctypes.c_int, [ctypes.c_int, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_uint64,
                                   ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64,
                                   ctypes.POINTER(ctypes.c_int64)]),
    "ozcs_compute_batch": (ctypes.c_int, [ctypes.POINTER(OzcsChunk), ctypes.c_uint32]),
    "ozcs_shard_chunks": (None, [ctypes.POINTER(OzcsChunk), ctypes.c_uint32, ctypes.c_uint32,
                                 ctypes.POINTER(ctypes.c_uint32)]),
    "ozcs_verify_batch": (ctypes.c_int, [ctypes.POINTER(OzcsVChunk), ctypes.c_uint32]),
    "ozcs_compute_device": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_uint32, ctypes.c_void_p,
                                           ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_void_p,
                                           ctypes.c_uint64, ctypes.c_void_p]),
    "ozcs_verify_device": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_uint32, ctypes.c_void_p,
                                          ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_void_p,
                                          ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_void_p,
                                          ctypes.c_void_p]),
    "ozcs_verify_device_noalloc": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_uint32, ctypes.c_void_p,
                                                  ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_void_p,
                                                  ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_void_p,
                                                  ctypes.c_void_p]),
    "ozcs_get_stats": (None, [ctypes.POINTER(OzcsStats)]),
    "ozcs_pipe_create": (ctypes.c_int, [ctypes.c_int, ctypes.c_uint32, ctypes.c_uint64, ctypes.POINTER(ctypes.c_void_p)]),
    "ozcs_pipe_destroy": (None, [ctypes.c_void_p]),
    "ozcs_pipe_slot_bytes": (ctypes.c_uint64, [ctypes.c_void_p]),
    "ozcs_pipe_sums_bytes": (ctypes.c_uint64, [ctypes.c_void_p]),
    "ozcs_pipe_acquire": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_int)]),
    "ozcs_pipe_acquire_wait": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_int)]),
    "ozcs_pipe_release": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    "ozcs_pipe_set_in_order": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    "ozcs_pipe_next_in_order": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_int),
                                               ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_uint64)]),
    "ozcs_pipe_submit_slot": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_uint32,
                                             ctypes.c_uint64, ctypes.c_void_p, ctypes.c_uint64]),
    "ozcs_pipe_submit": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_uint64,
                                        ctypes.c_void_p, ctypes.c_uint64, ctypes.POINTER(ctypes.c_int)]),
    "ozcs_pipe_poll": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_uint64)]),
    "ozcs_pipe_wait": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_uint64)]),
    "ozcs_pipe_drain": (ctypes.c_int, [ctypes.c_void_p]),
    "ozcs_pipe_ticket_slot": (ctypes.c_int, [ctypes.c_int]),
    "ozcs_pipe_next_completed": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_int),
                                                ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_uint64)]),
}


def build(force=False):
    """Compiles csrc/ into libozcs.so for gfx950 (hipcc cross-compiles without a GPU)."""
    if force:
        subprocess.check_call(["make", "-s", "-C", CSRC, "clean"])
    subprocess.check_call(["make", "-s", "-j8", "-C", CSRC, "libozcs.so"])
    return LIB_PATH


def source_hash():
    """SHA-256 over the library's sources (csrc/*.hip, csrc/*.h, the ABI header), the stamp that ties a
    committed profile to the kernels it was taken from."""
    import glob
    import hashlib
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.h")))
    files.append(os.path.join(os.path.dirname(_HERE), "include", "ozone_checksum.h"))
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


_lib = None


def lib():
    global _lib
    if _lib is None:
        # PyTorch's ROCm wheel bundles its own libamdhip64.so.7; whichever copy is loaded first serves
        # the whole process, and torch cannot see the GPU through a foreign one.  Let torch go first.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("%s is missing: run __graft_entry__.build() (no CPU fallback exists)" % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            f = getattr(L, name)
            f.restype = res
            f.argtypes = args
        _lib = L
    return _lib


class OzcsError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("ozcs error %d: %s" % (code, msg))
        self.code = code
        self.msg = msg


def check(rc):
    if rc != OK:
        raise OzcsError(rc, (lib().ozcs_last_error() or b"").decode("utf-8", "replace"))
    return rc


def _addr(buf):
    """bytes / bytearray / memoryview / numpy array -> (address, nbytes, keepalive)."""
    import numpy as np
    if isinstance(buf, np.ndarray):
        a = np.ascontiguousarray(buf)
        return a.ctypes.data, a.nbytes, a
    a = np.frombuffer(buf, dtype=np.uint8)
    return a.ctypes.data, a.nbytes, a


def compute(type_, bpc, data):
    """ozcs_compute over a host buffer -> concatenated digest bytes."""
    addr, n, keep = _addr(data)
    ds = DIGEST_SIZE.get(type_, 0)
    nwin = lib().ozcs_num_checksums(n, bpc)
    out = ctypes.create_string_buffer(max(1, nwin * ds))
    cnt = ctypes.c_uint64(0)
    check(lib().ozcs_compute(type_, bpc, addr, n, out, nwin * ds, ctypes.byref(cnt)))
    return out.raw[: cnt.value * ds]


def verify(type_, bpc, data, expected, start_index=0):
    """ozcs_verify -> (rc, first_mismatch); rc is OK / ERANGE / EEMPTY, other codes raise."""
    addr, n, keep = _addr(data)
    ds = DIGEST_SIZE.get(type_, 0)
    eb = bytes(expected)
    fm = ctypes.c_int64(-1)
    rc = lib().ozcs_verify(type_, bpc, addr, n, eb, (len(eb) // ds) if ds else 0, start_index, ctypes.byref(fm))
    if rc not in (OK, ERANGE, EEMPTY):
        check(rc)
    return rc, fm.value


def stats():
    s = OzcsStats()
    lib().ozcs_get_stats(ctypes.byref(s))
    return {n: getattr(s, n) for n, _ in OzcsStats._fields_}
