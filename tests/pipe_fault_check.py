"""Run as a child process with OZCS_FAULT_PIPE_DIGEST_LAUNCH=1: the first digest batch launch of the process is
refused.  Every ticket of that batch must report the failure (never stale digests as OK), CRC tickets and later
digest batches must be unaffected."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle  # noqa: E402
from hadoop_ozone_amd import native  # noqa: E402

MIB = 1 << 20


def main():
    assert os.environ.get("OZCS_FAULT_PIPE_DIGEST_LAUNCH") == "1"
    L = native.lib()
    native.check(L.ozcs_init(0))
    pipe = ctypes.c_void_p()
    native.check(L.ozcs_pipe_create(0, 8, MIB, ctypes.byref(pipe)))
    n = ctypes.c_uint64()
    # warm every slot's digest buffer with real MD5-sized content through CRC work first, so that "stale digests
    # handed over as OK" would be visible as non-sentinel bytes
    jobs = []
    for i in range(6):
        t = oracle.CRC32 if i in (1, 4) else (oracle.MD5 if i % 2 == 0 else oracle.SHA256)
        d = oracle.splitmix64(40 + i, MIB - i)
        out = ctypes.create_string_buffer(b"\xEE" * (16 * native.DIGEST_SIZE[t]), 16 * native.DIGEST_SIZE[t])
        ticket = ctypes.c_int(-1)
        native.check(L.ozcs_pipe_submit(pipe, t, 65536, d.ctypes.data, d.size, out, len(out), ctypes.byref(ticket)))
        jobs.append((t, d, out, ticket.value))
    failed = 0
    for t, d, out, ticket in jobs:
        rc = L.ozcs_pipe_wait(pipe, ticket, ctypes.byref(n))
        exp = oracle.compute(t, 65536, d)
        if rc == native.OK:
            assert out.raw[:len(exp)] == exp, "ticket %d: OK but wrong digests" % ticket
        else:
            assert rc == native.EHIP, rc
            assert out.raw == b"\xEE" * len(out.raw), "a failed ticket must not deliver digests"
            assert t != oracle.CRC32
            failed += 1
    assert failed >= 1, "the injected launch failure was not reported through any ticket"
    # the pipe keeps working: the next digest batch is launched normally
    d = oracle.splitmix64(99, MIB)
    out = ctypes.create_string_buffer(16 * 16)
    ticket = ctypes.c_int(-1)
    native.check(L.ozcs_pipe_submit(pipe, oracle.MD5, 65536, d.ctypes.data, d.size, out, len(out), ctypes.byref(ticket)))
    native.check(L.ozcs_pipe_wait(pipe, ticket.value, ctypes.byref(n)))
    assert n.value == 16 and out.raw == oracle.compute(oracle.MD5, 65536, d)
    native.check(L.ozcs_pipe_drain(pipe))
    L.ozcs_pipe_destroy(pipe)
    print("pipe fault check ok: %d ticket(s) reported the refused launch" % failed)


if __name__ == "__main__":
    main()
