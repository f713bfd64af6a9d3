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


def in_order(L):
    """The same refusal on an in-order pipe: the failed batch's outcomes come out at their turn with the error, the
    slots are back in the pool, everything before and after is delivered intact and in submission order."""
    pipe = ctypes.c_void_p()
    native.check(L.ozcs_pipe_create(0, 8, MIB, ctypes.byref(pipe)))
    native.check(L.ozcs_pipe_set_in_order(pipe, 1))
    jobs = []
    buf, tk = ctypes.c_void_p(), ctypes.c_int(-1)
    for i in range(8):
        t = oracle.CRC32C if i in (0, 3, 7) else (oracle.MD5 if i % 2 == 0 else oracle.SHA256)
        d = oracle.splitmix64(140 + i, MIB - 3 * i)
        out = ctypes.create_string_buffer(b"\xEE" * (16 * native.DIGEST_SIZE[t]), 16 * native.DIGEST_SIZE[t])
        native.check(L.ozcs_pipe_acquire_wait(pipe, ctypes.byref(buf), ctypes.byref(tk)))
        ctypes.memmove(buf.value, d.ctypes.data, d.size)
        native.check(L.ozcs_pipe_submit_slot(pipe, tk.value, t, 65536, d.size, out, len(out)))
        jobs.append((t, d, out, tk.value))
    failed = 0
    done, status, n = ctypes.c_int(-1), ctypes.c_int(0), ctypes.c_uint64()
    for t, d, out, ticket in jobs:
        native.check(L.ozcs_pipe_next_in_order(pipe, 1, ctypes.byref(done), ctypes.byref(status), ctypes.byref(n)))
        assert done.value == ticket, "out of order"
        exp = oracle.compute(t, 65536, d)
        if status.value == native.OK:
            assert out.raw[:len(exp)] == exp
        else:
            assert status.value == native.EHIP and t != oracle.CRC32C
            assert out.raw == b"\xEE" * len(out.raw), "a failed ticket must not deliver digests"
            failed += 1
    assert failed >= 1
    assert L.ozcs_pipe_next_in_order(pipe, 0, ctypes.byref(done), None, None) == native.EEMPTY
    # every slot is back: eight acquisitions succeed at once
    held = []
    for _ in range(8):
        native.check(L.ozcs_pipe_acquire(pipe, ctypes.byref(buf), ctypes.byref(tk)))
        held.append(tk.value)
    for h in held:
        native.check(L.ozcs_pipe_release(pipe, h))
    L.ozcs_pipe_destroy(pipe)
    print("pipe fault check ok (in-order pipe): %d ticket(s) reported the refused launch" % failed)


def main():
    assert os.environ.get("OZCS_FAULT_PIPE_DIGEST_LAUNCH") == "1"
    L = native.lib()
    native.check(L.ozcs_init(0))
    if sys.argv[1:] == ["in-order"]:
        return in_order(L)
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
