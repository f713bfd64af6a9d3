"""Run as a child process with OZCS_TEST_REPLICATE_DEVICES=3: the one GPU is listed three times, so ozcs_compute_batch
and ozcs_verify_batch shard their chunk lists over three "devices" with one host thread each.  Digests are checked
against the CPU oracle; a failing chunk's message must reach the calling thread."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import oracle  # noqa: E402
from hadoop_ozone_amd import native  # noqa: E402


def main():
    assert os.environ.get("OZCS_TEST_REPLICATE_DEVICES") == "3"
    L = native.lib()
    native.check(L.ozcs_init(0))
    assert L.ozcs_device_count() == 3 * max(1, L.ozcs_device_count() // 3)
    rng = np.random.default_rng(5)
    types = [oracle.CRC32, oracle.CRC32C, oracle.SHA256, oracle.MD5]
    n = 90
    arr = (native.OzcsChunk * n)()
    keep = []
    for i in range(n):
        t = types[i % 4]
        ln = int(rng.choice([0, 1, 777, 65536, 300001, 1 << 20, (4 << 20) + 5, 9 << 20]))
        bpc = int(rng.choice([512, 16384, 1 << 20]))
        d = oracle.splitmix64(100 + i, ln)
        ds = native.DIGEST_SIZE[t]
        out = ctypes.create_string_buffer(max(1, ((ln + bpc - 1) // bpc) * ds))
        keep.append((t, bpc, d, out))
        arr[i].data = d.ctypes.data if ln else None
        arr[i].len = ln
        arr[i].sums = ctypes.cast(out, ctypes.c_void_p).value
        arr[i].sums_cap = len(out)
        arr[i].bpc = bpc
        arr[i].type = t
    cuts = (ctypes.c_uint32 * 4)()
    L.ozcs_shard_chunks(arr, n, 3, cuts)
    assert 0 < cuts[1] < cuts[2] < n   # three non-empty shards: three worker threads
    native.check(L.ozcs_compute_batch(arr, n))
    for i, (t, bpc, d, out) in enumerate(keep):
        exp = oracle.compute(t, bpc, d)
        assert arr[i].status == native.OK and arr[i].n_sums * native.DIGEST_SIZE[t] == len(exp), i
        assert out.raw[:len(exp)] == exp, i
    # a bad chunk in the last shard (served by a worker thread): its status and message come back to this thread
    arr[n - 1].type = 42
    rc = L.ozcs_compute_batch(arr, n)
    assert rc == native.EINVAL and arr[n - 1].status == native.EINVAL and arr[0].status == native.OK
    assert b"Unrecognized ChecksumType: 42" in L.ozcs_last_error(), L.ozcs_last_error()
    arr[n - 1].type = keep[n - 1][0]
    # verify over the same three shards
    v = (native.OzcsVChunk * n)()
    stored = []
    for i, (t, bpc, d, out) in enumerate(keep):
        sb = ctypes.create_string_buffer(out.raw, len(out.raw))
        stored.append(sb)
        dd = d.copy()
        if i % 7 == 3 and d.size:
            dd[d.size // 2] ^= 1
        keep[i] = (t, bpc, dd, out)
        v[i].data = dd.ctypes.data if dd.size else None
        v[i].len = dd.size
        v[i].expected = ctypes.cast(sb, ctypes.c_void_p).value
        v[i].n_expected = (dd.size + bpc - 1) // bpc
        v[i].start_index = 0
        v[i].bpc = bpc
        v[i].type = t
    L.ozcs_verify_batch(v, n)
    for i, (t, bpc, dd, out) in enumerate(keep):
        if dd.size == 0:
            assert v[i].status == native.EEMPTY
            continue
        want = (dd.size // 2) // bpc if i % 7 == 3 else -1
        assert (v[i].status, v[i].first_mismatch) == (native.OK, want), (i, v[i].status, v[i].first_mismatch, want)
    # single calls from this thread stay on one device slot; a pipe on the replicated device works
    assert native.compute(oracle.CRC32, 1 << 20, keep[7][2]) == oracle.compute(oracle.CRC32, 1 << 20, keep[7][2])
    print("multi-device check ok: shards", list(cuts))


if __name__ == "__main__":
    main()
