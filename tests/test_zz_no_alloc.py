"""ozcs_verify_device_noalloc (the entry point that allocates nothing: its per-chunk verdict words live in a
file-scope device array of the library) against ozcs_verify_device on the same inputs, through the C ABI.

Bit-exact bar: the two-word summary must equal the first failing chunk / window of the per-chunk verdicts the
existing entry point writes, and both must equal what the CPU oracle says about the corrupted bytes.  Two
calls in a row on the same GPU, the second with a smaller input, so that verdict words left behind by the
first call would show.
"""
import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    import torch  # before the library: both then share one HIP runtime
    from hadoop_ozone_amd import native
    assert torch.cuda.is_available()
    L = native.lib()
    native.check(L.ozcs_init(0))
    torch.cuda.set_device(0)
    return torch, native, L


def per_chunk_verdicts(gpu, t, bpc, d, chunk_len, n_chunks, stride, d_exp, exp_stride, n_exp, start):
    torch, native, L = gpu
    fm = torch.full((max(1, n_chunks),), 7, dtype=torch.int64, device="cuda")
    native.check(L.ozcs_verify_device(0, t, bpc, d.data_ptr(), chunk_len, n_chunks, stride, d_exp.data_ptr(), exp_stride,
                                      n_exp, start, fm.data_ptr(), torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    return fm[:n_chunks].cpu().numpy()


def summary(gpu, t, bpc, d, chunk_len, n_chunks, stride, d_exp, exp_stride, n_exp, start):
    torch, native, L = gpu
    out = torch.full((2,), 99, dtype=torch.int64, device="cuda")
    native.check(L.ozcs_verify_device_noalloc(0, t, bpc, d.data_ptr(), chunk_len, n_chunks, stride, d_exp.data_ptr(),
                                              exp_stride, n_exp, start, out.data_ptr(),
                                              torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    return tuple(int(x) for x in out.cpu().numpy())


def reduce_verdicts(v):
    bad = np.nonzero(v != -1)[0]
    return (-1, -1) if len(bad) == 0 else (int(bad[0]), int(v[bad[0]]))


def case(gpu, t, bpc, chunk_len, n_chunks, flips, seed):
    """flips: (chunk, byte offset) pairs to corrupt after the stored digests were taken."""
    torch, native, L = gpu
    ds = native.DIGEST_SIZE[t]
    wpc = (chunk_len + bpc - 1) // bpc
    host = oracle.splitmix64(seed, n_chunks * chunk_len)
    stored = b"".join(oracle.compute(t, bpc, host[c * chunk_len:(c + 1) * chunk_len]) for c in range(n_chunks))
    bad = host.copy()
    for c, o in flips:
        bad[c * chunk_len + o] ^= 0x21
    d = torch.from_numpy(bad).cuda()
    d_exp = torch.from_numpy(np.frombuffer(stored, dtype=np.uint8).copy()).cuda()
    args = (t, bpc, d, chunk_len, n_chunks, chunk_len, d_exp, wpc * ds, wpc, 0)
    v = per_chunk_verdicts(gpu, *args)
    got = summary(gpu, *args)
    assert got == reduce_verdicts(v)
    # and what the oracle says about these bytes
    want = (-1, -1)
    for c in range(n_chunks):
        rc, first = oracle.verify(t, bpc, bad[c * chunk_len:(c + 1) * chunk_len], stored[c * wpc * ds:(c + 1) * wpc * ds], 0)
        assert rc in (0, 1)
        if first >= 0:
            want = (c, first)
            break
    assert got == want
    return got


def test_noalloc_matches_verify_device_two_calls_second_smaller(gpu):
    torch, native, L = gpu
    before = torch.cuda.memory_allocated()
    # first call: 300 chunks, failures in chunks 17 (window 2), 200 and 299
    assert case(gpu, oracle.CRC32C, 16384, 65536, 300, [(17, 2 * 16384 + 5), (200, 9), (299, 65535)], 11) == (17, 2)
    # second call, smaller: 10 clean chunks -- words 17.. of the array still hold the first call's verdicts
    assert case(gpu, oracle.CRC32C, 16384, 65536, 10, [], 12) == (-1, -1)
    # third, smaller still, one failure in the last window of the last chunk
    assert case(gpu, oracle.CRC32, 1 << 20, 4 << 20, 4, [(3, (4 << 20) - 1)], 13) == (3, 3)
    # digests too, ragged last window
    assert case(gpu, oracle.MD5, 4096, 10000, 6, [(4, 9000)], 14) == (4, 2)
    assert case(gpu, oracle.SHA256, 2048, 3072, 3, [], 15) == (-1, -1)
    assert torch.cuda.memory_allocated() >= before  # (the test's own tensors; the call itself allocates nothing)


def test_noalloc_start_index_and_short_stored_list(gpu):
    """An index past the stored list counts as failing, exactly as in ozcs_verify_device."""
    torch, native, L = gpu
    t, bpc, chunk_len, n_chunks = oracle.CRC32, 512, 4096, 5
    host = oracle.splitmix64(21, n_chunks * chunk_len)
    d = torch.from_numpy(host).cuda()
    # chunk c's stored list: 3 filler digests, then its 8 real ones; start_index 3 lines them up
    lists = []
    for c in range(n_chunks):
        lists.append(b"\xAA" * 12 + oracle.compute(t, bpc, host[c * chunk_len:(c + 1) * chunk_len]))
    d_exp = torch.from_numpy(np.frombuffer(b"".join(lists), dtype=np.uint8).copy()).cuda()
    for n_exp, start in ((11, 3), (9, 3), (11, 0)):
        args = (t, bpc, d, chunk_len, n_chunks, chunk_len, d_exp, 44, n_exp, start)
        assert summary(gpu, *args) == reduce_verdicts(per_chunk_verdicts(gpu, *args))
    assert summary(gpu, t, bpc, d, chunk_len, n_chunks, chunk_len, d_exp, 44, 11, 3) == (-1, -1)
    assert summary(gpu, t, bpc, d, chunk_len, n_chunks, chunk_len, d_exp, 44, 9, 3) == (0, 6)


def test_noalloc_limits_and_trivial_inputs(gpu):
    torch, native, L = gpu
    d = torch.zeros(4096, dtype=torch.uint8, device="cuda")
    out = torch.full((2,), 99, dtype=torch.int64, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    rc = L.ozcs_verify_device_noalloc(0, native.CRC32, 16, d.data_ptr(), 16, native.NOALLOC_MAX_CHUNKS + 1, 16,
                                      d.data_ptr(), 4, 1, 0, out.data_ptr(), s)
    assert rc == native.EINVAL and b"OZCS_NOALLOC_MAX_CHUNKS" in L.ozcs_last_error()
    assert L.ozcs_verify_device_noalloc(0, native.CRC32, 16, d.data_ptr(), 16, 1, 16, d.data_ptr(), 4, 1, 0, None, s) == native.EINVAL
    assert L.ozcs_verify_device_noalloc(0, 9, 16, d.data_ptr(), 16, 1, 16, d.data_ptr(), 4, 1, 0, out.data_ptr(), s) == native.EINVAL
    # NONE and empty inputs: nothing is hashed, the summary says "no failure"
    for t, chunk_len, n in ((native.NONE, 16, 4), (native.CRC32, 0, 4), (native.CRC32, 16, 0)):
        out.fill_(99)
        native.check(L.ozcs_verify_device_noalloc(0, t, 16, d.data_ptr(), chunk_len, n, 16, d.data_ptr(), 4, 1, 0,
                                                  out.data_ptr(), s))
        torch.cuda.synchronize()
        assert out.tolist() == [-1, -1]
    # the largest batch the array holds: 65536 chunks of 16 bytes, failure in the very last one
    n = native.NOALLOC_MAX_CHUNKS
    host = oracle.splitmix64(31, n * 16)
    stored = np.frombuffer(oracle.compute(native.CRC32, 16, host), dtype=np.uint8).copy()
    host[-1] ^= 1
    dd, de = torch.from_numpy(host).cuda(), torch.from_numpy(stored).cuda()
    native.check(L.ozcs_verify_device_noalloc(0, native.CRC32, 16, dd.data_ptr(), 16, n, 16, de.data_ptr(), 4, 1, 0,
                                              out.data_ptr(), s))
    torch.cuda.synchronize()
    assert out.tolist() == [n - 1, 0]


def test_noalloc_under_graph_capture(gpu):
    """What the entry point is for: it only enqueues kernels, so it can be captured into a HIP graph (where
    hipMalloc is not allowed) and replayed; every replay presets the verdict words it uses."""
    torch, native, L = gpu
    t, bpc, chunk_len, n_chunks = oracle.CRC32C, 16384, 131072, 40
    wpc = chunk_len // bpc
    host = oracle.splitmix64(77, n_chunks * chunk_len)
    stored = b"".join(oracle.compute(t, bpc, host[c * chunk_len:(c + 1) * chunk_len]) for c in range(n_chunks))
    d = torch.from_numpy(host.copy()).cuda()
    d_exp = torch.from_numpy(np.frombuffer(stored, dtype=np.uint8).copy()).cuda()
    out = torch.full((2,), 99, dtype=torch.int64, device="cuda")
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            native.check(L.ozcs_verify_device_noalloc(0, t, bpc, d.data_ptr(), chunk_len, n_chunks, chunk_len, d_exp.data_ptr(),
                                                      wpc * 4, wpc, 0, out.data_ptr(), torch.cuda.current_stream().cuda_stream))
    torch.cuda.current_stream().wait_stream(s)
    for flips, want in (([], [-1, -1]), ([(33, 5 * bpc + 1), (12, 7 * bpc)], [12, 7]), ([(39, chunk_len - 1)], [39, 7]), ([], [-1, -1])):
        bad = host.copy()
        for c, o in flips:
            bad[c * chunk_len + o] ^= 0x10
        d.copy_(torch.from_numpy(bad))
        out.fill_(99)
        g.replay()
        torch.cuda.synchronize()
        assert out.tolist() == want
