"""Parity of the HIP path with the CPU oracle, through the C ABI (bit-exact: all integer/byte work).

Small and medium inputs are compared digest by digest with the oracle and the committed golden vectors;
inputs at the benchmark's size are checked through size-independent properties (sampled windows against the
oracle, GF(2) linearity of CRC, invariance under chunk permutation, verify-after-compute round trip).
"""
import ctypes
import threading
import time

import numpy as np
import pytest

import oracle
from conftest import make_data

pytestmark = pytest.mark.gpu

MIB = 1 << 20
ALL = (oracle.CRC32, oracle.CRC32C, oracle.SHA256, oracle.MD5)


@pytest.fixture(scope="module")
def gpu():
    import torch  # before the library: both then share one HIP runtime
    from hadoop_ozone_amd import native
    assert torch.cuda.is_available()
    L = native.lib()
    native.check(L.ozcs_init(0))
    assert L.ozcs_available() == 1 and L.ozcs_device_count() >= 1
    torch.cuda.set_device(0)
    return torch, native, L


def dev_compute(gpu, t, bpc, dtensor, chunk_len, n_chunks, stride):
    torch, native, L = gpu
    ds = native.DIGEST_SIZE[t]
    wpc = (chunk_len + bpc - 1) // bpc
    sums = torch.full((max(1, wpc * n_chunks * ds),), 0xEE, dtype=torch.uint8, device=dtensor.device)
    native.check(L.ozcs_compute_device(0, t, bpc, dtensor.data_ptr(), chunk_len, n_chunks, stride, sums.data_ptr(),
                                       wpc * n_chunks * ds, torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    return bytes(sums[: wpc * n_chunks * ds].cpu().numpy())


def oracle_chunks(t, bpc, host, chunk_len, n_chunks, stride):
    return b"".join(oracle.compute(t, bpc, host[c * stride: c * stride + chunk_len]) for c in range(n_chunks))


# ------------------------------------------------------------------ golden vectors, host-pointer API

def test_kat(gpu, vectors):
    _, native, _ = gpu
    for rec in vectors["kat"]:
        got = native.compute(rec["type"], rec["bpc"], make_data(rec["data"]))
        assert got.hex() == "".join(rec["sums_hex"]), rec["name"]


def test_sha256_equals_the_literal_a_reference_test_holds(gpu, reference_held):
    # the one digest of known content that the reference's own tests hold for java.security.MessageDigest
    # (TestStringToSignProducer.java:86, computed there by StringToSignProducer.hash): through the host-pointer call,
    # as the ragged last window of a chunk, through the mirror of Checksum, and 300 times over in one device-resident
    # launch (full wavefronts of the one-lane-per-window kernel)
    torch, native, L = gpu
    from hadoop_ozone_amd import Checksum, ChecksumType
    for rec in reference_held:
        t = getattr(native, rec["type"])
        msg = np.frombuffer(rec["message_utf8"].encode("utf-8"), dtype=np.uint8)
        want = bytes.fromhex(rec["digest_hex"])
        assert native.compute(t, len(msg), msg) == want, rec["held_by"]
        assert native.compute(t, 1 << 20, msg) == want
        two = np.concatenate([np.zeros(256, dtype=np.uint8), msg])
        assert native.compute(t, 256, two)[32:] == want
        cd = Checksum(getattr(ChecksumType, rec["type"]), 4096).computeChecksum(msg.tobytes())
        assert cd.getChecksums() == [want]
        n, stride = 300, (len(msg) + 15) // 16 * 16 + 16
        host = np.zeros(n * stride, dtype=np.uint8)
        for c in range(n):
            host[c * stride: c * stride + len(msg)] = msg
        got = dev_compute(gpu, t, len(msg), torch.from_numpy(host).cuda(), len(msg), n, stride)
        assert got == want * n


def test_compute_golden(gpu, vectors):
    _, native, _ = gpu
    for rec in vectors["compute"]:
        got = native.compute(rec["type"], rec["bpc"], make_data(rec["data"]))
        assert got.hex() == "".join(rec["sums_hex"]), rec["name"]


def test_verify_golden(gpu, vectors):
    _, native, _ = gpu
    code = {0: native.OK, 1: native.OK, 2: native.EEMPTY, 3: native.EEMPTY, 4: native.ERANGE}
    for v in vectors["verify"]:
        sd = make_data(v["stored_data"])
        ds = native.DIGEST_SIZE[v["type"]]
        stored = oracle.compute(v["type"], v["bpc"], sd)
        if v["stored_count"] >= 0:
            stored = stored[: v["stored_count"] * ds]
        d = sd[v["data_off"]: v["data_off"] + v["data_len"]].copy()
        for o in v["flips"]:
            d[o] ^= 0xFF
        rc, first = native.verify(v["type"], v["bpc"], d, stored, v["start_index"])
        assert rc == code[v["rc"]], v["name"]
        assert first == v["first_mismatch"], v["name"]
        # and the oracle agrees with the fixture on the same inputs
        assert oracle.verify(v["type"], v["bpc"], d, stored, v["start_index"]) == (v["rc"], v["first_mismatch"])


# ------------------------------------------------------------------ device-resident API against the oracle

# shapes chosen so that the CRC launcher picks 16, 8, 4, 2 and 1 wavefronts per window in turn
CRC_SHAPES = [
    # (n_chunks, chunk_len, stride, bpc)
    (4, 4 * MIB, 4 * MIB, MIB),                    # 16 windows of 1 MiB: a whole workgroup per window
    (5, 4 * MIB + 4096 + 7, 4 * MIB + 8192, MIB),  # ragged last window of every chunk, padded stride
    (160, 256 * 1024, 256 * 1024, 65536),          # 640 windows
    (40, MIB, MIB, 32768),                         # 1280 windows
    (48, MIB, MIB, 16384),                         # 3072 windows of 16 KiB
    (16, MIB + 100, MIB + 128, 2048),              # 8208 windows, one wavefront each
    (3, 100000, 100000, 512),                      # HDFS-style 512-byte windows
    (2, 1000, 1000, 10),                           # TestChecksum-sized windows
    (1, 777, 777, 1),                              # one byte per checksum
    # >= 8192 windows: the small-window kernel (4, 8, 16, 32 lanes per window)
    (9, 512 * 1024 + 77, 512 * 1024 + 80, 512),
    (20, 450000, 450001, 1024),
    (33, 1000003, 1000003, 4096),
    (40, 4 * MIB, 4 * MIB, 16384),
    (3, 100000, 100000, 10),
    (7, 1500000, 1500007, 1000),
    (5, 9000001, 9000001, 4099),
    (2, 70 * MIB + 5, 70 * MIB + 16, 16400),
    # every window 16-byte aligned (the small-window kernel's fast path), ragged last windows
    (40, 512 * 1024 + 53, 512 * 1024 + 64, 512),
    (33, MIB + 48, MIB + 48, 1024),
    (20, 256 * 1024 + 17, 256 * 1024 + 32, 256),
    (300, 65536 + 3, 65536 + 16, 64),
    # full windows back to back (the gapless-layout kernel), window counts that leave the last wavefront ragged
    (37, 512 * 1024, 512 * 1024, 512),
    (1, 33 * MIB + 1024, 33 * MIB + 1024, 1024),
    (5, 5 * MIB, 5 * MIB, 4096),
    (64, 4 * MIB, 4 * MIB, 16384),
    (3, 4 * MIB, 4 * MIB, 256),
    (2, MIB, MIB, 128),
    (2, MIB, MIB, 64),
    (1, 3 * MIB, 3 * MIB, 48),      # not a whole number of stripes: the general aligned path
]


@pytest.mark.parametrize("shape", CRC_SHAPES, ids=lambda s: "%dx%d@%d" % (s[0], s[1], s[3]))
@pytest.mark.parametrize("t", [oracle.CRC32, oracle.CRC32C])
def test_crc_device_shapes(gpu, t, shape):
    torch, native, L = gpu
    n_chunks, chunk_len, stride, bpc = shape
    host = oracle.splitmix64(chunk_len * 31 + bpc + t, n_chunks * stride)
    d = torch.from_numpy(host).cuda()
    assert dev_compute(gpu, t, bpc, d, chunk_len, n_chunks, stride) == oracle_chunks(t, bpc, host, chunk_len, n_chunks, stride)


@pytest.mark.parametrize("t", [oracle.SHA256, oracle.MD5])
@pytest.mark.parametrize("shape", [(70, 3072, 3072, 2048), (3, 200000, 200064, 65536), (130, 1000, 1001, 55),
                                   (1, 64 * 300, 64 * 300, 64), (2, 4 * MIB, 4 * MIB, MIB)],
                         ids=lambda s: "%dx%d@%d" % (s[0], s[1], s[3]))
def test_digest_device_shapes(gpu, t, shape):
    torch, native, L = gpu
    n_chunks, chunk_len, stride, bpc = shape
    host = oracle.splitmix64(chunk_len + 7 * bpc + t, n_chunks * stride)
    d = torch.from_numpy(host).cuda()
    assert dev_compute(gpu, t, bpc, d, chunk_len, n_chunks, stride) == oracle_chunks(t, bpc, host, chunk_len, n_chunks, stride)


@pytest.mark.parametrize("t", ALL)
def test_misaligned_pointers_and_lengths(gpu, t):
    # window starts at every residue mod 16, lengths around the 4-byte / 16-byte / 64-byte boundaries
    torch, native, L = gpu
    host = oracle.splitmix64(99 + t, 300000)
    d = torch.from_numpy(host).cuda()
    ds = native.DIGEST_SIZE[t]
    for off in range(0, 17):
        for n, bpc in ((1, 1), (3, 3), (4, 4), (5, 4), (19, 19), (20, 20), (21, 7), (63, 64), (64, 64), (65, 64),
                       (1000 + off, 100), (70000 + off, 4099), (262144 + 3 * off, 65536 + off)):
            sub = d[off: off + n]
            got = dev_compute(gpu, t, bpc, sub, n, 1, n)
            assert got == oracle.compute(t, bpc, host[off: off + n]), (off, n, bpc)
            assert len(got) == ((n + bpc - 1) // bpc) * ds


def test_unaligned_sums_pointer(gpu):
    torch, native, L = gpu
    host = oracle.splitmix64(5, 64 * 1024)
    d = torch.from_numpy(host).cuda()
    for t in ALL:
        ds = native.DIGEST_SIZE[t]
        n = 64 * 1024 // 4096
        sums = torch.zeros(n * ds + 8, dtype=torch.uint8, device="cuda")
        for o in (1, 2, 3):
            native.check(L.ozcs_compute_device(0, t, 4096, d.data_ptr(), 64 * 1024, 1, 64 * 1024,
                                               sums.data_ptr() + o, n * ds, torch.cuda.current_stream().cuda_stream))
            torch.cuda.synchronize()
            assert bytes(sums[o: o + n * ds].cpu().numpy()) == oracle.compute(t, 4096, host)


def test_empty_none_and_capacity(gpu):
    torch, native, L = gpu
    d = torch.zeros(1024, dtype=torch.uint8, device="cuda")
    s = torch.zeros(64, dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    assert L.ozcs_compute_device(0, native.NONE, 16, d.data_ptr(), 1024, 1, 1024, s.data_ptr(), 0, st) == native.OK
    assert L.ozcs_compute_device(0, native.CRC32, 16, d.data_ptr(), 0, 1, 0, s.data_ptr(), 0, st) == native.OK
    assert L.ozcs_compute_device(0, native.CRC32, 16, d.data_ptr(), 1024, 1, 1024, s.data_ptr(), 64, st) == native.EINVAL
    assert L.ozcs_compute_device(0, 7, 16, d.data_ptr(), 1024, 1, 1024, s.data_ptr(), 64, st) == native.EINVAL
    assert L.ozcs_compute_device(5, native.CRC32, 512, d.data_ptr(), 1024, 1, 1024, s.data_ptr(), 64, st) == native.ENODEV
    assert native.compute(native.CRC32, 16, b"") == b""
    assert native.compute(native.NONE, 16, b"abcdef") == b""


# ------------------------------------------------------------------ verify on the device

@pytest.mark.parametrize("t,bpc", [(oracle.CRC32C, 16384), (oracle.CRC32, MIB), (oracle.SHA256, 65536), (oracle.MD5, 4096)])
def test_verify_device_first_mismatch_per_chunk(gpu, t, bpc):
    torch, native, L = gpu
    n_chunks, chunk_len = 12, 2 * MIB
    ds = native.DIGEST_SIZE[t]
    wpc = chunk_len // bpc
    host = oracle.splitmix64(4242 + t, n_chunks * chunk_len)
    d = torch.from_numpy(host).cuda()
    stored = torch.from_numpy(np.frombuffer(dev_compute(gpu, t, bpc, d, chunk_len, n_chunks, chunk_len), dtype=np.uint8).copy()).cuda()
    fm = torch.zeros(n_chunks, dtype=torch.int64, device="cuda")
    st = torch.cuda.current_stream().cuda_stream

    def run(start_index=0, n_expected=wpc):
        native.check(L.ozcs_verify_device(0, t, bpc, d.data_ptr(), chunk_len, n_chunks, chunk_len, stored.data_ptr(),
                                          wpc * ds, n_expected, start_index, fm.data_ptr(), st))
        torch.cuda.synchronize()
        return fm.cpu().tolist()

    assert run() == [-1] * n_chunks
    # corrupt data: chunk 3 window 1 (and a later window), chunk 7 last window, chunk 11 first byte
    expect = [-1] * n_chunks
    for k, (c, w) in enumerate(((3, min(1, wpc - 1)), (3, wpc - 1), (7, wpc - 1), (11, 0))):
        d[c * chunk_len + w * bpc + 5 + k] ^= 0x10
        expect[c] = w if expect[c] < 0 else min(expect[c], w)
    assert run() == expect
    # a stored list shorter than the computed one: the first unreachable index is reported
    short = [e if 0 <= e < wpc - 1 else wpc - 1 for e in expect]
    assert run(n_expected=wpc - 1) == short
    # wrong start index: every chunk mismatches at 0 (ChunkInputStream.java:497-500 computes it from offsets)
    if wpc > 2:
        assert run(start_index=1) == [0] * n_chunks


def test_verify_k_corrupted_windows_at_speed(gpu):
    # config 3 shape scaled to one GPU-second of oracle work: CRC32C, 16 KiB windows, k in {0, 1, 1000}
    torch, native, L = gpu
    bpc, chunk_len, n_chunks = 16384, 4 * MIB, 256   # 65,536 windows: eight lanes per window
    wpc = chunk_len // bpc
    g = torch.Generator(device="cuda")
    g.manual_seed(3)
    d = torch.empty(n_chunks * chunk_len, dtype=torch.uint8, device="cuda")
    d.view(torch.int64).random_(generator=g)
    stored = torch.from_numpy(np.frombuffer(dev_compute(gpu, oracle.CRC32C, bpc, d, chunk_len, n_chunks, chunk_len), dtype=np.uint8).copy()).cuda()
    # the stored sums themselves are right (sampled against the oracle)
    hs = stored.cpu().numpy()
    for w in (0, 1, 255, 256, 5000, n_chunks * wpc - 1):
        win = d[w * bpc: (w + 1) * bpc].cpu().numpy()
        assert bytes(hs[4 * w: 4 * w + 4]) == oracle.compute(oracle.CRC32C, bpc, win)
    fm = torch.zeros(n_chunks, dtype=torch.int64, device="cuda")
    rng = np.random.default_rng(8)
    for k in (0, 1, 1000):
        bad = rng.choice(n_chunks * wpc, size=k, replace=False)
        for w in bad:
            d[int(w) * bpc + int(rng.integers(0, bpc))] ^= 0x01
        native.check(L.ozcs_verify_device(0, oracle.CRC32C, bpc, d.data_ptr(), chunk_len, n_chunks, chunk_len,
                                          stored.data_ptr(), wpc * 4, wpc, 0, fm.data_ptr(),
                                          torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
        expect = [-1] * n_chunks
        for w in bad:
            c, i = divmod(int(w), wpc)
            expect[c] = i if expect[c] < 0 else min(expect[c], i)
        assert fm.cpu().tolist() == expect
        stored = torch.from_numpy(np.frombuffer(dev_compute(gpu, oracle.CRC32C, bpc, d, chunk_len, n_chunks, chunk_len), dtype=np.uint8).copy()).cuda()


# ------------------------------------------------------------------ benchmark-sized properties

def test_full_size_properties_crc32_1mib(gpu):
    """8 GiB of 4 MiB chunks at 1 MiB windows (the headline configuration with 32 blocks)."""
    torch, native, L = gpu
    nbytes = 32 * 256 * MIB
    chunk = 4 * MIB
    g = torch.Generator(device="cuda")
    g.manual_seed(77)
    a = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    for o in range(0, nbytes, 1 << 30):
        a[o: o + (1 << 30)].view(torch.int64).random_(generator=g)
    sa = np.frombuffer(dev_compute(gpu, oracle.CRC32, MIB, a, chunk, nbytes // chunk, chunk), dtype=">u4")
    assert sa.size == nbytes // MIB
    # (1) sampled windows against the oracle, first/last window of the batch included
    rng = np.random.default_rng(5)
    for w in [0, 1, 3, 4, sa.size - 1] + rng.integers(0, sa.size, 40).tolist():
        win = a[w * MIB: (w + 1) * MIB].cpu().numpy()
        assert int(sa[w]) == oracle.crc32(win.tobytes()), w
    # (2) CRC is affine over GF(2): crc(x ^ y) = crc(x) ^ crc(y) ^ crc(0...0) for equal lengths
    half = nbytes // 2
    x, y = a[:half], a[half:]
    z = torch.bitwise_xor(x, y)
    sz = np.frombuffer(dev_compute(gpu, oracle.CRC32, MIB, z, chunk, half // chunk, chunk), dtype=">u4")
    zero = oracle.crc32(bytes(MIB))
    nh = half // MIB
    assert np.array_equal(sz, sa[:nh] ^ sa[nh:] ^ np.uint32(zero))
    # (3) permuting chunks permutes their digests
    del z
    perm = torch.randperm(half // chunk, generator=torch.Generator().manual_seed(1))
    xp = x.view(half // chunk, chunk)[perm.cuda()].contiguous().view(-1)
    sp = np.frombuffer(dev_compute(gpu, oracle.CRC32, MIB, xp, chunk, half // chunk, chunk), dtype=">u4")
    assert np.array_equal(sp.reshape(-1, 4), sa[:nh].reshape(-1, 4)[perm.numpy()])
    del xp
    # (4) compute -> verify round trip on the full batch, then one flipped bit is found
    stored = torch.from_numpy(sa.view(np.uint8).copy()).cuda()
    fm = torch.zeros(nbytes // chunk, dtype=torch.int64, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    native.check(L.ozcs_verify_device(0, oracle.CRC32, MIB, a.data_ptr(), chunk, nbytes // chunk, chunk, stored.data_ptr(), 16, 4, 0, fm.data_ptr(), st))
    torch.cuda.synchronize()
    assert int(fm.max()) == -1 and int(fm.min()) == -1
    a[1234 * chunk + 3 * MIB + 99] ^= 0x80
    native.check(L.ozcs_verify_device(0, oracle.CRC32, MIB, a.data_ptr(), chunk, nbytes // chunk, chunk, stored.data_ptr(), 16, 4, 0, fm.data_ptr(), st))
    torch.cuda.synchronize()
    h = fm.cpu().numpy()
    assert h[1234] == 3 and (np.delete(h, 1234) == -1).all()


def test_sha256_md5_many_windows(gpu):
    # lane-per-window batching at a size where every lane of many wavefronts is busy: 8192 windows of 64 KiB
    torch, native, L = gpu
    bpc, n = 65536, 8192
    g = torch.Generator(device="cuda")
    g.manual_seed(9)
    a = torch.empty(n * bpc, dtype=torch.uint8, device="cuda")
    a.view(torch.int64).random_(generator=g)
    rng = np.random.default_rng(2)
    pick = [0, 63, 64, n - 1] + rng.integers(0, n, 24).tolist()
    for t in (oracle.SHA256, oracle.MD5):
        ds = native.DIGEST_SIZE[t]
        s = dev_compute(gpu, t, bpc, a, 4 * MIB, n * bpc // (4 * MIB), 4 * MIB)
        for w in pick:
            assert s[w * ds: (w + 1) * ds] == oracle.compute(t, bpc, a[w * bpc: (w + 1) * bpc].cpu().numpy()), (t, w)


# ------------------------------------------------------------------ batched and concurrent host calls

def test_compute_batch_mixed_types(gpu):
    torch, native, L = gpu
    specs = [(oracle.CRC32, MIB, 4 * MIB), (oracle.MD5, MIB, 4 * MIB), (oracle.CRC32, MIB, 4 * MIB), (oracle.CRC32, MIB, 3 * MIB + 5),
             (oracle.SHA256, 2048, 3072), (oracle.CRC32C, 16384, 100000), (oracle.NONE, 10, 1000), (oracle.CRC32, 256, 0)]
    n = len(specs)
    arr = (native.OzcsChunk * n)()
    bufs, outs = [], []
    for i, (t, bpc, ln) in enumerate(specs):
        b = oracle.splitmix64(100 + i, ln)
        ds = native.DIGEST_SIZE[t]
        o = ctypes.create_string_buffer(max(1, ((ln + bpc - 1) // bpc) * ds))
        bufs.append(b)
        outs.append(o)
        arr[i].data, arr[i].len = (b.ctypes.data if ln else None), ln
        arr[i].sums, arr[i].sums_cap = ctypes.cast(o, ctypes.c_void_p).value, len(o)
        arr[i].bpc, arr[i].type = bpc, t
    native.check(L.ozcs_compute_batch(arr, n))
    for i, (t, bpc, ln) in enumerate(specs):
        ds = native.DIGEST_SIZE[t]
        exp = oracle.compute(t, bpc, bufs[i])
        assert arr[i].status == native.OK and arr[i].n_sums == (len(exp) // ds if ds else 0)
        assert outs[i].raw[: len(exp)] == exp
    arr[1].type = 42
    assert L.ozcs_compute_batch(arr, n) == native.EINVAL and arr[1].status == native.EINVAL and arr[0].status == native.OK


def test_concurrent_callers(gpu):
    # many Java threads call concurrently in the reference's deployment (one writer per BlockOutputStream)
    torch, native, L = gpu
    errs = []

    def work(seed):
        try:
            for j in range(6):
                t = ALL[(seed + j) % 4]
                n = 50000 + 9973 * seed + j
                d = oracle.splitmix64(seed * 100 + j, n)
                if native.compute(t, 4096, d) != oracle.compute(t, 4096, d):
                    errs.append((seed, j))
        except Exception as e:  # noqa: BLE001
            errs.append(repr(e))

    th = [threading.Thread(target=work, args=(s,)) for s in range(8)]
    [x.start() for x in th]
    [x.join() for x in th]
    assert errs == []


# ------------------------------------------------------------------ the reference's own unit tests, through the mirror

def test_reference_TestChecksum(gpu):
    from hadoop_ozone_amd import Checksum, ChecksumData, ChecksumType, OzoneChecksumException
    data = bytes(oracle.splitmix64(1, 55) % 26 + 97)   # RandomStringUtils.randomAlphabetic(55)
    for t in (ChecksumType.SHA256, ChecksumType.CRC32, ChecksumType.CRC32C, ChecksumType.MD5):
        c = Checksum(t, 10)
        cd = c.computeChecksum(data)
        assert len(cd.getChecksums()) == 6                     # TestChecksum.java:56-58
        assert Checksum.verifyChecksum(data, cd) is True        # :60-61
        bad = bytearray(data)
        bad[50] = (bad[50] + 1) & 0xFF
        cd2 = c.computeChecksum(bytes(bad))
        assert cd2 != cd                                       # :67-80 testIncorrectChecksum
        with pytest.raises(OzoneChecksumException, match="Checksum mismatch at index 5"):
            Checksum.verifyChecksum(bytes(bad), cd)
        assert cd == c.computeChecksum(data)                   # testChecksumMismatchForDifferentChecksumTypes' premise
        assert cd.getChecksums() == [oracle.compute(int(t), 10, data[i:i + 10]) for i in range(0, 55, 10)]
    a = Checksum(ChecksumType.SHA256, 10).computeChecksum(data)
    b = Checksum(ChecksumType.CRC32, 10).computeChecksum(data)
    assert a != b                                              # :86-101
    # offsets (computeChecksum(byte[], off, len)) and the list overloads
    c = Checksum(ChecksumType.CRC32, 20)
    whole = oracle.splitmix64(2, 100).tobytes()
    cd = c.computeChecksum(whole)
    assert c.computeChecksum(whole, 20, 40).getChecksums() == cd.getChecksums()[1:3]
    windows = [whole[i:i + 20] for i in range(0, 100, 20)]
    assert c.computeChecksum(windows, reference_list_quirk=False) == cd
    assert c.computeChecksum(windows).getChecksums() == cd.getChecksums()[:4]   # the reference's iterator stops early (five elements: four)
    assert Checksum.verifyChecksum(windows[1:3], cd, 1, False) is True        # ChunkInputStream.java:497-503
    assert Checksum.verifyChecksum([whole[20:60]], cd, 1, True) is True       # V0 layout, Checksum.java:274-277
    with pytest.raises(OzoneChecksumException, match="Checksum mismatch at index 0"):
        Checksum.verifyChecksum(windows[1:3], cd, 2, False)
    # ragged list: every element is one window, truncated to bpc bytes (Checksum.java:186-198)
    rag = [whole[0:7], whole[7:40], whole[40:41]]
    got = c.computeChecksum(rag).getChecksums()
    assert got == [oracle.compute(oracle.CRC32, 20, r[:20]) for r in rag]
    with pytest.raises(OzoneChecksumException, match="Computed checksum has 5 number of checksums. Original checksum has 2"):
        Checksum.verifyChecksum(whole, ChecksumData(ChecksumType.CRC32, 20, cd.getChecksums()[:2]), 0)


@pytest.mark.parametrize("t", [oracle.CRC32, oracle.CRC32C])
def test_reference_TestChecksumByteBuffer(gpu, t):
    # TestChecksumByteBuffer.java:33-101 checks the engine on "hello world!" and on 1 KiB of random bytes fed one
    # byte at a time and as the growing prefixes [0, i).  The GPU path has no incremental engine (the window loop is
    # what it replaces), so the same inputs go in as whole messages: every prefix of the buffer is one window of
    # one batched call, and the concatenation of all prefixes -- what the reference's cumulative updates spell
    # out -- is one more.
    torch, native, L = gpu
    hello = b"hello world!"
    assert native.compute(t, 64, hello) == oracle.compute(t, 64, hello)
    rng = np.random.default_rng(1000 + t)
    for _ in range(3):
        buf = rng.integers(0, 256, 1 << 10, dtype=np.uint8)
        length = int(rng.integers(2, 1 << 10))
        n = length
        arr = (native.OzcsChunk * n)()
        outs = np.zeros((n, 4), dtype=np.uint8)
        for i in range(n):                      # prefix [0, i + 1)
            arr[i].data, arr[i].len = buf.ctypes.data, i + 1
            arr[i].sums, arr[i].sums_cap = outs[i].ctypes.data, 4
            arr[i].bpc, arr[i].type = 1 << 10, t
        native.check(L.ozcs_compute_batch(arr, n))
        engine = oracle.crc32 if t == oracle.CRC32 else oracle.crc32c
        for i in range(n):
            assert int.from_bytes(bytes(outs[i]), "big") == engine(buf[:i + 1]), i
        cumulative = np.concatenate([buf[:i] for i in range(length)])   # update(bytes, 0, i) for i < length, no reset
        assert native.compute(t, len(cumulative), cumulative) == engine(cumulative).to_bytes(4, "big")


@pytest.mark.parametrize("t", [oracle.CRC32, oracle.CRC32C])
def test_reference_TestChecksumImplsComputeSameValues(gpu, t):
    # TestChecksumImplsComputeSameValues.java:38-101: 64 MiB of random bytes, six window sizes, every engine must
    # give the same value for every window.  Here the engines are the GPU kernels (host-pointer call and
    # device-resident call), the oracle's slice-by-8 restatement and its carry-less-multiply engine.
    torch, native, L = gpu
    n = 64 << 20
    host = np.random.default_rng(64 + t).integers(0, 256, n, dtype=np.uint8)
    dtensor = torch.from_numpy(host).cuda()
    for bpc in (512, 1024, 2048, 4096, 32768, 1048576):
        want = oracle.compute(t, bpc, host)
        assert len(want) == 4 * (n // bpc)
        assert native.compute(t, bpc, host) == want
        assert dev_compute(gpu, t, bpc, dtensor, n, 1, n) == want
        if oracle.has_clmul():
            assert oracle.compute_clmul(t, bpc, host) == want


# ------------------------------------------------------------------ asynchronous pinned pipeline

def test_pipe_mixed_types_against_oracle(gpu):
    # the end-to-end write-path configuration in miniature: host -> device -> host through pinned slots, CRC32 : MD5 = 3 : 1
    torch, native, L = gpu
    pipe = ctypes.c_void_p()
    native.check(L.ozcs_pipe_create(0, 6, 4 * MIB, ctypes.byref(pipe)))
    try:
        jobs = []
        pending = {}
        rng = np.random.default_rng(17)

        def reap(ticket):
            j = pending.pop(ticket)
            n = ctypes.c_uint64()
            native.check(L.ozcs_pipe_wait(pipe, ticket, ctypes.byref(n)))
            j["n"] = n.value

        for i in range(40):
            t = oracle.MD5 if i % 4 == 3 else oracle.CRC32
            ln = 4 * MIB if i % 5 else int(rng.integers(1, 4 * MIB))
            bpc = MIB if i % 7 else 16384
            d = oracle.splitmix64(900 + i, ln)
            ds = native.DIGEST_SIZE[t]
            out = ctypes.create_string_buffer(((ln + bpc - 1) // bpc) * ds)
            ticket = ctypes.c_int(-1)
            if i % 2:   # producer fills the pinned slot in place
                buf = ctypes.c_void_p()
                while True:
                    rc = L.ozcs_pipe_acquire(pipe, ctypes.byref(buf), ctypes.byref(ticket))
                    if rc == native.EBUSY:
                        reap(next(iter(pending)))
                        continue
                    native.check(rc)
                    break
                ctypes.memmove(buf.value, d.ctypes.data, ln)
                native.check(L.ozcs_pipe_submit_slot(pipe, ticket.value, t, bpc, ln, out, len(out)))
            else:       # data lives elsewhere: the library stages it
                while True:
                    rc = L.ozcs_pipe_submit(pipe, t, bpc, d.ctypes.data, ln, out, len(out), ctypes.byref(ticket))
                    if rc == native.EBUSY:
                        reap(next(iter(pending)))
                        continue
                    native.check(rc)
                    break
            j = {"t": t, "bpc": bpc, "d": d, "out": out}
            jobs.append(j)
            pending[ticket.value] = j
        native.check(L.ozcs_pipe_drain(pipe))
        for j in jobs:
            exp = oracle.compute(j["t"], j["bpc"], j["d"])
            assert j["out"].raw[: len(exp)] == exp
        # all slots taken -> EBUSY; an unsubmitted ticket cannot be waited for; bad arguments are rejected
        held = []
        buf, ticket = ctypes.c_void_p(), ctypes.c_int()
        for _ in range(6):
            native.check(L.ozcs_pipe_acquire(pipe, ctypes.byref(buf), ctypes.byref(ticket)))
            held.append(ticket.value)
        assert L.ozcs_pipe_acquire(pipe, ctypes.byref(buf), ctypes.byref(ticket)) == native.EBUSY
        assert L.ozcs_pipe_wait(pipe, held[0], None) == native.EINVAL
        out = ctypes.create_string_buffer(64)
        assert L.ozcs_pipe_submit_slot(pipe, held[0], 77, 1024, 4096, out, 64) == native.EINVAL   # slot is released
        assert L.ozcs_pipe_submit_slot(pipe, held[1], oracle.CRC32, 1024, 8 * MIB, out, 64) == native.EINVAL
        for h in held[2:]:
            native.check(L.ozcs_pipe_submit_slot(pipe, h, oracle.NONE, 1024, 100, None, 0))
            n = ctypes.c_uint64(9)
            native.check(L.ozcs_pipe_wait(pipe, h, ctypes.byref(n)))
            assert n.value == 0
    finally:
        L.ozcs_pipe_destroy(pipe)


# ------------------------------------------------------------------ C++ host mirror, in a process without PyTorch

def test_cpp_host_mirror_standalone(gpu):
    import os
    import subprocess
    from conftest import ROOT
    exe = os.path.join(ROOT, "tests", "cpp", "host_mirror_test")
    assert os.path.exists(exe), "run __graft_entry__.build() first"
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "all passed" in r.stdout


# ------------------------------------------------------------------ every CRC kernel variant on random shapes

@pytest.mark.parametrize("env", [{"OZCS_CRC_LPW": "64", "OZCS_CRC_GROUP": g} for g in ("1", "2", "4", "8", "16")] +
                         [{"OZCS_CRC_LPW": l} for l in ("4", "8", "16", "32")] +
                         [{"OZCS_CRC_LPW": "8", "OZCS_CRC_FLAT": "66"}, {"OZCS_CRC_LPW": "4", "OZCS_CRC_FLAT": "34"},
                          {"OZCS_CRC_LPW": "8", "OZCS_CRC_NO_FLAT": "1"}, {"OZCS_CRC_LPW": "8", "OZCS_CRC_NO_ALIGNED": "1"},
                          {"OZCS_CRC_HOLD": "1"}, {"OZCS_CRC_HOLD": "-1"}, {"OZCS_CRC_LPW": "16", "OZCS_CRC_HOLD": "1"}],
                         ids=lambda e: "-".join("%s%s" % (k[9:], v) for k, v in e.items()))
def test_forced_kernel_variants_random_shapes(gpu, env):
    # the launcher's heuristics normally pick the variant; here each one must be right on ANY shape
    import os
    import subprocess
    import sys
    from conftest import ROOT
    e = dict(os.environ)
    e.update(env)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "variant_check.py"), "7"], env=e,
                       capture_output=True, text=True, timeout=170)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


@pytest.mark.parametrize("lanes", ["8", "16", "32"])
def test_crc_wavefronts_starting_inside_their_windows(gpu, lanes):
    # the gapless-layout kernel with four phases (wavefronts start 0, 1/4, 1/2, 3/4 into their windows and wrap
    # round; head and tail are merged with the operators of bpc/2 and bpc/4 bytes): every power-of-two window size
    import os
    import subprocess
    import sys
    from conftest import ROOT
    e = dict(os.environ)
    e.update({"OZCS_CRC_LPW": lanes, "OZCS_CRC_PHASES": "4"})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "variant_check.py"), "13", "pow2"], env=e,
                       capture_output=True, text=True, timeout=170)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


@pytest.mark.parametrize("split,lines", [("0", "0"), ("0", "1"), ("1", "1")])
def test_sha256_one_and_two_wavefront_forms(gpu, split, lines):
    # the launcher picks the form by window count; both must be right on any shape (MD5 rides along)
    import os
    import subprocess
    import sys
    from conftest import ROOT
    e = dict(os.environ)
    e["OZCS_SHA_SPLIT"] = split
    e["OZCS_MD5_LINES"] = lines   # the two loader forms (64-byte / whole-line steps) of MD5 and of one-wavefront SHA-256
    e["OZCS_SHA_LINES"] = lines
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "variant_check.py"), "11", "digests"], env=e,
                       capture_output=True, text=True, timeout=170)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


def test_bench_two_ranks_rehearsal(gpu):
    # the launch line the driver uses for N > 1, rehearsed on one GPU: gloo for the barrier / MAX, both ranks on GPU 0
    import json
    import os
    import socket
    import subprocess
    import sys
    from conftest import ROOT
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    tail = ["--gpus", "2", "--steps", "3", "--warmup", "1", "--blocks", "2", "--dist-backend", "gloo", "--device-index", "0"]
    launched = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py")] + tail
    # ... and the bare form, in which bench.py starts its two ranks itself (as a child, before any GPU call)
    bare = [sys.executable, os.path.join(ROOT, "bench.py")] + tail
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    for cmd in (launched, bare):
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=170, cwd=ROOT, env=env)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1   # rank 0 alone prints
        rec = json.loads(lines[0])
        assert rec["n_gpus"] == 2 and rec["scaling"] == "weak" and rec["value"] > 0 and "cpu_baseline" not in rec
        assert rec["config"]["blocks_per_gpu"] == 2


def test_stats_counters_and_pipe_slot_bytes(gpu):
    torch, native, L = gpu
    before = native.stats()
    data = oracle.splitmix64(77, 3 * MIB + 5)
    sums = native.compute(oracle.CRC32C, MIB, data)
    assert sums == oracle.compute(oracle.CRC32C, MIB, data)
    after = native.stats()
    assert after["calls"] == before["calls"] + 1
    assert after["bytes_hashed"] == before["bytes_hashed"] + data.size
    assert after["windows"] == before["windows"] + 4
    assert after["h2d_bytes"] == before["h2d_bytes"] + data.size
    assert after["d2h_bytes"] == before["d2h_bytes"] + 16
    assert after["kernel_launches"] == before["kernel_launches"] + 1
    pipe = ctypes.c_void_p()
    native.check(L.ozcs_pipe_create(0, 2, 123456, ctypes.byref(pipe)))
    try:
        assert L.ozcs_pipe_slot_bytes(pipe) == 123456
    finally:
        L.ozcs_pipe_destroy(pipe)
    assert L.ozcs_pipe_slot_bytes(None) == 0


def test_pipe_shared_by_threads(gpu):
    # one pipe, four producer threads (one per open BlockOutputStream): every ticket is waited by its holder
    import threading
    torch, native, L = gpu
    pipe = ctypes.c_void_p()
    native.check(L.ozcs_pipe_create(0, 6, MIB, ctypes.byref(pipe)))
    errors = []

    def producer(k):
        try:
            for i in range(12):
                t = oracle.CRC32C if (i + k) % 3 else oracle.MD5
                ln = MIB - 37 * i - k
                bpc = 65536 if i % 2 else 16384
                d = oracle.splitmix64(5000 + 100 * k + i, ln)
                ds = native.DIGEST_SIZE[t]
                out = ctypes.create_string_buffer(((ln + bpc - 1) // bpc) * ds)
                ticket = ctypes.c_int(-1)
                while True:
                    rc = L.ozcs_pipe_submit(pipe, t, bpc, d.ctypes.data, ln, out, len(out), ctypes.byref(ticket))
                    if rc != native.EBUSY:
                        break
                    time.sleep(0.0005)   # the other threads hold every slot: theirs to wait for, not ours
                native.check(rc)
                n = ctypes.c_uint64()
                native.check(L.ozcs_pipe_wait(pipe, ticket.value, ctypes.byref(n)))
                assert n.value == (ln + bpc - 1) // bpc
                assert out.raw == oracle.compute(t, bpc, d), (k, i)
        except Exception as e:   # noqa: BLE001
            errors.append(repr(e))

    try:
        ts = [threading.Thread(target=producer, args=(k,)) for k in range(4)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        assert not errors, errors
    finally:
        L.ozcs_pipe_destroy(pipe)


def test_largest_windows_and_offsets_beyond_4_gib(gpu):
    # bytesPerChecksum is a Java int: windows of hundreds of MiB, and a chunk whose offsets need 64 bits
    torch, native, L = gpu
    GIB = 1 << 30
    host = oracle.splitmix64(31337, 5 * GIB + 12345)
    d = torch.from_numpy(host).cuda()
    # CRC: 1 GiB - 1 windows over 5 GiB + 12345 bytes (six windows, the last one short, unaligned starts)
    for t in (oracle.CRC32, oracle.CRC32C):
        bpc = GIB - 1
        assert dev_compute(gpu, t, bpc, d, host.size, 1, host.size) == oracle.compute(t, bpc, host)
    # the largest bytesPerChecksum an int holds: one window covering 2 GiB - 1 bytes, then the rest
    bpc = (1 << 31) - 1
    assert dev_compute(gpu, oracle.CRC32, bpc, d, host.size, 1, host.size) == oracle.compute(oracle.CRC32, bpc, host)
    # two chunks 2.5 GiB apart: the second chunk's windows lie beyond 4 GiB
    stride = 5 * GIB // 2
    assert dev_compute(gpu, oracle.CRC32C, 1 << 20, d, 3 * MIB + 5, 2, stride) == oracle_chunks(oracle.CRC32C, 1 << 20, host, 3 * MIB + 5, 2, stride)
    # digests: one lane walks a 200 MiB window (MD5) / a 48 MiB window (SHA-256)
    sub = d[4 * GIB + 3: 4 * GIB + 3 + 500 * MIB]
    hsub = host[4 * GIB + 3: 4 * GIB + 3 + 500 * MIB]
    assert dev_compute(gpu, oracle.MD5, 200 * MIB + 1, sub, hsub.size, 1, hsub.size) == oracle.compute(oracle.MD5, 200 * MIB + 1, hsub)
    assert dev_compute(gpu, oracle.SHA256, 48 * MIB - 7, sub[: 100 * MIB], 100 * MIB, 1, 100 * MIB) == \
        oracle.compute(oracle.SHA256, 48 * MIB - 7, hsub[: 100 * MIB])


def test_compute_batch_random_sweep(gpu):
    # 500 chunks of random type, window size and length in ONE ozcs_compute_batch call (more than one job list
    # per digest type, chunks of one byte, windows of one byte, lengths around every block and stripe boundary)
    torch, native, L = gpu
    rng = np.random.default_rng(2024)
    n = 500
    arr = (native.OzcsChunk * n)()
    specs, bufs, outs = [], [], []
    edge = [1, 2, 3, 4, 5, 15, 16, 17, 55, 56, 57, 63, 64, 65, 119, 120, 127, 128, 129, 1023, 1024, 1025, 4095, 4096, 4097]
    for i in range(n):
        t = int(rng.choice([oracle.CRC32, oracle.CRC32C, oracle.SHA256, oracle.MD5]))
        bpc = int(rng.choice([1, 7, 10, 64, 100, 512, 1000, 4096, 16384, 65536]))
        ln = int(rng.choice(edge)) if i % 3 == 0 else int(rng.integers(1, min(60 * bpc, 300000) + 1))
        if bpc < 8:
            ln = min(ln, 2000)
        b = oracle.splitmix64(7000 + i, ln)
        ds = native.DIGEST_SIZE[t]
        o = ctypes.create_string_buffer(((ln + bpc - 1) // bpc) * ds)
        specs.append((t, bpc, ln))
        bufs.append(b)
        outs.append(o)
        arr[i].data, arr[i].len = b.ctypes.data, ln
        arr[i].sums, arr[i].sums_cap = ctypes.cast(o, ctypes.c_void_p).value, len(o)
        arr[i].bpc, arr[i].type = bpc, t
    native.check(L.ozcs_compute_batch(arr, n))
    for i, (t, bpc, ln) in enumerate(specs):
        assert arr[i].status == native.OK, (i, t, bpc, ln)
        assert outs[i].raw == oracle.compute(t, bpc, bufs[i]), (i, t, bpc, ln)


def test_compute_batch_runs_of_equal_chunks(gpu):
    """Equal CRC chunks that follow each other share one launch inside ozcs_compute_batch: runs of small and of large
    chunks, lengths that are no multiple of the 256-byte layout grain or of the window size, runs cut by another
    shape, type or a digest chunk, a run of one, and a run spanning staged and unstaged sizes."""
    torch, native, L = gpu
    C, CC, S, M = oracle.CRC32, oracle.CRC32C, oracle.SHA256, oracle.MD5
    plan = ([(C, 16384, 65536)] * 9 + [(C, 16384, 65537)] * 3 + [(M, 16384, 65536)] + [(C, 16384, 65536)] * 2 +
            [(CC, 16384, 65536)] * 5 + [(CC, 4096, 65536)] * 4 + [(C, MIB, 4 * MIB)] * 6 + [(C, MIB, 4 * MIB - 100)] * 3 +
            [(S, 65536, MIB)] + [(C, MIB, 4 * MIB)] + [(CC, 100000, 3 * MIB + 77)] * 4 + [(C, 512, 1000)] * 40 +
            [(C, 512, 999)] + [(C, 512, 1000)] * 2 + [(C, MIB, 9 * MIB)] * 3 + [(C, 1, 300)] * 3 + [(CC, 1 << 20, 1)] * 5)
    n = len(plan)
    arr = (native.OzcsChunk * n)()
    bufs, outs = [], []
    for i, (t, bpc, ln) in enumerate(plan):
        b = oracle.splitmix64(9100 + i, ln)
        o = ctypes.create_string_buffer(((ln + bpc - 1) // bpc) * native.DIGEST_SIZE[t])
        bufs.append(b)
        outs.append(o)
        arr[i].data, arr[i].len = b.ctypes.data, ln
        arr[i].sums, arr[i].sums_cap = ctypes.cast(o, ctypes.c_void_p).value, len(o)
        arr[i].bpc, arr[i].type = bpc, t
    before = native.stats()
    native.check(L.ozcs_compute_batch(arr, n))
    after = native.stats()
    for i, (t, bpc, ln) in enumerate(plan):
        assert arr[i].status == native.OK and arr[i].n_sums == (ln + bpc - 1) // bpc, (i, t, bpc, ln)
        assert outs[i].raw == oracle.compute(t, bpc, bufs[i]), (i, t, bpc, ln)
    # 14 runs of CRC chunks (the 9 MiB chunks are laid out last), and three window-descriptor launches: the lone small
    # CRC32 chunk (512, 999) and the MD5 and SHA-256 chunks
    assert after["kernel_launches"] - before["kernel_launches"] == 14 + 3


def test_compute_batch_chunks_larger_than_the_staging_piece(gpu):
    """Chunks of more than 128 MiB inside a batch are hashed as pieces of whole windows: lengths that are no multiple
    of the piece or of the window, a window size that does not divide the piece, a digest type, large chunks that are
    staged whole beside them, and a rejected chunk (digest list too short) among them, which must fail alone."""
    torch, native, L = gpu
    C, CC, S, M = oracle.CRC32, oracle.CRC32C, oracle.SHA256, oracle.MD5
    plan = [(C, MIB, 130 * MIB), (CC, 16384, 129 * MIB + 12345), (C, 3 * MIB + 1, 140 * MIB), (M, 65536, 128 * MIB + 1),
            (S, MIB, 9 * MIB), (C, MIB, 4 * MIB), (CC, 12 * MIB, 30 * MIB), (C, 100000, 8 * MIB + 1), (C, 512, 17 * MIB)]
    n = len(plan)
    arr = (native.OzcsChunk * (n + 1))()
    bufs, outs = [], []
    for i, (t, bpc, ln) in enumerate(plan):
        b = oracle.splitmix64(9300 + i, ln)
        o = ctypes.create_string_buffer(((ln + bpc - 1) // bpc) * native.DIGEST_SIZE[t])
        bufs.append(b)
        outs.append(o)
        arr[i].data, arr[i].len = b.ctypes.data, ln
        arr[i].sums, arr[i].sums_cap = ctypes.cast(o, ctypes.c_void_p).value, len(o)
        arr[i].bpc, arr[i].type = bpc, t
    # the rejected one: 16 MiB at 1 MiB windows needs 64 bytes of digests, 60 are offered
    short = ctypes.create_string_buffer(b"\xEE" * 60, 60)
    arr[n].data, arr[n].len = bufs[0].ctypes.data, 16 * MIB
    arr[n].sums, arr[n].sums_cap = ctypes.cast(short, ctypes.c_void_p).value, 60
    arr[n].bpc, arr[n].type = MIB, C
    assert L.ozcs_compute_batch(arr, n + 1) == native.EINVAL
    assert arr[n].status == native.EINVAL and arr[n].n_sums == 0 and short.raw == b"\xEE" * 60
    for i, (t, bpc, ln) in enumerate(plan):
        assert arr[i].status == native.OK and arr[i].n_sums == (ln + bpc - 1) // bpc, (i, t, bpc, ln)
        assert outs[i].raw == oracle.compute(t, bpc, bufs[i]), (i, t, bpc, ln)
    # the same list through ozcs_verify_batch: stored digests of chunk 1 get one flipped bit in window 300
    v = (native.OzcsVChunk * n)()
    stored = [bytearray(o.raw) for o in outs]
    stored[1][300 * 4] ^= 1
    keep = [(ctypes.c_char * len(x)).from_buffer(x) for x in stored]
    for i, (t, bpc, ln) in enumerate(plan):
        v[i].data, v[i].len = bufs[i].ctypes.data, ln
        v[i].expected, v[i].n_expected = ctypes.addressof(keep[i]), (ln + bpc - 1) // bpc
        v[i].start_index, v[i].bpc, v[i].type = 0, bpc, t
    assert L.ozcs_verify_batch(v, n) == native.OK
    assert [v[i].first_mismatch for i in range(n)] == [-1, 300] + [-1] * (n - 2)


def test_full_size_properties_other_configurations(gpu):
    """The per-GPU share of the 16 KiB verify configuration (8 GiB, 524,288 windows) and the whole SHA-256
    configuration (32 GiB at 1 MiB windows, 32,768 windows): sampled windows against the oracle, permutation
    invariance, and the compute -> verify round trip."""
    torch, native, L = gpu
    chunk = 4 * MIB
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator(device="cuda")
    g.manual_seed(4711)
    nbytes = 128 * 256 * MIB
    a = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    for o in range(0, nbytes, 1 << 30):
        a[o: o + (1 << 30)].view(torch.int64).random_(generator=g)
    rng = np.random.default_rng(8)

    # ---- CRC32C, 16 KiB windows, first 8 GiB ----
    n8 = 32 * 256 * MIB
    bpc = 16384
    sums = dev_compute(gpu, oracle.CRC32C, bpc, a, chunk, n8 // chunk, chunk)
    sc = np.frombuffer(sums, dtype=">u4")
    assert sc.size == n8 // bpc
    for w in [0, 1, 255, 256, sc.size - 1] + rng.integers(0, sc.size, 200).tolist():
        assert int(sc[w]) == oracle.crc32c(a[w * bpc: (w + 1) * bpc].cpu().numpy().tobytes()), w
    stored = torch.from_numpy(np.frombuffer(sums, dtype=np.uint8).copy()).cuda()
    fm = torch.zeros(n8 // chunk, dtype=torch.int64, device="cuda")
    wpc = chunk // bpc
    native.check(L.ozcs_verify_device(0, oracle.CRC32C, bpc, a.data_ptr(), chunk, n8 // chunk, chunk, stored.data_ptr(),
                                      wpc * 4, wpc, 0, fm.data_ptr(), st))
    torch.cuda.synchronize()
    assert int(fm.max()) == -1 and int(fm.min()) == -1
    # 1000 corrupted windows: every chunk reports its first one
    bad_w = np.unique(rng.integers(0, sc.size, 1000))
    expect = np.full(n8 // chunk, -1, dtype=np.int64)
    idx = torch.from_numpy(bad_w.astype(np.int64) * bpc + 77).cuda()
    a[idx] ^= 0x04
    for w in bad_w:
        c = w // wpc
        expect[c] = w % wpc if expect[c] < 0 else min(expect[c], w % wpc)
    native.check(L.ozcs_verify_device(0, oracle.CRC32C, bpc, a.data_ptr(), chunk, n8 // chunk, chunk, stored.data_ptr(),
                                      wpc * 4, wpc, 0, fm.data_ptr(), st))
    torch.cuda.synchronize()
    assert np.array_equal(fm.cpu().numpy(), expect)
    a[idx] ^= 0x04
    del stored, fm

    # ---- SHA-256, 1 MiB windows, all 32 GiB (the two-wavefront form: 32,768 windows) ----
    sh = dev_compute(gpu, oracle.SHA256, MIB, a, chunk, nbytes // chunk, chunk)
    assert len(sh) == (nbytes // MIB) * 32
    for w in [0, 3, 4, nbytes // MIB - 1] + rng.integers(0, nbytes // MIB, 12).tolist():
        win = a[w * MIB: (w + 1) * MIB].cpu().numpy()
        assert sh[w * 32: (w + 1) * 32] == oracle.compute(oracle.SHA256, MIB, win), w
    # the one-wavefront form on a quarter of the batch (65,536+ windows would be needed to select it by
    # count, so it is compared through the job path's neighbour: a different chunking of the same bytes)
    q = nbytes // 4
    sh2 = dev_compute(gpu, oracle.SHA256, MIB, a, 8 * MIB, q // (8 * MIB), 8 * MIB)   # 8 MiB chunks: same windows
    assert sh2 == sh[: (q // MIB) * 32]


def test_large_host_buffer_takes_the_piecewise_path(gpu):
    # ozcs_compute on 32 MiB or more is cut into pieces of whole windows and gathered through pinned memory
    torch, native, L = gpu
    d = oracle.splitmix64(8080, 72 * MIB + 12345)
    for t, bpc in ((oracle.CRC32, MIB), (oracle.CRC32C, 1000003), (oracle.CRC32, 3 * MIB + 16), (oracle.MD5, 65536),
                   (oracle.SHA256, 700001), (oracle.CRC32C, 16 * MIB)):
        assert native.compute(t, bpc, d) == oracle.compute(t, bpc, d), (t, bpc)


def test_large_host_buffer_verify(gpu):
    # ozcs_verify on 32 MiB or more goes through the batched path; outcomes as for small buffers
    torch, native, L = gpu
    d = oracle.splitmix64(9090, 40 * MIB + 77)
    for t, bpc in ((oracle.CRC32, MIB), (oracle.CRC32C, 1000003), (oracle.MD5, 262144)):
        stored = oracle.compute(t, bpc, d)
        ds = native.DIGEST_SIZE[t]
        n = len(stored) // ds
        assert native.verify(t, bpc, d, stored, 0) == (native.OK, -1)
        bad = d.copy()
        bad[17 * MIB + 5] ^= 2
        assert native.verify(t, bpc, bad, stored, 0) == (native.OK, (17 * MIB + 5) // bpc)
        assert native.verify(t, bpc, d, stored[: (n - 3) * ds], 0)[0] == native.ERANGE
        assert native.verify(t, bpc, d, b"", 0)[0] == native.EEMPTY
        assert native.verify(t, bpc, d[5 * bpc:], stored, 5) == (native.OK, -1)
        assert oracle.verify(t, bpc, bad, stored, 0) == (1, (17 * MIB + 5) // bpc)


def test_device_entry_points_under_graph_capture(gpu):
    # the device-resident calls only enqueue kernels (and one memset) on the caller's stream: they can be
    # captured into a HIP graph and replayed on new data
    torch, native, L = gpu
    n_chunks, chunk, bpc = 8, 4 * MIB, 16384
    d = torch.zeros(n_chunks * chunk, dtype=torch.uint8, device="cuda")
    wpc = chunk // bpc
    sums = torch.zeros(n_chunks * wpc * 4, dtype=torch.uint8, device="cuda")
    big = torch.zeros(n_chunks * 4 * 4, dtype=torch.uint8, device="cuda")
    fm = torch.zeros(n_chunks, dtype=torch.int64, device="cuda")
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            st = torch.cuda.current_stream().cuda_stream
            native.check(L.ozcs_compute_device(0, oracle.CRC32C, bpc, d.data_ptr(), chunk, n_chunks, chunk, sums.data_ptr(), sums.numel(), st))
            native.check(L.ozcs_compute_device(0, oracle.CRC32, MIB, d.data_ptr(), chunk, n_chunks, chunk, big.data_ptr(), big.numel(), st))
            native.check(L.ozcs_verify_device(0, oracle.CRC32C, bpc, d.data_ptr(), chunk, n_chunks, chunk, sums.data_ptr(), wpc * 4, wpc, 0,
                                              fm.data_ptr(), st))
    torch.cuda.current_stream().wait_stream(s)
    for seed in (1, 2):
        host = oracle.splitmix64(600 + seed, n_chunks * chunk)
        d.copy_(torch.from_numpy(host))
        g.replay()
        torch.cuda.synchronize()
        assert bytes(sums.cpu().numpy()) == oracle_chunks(oracle.CRC32C, bpc, host, chunk, n_chunks, chunk)
        assert bytes(big.cpu().numpy()) == oracle_chunks(oracle.CRC32, MIB, host, chunk, n_chunks, chunk)
        assert fm.cpu().tolist() == [-1] * n_chunks


@pytest.mark.parametrize("t,bpc,gib", [(oracle.CRC32, 16384, 1), (oracle.CRC32C, 32768, 2)])
def test_default_kernel_selection_four_phase_flat_crc(gpu, t, bpc, gib):
    # No launcher override: a gapless, 16-byte aligned batch of at least 16 windows per wavefront of the chip (65,536
    # windows on 256 CUs) takes the kernel the launcher itself picks there -- 8 lanes per window with four phases at
    # 16 KiB, 32 lanes with four phases at 32 KiB.  Every digest is compared with the oracle, compute and verify.
    import os
    torch, native, L = gpu
    assert not any(k.startswith("OZCS_CRC_") for k in os.environ), "launcher overrides are set"
    n = gib << 30
    assert n // bpc >= 65536
    g = torch.Generator(device="cuda")
    g.manual_seed(777 + bpc)
    d = torch.empty(n, dtype=torch.uint8, device="cuda")
    d.view(torch.int64).random_(generator=g)
    host = d.cpu().numpy()
    chunk = 4 * MIB
    got = dev_compute(gpu, t, bpc, d, chunk, n // chunk, chunk)
    want = oracle.compute(t, bpc, host)            # chunks back to back, bpc divides the chunk: one run of windows
    assert got == want
    # verify with the launcher's own choice as well: clean, then one flipped byte in the last window of a middle chunk
    ds, wpc, nch = 4, chunk // bpc, n // chunk
    stored = torch.from_numpy(np.frombuffer(want, dtype=np.uint8).copy()).cuda()
    fm = torch.zeros(nch, dtype=torch.int64, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    native.check(L.ozcs_verify_device(0, t, bpc, d.data_ptr(), chunk, nch, chunk, stored.data_ptr(), wpc * ds, wpc, 0,
                                      fm.data_ptr(), stream))
    torch.cuda.synchronize()
    assert int(fm.max()) == -1 and int(fm.min()) == -1
    c = nch // 2
    d[c * chunk + chunk - 3] ^= 0x20
    native.check(L.ozcs_verify_device(0, t, bpc, d.data_ptr(), chunk, nch, chunk, stored.data_ptr(), wpc * ds, wpc, 0,
                                      fm.data_ptr(), stream))
    torch.cuda.synchronize()
    v = fm.cpu().numpy()
    assert v[c] == wpc - 1 and (np.delete(v, c) == -1).all()



@pytest.mark.parametrize("bpc,mib", [(512, 320), (128, 1100), (128, 70), (4096, 600), (256, 100)])
def test_compute_small_windows_digests_held_to_the_end_of_a_wavefronts_reads(gpu, bpc, mib):
    # compute at windows of 4 KiB and less keeps a wavefront's digests in registers and stores them when its reads are
    # done: several passes per wavefront (more than one collector register), a batch that needs more than one launch
    # (128 B x 1,100 MiB: a launch of 4-lane windows takes 1 GiB), a last pass that is not full, and every digest against the oracle
    torch, native, L = gpu
    n = mib * (1 << 20) + 3 * bpc          # a ragged number of windows: the last pass is partial
    host = oracle.splitmix64(900 + bpc, n)
    d = torch.from_numpy(host).cuda()
    for t in (oracle.CRC32, oracle.CRC32C):
        got = dev_compute(gpu, t, bpc, d, n, 1, n)
        want = oracle.compute(t, bpc, host)
        assert len(got) == len(want) == 4 * (n // bpc)
        if got != want:
            a, b = np.frombuffer(got, dtype=np.uint32), np.frombuffer(want, dtype=np.uint32)
            bad = np.nonzero(a != b)[0]
            raise AssertionError("type %d bpc %d: %d of %d digests differ, first at window %d" % (t, bpc, len(bad), len(a), bad[0]))


@pytest.mark.parametrize("t", [oracle.SHA256, oracle.MD5])
def test_default_kernel_selection_whole_line_digest_loaders(gpu, t):
    # No override: from eight wavefronts per CU on (131,072 windows on 256 CUs) the launcher takes the loaders that
    # fetch whole 128-byte lines per step.  Every digest is compared with the oracle.
    import os
    torch, native, L = gpu
    assert "OZCS_SHA_LINES" not in os.environ and "OZCS_MD5_LINES" not in os.environ and "OZCS_SHA_SPLIT" not in os.environ
    bpc, nwin = 1024, 131072 + 64
    chunk = 64 * bpc + 520                          # a ragged last window of 520 bytes per chunk: 65 windows a chunk
    nch = (nwin + 64) // 65
    g = torch.Generator(device="cuda")
    g.manual_seed(4242 + t)
    d = torch.empty(nch * chunk, dtype=torch.uint8, device="cuda")
    d.view(torch.int64).random_(generator=g)
    host = d.cpu().numpy()
    got = dev_compute(gpu, t, bpc, d, chunk, nch, chunk)
    assert nch * 65 >= 131072
    assert got == oracle_chunks(t, bpc, host, chunk, nch, chunk)
