"""The JNI shim and its Java class cannot be built here (no JDK).  What can be checked without one: the shim is
valid C against the C ABI header (syntax and types, through a hand-written declaration stub of the JNI names it
uses), and every native method GpuChecksum.java declares has a definition in the shim with the same number of
parameters and matching JNI types."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "hadoop-ozone_amd", "jni", "ozcs_jni.c")
JAVA = os.path.join(ROOT, "hadoop-ozone_amd", "jni", "GpuChecksum.java")

JAVA_TO_JNI = {"int": "jint", "long": "jlong", "boolean": "jboolean", "byte[]": "jbyteArray", "int[]": "jintArray",
               "ByteBuffer": "jobject", "ByteBuffer[]": "jobjectArray", "void": "void"}


def test_shim_is_valid_c_against_the_abi_header():
    r = subprocess.run(["gcc", "-fsyntax-only", "-std=gnu11", "-Wall", "-Wextra", "-Werror",
                        "-I", os.path.join(ROOT, "tests", "jni_stub"), "-I", os.path.join(ROOT, "include"), SHIM],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_shim_holds_no_critical_region_and_makes_no_upcalls():
    src = open(SHIM).read()
    code = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    for banned in ("GetPrimitiveArrayCritical", "FindClass", "GetMethodID", "CallIntMethod"):
        assert banned not in code, banned


def test_every_java_native_method_has_a_matching_definition():
    java = open(JAVA).read()
    natives = re.findall(r"(?:public|private)\s+static\s+native\s+(\S+)\s+(\w+)\s*\(([^)]*)\)", java)
    assert len(natives) >= 10
    shim = re.sub(r"/\*.*?\*/", "", open(SHIM).read(), flags=re.S)
    defs = {m.group(2): (m.group(1), m.group(3)) for m in
            re.finditer(r"JNIEXPORT\s+(\w+)\s+JNICALL\s+JFN\((\w+)\)\s*\(([^)]*)\)", shim)}
    assert set(defs) == {name for _, name, _ in natives}
    for ret, name, params in natives:
        jtypes = [JAVA_TO_JNI[" ".join(p.split()[:-1])] for p in params.split(",") if p.strip()]
        cret, cparams = defs[name]
        ctypes_ = [" ".join(p.split()[:-1]) for p in cparams.split(",")]
        assert ctypes_[:2] == ["JNIEnv*", "jclass"], name
        assert ctypes_[2:] == jtypes, (name, ctypes_[2:], jtypes)
        assert cret == JAVA_TO_JNI[ret], name


# ------------------------------------------------------------------ the shim, executed: stand-in JNI function table

import ctypes  # noqa: E402

import numpy as np  # noqa: E402
import pytest  # noqa: E402

FAKE = os.path.join(ROOT, "tests", "jni_stub", "libozcs_jni_fake.so")
K_BYTES, K_INTS, K_DIRECT, K_OBJECTS = 1, 2, 3, 4
PFX = "Java_org_apache_hadoop_ozone_common_GpuChecksum_"


class FakeJvm:
    """ctypes view of the shim built against tests/jni_stub (jni.h + fake_jvm.c): Java arrays and direct buffers
    are numpy arrays wrapped in the stand-in's object records."""

    def __init__(self):
        import torch  # noqa: F401  (first, so that the shim's libozcs shares torch's HIP runtime)
        assert os.path.exists(FAKE), "run __graft_entry__.build() first"
        self.lib = L = ctypes.CDLL(FAKE)
        L.fake_env.restype = ctypes.c_void_p
        L.fake_object.restype = ctypes.c_void_p
        L.fake_object.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_long]
        L.fake_release.argtypes = [ctypes.c_void_p]
        L.fake_storage.restype = ctypes.c_void_p
        L.fake_storage.argtypes = [ctypes.c_void_p]
        L.fake_length.restype = ctypes.c_long
        L.fake_length.argtypes = [ctypes.c_void_p]
        self.env = L.fake_env()
        self.keep = []
        V, I, J, B = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_uint8
        sig = {"nativeAvailable": (B, [I]),
               "calculateChunkedSumsByteArray": (J, [I, I, V, I, V, I, I]),
               "calculateChunkedSumsDirect": (J, [I, I, V, I, I, V, I, I]),
               "verifyChunkedSumsByteArray": (J, [I, I, V, I, I, I, V, I, I]),
               "verifyChunkedSumsDirect": (J, [I, I, V, I, I, I, I, V, I, I]),
               "calculateWindowSumsDirect": (J, [I, I, V, I, I, V, V, V]),
               "pipeCreate": (J, [I, I, J]), "pipeDestroy": (None, [J]), "pipeAcquire": (V, [J, V]),
               "pipeSubmit": (I, [J, I, I, I, J, V]), "pipeWait": (J, [J, I]),
               "pipeNextCompleted": (J, [J, B, V]), "pipeTicketSlot": (I, [I]),
               "pipeRelease": (I, [J, I]), "pipeSlotBytes": (J, [J]), "pipeSumsBytes": (J, [J]),
               "pipeSetInOrder": (I, [J, B]), "pipeNextInOrder": (J, [J, B, V]), "pipeAcquireWait": (V, [J, V])}
        self.peak_local_refs = 0       # most local references any native call held at once
        for name, (ret, args) in sig.items():
            f = getattr(L, PFX + name)
            f.restype = ret
            f.argtypes = [V, V] + args
            setattr(self, name, (lambda f: lambda *a: self._call(f, *a))(f))

    def _call(self, f, *a):
        self.lib.fake_new_frame()      # a native method starts with an empty local reference frame
        r = f(self.env, None, *a)
        self.peak_local_refs = max(self.peak_local_refs, self.lib.fake_local_refs_peak())
        return r

    def obj(self, kind, arr):
        self.keep.append(arr)
        return self.lib.fake_object(kind, arr.ctypes.data, arr.size)

    def pending(self, clear=True):
        return self.lib.fake_exception_pending(1 if clear else 0)


@pytest.fixture(scope="module")
def jvm():
    return FakeJvm()


@pytest.mark.gpu
def test_shim_heap_arrays_through_its_pinned_pipe_and_beyond_a_slot(jvm):
    import oracle
    assert jvm.nativeAvailable(0) == 1
    for t, ds in ((oracle.CRC32, 4), (oracle.CRC32C, 4), (oracle.SHA256, 32), (oracle.MD5, 16)):
        for n, bpc, off in ((4 << 20, 1 << 20, 0), (100000, 16384, 7), (12 << 20, 1 << 20, 3), (1, 512, 0)):
            if ds > 4 and n > (4 << 20):
                continue
            heap = oracle.splitmix64(n + t, n + off + 5)          # the Java array, with bytes around the range
            want = oracle.compute(t, bpc, heap[off:off + n])
            sums = np.full(len(want) + 9, 0xEE, dtype=np.uint8)
            got = jvm.calculateChunkedSumsByteArray(bpc, t, jvm.obj(K_BYTES, sums), 4, jvm.obj(K_BYTES, heap), off, n)
            assert got == len(want) // ds and not jvm.pending()
            assert bytes(sums[4:4 + len(want)]) == want
            assert bytes(sums[:4]) == b"\xee" * 4 and bytes(sums[4 + len(want):]) == b"\xee" * 5
    # NONE: no digests; bad arguments: the library's EINVAL (-1), nothing written, no exception left pending
    heap = oracle.splitmix64(5, 100)
    sums = np.zeros(16, dtype=np.uint8)
    assert jvm.calculateChunkedSumsByteArray(10, 1, jvm.obj(K_BYTES, sums), 0, jvm.obj(K_BYTES, heap), 0, 100) == 0
    assert jvm.calculateChunkedSumsByteArray(10, oracle.CRC32, jvm.obj(K_BYTES, sums), 0, jvm.obj(K_BYTES, heap), 0, 100) == -1  # 40 B of sums needed
    assert jvm.calculateChunkedSumsByteArray(10, oracle.CRC32, jvm.obj(K_BYTES, sums), 0, jvm.obj(K_BYTES, heap), 95, 10) == -1
    assert jvm.calculateChunkedSumsByteArray(0, oracle.CRC32, jvm.obj(K_BYTES, sums), 0, jvm.obj(K_BYTES, heap), 0, 10) == -1
    assert jvm.calculateChunkedSumsByteArray(10, 9, jvm.obj(K_BYTES, sums), 0, jvm.obj(K_BYTES, heap), 0, 10) == -1
    assert not jvm.pending() and not sums.any()


@pytest.mark.gpu
def test_shim_direct_buffers_and_verify(jvm):
    import oracle
    n, bpc = 300000, 16384
    data = oracle.splitmix64(77, n + 64)
    want = oracle.compute(oracle.CRC32C, bpc, data[64:64 + n])
    sums = np.zeros(len(want) + 8, dtype=np.uint8)
    got = jvm.calculateChunkedSumsDirect(bpc, oracle.CRC32C, jvm.obj(K_DIRECT, sums), 8, sums.size,
                                         jvm.obj(K_DIRECT, data), 64, 64 + n)   # position / limit come from Java
    assert got == len(want) // 4 and bytes(sums[8:]) == want
    assert jvm.calculateChunkedSumsDirect(bpc, oracle.CRC32C, jvm.obj(K_DIRECT, sums), 8, sums.size,
                                          jvm.obj(K_BYTES, data), 64, 64 + n) == -1     # a heap buffer has no address
    assert jvm.calculateChunkedSumsDirect(bpc, oracle.CRC32C, jvm.obj(K_DIRECT, sums), 8, sums.size,
                                          jvm.obj(K_DIRECT, data), 64, n + 65) == -1    # limit beyond the capacity
    # verify: -1 = all windows match, k >= 0 = first failing window relative to the computed list, < -1 = error
    stored = np.frombuffer(want, dtype=np.uint8).copy()
    heap = data.copy()
    args = lambda d, start=0, nsums=len(want) // 4: (bpc, oracle.CRC32C, jvm.obj(K_BYTES, stored), 0, nsums, start,
                                                      jvm.obj(K_BYTES, d), 64, n)
    assert jvm.verifyChunkedSumsByteArray(*args(heap)) == -1
    heap[64 + 5 * bpc + 3] ^= 1
    assert jvm.verifyChunkedSumsByteArray(*args(heap)) == 5
    assert jvm.verifyChunkedSumsByteArray(*args(data, nsums=3)) < -1     # fewer stored sums than windows (OZCS_ERANGE)
    # the same through direct buffers (positions and limits from the Java side): the stored digests sit at position 8
    dsums = np.concatenate([np.zeros(8, dtype=np.uint8), stored])
    dargs = lambda d, lim=64 + n: (bpc, oracle.CRC32C, jvm.obj(K_DIRECT, dsums), 8, dsums.size, len(want) // 4, 0,
                                   jvm.obj(K_DIRECT, d), 64, lim)
    assert jvm.verifyChunkedSumsDirect(*dargs(data)) == -1
    assert jvm.verifyChunkedSumsDirect(*dargs(heap)) == 5
    assert jvm.verifyChunkedSumsDirect(*dargs(data, lim=data.size + 1)) == -2                      # limit beyond capacity
    assert jvm.verifyChunkedSumsDirect(bpc, oracle.CRC32C, jvm.obj(K_DIRECT, dsums), 8, 16, len(want) // 4, 0,
                                       jvm.obj(K_DIRECT, data), 64, 64 + n) == -2                  # digests beyond sums.limit()
    # startIndex: windows 2.. of the chunk against the stored list from index 2
    assert jvm.verifyChunkedSumsByteArray(bpc, oracle.CRC32C, jvm.obj(K_BYTES, stored), 0, len(want) // 4, 2,
                                          jvm.obj(K_BYTES, data), 64 + 2 * bpc, n - 2 * bpc) == -1


@pytest.mark.gpu
def test_shim_pipe_wrappers(jvm):
    import oracle
    pipe = jvm.pipeCreate(0, 4, 1 << 20)
    assert pipe != 0
    try:
        chunks, tickets, outs = [], [], []
        for i in range(4):
            tk = np.zeros(1, dtype=np.int32)
            buf = jvm.pipeAcquire(pipe, jvm.obj(K_INTS, tk))       # a direct ByteBuffer over the pinned slot
            assert buf and jvm.lib.fake_length(buf) == 1 << 20
            n = (1 << 20) - 1000 * i
            d = oracle.splitmix64(900 + i, n)
            ctypes.memmove(jvm.lib.fake_storage(buf), d.ctypes.data, n)
            out = np.zeros(64 * 16, dtype=np.uint8)
            t = oracle.MD5 if i % 2 else oracle.CRC32
            assert jvm.pipeSubmit(pipe, int(tk[0]), t, 65536, n, jvm.obj(K_DIRECT, out)) == 0
            assert jvm.pipeTicketSlot(int(tk[0])) == i
            chunks.append((t, d)); tickets.append(int(tk[0])); outs.append(out)
        tk = np.zeros(1, dtype=np.int32)
        assert jvm.pipeAcquire(pipe, jvm.obj(K_INTS, tk)) is None   # every slot held: null, as the Java side expects
        assert jvm.pipeWait(pipe, tickets[0]) == 16
        seen = {tickets[0]}
        while True:
            ts = np.zeros(2, dtype=np.int32)
            n = jvm.pipeNextCompleted(pipe, 1, jvm.obj(K_INTS, ts))
            if n < 0:
                assert n == -6                                       # OZCS_EEMPTY: nothing outstanding any more
                break
            assert ts[1] == 0 and n == 16
            seen.add(int(ts[0]))
        assert seen == set(tickets)
        for (t, d), out in zip(chunks, outs):
            want = oracle.compute(t, 65536, d)
            assert bytes(out[:len(want)]) == want
    finally:
        jvm.pipeDestroy(pipe)


@pytest.mark.gpu
def test_shim_one_window_per_buffer_in_one_call(jvm):
    # the List<ByteString> case: one digest per direct buffer, each cut to bytesPerSum, empty buffers skipped
    import oracle
    bpc = 4096
    lens = [4096, 4096, 0, 100, 9000, 4096, 1]
    bufs = [oracle.splitmix64(500 + i, n + 16) for i, n in enumerate(lens)]        # 8 bytes of slack either side
    objs = np.array([jvm.obj(K_DIRECT, b) for b in bufs], dtype=np.uint64)
    pos = np.full(len(lens), 8, dtype=np.int32)
    lim = np.array([8 + n for n in lens], dtype=np.int32)
    for t, ds in ((oracle.CRC32C, 4), (oracle.MD5, 16)):
        want = b"".join(oracle.compute(t, bpc, b[8:8 + min(n, bpc)]) for b, n in zip(bufs, lens) if n)
        sums = np.full(len(want) + 12, 0xEE, dtype=np.uint8)
        got = jvm.calculateWindowSumsDirect(bpc, t, jvm.obj(K_DIRECT, sums), 4, sums.size, jvm.obj(K_OBJECTS, objs),
                                            jvm.obj(K_INTS, pos), jvm.obj(K_INTS, lim))
        assert got == len(want) // ds and bytes(sums[4:4 + len(want)]) == want and bytes(sums[4 + len(want):]) == b"\xee" * 8
    short = np.zeros(8, dtype=np.uint8)   # room for two CRC digests only
    assert jvm.calculateWindowSumsDirect(bpc, oracle.CRC32, jvm.obj(K_DIRECT, short), 0, 8, jvm.obj(K_OBJECTS, objs),
                                         jvm.obj(K_INTS, pos), jvm.obj(K_INTS, lim)) == -1
    lim[1] = bufs[1].size + 1             # a limit beyond its buffer's capacity
    sums = np.zeros(64, dtype=np.uint8)
    assert jvm.calculateWindowSumsDirect(bpc, oracle.CRC32, jvm.obj(K_DIRECT, sums), 0, 64, jvm.obj(K_OBJECTS, objs),
                                         jvm.obj(K_INTS, pos), jvm.obj(K_INTS, lim)) == -1
    assert not jvm.pending()


@pytest.mark.gpu
def test_shim_256_windows_stay_within_the_guaranteed_local_references(jvm):
    # a 4 MiB chunk at 16 KiB windows handed over as 256 buffers: every GetObjectArrayElement creates a local
    # reference and a native frame is guaranteed only 16 -- the stand-in table counts them like -Xcheck:jni
    import oracle
    bpc, n = 16384, 256
    bufs = [oracle.splitmix64(7000 + i, bpc) for i in range(n)]
    objs = np.array([jvm.obj(K_DIRECT, b) for b in bufs], dtype=np.uint64)
    pos = np.zeros(n, dtype=np.int32)
    lim = np.full(n, bpc, dtype=np.int32)
    sums = np.zeros(4 * n, dtype=np.uint8)
    jvm.peak_local_refs = 0
    got = jvm.calculateWindowSumsDirect(bpc, oracle.CRC32, jvm.obj(K_DIRECT, sums), 0, sums.size,
                                        jvm.obj(K_OBJECTS, objs), jvm.obj(K_INTS, pos), jvm.obj(K_INTS, lim))
    assert got == n and not jvm.pending() and jvm.peak_local_refs <= 2
    assert bytes(sums) == b"".join(oracle.compute(oracle.CRC32, bpc, b) for b in bufs)
    # the stand-in does notice a frame that keeps them: 17 live references raise its pending-exception flag
    jvm.lib.fake_new_frame()
    GetObjectArrayElement = ctypes.CFUNCTYPE(ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32)
    table = ctypes.cast(ctypes.cast(jvm.env, ctypes.POINTER(ctypes.c_void_p))[0], ctypes.POINTER(ctypes.c_void_p))
    get = GetObjectArrayElement(table[8])
    arr = jvm.obj(K_OBJECTS, objs)
    for i in range(17):
        get(jvm.env, arr, i)
    assert jvm.lib.fake_local_refs_peak() == 17 and jvm.pending()


# ---- the Java class's routing, executed: tests/java_side.py restates GpuChecksum.computeWindows ----

@pytest.fixture()
def java(jvm):
    from java_side import JavaSide
    j = JavaSide(jvm, pipe_slots=4, pipe_slot_bytes=2 << 20)     # small, so that a 12 MiB input recycles its own slots
    yield j
    j.close()


def _windows_of(buf, bpc):
    # ChunkBufferImplWithByteBuffer.iterate (ChunkBufferImplWithByteBuffer.java:62-83): duplicates cut every bpc bytes
    out = []
    p = buf.pos
    while p < buf.lim:
        d = buf.duplicate()
        d.pos, d.lim = p, min(p + bpc, buf.lim)
        out.append(d)
        p = d.lim
    return out


@pytest.mark.gpu
def test_java_side_read_only_heap_buffers_reach_the_gpu_through_pinned_slots(jvm, java):
    # Checksum.computeChecksum(byte[]) / (ByteBuffer) make their argument read-only first (Checksum.java:142-145):
    # such a buffer has neither array() nor an address, so its windows are put() into pinned pipe slots
    import oracle
    from java_side import JBuffer
    for t, n, bpc, off in ((oracle.CRC32, 4 << 20, 1 << 20, 0), (oracle.CRC32C, 12 << 20, 16384, 5),
                           (oracle.CRC32, 55, 10, 0), (oracle.SHA256, 3072, 2048, 1), (oracle.MD5, 300000, 65536, 3),
                           (oracle.CRC32, (2 << 20) + 1, 1 << 20, 0)):
        heap = oracle.splitmix64(n + t, n + off + 9)
        buf = JBuffer(heap, off, off + n, direct=False, read_only=True)
        windows = _windows_of(buf, bpc)
        out = []
        java.calls.clear()
        jvm.peak_local_refs = 0
        assert java.computeWindows(t, bpc, windows, out)
        assert b"".join(out) == oracle.compute(t, bpc, heap[off:off + n]), (t, n, bpc)
        assert "pipeSubmit" in java.calls and "calculateWindowSumsDirect" not in java.calls
        assert not jvm.pending() and jvm.peak_local_refs <= 1
        # a read-only window has no array: ChecksumByteBufferImpl.update and MessageDigest.update consume it
        assert all(w.pos == w.lim for w in windows)
    # a writable heap buffer (BlockOutputStream's chunk buffer): same route; CRC leaves it unconsumed, a digest consumes
    heap = oracle.splitmix64(99, 1 << 20)
    for t, consumed in ((oracle.CRC32C, False), (oracle.MD5, True)):
        windows = _windows_of(JBuffer(heap), 1 << 18)
        out = []
        assert java.computeWindows(t, 1 << 18, windows, out)
        assert b"".join(out) == oracle.compute(t, 1 << 18, heap)
        assert all((w.pos == w.lim) == consumed for w in windows)


@pytest.mark.gpu
def test_java_side_lists_of_windows_ragged_and_direct(jvm, java):
    import oracle
    from java_side import JBuffer
    bpc = 4096
    # one ByteString per window (Checksum.java:282-288), read-only; a short element in the middle ends its slot's group,
    # an element longer than bpc contributes its first bpc bytes only (Checksum.java:186-198)
    lens = [4096, 4096, 100, 4096, 9000, 1, 4096]
    bufs = [oracle.splitmix64(600 + i, n + 8) for i, n in enumerate(lens)]
    windows = [JBuffer(b, 4, 4 + n, read_only=True) for b, n in zip(bufs, lens)]
    for t in (oracle.CRC32, oracle.SHA256):
        ws = [w.duplicate() for w in windows]
        out = []
        java.calls.clear()
        assert java.computeWindows(t, bpc, ws, out)
        assert out == [oracle.compute(t, bpc, b[4:4 + min(n, bpc)]) for b, n in zip(bufs, lens)]
        assert java.calls.count("pipeSubmit") == 3        # groups: [w0 w1 w2] [w3 w4 w5] [w6]: a short window ends its group
        assert [w.pos - 4 for w in ws] == [min(n, bpc) for n in lens]
    # direct buffers: in place, one native call; slices of one direct buffer are recognised as one chunk
    direct = oracle.splitmix64(41, 10 * bpc + 77)
    ws = _windows_of(JBuffer(direct, 3, direct.size, direct=True, read_only=True), bpc)
    out = []
    java.calls.clear()
    before = _stats()
    assert java.computeWindows(oracle.CRC32C, bpc, ws, out)
    assert b"".join(out) == oracle.compute(oracle.CRC32C, bpc, direct[3:])
    assert java.calls == ["calculateWindowSumsDirect"] and all(w.pos == w.lim for w in ws)
    after = _stats()
    assert after["kernel_launches"] - before["kernel_launches"] == 1      # ozcs_compute over the whole range
    # mixed direct and heap windows: everything is staged
    ws = [JBuffer(direct, 0, bpc, direct=True), JBuffer(bufs[0], 0, bpc)]
    out = []
    java.calls.clear()
    assert java.computeWindows(oracle.CRC32, bpc, ws, out) and "pipeSubmit" in java.calls
    assert out == [oracle.compute(oracle.CRC32, bpc, direct[:bpc]), oracle.compute(oracle.CRC32, bpc, bufs[0][:bpc])]


@pytest.mark.gpu
def test_java_side_declines_without_touching_anything(jvm, java):
    import oracle
    from java_side import JBuffer
    big = oracle.splitmix64(5, (2 << 20) + 4096)
    # a window larger than a pipe slot (bpc 4 MiB against 2 MiB slots): the Java engines keep the call
    ws = _windows_of(JBuffer(big, read_only=True), 4 << 20)
    out = []
    assert not java.computeWindows(oracle.CRC32, 4 << 20, ws, out) and out == [] and ws[0].pos == 0
    # an empty window (a consumed buffer of an IncrementalChunkBuffer): declined as well
    ws = [JBuffer(big, 0, 4096, read_only=True), JBuffer(big, 10, 10, read_only=True)]
    assert not java.computeWindows(oracle.CRC32, 4096, ws, out) and out == []
    # every slot held by other threads: declined, and the slots it did get are given back
    pipe = java._sharedPipe()
    held = []
    for _ in range(4):
        tk = np.zeros(1, dtype=np.int32)
        assert jvm.pipeAcquire(pipe, jvm.obj(K_INTS, tk))
        held.append(int(tk[0]))
    ws = _windows_of(JBuffer(big, 0, 1 << 20, read_only=True), 1 << 18)
    assert not java.computeWindows(oracle.CRC32, 1 << 18, ws, out) and out == [] and all(w.pos != w.lim for w in ws)
    for tk in held:
        assert jvm.pipeRelease(pipe, tk) == 0
    assert jvm.pipeRelease(pipe, held[0]) == -1          # released already
    assert java.computeWindows(oracle.CRC32, 1 << 18, ws, out)
    assert b"".join(out) == oracle.compute(oracle.CRC32, 1 << 18, big[:1 << 20])
    assert not jvm.pending()


@pytest.mark.gpu
def test_reference_TestChecksum_shapes_through_the_java_route(jvm, java):
    # TestChecksum.java:47-80 as the patched Checksum would run it: computeChecksum(byte[]) -> read-only heap view ->
    # iterate(bytesPerChecksum) -> GpuChecksum.computeWindows; verifyChecksum(byte[], ChecksumData) recomputes the same
    # way and compares with ChecksumData.verifyChecksumDataMatches.  55 alphabetic bytes at bytesPerChecksum 10 give
    # exactly 6 checksums (:47-62); changing data[50] changes the ChecksumData and fails verification at index 5 (:67-80).
    import oracle
    from hadoop_ozone_amd import ChecksumData, ChecksumType, OzoneChecksumException
    from java_side import JBuffer
    rng = np.random.default_rng(2020)
    data = rng.integers(ord("a"), ord("z") + 1, 55, dtype=np.uint8)           # RandomStringUtils.randomAlphabetic(55)

    def compute(t, bpc, arr):      # Checksum.computeChecksum(byte[]) with the one-hunk patch applied
        windows = _windows_of(JBuffer(arr, read_only=True), bpc)
        out = []
        assert java.computeWindows(int(t), bpc, windows, out)
        return ChecksumData(t, bpc, out)

    for t in (ChecksumType.SHA256, ChecksumType.CRC32, ChecksumType.CRC32C, ChecksumType.MD5):   # SHA256 is the test's default (:33-34)
        cd = compute(t, 10, data)
        assert len(cd.getChecksums()) == 6
        assert b"".join(cd.getChecksums()) == oracle.compute(int(t), 10, data)
        assert cd.verifyChecksumDataMatches(compute(t, 10, data), 0) is True                  # verifyChecksum(data, cd) (:60-61)
        bad = data.copy()
        bad[50] = ord("A") if bad[50] != ord("A") else ord("B")
        other = compute(t, 10, bad)
        assert other != cd and other.getChecksums()[:5] == cd.getChecksums()[:5]
        with pytest.raises(OzoneChecksumException, match="Checksum mismatch at index 5"):
            cd.verifyChecksumDataMatches(other, 0)
    # TestChecksumImplsComputeSameValues.java:39-41 window sizes over one buffer, both CRCs, through the same route
    big = oracle.splitmix64(64, 6 << 20)
    for bpc in (512, 1024, 2048, 4096, 32768, 1048576):
        for t in (ChecksumType.CRC32, ChecksumType.CRC32C):
            assert b"".join(compute(t, bpc, big).getChecksums()) == oracle.compute(int(t), bpc, big), (bpc, t)
    assert not jvm.pending()


def _stats():
    from hadoop_ozone_amd import native
    return native.stats()


def test_checksum_java_patch_applies_to_the_reference():
    # the reference-side change ships as a minimal-context diff; it must apply to the reference's own Checksum.java
    import shutil
    import tempfile
    rel = "hadoop-hdds/common/src/main/java/org/apache/hadoop/ozone/common/Checksum.java"
    ref = os.path.join("/root/reference", rel)
    if not os.path.exists(ref) or not shutil.which("patch"):
        pytest.skip("the reference tree (or patch) is not on this machine")
    patch = os.path.join(ROOT, "hadoop-ozone_amd", "jni", "Checksum.java.patch")
    with tempfile.TemporaryDirectory() as tmp:
        dst = os.path.join(tmp, rel)
        os.makedirs(os.path.dirname(dst))
        shutil.copy(ref, dst)
        r = subprocess.run(["patch", "-p1", "--dry-run", "-i", patch], cwd=tmp, capture_output=True, text=True)
        assert r.returncode == 0, r.stdout + r.stderr
        r = subprocess.run(["patch", "-p1", "-i", patch], cwd=tmp, capture_output=True, text=True)
        assert r.returncode == 0 and "fuzz" not in r.stdout, r.stdout + r.stderr
        text = open(dst).read()
    # the one seam every overload funnels through (Checksum.java:119-153, 210-290 all end in computeChecksum(ChunkBuffer))
    assert text.count("GpuChecksum.computeWindows(checksumType, bytesPerChecksum, windows") == 1
    assert text.count("{") == text.count("}") and text.count("(") == text.count(")")
    body = text[text.index("public ChecksumData computeChecksum(ChunkBuffer data)"):]
    assert body.index("data.iterate(bytesPerChecksum)") < body.index("if (!GpuChecksum.computeWindows(") < \
        body.index("checksumList.add(computeChecksum(b, function, bytesPerChecksum))")


def test_java_class_is_plausible_java():
    # no JDK here: at least brackets balance outside comments and strings, every statement-level line ends properly,
    # and the method the patch calls exists with the argument list the patch passes
    java = open(JAVA).read()
    code = re.sub(r"/\*.*?\*/", "", java, flags=re.S)
    code = re.sub(r"//[^\n]*", "", code)
    code = re.sub(r'"(?:[^"\\]|\\.)*"', '""', code)
    for a, b in ("{}", "()", "[]"):
        assert code.count(a) == code.count(b), (a, code.count(a), code.count(b))
    assert re.search(r"public static boolean computeWindows\(ChecksumType checksumType, int bytesPerSum,\s*"
                     r"List<ByteBuffer> windows,\s*List<ByteString> out\)", code)
    for line in code.splitlines():
        t = line.strip()
        if not t or t.startswith(("@", "case ", "default:", "*")):
            continue
        assert t[-1] in "{};,(:|&?+=" or t.endswith(("&&", "||")) or t[-1] == ")" , line
