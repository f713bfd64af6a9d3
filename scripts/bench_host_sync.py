#!/usr/bin/env python3
"""Rate of the synchronous host-pointer calls (ozcs_compute / ozcs_compute_batch) from pageable memory:
what a JNI caller with heap byte[] gets without the pinned pipe.  PCIe-inclusive, never the headline."""
import ctypes, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401
from hadoop_ozone_amd import native

MIB = 1 << 20
L = native.lib()
native.check(L.ozcs_init(0))
rng = np.random.default_rng(0)
for size in (4 * MIB, 16 * MIB, 32 * MIB, 64 * MIB, 1024 * MIB):
    d = rng.integers(0, 256, size, dtype=np.uint8)
    native.compute(native.CRC32, MIB, d)
    reps = max(2, (2048 * MIB) // size)
    t0 = time.perf_counter()
    for _ in range(reps):
        native.compute(native.CRC32, MIB, d)
    dt = time.perf_counter() - t0
    print(json.dumps({"call": "ozcs_compute", "bytes": size, "GBps": round(size * reps / dt / 1e9, 2)}), flush=True)
# batch of 64 x 4 MiB chunks
n = 64
bufs = [rng.integers(0, 256, 4 * MIB, dtype=np.uint8) for _ in range(n)]
arr = (native.OzcsChunk * n)()
outs = [ctypes.create_string_buffer(16) for _ in range(n)]
for i in range(n):
    arr[i].data, arr[i].len = bufs[i].ctypes.data, 4 * MIB
    arr[i].sums, arr[i].sums_cap = ctypes.cast(outs[i], ctypes.c_void_p).value, 16
    arr[i].bpc, arr[i].type = MIB, native.CRC32
native.check(L.ozcs_compute_batch(arr, n))
t0 = time.perf_counter()
for _ in range(8):
    native.check(L.ozcs_compute_batch(arr, n))
dt = time.perf_counter() - t0
print(json.dumps({"call": "ozcs_compute_batch 64 x 4 MiB", "GBps": round(8 * n * 4 * MIB / dt / 1e9, 2)}), flush=True)
# many caller threads, one 4 MiB heap chunk each per call (the shape of many open BlockOutputStreams)
import threading
for nthreads in (1, 2, 4, 8, 16):
    datas = [rng.integers(0, 256, 4 * MIB, dtype=np.uint8) for _ in range(nthreads)]
    reps = 128
    lib = native.lib()

    def work(d):
        out = ctypes.create_string_buffer(16)
        n = ctypes.c_uint64()
        for _ in range(reps):
            lib.ozcs_compute(native.CRC32, MIB, d.ctypes.data, d.size, out, 16, ctypes.byref(n))
    ts = [threading.Thread(target=work, args=(d,)) for d in datas]
    t0 = time.perf_counter()
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    dt = time.perf_counter() - t0
    print(json.dumps({"call": "ozcs_compute 4 MiB from %d threads" % nthreads,
                      "GBps": round(nthreads * reps * 4 * MIB / dt / 1e9, 2)}), flush=True)
