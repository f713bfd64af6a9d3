#!/usr/bin/env python3
"""ozcs_compute_batch from pageable memory over batch shapes (chunks x chunk size): the gather into pinned staging, the
copies and the kernels of one call.  PCIe-inclusive, never the headline.  OZCS_LIB selects a library build."""
import ctypes, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401
from hadoop_ozone_amd import native

MIB = 1 << 20
L = native.lib()
native.check(L.ozcs_init(0))
rng = np.random.default_rng(0)
for n, size in ((16, 4 * MIB), (64, 4 * MIB), (128, 4 * MIB), (256, MIB), (32, 8 * MIB), (1024, 256 * 1024), (16, 32 * MIB), (6, 80 * MIB)):
    bufs = [rng.integers(0, 256, size, dtype=np.uint8) for _ in range(n)]
    arr = (native.OzcsChunk * n)()
    cap = max(64, (size // MIB + 1) * 4)
    outs = [ctypes.create_string_buffer(cap) for _ in range(n)]
    for i in range(n):
        arr[i].data, arr[i].len = bufs[i].ctypes.data, size
        arr[i].sums, arr[i].sums_cap = ctypes.cast(outs[i], ctypes.c_void_p).value, cap
        arr[i].bpc, arr[i].type = MIB, native.CRC32
    native.check(L.ozcs_compute_batch(arr, n))
    reps = max(4, (4096 * MIB) // (n * size))
    best, t_all = 1e9, time.perf_counter()
    for _ in range(reps):
        t0 = time.perf_counter()
        native.check(L.ozcs_compute_batch(arr, n))
        best = min(best, time.perf_counter() - t0)
    dt = time.perf_counter() - t_all
    print(json.dumps({"lib": os.path.basename(os.environ.get("OZCS_LIB", "") or "product"), "batch": "%d x %d KiB" % (n, size >> 10),
                      "GBps_avg": round(reps * n * size / dt / 1e9, 2), "GBps_best": round(n * size / best / 1e9, 2)}), flush=True)
