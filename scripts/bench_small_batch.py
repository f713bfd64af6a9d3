import ctypes, json, os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import torch
from hadoop_ozone_amd import native
L = native.lib(); native.check(L.ozcs_init(0))
rng = np.random.default_rng(0)
for csz, bpc, n in ((65536, 16384, 4096), (16384, 16384, 8192), (1 << 20, 1 << 20, 512), (4 << 20, 1 << 20, 128)):
    big = rng.integers(0, 256, csz * n, dtype=np.uint8)
    arr = (native.OzcsChunk * n)()
    ds = 4; wpc = (csz + bpc - 1) // bpc
    out = np.zeros(n * wpc * ds, dtype=np.uint8)
    for i in range(n):
        arr[i].data, arr[i].len = big.ctypes.data + i * csz, csz
        arr[i].sums, arr[i].sums_cap = out.ctypes.data + i * wpc * ds, wpc * ds
        arr[i].bpc, arr[i].type = bpc, native.CRC32C
    native.check(L.ozcs_compute_batch(arr, n))
    t0 = time.perf_counter()
    for _ in range(3):
        native.check(L.ozcs_compute_batch(arr, n))
    dt = (time.perf_counter() - t0) / 3
    print(json.dumps({"chunks": n, "chunk_bytes": csz, "bpc": bpc, "GBps": round(csz * n / dt / 1e9, 2), "us_per_chunk": round(dt / n * 1e6, 2)}), flush=True)
