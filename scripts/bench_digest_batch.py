"""ozcs_compute_batch on many large SHA-256 / MD5 chunks (4 MiB chunks, 1 MiB windows): time per call.

Usage: python scripts/bench_digest_batch.py   (OZCS_LIB selects another build of the library)"""
import json, os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from hadoop_ozone_amd import native
L = native.lib(); native.check(L.ozcs_init(0))
rng = np.random.default_rng(0)
csz, bpc = 4 << 20, 1 << 20
for name, typ, ds in (("SHA256", native.SHA256, 32), ("MD5", native.MD5, 16)):
    for n in (32, 96, 128, 256, 512):
        big = rng.integers(0, 256, csz * n, dtype=np.uint8)
        arr = (native.OzcsChunk * n)()
        wpc = csz // bpc
        out = np.zeros(n * wpc * ds, dtype=np.uint8)
        for i in range(n):
            arr[i].data, arr[i].len = big.ctypes.data + i * csz, csz
            arr[i].sums, arr[i].sums_cap = out.ctypes.data + i * wpc * ds, wpc * ds
            arr[i].bpc, arr[i].type = bpc, typ
        native.check(L.ozcs_compute_batch(arr, n))
        t0 = time.perf_counter()
        for _ in range(2):
            native.check(L.ozcs_compute_batch(arr, n))
        dt = (time.perf_counter() - t0) / 2
        print(json.dumps({"type": name, "chunks": n, "chunk_bytes": csz, "bpc": bpc, "ms_per_call": round(dt * 1e3, 1),
                          "GBps": round(csz * n / dt / 1e9, 2)}), flush=True)
