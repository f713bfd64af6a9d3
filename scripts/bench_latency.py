#!/usr/bin/env python3
"""Per-call latency of small device-resident launches (one chunk) and of the synchronous host call: what one
writer thread sees per chunk.  Prints one JSON line per case; times are medians over 200 calls, in microseconds."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from hadoop_ozone_amd import native

MIB = 1 << 20
L = native.lib()
native.check(L.ozcs_init(0))
st = torch.cuda.current_stream().cuda_stream
rng = np.random.default_rng(0)
for name, t, chunk, bpc in (("CRC32 4 MiB chunk @ 1 MiB", native.CRC32, 4 * MIB, MIB), ("CRC32 4 MiB chunk @ 16 KiB", native.CRC32, 4 * MIB, 16384),
                            ("CRC32C 64 KiB chunk @ 16 KiB", native.CRC32C, 65536, 16384), ("CRC32 1 MiB chunk @ 1 MiB", native.CRC32, MIB, MIB),
                            ("MD5 4 MiB chunk @ 16 KiB", native.MD5, 4 * MIB, 16384), ("SHA256 4 MiB chunk @ 16 KiB", native.SHA256, 4 * MIB, 16384)):
    h = rng.integers(0, 256, chunk, dtype=np.uint8)
    d = torch.from_numpy(h).cuda()
    n = (chunk + bpc - 1) // bpc
    sums = torch.zeros(n * native.DIGEST_SIZE[t], dtype=torch.uint8, device="cuda")
    dev, host = [], []
    for i in range(220):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        native.check(L.ozcs_compute_device(0, t, bpc, d.data_ptr(), chunk, 1, chunk, sums.data_ptr(), sums.numel(), st))
        torch.cuda.synchronize()
        dev.append(time.perf_counter() - t0)
        t0 = time.perf_counter()
        native.compute(t, bpc, h)
        host.append(time.perf_counter() - t0)
    print(json.dumps({"case": name, "device_resident_launch_plus_sync_us": round(float(np.median(dev[20:])) * 1e6, 1),
                      "host_pointer_call_us": round(float(np.median(host[20:])) * 1e6, 1)}), flush=True)
