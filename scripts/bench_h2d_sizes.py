#!/usr/bin/env python3
"""What the host-to-device link gives to pinned copies of different sizes, issued back to back on one stream and
alternating over two and four streams (2 GiB per measurement).  The pinned pipe copies one 4 MiB chunk per call on one
stream: this shows whether larger or concurrent copies would carry more."""
import json
import time

import torch

MIB = 1 << 20
total = 2048 * MIB
h = torch.empty(256 * MIB, dtype=torch.uint8).pin_memory()
d = torch.empty(256 * MIB, dtype=torch.uint8, device="cuda")
for size in (1 * MIB, 4 * MIB, 16 * MIB, 64 * MIB, 256 * MIB):
    for nstreams in (1, 2, 4):
        streams = [torch.cuda.Stream() for _ in range(nstreams)]
        n = total // size
        best = 0.0
        for rep in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(n):
                o = (i * size) % (256 * MIB)
                with torch.cuda.stream(streams[i % nstreams]):
                    d[o:o + size].copy_(h[o:o + size], non_blocking=True)
            torch.cuda.synchronize()
            best = max(best, total / (time.perf_counter() - t0) / 1e9)
        print(json.dumps({"copy_MiB": size // MIB, "streams": nstreams, "GBps": round(best, 2)}), flush=True)
