#!/usr/bin/env python3
"""What bounds a gather-then-copy path: 512 MiB gathered by 4 threads into pinned memory alone, the same bytes copied to
the device from pinned memory alone (8 MiB pieces, one stream / four streams), and both at once on different buffers."""
import ctypes, json, os, sys, threading, time
import numpy as np
import torch

MIB = 1 << 20
TOTAL, PIECE, NT = 512 * MIB, 8 * MIB, 4
src = np.full(TOTAL, 3, dtype=np.uint8)
stage = torch.empty(TOTAL, dtype=torch.uint8).pin_memory()
other = torch.empty(TOTAL, dtype=torch.uint8).pin_memory()
dev = torch.empty(TOTAL, dtype=torch.uint8, device="cuda")
streams = [torch.cuda.Stream() for _ in range(4)]


def gather():
    def work(t):
        q = TOTAL // NT
        for off in range(t * q, (t + 1) * q, PIECE):
            ctypes.memmove(stage.data_ptr() + off, src.ctypes.data + off, PIECE)
    ts = [threading.Thread(target=work, args=(t,)) for t in range(NT)]
    for t in ts:
        t.start()
    return ts


def copies(buf, ns):
    for k, off in enumerate(range(0, TOTAL, PIECE)):
        with torch.cuda.stream(streams[k % ns]):
            dev[off:off + PIECE].copy_(buf[off:off + PIECE], non_blocking=True)


def timed(fn, reps=6):
    fn()
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return round(TOTAL / best / 1e9, 1)


def g_only():
    for t in gather():
        t.join()


def both(ns):
    ts = gather()
    copies(other, ns)
    for t in ts:
        t.join()


print(json.dumps({"gather alone (4 threads) GB/s": timed(g_only)}), flush=True)
for ns in (1, 4):
    print(json.dumps({"streams": ns, "copies alone GB/s": timed(lambda: copies(other, ns)),
                      "gather and copies of another buffer at once, GB/s of either": timed(lambda: both(ns))}), flush=True)
