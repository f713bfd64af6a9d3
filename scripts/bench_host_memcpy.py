#!/usr/bin/env python3
"""Host memory copy rate by thread count (pageable -> pageable and pageable -> pinned, 64 MiB per thread per pass, sources
larger than the caches): the ceiling of every path that gathers caller memory into pinned staging before the link."""
import ctypes, json, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

MIB = 1 << 20
PER = 64 * MIB
for pinned in (False, True):
    for nt in (1, 2, 4, 8, 12, 16):
        src = [np.full(PER, 7 + i, dtype=np.uint8) for i in range(nt)]
        if pinned:
            keep = [torch.empty(PER, dtype=torch.uint8).pin_memory() for _ in range(nt)]
            dst = [k.data_ptr() for k in keep]
        else:
            keep = [np.zeros(PER, dtype=np.uint8) for _ in range(nt)]
            dst = [k.ctypes.data for k in keep]
        reps = 12

        def work(i):
            for _ in range(reps):
                ctypes.memmove(dst[i], src[i].ctypes.data, PER)
        for i in range(nt):
            ctypes.memmove(dst[i], src[i].ctypes.data, PER)
        ts = [threading.Thread(target=work, args=(i,)) for i in range(nt)]
        t0 = time.perf_counter()
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        dt = time.perf_counter() - t0
        print(json.dumps({"copy": "pageable -> " + ("pinned" if pinned else "pageable"), "threads": nt,
                          "GBps": round(nt * reps * PER / dt / 1e9, 1)}), flush=True)
print(json.dumps({"cpus_allowed": len(os.sched_getaffinity(0))}))
