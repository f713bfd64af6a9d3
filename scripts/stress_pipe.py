#!/usr/bin/env python3
"""Soak test of the pinned pipe on the GPU box: random chunk sizes, checksum types and window sizes through the three ways
of taking completions -- per ticket from several threads, any completed ticket from a harvester thread beside a
submitting thread, and in submission order on an in-order pipe -- every digest compared with the oracle.
usage: scripts/stress_pipe.py [seconds per mode]   (prints one JSON line per mode; exit 1 on any mismatch)"""
import ctypes
import json
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401
import oracle
from hadoop_ozone_amd import native

MIB = 1 << 20
SLOT = 2 * MIB
TYPES = (oracle.CRC32, oracle.CRC32C, oracle.CRC32, oracle.MD5, oracle.SHA256, oracle.NONE)


def job(rng):
    t = TYPES[int(rng.integers(0, len(TYPES)))]
    bpc = int(rng.choice([512, 4096, 16384, 65536, 100000, MIB]))
    if t in (oracle.MD5, oracle.SHA256):
        bpc = max(bpc, 16384)
    ln = int(rng.integers(0, SLOT + 1)) if rng.integers(0, 8) else int(rng.integers(0, 100))
    d = oracle.splitmix64(int(rng.integers(1, 1 << 30)), ln)
    ds = native.DIGEST_SIZE[t]
    n = 0 if t == oracle.NONE else (ln + bpc - 1) // bpc
    out = ctypes.create_string_buffer(max(1, n * ds))
    return t, bpc, d, out, n, ds


def check(t, bpc, d, out, n, ds):
    want = b"" if t == oracle.NONE else oracle.compute(t, bpc, d)
    return out.raw[:n * ds] == want


def per_ticket(L, seconds, seed):
    pipe = ctypes.c_void_p()
    native.check(L.ozcs_pipe_create(0, 12, SLOT, ctypes.byref(pipe)))
    errors, done = [], [0]
    stop = time.time() + seconds

    def producer(k):
        rng = np.random.default_rng(seed * 100 + k)
        try:
            while time.time() < stop:
                t, bpc, d, out, n, ds = job(rng)
                tk = ctypes.c_int(-1)
                while True:
                    rc = L.ozcs_pipe_submit(pipe, t, bpc, d.ctypes.data if d.size else None, d.size, out, len(out), ctypes.byref(tk))
                    if rc != native.EBUSY:
                        break
                    time.sleep(0.0001)
                native.check(rc)
                got = ctypes.c_uint64()
                if k % 2:
                    native.check(L.ozcs_pipe_wait(pipe, tk.value, ctypes.byref(got)))
                else:
                    while True:
                        rc = L.ozcs_pipe_poll(pipe, tk.value, ctypes.byref(got))
                        if rc != native.EBUSY:
                            break
                    native.check(rc)
                if got.value != n or not check(t, bpc, d, out, n, ds):
                    errors.append((k, t, bpc, d.size))
                    return
                done[0] += 1
        except Exception as e:   # noqa: BLE001
            errors.append(repr(e))

    ts = [threading.Thread(target=producer, args=(k,)) for k in range(6)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    native.check(L.ozcs_pipe_drain(pipe))
    L.ozcs_pipe_destroy(pipe)
    return done[0], errors


def any_order(L, seconds, seed):
    pipe = ctypes.c_void_p()
    native.check(L.ozcs_pipe_create(0, 24, SLOT, ctypes.byref(pipe)))
    rng = np.random.default_rng(seed)
    jobs, lock, errors, done = {}, threading.Lock(), [], [0]
    stop = time.time() + seconds
    finished = threading.Event()

    def harvester():
        tk, st, got = ctypes.c_int(-1), ctypes.c_int(0), ctypes.c_uint64()
        try:
            while True:
                rc = L.ozcs_pipe_next_completed(pipe, 0, ctypes.byref(tk), ctypes.byref(st), ctypes.byref(got))
                if rc == native.EEMPTY:
                    if finished.is_set():
                        return
                    time.sleep(0.0002)
                    continue
                if rc == native.EBUSY:
                    time.sleep(0.0001)
                    continue
                native.check(rc)
                with lock:
                    j = jobs.pop(tk.value)
                if st.value != native.OK or got.value != j[4] or not check(*j):
                    errors.append((tk.value, j[0], j[1], j[2].size))
                    return
                done[0] += 1
        except Exception as e:   # noqa: BLE001
            errors.append(repr(e))

    h = threading.Thread(target=harvester)
    h.start()
    buf, tk = ctypes.c_void_p(), ctypes.c_int(-1)
    while time.time() < stop and not errors:
        j = job(rng)
        while True:
            rc = L.ozcs_pipe_acquire(pipe, ctypes.byref(buf), ctypes.byref(tk))
            if rc != native.EBUSY or errors:
                break
            time.sleep(0.0001)
        if errors:
            break
        native.check(rc)
        if j[2].size:
            ctypes.memmove(buf.value, j[2].ctypes.data, j[2].size)
        with lock:
            jobs[tk.value] = j
        native.check(L.ozcs_pipe_submit_slot(pipe, tk.value, j[0], j[1], j[2].size, j[3], len(j[3])))
    finished.set()
    h.join()
    L.ozcs_pipe_destroy(pipe)
    return done[0], errors + ([("left", len(jobs))] if jobs and not errors else [])


def in_order(L, seconds, seed):
    pipe = ctypes.c_void_p()
    native.check(L.ozcs_pipe_create(0, 16, SLOT, ctypes.byref(pipe)))
    native.check(L.ozcs_pipe_set_in_order(pipe, 1))
    rng = np.random.default_rng(seed)
    pending, errors, done = [], [], 0
    buf, tk, dn, st, got = ctypes.c_void_p(), ctypes.c_int(-1), ctypes.c_int(-1), ctypes.c_int(0), ctypes.c_uint64()
    stop = time.time() + seconds

    def deliver(block):
        nonlocal done
        rc = L.ozcs_pipe_next_in_order(pipe, block, ctypes.byref(dn), ctypes.byref(st), ctypes.byref(got))
        if rc in (native.EBUSY, native.EEMPTY):
            return False
        native.check(rc)
        ticket, j = pending.pop(0)
        if dn.value != ticket or st.value != native.OK or got.value != j[4] or not check(*j):
            errors.append((ticket, dn.value, j[0], j[1], j[2].size))
        done += 1
        return True

    while time.time() < stop and not errors:
        j = job(rng)
        rc = L.ozcs_pipe_acquire_wait(pipe, ctypes.byref(buf), ctypes.byref(tk))
        if rc == native.EBUSY:     # the backlog of undelivered outcomes is full
            deliver(1)
            continue
        native.check(rc)
        if j[2].size:
            ctypes.memmove(buf.value, j[2].ctypes.data, j[2].size)
        native.check(L.ozcs_pipe_submit_slot(pipe, tk.value, j[0], j[1], j[2].size, j[3], len(j[3])))
        pending.append((tk.value, j))
        if rng.integers(0, 4) == 0:
            while deliver(0):
                pass
    while pending and not errors:
        deliver(1)
    L.ozcs_pipe_destroy(pipe)
    return done, errors


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
    L = native.lib()
    native.check(L.ozcs_init(0))
    bad = False
    for name, fn in (("per ticket, 6 threads", per_ticket), ("any completed, harvester thread", any_order), ("in-order pipe", in_order)):
        n, errors = fn(L, seconds, 7)
        print(json.dumps({"mode": name, "seconds": seconds, "chunks_checked": n, "errors": [str(e) for e in errors[:3]]}), flush=True)
        bad = bad or bool(errors)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
