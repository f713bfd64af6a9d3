#!/usr/bin/env python3
"""End-to-end host -> device -> host rate of the pinned pipeline (the end-to-end write-path configuration): 4 MiB chunks through
ozcs_pipe slots, CRC32 : MD5 = 3 : 1 at 1 MiB windows, H2D copy / kernel / digest copy-back on three streams.
Prints one JSON line per variant.  PCIe-bound by design: this is NOT the headline device-resident number."""
import ctypes
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401  (shares its HIP runtime with the library)
from hadoop_ozone_amd import native

MIB = 1 << 20


def run_in_order(L, n_slots, n_chunks, mix, staged, bpc=MIB):
    """One strictly ordered stream on an in-order pipe (ozcs_pipe_set_in_order): the producer blocks for a slot, slots go
    back when the GPU is done with them, outcomes are taken in submission order as they become available."""
    pipe = ctypes.c_void_p()
    native.check(L.ozcs_pipe_create(0, n_slots, 4 * MIB, ctypes.byref(pipe)))
    native.check(L.ozcs_pipe_set_in_order(pipe, 1))
    src = np.random.default_rng(1).integers(0, 256, 4 * MIB, dtype=np.uint8)
    cap = (4 * MIB // bpc) * 32
    ring = 17 * n_slots                                   # digests of undelivered outcomes stay where they are
    outs = [ctypes.create_string_buffer(cap) for _ in range(ring)]
    buf, ticket, done = ctypes.c_void_p(), ctypes.c_int(), ctypes.c_int()
    delivered = 0
    max_backlog = 0
    for warm in (True, False):
        n_run = n_slots if warm else n_chunks
        t0 = time.perf_counter()
        submitted = delivered = 0
        while delivered < n_run:
            if submitted < n_run:
                t = native.MD5 if (mix and not warm and submitted % 4 == 3) else native.CRC32
                if staged:                                # the library copies the chunk into a slot (shared with its helpers)
                    rc = L.ozcs_pipe_submit(pipe, t, bpc, src.ctypes.data, 4 * MIB, outs[submitted % ring], cap, ctypes.byref(ticket))
                else:
                    rc = L.ozcs_pipe_acquire_wait(pipe, ctypes.byref(buf), ctypes.byref(ticket))
                    if rc == native.OK:
                        rc = L.ozcs_pipe_submit_slot(pipe, ticket.value, t, bpc, 4 * MIB, outs[submitted % ring], cap)
                if rc == native.OK:
                    submitted += 1
                elif rc != native.EBUSY:                  # EBUSY: no slot free or the backlog is full, deliver first
                    native.check(rc)
            block = 1 if (submitted == n_run or rc == native.EBUSY) else 0
            while L.ozcs_pipe_next_in_order(pipe, block, ctypes.byref(done), None, None) == native.OK:
                delivered += 1
                block = 0
            max_backlog = max(max_backlog, submitted - delivered)
        dt = time.perf_counter() - t0
    L.ozcs_pipe_destroy(pipe)
    return n_chunks * 4 * MIB / dt / 1e9, max_backlog


def run(L, n_slots, n_chunks, mix, staged, bpc=MIB, any_order=False, only=None):
    """any_order: a full pipe is relieved by whichever ticket completes first (many streams share the pipe, or the
    stream ships chunks as their digests arrive) instead of by the oldest ticket (one strictly ordered stream)."""
    pipe = ctypes.c_void_p()
    native.check(L.ozcs_pipe_create(0, n_slots, 4 * MIB, ctypes.byref(pipe)))
    src = np.random.default_rng(1).integers(0, 256, 4 * MIB, dtype=np.uint8)
    cap = (4 * MIB // bpc) * 32
    outs = [ctypes.create_string_buffer(cap) for _ in range(n_slots)]
    buf, ticket, n = ctypes.c_void_p(), ctypes.c_int(), ctypes.c_uint64()
    done_ticket = ctypes.c_int()

    def relieve():
        if any_order:
            native.check(L.ozcs_pipe_next_completed(pipe, 1, ctypes.byref(done_ticket), None, None))
        else:
            native.check(L.ozcs_pipe_wait(pipe, order.pop(0), ctypes.byref(n)))

    # fill every slot once
    held = []
    for _ in range(n_slots):
        native.check(L.ozcs_pipe_acquire(pipe, ctypes.byref(buf), ctypes.byref(ticket)))
        ctypes.memmove(buf.value, src.ctypes.data, 4 * MIB)
        held.append(ticket.value)
    for h in held:
        native.check(L.ozcs_pipe_submit_slot(pipe, h, native.CRC32, bpc, 4 * MIB, outs[L.ozcs_pipe_ticket_slot(h)], cap))
    native.check(L.ozcs_pipe_drain(pipe))
    order = []
    t0 = time.perf_counter()
    for i in range(n_chunks):
        t = only if only is not None else (native.MD5 if (mix and i % 4 == 3) else native.CRC32)
        if staged:
            while True:
                rc = L.ozcs_pipe_submit(pipe, t, bpc, src.ctypes.data, 4 * MIB, outs[i % n_slots], cap, ctypes.byref(ticket))
                if rc != native.EBUSY:
                    break
                relieve()
            native.check(rc)
        else:
            while True:
                rc = L.ozcs_pipe_acquire(pipe, ctypes.byref(buf), ctypes.byref(ticket))
                if rc != native.EBUSY:
                    break
                relieve()
            native.check(rc)
            native.check(L.ozcs_pipe_submit_slot(pipe, ticket.value, t, bpc, 4 * MIB,
                                                 outs[L.ozcs_pipe_ticket_slot(ticket.value)], cap))
        if not any_order:
            order.append(ticket.value)
    native.check(L.ozcs_pipe_drain(pipe))
    dt = time.perf_counter() - t0
    L.ozcs_pipe_destroy(pipe)
    return n_chunks * 4 * MIB / dt / 1e9


def h2d_ceiling():
    """Plain pinned host -> device copies of 4 MiB, back to back on one stream: what the link gives."""
    h = torch.empty(64 * 4 * MIB, dtype=torch.uint8).pin_memory()
    d = torch.empty(4 * MIB, dtype=torch.uint8, device="cuda")
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(512):
            o = (i % 64) * 4 * MIB
            d.copy_(h[o:o + 4 * MIB], non_blocking=True)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    return 512 * 4 * MIB / dt / 1e9


def main():
    L = native.lib()
    native.check(L.ozcs_init(0))
    if len(sys.argv) > 1 and sys.argv[1] == "digest-streams":   # one digest type only: what an MD5 or SHA-256 stream gets
        for name, t in (("MD5", native.MD5), ("SHA256", native.SHA256)):
            for bpc in (MIB, 16384):
                for slots, any_order in ((64, False), (256, False), (256, True), (1024, True)):
                    n = 8192 if slots >= 1024 else 4096
                    print(json.dumps({"types": name + " only", "bytes_per_checksum": bpc, "slots": slots,
                                      "harvest": "any completed ticket" if any_order else "oldest ticket first",
                                      "GBps": round(run(L, slots, n, False, False, bpc, any_order, only=t), 2)}), flush=True)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "mix-in-order":   # one stream that ships its chunks in order: CRC32:MD5 3:1 at 1 MiB windows
        for slots in (128, 256, 512):
            print(json.dumps({"slots": slots, "oldest_first_GBps": round(run(L, slots, 4096, True, False, MIB, False), 2)}), flush=True)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "in-order":   # the in-order pipe against waiting for the oldest ticket
        for staged in (False, True):
            for slots in (32, 64, 128, 256):
                g, backlog = run_in_order(L, slots, 4096, True, staged, MIB)
                print(json.dumps({"types": "CRC32:MD5 3:1", "bytes_per_checksum": MIB, "slots": slots,
                                  "input": "memcpy from pageable memory into the slot" if staged else "produced in the pinned slot",
                                  "harvest": "in-order pipe (slots recycled on completion, outcomes in submission order)",
                                  "GBps": round(g, 2), "most_outcomes_pending": backlog}), flush=True)
        for slots in (64, 128, 256):
            print(json.dumps({"types": "CRC32:MD5 3:1", "bytes_per_checksum": MIB, "slots": slots,
                              "harvest": "oldest ticket first (slot held until its ticket is waited for)",
                              "GBps": round(run(L, slots, 4096, True, False, MIB, False), 2)}), flush=True)
        g, backlog = run_in_order(L, 64, 4096, False, False, MIB)
        print(json.dumps({"types": "CRC32", "slots": 64, "harvest": "in-order pipe", "GBps": round(g, 2)}), flush=True)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "copy-streams":   # the workloads scripts/sweep_pipe_copy_streams.sh compares
        print(json.dumps({"types": "CRC32", "slots": 16, "harvest": "oldest first", "GBps": round(run(L, 16, 4096, False, False, MIB, False), 2)}), flush=True)
        print(json.dumps({"types": "CRC32", "slots": 16, "harvest": "oldest first", "input": "memcpy into the slot", "GBps": round(run(L, 16, 4096, False, True, MIB, False), 2)}), flush=True)
        print(json.dumps({"types": "CRC32", "slots": 64, "harvest": "in-order pipe", "GBps": round(run_in_order(L, 64, 4096, False, False, MIB)[0], 2)}), flush=True)
        print(json.dumps({"types": "CRC32:MD5 3:1", "slots": 64, "harvest": "any completed", "GBps": round(run(L, 64, 4096, True, False, MIB, True), 2)}), flush=True)
        print(json.dumps({"types": "CRC32:MD5 3:1", "slots": 128, "harvest": "in-order pipe", "GBps": round(run_in_order(L, 128, 4096, True, False, MIB)[0], 2)}), flush=True)
        print(json.dumps({"types": "CRC32:MD5 3:1", "slots": 256, "harvest": "in-order pipe", "GBps": round(run_in_order(L, 256, 8192, True, False, MIB)[0], 2)}), flush=True)
        print(json.dumps({"types": "CRC32:MD5 3:1", "slots": 64, "bytes_per_checksum": 16384, "harvest": "oldest first", "GBps": round(run(L, 64, 4096, True, False, 16384, False), 2)}), flush=True)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "mix":   # quick look: CRC32:MD5 3:1 at 1 MiB windows only
        for slots in (64, 128):
            print(json.dumps({"slots": slots, "any_order_GBps": round(run(L, slots, 4096, True, False, MIB, True), 2)}), flush=True)
        return
    print(json.dumps({"metric": "pinned hipMemcpyAsync H2D, 4 MiB pieces, no kernel", "value": round(h2d_ceiling(), 2),
                      "unit": "GB/s"}), flush=True)
    for mix, staged, slots, bpc, any_order in (
            (False, False, 16, MIB, False), (False, True, 16, MIB, False),
            (True, False, 16, MIB, False), (True, False, 64, MIB, False), (True, False, 128, MIB, False),
            (True, False, 256, MIB, False),
            (True, False, 16, MIB, True), (True, False, 32, MIB, True), (True, False, 64, MIB, True),
            (True, False, 96, MIB, True), (True, False, 128, MIB, True), (True, False, 256, MIB, True), (True, True, 128, MIB, True),
            (True, False, 16, 16384, False), (True, False, 64, 16384, False), (True, True, 64, 16384, False),
            (True, False, 64, 16384, True)):
        gbps = run(L, slots, 4096, mix, staged, bpc, any_order)
        print(json.dumps({"metric": "end-to-end pinned pipeline GB/s", "value": round(gbps, 2), "unit": "GB/s",
                          "types": "CRC32:MD5 3:1" if mix else "CRC32", "slots": slots, "chunk_bytes": 4 * MIB,
                          "bytes_per_checksum": bpc,
                          "harvest": "any completed ticket" if any_order else "oldest ticket first",
                          "input": "memcpy from pageable memory into the slot" if staged else "produced in the pinned slot"}),
              flush=True)


if __name__ == "__main__":
    main()
