"""The pinned pipeline beyond the basic flow: completion in any order, ticket tenancies under concurrent
ozcs_pipe_drain, and a refused digest launch.  Digests are compared with the CPU oracle (bit-exact)."""
import ctypes
import os
import subprocess
import sys
import threading
import time

import pytest

import oracle
from conftest import ROOT

pytestmark = pytest.mark.gpu

MIB = 1 << 20


@pytest.fixture(scope="module")
def gpu():
    import torch  # before the library: both then share one HIP runtime
    from hadoop_ozone_amd import native
    assert torch.cuda.is_available()
    L = native.lib()
    native.check(L.ozcs_init(0))
    torch.cuda.set_device(0)
    return torch, native, L


def test_pipe_next_completed_any_order(gpu):
    # many streams sharing one pipe: chunks are harvested as they complete, CRC chunks overtake MD5 chunks
    # submitted before them (a 1 MiB MD5 window keeps one lane busy for milliseconds)
    torch, native, L = gpu
    pipe = ctypes.c_void_p()
    native.check(L.ozcs_pipe_create(0, 8, 4 * MIB, ctypes.byref(pipe)))
    try:
        ticket, status, n = ctypes.c_int(-1), ctypes.c_int(99), ctypes.c_uint64()
        assert L.ozcs_pipe_next_completed(pipe, 0, ctypes.byref(ticket), ctypes.byref(status), ctypes.byref(n)) == native.EEMPTY
        assert L.ozcs_pipe_next_completed(pipe, 1, ctypes.byref(ticket), None, None) == native.EEMPTY
        jobs, order, submitted = {}, [], 0
        total = 48

        def harvest(block):
            rc = L.ozcs_pipe_next_completed(pipe, block, ctypes.byref(ticket), ctypes.byref(status), ctypes.byref(n))
            if rc == native.EBUSY:
                return False
            native.check(rc)
            j = jobs.pop(ticket.value)
            assert status.value == native.OK
            assert 0 <= L.ozcs_pipe_ticket_slot(ticket.value) < 8
            exp = oracle.compute(j["t"], j["bpc"], j["d"])
            assert n.value * native.DIGEST_SIZE[j["t"]] == len(exp)
            assert j["out"].raw[:len(exp)] == exp, j["i"]
            order.append(j["i"])
            return True

        while submitted < total:
            i = submitted
            t = (oracle.MD5, oracle.CRC32, oracle.CRC32C, oracle.SHA256)[i % 4] if i % 8 < 4 else oracle.CRC32
            bpc = MIB if i % 3 else 65536
            ln = 4 * MIB - 1000 * (i % 5)
            d = oracle.splitmix64(7000 + i, ln)
            out = ctypes.create_string_buffer(((ln + bpc - 1) // bpc) * native.DIGEST_SIZE[t])
            tk = ctypes.c_int(-1)
            rc = L.ozcs_pipe_submit(pipe, t, bpc, d.ctypes.data, ln, out, len(out), ctypes.byref(tk))
            if rc == native.EBUSY:
                assert harvest(1)
                continue
            native.check(rc)
            assert tk.value not in jobs
            jobs[tk.value] = {"t": t, "bpc": bpc, "d": d, "out": out, "i": i}
            submitted += 1
            harvest(0)
        while jobs:
            assert harvest(1)
        assert sorted(order) == list(range(total))
        assert order != list(range(total)), "every chunk completed in submission order: digest batches block CRC chunks"
        assert L.ozcs_pipe_next_completed(pipe, 1, ctypes.byref(ticket), None, None) == native.EEMPTY
    finally:
        L.ozcs_pipe_destroy(pipe)


def test_pipe_in_order_delivery_recycles_slots_behind_slow_digests(gpu):
    # BlockOutputStream's shape: outcomes in submission order, while the slot of a finished CRC chunk goes back to the
    # pool although an older MD5 / SHA-256 chunk (milliseconds per 1 MiB window) is still being hashed
    torch, native, L = gpu
    pipe = ctypes.c_void_p()
    n_slots = 6
    native.check(L.ozcs_pipe_create(0, n_slots, 4 * MIB, ctypes.byref(pipe)))
    try:
        ticket, status, n = ctypes.c_int(-1), ctypes.c_int(99), ctypes.c_uint64()
        assert L.ozcs_pipe_next_in_order(pipe, 0, ctypes.byref(ticket), None, None) == native.EINVAL   # mode not set
        native.check(L.ozcs_pipe_set_in_order(pipe, 1))
        assert L.ozcs_pipe_next_in_order(pipe, 1, ctypes.byref(ticket), None, None) == native.EEMPTY
        total = 40
        jobs, delivered, undelivered_peak = [], 0, 0
        buf, tk = ctypes.c_void_p(), ctypes.c_int(-1)

        def deliver(block):
            nonlocal delivered
            rc = L.ozcs_pipe_next_in_order(pipe, block, ctypes.byref(ticket), ctypes.byref(status), ctypes.byref(n))
            if rc in (native.EBUSY, native.EEMPTY):
                return False
            native.check(rc)
            j = jobs[delivered]                      # strictly the oldest undelivered submission
            assert ticket.value == j["ticket"] and status.value == native.OK
            exp = oracle.compute(j["t"], j["bpc"], j["d"])
            assert n.value * native.DIGEST_SIZE[j["t"]] == len(exp) and j["out"].raw[:len(exp)] == exp, delivered
            delivered += 1
            return True

        for i in range(total):
            t = oracle.MD5 if i % 8 == 0 else (oracle.SHA256 if i % 8 == 4 else (oracle.CRC32, oracle.CRC32C)[i % 2])
            bpc = MIB if t in (oracle.MD5, oracle.SHA256) else 65536
            ln = 4 * MIB - 999 * (i % 7)
            d = oracle.splitmix64(12000 + i, ln)
            out = ctypes.create_string_buffer(((ln + bpc - 1) // bpc) * native.DIGEST_SIZE[t])
            native.check(L.ozcs_pipe_acquire_wait(pipe, ctypes.byref(buf), ctypes.byref(tk)))   # blocks for a slot, never EBUSY here
            ctypes.memmove(buf.value, d.ctypes.data, ln)
            native.check(L.ozcs_pipe_submit_slot(pipe, tk.value, t, bpc, ln, out, len(out)))
            jobs.append({"t": t, "bpc": bpc, "d": d, "out": out, "ticket": tk.value})
            while deliver(0):
                pass
            undelivered_peak = max(undelivered_peak, len(jobs) - delivered)
        # per-ticket calls have no meaning once slots are recycled behind the tickets
        assert L.ozcs_pipe_wait(pipe, jobs[-1]["ticket"], None) == native.EINVAL
        assert L.ozcs_pipe_next_completed(pipe, 0, ctypes.byref(ticket), None, None) == native.EINVAL
        assert L.ozcs_pipe_set_in_order(pipe, 0) == native.EBUSY                                 # outcomes are queued
        native.check(L.ozcs_pipe_drain(pipe))            # GPU work done; outcomes stay queued
        while delivered < total:
            assert deliver(1)
        assert L.ozcs_pipe_next_in_order(pipe, 0, ctypes.byref(ticket), None, None) == native.EEMPTY
        # more chunks were in flight or waiting for delivery than the pipe has slots: slots were recycled behind
        # the digest chunk at the head of the queue
        assert undelivered_peak > n_slots, undelivered_peak
        native.check(L.ozcs_pipe_set_in_order(pipe, 0))
        # release: an acquired slot handed back unsubmitted is free again
        held = []
        for _ in range(n_slots):
            native.check(L.ozcs_pipe_acquire(pipe, ctypes.byref(buf), ctypes.byref(tk)))
            held.append(tk.value)
        assert L.ozcs_pipe_acquire(pipe, ctypes.byref(buf), ctypes.byref(tk)) == native.EBUSY
        assert L.ozcs_pipe_acquire_wait(pipe, ctypes.byref(buf), ctypes.byref(tk)) == native.EBUSY   # waiting cannot help
        native.check(L.ozcs_pipe_release(pipe, held[2]))
        assert L.ozcs_pipe_release(pipe, held[2]) == native.EINVAL
        native.check(L.ozcs_pipe_acquire(pipe, ctypes.byref(buf), ctypes.byref(tk)))
        assert L.ozcs_pipe_ticket_slot(tk.value) == L.ozcs_pipe_ticket_slot(held[2]) and tk.value != held[2]
        assert L.ozcs_pipe_sums_bytes(pipe) >= (4 * MIB // 512 + 1) * 32
    finally:
        L.ozcs_pipe_destroy(pipe)


def test_stats_phase_times_of_host_pointer_calls(gpu):
    # ABI 2: one call in OZCS_STATS_SAMPLE_EVERY per pooled context is timed with events on its own stream
    torch, native, L = gpu
    d = oracle.splitmix64(3, 8 * MIB)
    before = native.stats()
    for _ in range(40):
        native.compute(native.CRC32, MIB, d)
    after = native.stats()
    timed = after["timed_calls"] - before["timed_calls"]
    assert 2 <= timed <= 3, timed                                       # 40 calls, one in 16 timed
    assert after["timed_bytes"] - before["timed_bytes"] == timed * 8 * MIB
    h2d, ker, d2h = (after[k] - before[k] for k in ("h2d_us", "kernel_us", "d2h_us"))
    # 8 MiB over PCIe takes 150 us at the very least and under 10 ms; the kernel some microseconds
    assert 100 * timed <= h2d <= 10000 * timed, (h2d, timed)
    assert 1 <= ker <= 2000 * timed and d2h <= 2000 * timed, (ker, d2h)
    assert after["errors"] == before["errors"]


def test_pipe_ticket_names_one_tenancy(gpu):
    torch, native, L = gpu
    pipe = ctypes.c_void_p()
    native.check(L.ozcs_pipe_create(0, 1, MIB, ctypes.byref(pipe)))
    try:
        d = oracle.splitmix64(1, MIB)
        out = ctypes.create_string_buffer(64)
        t1, t2, n = ctypes.c_int(), ctypes.c_int(), ctypes.c_uint64(77)
        native.check(L.ozcs_pipe_submit(pipe, oracle.CRC32, 65536, d.ctypes.data, MIB, out, 64, ctypes.byref(t1)))
        native.check(L.ozcs_pipe_drain(pipe))   # ends the tenancy behind the holder's back
        assert out.raw == oracle.compute(oracle.CRC32, 65536, d)
        native.check(L.ozcs_pipe_wait(pipe, t1.value, ctypes.byref(n)))   # ... and the holder still learns the outcome
        assert n.value == 16
        native.check(L.ozcs_pipe_poll(pipe, t1.value, None))
        # the slot is let again: a different ticket, and the old one does not touch the new tenancy
        out2 = ctypes.create_string_buffer(b"\xEE" * 16, 16)
        native.check(L.ozcs_pipe_submit(pipe, oracle.MD5, MIB, d.ctypes.data, MIB, out2, 16, ctypes.byref(t2)))
        assert t2.value != t1.value and L.ozcs_pipe_ticket_slot(t2.value) == L.ozcs_pipe_ticket_slot(t1.value) == 0
        native.check(L.ozcs_pipe_wait(pipe, t1.value, ctypes.byref(n)))   # still the recorded outcome of tenancy 1
        assert n.value == 16
        native.check(L.ozcs_pipe_wait(pipe, t2.value, ctypes.byref(n)))
        assert n.value == 1 and out2.raw == oracle.compute(oracle.MD5, MIB, d)
        native.check(L.ozcs_pipe_poll(pipe, t1.value, ctypes.byref(n)))   # tenancy 1 is still on record behind tenancy 2
        assert n.value == 16
        assert L.ozcs_pipe_poll(pipe, 12345678, None) == native.EINVAL
        assert L.ozcs_pipe_poll(pipe, -1, None) == native.EINVAL
        assert L.ozcs_pipe_ticket_slot(-5) == -1
    finally:
        L.ozcs_pipe_destroy(pipe)


def test_pipe_ticket_answered_after_its_slot_was_drained_and_let_again(gpu):
    # a holder that comes for its ticket late: the slot was freed by ozcs_pipe_drain and has had other tenancies since.
    # The pipe remembers a slot's last 16 outcomes; the digests were delivered when the slot was drained.
    torch, native, L = gpu
    pipe = ctypes.c_void_p()
    native.check(L.ozcs_pipe_create(0, 1, 2 * MIB, ctypes.byref(pipe)))     # one slot: every submission is a new tenancy of it
    tickets, outs, datas = [], [], []
    n = ctypes.c_uint64()
    for i in range(18):
        ln = 65536 * (i + 1)
        d = oracle.splitmix64(610 + i, ln)
        out = ctypes.create_string_buffer(4 * (i + 1))
        t = ctypes.c_int(-1)
        native.check(L.ozcs_pipe_submit(pipe, oracle.CRC32C, 65536, d.ctypes.data, ln, out, len(out), ctypes.byref(t)))
        native.check(L.ozcs_pipe_drain(pipe))
        assert out.raw == oracle.compute(oracle.CRC32C, 65536, d)
        tickets.append(t.value)
        outs.append(out)
        datas.append(d)
    assert len(set(tickets)) == 18
    for i in range(2, 18):                                               # the last 16 tenancies are on record
        n.value = 0
        assert L.ozcs_pipe_wait(pipe, tickets[i], ctypes.byref(n)) == native.OK and n.value == i + 1
        assert L.ozcs_pipe_poll(pipe, tickets[i], ctypes.byref(n)) == native.OK and n.value == i + 1
    for i in range(2):                                                   # older ones are not
        assert L.ozcs_pipe_wait(pipe, tickets[i], ctypes.byref(n)) == native.EINVAL
    L.ozcs_pipe_destroy(pipe)


def test_pipe_waiters_against_concurrent_drain(gpu):
    # producers wait for their own tickets while another thread keeps draining the pipe: a waiter that lost the
    # lock must never complete (or read the digests of) a later tenancy of its slot
    torch, native, L = gpu
    pipe = ctypes.c_void_p()
    native.check(L.ozcs_pipe_create(0, 3, MIB, ctypes.byref(pipe)))
    errors, stop = [], threading.Event()

    def producer(k):
        try:
            for i in range(40):
                t = (oracle.CRC32C, oracle.MD5, oracle.CRC32, oracle.SHA256)[(i + k) % 4]
                ln = MIB - 41 * i - k
                bpc = 262144 if i % 2 else 16384
                d = oracle.splitmix64(9000 + 100 * k + i, ln)
                out = ctypes.create_string_buffer(((ln + bpc - 1) // bpc) * native.DIGEST_SIZE[t])
                ticket = ctypes.c_int(-1)
                while True:
                    rc = L.ozcs_pipe_submit(pipe, t, bpc, d.ctypes.data, ln, out, len(out), ctypes.byref(ticket))
                    if rc != native.EBUSY:
                        break
                    time.sleep(0.0002)
                native.check(rc)
                n = ctypes.c_uint64()
                native.check(L.ozcs_pipe_wait(pipe, ticket.value, ctypes.byref(n)))
                assert n.value == (ln + bpc - 1) // bpc
                assert out.raw == oracle.compute(t, bpc, d), (k, i)
        except Exception as e:   # noqa: BLE001
            errors.append(repr(e))

    def drainer():
        try:
            while not stop.is_set():
                native.check(L.ozcs_pipe_drain(pipe))
        except Exception as e:   # noqa: BLE001
            errors.append("drainer: " + repr(e))

    try:
        ts = [threading.Thread(target=producer, args=(k,)) for k in range(4)]
        dr = threading.Thread(target=drainer)
        dr.start()
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        stop.set()
        dr.join()
        assert not errors, errors
    finally:
        L.ozcs_pipe_destroy(pipe)


@pytest.mark.parametrize("mode", [[], ["in-order"]], ids=["per-ticket", "in-order"])
def test_pipe_refused_digest_launch_is_reported_by_its_tickets(gpu, mode):
    e = dict(os.environ)
    e["OZCS_FAULT_PIPE_DIGEST_LAUNCH"] = "1"
    e["OZCS_LIB"] = os.path.join(ROOT, "hadoop-ozone_amd", "csrc", "libozcs_testhooks.so")   # the hooks are not in the product library
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "pipe_fault_check.py")] + mode, env=e, capture_output=True,
                       text=True, timeout=110)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert "pipe fault check ok" in r.stdout


def test_batch_calls_sharded_over_replicated_devices(gpu):
    # the multi-device paths of ozcs_compute_batch / ozcs_verify_batch on a one-GPU box: the device is listed three times
    e = dict(os.environ)
    e["OZCS_TEST_REPLICATE_DEVICES"] = "3"
    e["OZCS_LIB"] = os.path.join(ROOT, "hadoop-ozone_amd", "csrc", "libozcs_testhooks.so")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "multi_device_check.py")], env=e, capture_output=True,
                       text=True, timeout=150)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert "multi-device check ok" in r.stdout


def test_pipe_large_digest_batches(gpu):
    # hundreds of SHA-256 / MD5 chunks parked at once: one batch holds far more chunks than a kernel-argument job list
    # could, and the kernel finds every window's chunk by bisection over ragged chunk sizes and window sizes
    torch, native, L = gpu
    n_slots = 300
    pipe = ctypes.c_void_p()
    native.check(L.ozcs_pipe_create(0, n_slots, 65536, ctypes.byref(pipe)))
    try:
        jobs = []
        for rnd in range(2):
            for i in range(n_slots):
                t = (oracle.SHA256, oracle.MD5, oracle.MD5, oracle.CRC32C)[i % 4]
                ln = 65536 - 97 * (i % 50) if i % 9 else 1 + i
                bpc = (4096, 16384, 1000, 65536)[i % 4]
                d = oracle.splitmix64(31000 + rnd * 1000 + i, ln)
                out = ctypes.create_string_buffer(((ln + bpc - 1) // bpc) * native.DIGEST_SIZE[t])
                tk = ctypes.c_int(-1)
                native.check(L.ozcs_pipe_submit(pipe, t, bpc, d.ctypes.data, ln, out, len(out), ctypes.byref(tk)))
                jobs.append((t, bpc, d, out))
            native.check(L.ozcs_pipe_drain(pipe))
        for k, (t, bpc, d, out) in enumerate(jobs):
            assert out.raw == oracle.compute(t, bpc, d), k
    finally:
        L.ozcs_pipe_destroy(pipe)
