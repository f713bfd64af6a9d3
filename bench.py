#!/usr/bin/env python3
"""Headline benchmark: device-resident CRC32 at 1 MiB bytes-per-checksum over a batch of 256 MiB
blocks (4 MiB chunks), the `Checksum.computeChecksum` work of BlockOutputStream.writeChunkToContainer
(BlockOutputStream.java:645-656) for every chunk of the batch, in one launch per step.

    python bench.py --gpus 1 --steps 20 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A step = one pass of the hot path over the rank's whole batch, inputs already resident in HBM.
Chunks are independent, so ranks shard the batch with no data-path collective (weak scaling: every
rank holds `--blocks` blocks).  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MIB = 1 << 20
BLOCK_BYTES = 256 * MIB   # ozone.scm.block.size, OzoneConfigKeys.java:115-116
CHUNK_BYTES = 4 * MIB     # ozone.scm.chunk.size / stream buffer, OzoneClientConfig.java:50-55
HBM_PEAK_GBPS = 8000.0    # MI355X HBM3E datasheet peak
CHECK_WINDOWS = 48        # digests of the timed run compared with the CPU oracle afterwards (first, last, random)
# Integer-VALU ceiling of the chip for SHA-256 / MD5: 256 CUs x 4 SIMDs, one wave64 vector instruction per
# 4 cycles per SIMD, at the 2.4 GHz maximum clock.
SIMDS = 256 * 4
MAX_CLOCK_HZ = 2.4e9
VALU_PEAK_GINSTR = SIMDS * MAX_CLOCK_HZ / 4 / 1e9   # 614.4 G wave-instructions/s
# Vector instructions one lane executes per 64-byte block in the one-wavefront kernels (counted in the ISA of
# digest_kernels.hip's check-free loop; SQ_INSTS_VALU per block in profiles/pmc_summary.json when present).
ISA_VALU_PER_BLOCK = {"SHA256": 1404, "MD5": 351}


def shard_blocks(total_blocks, rank, world):
    """Contiguous shard [lo, hi) of a block list for `rank` of `world` (chunk-list sharding, no overlap)."""
    lo = total_blocks * rank // world
    hi = total_blocks * (rank + 1) // world
    return lo, hi


def aggregate(bytes_per_rank_step, steps, max_elapsed_s, world):
    """Whole-job throughput in GB/s (1e9 input bytes per second) from the slowest rank's time."""
    total = bytes_per_rank_step * steps * world
    return total / max_elapsed_s / 1e9


def load_profile(workload, source_hash):
    """The committed rocprofv3 PMC record of this workload (profiles/pmc_summary.json), or (None, why): a record
    counts only when it was taken from the very sources the loaded library was built from."""
    path = os.path.join(ROOT, "profiles", "pmc_summary.json")
    try:
        with open(path) as f:
            ent = json.load(f).get(workload)
    except (OSError, ValueError):
        return None, "profiles/pmc_summary.json is missing or unreadable"
    if not ent:
        return None, "no PMC record for workload %s" % workload
    if ent.get("source_hash") != source_hash:
        return None, "PMC record of %s was taken from other kernel sources (%s, now %s): re-profile" % (
            workload, ent.get("source_hash"), source_hash)
    return ent, None


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start the ranks as a child process (before this process
    touches the GPU), relay its output and exit code."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run(cmd, env=env)
    sys.exit(r.returncode)


def nt_read_ceiling(gib):
    """A bare non-temporal read of `gib` GiB in the 1 MiB-window CRC kernel's own access pattern (G wavefronts per
    window, two 1 KiB rows each, one stripe ahead; tools/hbm_read_bench.hip, mode "crc"), measured on this box by a
    CHILD process that has finished before this process touches the GPU.  What the memory system delivers to that
    address stream with no arithmetic at all, timed like the kernel (twenty launches after ten).  The kernel's own
    request timing differs a little from the tool's, and the two figures agree within the spread between runs
    (about half a percent either way): a yardstick for "the arithmetic is free", not a bound."""
    import re
    import subprocess
    exe = os.path.join(ROOT, "tools", "hbm_read_bench")
    if not os.path.exists(exe):
        return {"measured_in_this_run": False, "note": "tools/hbm_read_bench is not built (make -C tools)"}
    try:
        r = subprocess.run([exe, str(gib), "crc"], capture_output=True, text=True, timeout=120)
    except (OSError, subprocess.SubprocessError) as e:
        return {"measured_in_this_run": False, "note": "tools/hbm_read_bench did not run: %r" % (e,)}
    pats = {}
    for m in re.finditer(r"crc pattern G=(\d+).*?avg\s+([\d.]+) GB/s\s+best\s+([\d.]+) GB/s", r.stdout):
        pats["G=%s" % m.group(1)] = {"avg_GBps": float(m.group(2)), "best_GBps": float(m.group(3))}
    if r.returncode != 0 or not pats:
        return {"measured_in_this_run": False, "note": "tools/hbm_read_bench failed (rc %d): %s" % (r.returncode, (r.stdout + r.stderr)[-200:])}
    top = max(pats, key=lambda k: pats[k]["avg_GBps"])
    return {"GBps": pats[top]["avg_GBps"], "pattern": "wavefronts per window " + top, "patterns": pats, "buffer_GiB": gib,
            "source": "tools/hbm_read_bench %d crc, child process, 20 launches after 10 warm-ups per pattern" % gib,
            "measured_in_this_run": True,
            "note": "bare read in the kernel's access pattern; kernel and bare read agree within the run-to-run spread "
                    "(about 0.5 %), so a fraction slightly above 1 is that spread, not a kernel faster than memory"}


def check_against_oracle(type_id, bpc, data, sums, n_windows, ds, seed=7):
    """After the timed loop: CHECK_WINDOWS windows of the batch (the first, the last, the rest at random) are copied to
    the host and their digests, as the timed launches left them in `sums`, compared with the CPU oracle.  The oracle
    is the checker here, never the thing measured."""
    import random
    import oracle
    wpc = (CHUNK_BYTES + bpc - 1) // bpc
    rng = random.Random(seed)
    picks = sorted({0, n_windows - 1} | {rng.randrange(n_windows) for _ in range(CHECK_WINDOWS - 2)})
    got = sums.cpu().numpy()
    bad = []
    for g in picks:
        c, i = divmod(g, wpc)
        lo = c * CHUNK_BYTES + i * bpc
        hi = min(lo + bpc, (c + 1) * CHUNK_BYTES)
        want = oracle.compute(type_id, bpc, data[lo:hi].cpu().numpy())
        if bytes(got[g * ds:(g + 1) * ds]) != want:
            bad.append(g)
    return {"windows": len(picks), "ok": not bad, "against": "CPU oracle (oracle/ozcs_oracle.c)",
            **({"mismatching_windows": bad[:8]} if bad else {})}


CPU_PASSES = 40  # passes of the headline CPU baseline: the host cores are shared, the best pass is reported


def cpu_baseline(sample, bpc):
    """The reference's CPU path restated in C and timed on this box's host cores over a bounded sample of the
    same workload.  Not the JVM (none exists here).  Two engines: PCLMULQDQ folding, which is what the JDK's
    java.util.zip.CRC32 intrinsic (the reference's production engine, ChecksumByteBufferFactory.java:74-76)
    is built on -- the reported value; and the slice-by-8 restatement of the reference's own
    ChecksumByteBuffer.CrcIntTable.update, which is also the parity oracle."""
    import zlib
    import oracle
    cores = min(16, os.cpu_count() or 1)
    n1 = min(sample.nbytes, 1 << 30)
    t1 = oracle.bench(oracle.CRC32, bpc, sample[:n1], threads=1, reps=2)
    tn = oracle.bench(oracle.CRC32, bpc, sample, threads=cores, reps=8)
    c1 = oracle.bench_clmul(oracle.CRC32, bpc, sample[:n1], threads=1, reps=3)
    cn = oracle.bench_clmul(oracle.CRC32, bpc, sample, threads=cores, reps=CPU_PASSES)
    mv = memoryview(sample[: min(sample.nbytes, 256 * MIB)])
    t0 = time.perf_counter()
    for o in range(0, len(mv), bpc):
        zlib.crc32(mv[o:o + bpc])
    tz = time.perf_counter() - t0
    return {
        "value": round(sample.nbytes / cn / 1e9, 3), "unit": "GB/s", "cores": cores, "kind": "port",
        "sample": "%d MiB of the same random bytes, CRC32 at %d B windows, best of %d passes on %d threads (about ten "
                  "seconds of CPU time); C restatement (%s), not the JVM" % (sample.nbytes // MIB, bpc, CPU_PASSES, cores,
                                                    "PCLMULQDQ folding as in the JDK intrinsic" if oracle.has_clmul()
                                                    else "table method: this CPU lacks PCLMULQDQ"),
        "one_core_GBps": round(n1 / c1 / 1e9, 3),
        "slice_by_8_GBps": round(sample.nbytes / tn / 1e9, 3),
        "slice_by_8_one_core_GBps": round(n1 / t1 / 1e9, 3),
        "zlib_crc32_one_core_GBps": round(len(mv) / tz / 1e9, 3),
    }


def cpu_baseline_other(type_name, type_id, sample, bpc):
    """CPU baseline for the configurations other than the headline one: the same algorithm on the host cores over
    a bounded sample.  CRC32C: PCLMULQDQ folding (the scheme of the JDK's CRC32C intrinsic) plus the slice-by-8
    restatement; SHA-256 / MD5: the plain C restatements of FIPS 180-4 / RFC 1321 (the JDK's MessageDigest uses
    SHA-NI / hand-tuned intrinsics where the CPU has them, so the JVM may be faster than this port)."""
    import oracle
    cores = min(16, os.cpu_count() or 1)
    if type_name == "CRC32C":
        tn = oracle.bench_clmul(type_id, bpc, sample, threads=cores, reps=6)
        t1 = oracle.bench_clmul(type_id, bpc, sample[: min(sample.nbytes, 1 << 30)], threads=1, reps=2)
        ts = oracle.bench(type_id, bpc, sample, threads=cores, reps=4)
        extra = {"one_core_GBps": round(min(sample.nbytes, 1 << 30) / t1 / 1e9, 3),
                 "slice_by_8_GBps": round(sample.nbytes / ts / 1e9, 3)}
        engine = "PCLMULQDQ folding as in the JDK intrinsic" if oracle.has_clmul() else "table method"
    else:
        tn = oracle.bench(type_id, bpc, sample, threads=cores, reps=2)
        t1 = oracle.bench(type_id, bpc, sample[: min(sample.nbytes, 256 * MIB)], threads=1, reps=1)
        extra = {"one_core_GBps": round(min(sample.nbytes, 256 * MIB) / t1 / 1e9, 3)}
        engine = "plain C from the specification, no SHA-NI"
    out = {"value": round(sample.nbytes / tn / 1e9, 3), "unit": "GB/s", "cores": cores, "kind": "port",
           "sample": "%d MiB of the same random bytes, %s at %d B windows, best pass on %d threads; C restatement (%s), "
                     "not the JVM" % (sample.nbytes // MIB, type_name, bpc, cores, engine)}
    out.update(extra)
    return out


def jvm_baseline(bpc, cores):
    """The reference's production engines (java.util.zip.CRC32 / CRC32C) timed by bench/java/CrcBaseline.java, when
    this box has a JDK; None otherwise (the build image and the project's GPU boxes have none).  Bounded: about
    10 s of CPU work, every step under a time limit, and no failure here fails the benchmark."""
    import shutil
    import subprocess
    import tempfile
    if not (shutil.which("javac") and shutil.which("java")):
        return None
    try:
        with tempfile.TemporaryDirectory() as tmp:
            subprocess.run(["javac", "-d", tmp, os.path.join(ROOT, "bench", "java", "CrcBaseline.java")], check=True,
                           capture_output=True, timeout=120)
            r = subprocess.run(["java", "-cp", tmp, "CrcBaseline", str(bpc), str(cores), "1.5"], check=True,
                               capture_output=True, text=True, timeout=120)
        for line in r.stdout.splitlines():
            if line.startswith("{"):
                return json.loads(line)
    except (OSError, ValueError, subprocess.SubprocessError):
        pass
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--blocks", type=int, default=64, help="256 MiB blocks resident per GPU (64 = 16 GiB)")
    ap.add_argument("--type", default="CRC32", choices=["CRC32", "CRC32C", "SHA256", "MD5"])
    ap.add_argument("--bpc", type=int, default=MIB)
    ap.add_argument("--mode", default="compute", choices=["compute", "verify"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ceiling", action="store_true", help="skip the bare-read measurement by tools/hbm_read_bench")
    ap.add_argument("--no-check", action="store_true", help="skip the comparison of sampled digests with the CPU oracle")
    ap.add_argument("--zeros", action="store_true", help="all-zero data instead of random bytes (the rate must not depend on the data)")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo + --device-index 0 rehearses the multi-rank path on a one-GPU box")
    ap.add_argument("--device-index", type=int, default=-1, help="GPU of this rank (default: LOCAL_RANK)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        self_launch(args)

    # the bare-read ceiling of this box, by a child process, before this process touches the GPU (one GPU only: the
    # child would otherwise compete with the other ranks)
    ceiling = None
    if args.gpus == 1 and args.type in ("CRC32", "CRC32C") and not args.no_ceiling:
        ceiling = nt_read_ceiling(min(16, max(1, args.blocks // 4)))

    import torch
    import torch.distributed as dist
    from hadoop_ozone_amd import native

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (there is no CPU path in this library)")
    if args.device_index >= 0:
        local_rank = args.device_index
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")

    L = native.lib()
    native.check(L.ozcs_init(1 << local_rank))
    type_id = getattr(native, args.type)
    ds = native.DIGEST_SIZE[type_id]

    # synthetic batch: this rank's shard of world*blocks blocks, uniform random bytes, generated on the device
    lo, hi = shard_blocks(args.blocks * world, rank, world)
    nbytes = (hi - lo) * BLOCK_BYTES
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234 + lo)
    data = torch.zeros(nbytes, dtype=torch.uint8, device=dev) if args.zeros else torch.empty(nbytes, dtype=torch.uint8, device=dev)
    step_fill = 1 << 30
    for o in range(0, 0 if args.zeros else nbytes, step_fill):
        n = min(step_fill, nbytes - o)
        data[o:o + n].view(torch.int64).random_(generator=gen)
    n_chunks = nbytes // CHUNK_BYTES
    wpc = (CHUNK_BYTES + args.bpc - 1) // args.bpc
    n_windows = n_chunks * wpc
    sums = torch.zeros(n_windows * ds, dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream()

    def compute_step():
        native.check(L.ozcs_compute_device(local_rank, type_id, args.bpc, data.data_ptr(), CHUNK_BYTES, n_chunks,
                                           CHUNK_BYTES, sums.data_ptr(), sums.numel(), stream.cuda_stream))

    first_mismatch = torch.zeros(n_chunks, dtype=torch.int64, device=dev)

    def verify_step():
        native.check(L.ozcs_verify_device(local_rank, type_id, args.bpc, data.data_ptr(), CHUNK_BYTES, n_chunks,
                                          CHUNK_BYTES, sums.data_ptr(), wpc * ds, wpc, 0, first_mismatch.data_ptr(),
                                          stream.cuda_stream))

    step = compute_step
    if args.mode == "verify":  # stored sums come from the compute kernel; every window must then match
        compute_step()
        step = verify_step

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for a, b in ev:
        a.record(stream)
        step()
        b.record(stream)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        dist.barrier()
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kernel_ms = [a.elapsed_time(b) for a, b in ev]
    avg_kernel_s = sum(kernel_ms) / len(kernel_ms) / 1e3

    # ---- outside the timed region: the digests the timed launches produced (verify: checked against) are proven on a
    # sample; a mismatch fails the run
    check = None
    if not args.no_check:
        check = check_against_oracle(type_id, args.bpc, data, sums, n_windows, ds, seed=7 + rank)
        if args.mode == "verify":
            check["verdicts_all_clean"] = bool(int(first_mismatch.max()) == -1 and int(first_mismatch.min()) == -1)
            check["ok"] = check["ok"] and check["verdicts_all_clean"]
        if not check["ok"]:
            print(json.dumps({"error": "digests of the benchmarked launches differ from the oracle", "rank": rank, "check": check}),
                  file=sys.stderr, flush=True)
            sys.exit(3)

    if rank == 0:
        value = aggregate(nbytes, args.steps, elapsed, world)
        # per launch: window bytes read + digest bytes written (verify: read, plus one verdict word per chunk)
        alg_bytes = nbytes + n_windows * ds + (n_chunks * 8 if args.mode == "verify" else 0)
        if args.mode == "verify":
            assert int(first_mismatch.max()) == -1 and int(first_mismatch.min()) == -1
        achieved = alg_bytes / avg_kernel_s / 1e9
        workload = "%s %s, %d x 256 MiB blocks per GPU (4 MiB chunks, %d B windows), device-resident" % (
            args.type, args.mode, args.blocks, args.bpc)
        src_hash = native.source_hash()
        prof, prof_note = load_profile("%s-%d-%d" % (args.type, args.bpc, args.blocks), src_hash)
        out = {
            "metric": "device-resident checksum GB/s (%s, %s bytes/checksum)" % (
                args.type, "1MB" if args.bpc == MIB else str(args.bpc)),
            "value": round(value, 2), "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u32", "data": "synthetic (all zeros)" if args.zeros else "synthetic",
            "config": {"workload": workload, "blocks_per_gpu": args.blocks, "chunk_bytes": CHUNK_BYTES,
                       "bytes_per_checksum": args.bpc, "windows_per_gpu": n_windows, "sharding": "by block, no collective"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 4),
                         "traffic": prof.get("hbm_bytes_per_launch") if prof else None,
                         "kernel_ms_avg": round(avg_kernel_s * 1e3, 4), "kernel_ms_min": round(min(kernel_ms), 4),
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "kernel_source_hash": src_hash},
        }
        if ceiling is not None:
            out["roofline"]["nt_read_ceiling"] = ceiling
            if ceiling.get("measured_in_this_run"):
                out["roofline"]["frac_of_nt_read_ceiling"] = round(achieved / ceiling["GBps"], 4)
        if check is not None:
            out["check"] = check
        if prof_note:
            out["roofline"]["traffic_note"] = prof_note
        # whole job against the HBM of all GPUs that took part (host-timed, launch gaps included); the per-kernel
        # figures above are rank 0's
        out["aggregate_hbm"] = {"achieved": round(value, 1), "peak": HBM_PEAK_GBPS * world, "unit": "GB/s",
                                "frac": round(value / (HBM_PEAK_GBPS * world), 4), "n_gpus": world}
        if args.type in ("SHA256", "MD5"):
            # The bound that matters for the digests: vector-instruction issue.  One lane hashes one window, 64 windows
            # per wavefront; a window of w bytes is (w + 8) // 64 + 1 blocks (data + padding).
            full, tail = divmod(CHUNK_BYTES, args.bpc)
            blocks = n_chunks * (full * ((args.bpc + 8) // 64 + 1) + ((tail + 8) // 64 + 1 if tail else 0))
            per_block = ISA_VALU_PER_BLOCK[args.type]
            source = "ISA count of the one-wavefront loop"
            if prof and prof.get("valu_per_block_lane"):
                per_block, source = prof["valu_per_block_lane"], "SQ_INSTS_VALU per block (profiles/pmc_summary.json)"
            ginstr = blocks * per_block / 64 / avg_kernel_s / 1e9
            hbm = out["roofline"]
            out["roofline"] = {
                "bound": "valu", "achieved": round(ginstr, 2), "peak": round(VALU_PEAK_GINSTR, 1),
                "unit": "G wave64-instr/s", "frac": round(ginstr / VALU_PEAK_GINSTR, 4), "traffic": hbm["traffic"],
                "valu_per_block_lane": per_block, "valu_per_block_source": source, "blocks_per_launch": blocks,
                "peak_is": "%d SIMDs x %.1f GHz / 4 cycles per wave64 vector instruction" % (SIMDS, MAX_CLOCK_HZ / 1e9),
                "kernel_ms_avg": hbm["kernel_ms_avg"], "kernel_ms_min": hbm["kernel_ms_min"],
                "kernel_source_hash": src_hash,
                "hbm": {"achieved": hbm["achieved"], "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": hbm["frac"],
                        "algorithmic_bytes_per_launch": alg_bytes},
                "note": "one lane per window: integer-VALU issue bounds SHA-256 / MD5, not HBM; the HBM fraction is kept "
                        "as a secondary figure because the metric is quoted in GB/s",
            }
            if prof_note:
                out["roofline"]["traffic_note"] = prof_note
        if world == 1 and not args.no_cpu_baseline and args.type == "CRC32":
            sample = data[: min(nbytes, 2 << 30)].cpu().numpy()
            out["cpu_baseline"] = cpu_baseline(sample, args.bpc)
            jvm = jvm_baseline(args.bpc, out["cpu_baseline"]["cores"])
            if jvm:   # the real thing beside the C restatement, when a JDK happens to exist
                out["cpu_baseline"]["jvm"] = jvm
        elif world == 1 and not args.no_cpu_baseline:
            sample = data[: min(nbytes, (2 << 30) if args.type == "CRC32C" else (1 << 30))].cpu().numpy()
            out["cpu_baseline"] = cpu_baseline_other(args.type, type_id, sample, args.bpc)
        elif world == 1:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
