"""The reference's own CRC benchmarks, same shapes: one 64 MiB buffer, bytes-per-checksum 512 ... 1 MiB, CRC32 and
CRC32C (BenchMarkCRCBatch.java:52-65 calculateChunkedSumsByteArray over the whole buffer; BenchMarkCRCStreaming.java:69,
155-168 update / getValue / reset per window), one caller thread.  Three ways per shape: the host-pointer call a JNI
caller with a heap array makes (copy in, kernel, digests out), the same buffer resident on the device, and the
CPU engines of the oracle on one thread beside them (carry-less-multiply folding, and the reference's slice-by-8).

Usage: python scripts/bench_reference_shapes.py"""
import ctypes, json, os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import torch
import oracle
from hadoop_ozone_amd import native

L = native.lib(); native.check(L.ozcs_init(0))
N = 64 << 20
host = np.random.default_rng(64).integers(0, 256, N, dtype=np.uint8)
dev = torch.from_numpy(host).cuda()
stream = torch.cuda.current_stream().cuda_stream


def timed(fn, reps):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t0) / reps


for name, t in (("CRC32", native.CRC32), ("CRC32C", native.CRC32C)):
    for bpc in (512, 1024, 2048, 4096, 32768, 1048576):
        n = N // bpc
        sums = np.zeros(4 * n, dtype=np.uint8)
        dsums = torch.zeros(4 * n, dtype=torch.uint8, device="cuda")
        got = ctypes.c_uint64(0)

        def host_call():
            native.check(L.ozcs_compute(t, bpc, host.ctypes.data, N, sums.ctypes.data, sums.nbytes, ctypes.byref(got)))

        def device_call():
            native.check(L.ozcs_compute_device(0, t, bpc, dev.data_ptr(), N, 1, N, dsums.data_ptr(), dsums.numel(), stream))
            torch.cuda.synchronize()

        # twenty such calls captured into one HIP graph (the calls only enqueue kernels on the caller's stream): what a
        # caller with many device-resident buffers of this size pays per call once launches are not issued one by one
        graph, side = torch.cuda.CUDAGraph(), torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            with torch.cuda.graph(graph, stream=side):
                st = torch.cuda.current_stream().cuda_stream
                for _ in range(20):
                    native.check(L.ozcs_compute_device(0, t, bpc, dev.data_ptr(), N, 1, N, dsums.data_ptr(), dsums.numel(), st))
        torch.cuda.current_stream().wait_stream(side)

        def graph_call():
            graph.replay()
            torch.cuda.synchronize()

        th, td, tg = timed(host_call, 5), timed(device_call, 20), timed(graph_call, 5) / 20
        want = oracle.compute(t, bpc, host)
        assert bytes(sums) == want and bytes(dsums.cpu().numpy()) == want
        row = {"type": name, "bytes_per_checksum": bpc, "buffer_MiB": 64,
               "host_pointer_call_GBps": round(N / th / 1e9, 2), "host_pointer_calls_per_s": round(1 / th, 1),
               "device_resident_GBps": round(N / td / 1e9, 1), "device_resident_call_us": round(td * 1e6, 1),
               "in_a_graph_of_20_GBps": round(N / tg / 1e9, 1), "in_a_graph_of_20_call_us": round(tg * 1e6, 1),
               "cpu_one_thread_slice_by_8_GBps": round(N / oracle.bench(t, bpc, host, threads=1, reps=2) / 1e9, 2)}
        if oracle.has_clmul():
            row["cpu_one_thread_clmul_GBps"] = round(N / oracle.bench_clmul(t, bpc, host, threads=1, reps=2) / 1e9, 2)
        print(json.dumps(row), flush=True)
