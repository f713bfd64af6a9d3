"""First contact with the GPU: loads the library next to torch, checks a few shapes against the oracle."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import oracle
from hadoop_ozone_amd import native

L = native.lib()
print("available", L.ozcs_available(), "devices", L.ozcs_device_count(), flush=True)
ok = True
for t in (native.CRC32, native.CRC32C, native.SHA256, native.MD5):
    for n, bpc in ((9, 9), (55, 10), (100, 20), (3072, 2048), (4 << 20, 1 << 20), ((2 << 20) + 12345, 1 << 20),
                   (20 * 16384 + 100, 16384), (1 << 20, 512), (3 * 65552 + 9, 65552)):
        d = oracle.splitmix64(n + t, n)
        t0 = time.time()
        got = native.compute(t, bpc, d)
        dt = time.time() - t0
        exp = oracle.compute(t, bpc, d)
        good = got == exp
        ok &= good
        print("type", t, "n", n, "bpc", bpc, "OK" if good else "MISMATCH", "%.1f ms" % (dt * 1e3), flush=True)
        if not good:
            ds = native.DIGEST_SIZE[t]
            bad = [i for i in range(len(exp) // ds) if got[i*ds:(i+1)*ds] != exp[i*ds:(i+1)*ds]]
            print("   bad windows", bad[:10], "of", len(exp) // ds, got[:ds].hex(), exp[:ds].hex())
print("ALL OK" if ok else "FAILURES")
# device-resident timing, config-2 shape
dev = torch.device("cuda:0")
for blocks in (1, 4, 16):
    nbytes = blocks * (256 << 20)
    x = torch.randint(0, 256, (nbytes,), dtype=torch.uint8, device=dev)
    nwin = nbytes >> 20
    sums = torch.empty(nwin * 4, dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    def run():
        native.check(L.ozcs_compute_device(0, native.CRC32, 1 << 20, x.data_ptr(), 4 << 20, nbytes >> 22, 4 << 20,
                                           sums.data_ptr(), nwin * 4, st))
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print("blocks", blocks, "%.3f ms  %.1f GB/s" % (ms, nbytes / ms / 1e6), flush=True)
    # sample check
    h = x[: 8 << 20].cpu().numpy()
    exp = oracle.compute(native.CRC32, 1 << 20, h)
    print("   sample parity", bytes(sums[:32].cpu().numpy()) == exp)
    del x, sums
sys.exit(0 if ok else 1)
