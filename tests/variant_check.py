"""Runs random batch shapes through ONE forced CRC kernel variant (OZCS_CRC_GROUP or OZCS_CRC_LPW in the
environment) and compares every digest with the oracle.  Started by test_gpu_parity.py, one process per
variant, because the overrides are read once per process.  Exit code 0 = parity."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import oracle
from hadoop_ozone_amd import native


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    digests = len(sys.argv) > 2 and sys.argv[2] == "digests"
    pow2 = len(sys.argv) > 2 and sys.argv[2] == "pow2"   # gapless batches of power-of-two windows only
    rng = np.random.default_rng(seed)
    L = native.lib()
    native.check(L.ozcs_init(0))
    st = torch.cuda.current_stream().cuda_stream
    bad = 0
    for case in range(40):
        bpc = int(rng.choice([1, 3, 16, 64, 100, 512, 1000, 1024, 4096, 4099, 16384, 65536, 100000, 1 << 20]))
        bpc = max(1, bpc + int(rng.integers(-2, 3)) * int(rng.integers(0, 2)))
        n_chunks = int(rng.integers(1, 6))
        max_len = min(40 * bpc + 17, 3 << 20)
        chunk_len = int(rng.integers(1, max_len + 1))
        stride = chunk_len + int(rng.integers(0, 40))
        off = int(rng.integers(0, 16))
        if case % 3 == 0 and bpc >= 16:   # every window 16-byte aligned: the kernels' fast path
            bpc = (bpc + 15) // 16 * 16
            stride = (stride + 15) // 16 * 16
            off = 0
            if case % 6 == 0 and bpc >= 512:   # and full windows back to back: the gapless-layout kernel
                bpc = bpc // 512 * 512
                chunk_len = max(1, chunk_len // bpc) * bpc
                stride = chunk_len
        if pow2:
            bpc = int(rng.choice([512, 2048, 4096, 8192, 16384, 32768, 65536]))
            n_chunks = int(rng.integers(1, 4))
            chunk_len = int(rng.integers(1, max(2, (3 << 20) // bpc))) * bpc
            stride = chunk_len
            off = 0
        t = oracle.CRC32 if case % 2 else oracle.CRC32C
        if digests:
            t = oracle.SHA256 if case % 4 else oracle.MD5
        host = oracle.splitmix64(seed * 1000 + case, off + n_chunks * stride)
        d = torch.from_numpy(host).cuda()
        wpc = (chunk_len + bpc - 1) // bpc
        ds = native.DIGEST_SIZE[t]
        sums = torch.zeros(wpc * n_chunks * ds + 4, dtype=torch.uint8, device="cuda")
        native.check(L.ozcs_compute_device(0, t, bpc, d.data_ptr() + off, chunk_len, n_chunks, stride,
                                           sums.data_ptr(), wpc * n_chunks * ds, st))
        torch.cuda.synchronize()
        got = bytes(sums[: wpc * n_chunks * ds].cpu().numpy())
        exp = b"".join(oracle.compute(t, bpc, host[off + c * stride: off + c * stride + chunk_len]) for c in range(n_chunks))
        if got != exp:
            bad += 1
            print("MISMATCH type=%d bpc=%d n_chunks=%d chunk_len=%d stride=%d off=%d" % (t, bpc, n_chunks, chunk_len, stride, off))
        # verify agrees with compute, and finds one flipped bit
        stored = torch.from_numpy(np.frombuffer(exp, dtype=np.uint8).copy()).cuda()
        fm = torch.zeros(n_chunks, dtype=torch.int64, device="cuda")
        c_bad = int(rng.integers(0, n_chunks))
        pos = int(rng.integers(0, chunk_len))
        d[off + c_bad * stride + pos] ^= 1 << int(rng.integers(0, 8))
        native.check(L.ozcs_verify_device(0, t, bpc, d.data_ptr() + off, chunk_len, n_chunks, stride, stored.data_ptr(),
                                          wpc * ds, wpc, 0, fm.data_ptr(), st))
        torch.cuda.synchronize()
        want = [-1] * n_chunks
        want[c_bad] = pos // bpc
        if fm.cpu().tolist() != want:
            bad += 1
            print("VERIFY MISMATCH bpc=%d chunk_len=%d got=%s want=%s" % (bpc, chunk_len, fm.cpu().tolist(), want))
    print("variant", {k: v for k, v in os.environ.items() if k.startswith("OZCS_")}, "bad =", bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
