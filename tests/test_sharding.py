"""Multi-GPU path on CPU: ranks shard the block list contiguously, exchange nothing on the data path, and the
whole-job rate comes from the slowest rank (gloo, world_size 2).  The per-rank work here is the oracle; on the
GPU box the same bench.py helpers drive the HIP library."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import bench
import oracle


def test_shard_blocks_partitions_exactly():
    for total in (0, 1, 7, 8, 64, 257):
        for world in (1, 2, 3, 4, 8):
            spans = [bench.shard_blocks(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def test_aggregate_uses_slowest_rank():
    assert bench.aggregate(10 ** 9, 4, 2.0, 8) == 16.0   # 8 ranks x 4 steps x 1 GB in 2 s


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_chunks, chunk_len, bpc, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = bench.shard_blocks(n_chunks, rank, world)
    # every rank regenerates only its own chunks (as each GPU holds only its shard)
    sums = b"".join(oracle.compute(oracle.CRC32, bpc, oracle.splitmix64(1000 + c, chunk_len)) for c in range(lo, hi))
    dist.barrier()
    elapsed = torch.tensor([0.5 + rank], dtype=torch.float64)   # rank 1 is "slower"
    dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    # results are gathered by plain copies (here: files), never by a collective on the data path
    np.save(os.path.join(out_dir, "sums_%d.npy" % rank), np.frombuffer(sums, dtype=np.uint8))
    if rank == 0:
        np.save(os.path.join(out_dir, "elapsed.npy"), elapsed.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_shard_by_chunk(tmp_path):
    n_chunks, chunk_len, bpc, world = 7, 300000, 65536, 2
    mp.spawn(_worker, args=(world, _free_port(), n_chunks, chunk_len, bpc, str(tmp_path)), nprocs=world, join=True)
    got = b"".join(np.load(str(tmp_path / ("sums_%d.npy" % r))).tobytes() for r in range(world))
    exp = b"".join(oracle.compute(oracle.CRC32, bpc, oracle.splitmix64(1000 + c, chunk_len)) for c in range(n_chunks))
    assert got == exp
    assert float(np.load(str(tmp_path / "elapsed.npy"))[0]) == 1.5


def _cuts(lens, parts):
    import ctypes
    from hadoop_ozone_amd import native
    L = native.lib()   # pure function: no GPU call is made
    arr = (native.OzcsChunk * max(1, len(lens)))()
    for i, n in enumerate(lens):
        arr[i].len = n
    cuts = (ctypes.c_uint32 * (parts + 1))()
    L.ozcs_shard_chunks(arr, len(lens), parts, cuts)
    return list(cuts)


def test_batch_is_sharded_over_devices_by_bytes():
    # equal chunks: equal counts
    assert _cuts([4 << 20] * 8, 4) == [0, 2, 4, 6, 8]
    assert _cuts([4 << 20] * 64, 8) == list(range(0, 65, 8))
    # one large chunk in front: it gets a device to itself, the small ones share the other
    assert _cuts([64 << 20] + [1 << 20] * 64, 2) == [0, 1, 65]
    # mixed sizes: every part within one chunk of its byte share, parts contiguous and complete
    rng = np.random.default_rng(3)
    for parts in (2, 3, 8):
        lens = [int(x) for x in rng.integers(1, 8 << 20, 300)]
        cuts = _cuts(lens, parts)
        assert cuts[0] == 0 and cuts[-1] == len(lens) and all(a <= b for a, b in zip(cuts, cuts[1:]))
        share = sum(lens) / parts
        for d in range(parts):
            assert abs(sum(lens[cuts[d]:cuts[d + 1]]) - share) <= max(lens)
        by_count = [len(lens) * d // parts for d in range(parts + 1)]
        worst = max(sum(lens[cuts[d]:cuts[d + 1]]) for d in range(parts))
        assert worst <= max(sum(lens[by_count[d]:by_count[d + 1]]) for d in range(parts)) + max(lens)
    # degenerate lists
    assert _cuts([], 4) == [0, 0, 0, 0, 0]
    assert _cuts([0, 0, 0, 0], 2) == [0, 2, 4]
    assert _cuts([5], 3) == [0, 1, 1, 1]
    assert _cuts([1, 1], 1) == [0, 2]


def test_bench_starts_its_own_ranks(tmp_path):
    # `python bench.py --gpus 2` with no launcher around it: the ranks are started as a child before any GPU call.
    # Without a GPU every rank stops at "needs a GPU"; what is checked here is that two ranks were started.
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--blocks", "1"], env=env, capture_output=True, text=True, timeout=170, cwd=str(tmp_path))
    if torch.cuda.is_available():
        assert r.returncode == 0, r.stderr[-2000:]
    else:
        assert r.returncode != 0
        assert "bench.py needs a GPU" in r.stderr and "local_rank: 1" in r.stderr, r.stderr[-2000:]
