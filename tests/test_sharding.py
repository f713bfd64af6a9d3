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
