"""world_size-2 gloo tests (CPU) of the multi-GPU host logic: batch sharding and the single
broadcast the forward path needs.  The convolution itself is not run here (no GPU)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from fft_conv_pytorch_amd.distributed import broadcast_buffer, shard_range


def test_shard_range_partitions_exactly():
    for total in (1, 7, 8, 32, 64, 65):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            for (a0, a1), (b0, b1) in zip(spans, spans[1:]):
                assert a1 == b0 and a1 >= a0
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(8, 2, 2)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # rank 0 owns the "kernel spectrum"; the others only know its size
        buf = broadcast_buffer(lambda: torch.arange(1024, dtype=torch.float32) * 0.5,
                               lambda: torch.empty(1024, dtype=torch.float32), src=0)
        ok_bcast = torch.equal(buf, torch.arange(1024, dtype=torch.float32) * 0.5)
        # weak-scaling bookkeeping as in bench.py: per-rank work units, max-over-ranks time
        lo, hi = shard_range(33, world, rank)
        units = torch.tensor([float(hi - lo)])
        dist.all_reduce(units, op=dist.ReduceOp.SUM)
        t = torch.tensor([1.0 + rank], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        out[rank] = (ok_bcast, float(units), float(t))
    finally:
        dist.destroy_process_group()


def test_broadcast_and_bookkeeping_world2_gloo():
    world = 2
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
        assert len(out) == world
        for rank in range(world):
            ok, units, tmax = out[rank]
            assert ok and units == 33.0 and tmax == 2.0
