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


# ---- spectrum broadcast vs local transform: decided by the all-gathered plan signatures
class _FakePlan:
    def __init__(self, sig, nbytes=64):
        self._sig, self.spectrum_bytes, self.workspace_bytes = sig, nbytes, 0

    def signature(self):
        return (self.spectrum_bytes,) + self._sig


def _sig_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from fft_conv_pytorch_amd import distributed as D, functional as F_
        made = []

        def fake_transform(plan, kernel):
            made.append(1)
            return F_.KernelSpectrum(plan, torch.full((16,), float(rank + 1)))

        F_.transform_kernel = fake_transform
        kernel = torch.zeros(4)
        # equal layouts on both ranks: rank 0 transforms, rank 1 receives rank 0's spectrum
        spec = D.broadcast_kernel_spectrum(_FakePlan((1024, 1, 1, 512, 0, 0, 0, 4)), kernel, src=0)
        same = (len(made), float(spec.buf[0]))
        # rank 1's planner picked another tile (unequal shards can do that): nobody broadcasts, everyone transforms
        made.clear()
        spec = D.broadcast_kernel_spectrum(_FakePlan((1024 if rank == 0 else 2048, 1, 1, 512, 0, 0, 0, 4)), kernel, src=0)
        differ = (len(made), float(spec.buf[0]))
        # equal tile but different byte size must not be broadcast either
        made.clear()
        spec = D.broadcast_kernel_spectrum(_FakePlan((1024, 1, 1, 512, 0, 0, 0, 4), nbytes=64 + 64 * rank), kernel, src=0)
        size = (len(made), float(spec.buf[0]))
        out[rank] = (same, differ, size)
    finally:
        dist.destroy_process_group()


def test_spectrum_broadcast_needs_agreeing_plan_signatures_world2_gloo():
    world = 2
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_sig_worker, args=(world, _free_port(), out), nprocs=world, join=True)
        assert out[0] == ((1, 1.0), (1, 1.0), (1, 1.0))
        assert out[1] == ((0, 1.0), (1, 2.0), (1, 2.0))     # received once, then two local transforms


# ---- strong scaling as bench.py --scaling strong does it: the node-level batch of a BASELINE.json problem is split over
# the ranks; the whole-problem throughput is (outputs of ALL ranks) / (max over ranks of the time)
def _strong_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import importlib.util
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
        bench = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(bench)
        res = {}
        for name, total in bench.STRONG_BATCH.items():
            lo, hi = shard_range(total, world, rank)
            mine = torch.zeros(total, dtype=torch.int64)
            mine[lo:hi] = 1                       # batch items this rank convolves
            dist.all_reduce(mine, op=dist.ReduceOp.SUM)
            # per-item outputs x the whole batch, over the slowest rank's time: what `value` is made of
            per_item, t_rank = 1000, torch.tensor([0.5 * (hi - lo)], dtype=torch.float64)
            dist.all_reduce(t_rank, op=dist.ReduceOp.MAX)
            res[name] = (bool((mine == 1).all()), hi - lo, per_item * total / float(t_rank))
        out[rank] = res
    finally:
        dist.destroy_process_group()


def test_strong_scaling_split_world2_gloo():
    world = 2
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_strong_worker, args=(world, _free_port(), out), nprocs=world, join=True)
        assert len(out) == world
        for name in ("cfgA", "cfgB", "cfgC", "cfgD"):
            covered0, n0, v0 = out[0][name]
            covered1, n1, v1 = out[1][name]
            assert covered0 and covered1, "every batch item convolved exactly once"
            assert abs(n0 - n1) <= 1 and v0 == v1          # balanced shards, one agreed throughput figure
