"""Multi-GPU use of the forward path: one process per GPU, batch-sharded, no data-path collective.

Every (batch item, channel group) unit of the forward convolution is independent
(/root/reference/fft_conv_pytorch/functional.py:11-16 contracts only over the input
channels of a group), so N GPUs simply convolve N disjoint batch shards.  The only
exchange is the one-off broadcast of the transformed kernel from one rank per weight
version (RCCL over xGMI when the backend is "nccl"; about 0.5-1 MB for a 1-D 8x8
kernel at a 2048-point tile), after which ranks never talk during forward.
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(total: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous, balanced [start, stop) slice of ``total`` batch items for ``rank``."""
    if not (0 <= rank < world_size):
        raise ValueError(f"rank {rank} outside world of size {world_size}")
    base, extra = divmod(total, world_size)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def broadcast_buffer(make_local: Callable[[], torch.Tensor], like: Callable[[], torch.Tensor], src: int = 0,
                     group: Optional[dist.ProcessGroup] = None) -> torch.Tensor:
    """Rank ``src`` builds a buffer with ``make_local``; every other rank allocates ``like()`` and
    receives it.  Returns the (now identical) buffer on every rank."""
    rank = dist.get_rank(group)
    buf = make_local() if rank == src else like()
    dist.broadcast(buf, src=src, group=group)
    return buf


def broadcast_kernel_spectrum(plan, kernel: torch.Tensor, src: int = 0, group: Optional[dist.ProcessGroup] = None):
    """Transform the kernel on rank ``src`` only and broadcast the spectrum (not the raw weights'
    full-length FFT, which the reference would have needed: 8.4 MB at the metric configuration)."""
    from . import functional as F_

    holder = {}

    def make_local():
        holder["spec"] = F_.transform_kernel(plan, kernel)
        return holder["spec"].buf

    def like():
        holder["spec"] = F_.KernelSpectrum(
            plan, torch.empty(max(plan.spectrum_bytes, 16) // 4, dtype=torch.float32, device=kernel.device),
            torch.empty(plan.workspace_bytes // 4, dtype=torch.float32, device=kernel.device)
            if plan.workspace_bytes else None)
        return holder["spec"].buf

    broadcast_buffer(make_local, like, src=src, group=group)
    return holder["spec"]


def fft_conv_sharded(signal_shard: torch.Tensor, kernel: torch.Tensor, bias=None, stride=1, padding=0, dilation=1,
                     groups: int = 1, padding_mode: str = "constant", src: int = 0,
                     group: Optional[dist.ProcessGroup] = None) -> torch.Tensor:
    """Forward convolution of this rank's batch shard; the kernel spectrum comes from rank ``src``."""
    from . import functional as F_

    plan = F_._plan_for(signal_shard, kernel, bias, stride, padding, dilation, groups, padding_mode)
    spectrum = broadcast_kernel_spectrum(plan, kernel, src=src, group=group)
    return F_._forward_native(signal_shard, spectrum, bias)
