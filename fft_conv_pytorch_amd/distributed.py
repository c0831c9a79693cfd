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


def _plan_signatures_agree(plan, device, group) -> bool:
    """True when every rank's plan lays the kernel spectrum out identically (same bytes, tile, dilation
    handling, segmentation, channel blocking).  The 1-D planner looks at the LOCAL batch, and shards may
    differ by one item (33 items on 2 ranks -> 17 + 16), so this is checked, not assumed."""
    sig = torch.tensor(plan.signature(), dtype=torch.int64, device=device)
    world = dist.get_world_size(group)
    gathered = [torch.empty_like(sig) for _ in range(world)]
    dist.all_gather(gathered, sig, group=group)
    return all(torch.equal(g, gathered[0]) for g in gathered)


def broadcast_kernel_spectrum(plan, kernel: torch.Tensor, src: int = 0, group: Optional[dist.ProcessGroup] = None):
    """Transform the kernel on rank ``src`` only and broadcast the spectrum (not the raw weights'
    full-length FFT, which the reference would have needed: 8.4 MB at the metric configuration).

    Collective: every rank of ``group`` must call it.  When the ranks' plans disagree on the spectrum
    layout (unequal shards can make the planner pick another tile or dilation handling), a foreign
    spectrum would be garbage or the wrong size: every rank then transforms its own copy of ``kernel``
    instead -- all ranks reach the same decision from the same all-gathered signatures, so nobody is
    left waiting in a broadcast."""
    from . import functional as F_

    comm_device = kernel.device if dist.get_backend(group) == "nccl" else torch.device("cpu")
    if not _plan_signatures_agree(plan, comm_device, group):
        return F_.transform_kernel(plan, kernel)

    holder = {}

    def make_local():
        holder["spec"] = F_.transform_kernel(plan, kernel)
        return holder["spec"].buf

    def like():
        holder["spec"] = F_.KernelSpectrum(
            plan, torch.empty(max(plan.spectrum_bytes, 16) // 4, dtype=torch.float32, device=kernel.device))
        return holder["spec"].buf

    broadcast_buffer(make_local, like, src=src, group=group)
    return holder["spec"]


def fft_conv_sharded(signal_shard: torch.Tensor, kernel: torch.Tensor, bias=None, stride=1, padding=0, dilation=1,
                     groups: int = 1, padding_mode: str = "constant", src: int = 0,
                     group: Optional[dist.ProcessGroup] = None) -> torch.Tensor:
    """Forward convolution of this rank's batch shard; the kernel spectrum comes from rank ``src``
    (every rank holds the same ``kernel``, as after loading one checkpoint)."""
    from . import functional as F_

    plan = F_._plan_for(signal_shard, kernel, bias, stride, padding, dilation, groups, padding_mode)
    spectrum = broadcast_kernel_spectrum(plan, kernel, src=src, group=group)
    return F_._forward_native(signal_shard, spectrum, bias)
