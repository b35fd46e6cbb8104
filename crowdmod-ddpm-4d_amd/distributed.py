"""Multi-GPU sampling: batch-shard the independent chains, one process per GPU.

The reference has no distributed code.  Sampling chains are
independent -- GroupNorm and attention are per sample (layers.py:15-16,30) -- so the
only collective is one gather of the finished samples: `torch.distributed` all_gather
(backend "nccl" == RCCL over xGMI on the MI355X node, "gloo" in the CPU tests).  Each
chain's noise is addressed by its GLOBAL sample index, so the gathered result is
bit-identical to the single-process run whatever the world size.
"""
from __future__ import annotations

from typing import Callable, Tuple

import numpy as np


def shard_range(global_batch: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block [lo, hi) of the global batch owned by `rank` (blocks differ by <= 1)."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world of {world}")
    base, rem = divmod(global_batch, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_samples(local, global_batch: int, rank: int, world: int, group=None):
    """All-gather the per-rank blocks (possibly of unequal length) into the full batch.

    `local` is a torch tensor (CUDA with nccl, CPU with gloo) or a numpy array (CPU)."""
    import torch
    import torch.distributed as dist
    is_np = isinstance(local, np.ndarray)
    t = torch.from_numpy(np.ascontiguousarray(local)) if is_np else local.contiguous()
    if world == 1:
        return local
    sizes = [shard_range(global_batch, r, world) for r in range(world)]
    maxn = max(hi - lo for lo, hi in sizes)
    pad = torch.zeros((maxn,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    pad[: t.shape[0]] = t
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad, group=group)
    out = torch.cat([bufs[r][: hi - lo] for r, (lo, hi) in enumerate(sizes)], dim=0)
    return out.cpu().numpy() if is_np else out


def sample_sharded(generate: Callable, past_global, global_batch: int, rank: int, world: int, group=None):
    """Run `generate(past_shard, nsamples, sample_id_base)` on this rank's block and gather.

    `generate` is e.g. `lambda p, n, base: model._generate_ddpm(p, sampler, n, sample_id_base=base)[0]`."""
    lo, hi = shard_range(global_batch, rank, world)
    local = generate(past_global[lo:hi], hi - lo, lo)
    return gather_samples(local, global_batch, rank, world, group)


# ------------------------------------------------------------------------------------
# data-parallel training (plain DP, one gradient all-reduce per step)
# ------------------------------------------------------------------------------------
def init_process_group(backend: str = None):
    """One process per GPU (torch.distributed.run sets RANK / WORLD_SIZE / MASTER_*).  Backend "nccl" is
    RCCL over xGMI on the MI355X node; "gloo" when no GPU is visible (CPU tests)."""
    import torch
    import torch.distributed as dist
    if dist.is_initialized():
        return
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        import os
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", 0)))
    dist.init_process_group(backend=backend)


def mean_over_ranks(value: float, group=None) -> float:
    """Mean of a host scalar over the ranks (fp64 all-reduce; NaN on any rank gives NaN everywhere): the
    epoch loss that drives ReduceLROnPlateau / the NaN stop in data-parallel training."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return float(value)
    dev = "cpu"
    if dist.get_backend(group) == "nccl":
        dev = torch.device("cuda", torch.cuda.current_device())
    t = torch.tensor([float(value)], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return float(t.item()) / dist.get_world_size(group)


class _DevView:
    """Zero-copy torch view of a raw device allocation through __cuda_array_interface__."""

    def __init__(self, ptr: int, numel: int):
        self.__cuda_array_interface__ = {"shape": (numel,), "typestr": "<f4", "data": (int(ptr), False), "version": 2}


class GradAverager:
    """grad <- mean over ranks of grad, in place on the library's flat gradient buffer (28.9 MB fp32 for
    the ATC model: ONE all-reduce per step, no bucketing needed at this size).  With the nccl backend the
    buffer is wrapped zero-copy and reduced by RCCL; with gloo it is staged through the host."""

    def __init__(self, group=None):
        self.group = group
        self._view = None
        self._key = None

    def __call__(self, net):
        import torch
        import torch.distributed as dist
        from . import native
        world = dist.get_world_size(self.group)
        if world == 1:
            return
        ptr, n = net.flat_grads()
        if dist.get_backend(self.group) == "nccl":
            if self._key != (ptr, n):
                self._view = torch.as_tensor(_DevView(ptr, n), device=f"cuda:{net.device}")
                self._key = (ptr, n)
            dist.all_reduce(self._view, op=dist.ReduceOp.SUM, group=self.group)
            self._view.div_(world)
            torch.cuda.current_stream(self._view.device).synchronize()
        else:
            host = np.empty(n, dtype=np.float32)
            native.check(native.lib().cm_memcpy_d2h(net.device, host.ctypes.data, ptr, host.nbytes))
            t = torch.from_numpy(host)
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
            t.div_(world)
            native.check(native.lib().cm_memcpy_h2d(net.device, ptr, host.ctypes.data, host.nbytes))
