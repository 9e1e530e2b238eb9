"""Multi-GPU layer: one process per GPU, patches sharded across ranks, ONE collective to reassemble the masks.

The reference's inference path is single-process and loops over images one at a time
(/root/reference/segmentor.py:96, pixel_latent_vector.py:63); patches are independent, so the build shards them
with no data-path collective and adds a single RCCL all-gather of the uint8 masks (2 MiB per rank at 8 patches of
512x512) -- SURVEY.md 8e.  Rank/world come from the same environment variables the reference reads
(WORLD_SIZE / RANK / LOCAL_RANK, ldiffusion.py:34-35,42).  Backend "nccl" is RCCL over xGMI on ROCm; the same code
runs on "gloo" with CPU tensors, which is how the world_size-2 CPU tests cover it.
"""
from __future__ import annotations

import os
from typing import Tuple

import torch


def world_info() -> Tuple[int, int, int]:
    """(rank, world_size, local_rank) from the torchrun / deepspeed environment."""
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def shard_range(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block [lo, hi) of `total` patches owned by `rank`; earlier ranks take the remainder."""
    if world < 1 or not (0 <= rank < world) or total < 0:
        raise ValueError(f"bad shard request total={total} rank={rank} world={world}")
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_masks(local: torch.Tensor, total: int, group=None) -> torch.Tensor:
    """All-gather per-rank blocks [n_r, ...] (any dtype, e.g. uint8 masks [n_r, H, W] or features [n_r, N, H, W]) into
    the full [total, ...] tensor on every rank, in patch order.  Even shards use one all_gather_into_tensor; ragged
    shards pad to the largest block."""
    import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized():
        if local.shape[0] != total:
            raise ValueError("gather_masks without a process group needs the full batch")
        return local
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    lo, hi = shard_range(total, rank, world)
    if local.shape[0] != hi - lo:
        raise ValueError(f"rank {rank} holds {local.shape[0]} patches, expected {hi - lo}")
    local = local.contiguous()
    if total % world == 0:
        out = torch.empty((total,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        if dist.get_backend(group) == "gloo":
            parts = list(out.chunk(world, 0))
            dist.all_gather(parts, local, group=group)
        else:
            dist.all_gather_into_tensor(out, local, group=group)
        return out
    nmax = -(-total // world)
    pad = torch.zeros((nmax,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    return torch.cat([parts[r][: shard_range(total, r, world)[1] - shard_range(total, r, world)[0]] for r in range(world)], 0)
