"""Multi-GPU plumbing of the decoder path: utterances are independent, so the path shards statically across the
ranks of one node (one process per GPU) with a single weight broadcast and no collective inside the diffusion loop
(SURVEY.md §8(e)).  Works with any torch.distributed backend ("nccl" == RCCL on ROCm; "gloo" in the CPU tests)."""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, Tuple

import numpy as np
import torch

from .params import DecoderConfig, param_shapes


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [lo, hi) slice of `n_items` utterances owned by `rank`; sizes differ by at most one."""
    base, rem = divmod(int(n_items), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def pack_state_dict(cfg: DecoderConfig, sd: Dict[str, "np.ndarray | torch.Tensor"], device) -> torch.Tensor:
    """One flat fp32 blob in `param_shapes(cfg)` order (the payload of the weight broadcast)."""
    parts = []
    for k, shape in param_shapes(cfg).items():
        t = torch.as_tensor(sd[k], dtype=torch.float32)
        if tuple(t.shape) != tuple(shape):
            raise ValueError(f"{k}: shape {tuple(t.shape)} != {tuple(shape)}")
        parts.append(t.reshape(-1))
    return torch.cat(parts).to(device)


def unpack_state_dict(cfg: DecoderConfig, flat: torch.Tensor) -> "OrderedDict[str, torch.Tensor]":
    out, off = OrderedDict(), 0
    for k, shape in param_shapes(cfg).items():
        n = int(np.prod(shape))
        out[k] = flat[off:off + n].view(*shape)
        off += n
    if off != flat.numel():
        raise ValueError("blob size does not match the architecture")
    return out


def broadcast_state_dict(cfg: DecoderConfig, sd_on_rank0, rank: int, world: int, device) -> "OrderedDict[str, torch.Tensor]":
    """Rank 0 holds `sd_on_rank0`; every rank returns views into its copy of the packed blob."""
    total = sum(int(np.prod(s)) for s in param_shapes(cfg).values())
    if rank == 0:
        flat = pack_state_dict(cfg, sd_on_rank0, device)
    else:
        flat = torch.empty(total, dtype=torch.float32, device=device)
    if world > 1:
        import torch.distributed as dist
        dist.broadcast(flat, src=0)
    return unpack_state_dict(cfg, flat)


def max_over_ranks(value: float, world: int, device) -> float:
    if world <= 1:
        return float(value)
    import torch.distributed as dist
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
