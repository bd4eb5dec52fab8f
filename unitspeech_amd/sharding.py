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


def broadcast_state_dict(cfg: DecoderConfig, sd_on_rank0, rank: int, world: int, device, timing: dict = None) -> "OrderedDict[str, torch.Tensor]":
    """Rank 0 holds `sd_on_rank0`; every rank returns views into its copy of the packed blob.  `timing` (optional dict) receives
    `broadcast_ms` -- the collective alone, between device synchronisations, slowest rank's view after the caller's max-reduce is
    not needed: the broadcast ends everywhere within one ring pass -- and `bytes`."""
    import time
    total = sum(int(np.prod(s)) for s in param_shapes(cfg).values())
    if rank == 0:
        flat = pack_state_dict(cfg, sd_on_rank0, device)
    else:
        flat = torch.empty(total, dtype=torch.float32, device=device)
    if timing is not None:
        timing["bytes"] = int(total) * 4
        timing["broadcast_ms"] = None
    if world > 1:
        import torch.distributed as dist
        dev = torch.device(device)
        if dev.type == "cuda":
            torch.cuda.synchronize(dev)
        dist.barrier()
        t0 = time.perf_counter()
        dist.broadcast(flat, src=0)
        if dev.type == "cuda":
            torch.cuda.synchronize(dev)
        if timing is not None:
            timing["broadcast_ms"] = 1e3 * (time.perf_counter() - t0)
    return unpack_state_dict(cfg, flat)


def max_over_ranks(value: float, world: int, device) -> float:
    if world <= 1:
        return float(value)
    import torch.distributed as dist
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def allreduce_gradients(params, world: int, blob: "torch.Tensor | None" = None, group=None) -> int:
    """Data-parallel gradient averaging for the decoder's training step (pre-training shards the batch
    over the GPUs of a node; the reference trains on one GPU).  Gradients that are views into `blob` (the single buffer
    `us_estimator_backward` writes, `model._engine.last_grad_blob`) travel as ONE all-reduce of that buffer -- 476 MB for the
    full-size decoder, i.e. one large ring pass over xGMI instead of 230 latency-bound ones; every other gradient is flattened
    into a second, small buffer.  Returns the number of collectives issued.  In place; averages (sum / world)."""
    if world <= 1:
        return 0
    import torch.distributed as dist
    grads = [p.grad for p in params if p.grad is not None]
    n_coll = 0
    loose = grads
    if blob is not None:
        lo = blob.data_ptr()
        hi = lo + blob.numel() * blob.element_size()
        inside = [g for g in grads if g.device == blob.device and lo <= g.data_ptr() < hi and g.is_contiguous()]
        if inside:
            dist.all_reduce(blob, group=group)
            blob.div_(world)
            n_coll += 1
            ids = {id(g) for g in inside}
            loose = [g for g in grads if id(g) not in ids]
    if loose:
        flat = torch.cat([g.reshape(-1).to(torch.float32) for g in loose])
        dist.all_reduce(flat, group=group)
        flat.div_(world)
        n_coll += 1
        off = 0
        for g in loose:
            n = g.numel()
            g.copy_(flat[off:off + n].view_as(g))
            off += n
    return n_coll
