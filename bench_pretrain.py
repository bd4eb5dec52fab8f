#!/usr/bin/env python3
"""Decoder pre-training step benchmark (reference train_STEP1.py:215-249 with the decoder's share of
compute_train_step_loss :380): `decoder.compute_loss(y, y_mask, mu_y, spk_emb)` on a batch of B crops of 176 frames ->
backward -> [data-parallel: one all-reduce of the gradient buffer over RCCL] -> clip_grad_norm_(1) + Adam(lr=1e-4) in the
HIP optimiser.  The reference trains on one GPU; with --gpus N every rank takes its own B crops (weak scaling) and the
gradients are averaged.  Synthetic data and weights.

    python bench_pretrain.py [--batch 32] [--iters 10]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench_pretrain.py --gpus N
Prints one JSON line on rank 0: crops/s over all ranks, ms per step, and where the step's time goes.
"""
import argparse
import json
import os
import time

import numpy as np
import torch

from unitspeech_amd import DecoderConfig, FusedAdam, UnitSpeech, synthetic_state_dict
from unitspeech_amd.sharding import allreduce_gradients, broadcast_state_dict, max_over_ranks


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--batch", type=int, default=32, help="crops per GPU (conf/hydra_config.py:140 batch_size)")
    ap.add_argument("--frames", type=int, default=176, help="crop length, fix_len_compatibility(2 * 22050 // 256)")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    a = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {a.gpus}")
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)   # RCCL
    cfg = DecoderConfig()
    sd = broadcast_state_dict(cfg, synthetic_state_dict(cfg, 0) if rank == 0 else None, rank, world, dev)
    model = UnitSpeech(cfg.n_feats, cfg.dim, list(cfg.dim_mults), cfg.beta_min, cfg.beta_max, cfg.pe_scale, cfg.spk_emb_dim)
    model.load_state_dict(sd, strict=True)
    model = model.to(dev).train()
    del sd
    opt = FusedAdam(model.parameters(), lr=1e-4)                        # conf/hydra_config.py:161
    g = np.random.Generator(np.random.Philox(key=7000 + rank))
    B, T = a.batch, a.frames
    y = torch.from_numpy(g.standard_normal((B, cfg.n_feats, T), dtype=np.float32)).clamp(-1, 1).to(dev)
    mu_y = torch.from_numpy(g.standard_normal((B, cfg.n_feats, T), dtype=np.float32) * 0.5).to(dev)
    lengths = torch.from_numpy(g.integers(T // 2, T + 1, size=B)).to(dev)
    y_mask = (torch.arange(T, device=dev)[None, :] < lengths[:, None]).unsqueeze(1).float()
    spk = torch.from_numpy(g.standard_normal((B, 1, cfg.spk_emb_dim), dtype=np.float32)).to(dev)
    spk = spk / spk.norm(dim=-1, keepdim=True)
    torch.manual_seed(100 + rank)
    marks = {"fwd": 0.0, "bwd": 0.0, "allreduce": 0.0, "optim": 0.0}

    def step(timed):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)] if timed else None
        if timed: ev[0].record()
        loss, _ = model.compute_loss(y, y_mask, mu_y, spk_emb=spk)
        if timed: ev[1].record()
        opt.zero_grad(set_to_none=True)
        loss.backward()
        if timed: ev[2].record()
        allreduce_gradients(list(model.parameters()), world, blob=model._get_engine().last_grad_blob)
        if timed: ev[3].record()
        opt.step(max_norm=1)
        if timed: ev[4].record()
        return loss, ev

    for _ in range(a.warmup):
        loss, _ = step(False)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    evs = []
    for _ in range(a.iters):
        loss, ev = step(True)
        evs.append(ev)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = max_over_ranks(time.perf_counter() - t0, world, dev)
    for ev in evs:
        for k, (i, j) in zip(marks, ((0, 1), (1, 2), (2, 3), (3, 4))):
            marks[k] += ev[i].elapsed_time(ev[j]) / a.iters
    assert torch.isfinite(loss), "non-finite loss"
    if rank == 0:
        eng = model._get_engine()
        flops_fwd = eng.lib.us_estimator_flops(eng.handle, T) * B
        ms = 1e3 * elapsed / a.iters
        print(json.dumps({"metric": "decoder pre-training crops/s (176-frame crops, fwd+bwd+allreduce+clip+Adam)",
                          "value": world * B * a.iters / elapsed, "unit": "crops/s", "n_gpus": world, "batch_per_gpu": B,
                          "frames": T, "iters": a.iters, "ms_per_step": ms, "scaling": "weak", "dtype": "f32", "data": "synthetic",
                          "ms_breakdown": {k: round(v, 3) for k, v in marks.items()},
                          "direct_form_tflops": 3 * flops_fwd * world / (ms * 1e-3) / 1e12, "last_loss": float(loss)}))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
