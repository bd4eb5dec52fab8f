#!/usr/bin/env python3
"""Round-4 golden vectors from the REFERENCE decoder (build container only; loader in tools/make_goldens.py).

    PYTHONDONTWRITEBYTECODE=1 python tools/make_goldens_r4.py

  grads_8x176     `loss_t` (unitspeech/unitspeech.py:393-405) under torch autograd on 8 ragged 176-frame crops of the full-size decoder -- the
                  inputs of tests/test_hip_parity_r3.py::_crops(8, 176, key=33) -- in fp32 AND in fp64.  Stored per parameter tensor: the
                  fp64 norm, the reference's own fp32-vs-fp64 distance (its noise floor: the bar a per-tensor comparison can be held to),
                  and for every tensor of at most 1,024 elements (the eight Rezero gains, biases, GroupNorm affine parameters) the fp64
                  values themselves; larger tensors as an odd-strided 4,096-element fp64 sample.
"""
from __future__ import annotations

import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_goldens import FULL, ReplayRandn, build, load_reference, save  # noqa: E402

SMALL = 1024
SAMPLE = 4096


def crops(B, T, key):
    """tests/test_hip_parity_r3.py::_crops, restated (the generator's draws in the same order)."""
    g = np.random.Generator(np.random.Philox(key=key))
    x0 = torch.from_numpy(g.standard_normal((B, FULL.n_feats, T), dtype=np.float32)).clamp(-1, 1)
    cond = torch.from_numpy(g.standard_normal((B, FULL.n_feats, T), dtype=np.float32) * 0.5)
    lengths = [T - 8 * (b % 3) for b in range(B)]
    mask = torch.zeros(B, 1, T)
    for b, n in enumerate(lengths):
        mask[b, 0, :n] = 1.0
    spk = torch.from_numpy(g.standard_normal((B, 1, FULL.spk_emb_dim), dtype=np.float32))
    spk = spk / spk.norm(dim=-1, keepdim=True)
    t = torch.from_numpy(g.uniform(0.05, 0.95, size=(B,)).astype(np.float32))
    z = torch.from_numpy(g.standard_normal((B, FULL.n_feats, T), dtype=np.float32))
    return x0, mask, cond, spk, t, z


def sample_stride(numel):
    return (numel // SAMPLE + 1) | 1


def grads(U, dtype, args):
    x0, mask, cond, spk, t, z = (a.to(dtype) for a in args)
    m = build(U, FULL, 0, dtype).train()
    t0 = time.time()
    with ReplayRandn([z]):
        loss, _ = m.loss_t(x0, mask, cond, t, spk)
    loss.backward()
    print(f"   {dtype}: loss {float(loss):.9f}  ({time.time() - t0:.0f} s)", flush=True)
    return float(loss), {n: p.grad.detach().double() for n, p in m.named_parameters() if p.grad is not None}


def main():
    torch.set_num_threads(8)
    U = load_reference()
    args = crops(8, 176, key=33)
    l32, g32 = grads(U, torch.float32, args)
    l64, g64 = grads(U, torch.float64, args)
    assert len(g64) == 228 and set(g32) == set(g64)
    out = {"loss_fp32": l32, "loss_fp64": l64, "names": np.array(sorted(g64))}
    worst = []
    for i, n in enumerate(sorted(g64)):
        a, b = g32[n], g64[n]
        out[f"norm_{i}"] = float(b.norm())
        floor = float((a - b).norm() / (b.norm() + 1e-300))
        out[f"floor_{i}"] = floor
        worst.append((floor, n, b.numel()))
        if b.numel() <= SMALL:
            out[f"val_{i}"] = b.numpy()
            out[f"val32_{i}"] = a.numpy()
        else:
            out[f"val_{i}"] = b.reshape(-1)[::sample_stride(b.numel())].numpy()
    worst.sort()
    print("   the reference's own fp32-vs-fp64 distance per tensor, largest ten:")
    for f, n, k in worst[-10:]:
        print(f"     {f:.2e}  {n} ({k} el.)")
    whole = float(torch.sqrt(sum(((g32[n] - g64[n]) ** 2).sum() for n in g64)) / torch.sqrt(sum((g64[n] ** 2).sum() for n in g64)))
    out["whole_floor"] = whole
    print(f"   whole-gradient fp32-vs-fp64 relative L2 {whole:.2e}")
    save("grads_full_8x176_fp64", **out)


if __name__ == "__main__":
    main()
