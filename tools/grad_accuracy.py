#!/usr/bin/env python3
"""Relative error of every parameter gradient of one full-size fine-tune loss (B crops of 176 frames) between the f16x3 backward
(default) and the exact-fp32 MFMA backward (US_F16X3=0), same weights, same draws: how much the two-plane fp16 operands cost where
gradients are small (a mean-reduced loss puts dL/dy around 1/(B*80*176)).   python tools/grad_accuracy.py [--batch 1]"""
import argparse
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np
import torch

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=1)
a = ap.parse_args()
from unitspeech_amd import DecoderConfig, UnitSpeech, synthetic_state_dict  # noqa: E402

cfg = DecoderConfig()
sd = {k: torch.from_numpy(v) for k, v in synthetic_state_dict(cfg, 0).items()}
g = np.random.Generator(np.random.Philox(key=99))
B, T = a.batch, 176
x0 = torch.from_numpy(g.standard_normal((B, cfg.n_feats, T), dtype=np.float32)).clamp(-1, 1).cuda()
cond = torch.from_numpy(g.standard_normal((B, cfg.n_feats, T), dtype=np.float32) * 0.5).cuda()
mask = torch.ones(B, 1, T).cuda()
spk = torch.from_numpy(g.standard_normal((B, 1, cfg.spk_emb_dim), dtype=np.float32)).cuda()
spk = spk / spk.norm(dim=-1, keepdim=True)


def grads(env):
    os.environ.update(env)
    m = UnitSpeech(cfg.n_feats, cfg.dim, list(cfg.dim_mults), cfg.beta_min, cfg.beta_max, cfg.pe_scale, cfg.spk_emb_dim)
    m.load_state_dict(sd)
    m = m.cuda().train()
    random.seed(0); torch.manual_seed(0)
    loss, _ = m.compute_loss(x0, mask, cond, spk)
    loss.backward()
    torch.cuda.synchronize()
    out = {n: p.grad.detach().double().cpu() for n, p in m.named_parameters() if p.grad is not None}
    for k in env:
        os.environ.pop(k)
    return float(loss), out


l_ref, ref = grads({"US_F16X3": "0"})
l_new, new = grads({})
rel = {n: float((new[n] - ref[n]).norm() / (ref[n].norm() + 1e-300)) for n in ref}
worst = sorted(rel.items(), key=lambda kv: -kv[1])[:8]
tot = float(torch.sqrt(sum(((new[n] - ref[n]) ** 2).sum() for n in ref)) / torch.sqrt(sum((ref[n] ** 2).sum() for n in ref)))
print(f"B={B}: loss fp32 {l_ref:.7f} f16x3 {l_new:.7f}; {len(ref)} gradients; whole-gradient relative L2 error {tot:.3e}; median per tensor "
      f"{float(np.median(list(rel.values()))):.3e}; typical |gy| scale 1/(B*80*176) = {1.0 / (B * 80 * 176):.1e}")
for n, v in worst:
    print(f"   {v:.3e}  {n}  (|g| = {float(ref[n].norm()):.3e})")
