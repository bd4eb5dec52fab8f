#!/usr/bin/env python3
"""G9: one full fine-tune iteration of the REFERENCE decoder (build container only): fine_tune -> loss.backward() ->
clip_grad_norm_(1) -> Adam(lr=2e-5).step()  (finetune.py:131-165), replaying the inputs and draws of
tests/golden/finetune_tiny.npz.  Records the total gradient norm and, for a handful of tensors, the clipped gradient and
the parameter after the step -> tests/golden/finetune_tiny_adam.npz.
    PYTHONDONTWRITEBYTECODE=1 python tools/make_golden_adam.py
"""
import os
import random
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_goldens import OUT, TINY, ReplayRandn, build, load_reference  # noqa: E402

KEYS = ["estimator.final_conv.weight", "estimator.final_conv.bias", "estimator.downs.0.0.block1.block.0.weight",
        "estimator.downs.1.0.res_conv.weight", "estimator.downs.2.1.block2.block.1.weight", "estimator.mid_attn.fn.g",
        "estimator.mid_attn.fn.fn.to_out.weight", "estimator.mid_attn.fn.fn.to_qkv.weight", "estimator.ups.0.3.conv.weight",
        "estimator.downs.0.3.conv.bias", "estimator.mlp.0.weight", "estimator.ups.2.1.mlp.1.weight"]


def main():
    U = load_reference()
    g = {k: torch.from_numpy(np.asarray(v)) for k, v in np.load(os.path.join(OUT, "finetune_tiny.npz")).items()}
    m = build(U, TINY, 0).train()
    opt = torch.optim.Adam(m.parameters(), lr=2e-5)
    random.seed(int(g["py_seed"]))
    orig_rand = torch.rand
    torch.rand = lambda *a, **k: g["t_draw"].clone()
    try:
        with ReplayRandn([g["z_draw"]]):
            loss = m.fine_tune(g["cond_x"], g["y"], g["y_mask"], g["y_lengths"], g["y"].shape[-1], g["attn"], g["spk_emb"],
                               int(g["segment_size"]), 80)
    finally:
        torch.rand = orig_rand
    assert abs(loss.item() - float(g["loss"])) < 1e-7, (loss.item(), float(g["loss"]))
    loss.backward()
    norm = torch.nn.utils.clip_grad_norm_(m.parameters(), max_norm=1)
    sd = dict(m.named_parameters())
    out = {"grad_norm": norm.detach().numpy(), "keys": np.array(KEYS)}
    for i, k in enumerate(KEYS):
        out[f"grad_{i}"] = sd[k].grad.detach().numpy().copy()
    opt.step()
    for i, k in enumerate(KEYS):
        out[f"param_{i}"] = sd[k].detach().numpy().copy()
    np.savez_compressed(os.path.join(OUT, "finetune_tiny_adam.npz"), **out)
    print(f"grad_norm={float(norm):.6f}; wrote finetune_tiny_adam.npz ({len(KEYS)} tensors)")


if __name__ == "__main__":
    main()
