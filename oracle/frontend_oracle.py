"""CPU restatement of the conditioning producer's two learned modules -- TEST INFRASTRUCTURE ONLY.

Only `tests/`, `__graft_entry__.smoke()` and the benches' `cpu_baseline` legs may import this file; the product
(`unitspeech_amd/`) never does (tests/test_cabi.py::test_product_does_not_import_oracle).

Functional (state_dict in, tensors out) restatement of
  * `Encoder.forward`                       /root/reference/unitspeech/encoder.py:294-308 (+ :12-30 LayerNorm, :33-65 ConvReluNorm,
                                            :68-187 MultiHeadAttention with relative positions, :190-211 FFN, :214-250 EncoderModule)
  * `DurationPredictor.forward(reverse=True)`  /root/reference/unitspeech/duration_predictor.py:47-63 (+ :9-21 LayerNorm)
in eval mode (every Dropout is the identity).  The relative-position terms are written with explicit offsets
(score[i, j] += q_i . rel_k[j - i + W] for |j - i| <= W) instead of the reference's pad-and-reshape skewing (:168-182),
so the two are independent statements of the same arithmetic.  Pinned by tests/golden/frontend_{tiny,full}.npz, which
tools/make_goldens_frontend.py generated from the reference modules themselves.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


def sequence_mask(lengths, max_length):
    ar = torch.arange(int(max_length), dtype=lengths.dtype, device=lengths.device)
    return ar.unsqueeze(0) < lengths.unsqueeze(1)


def channel_layer_norm(x, gamma, beta, eps):
    """encoder.py:21-30: statistics over the channel axis of [B, C, T]."""
    mean = x.mean(1, keepdim=True)
    var = ((x - mean) ** 2).mean(1, keepdim=True)
    return (x - mean) * torch.rsqrt(var + eps) * gamma.view(1, -1, 1) + beta.view(1, -1, 1)


def _conv(sd, prefix, x, pad):
    return F.conv1d(x, sd[prefix + ".weight"], sd[prefix + ".bias"], padding=pad)


def relative_attention(sd, prefix, x, attn_mask, n_heads, window):
    """MultiHeadAttention.forward(x, x, attn_mask) (encoder.py:105-144), self-attention, heads_share=True."""
    q, k, v = (_conv(sd, f"{prefix}.conv_{n}", x, 0) for n in "qkv")
    b, d, t = q.shape
    kc = d // n_heads
    q = q.view(b, n_heads, kc, t).transpose(2, 3)
    k = k.view(b, n_heads, kc, t).transpose(2, 3)
    v = v.view(b, n_heads, kc, t).transpose(2, 3)
    scores = torch.matmul(q, k.transpose(-2, -1)) / math.sqrt(kc)
    if window is not None:
        rel_k, rel_v = sd[prefix + ".emb_rel_k"][0], sd[prefix + ".emb_rel_v"][0]        # [2W+1, kc]
        ar = torch.arange(t)
        off = ar.unsqueeze(0) - ar.unsqueeze(1)                                       # off[i, j] = j - i
        inside = off.abs() <= window
        idx = (off + window).clamp(0, 2 * window)
        q_rel = torch.matmul(q, rel_k.t())                                            # [b, h, t, 2W+1]
        local = torch.gather(q_rel, 3, idx.view(1, 1, t, t).expand(b, n_heads, t, t)) * inside
        scores = scores + local / math.sqrt(kc)
    scores = scores.masked_fill(attn_mask == 0, -1e4)
    p = torch.softmax(scores, dim=-1)
    out = torch.matmul(p, v)
    if window is not None:
        # weight of relative offset r - W for query i is p[i, i + r - W] (zero outside the sequence)
        rel_w = torch.zeros(b, n_heads, t, 2 * window + 1, dtype=p.dtype)
        for r in range(2 * window + 1):
            j = ar + (r - window)
            ok = (j >= 0) & (j < t)
            rel_w[:, :, ok, r] = p[:, :, ar[ok], j[ok]]
        out = out + torch.matmul(rel_w, rel_v)
    out = out.transpose(2, 3).contiguous().view(b, d, t)
    return _conv(sd, prefix + ".conv_o", out, 0)


def encoder_forward(sd, ids, lengths, *, n_heads, n_layers, kernel_size, window_size, prenet_layers=3):
    """-> (mu_x [B, n_feats, L], x [B, C, L], x_mask [B, 1, L]); encoder.py:294-308."""
    emb = sd["emb.weight"]
    c = emb.shape[1]
    x = (emb[ids] * math.sqrt(c)).transpose(1, -1)
    x_mask = sequence_mask(lengths, x.shape[2]).unsqueeze(1).to(x.dtype)
    # prenet (ConvReluNorm, :58-65)
    x_org = x
    for i in range(prenet_layers):
        w = sd[f"prenet.conv_layers.{i}.weight"]
        x = _conv(sd, f"prenet.conv_layers.{i}", x * x_mask, w.shape[2] // 2)
        x = channel_layer_norm(x, sd[f"prenet.norm_layers.{i}.gamma"], sd[f"prenet.norm_layers.{i}.beta"], 1e-4)
        x = torch.relu(x)
    x = (x_org + _conv(sd, "prenet.proj", x, 0)) * x_mask
    # transformer blocks (EncoderModule, :239-250)
    attn_mask = x_mask.unsqueeze(2) * x_mask.unsqueeze(-1)
    for i in range(n_layers):
        x = x * x_mask
        y = relative_attention(sd, f"encoder.attn_layers.{i}", x, attn_mask, n_heads, window_size)
        x = channel_layer_norm(x + y, sd[f"encoder.norm_layers_1.{i}.gamma"], sd[f"encoder.norm_layers_1.{i}.beta"], 1e-4)
        y = _conv(sd, f"encoder.ffn_layers.{i}.conv_1", x * x_mask, kernel_size // 2)
        y = torch.relu(y)
        y = _conv(sd, f"encoder.ffn_layers.{i}.conv_2", y * x_mask, kernel_size // 2) * x_mask
        x = channel_layer_norm(x + y, sd[f"encoder.norm_layers_2.{i}.gamma"], sd[f"encoder.norm_layers_2.{i}.beta"], 1e-4)
    x = x * x_mask
    mu_x = _conv(sd, "proj_m", x, 0) * x_mask
    return mu_x, x, x_mask


def duration_predictor_forward(sd, x, x_mask, g=None):
    """logw [B, 1, L] = DurationPredictor.forward(x, x_mask, w=None, g=g, reverse=True); duration_predictor.py:47-63."""
    if g is not None:
        x = torch.cat([x, g.transpose(1, 2).repeat(1, 1, x.shape[-1])], dim=1)
    k = sd["conv_1.weight"].shape[2]
    for i in (1, 2):
        x = torch.relu(_conv(sd, f"conv_{i}", x * x_mask, k // 2))
        x = channel_layer_norm(x, sd[f"norm_{i}.gamma"], sd[f"norm_{i}.beta"], 1e-5)
    return _conv(sd, "proj", x * x_mask, 0) * x_mask
