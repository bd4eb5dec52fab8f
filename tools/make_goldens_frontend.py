#!/usr/bin/env python3
"""Golden vectors of the conditioning producer's learned modules from the REFERENCE classes (build container only).

    PYTHONDONTWRITEBYTECODE=1 python tools/make_goldens_frontend.py

Instantiates `unitspeech.encoder.Encoder` and `unitspeech.duration_predictor.DurationPredictor` of /root/reference (loader of
tools/make_goldens.py), loads the seeded weights of `unitspeech_amd.encoder.synthetic_*_state_dict` (so every tensor,
including the zero-initialised `prenet.proj`, is exercised), asserts that the key order equals the reference module's,
and stores inputs and outputs:
  frontend_tiny.npz   16 channels, 2 layers, 2 heads, window 4; B = 3, L = 12, lengths (12, 7, 3): a length below the
                      window exercises the sliced relative embeddings (encoder.py:154-166), ragged lengths the masks
  frontend_full.npz   conf/hydra_config.py sizes (192 / 768 / 6 layers / 2 heads / window 4; predictor 192+256 -> 256);
                      B = 2, L = 50, lengths (50, 37)
  frontend_nowin.npz  tiny, window_size=None (no relative terms), no speaker conditioning in the predictor
plus, for the end-to-end row, the reference's own `execute_text_to_speech` driven by these two modules (tiny decoder).
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_goldens import OUT, TINY, ReplayRandn, build, load_reference, save  # noqa: E402
from unitspeech_amd.encoder import (DurationPredictorConfig, EncoderConfig, synthetic_duration_predictor_state_dict,  # noqa: E402
                                    synthetic_encoder_state_dict)

TINY_E = EncoderConfig(n_vocab=20, n_feats=8, n_channels=16, filter_channels=32, n_heads=2, n_layers=2, kernel_size=3, window_size=4)
TINY_D = DurationPredictorConfig(in_channels=16, filter_channels=24, kernel_size=3, spk_emb_dim=12)
FULL_E = EncoderConfig()
FULL_D = DurationPredictorConfig()
NOWIN_E = EncoderConfig(n_vocab=20, n_feats=8, n_channels=16, filter_channels=32, n_heads=2, n_layers=2, kernel_size=3, window_size=None)
NOWIN_D = DurationPredictorConfig(in_channels=16, filter_channels=24, kernel_size=3, spk_emb_dim=0)


def modules(ecfg, dcfg, seed):
    import unitspeech.duration_predictor as D
    import unitspeech.encoder as E
    enc = E.Encoder(ecfg.n_vocab, ecfg.n_feats, ecfg.n_channels, ecfg.filter_channels, ecfg.n_heads, ecfg.n_layers, ecfg.kernel_size, 0.1,
                    window_size=ecfg.window_size)
    sd = {k: torch.from_numpy(v) for k, v in synthetic_encoder_state_dict(ecfg, seed).items()}
    assert list(sd) == list(enc.state_dict()), "encoder state_dict key order mismatch"
    enc.load_state_dict(sd, strict=True)
    dp = D.DurationPredictor(dcfg.in_channels, dcfg.filter_channels, dcfg.kernel_size, 0.1, spk_emb_dim=dcfg.spk_emb_dim)
    sd = {k: torch.from_numpy(v) for k, v in synthetic_duration_predictor_state_dict(dcfg, seed).items()}
    assert list(sd) == list(dp.state_dict()), "duration predictor state_dict key order mismatch"
    dp.load_state_dict(sd, strict=True)
    return enc.eval(), dp.eval()


def inputs(ecfg, dcfg, B, L, lengths, key):
    g = np.random.Generator(np.random.Philox(key=key))
    ids = torch.from_numpy(g.integers(0, ecfg.n_vocab, size=(B, L)).astype(np.int64))
    spk = None
    if dcfg.spk_emb_dim:
        spk = torch.from_numpy(g.standard_normal((B, 1, dcfg.spk_emb_dim), dtype=np.float32))
        spk = spk / spk.norm(dim=-1, keepdim=True)
    return ids, torch.LongTensor(lengths), spk


def one(name, ecfg, dcfg, B, L, lengths, key):
    enc, dp = modules(ecfg, dcfg, 0)
    ids, lens, spk = inputs(ecfg, dcfg, B, L, lengths, key)
    with torch.no_grad():
        mu_x, x, x_mask = enc(ids, lens)
        logw = dp(x, x_mask, w=None, g=spk, reverse=True)
    extra = {} if spk is None else {"spk_emb": spk}
    save(name, ids=ids, lengths=lens, mu_x=mu_x, x=x, x_mask=x_mask, logw=logw, seed=0, key=key, **extra)
    print(f"{name}: mean|mu_x|={mu_x.abs().mean():.3f} mean|x|={x.abs().mean():.3f} mean|logw|={logw.abs().mean():.3f}", flush=True)


def tts_with_modules(U):
    """execute_text_to_speech (:413-450) of the reference with the two REAL modules (tiny sizes) instead of the stand-ins."""
    ecfg = EncoderConfig(n_vocab=20, n_feats=TINY.n_feats, n_channels=16, filter_channels=32, n_heads=2, n_layers=2, kernel_size=3, window_size=4)
    dcfg = DurationPredictorConfig(in_channels=16, filter_channels=24, kernel_size=3, spk_emb_dim=TINY.spk_emb_dim)
    enc, dp = modules(ecfg, dcfg, 1)
    ids, lens, spk = inputs(ecfg, dcfg, 1, 9, [9], 31)      # the reference's guidance branches assume one utterance (:298-331)
    m = build(U, TINY, 0)
    with torch.no_grad():
        _, x, x_mask = enc(ids, lens)
        logw = dp(x, x_mask, w=None, g=spk, reverse=True)
        frames = int(torch.clamp_min(torch.sum(torch.ceil(torch.exp(logw) * x_mask), [1, 2]), 1).max())
    n_down = len(TINY.dim_mults) - 1
    tp = -(-frames // (1 << n_down)) * (1 << n_down)
    n_steps = 4
    gz = np.random.Generator(np.random.Philox(key=4243))
    z = torch.from_numpy(gz.standard_normal((1, TINY.n_feats, tp), dtype=np.float32))
    noise = [torch.from_numpy(gz.standard_normal((1, TINY.n_feats, tp), dtype=np.float32)) for _ in range(n_steps)]
    orig = torch.randn_like                      # the reference draws z with randn_like (:441), then one randn per step
    torch.randn_like = lambda *a, **k: z.clone()
    try:
        with ReplayRandn(noise):
            enc_out, dec_out, attn = m.execute_text_to_speech(ids, lens, spk, enc, dp, n_down, diffusion_steps=n_steps, length_scale=1.0,
                                                              text_gradient_scale=1.0, spk_gradient_scale=1.0)
    finally:
        torch.randn_like = orig
    save("tts_modules_tiny", ids=ids, lengths=lens, spk_emb=spk, enc_out=enc_out, dec_out=dec_out, attn=attn, logw=logw,
         n_steps=n_steps, frames=frames, tp=tp, noise_key=4243)
    print(f"tts_modules_tiny: frames={frames} tp={tp} mean|dec|={dec_out.abs().mean():.3f}", flush=True)


if __name__ == "__main__":
    U = load_reference()
    os.makedirs(OUT, exist_ok=True)
    one("frontend_tiny", TINY_E, TINY_D, 3, 12, [12, 7, 3], 11)
    one("frontend_full", FULL_E, FULL_D, 2, 50, [50, 37], 12)
    one("frontend_nowin", NOWIN_E, NOWIN_D, 2, 10, [10, 4], 13)
    tts_with_modules(U)
