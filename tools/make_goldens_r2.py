#!/usr/bin/env python3
"""Round-2 golden vectors from the REFERENCE decoder (build container only; see tools/make_goldens.py for the loader).

    PYTHONDONTWRITEBYTECODE=1 python tools/make_goldens_r2.py [names...]

  loop_T1024      50-step loop at the BASELINE shape 80x1024 (lengths=[984]), reference fp32 and fp64 columns
  allgrads        loss_t + EVERY parameter gradient + input gradients (x0, cond, spk_emb), tiny config, both the shipped
                  weight recipe and the strong-attention one (g = 1, unscaled to_qkv); full size: input gradients
  attn_eval       one evaluation with the strong-attention recipe, tiny and full
  tts             the reference's own execute_text_to_speech (:413-450) with the seeded front-end stand-ins of
                  unitspeech_amd/frontend.py, tiny and full, + the de-normalised mel of inference.py:140
  ckpt            checkpoints written with torch.save from the reference module in the layouts of
                  train_STEP1.py:297-304 and finetune.py:169-173 (dim=8, dim_mults=(1,2,4) config)
"""
from __future__ import annotations

import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_goldens import FULL, OUT, TINY, ReplayRandn, build, load_reference, run_loop, save, tt  # noqa: E402
from unitspeech_amd.frontend import SyntheticFrontEnd, text_to_ids  # noqa: E402
from unitspeech_amd.params import DecoderConfig, synthetic_inputs, synthetic_state_dict  # noqa: E402

CK = DecoderConfig(dim=8, dim_mults=(1, 2, 4))
GRAD_SAMPLE = 8192        # gradients larger than this are stored as a strided sample + fp64 sum and sum of squares


def build_recipe(U, cfg, seed, dtype=torch.float32, **recipe):
    m = U.UnitSpeech(cfg.n_feats, cfg.dim, list(cfg.dim_mults), cfg.beta_min, cfg.beta_max, int(cfg.pe_scale), cfg.spk_emb_dim)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic_state_dict(cfg, seed, **recipe).items()}, strict=True)
    return m.to(dtype).eval()


def g_loop_T1024(U):
    T, n = 1024, 50
    inp = tt(synthetic_inputs(FULL, 1, T, seed=8, n_steps=n, lengths=[984]))
    t0 = time.time()
    out = run_loop(build(U, FULL, 0), inp, n, 1.0, 1.0)
    print(f"   fp32 loop {time.time() - t0:.0f} s  mean|out|={out.abs().mean():.3f} max={out.abs().max():.1f}", flush=True)
    t0 = time.time()
    out64 = run_loop(build(U, FULL, 0, torch.float64), {k: v.double() for k, v in inp.items()}, n, 1.0, 1.0)
    print(f"   fp64 loop {time.time() - t0:.0f} s  fp32-vs-fp64 L1={(out.double() - out64).abs().mean():.3e}", flush=True)
    save("loop_full_N50_T1024", out=out, out_fp64=out64, lengths=np.array([984]), seed=8,
         noise_abs_sum=inp["noise"].double().abs().sum(), z_abs_sum=inp["z"].double().abs().sum(), w_text=1.0, w_spk=1.0)


def loss_with_input_grads(m, cfg, T, t, zz, inp):
    x0 = inp["z"].clone().requires_grad_(True)
    cond = inp["cond"].clone().requires_grad_(True)
    spk = inp["spk_emb"].clone().requires_grad_(True)
    with ReplayRandn([zz]):
        loss, xt = m.loss_t(x0, inp["mask"], cond, t, spk)
    loss.backward()
    return loss, xt, x0.grad, cond.grad, spk.grad


def sample_stride(numel):
    """Odd stride giving <= GRAD_SAMPLE samples that walk every axis of a conv weight (tests use the same rule)."""
    return (numel // GRAD_SAMPLE + 1) | 1


def g_allgrads(U):
    T = 32
    for tag, recipe in (("", {}), ("_attn", dict(rezero_g=1.0, qkv_scale=1.0))):
        m = build_recipe(U, TINY, 0, **recipe).train()
        inp = tt(synthetic_inputs(TINY, 2, T, seed=6, lengths=[T, T - 8]))
        t = torch.tensor([0.3, 0.8], dtype=torch.float32)
        zz = torch.from_numpy(np.random.Generator(np.random.Philox(key=77)).standard_normal((2, TINY.n_feats, T), dtype=np.float32))
        loss, xt, gx0, gcond, gspk = loss_with_input_grads(m, TINY, T, t, zz, inp)
        full = {n: p.grad for n, p in m.named_parameters() if p.grad is not None}
        assert len(full) == 228                    # all but text_uncon / spk_uncon, which loss_t never reads
        gn = torch.sqrt(sum((g.double() ** 2).sum() for g in full.values()))
        grads = {}
        for n, g in full.items():
            flat = g.reshape(-1)
            grads["gradsum:" + n] = flat.double().sum()
            grads["gradsq:" + n] = (flat.double() ** 2).sum()
            grads["grad:" + n] = g if flat.numel() <= GRAD_SAMPLE else flat[::sample_stride(flat.numel())]
        save(f"loss_tiny_allgrads{tag}", t=t, z=zz, loss=loss.detach(), xt=xt.detach(), grad_norm=gn,
             grad_x0=gx0, grad_cond=gcond, grad_spk_emb=gspk, **grads)
        print(f"   tiny{tag}: loss={loss.item():.6f} grad_norm={gn.item():.6f} n_grads={len(full)}"
              f" |gx0|={gx0.abs().mean():.3e} |gcond|={gcond.abs().mean():.3e} |gspk|={gspk.abs().mean():.3e}")
    T = 64
    m = build(U, FULL, 0).train()
    inp = tt(synthetic_inputs(FULL, 2, T, seed=6, lengths=[T, T - 8]))
    t = torch.tensor([0.3, 0.8], dtype=torch.float32)
    zz = torch.from_numpy(np.random.Generator(np.random.Philox(key=77)).standard_normal((2, FULL.n_feats, T), dtype=np.float32))
    loss, xt, gx0, gcond, gspk = loss_with_input_grads(m, FULL, T, t, zz, inp)
    save("loss_full_inputgrads", t=t, z=zz, loss=loss.detach(), grad_x0=gx0, grad_cond=gcond, grad_spk_emb=gspk)
    print(f"   full: loss={loss.item():.6f} |gx0|={gx0.abs().mean():.3e} |gcond|={gcond.abs().mean():.3e} |gspk|={gspk.abs().mean():.3e}")


def g_attn_eval(U):
    for tag, cfg, T in (("tiny", TINY, 32), ("full", FULL, 64)):
        inp = tt(synthetic_inputs(cfg, 3, T, seed=2, lengths=[T, T - 5, T - 16]))
        t = torch.tensor([0.99, 0.5, 0.013], dtype=torch.float32)
        recipe = dict(rezero_g=1.0, qkv_scale=1.0)
        with torch.no_grad():
            out = build_recipe(U, cfg, 0, **recipe).estimator(inp["z"], inp["mask"], inp["cond"], t, inp["spk_emb"])
            out0 = build_recipe(U, cfg, 0, rezero_g=0.0, qkv_scale=1.0).estimator(inp["z"], inp["mask"], inp["cond"], t, inp["spk_emb"])
            out64 = build_recipe(U, cfg, 0, torch.float64, **recipe).estimator(
                inp["z"].double(), inp["mask"].double(), inp["cond"].double(), t.double(), inp["spk_emb"].double())
        save(f"estimator_{tag}_attn", t=t, out=out, out_fp64=out64, lengths=np.array([T, T - 5, T - 16]))
        print(f"   {tag}: mean|out|={out.abs().mean():.4f} attention share (mean|out - out(g=0)|)={(out - out0).abs().mean():.4f}"
              f"  fp32-vs-fp64 L1={(out.double() - out64).abs().mean():.3e}")


def g_tts(U):
    for tag, cfg, text, n in (("tiny", TINY, "hello there, general", 10), ("full", FULL, "mi355x", 10)):
        m = build(U, cfg, 0)
        fe = SyntheticFrontEnd(cfg.n_feats)
        phoneme, lengths = text_to_ids(text)
        spk = tt(synthetic_inputs(cfg, 1, 8, seed=9))["spk_emb"]
        # dry call to learn T (the reference draws z with randn_like, then N x randn)
        cond_x, x, x_mask = fe.text_encoder(phoneme, lengths)
        w_ceil = torch.ceil(torch.exp(fe.duration_predictor(x, x_mask)) * x_mask)
        ylen = int(torch.clamp_min(w_ceil.sum([1, 2]), 1).long().max())
        Tp = U.fix_len_compatibility(ylen, len(cfg.dim_mults) - 1)
        g = np.random.Generator(np.random.Philox(key=4242))
        z = torch.from_numpy(g.standard_normal((1, cfg.n_feats, Tp), dtype=np.float32))
        noise = torch.from_numpy(g.standard_normal((n, 1, cfg.n_feats, Tp), dtype=np.float32))
        orig = torch.randn_like
        torch.randn_like = lambda *a, **k: z.clone()
        try:
            with ReplayRandn(list(noise)):
                enc, dec, attn = m.execute_text_to_speech(phoneme, lengths, spk, fe.text_encoder, fe.duration_predictor,
                                                          len(cfg.dim_mults) - 1, diffusion_steps=n, length_scale=1.0,
                                                          text_gradient_scale=1.0, spk_gradient_scale=1.0)
        finally:
            torch.randn_like = orig
        mel_min, mel_max = torch.tensor(-11.5), torch.tensor(2.0)
        mel = (dec + 1) / 2 * (mel_max - mel_min) + mel_min                    # inference.py:140
        # z and the per-step noise are regenerated by the tests from Philox(key=4242) in this order; checksums only
        save(f"tts_{tag}", phoneme=phoneme, phoneme_lengths=lengths, spk_emb=spk, z_abs_sum=z.double().abs().sum(),
             noise_abs_sum=noise.double().abs().sum(), enc_out=enc, dec_out=dec, attn=attn, mel=mel, mel_min=mel_min,
             mel_max=mel_max, n_steps=n, y_length=ylen, text=np.array(text))
        print(f"   {tag}: text={text!r} L={phoneme.shape[1]} y_len={ylen} T'={Tp} mean|dec|={dec.abs().mean():.3f}")


def g_ckpt(U):
    m = build(U, CK, 3)
    spk = tt(synthetic_inputs(CK, 1, 8, seed=9))["spk_emb"]
    # train_STEP1.py:297-304 (decoder checkpoint written by the trainer)
    table = torch.nn.Embedding.from_pretrained(torch.arange(4 * CK.spk_emb_dim, dtype=torch.float32).reshape(4, -1) / 1024)   # :133
    torch.save({"model": m.state_dict(), "spk_emb": table.state_dict(), "mel_min": torch.tensor(-11.5),
                "mel_max": torch.tensor(2.0), "iteration": 1234}, os.path.join(OUT, "ckpt_pretrained_small.pt"))
    # finetune.py:169-173 (speaker-adapted checkpoint)
    d = torch.load(os.path.join(OUT, "ckpt_pretrained_small.pt"))        # finetune.py:62 loads the pre-trained dict and mutates it
    d["model"] = m.state_dict(); d["mel_min"] = torch.tensor(-11.5); d["mel_max"] = torch.tensor(2.0); d["spk_emb"] = spk
    torch.save(d, os.path.join(OUT, "ckpt_finetuned_small.pt"))
    for f in ("ckpt_pretrained_small.pt", "ckpt_finetuned_small.pt"):
        print(f"   wrote {f} ({os.path.getsize(os.path.join(OUT, f)) / 1024:.0f} KiB)")


ALL = {"loop_T1024": g_loop_T1024, "allgrads": g_allgrads, "attn_eval": g_attn_eval, "tts": g_tts, "ckpt": g_ckpt}

if __name__ == "__main__":
    torch.set_num_threads(int(os.environ.get("GOLDEN_THREADS", "8")))
    U = load_reference()
    for name in (sys.argv[1:] or list(ALL)):
        print(name, flush=True)
        ALL[name](U)
