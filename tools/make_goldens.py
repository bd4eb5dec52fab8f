#!/usr/bin/env python3
"""Generate golden vectors from the REFERENCE decoder (build container only).

Imports `/root/reference/unitspeech/unitspeech.py` on CPU (absent third-party modules that the
decoder math never touches are stubbed with MagicMock), loads the procedurally
generated weights of `unitspeech_amd.params.synthetic_state_dict` into the reference modules and
records inputs + outputs as small `.npz` files under `tests/golden/`.

The reference never travels to the GPU box; only these data files do.  Re-run with
    PYTHONDONTWRITEBYTECODE=1 python tools/make_goldens.py
"""
from __future__ import annotations

import importlib.machinery
import os
import random
import sys
from unittest.mock import MagicMock

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from unitspeech_amd.params import DecoderConfig, synthetic_inputs, synthetic_state_dict  # noqa: E402

REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")

TINY = DecoderConfig(dim=16)
FULL = DecoderConfig()


def load_reference():
    sys.path.insert(0, REF)
    sys.dont_write_bytecode = True
    for _ in range(80):
        try:
            import unitspeech.unitspeech as U
            return U
        except ModuleNotFoundError as e:
            if e.name.startswith(("unitspeech", "conf")):
                raise
            m = MagicMock()
            m.__spec__ = importlib.machinery.ModuleSpec(e.name, None)
            m.__path__ = []
            sys.modules[e.name] = m
            if e.name == "s3prl.upstream.interfaces":
                m.UpstreamBase = type("UpstreamBase", (object,), {})
            for k in [k for k in sys.modules if k.startswith(("unitspeech", "conf"))]:
                del sys.modules[k]
    raise RuntimeError("could not import the reference decoder")


def build(U, cfg: DecoderConfig, seed: int, dtype=torch.float32):
    m = U.UnitSpeech(cfg.n_feats, cfg.dim, list(cfg.dim_mults), cfg.beta_min, cfg.beta_max,
                     int(cfg.pe_scale), cfg.spk_emb_dim)
    sd = {k: torch.from_numpy(v) for k, v in synthetic_state_dict(cfg, seed).items()}
    assert list(sd.keys()) == list(m.state_dict().keys()), "state_dict key order mismatch"
    m.load_state_dict(sd, strict=True)
    return m.to(dtype).eval()


class ReplayRandn:
    """Replace torch.randn with an iterator over pre-drawn tensors."""

    def __init__(self, draws):
        self.draws = list(draws)
        self.i = 0

    def __enter__(self):
        self.orig = torch.randn
        torch.randn = self
        return self

    def __exit__(self, *a):
        torch.randn = self.orig

    def __call__(self, *shape, **kw):
        d = self.draws[self.i]
        self.i += 1
        if len(shape) == 1 and not isinstance(shape[0], int):
            shape = tuple(shape[0])
        assert tuple(d.shape) == tuple(shape), (d.shape, shape)
        return d.to(kw.get("dtype", d.dtype))


def tt(d, dtype=torch.float32):
    return {k: torch.from_numpy(v).to(dtype) for k, v in d.items()}


def save(name, **arrs):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **{k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v))
                                 for k, v in arrs.items()})
    print(f"  wrote {path}  ({os.path.getsize(path) / 1024:.1f} KiB)")


def weight_fingerprint(cfg, seed):
    sd = synthetic_state_dict(cfg, seed)
    tot = np.float64(0)
    for v in sd.values():
        tot += np.abs(v.astype(np.float64)).sum()
    first = sd["estimator.final_block.block.0.weight"].ravel()[:8]
    return np.array([tot]), first


def run_loop(model, inp, n, wt, ws):
    with ReplayRandn(list(inp["noise"])):
        return model.forward(inp["z"], inp["mask"], inp["cond"], inp["spk_emb"], n, wt, ws)


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    U = load_reference()

    # ---- G0 weight-generator fingerprint (detects drift of the NumPy stream) -------------------
    for tag, cfg in (("tiny", TINY), ("full", FULL)):
        tot, first = weight_fingerprint(cfg, 0)
        save(f"weights_fingerprint_{tag}", abs_sum=tot, first8=first)

    # ---- G1 schedule tables ---------------------------------------------------------------------
    print("G1 schedule")
    model = build(U, TINY, 0)
    for n in (2, 10, 50):
        inp = tt(synthetic_inputs(TINY, 1, 8, seed=1, n_steps=n))
        # run the reference loop with scales 0 (one cheap estimator call per step) just to build tables
        run_loop(model, inp, n, 0.0, 0.0)
        names = ["betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_one_minus_alphas_cumprod",
                 "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod", "posterior_variance"]
        save(f"schedule_N{n}", **{k: getattr(model, k) for k in names})

    # ---- G2 time embedding ----------------------------------------------------------------------
    print("G2 time embedding")
    for tag, cfg in (("tiny", TINY), ("full", FULL)):
        m = build(U, cfg, 0)
        t = torch.tensor([1e-5, 0.01, 0.5, 0.99, 0.995], dtype=torch.float32)
        e = m.estimator.time_pos_emb(t, scale=m.estimator.pe_scale)
        save(f"temb_{tag}", t=t, posemb=e, mlp=m.estimator.mlp(e))
        if tag == "full":
            full_model = m

    # ---- G3 blocks in isolation (tiny width) ----------------------------------------------------
    print("G3 blocks")
    est = model.estimator
    g = np.random.Generator(np.random.Philox(key=1234))
    B, H, W = 2, 20, 12
    mask = torch.ones(B, 1, 1, W); mask[1, :, :, 9:] = 0
    temb = torch.from_numpy(g.standard_normal((B, TINY.temb_dim), dtype=np.float32))
    x32 = torch.from_numpy(g.standard_normal((B, 32, H, W), dtype=np.float32))
    x16 = torch.from_numpy(g.standard_normal((B, 16, H, W), dtype=np.float32))
    with torch.no_grad():
        save("blocks_tiny", mask=mask, temb=temb, x32=x32, x16=x16,
             block=est.downs[1][1].block1(x32, mask),                       # Block 32->32
             resnet_same=est.downs[1][1](x32, mask, temb),                  # ResnetBlock 32->32 (identity res)
             resnet_proj=est.downs[1][0](x16, mask, temb),                  # ResnetBlock 16->32 (1x1 res_conv)
             attn=est.downs[1][2](x32),                                     # Residual(Rezero(LinearAttention(32)))
             down=est.downs[1][3](x32 * mask),                              # Downsample(32)
             up=est.ups[1][3](x32 * mask))                                  # Upsample(32)

    # ---- G4 one estimator evaluation, B'=3 ------------------------------------------------------
    print("G4 estimator eval")
    for tag, cfg, T, m in (("tiny", TINY, 32, model), ("full", FULL, 64, full_model)):
        inp = tt(synthetic_inputs(cfg, 3, T, seed=2, lengths=[T, T - 5, T - 16]))
        t = torch.tensor([0.99, 0.5, 0.013], dtype=torch.float32)
        with torch.no_grad():
            out = m.estimator(inp["z"], inp["mask"], inp["cond"], t, inp["spk_emb"])
            out64 = build(U, cfg, 0, torch.float64).estimator(
                inp["z"].double(), inp["mask"].double(), inp["cond"].double(), t.double(), inp["spk_emb"].double())
        save(f"estimator_{tag}", x=inp["z"], mask=inp["mask"], mu=inp["cond"], t=t, spk_emb=inp["spk_emb"],
             out=out, out_fp64=out64)
        print(f"   {tag}: mean|out|={out.abs().mean():.4f}  fp32-vs-fp64 L1={(out.double() - out64).abs().mean():.3e}")

    # ---- G5/G6 sampler loops with explicit noise ------------------------------------------------
    print("G6 loops")
    T = 32
    for (wt, ws) in ((1.0, 1.0), (1.0, 0.0), (0.0, 1.0), (0.0, 0.0)):
        inp = tt(synthetic_inputs(TINY, 1, T, seed=3, n_steps=10, lengths=[T - 3]))
        out = run_loop(model, inp, 10, wt, ws)
        save(f"loop_tiny_N10_w{int(wt)}{int(ws)}", **inp, out=out, w_text=wt, w_spk=ws)
    # batched semantics = independent B=1 runs: 2 items run one at a time
    inp = tt(synthetic_inputs(TINY, 2, T, seed=4, n_steps=10, lengths=[T, T - 8]))
    outs = []
    for b in range(2):
        one = {k: (v[:, b:b + 1] if k == "noise" else v[b:b + 1]) for k, v in inp.items()}
        outs.append(run_loop(model, one, 10, 1.0, 1.0))
    save("loop_tiny_N10_B2", **inp, out=torch.cat(outs, 0), w_text=1.0, w_spk=1.0)

    T = 64
    full64 = build(U, FULL, 0, torch.float64)
    for n in (10, 50):
        inp = tt(synthetic_inputs(FULL, 1, T, seed=5, n_steps=n, lengths=[T - 4]))
        out = run_loop(full_model, inp, n, 1.0, 1.0)
        out64 = run_loop(full64, {k: v.double() for k, v in inp.items()}, n, 1.0, 1.0)
        print(f"   full N={n}: mean|out|={out.abs().mean():.3f} max={out.abs().max():.1f} finite={bool(torch.isfinite(out).all())}"
              f"  fp32-vs-fp64 L1={(out.double() - out64).abs().mean():.3e}")
        # noise is regenerated from the seed by the tests (1 MB at N=50); keep a checksum instead
        keep = {k: v for k, v in inp.items() if k != "noise"}
        save(f"loop_full_N{n}", **keep, out=out, out_fp64=out64, noise_abs_sum=inp["noise"].double().abs().sum(),
             w_text=1.0, w_spk=1.0)

    # ---- G7 loss_t + gradients ------------------------------------------------------------------
    print("G7 loss/grad")
    for tag, cfg, T, seed in (("tiny", TINY, 32, 0), ("full", FULL, 64, 0)):
        m = build(U, cfg, seed).train()
        inp = tt(synthetic_inputs(cfg, 2, T, seed=6, lengths=[T, T - 8]))
        t = torch.tensor([0.3, 0.8], dtype=torch.float32)
        zz = torch.from_numpy(np.random.Generator(np.random.Philox(key=77)).standard_normal(
            (2, cfg.n_feats, T), dtype=np.float32))
        with ReplayRandn([zz]):
            loss, xt = m.loss_t(inp["z"], inp["mask"], inp["cond"], t, inp["spk_emb"])
        loss.backward()
        names = ["estimator.final_conv.weight", "estimator.final_block.block.0.bias",
                 "estimator.final_block.block.1.weight", "estimator.downs.0.0.block1.block.0.weight",
                 "estimator.downs.0.0.res_conv.weight", "estimator.downs.1.2.fn.g",
                 "estimator.downs.1.2.fn.fn.to_out.bias", "estimator.mid_block1.mlp.1.bias",
                 "estimator.ups.0.3.conv.bias", "estimator.mlp.0.bias"]
        params = dict(m.named_parameters())
        gn = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in params.values() if p.grad is not None))
        grads = {("grad:" + n): params[n].grad for n in names}
        save(f"loss_{tag}", x0=inp["z"], mask=inp["mask"], cond=inp["cond"], spk_emb=inp["spk_emb"], t=t, z=zz,
             loss=loss.detach(), xt=xt.detach(), grad_norm=gn, **grads)
        print(f"   {tag}: loss={loss.item():.6f} grad_norm={gn.item():.6f}")

    # ---- G8 fine_tune segmenting ----------------------------------------------------------------
    print("G8 fine_tune")
    m = build(U, TINY, 0).train()
    L, Lu, seg = 48, 20, 32
    gg = np.random.Generator(np.random.Philox(key=88))
    y = torch.from_numpy(gg.standard_normal((1, 80, L), dtype=np.float32))
    cond_x = torch.from_numpy(gg.standard_normal((1, 80, Lu), dtype=np.float32))
    dur = torch.tensor([[3, 2, 3, 2, 3, 2, 3, 2, 3, 2, 3, 2, 3, 2, 3, 2, 2, 2, 2, 2]], dtype=torch.float32)
    assert int(dur.sum()) == L
    y_lengths = torch.LongTensor([L])
    y_mask = U.sequence_mask(y_lengths, L).unsqueeze(1).float()
    x_mask = torch.ones(1, 1, Lu)
    attn = U.generate_path(dur, (x_mask.unsqueeze(-1) * y_mask.unsqueeze(2)).squeeze(1))
    spk = tt(synthetic_inputs(TINY, 1, 8, seed=7))["spk_emb"]
    random.seed(123); torch.manual_seed(123)
    # capture the torch draws made inside (t then z) so the test can replay them
    t_draw = torch.rand(1); z_draw = torch.randn(1, 80, seg)
    random.seed(123)
    orig_rand = torch.rand
    torch.rand = lambda *a, **k: t_draw.clone()
    try:
        with ReplayRandn([z_draw]):
            loss = m.fine_tune(cond_x, y, y_mask, y_lengths, L, attn, spk, seg, 80)
    finally:
        torch.rand = orig_rand
    save("finetune_tiny", cond_x=cond_x, y=y, y_mask=y_mask, y_lengths=y_lengths, attn=attn, spk_emb=spk,
         t_draw=t_draw, z_draw=z_draw, py_seed=123, segment_size=seg, loss=loss.detach(), dur=dur)
    print(f"   fine_tune loss={loss.item():.6f}")


if __name__ == "__main__":
    main()
