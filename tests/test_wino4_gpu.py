"""Winograd F(4x4,3x3) / F(2x4,3x3) on the interpolation points {0, +-5/8, +-3/2} (csrc/wino4.hip, round 4): the stride-1 3x3 convolutions of
`Block` (unitspeech/unitspeech.py:46-55) at U-Net levels 1-3 in inference.  VERDICT r3 item 7(a) asked for the measurement on the GPU, judged
by the existing parity tolerances (2e-6 per evaluation, 1e-3 over 50 steps at T = 1024) -- the default configuration US_WINO4="0,44,44,24" is
what every other test file of this suite now runs; here: each form on its own against the oracle module by module (odd tile counts, padded
frames, the GroupNorm-fused input transform), every level combination against the reference goldens, batch independence, and the training
mode of the weight store (the inference-only pack is skipped while training and refreshed by the first inference call afterwards)."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from oracle import decoder_oracle as O
from unitspeech_amd import DecoderConfig, UnitSpeech, _lib, synthetic_inputs, synthetic_state_dict

pytestmark = pytest.mark.gpu

FULL = DecoderConfig()
DEV = "cuda:0"
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def G(d):
    return {k: torch.from_numpy(np.asarray(v)) for k, v in d.items()}


def l1(a, b):
    return float((a.double().cpu() - b.double().cpu()).abs().mean())


@pytest.fixture(scope="module")
def sd_np():
    return synthetic_state_dict(FULL, 0)


def build(sd_np, monkeypatch, forms):
    monkeypatch.setenv("US_WINO4", forms)                   # read when the handle is created
    m = UnitSpeech(FULL.n_feats, FULL.dim, list(FULL.dim_mults), FULL.beta_min, FULL.beta_max, FULL.pe_scale, FULL.spk_emb_dim)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd_np.items()}, strict=True)
    return m.to(DEV).eval()


def debug_block(model, kind, prefix, level, x_nchw, mask_full, temb, cout):
    eng = model._sync(torch.device(DEV))
    lib = eng.lib
    B, _, H, W = x_nchw.shape
    T = mask_full.shape[-1]
    x = x_nchw.permute(0, 2, 3, 1).contiguous().to(DEV)
    out = torch.empty(B, H, W, cout, device=DEV)
    ws = torch.empty(int(lib.us_workspace_bytes(eng.handle, B, T)), dtype=torch.uint8, device=DEV)
    m = mask_full.reshape(B, T).contiguous().to(DEV)
    te = temb.contiguous().to(DEV) if temb is not None else None
    rc = lib.us_debug_block(eng.handle, kind, prefix.encode(), level, C.c_void_p(x.data_ptr()), C.c_void_p(m.data_ptr()),
                            C.c_void_p(te.data_ptr()) if te is not None else None, C.c_void_p(out.data_ptr()), B, T, C.c_void_p(ws.data_ptr()),
                            ws.numel(), None)
    _lib.check(rc, eng.handle, "us_debug_block")
    torch.cuda.synchronize()
    return out.permute(0, 3, 1, 2).cpu()


# (form of the level, level, prefix of a ResnetBlock at that level, channels in / out, T)
CASES = [
    ("0,44,0,0", 1, "estimator.downs.1.1", 256, 256, 136),     # 40 x 68: 10 x 17 tiles, identity residual
    ("0,44,0,0", 1, "estimator.downs.1.0", 128, 256, 72),      # K = 128 -> 256 with the 1x1 res_conv, 40 x 36
    ("0,0,44,0", 2, "estimator.downs.2.1", 512, 512, 136),     # 20 x 34: the last tile column has 2 of its 4 outputs inside
    ("0,0,24,0", 2, "estimator.downs.2.1", 512, 512, 72),      # F(2x4) at level 2, 20 x 18
    ("0,0,0,24", 3, "estimator.mid_block1", 1024, 1024, 136),  # 10 x 17: 5 x 5 tiles, 3 outputs outside
    ("0,0,0,44", 3, "estimator.mid_block1", 1024, 1024, 72),   # F(4x4) at level 3: 10 rows = 2.5 tiles
]


@pytest.mark.parametrize("forms,level,prefix,cin,cout,T", CASES)
def test_resnet_block_with_the_4_wide_winograd_forms_vs_oracle(sd_np, monkeypatch, forms, level, prefix, cin, cout, T):
    """One whole `ResnetBlock` (:58-75) through us_debug_block: block1's 3x3 takes the plain input transform, block2's the one with
    block1's GroupNorm + Mish + time embedding evaluated on the fly; two items, one with padded frames; against the oracle module."""
    model = build(sd_np, monkeypatch, forms)
    sd = O.to_torch(sd_np)
    B, H, W = 2, FULL.n_feats >> level, T >> level
    g = np.random.Generator(np.random.Philox(key=1000 + level))
    x = torch.from_numpy(g.standard_normal((B, cin, H, W), dtype=np.float32))
    temb = torch.from_numpy(g.standard_normal((B, FULL.dim + FULL.spk_emb_dim), dtype=np.float32))
    mask_full = torch.ones(B, 1, T)
    mask_full[1, :, T - 24:] = 0
    mask = mask_full[:, :, ::(1 << level)].reshape(B, 1, 1, W)
    got = debug_block(model, 1, prefix, level, x * mask, mask_full, temb, cout)
    ref = O.resnet_block(sd, prefix, x, mask, temb)
    e = l1(got, ref) / float(ref.abs().mean())
    print(f"\n{forms} level {level} {cin}->{cout} {H}x{W}: relative L1 vs the oracle ResnetBlock {e:.2e}")
    assert torch.isfinite(got).all() and e <= 2e-6
    # and the block on its own (Block :46-55, the plain input transform; an all-ones mask shows the padded columns too)
    ones = torch.ones(B, 1, T)
    got_b = debug_block(model, 0, prefix, level, x, ones, None, cout)
    ref_b = O.block(sd, f"{prefix}.block1", x, torch.ones(B, 1, 1, W))
    assert l1(got_b, ref_b) / float(ref_b.abs().mean()) <= 2e-6


@pytest.mark.parametrize("forms", ["0,44,44,24", "0,44,44,44", "0,24,24,24", "0,0,0,0"])
def test_evaluation_and_50_step_loop_vs_reference_goldens(sd_np, monkeypatch, forms):
    """The parity bars of the path, per configuration of the forms: one full-size evaluation against the reference's fp32 and fp64 outputs
    (2e-6), the 50-step decode at 80x1024 against the reference's (1e-3, north star).  "0,0,0,0" keeps F(2x2,3x3) everywhere (round 3)."""
    model = build(sd_np, monkeypatch, forms)
    g = G(np.load(os.path.join(GOLD, "estimator_full.npz")))
    with torch.no_grad():
        out = model.estimator(g["x"].to(DEV), g["mask"].to(DEV), g["mu"].to(DEV), g["t"].to(DEV), g["spk_emb"].to(DEV))
    e32, e64 = l1(out, g["out"]), l1(out, g["out_fp64"])
    assert (out.cpu() * (1 - g["mask"])).abs().max().item() == 0.0
    gl = G(np.load(os.path.join(GOLD, "loop_full_N50_T1024.npz")))
    inp = G(synthetic_inputs(FULL, 1, 1024, seed=int(gl["seed"]), n_steps=50, lengths=[int(gl["lengths"][0])]))
    lo = model(inp["z"].to(DEV), inp["mask"].to(DEV), inp["cond"].to(DEV), inp["spk_emb"].to(DEV), 50, 1.0, 1.0, noise=inp["noise"].to(DEV))
    l32, l64 = l1(lo, gl["out"]), l1(lo, gl["out_fp64"])
    print(f"\nUS_WINO4={forms}: evaluation L1 vs reference fp32 {e32:.2e} / fp64 {e64:.2e}; 50 steps at T=1024 vs reference fp32 {l32:.2e} / fp64 {l64:.2e} "
          f"(the reference's own fp32 vs fp64: {l1(gl['out'], gl['out_fp64']):.2e})")
    assert e32 <= 2e-6 and e64 <= 2e-6 and l32 <= 1e-3 and l64 <= 1e-3


@pytest.mark.parametrize("T", [8, 40])
def test_short_utterances_vs_oracle(sd_np, monkeypatch, T):
    """T = 8: one column at level 3, two at level 2, four at level 1 -- a single, mostly empty tile column at every level of the 4-wide forms;
    T = 40: 5 / 10 / 20 columns (partial last tiles at levels 3 and 2).  Full-size weights, a padded item, against the oracle."""
    model = build(sd_np, monkeypatch, "0,44,44,24")
    sd = O.to_torch(sd_np)
    inp = G(synthetic_inputs(FULL, 2, T, seed=41, lengths=[T, max(T - 8, 4)]))
    t = torch.tensor([0.8, 0.15])
    with torch.no_grad():
        out = model.estimator(inp["z"].to(DEV), inp["mask"].to(DEV), inp["cond"].to(DEV), t.to(DEV), inp["spk_emb"].to(DEV))
    ref = O.estimator_forward(sd, inp["z"], inp["mask"], inp["cond"], t, inp["spk_emb"])
    e = l1(out, ref)
    print(f"\nT={T} evaluation: L1 vs oracle {e:.2e} (mean |ref| {ref.abs().mean():.3f})")
    assert torch.isfinite(out).all() and e <= 2e-6


def test_ragged_evaluation_and_batch_independence(sd_np, monkeypatch):
    """T = 136 (17 / 34 / 68 columns at levels 3 / 2 / 1: partial tile columns everywhere) with a padded and a fully padded item, against the
    oracle; and an item's result must not depend on what it is batched with, bit for bit (the tile of a Winograd-domain GEMM follows the
    launch geometry, the order in which an output element is summed does not)."""
    model = build(sd_np, monkeypatch, "0,44,44,24")
    sd = O.to_torch(sd_np)
    T = 136
    inp = G(synthetic_inputs(FULL, 3, T, seed=31, lengths=[T, T - 16, 8]))
    t = torch.tensor([0.9, 0.4, 0.07])
    with torch.no_grad():
        out = model.estimator(inp["z"].to(DEV), inp["mask"].to(DEV), inp["cond"].to(DEV), t.to(DEV), inp["spk_emb"].to(DEV))
        one = model.estimator(inp["z"][1:2].to(DEV), inp["mask"][1:2].to(DEV), inp["cond"][1:2].to(DEV), t[1:2].to(DEV), inp["spk_emb"][1:2].to(DEV))
    ref = O.estimator_forward(sd, inp["z"], inp["mask"], inp["cond"], t, inp["spk_emb"])
    e = l1(out, ref)
    print(f"\nragged T=136 evaluation: L1 vs oracle {e:.2e} (mean |ref| {ref.abs().mean():.3f})")
    assert e <= 2e-6 and torch.equal(out[1:2], one)


def test_training_mode_skips_the_inference_only_pack_and_inference_refreshes_it(sd_np, monkeypatch):
    """us_decoder_set_training: a training forward's weight sync leaves the F(4x4) packs stale (an optimiser step would otherwise re-pack
    36 + 16 + 16 matrices per convolution); the first inference call afterwards loads those weights again.  The weights CHANGE in between:
    an inference result computed from a stale pack would show."""
    model = build(sd_np, monkeypatch, "0,44,44,24")
    eng = model._sync(torch.device(DEV))
    lib = eng.lib
    assert lib.us_decoder_stale_inference_forms(eng.handle) == 0
    T = 64
    inp = G(synthetic_inputs(FULL, 1, T, seed=5))
    x, mask, cond, spk = (inp[k].to(DEV) for k in ("z", "mask", "cond", "spk_emb"))
    t = torch.tensor([0.5], device=DEV)
    with torch.no_grad():
        before = model.estimator(x, mask, cond, t, spk)
    model.train()
    torch.manual_seed(0)
    loss, _ = model.compute_loss(x.clamp(-1, 1), mask, cond, spk)
    loss.backward()
    with torch.no_grad():
        for n, p in model.named_parameters():
            if "mid_block1.block1.block.0.weight" in n:       # (a GroupNorm follows: scaling the weight would change nothing)
                p.add_(torch.randn_like(p) * p.std())         # bumps the version: re-loaded by the next sync
    loss2, _ = model.compute_loss(x.clamp(-1, 1), mask, cond, spk)      # a training sync: the changed tensor's inference pack goes stale
    torch.cuda.synchronize()
    assert lib.us_decoder_stale_inference_forms(eng.handle) >= 1
    model.eval()
    with torch.no_grad():
        after = model.estimator(x, mask, cond, t, spk)
    assert lib.us_decoder_stale_inference_forms(eng.handle) == 0
    sd2 = O.to_torch({k: v.detach().cpu().numpy() for k, v in model.state_dict().items()})
    ref = O.estimator_forward(sd2, inp["z"], inp["mask"], inp["cond"], t.cpu(), inp["spk_emb"])
    assert l1(after, ref) <= 2e-6 and l1(after, before) > 1e-3
