"""GPU parity tests added in round 3 (all through the C ABI).

* The f16x3 operand range (VERDICT r2 "What's weak" 1): a tensor beyond the fp16 range that meets a split-precision GEMM
  is never clamped -- the handle reports it (`us_range_status`), the affected results are non-finite, a NaN stays a NaN, and the
  Python mirror repeats an inference call on the exact-fp32 engine, which matches the oracle.  Reference semantics: plain fp32
  everywhere, `unitspeech/unitspeech.py:46-96`.
* The loss scaling of the backward now lives inside `us_estimator_backward` (data-driven power of two for ANY magnitude of grad_out).
* loss and every parameter gradient of a pre-training batch (8 crops of 176 frames, full size) against the oracle's
  `loss_t` under torch autograd on the CPU (`train_STEP1.py:381`, `unitspeech/unitspeech.py:393-411`).
"""
import ctypes as C
import os
import warnings

import numpy as np
import pytest
import torch

from oracle import decoder_oracle as O
from unitspeech_amd import DecoderConfig, UnitSpeech, _lib, synthetic_inputs, synthetic_state_dict
from unitspeech_amd.unitspeech import RangeError

pytestmark = pytest.mark.gpu

FULL = DecoderConfig()
DEV = "cuda:0"
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def G(d):
    return {k: (torch.from_numpy(np.asarray(v)) if np.asarray(v).dtype.kind != "U" else str(v)) for k, v in d.items()}


def l1(a, b):
    return (a.double().cpu() - b.double().cpu()).abs().mean().item()


def build(sd_np, exact=False, train=False):
    cfg = FULL
    m = UnitSpeech(cfg.n_feats, cfg.dim, list(cfg.dim_mults), cfg.beta_min, cfg.beta_max, cfg.pe_scale, cfg.spk_emb_dim)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd_np.items()}, strict=True)
    m.exact = exact
    m = m.to(DEV)
    return m.train() if train else m.eval()


@pytest.fixture(scope="module")
def sd_np():
    return synthetic_state_dict(FULL, 0)


def _debug_block(eng, kind, prefix, level, x_nchw, T, cout):
    """One `Block` of the score network through us_debug_block on the given engine; returns (out [B,C,H,W], range status)."""
    lib = eng.lib
    B, _, H, W = x_nchw.shape
    x = x_nchw.permute(0, 2, 3, 1).contiguous().to(DEV)
    out = torch.empty(B, H, W, cout, device=DEV)
    ws = torch.empty(int(lib.us_workspace_bytes(eng.handle, B, T)), dtype=torch.uint8, device=DEV)
    m = torch.ones(B, T, device=DEV)
    rc = lib.us_debug_block(eng.handle, kind, prefix.encode(), level, C.c_void_p(x.data_ptr()), C.c_void_p(m.data_ptr()), None,
                            C.c_void_p(out.data_ptr()), B, T, C.c_void_p(ws.data_ptr()), ws.numel(), None)
    _lib.check(rc, eng.handle, "us_debug_block")
    st = C.c_uint(0)
    _lib.check(lib.us_range_status(eng.handle, C.byref(st), 1, None), eng.handle, "us_range_status")
    return out.permute(0, 3, 1, 2).cpu(), int(st.value)


# ---------------------------------------------------------------------------------------------------------------
# range: the C ABI reports, never clamps
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("level,prefix,C_,fine,beyond", [(0, "estimator.downs.0.1", 128, (1.0,), 1e5), (1, "estimator.downs.1.1", 256, (1.0, 2e4), 3e7)])
def test_block_with_activations_beyond_the_fp16_range_is_reported_and_exact_on_the_fp32_handle(sd_np, level, prefix, C_, fine, beyond):
    """`Block` (conv3x3 -> GroupNorm -> Mish, :46-55) on an input beyond the f16x3 operand range: level 0 runs the direct convolution (split
    inside the kernel; |x| ~ 1e5), level 1 the Winograd F(4x4) form, whose input transform scales by 2^-5 (wino4.hip: V overflows near
    |x| ~ 1e5 x 20 / 32, so 2e4 * N(0,1) -- values up to 9e4, beyond fp16 themselves -- stays on the fast path at full accuracy, and 3e7 does
    not).  Default handle beyond the range: status US_RANGE_ACT and a non-finite output (round 2 returned a finite, silently clamped one);
    exact-fp32 handle: status 0 and the oracle's values.  An in-range input leaves the status at 0 on both."""
    model = build(sd_np)
    T = 64
    H, W = FULL.n_feats >> level, T >> level
    g = np.random.Generator(np.random.Philox(key=5 + level))
    x = torch.from_numpy(g.standard_normal((2, C_, H, W), dtype=np.float32))
    sd = O.to_torch(sd_np)
    ones = torch.ones(2, 1, 1, W)
    for scale in fine + (beyond,):
        xs = x * scale
        ref = O.block(sd, prefix + ".block1", xs, ones)
        eng = model._sync(torch.device(DEV))
        got, st = _debug_block(eng, 0, prefix, level, xs, T, C_)
        eng_x = model._sync(torch.device(DEV), exact=True)
        got_x, st_x = _debug_block(eng_x, 0, prefix, level, xs, T, C_)
        e_x = l1(got_x, ref)
        print(f"\nlevel {level}, |x| ~ {scale:g}: f16x3 status {st}, exact status {st_x}, exact L1 vs oracle {e_x:.2e}"
              + (f", f16x3 L1 {l1(got, ref):.2e}" if st == 0 else f", f16x3 finite share {torch.isfinite(got).float().mean().item():.3f}"))
        assert st_x == 0 and e_x <= 2e-6 * max(1.0, ref.abs().mean().item())
        if scale in fine:
            assert st == 0 and l1(got, ref) <= 2e-6 * max(1.0, ref.abs().mean().item())
        else:
            assert st & _lib.US_RANGE_ACT
            assert not torch.isfinite(got).all()             # loud: an overflow is an infinity, not 65504


def test_nan_and_tiny_operands(sd_np):
    """A NaN activation stays a NaN (the clamp used to launder it into 65504) and does not count as a range event; operands around
    1e-6 (the floor of the lo plane) keep fp32-level accuracy through the GroupNorm that follows."""
    model = build(sd_np)
    T, level, prefix, C_ = 64, 0, "estimator.downs.0.1", 128
    H, W = FULL.n_feats, T
    g = np.random.Generator(np.random.Philox(key=11))
    x = torch.from_numpy(g.standard_normal((1, C_, H, W), dtype=np.float32))
    eng = model._sync(torch.device(DEV))
    xn = x.clone()
    xn[0, 3, 10, 20] = float("nan")
    got, st = _debug_block(eng, 0, prefix, level, xn, T, C_)
    assert st == 0 and torch.isnan(got).any()
    sd = O.to_torch(sd_np)
    ones = torch.ones(1, 1, 1, W)
    xt = x * 1e-6
    ref = O.block(sd, prefix + ".block1", xt, ones)
    got, st = _debug_block(eng, 0, prefix, level, xt, T, C_)
    eng_x = model._sync(torch.device(DEV), exact=True)
    got_x, _ = _debug_block(eng_x, 0, prefix, level, xt, T, C_)
    e, e_x = l1(got, ref), l1(got_x, ref)
    print(f"\n|x| ~ 1e-6: f16x3 L1 vs oracle {e:.2e}, exact-fp32 handle {e_x:.2e}, mean |out| {ref.abs().mean().item():.3f}")
    assert st == 0 and e <= 2e-6 and e_x <= 2e-6


# ---------------------------------------------------------------------------------------------------------------
# range: the Python mirror repeats the call on the exact-fp32 engine
# ---------------------------------------------------------------------------------------------------------------
def test_estimator_with_a_1e5_activation_falls_back_to_the_exact_engine_and_matches_the_oracle(sd_np):
    """Weights whose last Upsample (`ups.2.3`, :21) emits values around 1e5-1e6: they feed the final Block's direct f16x3 convolution.  The
    default engine reports the range event, the mirror warns and repeats the evaluation on the exact-fp32 engine; the result is
    the oracle's to 2e-6 of the output scale.  (GroupNorm makes the final Block scale-invariant, so the reference's own fp32
    arithmetic is well conditioned here.)"""
    sd = dict(sd_np)
    sd["estimator.ups.2.3.conv.weight"] = (sd_np["estimator.ups.2.3.conv.weight"] * np.float32(1e6)).astype(np.float32)
    sd["estimator.ups.2.3.conv.bias"] = (sd_np["estimator.ups.2.3.conv.bias"] * np.float32(1e6)).astype(np.float32)
    model = build(sd)
    T, B = 64, 2
    inp = G(synthetic_inputs(FULL, B, T, seed=3, lengths=[T, T - 8]))
    t = torch.tensor([0.3, 0.8])
    taps = {}
    ref = O.estimator_forward(O.to_torch(sd), inp["z"], inp["mask"], inp["cond"], t, inp["spk_emb"], FULL.pe_scale, taps=taps)
    big = (taps["ups.2"] * inp["mask"].unsqueeze(1)).abs()             # what the final Block's convolution reads (`x * mask`, :54)
    assert (big > 65520).float().mean().item() > 0.1 and torch.isfinite(ref).all()
    args = [inp[k].to(DEV) for k in ("z", "mask", "cond")] + [t.to(DEV), inp["spk_emb"].to(DEV)]
    with pytest.warns(RuntimeWarning, match="exact-fp32"), torch.no_grad():
        out = model.estimator(*args)
    e = l1(out, ref)
    scale = ref.abs().mean().item()
    print(f"\n1e5 activation: exact-engine fall-back L1 vs oracle {e:.2e} (mean |out| {scale:.3f}, max |ups.2| {taps['ups.2'].abs().max().item():.3g})")
    assert torch.isfinite(out).all() and e <= 2e-6 * max(1.0, scale)
    # with the check switched off the caller gets what the default engine computed: non-finite where the overflow mattered
    model.range_check = False
    with torch.no_grad():
        raw = model.estimator(*args)
    assert not torch.isfinite(raw).all()
    assert model.range_status() & _lib.US_RANGE_ACT
    # the sampler takes the same route
    model.range_check = True
    with pytest.warns(RuntimeWarning, match="exact-fp32"):
        dec = model(inp["z"].to(DEV), inp["mask"].to(DEV), inp["cond"].to(DEV), inp["spk_emb"].to(DEV), 2, 1.0, 1.0, rng="philox", seed=1)
    model.exact = True
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        dec_x = model(inp["z"].to(DEV), inp["mask"].to(DEV), inp["cond"].to(DEV), inp["spk_emb"].to(DEV), 2, 1.0, 1.0, rng="philox", seed=1)
    assert torch.isfinite(dec).all() and torch.equal(dec, dec_x)


def test_weight_beyond_the_fp16_range_moves_the_model_to_the_exact_engine(sd_np):
    """One convolution weight of 3e5 at a corner tap (so that U = G g G^T of the Winograd pack carries it undamped): the pack reports
    US_RANGE_WEIGHT, the first call is repeated on the exact engine and later calls go there directly."""
    sd = dict(sd_np)
    w = sd_np["estimator.downs.1.1.block1.block.0.weight"].copy()
    w[3, 5, 0, 0] = 3e5
    sd["estimator.downs.1.1.block1.block.0.weight"] = w
    model = build(sd)
    T, B = 64, 1
    inp = G(synthetic_inputs(FULL, B, T, seed=4))
    t = torch.tensor([0.5])
    ref = O.estimator_forward(O.to_torch(sd), inp["z"], inp["mask"], inp["cond"], t, inp["spk_emb"], FULL.pe_scale)
    args = [inp[k].to(DEV) for k in ("z", "mask", "cond")] + [t.to(DEV), inp["spk_emb"].to(DEV)]
    with pytest.warns(RuntimeWarning, match="exact-fp32"), torch.no_grad():
        out = model.estimator(*args)
    assert l1(out, ref) <= 2e-6 * max(1.0, ref.abs().mean().item())
    assert model._get_engine(False).weights_out_of_range
    with warnings.catch_warnings(), torch.no_grad():
        warnings.simplefilter("error")                       # no second detour through the default engine
        out2 = model.estimator(*args)
    assert torch.equal(out, out2)


def test_training_reports_a_range_event_one_call_late(sd_np):
    """Training calls cannot be repeated transparently (the tape is consumed): the status is copied asynchronously after every backward
    and the next forward raises RangeError."""
    sd = dict(sd_np)
    sd["estimator.ups.2.3.conv.weight"] = (sd_np["estimator.ups.2.3.conv.weight"] * np.float32(1e6)).astype(np.float32)
    model = build(sd, train=True)
    T = 64
    inp = G(synthetic_inputs(FULL, 1, T, seed=8))
    x0, mask, cond, spk = (inp[k].to(DEV) for k in ("z", "mask", "cond", "spk_emb"))
    torch.manual_seed(0)
    loss, _ = model.compute_loss(x0, mask, cond, spk)
    loss.backward()
    torch.cuda.synchronize()
    with pytest.raises(RangeError):
        model.compute_loss(x0, mask, cond, spk)
    model.exact = True                                        # the documented way out
    model.zero_grad(set_to_none=True)                         # (the first pass left non-finite gradients behind)
    torch.manual_seed(0)
    loss, _ = model.compute_loss(x0, mask, cond, spk)
    loss.backward()
    bad = [n for n, p in model.named_parameters() if p.grad is not None and not torch.isfinite(p.grad).all()]
    assert torch.isfinite(loss) and not bad, (float(loss), bad[:8])


@pytest.mark.parametrize("backward", ["eager", "graph"])
def test_finetune_graph_reports_a_range_event_one_step_late(sd_np, backward):
    """under FineTuneGraph the training forward only ever runs inside the capture, where `_EstimatorFn.forward` cannot poll the
    range word, so an activation beyond the fp16 range never raised and Adam kept stepping on non-finite gradients.  `step()` polls on
    the host before the replay now (and posts the word itself behind a fully captured iteration): the overflow of step i raises at
    step i + 1.  Also: an inference call on the same engine afterwards must not be repeated on the exact engine because of that stale
    training event (`_Engine.range_clear`)."""
    from unitspeech_amd.graph import FineTuneGraph
    from unitspeech_amd.util import generate_path, sequence_mask
    sd = dict(sd_np)
    sd["estimator.ups.2.3.conv.weight"] = (sd_np["estimator.ups.2.3.conv.weight"] * np.float32(1e6)).astype(np.float32)
    model = build(sd, train=True)
    gen = np.random.Generator(np.random.Philox(key=322))
    L, Lu, seg = 96, 32, 64
    y = torch.from_numpy(gen.standard_normal((1, 80, L), dtype=np.float32)).clamp(-1, 1).to(DEV)
    cond_x = torch.from_numpy(gen.standard_normal((1, 80, Lu), dtype=np.float32) * 0.5).to(DEV)
    y_len = torch.LongTensor([L]).to(DEV)
    y_mask = sequence_mask(y_len, L).unsqueeze(1).float()
    attn = generate_path(torch.full((1, Lu), 3.0, device=DEV), (torch.ones(1, 1, Lu, device=DEV).unsqueeze(-1) * y_mask.unsqueeze(2)).squeeze(1))
    spk = G(synthetic_inputs(FULL, 1, 8, seed=12))["spk_emb"].to(DEV)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        try:
            graph = FineTuneGraph(model, spk, 1, seg, 80, backward=backward)   # (its eager warm-up iterations may already raise: equally loud)
            graph.step(cond_x, y, y_len, attn)
            torch.cuda.synchronize()
            with pytest.raises(RangeError):
                graph.step(cond_x, y, y_len, attn)
        except RangeError:
            pass
    # a model whose weights are in range; an overflow in the LAST training iteration (nobody polls after it) must not make the next
    # inference call on the same engine fall back to the exact engine
    ok = build(sd_np, train=True)
    g2 = FineTuneGraph(ok, spk, 1, seg, 80, backward=backward)
    g2.step(cond_x, y, y_len, attn)
    g2.spk.mul_(1e8)              # the graph's static speaker embedding: the time projections push h1 beyond the fp16 range
    g2.step(cond_x, y, y_len, attn)
    torch.cuda.synchronize()
    eng = ok._get_engine()
    assert eng.range_status(reset=False) & _lib.US_RANGE_ACT          # the stale event stands
    inp = G(synthetic_inputs(FULL, 1, 32, seed=5, n_steps=2))
    ok.eval()
    with warnings.catch_warnings():
        warnings.simplefilter("error")            # a repeat on the exact engine warns: that would fail the test
        out = ok(inp["z"].to(DEV), inp["mask"].to(DEV), inp["cond"].to(DEV), inp["spk_emb"].to(DEV), 2, 1.0, 1.0, noise=inp["noise"].to(DEV))
    assert torch.isfinite(out).all()


# ---------------------------------------------------------------------------------------------------------------
# loss scaling inside us_estimator_backward
# ---------------------------------------------------------------------------------------------------------------
def _crops(B, T, key):
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


class _ReplayRandn:
    def __init__(self, draws):
        self.draws, self.i = list(draws), 0

    def __enter__(self):
        self.orig = torch.randn
        torch.randn = self
        return self

    def __exit__(self, *a):
        torch.randn = self.orig

    def __call__(self, *shape, **kw):
        d = self.draws[self.i]
        self.i += 1
        return d.to(device=kw.get("device", d.device), dtype=kw.get("dtype", d.dtype))


def _hip_grads(sd_np, x0, mask, cond, spk, t, z, exact=False, loss_factor=1.0):
    m = build(sd_np, exact=exact, train=True)
    with _ReplayRandn([z.to(DEV)]):
        loss, _ = m.loss_t(x0.to(DEV), mask.to(DEV), cond.to(DEV), t.to(DEV), spk.to(DEV))
    (loss * loss_factor).backward()
    torch.cuda.synchronize()
    assert m.range_status() == 0
    return float(loss), {n: p.grad.detach().double().cpu() for n, p in m.named_parameters() if p.grad is not None}


def _rel(new, ref, show=3, scalar_floor=0.0):
    """(whole-gradient relative L2, median per-tensor relative L2, worst per-tensor relative L2), every tensor -- the eight one-element
    Rezero gains included -- against its OWN norm.  (Round 3 measured the gains against the largest of them: they were sums formed from
    M1 = G^T q with float atomics and moved by 1e-3 between runs.  They are fixed-order fp64 sums of grad_out * fn(x) now, train.hip.)
    scalar_floor > 0 (the comparison of two fp32 pipelines on ONE crop only): a gain's gradient is a sum of ~1e6 products of like magnitude, so
    the rounding noise it inherits from its operands is ABSOLUTE, about the same for all eight, however far the sum itself cancels -- the
    crop of key 21 has one gain at 2e-6 beside peers of 5e-5 ... 2e-3, and its 2e-9 of noise (1e-5 of a typical gain) reads as 1e-3 of
    itself.  Such a scalar is measured against `scalar_floor` times the MEDIAN magnitude of the eight (round 3 used the largest).  The
    8 x 176 comparison with the oracle below keeps every tensor on its own norm."""
    whole = float(torch.sqrt(sum(((new[n] - ref[n]) ** 2).sum() for n in ref)) / torch.sqrt(sum((ref[n] ** 2).sum() for n in ref)))
    scal = sorted(float(ref[n].abs().max()) for n in ref if ref[n].numel() == 1)
    floor = scalar_floor * scal[len(scal) // 2] if scal else 0.0
    per = sorted((float((new[n] - ref[n]).norm() / (max(float(ref[n].norm()), floor if ref[n].numel() == 1 else 0.0) + 1e-300)), n) for n in ref)
    print("\n  worst tensors: " + ", ".join(f"{n} {e:.1e} (|ref| {float(ref[n].norm()):.1e}, {ref[n].numel()} el.)" for e, n in per[-show:]))
    return whole, per[len(per) // 2][0], per[-1][0]


def test_backward_is_insensitive_to_the_magnitude_of_the_incoming_gradient(sd_np):
    """one 176-frame crop with the loss multiplied by 1e-3 (a loss-term weight, 1 / accumulation steps, an outer scaler)
    put dL/dscore at ~7e-8, far below fp16's normal range, and the old element-count gate never rescaled B = 1.  The entry point now
    scales by a data-driven power of two itself: the gradients are those of the exact-fp32 backward times 1e-3."""
    args = _crops(1, 176, key=21)
    _, ref = _hip_grads(sd_np, *args, exact=True)
    _, small = _hip_grads(sd_np, *args, loss_factor=1e-3)
    small = {n: g * 1e3 for n, g in small.items()}
    whole, median, worst = _rel(small, ref, scalar_floor=1.0)
    print(f"\nloss x 1e-3 at one crop: whole-gradient relative L2 vs exact fp32 {whole:.2e}, median tensor {median:.2e}, worst {worst:.2e}")
    assert len(ref) == 228 and whole <= 6e-7 and median <= 1.5e-6 and worst <= 1e-4


def test_pretraining_batch_loss_and_every_gradient_vs_oracle_autograd(sd_np):
    """SURVEY 8(f4) at the size it names per GPU slice: 8 crops of 176 frames, full-size decoder, ragged lengths.  Oracle: `loss_t`
    (`oracle/decoder_oracle.py`, pinned to the reference) under torch autograd on the CPU in fp32.  Columns: the default backward
    (f16x3) and the exact-fp32 handle, each against the oracle; bar: whole-gradient relative L2 <= 1e-6."""
    B, T = 8, 176
    x0, mask, cond, spk, t, z = _crops(B, T, key=33)
    sd = {k: v.clone().requires_grad_(True) for k, v in O.to_torch(sd_np).items()}
    loss_ref, _ = O.loss_t(sd, x0, mask, cond, t, spk, z, FULL.n_feats, FULL.beta_min, FULL.beta_max, FULL.pe_scale)
    loss_ref.backward()
    ref = {k: v.grad.detach().double() for k, v in sd.items() if v.grad is not None and k.startswith("estimator.")}
    assert len(ref) == 228
    l_new, new = _hip_grads(sd_np, x0, mask, cond, spk, t, z)
    l_x, ex = _hip_grads(sd_np, x0, mask, cond, spk, t, z, exact=True)
    new = {k: new[k] for k in ref}
    ex = {k: ex[k] for k in ref}
    w_new, med_new, worst_new = _rel(new, ref)
    w_x, med_x, worst_x = _rel(ex, ref)
    print(f"\n8 x 176 crops vs oracle autograd: loss {float(loss_ref):.7f} / f16x3 {l_new:.7f} / exact {l_x:.7f}; "
          f"whole-gradient relative L2 f16x3 {w_new:.2e} (median tensor {med_new:.2e}, worst {worst_new:.2e}), "
          f"exact-fp32 {w_x:.2e} (median {med_x:.2e}, worst {worst_x:.2e})")
    assert abs(l_new - float(loss_ref)) <= 2e-6 * max(1.0, abs(float(loss_ref)))
    assert abs(l_x - float(loss_ref)) <= 2e-6 * max(1.0, abs(float(loss_ref)))
    assert w_new <= 1e-6 and w_x <= 1e-6
    assert med_new <= 3e-6 and med_x <= 3e-6
    # the per-tensor worst, asserted: every one of the 228 tensors within 1e-4 of its own norm, scalars included
    assert worst_new <= 1e-4 and worst_x <= 1e-4
    # ... and against the REFERENCE's own autograd in fp64 on the same crops (tools/make_goldens_r4.py: tensors of <= 1,024 elements whole,
    # larger ones as an odd-strided sample; the reference's fp32 run is within 4.9e-6 of it on every tensor, 3.5e-6 on the gains)
    gold = np.load(os.path.join(GOLD, "grads_full_8x176_fp64.npz"))
    names = [str(n) for n in gold["names"]]
    assert len(names) == 228 and abs(l_new - float(gold["loss_fp64"])) <= 2e-6
    for tag, got in (("f16x3", new), ("exact", ex)):
        worst = (0.0, "")
        for i, n in enumerate(names):
            g64 = torch.from_numpy(gold[f"val_{i}"]).double().reshape(-1)
            mine = got[n].reshape(-1)
            if mine.numel() > 1024:
                mine = mine[::(mine.numel() // 4096 + 1) | 1]
            assert mine.numel() == g64.numel(), n
            worst = max(worst, (float((mine - g64).norm() / (g64.norm() + 1e-300)), n))
        print(f"  {tag} vs the reference's fp64 autograd: worst tensor {worst[1]} {worst[0]:.2e}")
        assert worst[0] <= 1e-4, worst


def test_weight_gradient_chains_on_the_second_stream_give_the_gradients_of_the_one_stream_backward(sd_np, monkeypatch):
    """The eager backward runs each parameter's max -> zero -> wgrad -> unpack chain on the handle's second stream, ordered against the
    data-gradient chain by events in both directions (train_host.inc, BwdCtx).  US_WGRAD_STREAM=0 (read when the handle is created) keeps
    everything on the caller's stream: same gradients up to the order of the fp32 atomics (four ragged crops; every per-level buffer
    of the backward pass is written at least twice)."""
    args = _crops(4, 176, key=57)
    l2, two = _hip_grads(sd_np, *args)
    monkeypatch.setenv("US_WGRAD_STREAM", "0")
    l1_, one = _hip_grads(sd_np, *args)
    assert l1_ == l2 and len(two) == 228
    whole, median, worst = _rel(two, one)
    print(f"\ntwo streams vs one: whole-gradient relative L2 {whole:.2e}, median tensor {median:.2e}, worst {worst:.2e}")
    assert whole <= 2e-7 and worst <= 1e-4
    monkeypatch.delenv("US_WGRAD_STREAM")
    _, again = _hip_grads(sd_np, *args)     # a second two-stream handle: same result again
    w2, _, worst2 = _rel(again, one)
    assert w2 <= 2e-7 and worst2 <= 1e-4


def test_backward_needs_no_zero_fill_for_the_convolution_weight_gradients(sd_np):
    """The gradient blob of a backward call is torch.empty: only its head (the tensors the library accumulates into) is zero-filled, the
    convolution weights behind it are overwritten element by element (us_grad_is_overwritten).  The blob of the second call below
    arrives NaN-filled: every gradient must come out finite and equal to the first call's."""
    args = _crops(1, 176, key=71)
    x0, mask, cond, spk, t, z = args
    m = build(sd_np, train=True)
    eng = m._get_engine()

    def run():
        for p in m.parameters():
            p.grad = None
        with _ReplayRandn([z.to(DEV)]):
            loss, _ = m.loss_t(x0.to(DEV), mask.to(DEV), cond.to(DEV), t.to(DEV), spk.to(DEV))
        loss.backward()
        torch.cuda.synchronize()
        return {n: p.grad.detach().double().cpu() for n, p in m.named_parameters() if p.grad is not None}

    ref = run()
    n = eng.last_grad_blob.numel()
    keys = [k for k in eng.grad_overwritten.__self__.__dict__["_grad_over"]][0]
    flags = eng.grad_overwritten(keys)
    over = [k for k, f in zip(keys, flags) if f]
    assert len(over) == 52                                            # 44 ResnetBlock / final convolutions and res_convs, 8 to_qkv, 6 down / up -- minus the 2-channel first block's
    assert all(k.endswith(".weight") for k in over)
    assert not any(k.endswith(".bias") or ".mlp." in k or ".block.1." in k or k.endswith(".g") for k in over)
    # the second call's blob arrives NaN-filled: torch.empty is wrapped for tensors of exactly the blob's size
    orig_empty, hits = torch.empty, []

    def poisoned_empty(*a, **kw):
        out = orig_empty(*a, **kw)
        if out.dtype == torch.float32 and out.numel() == n and out.is_cuda:
            out.fill_(float("nan"))
            hits.append(1)
        return out

    torch.empty = poisoned_empty
    try:
        new = run()
    finally:
        torch.empty = orig_empty
    assert hits, "the backward did not allocate its blob through torch.empty: the test proves nothing"
    assert all(torch.isfinite(g).all() for g in new.values())
    whole, _, worst = _rel(new, ref)
    assert whole <= 2e-7 and worst <= 1e-4


def test_winograd_split_k_slabs_give_the_same_gradients(sd_np, monkeypatch):
    """US_WINO_SPLITK=1 (off by default: measured slower): the Winograd-domain GEMMs of a one-crop training pass slice K and the
    output transform sums the raw slabs in slice order.  Same loss, gradients equal up to the regrouping of the K sum."""
    args = _crops(1, 176, key=91)
    l0, ref = _hip_grads(sd_np, *args)
    monkeypatch.setenv("US_WINO_SPLITK", "1")
    l1_, got = _hip_grads(sd_np, *args)
    whole, median, worst = _rel(got, ref, scalar_floor=1.0)
    print(f"\nWinograd split-K slabs vs single pass: loss {l1_:.7f} / {l0:.7f}, whole-gradient relative L2 {whole:.2e}, median tensor {median:.2e}, worst {worst:.2e}")
    assert abs(l1_ - l0) <= 2e-6 * max(1.0, abs(l0)) and whole <= 6e-7 and worst <= 1e-4
