"""Conditioning producer's learned modules: `Encoder` (unitspeech/encoder.py:253-308) and `DurationPredictor`
(unitspeech/duration_predictor.py:24-63).

CPU: the oracle restatement (oracle/frontend_oracle.py) against outputs of the REFERENCE classes (tests/golden/frontend_*.npz,
tools/make_goldens_frontend.py), and the host side of the drop-in modules (state_dict layout, argument checks).
GPU: the HIP modules through the C ABI against the same goldens and against the oracle on other shapes; the reference's own
`execute_text_to_speech` output with the HIP encoder + duration predictor + decoder in place of the reference's modules.
"""
import numpy as np
import pytest
import torch

from oracle import frontend_oracle as FO
from unitspeech_amd.encoder import (DurationPredictor, DurationPredictorConfig, Encoder, EncoderConfig, duration_predictor_state_shapes,
                                    encoder_state_shapes, synthetic_duration_predictor_state_dict, synthetic_encoder_state_dict)
from unitspeech_amd.params import DecoderConfig, synthetic_state_dict

TINY_E = EncoderConfig(n_vocab=20, n_feats=8, n_channels=16, filter_channels=32, n_heads=2, n_layers=2, kernel_size=3, window_size=4)
TINY_D = DurationPredictorConfig(in_channels=16, filter_channels=24, kernel_size=3, spk_emb_dim=12)
NOWIN_E = EncoderConfig(n_vocab=20, n_feats=8, n_channels=16, filter_channels=32, n_heads=2, n_layers=2, kernel_size=3, window_size=None)
NOWIN_D = DurationPredictorConfig(in_channels=16, filter_channels=24, kernel_size=3, spk_emb_dim=0)
CASES = {"frontend_tiny": (TINY_E, TINY_D), "frontend_full": (EncoderConfig(), DurationPredictorConfig()), "frontend_nowin": (NOWIN_E, NOWIN_D)}
# fp32 against fp32 with a different summation order (the reference's convolutions run through oneDNN / ATen matmul): six
# LayerNorm'ed transformer blocks keep every activation O(1), so an absolute bound is the natural one
TOL = 2e-5


def T(d):
    return {k: torch.from_numpy(np.asarray(v)) for k, v in d.items()}


def tsd(d):
    return {k: torch.from_numpy(v) for k, v in d.items()}


def oracle_run(ecfg, dcfg, ids, lengths, spk):
    esd, dsd = tsd(synthetic_encoder_state_dict(ecfg, 0)), tsd(synthetic_duration_predictor_state_dict(dcfg, 0))
    mu_x, x, x_mask = FO.encoder_forward(esd, ids, lengths, n_heads=ecfg.n_heads, n_layers=ecfg.n_layers, kernel_size=ecfg.kernel_size,
                                         window_size=ecfg.window_size)
    return mu_x, x, x_mask, FO.duration_predictor_forward(dsd, x, x_mask, spk)


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_matches_the_reference_modules(golden, name):
    g = T(golden(name))
    ecfg, dcfg = CASES[name]
    mu_x, x, x_mask, logw = oracle_run(ecfg, dcfg, g["ids"], g["lengths"], g.get("spk_emb"))
    assert torch.equal(x_mask, g["x_mask"])
    for got, key in ((mu_x, "mu_x"), (x, "x"), (logw, "logw")):
        assert got.shape == g[key].shape
        assert float((got - g[key]).abs().max()) <= TOL, key
    # padded symbols are exactly zero in every output (encoder.py:250,307; duration_predictor.py:59)
    pad = x_mask == 0
    assert float(mu_x.abs().mul(pad).max()) == 0.0 and float(x.abs().mul(pad).max()) == 0.0 and float(logw.abs().mul(pad).max()) == 0.0


def test_state_dict_layout_is_the_reference_modules():
    """Key order and shapes: the goldens' generator asserts the same lists against the reference classes."""
    for ecfg, dcfg in CASES.values():
        enc = Encoder(ecfg.n_vocab, ecfg.n_feats, ecfg.n_channels, ecfg.filter_channels, ecfg.n_heads, ecfg.n_layers, ecfg.kernel_size, 0.1,
                      window_size=ecfg.window_size)
        assert [(k, tuple(v.shape)) for k, v in enc.state_dict().items()] == list(encoder_state_shapes(ecfg).items())
        dp = DurationPredictor(dcfg.in_channels, dcfg.filter_channels, dcfg.kernel_size, 0.1, spk_emb_dim=dcfg.spk_emb_dim)
        assert [(k, tuple(v.shape)) for k, v in dp.state_dict().items()] == list(duration_predictor_state_shapes(dcfg).items())
        enc.load_state_dict(tsd(synthetic_encoder_state_dict(ecfg, 0)), strict=True)
        dp.load_state_dict(tsd(synthetic_duration_predictor_state_dict(dcfg, 0)), strict=True)
    full = encoder_state_shapes(EncoderConfig())
    n_params = sum(int(np.prod(s)) for s in full.values())
    assert n_params == sum(p.numel() for p in Encoder(150, 80, 192, 768, 2, 6, 3, 0.1, window_size=4).parameters())


def test_c_abi_key_tables_match_the_state_dict_layout():
    """Handle creation does no GPU work: the library's own key list (what `us_frontend_load_weight` accepts, in order) against the
    layout the goldens' generator asserted equal to the reference modules'; bad configurations are refused with a message."""
    import ctypes as C
    from unitspeech_amd import _lib
    lib = _lib.load()
    for ecfg, dcfg in CASES.values():
        h = C.c_void_p()
        c = _lib.us_encoder_config(ecfg.n_vocab, ecfg.n_feats, ecfg.n_channels, ecfg.filter_channels, ecfg.n_heads, ecfg.n_layers,
                                   ecfg.kernel_size, ecfg.window_size or 0)
        assert lib.us_encoder_create(C.byref(h), C.byref(c)) == 0
        keys = [lib.us_frontend_weight_key(h, i).decode() for i in range(lib.us_frontend_num_weights(h))]
        assert keys == list(encoder_state_shapes(ecfg)) and lib.us_frontend_weight_key(h, len(keys)) is None
        assert lib.us_duration_predictor_forward(h, None, None, None, None, 1, 1, None, 0, None) == -1        # wrong kind of handle: EINVAL
        assert lib.us_frontend_destroy(h) == 0
        h = C.c_void_p()
        c = _lib.us_duration_config(dcfg.in_channels, dcfg.filter_channels, dcfg.kernel_size, dcfg.spk_emb_dim)
        assert lib.us_duration_predictor_create(C.byref(h), C.byref(c)) == 0
        keys = [lib.us_frontend_weight_key(h, i).decode() for i in range(lib.us_frontend_num_weights(h))]
        assert keys == list(duration_predictor_state_shapes(dcfg))
        assert lib.us_frontend_destroy(h) == 0
    h = C.c_void_p()
    bad = _lib.us_encoder_config(150, 80, 192, 768, 5, 6, 3, 4)             # 192 channels over 5 heads
    assert lib.us_encoder_create(C.byref(h), C.byref(bad)) == -1 and b"bad configuration" in lib.us_frontend_last_error(None)
    even = _lib.us_duration_config(192, 256, 4, 256)                        # even kernel size
    assert lib.us_duration_predictor_create(C.byref(h), C.byref(even)) == -1


def test_host_side_argument_checks():
    enc = Encoder(20, 8, 16, 32, 2, 2, 3, 0.1, window_size=4).eval()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        enc(torch.zeros(1, 4, dtype=torch.long), torch.LongTensor([4]))
    with pytest.raises(ValueError):
        enc(torch.zeros(4, dtype=torch.long), torch.LongTensor([4]))
    with pytest.raises(NotImplementedError):
        Encoder(20, 8, 16, 32, 2, 2, 3, 0.1, n_contentvec=4)
    dp = DurationPredictor(16, 24, 3, 0.1, spk_emb_dim=12).eval()
    with pytest.raises(NotImplementedError):
        dp(torch.zeros(1, 16, 4), torch.ones(1, 1, 4), w=torch.ones(1, 1, 4), g=torch.zeros(1, 1, 12))
    with pytest.raises(ValueError):
        dp(torch.zeros(1, 16, 4), torch.ones(1, 1, 4), g=None, reverse=True)
    with pytest.raises(ValueError):
        dp(torch.zeros(1, 15, 4), torch.ones(1, 1, 4), g=torch.zeros(1, 1, 12), reverse=True)


# ---- GPU ------------------------------------------------------------------------------------------------------------------

def hip_modules(ecfg, dcfg, seed=0, dev="cuda"):
    enc = Encoder(ecfg.n_vocab, ecfg.n_feats, ecfg.n_channels, ecfg.filter_channels, ecfg.n_heads, ecfg.n_layers, ecfg.kernel_size, 0.1,
                  window_size=ecfg.window_size)
    enc.load_state_dict(tsd(synthetic_encoder_state_dict(ecfg, seed)), strict=True)
    dp = DurationPredictor(dcfg.in_channels, dcfg.filter_channels, dcfg.kernel_size, 0.1, spk_emb_dim=dcfg.spk_emb_dim)
    dp.load_state_dict(tsd(synthetic_duration_predictor_state_dict(dcfg, seed)), strict=True)
    return enc.to(dev).eval(), dp.to(dev).eval()


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(CASES))
def test_hip_modules_match_the_reference_goldens(golden, name):
    g = T(golden(name))
    ecfg, dcfg = CASES[name]
    enc, dp = hip_modules(ecfg, dcfg)
    spk = g["spk_emb"].cuda() if "spk_emb" in g else None
    mu_x, x, x_mask = enc(g["ids"].cuda(), g["lengths"].cuda())
    logw = dp(x, x_mask, w=None, g=spk, reverse=True)
    assert torch.equal(x_mask.cpu(), g["x_mask"])
    for got, key in ((mu_x, "mu_x"), (x, "x"), (logw, "logw")):
        assert got.shape == g[key].shape
        assert float((got.cpu() - g[key]).abs().max()) <= TOL, key
    pad = (x_mask == 0)
    assert float(mu_x.abs().mul(pad).max()) == 0.0 and float(x.abs().mul(pad).max()) == 0.0 and float(logw.abs().mul(pad).max()) == 0.0


@pytest.mark.gpu
@pytest.mark.parametrize("B,L,lengths", [(1, 1, [1]), (2, 5, [5, 2]), (4, 37, [37, 1, 20, 36]), (1, 300, [293])])
def test_hip_modules_match_the_oracle_on_other_shapes(B, L, lengths):
    """Single symbol, lengths at and below the window, a batch whose items differ in length, a long utterance."""
    ecfg, dcfg = TINY_E, TINY_D
    gen = np.random.Generator(np.random.Philox(key=100 + L))
    ids = torch.from_numpy(gen.integers(0, ecfg.n_vocab, size=(B, L)).astype(np.int64))
    spk = torch.from_numpy(gen.standard_normal((B, 1, dcfg.spk_emb_dim), dtype=np.float32))
    lens = torch.LongTensor(lengths)
    ref = oracle_run(ecfg, dcfg, ids, lens, spk)
    enc, dp = hip_modules(ecfg, dcfg)
    mu_x, x, x_mask = enc(ids.cuda(), lens.cuda())
    logw = dp(x, x_mask, w=None, g=spk.cuda(), reverse=True)
    for got, want, key in zip((mu_x, x, x_mask, logw), ref, ("mu_x", "x", "x_mask", "logw")):
        assert float((got.cpu() - want).abs().max()) <= TOL, key
    # an item's result does not depend on what it is batched with, bit for bit (padding is masked, not averaged in)
    if B > 1:
        b = B - 1
        mu1, x1, m1 = enc(ids[b:b + 1, :lengths[b]].cuda(), lens[b:b + 1].cuda())
        assert torch.equal(mu1, mu_x[b:b + 1, :, :lengths[b]]) and torch.equal(x1, x[b:b + 1, :, :lengths[b]])
        lw1 = dp(x1, m1, w=None, g=spk[b:b + 1].cuda(), reverse=True)
        assert torch.equal(lw1, logw[b:b + 1, :, :lengths[b]])


@pytest.mark.gpu
def test_hip_modules_follow_weight_updates_and_report_errors():
    enc, dp = hip_modules(TINY_E, TINY_D)
    ids, lens = torch.arange(6).view(1, 6).cuda(), torch.LongTensor([6]).cuda()
    mu0, _, _ = enc(ids, lens)
    with torch.no_grad():
        enc.proj_m.bias.add_(1.0)
    mu1, _, _ = enc(ids, lens)
    assert float((mu1 - mu0 - 1.0).abs().max()) <= 1e-6
    bad = ids.clone()
    bad[0, 2] = TINY_E.n_vocab + 3                       # outside the embedding table: NaNs, not a silent clamp
    mu2, _, _ = enc(bad, lens)
    assert not torch.isfinite(mu2).all()
    enc.train()
    with pytest.raises(RuntimeError, match="inference-only"):
        enc(ids, lens)


@pytest.mark.gpu
def test_c_abi_reports_unknown_keys_wrong_shapes_and_missing_weights():
    import ctypes as C
    from unitspeech_amd import _lib
    lib = _lib.load()
    h = C.c_void_p()
    c = _lib.us_duration_config(16, 24, 3, 12)
    assert lib.us_duration_predictor_create(C.byref(h), C.byref(c)) == 0
    w = torch.zeros(24, 28, 3, device="cuda")
    shp = (C.c_int64 * 3)(24, 28, 3)
    assert lib.us_frontend_load_weight(h, b"conv_9.weight", w.data_ptr(), shp, 3, None) == -2            # ENOKEY
    bad = (C.c_int64 * 3)(24, 27, 3)
    assert lib.us_frontend_load_weight(h, b"conv_1.weight", w.data_ptr(), bad, 3, None) == -3            # ESHAPE
    assert lib.us_frontend_load_weight(h, b"conv_1.weight", w.data_ptr(), shp, 3, None) == 0
    x, m, g, out = (torch.zeros(n, device="cuda") for n in (16 * 4, 4, 12, 4))
    assert lib.us_duration_predictor_forward(h, x.data_ptr(), m.data_ptr(), g.data_ptr(), out.data_ptr(), 1, 4, None, 0, None) == -4      # EWEIGHTS
    assert b"has not been loaded" in lib.us_frontend_last_error(h)
    assert lib.us_duration_predictor_forward(h, x.data_ptr(), m.data_ptr(), None, out.data_ptr(), 1, 4, None, 0, None) == -1               # g missing
    torch.cuda.synchronize()
    assert lib.us_frontend_destroy(h) == 0


@pytest.mark.gpu
def test_text_to_speech_with_the_hip_modules_matches_the_reference(golden):
    """The reference's `execute_text_to_speech` (unitspeech/unitspeech.py:413-450) run with the reference's Encoder, DurationPredictor and
    decoder, against the same call with all three replaced by the HIP modules."""
    from unitspeech_amd import UnitSpeech
    g = golden("tts_modules_tiny")
    cfg = DecoderConfig(dim=16)
    ecfg = EncoderConfig(n_vocab=20, n_feats=cfg.n_feats, n_channels=16, filter_channels=32, n_heads=2, n_layers=2, kernel_size=3, window_size=4)
    dcfg = DurationPredictorConfig(in_channels=16, filter_channels=24, kernel_size=3, spk_emb_dim=cfg.spk_emb_dim)
    enc, dp = hip_modules(ecfg, dcfg, seed=1)
    model = UnitSpeech(cfg.n_feats, cfg.dim, list(cfg.dim_mults), cfg.beta_min, cfg.beta_max, cfg.pe_scale, cfg.spk_emb_dim)
    model.load_state_dict(tsd(synthetic_state_dict(cfg, 0)))
    model = model.cuda().eval()
    ids, lens, spk = (torch.from_numpy(g[k]).cuda() for k in ("ids", "lengths", "spk_emb"))
    n_steps, tp = int(g["n_steps"]), int(g["tp"])
    gz = np.random.Generator(np.random.Philox(key=int(g["noise_key"])))
    z = torch.from_numpy(gz.standard_normal((1, cfg.n_feats, tp), dtype=np.float32)).cuda()
    noise = torch.from_numpy(np.stack([gz.standard_normal((1, cfg.n_feats, tp), dtype=np.float32) for _ in range(n_steps)])).cuda()
    orig = torch.randn_like                       # z is the reference's randn_like draw (:441)
    torch.randn_like = lambda *a, **k: z.clone()
    try:
        enc_out, dec_out, attn = model.execute_text_to_speech(ids, lens, spk, enc, dp, len(cfg.dim_mults) - 1, diffusion_steps=n_steps,
                                                              length_scale=1.0, text_gradient_scale=1.0, spk_gradient_scale=1.0, noise=noise)
    finally:
        torch.randn_like = orig
    assert enc_out.shape[-1] == int(g["frames"]) and torch.equal(attn.cpu(), torch.from_numpy(g["attn"]))
    assert float((enc_out.cpu() - torch.from_numpy(g["enc_out"])).abs().max()) <= TOL
    ref = torch.from_numpy(g["dec_out"])
    assert float((dec_out.cpu() - ref).abs().max()) <= 1e-4 * float(ref.abs().max())
