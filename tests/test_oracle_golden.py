"""The CPU oracle (oracle/decoder_oracle.py) against every golden vector captured from the reference
decoder by tools/make_goldens.py.  These pins are what entitles the oracle to judge the HIP path."""
import random

import numpy as np
import pytest
import torch

from oracle import decoder_oracle as O
from unitspeech_amd.params import (DecoderConfig, n_params, param_shapes, synthetic_inputs,
                                   synthetic_state_dict)

TINY = DecoderConfig(dim=16)
FULL = DecoderConfig()


@pytest.fixture(scope="module")
def sd_tiny():
    return O.to_torch(synthetic_state_dict(TINY, 0))


@pytest.fixture(scope="module")
def sd_full():
    return O.to_torch(synthetic_state_dict(FULL, 0))


def T(d):
    return {k: torch.from_numpy(np.asarray(v)) for k, v in d.items()}


def test_param_inventory():
    sh = param_shapes(FULL)
    assert len(sh) == 230
    assert n_params(FULL) == 119_145_177


@pytest.mark.parametrize("tag,cfg", [("tiny", TINY), ("full", FULL)])
def test_weight_generator_fingerprint(golden, tag, cfg):
    g = golden(f"weights_fingerprint_{tag}")
    sd = synthetic_state_dict(cfg, 0)
    tot = sum(np.abs(v.astype(np.float64)).sum() for v in sd.values())
    assert abs(tot - g["abs_sum"][0]) <= 1e-9 * g["abs_sum"][0]
    np.testing.assert_array_equal(sd["estimator.final_block.block.0.weight"].ravel()[:8], g["first8"])


@pytest.mark.parametrize("n", [2, 10, 50])
def test_schedule_tables_bit_exact(golden, n):
    g = golden(f"schedule_N{n}")
    tb = O.schedule_tables(n, 0.05, 20.0)
    for k, v in g.items():
        np.testing.assert_array_equal(tb[k].numpy(), v, err_msg=k)


@pytest.mark.parametrize("tag,cfg", [("tiny", TINY), ("full", FULL)])
def test_time_embedding(golden, tag, cfg, sd_tiny, sd_full):
    sd = sd_tiny if tag == "tiny" else sd_full
    g = T(golden(f"temb_{tag}"))
    e = O.sinusoidal_pos_emb(g["t"], cfg.dim, cfg.pe_scale)
    np.testing.assert_array_equal(e.numpy(), g["posemb"].numpy())
    spk = torch.zeros(g["t"].shape[0], 1, cfg.spk_emb_dim)
    te = O.time_embedding(sd, g["t"], spk, cfg.dim, cfg.pe_scale)[:, :cfg.dim]
    np.testing.assert_allclose(te.numpy(), g["mlp"].numpy(), rtol=0, atol=1e-6)


def test_blocks(golden, sd_tiny):
    g = T(golden("blocks_tiny"))
    sd = sd_tiny
    p = "estimator.downs.1"
    tol = dict(rtol=0, atol=2e-6)
    np.testing.assert_allclose(O.block(sd, f"{p}.1.block1", g["x32"], g["mask"]).numpy(), g["block"].numpy(), **tol)
    np.testing.assert_allclose(O.resnet_block(sd, f"{p}.1", g["x32"], g["mask"], g["temb"]).numpy(),
                               g["resnet_same"].numpy(), **tol)
    np.testing.assert_allclose(O.resnet_block(sd, f"{p}.0", g["x16"], g["mask"], g["temb"]).numpy(),
                               g["resnet_proj"].numpy(), **tol)
    np.testing.assert_allclose(O.linear_attention(sd, f"{p}.2", g["x32"]).numpy(), g["attn"].numpy(), **tol)
    import torch.nn.functional as F
    down = F.conv2d(g["x32"] * g["mask"], sd[f"{p}.3.conv.weight"], sd[f"{p}.3.conv.bias"], stride=2, padding=1)
    np.testing.assert_allclose(down.numpy(), g["down"].numpy(), **tol)
    up = F.conv_transpose2d(g["x32"] * g["mask"], sd["estimator.ups.1.3.conv.weight"],
                            sd["estimator.ups.1.3.conv.bias"], stride=2, padding=1)
    np.testing.assert_allclose(up.numpy(), g["up"].numpy(), **tol)


@pytest.mark.parametrize("tag", ["tiny", "full"])
def test_estimator_eval(golden, tag, sd_tiny, sd_full):
    sd = sd_tiny if tag == "tiny" else sd_full
    g = T(golden(f"estimator_{tag}"))
    out = O.estimator_forward(sd, g["x"], g["mask"], g["mu"], g["t"], g["spk_emb"])
    # same ATen kernels in the same order => (near) bit-identical; the fp64 column bounds the noise floor
    assert (out - g["out"]).abs().mean().item() <= 1e-6
    assert (out.double() - g["out_fp64"]).abs().mean().item() <= 2e-6


@pytest.mark.parametrize("w", ["11", "10", "01", "00"])
def test_loop_tiny_cfg_variants(golden, w, sd_tiny):
    g = T(golden(f"loop_tiny_N10_w{w}"))
    out = O.reverse_diffusion(sd_tiny, g["z"], g["mask"], g["cond"], g["spk_emb"], 10,
                              float(g["w_text"]), float(g["w_spk"]), noise=g["noise"])
    l1 = (out - g["out"]).abs().mean().item()
    assert l1 <= 1e-4, l1


def test_loop_tiny_batched_equals_independent_runs(golden, sd_tiny):
    g = T(golden("loop_tiny_N10_B2"))
    out = O.reverse_diffusion(sd_tiny, g["z"], g["mask"], g["cond"], g["spk_emb"], 10, 1.0, 1.0, noise=g["noise"])
    assert (out - g["out"]).abs().mean().item() <= 1e-4


def test_loop_full_N10(golden, sd_full):
    g = T(golden("loop_full_N10"))
    T_ = g["z"].shape[-1]
    inp = T(synthetic_inputs(FULL, 1, T_, seed=5, n_steps=10, lengths=[T_ - 4]))
    assert abs(inp["noise"].double().abs().sum().item() - float(g["noise_abs_sum"])) < 1e-6 * float(g["noise_abs_sum"])
    np.testing.assert_array_equal(inp["z"].numpy(), g["z"].numpy())
    out = O.reverse_diffusion(sd_full, g["z"], g["mask"], g["cond"], g["spk_emb"], 10, 1.0, 1.0, noise=inp["noise"])
    l1 = (out - g["out"]).abs().mean().item()
    assert l1 <= 1e-3, l1          # north-star tolerance (mean|out| ~ 92 for these untrained weights)


@pytest.mark.parametrize("tag,cfg", [("tiny", TINY)])
def test_loss_and_grads(golden, tag, cfg, sd_tiny):
    g = T(golden(f"loss_{tag}"))
    sd = {k: v.clone().requires_grad_(True) for k, v in sd_tiny.items()}
    loss, xt = O.loss_t(sd, g["x0"], g["mask"], g["cond"], g["t"], g["spk_emb"], g["z"], cfg.n_feats)
    assert abs(loss.item() - float(g["loss"])) <= 1e-6
    np.testing.assert_allclose(xt.detach().numpy(), g["xt"].numpy(), rtol=0, atol=1e-6)
    loss.backward()
    for k, v in g.items():
        if k.startswith("grad:"):
            np.testing.assert_allclose(sd[k[5:]].grad.numpy(), v.numpy(), rtol=1e-4, atol=1e-6, err_msg=k)
    gn = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in sd.values() if p.grad is not None))
    assert abs(gn.item() - float(g["grad_norm"])) <= 1e-4 * float(g["grad_norm"])


def test_fine_tune_segment_and_loss(golden, sd_tiny):
    g = T(golden("finetune_tiny"))
    random.seed(int(g["py_seed"]))
    seg = int(g["segment_size"])
    y_cut, y_cut_mask, cond_y = O.fine_tune_segment(g["cond_x"], g["y"], g["y_mask"], g["y_lengths"],
                                                    g["y"].shape[-1], g["attn"], seg, 80)
    t = torch.clamp(g["t_draw"], 1e-5, 1.0 - 1e-5)           # compute_loss, unitspeech/unitspeech.py:407-411
    loss, _ = O.loss_t(sd_tiny, y_cut, y_cut_mask, cond_y, t, g["spk_emb"], g["z_draw"], 80)
    assert abs(loss.item() - float(g["loss"])) <= 1e-6


def test_fine_tune_iteration_clip_and_adam_vs_reference(golden, sd_tiny):
    """G9: oracle loss -> torch autograd -> clip_grad_norm_(1) -> Adam(lr=2e-5) reproduces the reference's gradient norm,
    clipped gradients and updated parameters (finetune.py:163-165)."""
    g = T(golden("finetune_tiny"))
    a = golden("finetune_tiny_adam")
    sd = {k: v.clone().requires_grad_(True) for k, v in sd_tiny.items()}
    opt = torch.optim.Adam(list(sd.values()), lr=2e-5)
    random.seed(int(g["py_seed"]))
    y_cut, y_cut_mask, cond_y = O.fine_tune_segment(g["cond_x"], g["y"], g["y_mask"], g["y_lengths"], g["y"].shape[-1],
                                                    g["attn"], int(g["segment_size"]), 80)
    t = torch.clamp(g["t_draw"], 1e-5, 1.0 - 1e-5)
    loss, _ = O.loss_t(sd, y_cut, y_cut_mask, cond_y, t, g["spk_emb"], g["z_draw"], 80)
    loss.backward()
    with_grad = [v for v in sd.values() if v.grad is not None]
    norm = torch.nn.utils.clip_grad_norm_(with_grad, 1)
    assert abs(float(norm) - float(a["grad_norm"])) <= 1e-5 * float(a["grad_norm"])
    grads = {k: v.grad.clone() for k, v in sd.items() if v.grad is not None}
    opt.step()
    for i, k in enumerate([str(x) for x in a["keys"]]):
        gr, pr = torch.from_numpy(np.asarray(a[f"grad_{i}"])), torch.from_numpy(np.asarray(a[f"param_{i}"]))
        assert (grads[k] - gr).abs().max() <= 2e-5 * gr.abs().max() + 1e-9, k
        # first Adam step moves every element by ~lr: compare the UPDATE, not the parameter
        d_ref, d_got = pr - sd_tiny[k], sd[k].detach() - sd_tiny[k]
        err = (d_got - d_ref).abs()
        solid = gr.abs() > 1e-6                                   # elements with |g| ~ eps = 1e-8 amplify rounding noise
        assert (err[solid].max() <= 2e-7 if solid.any() else True) and err.max() <= 2e-6, k


def test_helpers():
    assert O.fix_len_compatibility(172, 3) == 176 and O.fix_len_compatibility(176, 3) == 176
    m = O.sequence_mask(torch.LongTensor([3, 5]), 6)
    assert m.tolist() == [[True] * 3 + [False] * 3, [True] * 5 + [False]]
    dur = torch.tensor([[2., 1., 3.]])
    path = O.generate_path(dur, torch.ones(1, 3, 6))
    assert path[0].argmax(0).tolist() == [0, 0, 1, 2, 2, 2]
