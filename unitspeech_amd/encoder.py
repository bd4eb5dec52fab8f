"""`Encoder` and `DurationPredictor` of the conditioning producer on the HIP library.

Drop-in for `unitspeech/encoder.py:253-308` (`Encoder`: text encoder and unit encoder are two instances) and
`unitspeech/duration_predictor.py:24-63` (`DurationPredictor`): same constructor arguments, same `state_dict` keys in
the same order (the sub-modules below are parameter containers, exactly as in `unitspeech_amd.unitspeech`), same call
signatures and return values -- `Encoder(x, x_lengths) -> (mu_x, x, x_mask)`, `DurationPredictor(x, x_mask, w=None,
g=spk_emb, reverse=True) -> logw` -- so `UnitSpeech.execute_text_to_speech(phoneme, lengths, spk_emb, text_encoder,
duration_predictor, ...)` takes them where it takes the reference's modules.

Inference only: the modules must be in eval mode (the reference's Dropouts are then the identity), and the duration
predictor's training branch (`reverse=False`: the MSE against log durations, :60-61) is not built.  There is no CPU
fallback: tensors must live on a ROCm device.
"""
from __future__ import annotations

import ctypes as C
import hashlib
from collections import OrderedDict
from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import numpy as np
import torch

from . import _lib

PRENET_LAYERS, PRENET_KERNEL = 3, 5            # unitspeech/encoder.py:283-284


@dataclass(frozen=True)
class EncoderConfig:
    """`Encoder.__init__` arguments; defaults are conf/hydra_config.py:85-105 (n_vocab: 149 symbols + 1 / 1000 units)."""
    n_vocab: int = 150
    n_feats: int = 80
    n_channels: int = 192
    filter_channels: int = 768
    n_heads: int = 2
    n_layers: int = 6
    kernel_size: int = 3
    window_size: Optional[int] = 4


@dataclass(frozen=True)
class DurationPredictorConfig:
    """`DurationPredictor.__init__` arguments; defaults are conf/hydra_config.py:112-116."""
    in_channels: int = 192
    filter_channels: int = 256
    kernel_size: int = 3
    spk_emb_dim: int = 256


def encoder_state_shapes(cfg: EncoderConfig) -> "OrderedDict[str, Tuple[int, ...]]":
    """state_dict keys and shapes in the reference module's registration order (unitspeech/encoder.py:270-291)."""
    c, d = cfg.n_channels, cfg.n_channels // cfg.n_heads
    sh: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    sh["emb.weight"] = (cfg.n_vocab, c)
    for i in range(PRENET_LAYERS):
        sh[f"prenet.conv_layers.{i}.weight"] = (c, c, PRENET_KERNEL)
        sh[f"prenet.conv_layers.{i}.bias"] = (c,)
    for i in range(PRENET_LAYERS):
        sh[f"prenet.norm_layers.{i}.gamma"] = (c,)
        sh[f"prenet.norm_layers.{i}.beta"] = (c,)
    sh["prenet.proj.weight"] = (c, c, 1)
    sh["prenet.proj.bias"] = (c,)
    for i in range(cfg.n_layers):
        p = f"encoder.attn_layers.{i}"
        if cfg.window_size:
            sh[p + ".emb_rel_k"] = (1, 2 * cfg.window_size + 1, d)
            sh[p + ".emb_rel_v"] = (1, 2 * cfg.window_size + 1, d)
        for n in "qkvo":
            sh[f"{p}.conv_{n}.weight"] = (c, c, 1)
            sh[f"{p}.conv_{n}.bias"] = (c,)
    for i in range(cfg.n_layers):
        sh[f"encoder.norm_layers_1.{i}.gamma"] = (c,)
        sh[f"encoder.norm_layers_1.{i}.beta"] = (c,)
    for i in range(cfg.n_layers):
        p = f"encoder.ffn_layers.{i}"
        sh[p + ".conv_1.weight"] = (cfg.filter_channels, c, cfg.kernel_size)
        sh[p + ".conv_1.bias"] = (cfg.filter_channels,)
        sh[p + ".conv_2.weight"] = (c, cfg.filter_channels, cfg.kernel_size)
        sh[p + ".conv_2.bias"] = (c,)
    for i in range(cfg.n_layers):
        sh[f"encoder.norm_layers_2.{i}.gamma"] = (c,)
        sh[f"encoder.norm_layers_2.{i}.beta"] = (c,)
    sh["proj_m.weight"] = (cfg.n_feats, c, 1)
    sh["proj_m.bias"] = (cfg.n_feats,)
    return sh


def duration_predictor_state_shapes(cfg: DurationPredictorConfig) -> "OrderedDict[str, Tuple[int, ...]]":
    f, cin = cfg.filter_channels, cfg.in_channels + cfg.spk_emb_dim
    return OrderedDict([("conv_1.weight", (f, cin, cfg.kernel_size)), ("conv_1.bias", (f,)), ("norm_1.gamma", (f,)), ("norm_1.beta", (f,)),
                        ("conv_2.weight", (f, f, cfg.kernel_size)), ("conv_2.bias", (f,)), ("norm_2.gamma", (f,)), ("norm_2.beta", (f,)),
                        ("proj.weight", (1, f, 1)), ("proj.bias", (1,))])


def _synthetic(shapes, seed: int, tag: str) -> Dict[str, np.ndarray]:
    """Seeded weights from (seed, tensor name) with NumPy's Philox stream, like params.synthetic_state_dict: every tensor is
    non-trivial (the reference initialises `prenet.proj` to zero, which would hide the whole prenet from a parity test)."""
    out = OrderedDict()
    for name, shape in shapes.items():
        key = int.from_bytes(hashlib.sha256(f"{tag}/{seed}/{name}".encode()).digest()[:8], "little")
        g = np.random.Generator(np.random.Philox(key=key))
        z = g.standard_normal(shape, dtype=np.float32)
        if name.endswith("gamma"):
            v = 1.0 + 0.1 * z
        elif name.endswith(("beta", "bias")):
            v = 0.1 * z
        elif name == "emb.weight":
            v = z * shape[1] ** -0.5                       # encoder.py:284
        elif "emb_rel" in name:
            v = z * shape[2] ** -0.5                       # :88-92
        else:                                             # Conv1d [out, in, k]
            v = z * (shape[1] * shape[2]) ** -0.5
        out[name] = v.astype(np.float32)
    return out


def synthetic_encoder_state_dict(cfg: EncoderConfig, seed: int = 0) -> Dict[str, np.ndarray]:
    return _synthetic(encoder_state_shapes(cfg), seed, "encoder")


def synthetic_duration_predictor_state_dict(cfg: DurationPredictorConfig, seed: int = 0) -> Dict[str, np.ndarray]:
    return _synthetic(duration_predictor_state_shapes(cfg), seed, "duration_predictor")


# ---- parameter containers (state_dict layout only; the arithmetic lives in csrc/frontend.hip) ----------------------------

class _Conv1dParams(torch.nn.Module):
    def __init__(self, cin, cout, k):
        super().__init__()
        self.weight = torch.nn.Parameter(torch.empty(cout, cin, k).normal_(0, (cin * k) ** -0.5))
        self.bias = torch.nn.Parameter(torch.zeros(cout))


class _NormParams(torch.nn.Module):
    def __init__(self, c):
        super().__init__()
        self.gamma = torch.nn.Parameter(torch.ones(c))
        self.beta = torch.nn.Parameter(torch.zeros(c))


class _EmbParams(torch.nn.Module):
    def __init__(self, n, c):
        super().__init__()
        self.weight = torch.nn.Parameter(torch.empty(n, c).normal_(0, c ** -0.5))


class _Prenet(torch.nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv_layers = torch.nn.ModuleList([_Conv1dParams(c, c, PRENET_KERNEL) for _ in range(PRENET_LAYERS)])
        self.norm_layers = torch.nn.ModuleList([_NormParams(c) for _ in range(PRENET_LAYERS)])
        self.proj = _Conv1dParams(c, c, 1)
        with torch.no_grad():
            self.proj.weight.zero_()           # encoder.py:54-55


class _AttnParams(torch.nn.Module):
    def __init__(self, c, n_heads, window):
        super().__init__()
        d = c // n_heads
        if window:
            self.emb_rel_k = torch.nn.Parameter(torch.randn(1, 2 * window + 1, d) * d ** -0.5)
            self.emb_rel_v = torch.nn.Parameter(torch.randn(1, 2 * window + 1, d) * d ** -0.5)
        self.conv_q = _Conv1dParams(c, c, 1)
        self.conv_k = _Conv1dParams(c, c, 1)
        self.conv_v = _Conv1dParams(c, c, 1)
        self.conv_o = _Conv1dParams(c, c, 1)


class _FfnParams(torch.nn.Module):
    def __init__(self, c, f, k):
        super().__init__()
        self.conv_1 = _Conv1dParams(c, f, k)
        self.conv_2 = _Conv1dParams(f, c, k)


class _Transformer(torch.nn.Module):
    def __init__(self, c, f, n_heads, n_layers, k, window):
        super().__init__()
        self.attn_layers = torch.nn.ModuleList([_AttnParams(c, n_heads, window) for _ in range(n_layers)])
        self.norm_layers_1 = torch.nn.ModuleList([_NormParams(c) for _ in range(n_layers)])
        self.ffn_layers = torch.nn.ModuleList([_FfnParams(c, f, k) for _ in range(n_layers)])
        self.norm_layers_2 = torch.nn.ModuleList([_NormParams(c) for _ in range(n_layers)])


class _FrontEndModule(torch.nn.Module):
    """Owns one `us_frontend_handle` and pushes parameters whose storage or version changed since the last call."""

    def _init_engine(self):
        self._h = C.c_void_p()
        self._device = None
        self._versions = {}

    def _create(self, lib, device):          # overridden
        raise NotImplementedError

    def _sync(self, device: torch.device):
        if device.type != "cuda":
            raise RuntimeError("the HIP front end needs tensors on a ROCm device (no CPU fallback); got " + str(device))
        if self.training:
            raise RuntimeError(f"{type(self).__name__} is inference-only (the reference's Dropout layers are not built): call .eval()")
        lib = _lib.load()
        if not self._h or self._device != device:
            self._close()
            with torch.cuda.device(device):
                self._create(lib, device)
            self._device, self._versions = device, {}
        stream = C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
        # the handle allocates its weight store on the CURRENT device and refuses calls made under another one (frontend.hip: fe_device)
        with torch.cuda.device(device):
            for key, t in self.state_dict(keep_vars=True).items():
                tag = (t.data_ptr(), t._version, t.device)
                if self._versions.get(key) == tag:
                    continue
                in_place = t.dtype == torch.float32 and t.device == device and t.is_contiguous()
                src = t.detach() if in_place else t.detach().to(device=device, dtype=torch.float32).contiguous()
                shape = (C.c_int64 * src.dim())(*src.shape)
                rc = lib.us_frontend_load_weight(self._h, key.encode(), src.data_ptr(), shape, src.dim(), stream)
                self._check(lib, rc, f"us_frontend_load_weight({key})")
                if not in_place:
                    torch.cuda.current_stream(device).synchronize()      # the temporary must outlive the copy
                self._versions[key] = tag
        return lib, stream

    def _workspace(self, lib, device, B, L):
        """Caller-owned activation scratch of one forward call (torch's caching allocator: no hipMalloc / hipFree in the call)."""
        n = int(lib.us_frontend_workspace_bytes(self._h, B, L))
        ws = getattr(self, "_ws", None)
        if ws is None or ws.numel() < n or ws.device != device:
            self._ws = None
            self._ws = ws = torch.empty(n, dtype=torch.uint8, device=device)
        return ws

    def _check(self, lib, rc, what):
        if rc != _lib.US_OK:
            msg = lib.us_frontend_last_error(self._h)
            raise RuntimeError(f"libunitspeech_hip: {what} failed with {_lib.ERRORS.get(rc, rc)}: {msg.decode() if msg else ''}")

    def _close(self):
        if getattr(self, "_h", None):
            _lib.load().us_frontend_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self._close()
        except Exception:
            pass


class Encoder(_FrontEndModule):
    """`unitspeech/encoder.py:253` `Encoder(n_vocab, n_feats, n_channels, filter_channels, n_heads, n_layers, kernel_size, p_dropout,
    n_contentvec=0, window_size=None)`."""

    def __init__(self, n_vocab, n_feats, n_channels, filter_channels, n_heads, n_layers, kernel_size, p_dropout=0.0, n_contentvec=0,
                 window_size=None):
        super().__init__()
        if n_contentvec:
            raise NotImplementedError("Encoder(n_contentvec > 0) (a Linear front instead of the Embedding, encoder.py:281) is not built: "
                                      "no configuration of the reference uses it")
        self.cfg = EncoderConfig(int(n_vocab), int(n_feats), int(n_channels), int(filter_channels), int(n_heads), int(n_layers),
                                 int(kernel_size), int(window_size) if window_size else None)
        self.p_dropout = p_dropout
        self.emb = _EmbParams(n_vocab, n_channels)
        self.prenet = _Prenet(n_channels)
        self.encoder = _Transformer(n_channels, filter_channels, n_heads, n_layers, kernel_size, window_size)
        self.proj_m = _Conv1dParams(n_channels, n_feats, 1)
        self._init_engine()

    def _create(self, lib, device):
        c = _lib.us_encoder_config(self.cfg.n_vocab, self.cfg.n_feats, self.cfg.n_channels, self.cfg.filter_channels, self.cfg.n_heads,
                                   self.cfg.n_layers, self.cfg.kernel_size, self.cfg.window_size or 0)
        _lib.check(lib.us_encoder_create(C.byref(self._h), C.byref(c)), None, "us_encoder_create")

    @torch.no_grad()
    def forward(self, x, x_lengths):
        """x [B, L] symbol ids, x_lengths [B] -> (mu_x [B, n_feats, L], x [B, n_channels, L], x_mask [B, 1, L])."""
        if x.dim() != 2 or x_lengths.dim() != 1 or x_lengths.shape[0] != x.shape[0]:
            raise ValueError(f"Encoder: expected ids [B, L] and lengths [B], got {tuple(x.shape)} and {tuple(x_lengths.shape)}")
        device = x.device
        lib, stream = self._sync(device)
        b, l = x.shape
        ids = x.to(torch.int64).contiguous()
        lens = x_lengths.to(device=device, dtype=torch.int64).contiguous()
        mu_x = torch.empty(b, self.cfg.n_feats, l, device=device)
        h = torch.empty(b, self.cfg.n_channels, l, device=device)
        mask = torch.empty(b, 1, l, device=device)
        ws = self._workspace(lib, device, b, l)
        with torch.cuda.device(device):
            rc = lib.us_encoder_forward(self._h, ids.data_ptr(), lens.data_ptr(), mu_x.data_ptr(), h.data_ptr(), mask.data_ptr(), b, l,
                                        ws.data_ptr(), ws.numel(), stream)
        self._check(lib, rc, "us_encoder_forward")
        return mu_x, h, mask


class DurationPredictor(_FrontEndModule):
    """`unitspeech/duration_predictor.py:24` `DurationPredictor(in_channels, filter_channels, kernel_size, p_dropout, spk_emb_dim=0)`."""

    def __init__(self, in_channels, filter_channels, kernel_size, p_dropout=0.0, spk_emb_dim=0):
        super().__init__()
        self.cfg = DurationPredictorConfig(int(in_channels), int(filter_channels), int(kernel_size), int(spk_emb_dim))
        self.p_dropout = p_dropout
        cin = in_channels + spk_emb_dim
        self.conv_1 = _Conv1dParams(cin, filter_channels, kernel_size)
        self.norm_1 = _NormParams(filter_channels)
        self.conv_2 = _Conv1dParams(filter_channels, filter_channels, kernel_size)
        self.norm_2 = _NormParams(filter_channels)
        self.proj = _Conv1dParams(filter_channels, 1, 1)
        self._init_engine()

    def _create(self, lib, device):
        c = _lib.us_duration_config(self.cfg.in_channels, self.cfg.filter_channels, self.cfg.kernel_size, self.cfg.spk_emb_dim)
        _lib.check(lib.us_duration_predictor_create(C.byref(self._h), C.byref(c)), None, "us_duration_predictor_create")

    @torch.no_grad()
    def forward(self, x, x_mask, w=None, g=None, reverse=False):
        """x [B, in_channels, L], x_mask [B, 1, L], g [B, 1, spk_emb_dim] -> logw [B, 1, L] (reverse=True only)."""
        if not reverse:
            raise NotImplementedError("DurationPredictor(reverse=False) (the training loss, duration_predictor.py:60-61) is not built")
        if x.dim() != 3 or x.shape[1] != self.cfg.in_channels or x_mask.shape != (x.shape[0], 1, x.shape[2]):
            raise ValueError(f"DurationPredictor: expected x [B, {self.cfg.in_channels}, L] and x_mask [B, 1, L], got {tuple(x.shape)} "
                             f"and {tuple(x_mask.shape)}")
        if (g is None) != (self.cfg.spk_emb_dim == 0) or (g is not None and tuple(g.shape) != (x.shape[0], 1, self.cfg.spk_emb_dim)):
            raise ValueError(f"DurationPredictor: g must be [B, 1, {self.cfg.spk_emb_dim}] (None iff spk_emb_dim == 0)")
        device = x.device
        lib, stream = self._sync(device)
        f32 = lambda t: t.detach().to(device=device, dtype=torch.float32).contiguous()
        xs, ms = f32(x), f32(x_mask)
        gs = f32(g) if g is not None else None
        logw = torch.empty(x.shape[0], 1, x.shape[2], device=device)
        ws = self._workspace(lib, device, x.shape[0], x.shape[2])
        with torch.cuda.device(device):
            rc = lib.us_duration_predictor_forward(self._h, xs.data_ptr(), ms.data_ptr(), gs.data_ptr() if gs is not None else None,
                                                   logw.data_ptr(), x.shape[0], x.shape[2], ws.data_ptr(), ws.numel(), stream)
        self._check(lib, rc, "us_duration_predictor_forward")
        return logw
