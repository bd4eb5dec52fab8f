"""Checkpoint formats of the decoder: read what the reference writes, write what it reads.

Two layouts exist in the reference, both plain `torch.save` dicts whose "model" entry is the decoder's `state_dict`
(230 tensors at the full size; the schedule tables are non-persistent buffers and never appear, `unitspeech/unitspeech.py:271`):

  trainer       `train_STEP1.py:297-304`: {"model", "spk_emb": speaker_embeddings.state_dict() (= {"weight": [n_speakers, D]}),
                "mel_min", "mel_max", "iteration"}
  fine-tuned    `finetune.py:167-173`: the dict that was loaded, with "model", "mel_min", "mel_max" and "spk_emb" (the adapted
                speaker's [1, 1, D] embedding) overwritten in place -- so any other key of the pre-trained file ("iteration")
                survives in its original position

`inference.py:66-74,107-108,124` reads "model", "mel_max", "mel_min", "spk_emb" back.  The packed fp32 blob of
`sharding.pack_state_dict` (payload of the RCCL weight broadcast) is a third, in-memory form of the same tensors.
"""
from __future__ import annotations

from collections import OrderedDict
from dataclasses import dataclass, field
from typing import Any, Dict, Optional

import torch

from .params import DecoderConfig, param_shapes


@dataclass
class DecoderCheckpoint:
    model: "OrderedDict[str, torch.Tensor]"
    spk_emb: Any                      # [1, 1, D] tensor (fine-tuned) or {"weight": [n_speakers, D]} (trainer)
    mel_min: torch.Tensor
    mel_max: torch.Tensor
    iteration: Optional[int] = None
    raw: Dict[str, Any] = field(default_factory=dict)     # the file's dict as loaded (all keys, original order)

    def speaker_embedding(self, speaker_id: int = 0) -> torch.Tensor:
        """[1, 1, D] embedding the decoder is conditioned on: the stored tensor, or row `speaker_id` of the trainer's table."""
        if isinstance(self.spk_emb, dict):
            return self.spk_emb["weight"][int(speaker_id)].reshape(1, 1, -1)
        t = self.spk_emb
        return t.reshape(1, 1, -1) if t.numel() == t.shape[-1] else t[int(speaker_id)].reshape(1, 1, -1)


def infer_config(model_sd: Dict[str, torch.Tensor], beta_min=0.05, beta_max=20.0, pe_scale=1000.0) -> DecoderConfig:
    """Constructor arguments recovered from tensor shapes (the reference keeps them in its config files, conf/hydra_config.py:122-131)."""
    dim = int(model_sd["estimator.mlp.2.weight"].shape[0])
    mults = []
    while f"estimator.downs.{len(mults)}.0.block1.block.0.weight" in model_sd:
        mults.append(int(model_sd[f"estimator.downs.{len(mults)}.0.block1.block.0.weight"].shape[0]) // dim)
    return DecoderConfig(n_feats=int(model_sd["text_uncon"].shape[1]), dim=dim, dim_mults=tuple(mults), beta_min=beta_min,
                         beta_max=beta_max, pe_scale=pe_scale, spk_emb_dim=int(model_sd["spk_uncon"].shape[-1]))


def load_decoder_checkpoint(path: str, map_location="cpu") -> DecoderCheckpoint:
    d = torch.load(path, map_location=map_location, weights_only=True)
    for k in ("model", "mel_min", "mel_max", "spk_emb"):
        if k not in d:
            raise KeyError(f"{path}: not a decoder checkpoint (no '{k}' entry; has {sorted(d)})")
    model = OrderedDict(d["model"])
    cfg = infer_config(model)
    want = param_shapes(cfg)
    if list(model.keys()) != list(want.keys()):
        missing, extra = [k for k in want if k not in model], [k for k in model if k not in want]
        raise KeyError(f"{path}: state_dict keys do not match the decoder architecture (missing {missing[:3]}, unexpected {extra[:3]})")
    for k, shape in want.items():
        if tuple(model[k].shape) != tuple(shape):
            raise ValueError(f"{path}: {k} has shape {tuple(model[k].shape)}, expected {tuple(shape)}")
    it = d.get("iteration")
    return DecoderCheckpoint(model=model, spk_emb=d["spk_emb"], mel_min=torch.as_tensor(d["mel_min"]), mel_max=torch.as_tensor(d["mel_max"]),
                             iteration=int(it) if it is not None else None, raw=dict(d))


def build_decoder(ckpt: DecoderCheckpoint, device=None):
    """`UnitSpeech(...)` + `load_state_dict(ckpt["model"])` (inference.py:55-74)."""
    from .unitspeech import UnitSpeech
    cfg = infer_config(ckpt.model)
    m = UnitSpeech(cfg.n_feats, cfg.dim, list(cfg.dim_mults), cfg.beta_min, cfg.beta_max, cfg.pe_scale, cfg.spk_emb_dim)
    m.load_state_dict(ckpt.model, strict=True)
    return m.to(device) if device is not None else m


def save_finetuned_checkpoint(path: str, decoder, spk_emb: torch.Tensor, mel_min, mel_max, base: Optional[DecoderCheckpoint] = None) -> None:
    """`finetune.py:167-173`: the loaded dict (`base.raw`; empty when fine-tuning did not start from a file) with the four entries
    overwritten in place, then `torch.save`."""
    out: Dict[str, Any] = dict(base.raw) if base is not None else {}
    out["model"] = OrderedDict((k, v.detach().cpu()) for k, v in decoder.state_dict().items())
    out["mel_min"] = torch.as_tensor(mel_min).cpu()
    out["mel_max"] = torch.as_tensor(mel_max).cpu()
    out["spk_emb"] = spk_emb.detach().cpu()
    torch.save(out, path)
