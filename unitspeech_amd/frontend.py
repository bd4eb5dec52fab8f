"""Deterministic stand-ins for the conditioning producer's two learned modules.

The reference's `execute_text_to_speech` (`unitspeech/unitspeech.py:413-450`) takes the text encoder and the
duration predictor as callables: `text_encoder(phoneme, lengths) -> (cond_x, x, x_mask)` (`unitspeech/encoder.py:294`)
and `duration_predictor(x, x_mask, w=None, g=spk_emb, reverse=True) -> logw` (`unitspeech/duration_predictor.py:47`).
Their checkpoints are not available offline, so the `--synthetic` CLI mode and the `tts_*` goldens
(`tools/make_goldens_r2.py`, which hands the same two callables to the REFERENCE's `execute_text_to_speech`) use these
seeded stand-ins: an embedding table from NumPy's Philox stream (independent of torch's RNG) and a closed-form duration.
Device-agnostic torch ops only; no learned state.
"""
from __future__ import annotations

import numpy as np
import torch


class SyntheticFrontEnd:
    def __init__(self, n_feats: int, device="cpu", seed: int = 1234, vocab: int = 512):
        g = np.random.Generator(np.random.Philox(key=seed))
        table = (g.standard_normal((vocab, n_feats), dtype=np.float32) * np.float32(0.5)).astype(np.float32)
        self.table = torch.from_numpy(table).to(device)
        self.device = device

    def text_encoder(self, phoneme, phoneme_lengths):
        x = self.table[phoneme % self.table.shape[0]].transpose(1, 2)            # [B, n_feats, L]
        ar = torch.arange(phoneme.shape[1], device=phoneme.device)
        x_mask = (ar.unsqueeze(0) < phoneme_lengths.unsqueeze(1)).unsqueeze(1).to(x.dtype)
        return x * x_mask, x, x_mask

    def duration_predictor(self, x, x_mask, w=None, g=None, reverse=True):
        # 3..8 frames per symbol from the symbol id pattern (exact small integers + 0.5, so exp(log(.)) followed by the
        # reference's ceil (:425) cannot flip between CPU and GPU libm)
        frames = 2.5 + torch.floor(6.0 * torch.sigmoid(4.0 * x[:, :1]))
        return torch.log(frames) * x_mask


def text_to_ids(text: str, device="cpu"):
    """Characters -> ids with an interspersed blank, as `intersperse` does for phonemes (`unitspeech/util.py:62`)."""
    ids = [0]
    for ch in text.strip().lower():
        ids += [1 + (ord(ch) % 200), 0]
    t = torch.LongTensor(ids).unsqueeze(0).to(device)
    return t, torch.LongTensor([t.shape[-1]]).to(device)
