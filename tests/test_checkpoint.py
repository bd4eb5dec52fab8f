"""Checkpoint layouts of the reference against files written with `torch.save` from the REFERENCE module
(tools/make_goldens_r2.py `ckpt`): the trainer's layout (train_STEP1.py:297-304, with `iteration` and the speaker table) and the
fine-tuned one (finetune.py:167-173)."""
import os

import numpy as np
import torch

from unitspeech_amd import DecoderConfig, synthetic_state_dict
from unitspeech_amd.checkpoint import (build_decoder, infer_config, load_decoder_checkpoint, save_finetuned_checkpoint)
from unitspeech_amd.params import param_shapes
from unitspeech_amd.sharding import pack_state_dict, unpack_state_dict

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CK = DecoderConfig(dim=8, dim_mults=(1, 2, 4))


def test_reads_the_trainer_layout():
    ck = load_decoder_checkpoint(os.path.join(GOLDEN, "ckpt_pretrained_small.pt"))
    assert ck.iteration == 1234
    assert list(ck.raw) == ["model", "spk_emb", "mel_min", "mel_max", "iteration"]
    assert isinstance(ck.spk_emb, dict) and ck.spk_emb["weight"].shape == (4, CK.spk_emb_dim)
    assert ck.speaker_embedding(2).shape == (1, 1, CK.spk_emb_dim) and float(ck.speaker_embedding(2)[0, 0, 0]) == 2 * 256 / 1024
    assert infer_config(ck.model) == CK
    assert float(ck.mel_min) == -11.5 and float(ck.mel_max) == 2.0
    want = synthetic_state_dict(CK, 3)
    assert list(ck.model) == list(want)
    for k, v in want.items():
        np.testing.assert_array_equal(ck.model[k].numpy(), v)


def test_reads_the_finetuned_layout_and_builds_the_decoder():
    ck = load_decoder_checkpoint(os.path.join(GOLDEN, "ckpt_finetuned_small.pt"))
    assert ck.iteration == 1234                      # finetune.py mutates the loaded dict: `iteration` survives (:167-173)
    assert ck.spk_emb.shape == (1, 1, CK.spk_emb_dim) and torch.equal(ck.speaker_embedding(), ck.spk_emb)
    dec = build_decoder(ck)                          # construction + load_state_dict run on CPU; only compute needs the GPU
    sd = dec.state_dict()
    assert list(sd) == list(param_shapes(CK))
    for k in sd:
        assert torch.equal(sd[k], ck.model[k])


def test_written_checkpoint_round_trips_in_reference_key_order(tmp_path):
    base = load_decoder_checkpoint(os.path.join(GOLDEN, "ckpt_pretrained_small.pt"))
    dec = build_decoder(base)
    with torch.no_grad():
        dec.text_uncon.add_(1.0)                     # "fine-tune"
    spk = torch.ones(1, 1, CK.spk_emb_dim) / 16
    path = str(tmp_path / "7.pt")
    save_finetuned_checkpoint(path, dec, spk, base.mel_min, base.mel_max, base=base)
    raw = torch.load(path, weights_only=True)
    assert list(raw) == ["model", "spk_emb", "mel_min", "mel_max", "iteration"] and raw["iteration"] == 1234
    assert list(raw["model"]) == list(param_shapes(CK))          # what the reference's load_state_dict(strict) expects
    back = load_decoder_checkpoint(path)
    assert torch.equal(back.spk_emb, spk)
    assert torch.equal(back.model["text_uncon"], base.model["text_uncon"] + 1.0)
    # without a base file the four entries of finetune.py:169-172 are written
    save_finetuned_checkpoint(path, dec, spk, -11.5, 2.0)
    assert set(torch.load(path, weights_only=True)) == {"model", "spk_emb", "mel_min", "mel_max"}


def test_packed_blob_is_the_identity_on_reference_checkpoints():
    for name in ("ckpt_pretrained_small.pt", "ckpt_finetuned_small.pt"):
        ck = load_decoder_checkpoint(os.path.join(GOLDEN, name))
        flat = pack_state_dict(CK, ck.model, "cpu")
        assert flat.numel() == sum(v.numel() for v in ck.model.values())
        back = unpack_state_dict(CK, flat)
        assert list(back) == list(ck.model)
        for k in back:
            assert torch.equal(back[k], ck.model[k])
