#!/usr/bin/env python3
"""Speaker adaptation with the HIP decoder: the reference's `finetune.py` command line (:177-191) and inner loop (:131-165:
`decoder.fine_tune` -> `loss.backward()` -> `clip_grad_norm_(1)` -> `Adam(lr=2e-5).step()`), decoder swapped for
`unitspeech_amd.UnitSpeech`.

  default       the reference's pre-steps (speaker embedder, unit extractor, unit encoder, mel extraction; finetune.py:47-128)
                come from a checkout of the reference given with --reference_root and stay on the stock PyTorch path.
  --synthetic   seeded synthetic decoder weights and a synthetic (mel, units, durations, speaker embedding) tuple: runs the
                fine-tuning loop itself (BASELINE.json configs[3]) without any downloaded model.
  --features F  the OUTPUTS of the reference's pre-steps (finetune.py:86-128) from a `.pt` (torch.save of a dict) or `.npz` file, so the
                speaker embedder / unit extractor can run wherever their checkpoints live and the adaptation here:
                  mel        [1, 80, L]   normalised to [-1, 1] as finetune.py:104 leaves it (or raw with "mel_is_normalized": False)
                  spk_emb    [1, 256] or [1, 1, 256]   (divided by its norm here, :110)
                  duration   [1, Lu]      frames per unit (process_unit, :114)
                  cond_x     [1, 80, Lu]  the unit encoder's output (:123); or  unit [1, Lu] int64 + --unit_encoder_checkpoint (:66-79)
                  mel_min, mel_max        scalars (else the decoder checkpoint's, :98-99)
Saves {"model", "spk_emb", "mel_min", "mel_max"} like finetune.py:167-173.
"""
from __future__ import annotations

import argparse
import os
import sys
import time

import numpy as np
import torch

from unitspeech_amd import DecoderConfig, FusedAdam, UnitSpeech, synthetic_state_dict
from unitspeech_amd.checkpoint import build_decoder, infer_config, load_decoder_checkpoint, save_finetuned_checkpoint
from unitspeech_amd.util import fix_len_compatibility, generate_path, sequence_mask


def load_features(args, cfg, base, device):
    """--features: (mel, cond_x, duration, spk_emb, mel_min, mel_max) on `device` from the file the reference's pre-steps were saved to."""
    path = args.features
    if path.endswith(".npz"):
        with np.load(path) as f:
            d = {k: torch.from_numpy(np.asarray(f[k])) for k in f.files}
    else:
        d = torch.load(path, map_location="cpu")
    d = {k: (torch.as_tensor(v) if not isinstance(v, torch.Tensor) else v) for k, v in d.items()}
    for k in ("mel", "spk_emb", "duration"):
        if k not in d:
            raise SystemExit(f"--features {path}: missing `{k}`")

    def scalar(name):
        if name in d:
            return d[name].float().reshape(())
        if base is not None and getattr(base, name, None) is not None:
            return getattr(base, name).float().reshape(())
        raise SystemExit(f"--features {path}: no `{name}` in the file and no decoder checkpoint to take it from (finetune.py:98-99)")
    mel_min, mel_max = scalar("mel_min"), scalar("mel_max")
    mel = d["mel"].float()
    if mel.dim() == 2:
        mel = mel.unsqueeze(0)
    if mel.dim() != 3 or mel.shape[0] != 1 or mel.shape[1] != cfg.n_feats:
        raise SystemExit(f"--features: mel must be [1, {cfg.n_feats}, L], got {tuple(mel.shape)}")
    if "mel_is_normalized" in d and not bool(d["mel_is_normalized"]):
        mel = (mel - mel_min) / (mel_max - mel_min) * 2 - 1                      # finetune.py:104
    spk = d["spk_emb"].float().reshape(1, 1, -1)
    if spk.shape[-1] != cfg.spk_emb_dim:
        raise SystemExit(f"--features: spk_emb must have {cfg.spk_emb_dim} elements, got {spk.shape[-1]}")
    spk = spk / spk.norm()                                                       # :110
    duration = d["duration"].float().reshape(1, -1)
    if "cond_x" in d:
        cond_x = d["cond_x"].float()
        if cond_x.dim() == 2:
            cond_x = cond_x.unsqueeze(0)
    elif "unit" in d:
        if not args.unit_encoder_checkpoint:
            raise SystemExit("--features with `unit` needs --unit_encoder_checkpoint (or store the unit encoder's output as `cond_x`)")
        from unitspeech_amd.encoder import Encoder, EncoderConfig
        sd = torch.load(args.unit_encoder_checkpoint, map_location="cpu")
        sd = sd["model"] if "model" in sd else sd
        ec = EncoderConfig(n_vocab=int(sd["emb.weight"].shape[0]), n_feats=cfg.n_feats)
        unit_encoder = Encoder(ec.n_vocab, ec.n_feats, ec.n_channels, ec.filter_channels, ec.n_heads, ec.n_layers, ec.kernel_size, 0.1,
                               window_size=ec.window_size)
        unit_encoder.load_state_dict(sd)
        unit = d["unit"].long().reshape(1, -1).to(device)
        with torch.no_grad():
            cond_x, _, _ = unit_encoder.to(device).eval()(unit, torch.LongTensor([unit.shape[-1]]).to(device))        # :122-123
    else:
        raise SystemExit(f"--features {path}: give `cond_x` (the unit encoder's output) or `unit`")
    if cond_x.shape[0] != 1 or cond_x.shape[1] != cfg.n_feats or cond_x.shape[-1] != duration.shape[-1]:
        raise SystemExit(f"--features: cond_x {tuple(cond_x.shape)} and duration {tuple(duration.shape)} disagree")
    if int(duration.sum()) > mel.shape[-1] + duration.shape[-1]:
        raise SystemExit(f"--features: the durations cover {int(duration.sum())} frames, the mel has {mel.shape[-1]}")
    return mel.to(device), cond_x.float().to(device), duration.to(device), spk.to(device), mel_min, mel_max


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference_sample", type=str, default="reference.wav", help="Sample used to adapt the model to the speaker.")
    ap.add_argument("--ID", type=int, default=-1, help="Unique value used to identify the finetuned decoder.")
    ap.add_argument("--n_iters", type=int, default=500, help="Number of fine-tuning iterations.")
    ap.add_argument("--learning_rate", type=float, default=2e-5, help="Learning rate of the optimizer during fine-tuning.")
    ap.add_argument("--torch_optimizer", action="store_true", help="clip_grad_norm_ + torch.optim.Adam instead of the HIP clip+Adam")
    ap.add_argument("--no_graph", action="store_true", help="launch every kernel of an iteration eagerly instead of replaying the captured HIP graph of the forward (unitspeech_amd.graph)")
    ap.add_argument("--synthetic", action="store_true")
    ap.add_argument("--features", type=str, default=None, help="file with the pre-step tensors of finetune.py:86-128 (see the module docstring)")
    ap.add_argument("--unit_encoder_checkpoint", type=str, default=None, help="--features with `unit` instead of `cond_x`: the unit encoder's "
                                                                              "checkpoint ({'model': state_dict}, finetune.py:77-78)")
    ap.add_argument("--learned_frontend", action="store_true", help="--synthetic: cond_x from the HIP unit encoder (seeded weights) on synthetic units")
    ap.add_argument("--reference_root", type=str, default=None)
    ap.add_argument("--out_dir", type=str, default="checkpoints/inference")
    ap.add_argument("--decoder_checkpoint", type=str, default=None,
                    help="pre-trained decoder checkpoint to start from (train_STEP1.py:297-304 layout); --synthetic uses seeded weights otherwise")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--report_memory", action="store_true", help="print the allocated device memory with every progress line and the f16x3 range status at the end")
    args = ap.parse_args()
    if not torch.cuda.is_available():
        raise SystemExit("CUDA/ROCm is not available: the HIP decoder has no CPU fallback")
    device = torch.device("cuda", 0)
    torch.manual_seed(args.seed)
    import random
    random.seed(args.seed)
    cfg = DecoderConfig()
    n_down = len(cfg.dim_mults) - 1
    segment = fix_len_compatibility(2 * 22050 // 256, n_down)                 # out_size, finetune.py:40-44 (= 176)
    base = None
    if args.decoder_checkpoint:
        base = load_decoder_checkpoint(args.decoder_checkpoint)                  # finetune.py:61-63
        decoder = build_decoder(base)
        cfg = infer_config(base.model)
    else:
        decoder = UnitSpeech(cfg.n_feats, cfg.dim, list(cfg.dim_mults), cfg.beta_min, cfg.beta_max, cfg.pe_scale, cfg.spk_emb_dim)

    if args.features:
        if base is None and args.synthetic:
            decoder.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic_state_dict(cfg, 0).items()})
        elif base is None:
            raise SystemExit("--features needs --decoder_checkpoint (or --synthetic for seeded decoder weights)")
        mel, cond_x, duration, spk_emb, mel_min, mel_max = load_features(args, cfg, base, device)
    elif args.synthetic:
        if base is None:
            decoder.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic_state_dict(cfg, 0).items()})
        g = np.random.Generator(np.random.Philox(key=args.ID & 0xffff))
        L = 600
        Lu = L // 3
        mel = torch.from_numpy(g.standard_normal((1, cfg.n_feats, L), dtype=np.float32)).clamp(-1, 1).to(device)
        cond_x = (torch.from_numpy(g.standard_normal((1, cfg.n_feats, Lu), dtype=np.float32)) * 0.5).to(device)
        if args.learned_frontend:
            # finetune.py:66-78,122-123: cond_x is the (frozen, eval-mode) unit encoder's output for the utterance's unit sequence;
            # here the HIP Encoder at the reference's sizes (n_vocab = n_units = 1000) with seeded weights on synthetic units
            from unitspeech_amd.encoder import Encoder, EncoderConfig, synthetic_encoder_state_dict
            ec = EncoderConfig(n_vocab=1000, n_feats=cfg.n_feats)
            unit_encoder = Encoder(ec.n_vocab, ec.n_feats, ec.n_channels, ec.filter_channels, ec.n_heads, ec.n_layers, ec.kernel_size, 0.1,
                                   window_size=ec.window_size)
            unit_encoder.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic_encoder_state_dict(ec, 0).items()})
            unit = torch.from_numpy(g.integers(0, ec.n_vocab, size=(1, Lu)).astype(np.int64)).to(device)
            cond_x, _, _ = unit_encoder.to(device).eval()(unit, torch.LongTensor([Lu]).to(device))
        duration = torch.full((1, Lu), 3.0, device=device)
        spk = torch.from_numpy(g.standard_normal((1, 1, cfg.spk_emb_dim), dtype=np.float32)).to(device)
        spk_emb = spk / spk.norm()
        mel_min, mel_max = torch.tensor(-11.5), torch.tensor(2.0)
    else:
        if not args.reference_root:
            raise SystemExit("give --features (the pre-step tensors), --reference_root (reference checkout with its checkpoints) or use --synthetic")
        raise SystemExit("running the pre-steps here needs the reference's WavLM/ECAPA speaker embedder, mHuBERT unit extractor and unit "
                         "encoder checkpoints (finetune.py:47-128), none of which are available offline; run the pre-steps with the "
                         "reference, save their tensors and pass the file with --features")
    decoder = decoder.to(device).train()
    # finetune.py:81 uses torch.optim.Adam; FusedAdam is the same update (clip + Adam) in three HIP launches
    opt = (torch.optim.Adam if args.torch_optimizer else FusedAdam)(decoder.parameters(), lr=args.learning_rate)
    mel_lengths = torch.LongTensor([mel.shape[-1]]).to(device)
    mel_mask = sequence_mask(mel_lengths, mel.shape[-1]).unsqueeze(1).to(mel.dtype)
    x_mask = torch.ones(1, 1, cond_x.shape[-1], device=device)
    attn = generate_path(duration, (x_mask.unsqueeze(-1) * mel_mask.unsqueeze(2)).squeeze(1))

    def probe():
        """--report_memory: the diffusion loss of the utterance's first segment at 8 FIXED (t, z) draws, outside the training run's random stream:
        the same probe before and after the adaptation says whether it made progress (single iterations' losses vary 5x with their t)."""
        cpu_state, dev_state = torch.get_rng_state(), torch.cuda.get_rng_state(device)
        was_training = decoder.training
        decoder.eval()
        seg_mask = torch.ones(1, 1, segment, device=device)
        a0 = attn[:, :, :segment]
        cond_y = torch.matmul(a0.transpose(1, 2), cond_x.transpose(1, 2)).transpose(1, 2)          # unitspeech.py:484-486 for a crop at offset 0
        tot = 0.0
        with torch.no_grad():
            for k in range(8):
                torch.manual_seed(4321 + k)
                loss, _ = decoder.compute_loss(mel[:, :, :segment], seg_mask, cond_y, spk_emb=spk_emb)
                tot += float(loss)
        decoder.train(was_training)
        torch.set_rng_state(cpu_state)
        torch.cuda.set_rng_state(dev_state, device)
        return tot / 8

    if args.report_memory:
        print(f"probe loss before {probe():.5f}")
    graph = None
    if not args.no_graph:
        from unitspeech_amd.graph import FineTuneGraph
        graph = FineTuneGraph(decoder, spk_emb, mel.shape[0], segment, cfg.n_feats)    # forward + backward of an iteration as one HIP graph
    t0 = time.perf_counter()
    for it in range(args.n_iters):                                                           # finetune.py:131-165
        if graph is not None:
            loss = graph.step(cond_x, mel, mel_lengths, attn)
        else:
            loss = decoder.fine_tune(cond_x, mel, mel_mask, mel_lengths, mel.shape[-1], attn, spk_emb, segment, cfg.n_feats)
            opt.zero_grad(set_to_none=True)
            loss.backward()
        if args.torch_optimizer:
            torch.nn.utils.clip_grad_norm_(decoder.parameters(), 1)
            opt.step()
        else:
            opt.step(max_norm=1)
        if it % 50 == 0 or it == args.n_iters - 1:
            print(f"iter {it:4d}  diffusion loss {loss.item():.5f}")
            if args.report_memory:
                print(f"          allocated {torch.cuda.memory_allocated(device) >> 20} MiB")
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{args.n_iters} iterations in {dt:.2f} s ({1e3 * dt / max(args.n_iters, 1):.1f} ms/iter)")
    if args.report_memory:
        print(f"range status {decoder.range_status()}")
        print(f"probe loss after {probe():.5f}")
    os.makedirs(args.out_dir, exist_ok=True)
    path = os.path.join(args.out_dir, f"{args.ID}.pt")
    save_finetuned_checkpoint(path, decoder, spk_emb, mel_min, mel_max, base=base)           # finetune.py:167-173
    print(f"saved {path}")


if __name__ == "__main__":
    main()
