#!/usr/bin/env python3
"""Text -> mel inference with the HIP decoder: the reference's `inference.py` command line (same flags, :164-200) with the
decoder swapped for `unitspeech_amd.UnitSpeech`.

Two modes:
  default       the reference's call sequence (inference.py:34-160): phonemiser, text encoder, duration predictor, BigVGAN
                and the checkpoints come from a checkout of the reference given with --reference_root (they stay on the
                stock PyTorch path; only `decoder.execute_text_to_speech` runs on the HIP library).
  --synthetic   no checkpoints / espeak / vocoder are needed: seeded synthetic decoder weights, a deterministic stand-in for
                the text encoder + duration predictor (same call signatures), output = de-normalised mel saved as .npy.
                This is the plumbing check of BASELINE.json configs[0] (10 diffusion steps, short text).
"""
from __future__ import annotations

import argparse
import os
import sys
import time

import numpy as np
import torch

from unitspeech_amd import DecoderConfig, UnitSpeech, synthetic_state_dict


class SyntheticFrontEnd:
    """Stand-ins with the reference's signatures: `text_encoder(phoneme, lengths) -> (cond_x, x, x_mask)`
    (unitspeech/encoder.py:294) and `duration_predictor(x, x_mask, w=None, g=spk_emb, reverse=True) -> logw`
    (unitspeech/duration_predictor.py:47)."""

    def __init__(self, n_feats: int, device):
        g = torch.Generator().manual_seed(1234)
        self.table = torch.randn(512, n_feats, generator=g).to(device) * 0.5
        self.device = device

    def text_encoder(self, phoneme, phoneme_lengths):
        x = self.table[phoneme % self.table.shape[0]].transpose(1, 2)            # [B, n_feats, L]
        ar = torch.arange(phoneme.shape[1], device=self.device)
        x_mask = (ar.unsqueeze(0) < phoneme_lengths.unsqueeze(1)).unsqueeze(1).float()
        return x * x_mask, x, x_mask

    def duration_predictor(self, x, x_mask, w=None, g=None, reverse=True):
        # 3..8 frames per symbol, deterministic in the symbol embedding
        frames = 3.0 + 5.0 * torch.sigmoid(x.mean(1, keepdim=True))
        return torch.log(frames) * x_mask


def text_to_ids(text: str, device):
    ids = [0]
    for ch in text.strip().lower():
        ids += [1 + (ord(ch) % 200), 0]                    # interspersed blank, as `intersperse` does (unitspeech/util.py:62)
    t = torch.LongTensor(ids).unsqueeze(0).to(device)
    return t, torch.LongTensor([t.shape[-1]]).to(device)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--generated_sample_path", type=str, default="audio.wav", help="The path to save the generated audio.")
    ap.add_argument("--text", type=str, required=True, help="The desired transcript to be generated.")
    ap.add_argument("--ID", type=int, default=-10, help="The speaker ID to be used for the generation.")
    ap.add_argument("--text_gradient_scale", type=float, default=1.0)
    ap.add_argument("--spk_gradient_scale", type=float, default=1.0)
    ap.add_argument("--length_scale", type=float, default=1.0)
    ap.add_argument("--diffusion_steps", type=int, default=50)
    ap.add_argument("--synthetic", action="store_true", help="synthetic weights and front-end stand-ins (no checkpoints needed)")
    ap.add_argument("--reference_root", type=str, default=None, help="checkout of adrianstanea/UnitSpeech (non-synthetic mode)")
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()

    if not torch.cuda.is_available():
        raise SystemExit("CUDA/ROCm is not available: the HIP decoder has no CPU fallback (reference: inference.py:38-39)")
    device = torch.device("cuda", 0)
    torch.manual_seed(args.seed)
    cfg = DecoderConfig()
    n_down = len(cfg.dim_mults) - 1
    decoder = UnitSpeech(cfg.n_feats, cfg.dim, list(cfg.dim_mults), cfg.beta_min, cfg.beta_max, cfg.pe_scale, cfg.spk_emb_dim)

    if args.synthetic:
        decoder.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic_state_dict(cfg, 0).items()})
        decoder = decoder.to(device).eval()
        fe = SyntheticFrontEnd(cfg.n_feats, device)
        text_encoder, duration_predictor = fe.text_encoder, fe.duration_predictor
        spk = torch.randn(1, 1, cfg.spk_emb_dim, generator=torch.Generator().manual_seed(args.ID & 0xffff)).to(device)
        spk_emb = spk / spk.norm()
        mel_min, mel_max = torch.tensor(-11.5, device=device), torch.tensor(2.0, device=device)
        phoneme, phoneme_lengths = text_to_ids(args.text, device)
        vocoder = None
    else:
        if not args.reference_root:
            raise SystemExit("give --reference_root (reference checkout with its checkpoints) or use --synthetic")
        sys.path.insert(0, args.reference_root)
        from conf.hydra_config import MainConfig as rcfg                                   # noqa: E402
        from unitspeech.duration_predictor import DurationPredictor                         # noqa: E402
        from unitspeech.encoder import Encoder                                               # noqa: E402
        from unitspeech.text import cleaned_text_to_sequence, phonemize, symbols           # noqa: E402
        from unitspeech.util import get_phonemizer, get_vocoder, intersperse                # noqa: E402
        root = args.reference_root
        vocoder = get_vocoder(config_path=os.path.join(root, rcfg.vocoder.config_path), checkpoint=os.path.join(root, rcfg.vocoder.ckpt_path),
                              device=device)
        ck = os.path.join(root, rcfg.decoder.checkpoint if args.ID < 0 else f"{rcfg.finetune.finetuned_decoders_path}/{args.ID}.pt")
        dd = torch.load(ck, map_location="cpu")                                            # inference.py:66-74
        decoder.load_state_dict(dd["model"])
        decoder = decoder.to(device).eval()
        mel_max, mel_min, spk_emb = dd["mel_max"].to(device), dd["mel_min"].to(device), dd["spk_emb"].to(device)
        e = rcfg.encoder
        text_encoder = Encoder(n_vocab=len(symbols) + 1, n_feats=cfg.n_feats, n_channels=e.n_channels, filter_channels=e.filter_channels,
                               n_heads=e.n_heads, n_layers=e.n_layers, kernel_size=e.kernel_size, p_dropout=e.p_dropout,
                               window_size=e.window_size).to(device)
        text_encoder.load_state_dict(torch.load(os.path.join(root, rcfg.text_encoder.checkpoint), map_location="cpu")["model"])
        d = rcfg.duration_predictor
        duration_predictor = DurationPredictor(in_channels=d.in_channels, filter_channels=d.filter_channels, kernel_size=d.kernel_size,
                                               p_dropout=d.p_dropout, spk_emb_dim=d.spk_emb_dim).to(device)
        duration_predictor.load_state_dict(torch.load(os.path.join(root, d.checkpoint), map_location="cpu")["model"])
        text_encoder.eval(); duration_predictor.eval()
        ph = phonemize(args.text, get_phonemizer(rcfg.inference.language))
        seq = intersperse(cleaned_text_to_sequence(ph), len(symbols))
        phoneme = torch.LongTensor(seq).unsqueeze(0).to(device)
        phoneme_lengths = torch.LongTensor([phoneme.shape[-1]]).to(device)

    t0 = time.perf_counter()
    with torch.no_grad():
        y_enc, y_dec, attn = decoder.execute_text_to_speech(
            phoneme=phoneme, phoneme_lengths=phoneme_lengths, spk_emb=spk_emb, text_encoder=text_encoder,
            duration_predictor=duration_predictor, num_downsamplings_in_unet=n_down, diffusion_steps=args.diffusion_steps,
            length_scale=args.length_scale, text_gradient_scale=args.text_gradient_scale, spk_gradient_scale=args.spk_gradient_scale)
        mel = (y_dec + 1) / 2 * (mel_max - mel_min) + mel_min                              # inference.py:140
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    frames = y_dec.shape[-1]
    print(f"decoded {frames} mel frames in {dt:.3f} s ({frames / dt:.1f} frames/s, RTF {dt / (frames * 256 / 22050):.3f}), "
          f"{args.diffusion_steps} diffusion steps, finite={bool(torch.isfinite(mel).all())}")
    if vocoder is None:
        out = os.path.splitext(args.generated_sample_path)[0] + ".mel.npy"
        np.save(out, mel.squeeze(0).cpu().numpy())
        print(f"saved mel-spectrogram to {out} (no vocoder in --synthetic mode)")
    else:
        from scipy.io.wavfile import write
        audio = vocoder.forward(mel).cpu().squeeze().clamp(-1, 1).numpy()
        write(args.generated_sample_path, 22050, audio)
        print(f"saved {args.generated_sample_path}")


if __name__ == "__main__":
    main()
