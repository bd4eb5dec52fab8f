#!/usr/bin/env python3
"""Text -> mel inference with the HIP decoder: the reference's `inference.py` command line (same flags, :164-200) with the
decoder swapped for `unitspeech_amd.UnitSpeech`.

Two modes:
  default       the reference's call sequence (inference.py:34-160): phonemiser, text encoder, duration predictor, BigVGAN
                and the checkpoints come from a checkout of the reference given with --reference_root (phonemiser and vocoder
                stay on the stock PyTorch path; text encoder, duration predictor, alignment and decoder run on the HIP library).
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
from unitspeech_amd.checkpoint import build_decoder, load_decoder_checkpoint
from unitspeech_amd.encoder import (DurationPredictor, DurationPredictorConfig, Encoder, EncoderConfig, synthetic_duration_predictor_state_dict,
                                    synthetic_encoder_state_dict)
from unitspeech_amd.frontend import SyntheticFrontEnd, text_to_ids


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
    ap.add_argument("--learned_frontend", action="store_true",
                    help="--synthetic: run the HIP text encoder + duration predictor with seeded weights instead of the closed-form stand-ins")
    ap.add_argument("--reference_root", type=str, default=None, help="checkout of adrianstanea/UnitSpeech (non-synthetic mode)")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--noise_key", type=int, default=None,
                    help="draw z and the per-step noise from NumPy Philox(key) in the reference's order (z, then one tensor per step) instead of "
                         "torch's device generator: the stream the committed goldens were drawn from (tests/golden/tts_*.npz: 4242)")
    ap.add_argument("--spk_seed", type=int, default=None,
                    help="--synthetic: the speaker embedding of unitspeech_amd.synthetic_inputs(seed) instead of the --ID-keyed one")
    args = ap.parse_args()

    if not torch.cuda.is_available():
        raise SystemExit("CUDA/ROCm is not available: the HIP decoder has no CPU fallback (reference: inference.py:38-39)")
    device = torch.device("cuda", 0)
    torch.manual_seed(args.seed)
    cfg = DecoderConfig()
    n_down = len(cfg.dim_mults) - 1

    if args.synthetic:
        decoder = UnitSpeech(cfg.n_feats, cfg.dim, list(cfg.dim_mults), cfg.beta_min, cfg.beta_max, cfg.pe_scale, cfg.spk_emb_dim)
        decoder.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic_state_dict(cfg, 0).items()})
        decoder = decoder.to(device).eval()
        if args.learned_frontend:
            # the HIP text encoder + duration predictor at the reference's sizes (conf/hydra_config.py:85-116) with seeded weights
            ec, dc = EncoderConfig(n_vocab=256), DurationPredictorConfig()
            text_encoder = Encoder(ec.n_vocab, ec.n_feats, ec.n_channels, ec.filter_channels, ec.n_heads, ec.n_layers, ec.kernel_size, 0.1,
                                   window_size=ec.window_size)
            text_encoder.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic_encoder_state_dict(ec, 0).items()})
            duration_predictor = DurationPredictor(dc.in_channels, dc.filter_channels, dc.kernel_size, 0.1, spk_emb_dim=dc.spk_emb_dim)
            duration_predictor.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic_duration_predictor_state_dict(dc, 0).items()})
            text_encoder, duration_predictor = text_encoder.to(device).eval(), duration_predictor.to(device).eval()
        else:
            fe = SyntheticFrontEnd(cfg.n_feats, device)
            text_encoder, duration_predictor = fe.text_encoder, fe.duration_predictor
        spk = np.random.Generator(np.random.Philox(key=args.ID & 0xffff)).standard_normal((1, 1, cfg.spk_emb_dim), dtype=np.float32)
        spk_emb = torch.from_numpy(spk / np.linalg.norm(spk)).to(device)
        if args.spk_seed is not None:
            from unitspeech_amd import synthetic_inputs
            spk_emb = torch.from_numpy(synthetic_inputs(cfg, 1, 8, seed=args.spk_seed)["spk_emb"]).to(device)
        mel_min, mel_max = torch.tensor(-11.5, device=device), torch.tensor(2.0, device=device)
        phoneme, phoneme_lengths = text_to_ids(args.text, device)
        vocoder = None
    else:
        if not args.reference_root:
            raise SystemExit("give --reference_root (reference checkout with its checkpoints) or use --synthetic")
        sys.path.insert(0, args.reference_root)
        from conf.hydra_config import MainConfig as rcfg                                   # noqa: E402
        from unitspeech.text import cleaned_text_to_sequence, phonemize, symbols           # noqa: E402
        from unitspeech.util import get_phonemizer, get_vocoder, intersperse                # noqa: E402
        root = args.reference_root
        vocoder = get_vocoder(config_path=os.path.join(root, rcfg.vocoder.config_path), checkpoint=os.path.join(root, rcfg.vocoder.ckpt_path),
                              device=device)
        ck = os.path.join(root, rcfg.decoder.checkpoint if args.ID < 0 else f"{rcfg.finetune.finetuned_decoders_path}/{args.ID}.pt")
        dd = load_decoder_checkpoint(ck)                                                   # inference.py:66-74,107-108,124
        decoder = build_decoder(dd, device).eval()
        mel_max, mel_min, spk_emb = dd.mel_max.to(device), dd.mel_min.to(device), dd.speaker_embedding(max(args.ID, 0)).to(device)
        # text encoder and duration predictor: the HIP modules (same constructor arguments and state_dict keys as the reference's
        # classes, inference.py:77-105), loaded from the reference's own checkpoints
        e = rcfg.text_encoder
        text_encoder = Encoder(n_vocab=len(symbols) + 1, n_feats=cfg.n_feats, n_channels=e.n_channels, filter_channels=e.filter_channels,
                               n_heads=e.n_heads, n_layers=e.n_layers, kernel_size=e.kernel_size, p_dropout=e.p_dropout,
                               window_size=e.window_size).to(device)
        text_encoder.load_state_dict(torch.load(os.path.join(root, e.checkpoint), map_location="cpu")["model"])
        d = rcfg.duration_predictor
        duration_predictor = DurationPredictor(in_channels=d.in_channels, filter_channels=d.filter_channels, kernel_size=d.kernel_size,
                                               p_dropout=d.p_dropout, spk_emb_dim=d.spk_emb_dim).to(device)
        duration_predictor.load_state_dict(torch.load(os.path.join(root, d.checkpoint), map_location="cpu")["model"])
        text_encoder.eval(); duration_predictor.eval()
        ph = phonemize(args.text, get_phonemizer(rcfg.inference.language))
        seq = intersperse(cleaned_text_to_sequence(ph), len(symbols))
        phoneme = torch.LongTensor(seq).unsqueeze(0).to(device)
        phoneme_lengths = torch.LongTensor([phoneme.shape[-1]]).to(device)

    restore = None
    if args.noise_key is not None:
        # the reference draws z with randn_like (:441) and then one randn per step (:367); serve both from one NumPy Philox stream
        gen = np.random.Generator(np.random.Philox(key=args.noise_key))
        restore = (torch.randn_like, torch.randn)

        def draw(shape):
            return torch.from_numpy(gen.standard_normal(tuple(int(v) for v in shape), dtype=np.float32)).to(device)
        torch.randn_like = lambda x, **k: draw(x.shape)
        torch.randn = lambda *shape, **k: draw(shape[0] if len(shape) == 1 and not isinstance(shape[0], int) else shape)
    t0 = time.perf_counter()
    with torch.no_grad():
        # the de-normalisation of inference.py:140 happens in the sampler's last pass: `mel` is the vocoder's input as it stands
        y_enc, mel, attn = decoder.execute_text_to_speech(
            phoneme=phoneme, phoneme_lengths=phoneme_lengths, spk_emb=spk_emb, text_encoder=text_encoder,
            duration_predictor=duration_predictor, num_downsamplings_in_unet=n_down, diffusion_steps=args.diffusion_steps,
            length_scale=args.length_scale, text_gradient_scale=args.text_gradient_scale, spk_gradient_scale=args.spk_gradient_scale,
            mel_range=(mel_min, mel_max))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if restore is not None:
        torch.randn_like, torch.randn = restore
    frames = mel.shape[-1]
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
