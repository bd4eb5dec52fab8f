#!/usr/bin/env python3
"""Conditioning producer benchmark: text Encoder + DurationPredictor + alignment for one utterance, the step
`execute_text_to_speech` runs once before the 50 decoder evaluations (unitspeech/unitspeech.py:421-438).

    python bench_frontend.py [--symbols 200] [--iters 50]

Reports milliseconds per utterance on the GPU (HIP modules through the C ABI), the same on the host CPU with the oracle
restatement (`cpu_baseline`), and a roofline line: the step is latency-bound (≈60 small launches over a few hundred symbols), so
the HBM figure -- weights read once + activations written and read once per layer -- is a small fraction of the peak by nature.
"""
import argparse
import json
import os
import time

import numpy as np
import torch

from unitspeech_amd.encoder import (DurationPredictor, DurationPredictorConfig, Encoder, EncoderConfig, synthetic_duration_predictor_state_dict,
                                    synthetic_encoder_state_dict)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--symbols", type=int, default=200, help="phoneme ids per utterance (with interspersed blanks)")
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    ec, dc = EncoderConfig(), DurationPredictorConfig()
    esd, dsd = synthetic_encoder_state_dict(ec, 0), synthetic_duration_predictor_state_dict(dc, 0)
    enc = Encoder(ec.n_vocab, ec.n_feats, ec.n_channels, ec.filter_channels, ec.n_heads, ec.n_layers, ec.kernel_size, 0.1, window_size=ec.window_size)
    enc.load_state_dict({k: torch.from_numpy(v) for k, v in esd.items()})
    dp = DurationPredictor(dc.in_channels, dc.filter_channels, dc.kernel_size, 0.1, spk_emb_dim=dc.spk_emb_dim)
    dp.load_state_dict({k: torch.from_numpy(v) for k, v in dsd.items()})
    enc, dp = enc.to(dev).eval(), dp.to(dev).eval()
    g = np.random.Generator(np.random.Philox(key=5))
    B, L = a.batch, a.symbols
    ids = torch.from_numpy(g.integers(0, ec.n_vocab, size=(B, L)).astype(np.int64))
    lens = torch.LongTensor([L] * B)
    spk = torch.from_numpy(g.standard_normal((B, 1, dc.spk_emb_dim), dtype=np.float32))
    ids_d, lens_d, spk_d = ids.to(dev), lens.to(dev), spk.to(dev)

    def step():
        mu_x, x, x_mask = enc(ids_d, lens_d)
        return mu_x, dp(x, x_mask, w=None, g=spk_d, reverse=True)

    for _ in range(a.warmup):
        out = step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.iters):
        out = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.iters
    n_w = sum(v.size for v in esd.values()) + sum(v.size for v in dsd.values())
    rows = B * L
    # activations written + read once per layer output: 6 x (q, k, v, attention, out, ffn hidden 768, ffn out) + prenet + predictor
    act = rows * 4 * 2 * (ec.n_layers * (6 * ec.n_channels + ec.filter_channels) + 8 * ec.n_channels + 3 * dc.filter_channels + 448)
    byt = 4 * n_w + act
    res = {"metric": "conditioning producer ms/utterance (text Encoder + DurationPredictor, eval)", "value": dt * 1e3, "unit": "ms",
           "higher_is_better": False, "n_gpus": 1, "dtype": "f32", "data": "synthetic",
           "config": {"workload": f"Encoder 192/768/6 layers/2 heads/window 4 + DurationPredictor 448->256, B={B}, {L} symbols", "launch": "eager"},
           "iters": a.iters, "finite": bool(torch.isfinite(out[0]).all() and torch.isfinite(out[1]).all()),
           "roofline": {"bound": "hbm", "achieved": byt / dt / 1e9, "peak": 8000.0, "unit": "GB/s", "frac": byt / dt / 8e12, "traffic": None,
                        "bytes_per_utterance": byt, "note": "latency-bound by nature: ~60 launches over a few hundred symbols, once per utterance"}}
    if not a.no_cpu_baseline:
        from oracle import frontend_oracle as FO
        cores = min(len(os.sched_getaffinity(0)), int(os.environ.get("UNITSPEECH_CPU_THREADS", "16")))
        torch.set_num_threads(cores)
        te, td = {k: torch.from_numpy(v) for k, v in esd.items()}, {k: torch.from_numpy(v) for k, v in dsd.items()}

        def cpu_step():
            mu_x, x, x_mask = FO.encoder_forward(te, ids, lens, n_heads=ec.n_heads, n_layers=ec.n_layers, kernel_size=ec.kernel_size,
                                                 window_size=ec.window_size)
            return mu_x, FO.duration_predictor_forward(td, x, x_mask, spk)
        ref = cpu_step()
        t1 = time.perf_counter()
        n = 5
        for _ in range(n):
            ref = cpu_step()
        cdt = (time.perf_counter() - t1) / n
        model_name = next((ln.split(":", 1)[1].strip() for ln in open("/proc/cpuinfo") if ln.lower().startswith("model name")), "unknown")
        res["cpu_baseline"] = {"value": cdt * 1e3, "unit": "ms", "cores": cores, "kind": "port", "cpu": model_name,
                               "sample": f"{n} utterances through oracle/frontend_oracle.py after one warm-up"}
        res["max_abs_diff_vs_oracle"] = max(float((out[0].cpu() - ref[0]).abs().max()), float((out[1].cpu() - ref[1]).abs().max()))
    print(json.dumps(res))


if __name__ == "__main__":
    main()
