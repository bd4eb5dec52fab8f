#!/usr/bin/env python3
"""Fine-tune step benchmark (BASELINE.json configs[3]: 500-iteration speaker adaptation, B=1, 176-frame crops, Adam 2e-5).

Runs the reference's inner loop of finetune.py:131-165 (decoder.fine_tune -> loss.backward -> clip_grad_norm_(1) ->
Adam.step) on synthetic data with the HIP decoder and reports seconds per iteration; optionally checks the loss
trajectory of the first iterations against the CPU oracle driven by torch autograd with the same draws.
    python bench_finetune.py [--iters 50] [--check 3]
"""
import argparse
import json
import random
import time

import numpy as np
import torch

from unitspeech_amd import DecoderConfig, FusedAdam, UnitSpeech, synthetic_state_dict
from unitspeech_amd.util import generate_path, sequence_mask


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames", type=int, default=600, help="length of the reference utterance's mel")
    ap.add_argument("--segment", type=int, default=176, help="out_size: fix_len_compatibility(2*22050//256)")
    ap.add_argument("--check", type=int, default=0, help="compare the first K losses with the CPU oracle (slow)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-iters", type=int, default=2, help="iterations of the CPU oracle (torch autograd + Adam) timed")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of the captured HIP graph (unitspeech_amd.graph)")
    ap.add_argument("--backward", choices=["graph", "eager"], default=None,
                    help="with the graph: the backward captured too, or run eagerly after the replayed forward (FineTuneGraph; default: its own)")
    ap.add_argument("--optimizer", choices=["fused", "torch"], default="fused",
                    help="fused: HIP clip+Adam in 3 launches (unitspeech_amd.FusedAdam); torch: clip_grad_norm_ + torch.optim.Adam")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    cfg = DecoderConfig()
    sd_np = synthetic_state_dict(cfg, 0)
    model = UnitSpeech(cfg.n_feats, cfg.dim, list(cfg.dim_mults), cfg.beta_min, cfg.beta_max, cfg.pe_scale, cfg.spk_emb_dim)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd_np.items()})
    model = model.to(dev).train()
    fused = a.optimizer == "fused"
    opt = (FusedAdam if fused else torch.optim.Adam)(model.parameters(), lr=2e-5)

    g = np.random.Generator(np.random.Philox(key=2024))
    L, Lu = a.frames, a.frames // 3
    y = torch.from_numpy(g.standard_normal((1, cfg.n_feats, L), dtype=np.float32)).clamp(-1, 1)
    cond_x = torch.from_numpy(g.standard_normal((1, cfg.n_feats, Lu), dtype=np.float32) * 0.5)
    dur = torch.full((1, Lu), 3.0)
    y_lengths = torch.LongTensor([L])
    y_mask = sequence_mask(y_lengths, L).unsqueeze(1).float()
    attn = generate_path(dur, (torch.ones(1, 1, Lu).unsqueeze(-1) * y_mask.unsqueeze(2)).squeeze(1))
    spk = torch.from_numpy(g.standard_normal((1, 1, cfg.spk_emb_dim), dtype=np.float32))
    spk = spk / spk.norm()
    dv = lambda t: t.to(dev)
    cond_x_d, y_d, y_mask_d, y_len_d, attn_d, spk_d = map(dv, (cond_x, y, y_mask, y_lengths, attn, spk))

    random.seed(0); torch.manual_seed(0)
    graph = None
    if not a.no_graph:
        from unitspeech_amd.graph import FineTuneGraph
        graph = FineTuneGraph(model, spk_d, 1, a.segment, cfg.n_feats, backward=a.backward)

    def step():
        if graph is not None:
            loss = graph.step(cond_x_d, y_d, y_len_d, attn_d)
        else:
            loss = model.fine_tune(cond_x_d, y_d, y_mask_d, y_len_d, L, attn_d, spk_d, a.segment, cfg.n_feats)
            opt.zero_grad(set_to_none=True)
            loss.backward()
        if fused:
            opt.step(max_norm=1)
        else:
            torch.nn.utils.clip_grad_norm_(model.parameters(), 1)
            opt.step()
        return loss

    losses = []
    for _ in range(a.warmup):
        losses.append(step().item())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.iters):
        loss = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.iters
    last_loss = loss.item()
    # time to ENQUEUE an iteration, taken on an empty queue (two untimed extra iterations): the host is the limit when it is close to dt
    t1 = time.perf_counter()
    for _ in range(2):
        step()
    host_dt = (time.perf_counter() - t1) / 2
    torch.cuda.synchronize()
    eng = model._get_engine()
    fwd_flops = eng.lib.us_estimator_flops(eng.handle, a.segment)          # direct-convolution count of SURVEY.md 8(d), one item
    step_flops = 3.0 * fwd_flops                                           # forward + data gradients + weight gradients
    achieved = step_flops / dt / 1e12
    res = {"metric": "fine-tune seconds/iteration (B=1, 176-frame crop, fwd+bwd+clip+Adam)", "value": dt, "unit": "s/iter",
           "higher_is_better": False, "n_gpus": 1, "dtype": "f32", "data": "synthetic",
           "config": {"workload": f"finetune.py:131-165 inner loop, B=1, {a.segment}-frame crops of a {L}-frame utterance, full-size decoder, "
                                  f"Adam lr 2e-5, clip 1.0", "optimizer": a.optimizer, "launch": "eager" if graph is None else ("hip graph" if graph.backward == "graph" else "hip graph (forward) + eager backward")},
           "iters": a.iters, "host_enqueue_s_per_iter": host_dt, "est_500_iter_s": 500 * dt, "first_losses": losses, "last_loss": last_loss,
           "roofline": {"bound": "mfma", "achieved": achieved, "peak": 157.3, "unit": "TFLOP/s", "frac": achieved / 157.3, "traffic": None,
                        "flops_per_iteration": step_flops,
                        "flops_basis": "direct-form convolution FLOPs (3x the forward count of SURVEY.md 8(d)) over wall time of the whole "
                                       "iteration incl. clip + Adam; the Winograd forward / data-gradient convolutions execute 16/36 of them"}}
    if not a.no_cpu_baseline:
        import os
        from oracle import decoder_oracle as O
        cores = min(len(os.sched_getaffinity(0)), int(os.environ.get("UNITSPEECH_CPU_THREADS", "16")))
        torch.set_num_threads(cores)
        sdc = {k: torch.from_numpy(v).clone().requires_grad_(True) for k, v in sd_np.items()}
        optc = torch.optim.Adam(list(sdc.values()), lr=2e-5)
        rs = random.Random(0)
        gen = torch.Generator().manual_seed(0)
        times = []
        for i in range(a.cpu_baseline_iters + 1):                         # first iteration untimed (allocator / thread-pool warm-up)
            t1 = time.perf_counter()
            y_cut, m_cut, cond_y = O.fine_tune_segment(cond_x, y, y_mask, y_lengths, L, attn, a.segment, cfg.n_feats, rng=rs)
            tt = torch.clamp(torch.rand(1, generator=gen), 1e-5, 1 - 1e-5)
            zz = torch.randn(y_cut.shape, generator=gen)
            lossc, _ = O.loss_t(sdc, y_cut, m_cut, cond_y, tt, spk, zz, cfg.n_feats)
            optc.zero_grad()
            lossc.backward()
            torch.nn.utils.clip_grad_norm_(list(sdc.values()), 1)
            optc.step()
            if i > 0:
                times.append(time.perf_counter() - t1)
        cpu_dt = sum(times) / len(times)
        model_name = next((ln.split(":", 1)[1].strip() for ln in open("/proc/cpuinfo") if ln.lower().startswith("model name")), "unknown")
        res["cpu_baseline"] = {"value": cpu_dt, "unit": "s/iter", "cores": cores, "kind": "port", "cpu": model_name,
                               "sample": f"{a.cpu_baseline_iters} iterations of the CPU oracle (torch autograd + clip + torch.optim.Adam) on the "
                                         f"same crop shape, after one warm-up iteration"}

    if a.check > 0:
        # same python/torch draws replayed on the CPU oracle (fresh weights, same optimiser)
        from oracle import decoder_oracle as O
        sd = {k: torch.from_numpy(v).clone().requires_grad_(True) for k, v in sd_np.items()}
        opt2 = torch.optim.Adam(list(sd.values()), lr=2e-5)
        random.seed(0); torch.manual_seed(0)
        ref_losses = []
        for _ in range(min(a.check, a.warmup)):
            y_cut, m_cut, cond_y = O.fine_tune_segment(cond_x, y, y_mask, y_lengths, L, attn, a.segment, cfg.n_feats)
            t = torch.clamp(torch.rand(1, device=dev).cpu(), 1e-5, 1 - 1e-5)          # same device generator stream as the HIP run
            z = torch.randn(y_cut.shape, device=dev).cpu()
            lossr, _ = O.loss_t(sd, y_cut, m_cut, cond_y, t, spk, z, cfg.n_feats)
            opt2.zero_grad()
            lossr.backward()
            torch.nn.utils.clip_grad_norm_(list(sd.values()), 1)
            opt2.step()
            ref_losses.append(lossr.item())
        res["oracle_losses"] = ref_losses
    print(json.dumps(res))


if __name__ == "__main__":
    main()
