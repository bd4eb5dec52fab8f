#!/usr/bin/env python3
"""Benchmark of the diffusion-decoder hot path (BASELINE.json metric: mel-frames/sec @ 50 diffusion steps, 80x1024).

One "step" = one complete `UnitSpeech.forward` (reverse diffusion, 50 steps, text+speaker CFG = 3 score-network
evaluations per step) over a batch of synthetic utterances that is already resident in HBM.  At N=1 the workload
is BASELINE.json configs[1]: B=1, 80x1024 mel.  With --gpus N every rank runs the same per-GPU workload on its own
utterances (weak scaling, no data-path collective; the decoder weights are broadcast once from rank 0 over RCCL).

Launch:  python bench.py [--gpus 1] [--steps K] [--warmup W]
         python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
                bench.py --gpus N --steps K --warmup W
Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from unitspeech_amd import DecoderConfig, UnitSpeech, synthetic_inputs, synthetic_state_dict  # noqa: E402
from unitspeech_amd.sharding import broadcast_state_dict, max_over_ranks  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md: dense fp32 matrix peak (v_mfma_f32_32x32x2_f32)
HOP, SR = 256, 22050              # conf/hydra_config.py:37,39  -> seconds of speech per mel frame


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=1, help="utterances per GPU (configs[1]: 1; configs[2]/[4]: 64)")
    ap.add_argument("--frames", type=int, default=1024)
    ap.add_argument("--diffusion-steps", type=int, default=50)
    ap.add_argument("--micro-batch", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-steps", type=int, default=4, help="diffusion steps timed on the host CPU")
    return ap.parse_args()


def cpu_baseline(cfg, frames, n_diff, steps_timed):
    """The CPU oracle (oracle/decoder_oracle.py, a torch-CPU restatement pinned to the reference) on the host cores:
    `steps_timed` diffusion steps of the same B=1 text+spk CFG workload, scaled linearly to n_diff steps."""
    from oracle import decoder_oracle as O
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    # a 1-GPU box exposes every host thread but grants a 16-CPU share; oversubscribing it makes ATen ~4x slower
    cores = min(cores, int(os.environ.get("UNITSPEECH_CPU_THREADS", "16")))
    torch.set_num_threads(cores)
    sd = O.to_torch(synthetic_state_dict(cfg, 0))
    inp = {k: torch.from_numpy(v) for k, v in synthetic_inputs(cfg, 1, frames, seed=0, n_steps=steps_timed).items()}
    tu = sd["text_uncon"].repeat(1, 1, frames)
    su = sd["spk_uncon"] / sd["spk_uncon"].norm()
    t = torch.full((1,), 0.5)
    with torch.no_grad():
        O.classifier_free_guidance(sd, inp["z"], inp["mask"], inp["cond"], t, inp["spk_emb"], tu, su, 1.0, 1.0, cfg.pe_scale)  # warm-up
        t0 = time.perf_counter()
        O.reverse_diffusion(sd, inp["z"], inp["mask"], inp["cond"], inp["spk_emb"], steps_timed, 1.0, 1.0, noise=inp["noise"])
        dt = time.perf_counter() - t0
    per_step = dt / steps_timed
    return {"value": frames / (per_step * n_diff), "unit": "mel-frames/s", "cores": cores, "kind": "port",
            "sample": f"{steps_timed} of {n_diff} diffusion steps (3 score evaluations each) of the B=1 80x{frames} workload, "
                      f"{dt:.1f} s measured, scaled linearly",
            "sec_per_diffusion_step": per_step}


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus and world > 1:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU: the HIP decoder has no CPU fallback")
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist_
        dist = dist_
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    cfg = DecoderConfig()
    B, T, N = a.batch, a.frames, a.diffusion_steps
    # rank 0 generates the synthetic checkpoint; the others receive it as ONE packed 476.6 MB fp32 blob over RCCL
    sd = broadcast_state_dict(cfg, synthetic_state_dict(cfg, 0) if rank == 0 else None, rank, world, device)
    model = UnitSpeech(cfg.n_feats, cfg.dim, list(cfg.dim_mults), cfg.beta_min, cfg.beta_max, cfg.pe_scale, cfg.spk_emb_dim)
    model = model.to(device).eval()
    model.load_state_dict(sd, strict=True)
    model.micro_batch = a.micro_batch
    del sd

    # this rank's shard of utterances: items [rank*B, (rank+1)*B) of the global batch, resident in HBM
    inp = {k: torch.from_numpy(v).to(device) for k, v in synthetic_inputs(cfg, B, T, seed=1000 + rank).items()}

    def step():
        return model(inp["z"], inp["mask"], inp["cond"], inp["spk_emb"], N, 1.0, 1.0, rng="philox", seed=1234,
                     utt_offset=rank * B)

    for _ in range(a.warmup):
        out = step()
    torch.cuda.synchronize()
    eng = model._sync(device)          # creates the handle / pushes the weights when no warm-up step has run yet
    eng.lib.us_profile_enable(eng.handle, 1)
    eng.lib.us_profile_read(eng.handle, None, None, None, None, None, 1)

    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        out = step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    elapsed = max_over_ranks(elapsed, world, device)
    assert torch.isfinite(out).all(), "non-finite decoder output"

    conv_ms, conv_fl, ev_ms = C.c_double(), C.c_double(), C.c_double()
    conv_n, ev_n = C.c_int64(), C.c_int64()
    eng.lib.us_profile_read(eng.handle, C.byref(conv_ms), C.byref(conv_fl), C.byref(conv_n), C.byref(ev_ms), C.byref(ev_n), 1)
    eng.lib.us_profile_enable(eng.handle, 0)

    if rank == 0:
        frames_total = world * B * T * a.steps
        value = frames_total / elapsed
        ms_per_step = 1e3 * elapsed / a.steps
        n_cfg = 3
        flops_eval_item = eng.lib.us_estimator_flops(eng.handle, T)
        flops_step = flops_eval_item * n_cfg * B * N
        # dominant kernel: conv_igemm_kernel<32> (every 3x3 / 1x1 / strided / transposed convolution of the U-Net)
        launches = max(int(conv_n.value), 1)
        avg_ms = conv_ms.value / launches
        flops_per_launch = conv_fl.value / launches
        achieved = (flops_per_launch / (avg_ms * 1e-3)) / 1e12 if avg_ms > 0 else 0.0
        roofline = {"bound": "mfma", "kernel": "conv_igemm_kernel (v_mfma_f32_32x32x2_f32 implicit GEMM)",
                    "achieved": achieved, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                    "frac": achieved / PEAK_F32_MFMA_TFLOPS, "traffic": None,
                    "flops_per_launch": flops_per_launch, "avg_launch_ms": avg_ms, "launches_sampled": launches,
                    "conv_share_of_eval_time": (conv_ms.value / ev_ms.value) if ev_ms.value > 0 else None,
                    "sampled_eval_ms": (ev_ms.value / max(int(ev_n.value), 1)),
                    # achieved counts the FLOPs the MFMA units EXECUTE (the Winograd GEMMs at their 2.25x reduced count);
                    # whole_job_tflops is the direct-convolution count of SURVEY.md 8(d) over wall time, which the
                    # Winograd levels push past what the matrix cores execute
                    "flops_basis": "executed MFMA FLOPs per conv_igemm launch (Winograd F(2x2,3x3) GEMMs at their reduced count)",
                    "whole_job_tflops": flops_step * world / (ms_per_step * 1e-3) / 1e12}
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc):
            try:
                roofline["traffic"] = json.load(open(pmc)).get("hbm_bytes_per_launch")
            except Exception:
                pass
        res = {"metric": "mel-frames/sec @ 50 diffusion steps, 80x1024", "value": value, "unit": "mel-frames/s",
               "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms_per_step,
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": f"B={B} utterance(s)/GPU, 80x{T} mel, {N} diffusion steps, text+spk CFG (3 score evals/step), "
                                      f"full-size decoder (119.1M params, synthetic weights), built-in Philox noise",
                          "batch_per_gpu": B, "frames": T, "diffusion_steps": N, "parallelism": f"utterance-sharded x{world}"},
               "rtf": (elapsed / a.steps) / (B * T * HOP / SR),
               "roofline": roofline}
        if not a.no_cpu_baseline and world == 1:
            res["cpu_baseline"] = cpu_baseline(cfg, T, N, a.cpu_baseline_steps)
        print(json.dumps(res))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
