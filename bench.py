#!/usr/bin/env python3
"""Benchmark of the diffusion-decoder hot path (BASELINE.json metric: mel-frames/sec @ 50 diffusion steps, 80x1024).

One "step" = one complete `UnitSpeech.forward` (reverse diffusion, 50 steps, text+speaker CFG = 3 score-network
evaluations per step) over a batch of synthetic utterances that is already resident in HBM.  At N=1 the workload
is BASELINE.json configs[1]: B=1, 80x1024 mel.  With --gpus N every rank runs the same per-GPU workload on its own
utterances (weak scaling, no data-path collective; the decoder weights are broadcast once from rank 0 over RCCL).

Launch:  python bench.py [--gpus N] [--steps K] [--warmup W]      (N > 1 without a launcher: bench.py starts its own N ranks
                                                                   through torch.distributed.run before touching the GPU)
         python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
                bench.py --gpus N --steps K --warmup W
         python bench.py --config 64x8                              (BASELINE.json configs[4]: 64 utterances/GPU x 8 GPUs)
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
PEAK_F16_MFMA_TFLOPS = 2500.0     # same guide: dense bf16 / fp16 matrix peak (v_mfma_f32_32x32x16_f16), no sparsity
HOP, SR = 256, 22050              # conf/hydra_config.py:37,39  -> seconds of speech per mel frame


CONFIGS = {            # BASELINE.json configs -> (utterances per GPU, GPUs)
    "1x1": (1, 1),     # configs[1]: B=1, the bench line
    "64x1": (64, 1),   # configs[2]: B=64 on one GPU (roofline run)
    "64x8": (64, 8),   # configs[4]: B=512 sharded over 8 GPUs
}


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None)
    ap.add_argument("--config", choices=sorted(CONFIGS), default=None, help="BASELINE.json config shorthand: utterances/GPU x GPUs")
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=None, help="utterances per GPU (configs[1]: 1; configs[2]/[4]: 64)")
    ap.add_argument("--frames", type=int, default=1024)
    ap.add_argument("--diffusion-steps", type=int, default=50)
    ap.add_argument("--micro-batch", type=int, default=0)
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="N > 1 ranks that all use cuda:0 and talk over gloo: exercises the launch / broadcast / sharding / timing path of a "
                         "multi-GPU run on a one-GPU box (RCCL refuses two ranks on one device); the throughput it prints is not a scaling number")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-steps", type=int, default=4, help="diffusion steps timed on the host CPU")
    ap.add_argument("--profile-steps", type=int, default=2, help="untimed decodes after the timed region whose conv launches are event-sampled")
    ap.add_argument("--no-anchor", action="store_true",
                    help="skip the untimed anchor legs after the timed region: one decode on an exact-fp32 (v_mfma_f32_32x32x2_f32) handle with "
                         "the same inputs and noise, and the replay of the reference's 50-step T=1024 golden")
    a = ap.parse_args(argv)
    cb, cg = CONFIGS[a.config] if a.config else (1, 1)
    if a.batch is None:
        a.batch = cb
    if a.gpus is None:
        a.gpus = cg
    if a.gpus < 1 or a.batch < 1:
        ap.error("--gpus and --batch must be >= 1")
    return a


def spawn_ranks(a) -> int:
    """`python bench.py --gpus N` with no launcher in the environment: start N ranks (one per GPU) as a child
    `torch.distributed.run` and return its exit code.  Runs BEFORE this process makes any GPU call (importing torch does not
    initialise HIP), so no initialised process is ever replaced or forked."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    argv = [x for x in sys.argv[1:]]
    if "--gpus" not in " ".join(argv):
        argv += ["--gpus", str(a.gpus)]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def cpu_model_name() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


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
    return {"value": frames / (per_step * n_diff), "unit": "mel-frames/s", "cores": cores, "kind": "port", "cpu": cpu_model_name(),
            "sample": f"{steps_timed} of {n_diff} diffusion steps (3 score evaluations each) of the B=1 80x{frames} workload, "
                      f"{dt:.1f} s measured, scaled linearly",
            "sec_per_diffusion_step": per_step}


class HipWorkload:
    """One rank's share of the job on the HIP decoder: its utterances resident in HBM, one `step()` = one complete
    `UnitSpeech.forward` over them.  (The gloo test of the N>1 path substitutes a CPU stand-in with the same four methods.)"""

    def __init__(self, cfg, sd, device, a, rank):
        self.model = UnitSpeech(cfg.n_feats, cfg.dim, list(cfg.dim_mults), cfg.beta_min, cfg.beta_max, cfg.pe_scale, cfg.spk_emb_dim)
        self.model = self.model.to(device).eval()
        self.model.load_state_dict(sd, strict=True)
        self.model.micro_batch = a.micro_batch
        self.device, self.a, self.rank = device, a, rank
        # this rank's shard of utterances: items [rank*B, (rank+1)*B) of the global batch, resident in HBM
        self.inp = {k: torch.from_numpy(v).to(device) for k, v in synthetic_inputs(cfg, a.batch, a.frames, seed=1000 + rank).items()}
        self.eng = None

    def step(self):
        i, a = self.inp, self.a
        return self.model(i["z"], i["mask"], i["cond"], i["spk_emb"], a.diffusion_steps, 1.0, 1.0, rng="philox", seed=1234,
                          utt_offset=self.rank * a.batch)

    def sync(self):
        torch.cuda.synchronize()

    def exact_anchor(self):
        """The headline runs on f16x3 GEMMs (fp32-accurate split-operand products on the fp16 matrix cores).  Anchor it, in the same
        record, to the reference's own arithmetic: the decode of this rank's FIRST utterance (same z, same Philox noise) on a
        US_CREATE_EXACT_FP32 handle -- every GEMM on v_mfma_f32_32x32x2_f32 -- timed once after a warm-up, and the mel-L1 distance of the two
        outputs.  One utterance whatever the batch (a B = 64 decode on the exact engine would take a minute): a B = 1 figure."""
        a, i = self.a, self.inp
        one = lambda: self.model(i["z"][:1], i["mask"][:1], i["cond"][:1], i["spk_emb"][:1], a.diffusion_steps, 1.0, 1.0, rng="philox",
                                 seed=1234, utt_offset=self.rank * a.batch)
        default = one()
        self.sync()
        self.model.exact = True
        try:
            one()                                 # creates the exact engine, packs its weights
            self.sync()
            t0 = time.perf_counter()
            ex = one()
            self.sync()
            dt = time.perf_counter() - t0
        finally:
            self.model.exact = False
        return {"value": a.frames / dt, "unit": "mel-frames/s", "ms_per_step": 1e3 * dt, "batch": 1,
                "mel_l1_vs_default": float((ex - default).abs().mean()), "mean_abs_out": float(ex.abs().mean()),
                "engine": "US_CREATE_EXACT_FP32: every GEMM on v_mfma_f32_32x32x2_f32 (the arithmetic of the fp32 reference), one decode of one utterance"}

    def golden_anchor(self, cfg):
        """tests/golden/loop_full_N50_T1024.npz: the REFERENCE's own 50-step decode at 80x1024 (fp32 and fp64 columns; tools/make_goldens_r2.py)
        replayed on this engine with the golden's explicit noise: mel-L1 against both columns (north-star tolerance 1e-3)."""
        path = os.path.join(ROOT, "tests", "golden", "loop_full_N50_T1024.npz")
        if not os.path.exists(path) or cfg != DecoderConfig():
            return None
        g = np.load(path)
        inp = {k: torch.from_numpy(v).to(self.device) for k, v in
               synthetic_inputs(cfg, 1, 1024, seed=int(g["seed"]), n_steps=50, lengths=[int(g["lengths"][0])]).items()}
        out = self.model(inp["z"], inp["mask"], inp["cond"], inp["spk_emb"], 50, 1.0, 1.0, noise=inp["noise"]).double().cpu()
        ref32, ref64 = torch.from_numpy(g["out"]).double(), torch.from_numpy(g["out_fp64"]).double()
        return {"vs_reference_fp32": float((out - ref32).abs().mean()), "vs_reference_fp64": float((out - ref64).abs().mean()),
                "reference_fp32_vs_fp64": float((ref32 - ref64).abs().mean()), "mean_abs_out": float(ref32.abs().mean()),
                "tolerance": 1e-3, "golden": "tests/golden/loop_full_N50_T1024.npz"}

    def profile_begin(self):
        self.eng = self.model._sync(self.device)      # creates the handle / pushes the weights when no warm-up step has run yet
        self.eng.lib.us_profile_enable(self.eng.handle, 1)
        self.eng.lib.us_profile_read(self.eng.handle, None, None, None, None, None, 1)

    def profile_end(self):
        conv_ms, conv_fl, ev_ms = C.c_double(), C.c_double(), C.c_double()
        conv_n, ev_n = C.c_int64(), C.c_int64()
        lib, h = self.eng.lib, self.eng.handle
        lib.us_profile_read(h, C.byref(conv_ms), C.byref(conv_fl), C.byref(conv_n), C.byref(ev_ms), C.byref(ev_n), 0)
        f_ms, f_fl, f_n = C.c_double(), C.c_double(), C.c_int64()
        lib.us_profile_read_f16(h, C.byref(f_ms), C.byref(f_fl), C.byref(f_n))
        lib.us_profile_read(h, None, None, None, None, None, 1)
        lib.us_profile_enable(h, 0)
        return {"conv_ms": conv_ms.value, "conv_flops": conv_fl.value, "conv_launches": int(conv_n.value), "eval_ms": ev_ms.value,
                "evals": int(ev_n.value), "flops_eval_item": lib.us_estimator_flops(h, self.a.frames),
                "f16_ms": f_ms.value, "f16_flops": f_fl.value, "f16_launches": int(f_n.value)}


# the sources whose kernels the HBM-side counters of profiles/*_pmc_summary.json were collected on (inference path)
TRAFFIC_SOURCES = ("conv_igemm.hip", "wino.hip", "wino4.hip", "wino4_coef.h", "ops.hip", "attn.hip", "decoder.hip", "kernels.h", "pack_f16.h")


def pmc_traffic():
    """HBM-side bytes from the committed counter summary (separate rocprofv3 --pmc passes, tools/pmc_collect.sh + pmc_summary.py): figures
    of the tree they were collected on, not of the run that prints them.  So the summary carries a fingerprint of the inference kernels'
    sources (`source_sha256`, unitspeech_amd/_build.py), and a summary whose fingerprint differs from the tree bench.py runs from is
    REFUSED: `traffic` is null and `"stale": true` (the refused figures stay visible under `stale_values`).  (The GPU box has no .git, so
    the check is on file contents, not on `git diff <commit> HEAD`.)  Returns (bytes per conv launch, bytes of ALL kernels per
    score-network evaluation at B' = 3, 80x1024, source)."""
    from unitspeech_amd._build import source_fingerprint
    now = source_fingerprint(TRAFFIC_SOURCES)
    for name in ("r04_pmc_summary.json", "r03_pmc_summary.json"):
        path = os.path.join(ROOT, "profiles", name)
        if os.path.exists(path):
            try:
                d = json.load(open(path))
            except Exception:
                continue
            src = {"file": "profiles/" + name, "commit": d.get("commit"), "source_sha256": d.get("source_sha256"), "tree_sha256": now}
            per_launch, per_eval = d.get("hbm_bytes_per_launch"), d.get("all_kernels_hbm_side_bytes_per_evaluation")
            if d.get("source_sha256") != now:
                src.update({"stale": True, "stale_values": {"hbm_bytes_per_launch": per_launch, "all_kernels_per_eval": per_eval}})
                return None, None, src
            src["stale"] = False
            return per_launch, per_eval, src
    return None, None, None


def algorithmic_bytes_per_eval(n_params, Bp, T):
    """the weights once (fp32) + B' * T * 964 B of tensor I/O (x, mu [80, T] and the mask in, the score out)."""
    return 4.0 * n_params + float(Bp) * T * 964.0


def rank_census(dist, rank, world, device, backend):
    """What the process group actually is, gathered from every rank AFTER init_process_group: the line must not repeat WORLD_SIZE."""
    if dist is None:
        name = torch.cuda.get_device_name(device) if device.type == "cuda" else "cpu"
        return 1, [{"rank": 0, "device": str(device), "name": name}]
    me = {"rank": dist.get_rank(), "device": str(device), "name": torch.cuda.get_device_name(device) if device.type == "cuda" else "cpu",
          "host": os.uname().nodename, "pid": os.getpid()}
    if device.type == "cuda":
        try:
            me["uuid"] = str(torch.cuda.get_device_properties(device).uuid)
        except Exception:
            pass
    got = [None] * world
    dist.all_gather_object(got, me)
    return dist.get_world_size(), got


def run_rank(a, rank, local, world, backend="nccl", workload_cls=HipWorkload, device=None):
    """One rank of the job; rank 0 returns the result dict, the others None."""
    if device is None:
        torch.cuda.set_device(local)
        device = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist_
        dist = dist_
        if not dist.is_initialized():
            kw = {"device_id": device} if backend == "nccl" else {}
            dist.init_process_group(backend, rank=rank, world_size=world, **kw)

    if backend == "nccl" and world > 1 and torch.cuda.device_count() < world:
        raise SystemExit(f"{world} ranks over RCCL need {world} visible GPUs, this process sees {torch.cuda.device_count()} "
                         "(--rehearse-on-one-gpu runs the N-rank path on one device over gloo)")
    n_ranks, ranks = rank_census(dist, rank, world, device, backend)
    if n_ranks != world:
        raise SystemExit(f"the process group has {n_ranks} ranks, the launcher announced {world}")

    cfg = a.cfg if getattr(a, "cfg", None) is not None else DecoderConfig()
    B, T, N = a.batch, a.frames, a.diffusion_steps
    # rank 0 generates the synthetic checkpoint; the others receive it as ONE packed 476.6 MB fp32 blob over RCCL
    timing = {}
    sd = broadcast_state_dict(cfg, synthetic_state_dict(cfg, 0) if rank == 0 else None, rank, world, device, timing=timing)
    n_params = int(sum(v.numel() for v in sd.values()))
    wl = workload_cls(cfg, sd, device, a, rank)
    del sd

    out = None
    for _ in range(a.warmup):
        out = wl.step()
    wl.sync()

    # timed region: exactly `steps` decodes between barrier + synchronize on both sides; no event sampling inside
    if dist is not None:
        dist.barrier()
    wl.sync()
    per_step = []
    t0 = time.perf_counter()
    for _ in range(a.steps):
        s0 = time.perf_counter()
        out = wl.step()
        wl.sync()
        per_step.append(time.perf_counter() - s0)
    wl.sync()
    busy = time.perf_counter() - t0                 # this rank's own K steps, before it waits for the others
    if dist is not None:
        dist.barrier()
    own = time.perf_counter() - t0
    elapsed = max_over_ranks(own, world, device)
    slowest, fastest = max_over_ranks(busy, world, device), -max_over_ranks(-busy, world, device)
    median_step = max_over_ranks(float(np.median(per_step)) if per_step else 0.0, world, device)
    assert out is None or torch.isfinite(out).all(), "non-finite decoder output"
    # per-launch times for the roofline object: separate, untimed decodes with the library's event sampling switched on
    wl.profile_begin()
    for _ in range(max(1, min(a.steps, a.profile_steps))):
        wl.step()
    wl.sync()
    prof = wl.profile_end()
    anchors = {}
    if rank == 0 and not a.no_anchor and hasattr(wl, "exact_anchor"):
        anchors["exact_fp32"] = wl.exact_anchor()
        anchors["golden_mel_l1"] = wl.golden_anchor(cfg)

    res = None
    if rank == 0:
        frames_total = world * B * T * a.steps
        value = frames_total / elapsed
        ms_per_step = 1e3 * elapsed / a.steps
        n_cfg = 3
        flops_step = prof["flops_eval_item"] * n_cfg * B * N
        # dominant kernel: conv_igemm_kernel (every 3x3 / 1x1 / strided / transposed convolution of the U-Net).  Since round 2 its
        # launches run as f16x3 GEMMs: fp32-accurate products formed as three fp16 MFMA products of two-plane split operands, so
        # the matrix cores EXECUTE three fp16 FLOPs per fp32-equivalent FLOP.  `achieved` / `peak` / `frac` are stated for what the
        # hardware executes (fp16 MFMA FLOPs of the f16x3 launches against the dense fp16 peak); `fp32_equivalent` restates the
        # same launches plus the few remaining fp32-MFMA ones as 2*M*N*K FLOPs against the fp32 matrix peak, comparable with round 1.
        launches = max(prof["conv_launches"], 1)
        avg_ms = prof["conv_ms"] / launches
        flops_per_launch = prof["conv_flops"] / launches
        eq = (flops_per_launch / (avg_ms * 1e-3)) / 1e12 if avg_ms > 0 else 0.0
        f16_n = prof.get("f16_launches", 0)
        traffic, traffic_all, traffic_src = pmc_traffic()
        alg_bytes = algorithmic_bytes_per_eval(n_params, 3, 1024)      # at the shape the counters were collected on (B' = 3, T = 1024)
        if f16_n > 0:
            f16_avg_ms = prof["f16_ms"] / f16_n
            f16_exec_per_launch = 3.0 * prof["f16_flops"] / f16_n
            achieved = f16_exec_per_launch / (f16_avg_ms * 1e-3) / 1e12
            peak, kernel = PEAK_F16_MFMA_TFLOPS, "conv_igemm_kernel, f16x3 forms (v_mfma_f32_32x32x16_f16; fp32-accurate split-operand GEMM)"
            per_launch, avg, n_l = f16_exec_per_launch, f16_avg_ms, f16_n
            basis = ("fp16 MFMA FLOPs executed per f16x3 conv_igemm launch = 3 x 2*M*N*K (three products per fp32-equivalent product; "
                     "Winograd F(2x2,3x3) GEMMs at their reduced count)")
        else:
            achieved, peak, kernel = eq, PEAK_F32_MFMA_TFLOPS, "conv_igemm_kernel (v_mfma_f32_32x32x2_f32 implicit GEMM)"
            per_launch, avg, n_l = flops_per_launch, avg_ms, launches
            basis = "executed MFMA FLOPs per conv_igemm launch (Winograd F(2x2,3x3) GEMMs at their reduced count)"
        roofline = {"bound": "mfma", "kernel": kernel, "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                    "traffic": traffic, "traffic_source": traffic_src,
                    "traffic_all_kernels_per_eval": traffic_all, "algorithmic_bytes_per_eval": alg_bytes,
                    "traffic_over_algorithmic": (traffic_all / alg_bytes) if traffic_all else None,
                    "flops_per_launch": per_launch, "avg_launch_ms": avg, "launches_sampled": n_l, "flops_basis": basis,
                    "fp32_equivalent": {"achieved": eq, "peak": PEAK_F32_MFMA_TFLOPS, "frac": eq / PEAK_F32_MFMA_TFLOPS,
                                        "flops_per_launch": flops_per_launch, "avg_launch_ms": avg_ms, "launches_sampled": launches,
                                        "note": "all conv_igemm launches as 2*M*N*K FLOPs over their time; above 1.0 of the fp32 MFMA peak "
                                                "means faster than any exact-fp32 matrix-core kernel could run them"},
                    "conv_share_of_eval_time": (prof["conv_ms"] / prof["eval_ms"]) if prof["eval_ms"] > 0 else None,
                    "sampled_eval_ms": (prof["eval_ms"] / max(prof["evals"], 1)),
                    # whole_job_tflops is the direct-convolution count of SURVEY.md 8(d) over wall time
                    "whole_job_tflops": flops_step * world / (ms_per_step * 1e-3) / 1e12}
        res = {"metric": "mel-frames/sec @ 50 diffusion steps, 80x1024", "value": value, "unit": "mel-frames/s",
               "n_gpus": n_ranks, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms_per_step,
               "ms_per_step_median": 1e3 * median_step, "value_at_median": world * B * T / median_step if median_step > 0 else None,
               "rccl_ranks": {"backend": backend if world > 1 else None, "world_size": n_ranks, "ranks": ranks},
               "broadcast_ms": timing.get("broadcast_ms"), "broadcast_bytes": timing.get("bytes"),
               "rank_busy_s": {"min": fastest, "max": slowest, "note": "each rank's own K steps before the closing barrier"},
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "dtype_note": "fp32 storage and fp32 accumulation everywhere; GEMM products at fp32 accuracy as f16x3 (two fp16 planes per "
                             "operand, three fp16 MFMA products; per-evaluation error vs fp64 at the fp32 reference's level)",
               "config": {"workload": f"B={B} utterance(s)/GPU, 80x{T} mel, {N} diffusion steps, text+spk CFG (3 score evals/step), "
                                      f"full-size decoder (119.1M params, synthetic weights), built-in Philox noise",
                          "batch_per_gpu": B, "frames": T, "diffusion_steps": N, "parallelism": f"utterance-sharded x{world}"},
               "rtf": (elapsed / a.steps) / (B * T * HOP / SR),
               "roofline": roofline}
        for k, v in anchors.items():
            if v is not None:
                res[k] = v
        if "exact_fp32" in res and B == 1:
            res["exact_fp32"]["default_over_exact"] = value / world / res["exact_fp32"]["value"] if res["exact_fp32"]["value"] > 0 else None
        if not a.no_cpu_baseline and world == 1:
            res["cpu_baseline"] = cpu_baseline(cfg, T, N, a.cpu_baseline_steps)
    if dist is not None and backend == "nccl":
        dist.destroy_process_group()
    return res


def main():
    a = parse()
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and a.gpus > 1:
        sys.exit(spawn_ranks(a))       # no launcher: start our own ranks; nothing in this process has touched the GPU
    world = int(env_world) if env_world is not None else 1
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} disagrees with WORLD_SIZE={world} set by the launcher")
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU: the HIP decoder has no CPU fallback")
    if a.rehearse_on_one_gpu:
        res = run_rank(a, rank, 0, world, backend="gloo")
        if res is not None:
            res["rehearsal"] = "all ranks on cuda:0 over gloo: not a scaling measurement"
        import torch.distributed as dist
        if world > 1 and dist.is_initialized():
            dist.destroy_process_group()
    else:
        res = run_rank(a, rank, local, world)
    if res is not None:
        print(json.dumps(res))


if __name__ == "__main__":
    main()
