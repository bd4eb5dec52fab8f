"""The inference.py / finetune.py command-line mirrors run end to end in --synthetic mode (GPU)."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def test_inference_cli_synthetic(tmp_path):
    out = tmp_path / "sample.wav"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "inference.py"), "--synthetic", "--text", "buna ziua", "--diffusion_steps", "3",
                        "--generated_sample_path", str(out)], capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    mel = np.load(str(out)[:-4] + ".mel.npy")
    assert mel.shape[0] == 80 and mel.shape[1] > 0 and np.isfinite(mel).all()


def test_inference_cli_with_the_hip_text_encoder_and_duration_predictor(tmp_path):
    out = tmp_path / "sample.wav"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "inference.py"), "--synthetic", "--learned_frontend", "--text", "buna ziua",
                        "--diffusion_steps", "3", "--generated_sample_path", str(out)], capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    mel = np.load(str(out)[:-4] + ".mel.npy")
    assert mel.shape[0] == 80 and mel.shape[1] >= 19 and np.isfinite(mel).all()        # 19 symbols, at least one frame each


def test_finetune_cli_synthetic(tmp_path):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "finetune.py"), "--synthetic", "--learned_frontend", "--n_iters", "3", "--ID", "5",
                        "--out_dir", str(tmp_path)],
                       capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    import torch
    ck = torch.load(tmp_path / "5.pt", map_location="cpu")
    assert set(ck) == {"model", "spk_emb", "mel_min", "mel_max"} and len(ck["model"]) == 230


def test_pretrain_step_bench_runs(tmp_path):
    """bench_pretrain.py (B crops per GPU, fwd + bwd + [all-reduce] + HIP clip+Adam) at a small batch: finite loss, sane JSON."""
    import json
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench_pretrain.py"), "--batch", "2", "--iters", "2", "--warmup", "1"],
                       capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["n_gpus"] == 1 and d["batch_per_gpu"] == 2 and d["value"] > 0 and np.isfinite(d["last_loss"])
    assert set(d["ms_breakdown"]) == {"fwd", "bwd", "allreduce", "optim"}


def test_frontend_bench_runs():
    """bench_frontend.py (text Encoder + DurationPredictor per utterance, with its oracle comparison): sane JSON, parity inside the bench."""
    import json
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench_frontend.py"), "--symbols", "60", "--iters", "3"], capture_output=True, text=True,
                       timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["finite"] and d["value"] > 0 and d["roofline"]["bound"] == "hbm" and d["cpu_baseline"]["kind"] == "port"
    assert d["max_abs_diff_vs_oracle"] <= 2e-5


def test_finetune_loss_trajectory_of_three_iterations_vs_oracle():
    """BASELINE configs[3] beyond one iteration: bench_finetune.py --check 3 replays the first three fine-tune iterations (crop, t, z, forward,
    backward, clip, Adam: finetune.py:131-165) on the CPU oracle under torch autograd with the same draws; the losses of iterations 2 and
    3 depend on the weights the earlier updates produced, so agreement pins the whole loop, optimiser included."""
    import json
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench_finetune.py"), "--iters", "3", "--warmup", "3", "--check", "3",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    got, ref = d["first_losses"], d["oracle_losses"]
    assert len(got) == 3 and len(ref) == 3 and len(set(got)) == 3
    for a, b in zip(got, ref):
        assert abs(a - b) <= 2e-5 * max(1.0, abs(b)), (got, ref)
