"""The inference.py / finetune.py command-line mirrors run end to end in --synthetic mode (GPU)."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def test_inference_cli_ten_steps_matches_the_reference_golden(tmp_path, golden):
    """BASELINE configs[0] through the command line: `inference.py --synthetic`, 10 diffusion steps, the text, speaker embedding, z and
    per-step noise of tests/golden/tts_full.npz -- the REFERENCE's own `execute_text_to_speech` + inference.py:140 on the same stand-in
    front end (tools/make_goldens_r2.py) -- and the saved mel against that golden's de-normalised mel."""
    g = golden("tts_full")
    out = tmp_path / "sample.wav"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "inference.py"), "--synthetic", "--text", str(g["text"]), "--diffusion_steps",
                        str(int(g["n_steps"])), "--noise_key", "4242", "--spk_seed", "9", "--generated_sample_path", str(out)],
                       capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    mel = np.load(str(out)[:-4] + ".mel.npy")
    ref = g["mel"][0]
    assert int(g["n_steps"]) == 10 and mel.shape == ref.shape and np.isfinite(mel).all()
    err = float(np.abs(mel.astype(np.float64) - ref).mean())
    tol = 1e-3 * float(g["mel_max"] - g["mel_min"]) / 2          # the north-star 1e-3 on the normalised mel, through the de-normalisation
    print(f"\ninference.py, 10 steps: mel-L1 vs the reference golden {err:.3e} (tolerance {tol:.3e}, mean|mel| {np.abs(ref).mean():.2f})")
    assert err <= tol


def test_inference_cli_with_the_hip_text_encoder_and_duration_predictor(tmp_path):
    out = tmp_path / "sample.wav"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "inference.py"), "--synthetic", "--learned_frontend", "--text", "buna ziua",
                        "--diffusion_steps", "10", "--generated_sample_path", str(out)], capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    mel = np.load(str(out)[:-4] + ".mel.npy")
    assert mel.shape[0] == 80 and mel.shape[1] >= 19 and np.isfinite(mel).all()        # 19 symbols, at least one frame each


def _losses(stdout):
    return [float(line.split()[-1]) for line in stdout.splitlines() if line.startswith("iter ")]


def test_finetune_cli_synthetic_and_from_a_features_file(tmp_path):
    """`finetune.py --synthetic` (the loop of finetune.py:131-165 on seeded tensors), and the same tensors handed over as the file a user
    of the reference would write after its pre-steps (finetune.py:86-128: mel, cond_x, duration, spk_emb, mel_min / mel_max):
    `--features` must run the same loop on them -- same losses, same checkpoint layout."""
    import torch
    r = subprocess.run([sys.executable, os.path.join(ROOT, "finetune.py"), "--synthetic", "--n_iters", "3", "--ID", "5", "--out_dir", str(tmp_path)],
                       capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    ck = torch.load(tmp_path / "5.pt", map_location="cpu")
    assert set(ck) == {"model", "spk_emb", "mel_min", "mel_max"} and len(ck["model"]) == 230
    want = _losses(r.stdout)
    assert len(want) == 2 and all(np.isfinite(want))
    # the tensors --synthetic draws for --ID 5 (finetune.py), un-normalised speaker embedding, raw mel + range: the loader normalises both
    g = np.random.Generator(np.random.Philox(key=5))
    mel = torch.from_numpy(g.standard_normal((1, 80, 600), dtype=np.float32)).clamp(-1, 1)
    cond_x = torch.from_numpy(g.standard_normal((1, 80, 200), dtype=np.float32)) * 0.5
    spk = torch.from_numpy(g.standard_normal((1, 1, 256), dtype=np.float32))
    mel_min, mel_max = torch.tensor(-11.5), torch.tensor(2.0)
    feats = {"mel": (mel + 1) / 2 * (mel_max - mel_min) + mel_min, "mel_is_normalized": False, "cond_x": cond_x,
             "duration": torch.full((1, 200), 3.0), "spk_emb": spk.reshape(1, 256) * 3.0, "mel_min": mel_min, "mel_max": mel_max}
    torch.save(feats, tmp_path / "features.pt")
    r2 = subprocess.run([sys.executable, os.path.join(ROOT, "finetune.py"), "--synthetic", "--features", str(tmp_path / "features.pt"), "--n_iters", "3",
                         "--ID", "6", "--out_dir", str(tmp_path)], capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r2.returncode == 0, r2.stderr[-2000:]
    got = _losses(r2.stdout)
    print(f"\nfinetune.py losses: --synthetic {want}, --features {got}")
    assert len(got) == 2
    for a, b in zip(got, want):
        assert abs(a - b) <= 1e-4 * max(1.0, abs(b)), (got, want)      # the raw mel went through one de-normalise / normalise round trip
    ck2 = torch.load(tmp_path / "6.pt", map_location="cpu")
    assert set(ck2) == {"model", "spk_emb", "mel_min", "mel_max"} and abs(float(ck2["spk_emb"].norm()) - 1.0) < 1e-5
    # a file without cond_x / unit is refused with a message, not a traceback
    torch.save({k: v for k, v in feats.items() if k != "cond_x"}, tmp_path / "bad.pt")
    r3 = subprocess.run([sys.executable, os.path.join(ROOT, "finetune.py"), "--synthetic", "--features", str(tmp_path / "bad.pt"), "--n_iters", "1",
                         "--out_dir", str(tmp_path)], capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r3.returncode != 0 and "cond_x" in r3.stderr and "Traceback" not in r3.stderr


def test_finetune_cli_with_the_hip_unit_encoder(tmp_path):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "finetune.py"), "--synthetic", "--learned_frontend", "--n_iters", "3", "--ID", "5",
                        "--out_dir", str(tmp_path)],
                       capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    assert all(np.isfinite(_losses(r.stdout)))


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


def test_config3_finetune_500_iterations_through_the_cli(tmp_path):
    """BASELINE configs[3] at its stated length: `finetune.py` 500-iteration speaker adaptation (finetune.py:131-165 of the reference: crop,
    t, z, forward, backward, clip_grad_norm_(1), Adam(2e-5)) on one synthetic 600-frame utterance.  Finite losses that go down, device memory
    flat over the run, no f16x3 range event, a checkpoint in the reference's layout, and the wall time of the loop (<= 10 s)."""
    import re
    import torch
    r = subprocess.run([sys.executable, os.path.join(ROOT, "finetune.py"), "--synthetic", "--n_iters", "500", "--ID", "7", "--out_dir", str(tmp_path),
                        "--report_memory"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    losses = _losses(r.stdout)
    assert len(losses) == 11 and all(np.isfinite(losses))            # iterations 0, 50, ..., 450, 499
    pb, pa = (float(re.search(rf"probe loss {w} ([0-9.]+)", r.stdout).group(1)) for w in ("before", "after"))
    assert np.isfinite(pa) and pa < pb, (pb, pa)                      # the same 8 fixed (t, z) draws before and after: the adaptation made progress
    m = re.search(r"500 iterations in ([0-9.]+) s", r.stdout)
    mem = [int(v) for v in re.findall(r"allocated (\d+) MiB", r.stdout)]
    rng = re.search(r"range status (\d+)", r.stdout)
    print(f"\nfinetune.py 500 iterations: {m.group(1)} s, probe loss {pb:.4f} -> {pa:.4f}, allocated MiB {mem[:2]} ... {mem[-1]}")
    assert m and float(m.group(1)) <= 10.0
    assert len(mem) >= 3 and mem[-1] <= mem[1] + 64                    # flat after the first iterations (graph capture, optimiser state)
    assert rng and int(rng.group(1)) == 0
    ck = torch.load(tmp_path / "7.pt", map_location="cpu")
    assert set(ck) == {"model", "spk_emb", "mel_min", "mel_max"} and all(torch.isfinite(v).all() for v in ck["model"].values())
