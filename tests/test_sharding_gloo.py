"""The N>1 path on CPU: two gloo ranks, weight broadcast + utterance sharding + max-over-ranks timing reduce."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from unitspeech_amd import DecoderConfig, synthetic_state_dict
from unitspeech_amd.sharding import broadcast_state_dict, max_over_ranks, shard_range

CFG = DecoderConfig(dim=16)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sd0 = synthetic_state_dict(CFG, 0) if rank == 0 else None
    sd = broadcast_state_dict(CFG, sd0, rank, world, "cpu")
    checksum = float(sum(v.double().abs().sum() for v in sd.values()))
    lo, hi = shard_range(11, rank, world)
    slowest = max_over_ranks(1.0 + rank, world, "cpu")
    dist.barrier()
    np.save(os.path.join(out_dir, f"r{rank}.npy"), np.array([checksum, lo, hi, slowest]))
    dist.destroy_process_group()


def test_two_rank_broadcast_and_sharding(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    ref = synthetic_state_dict(CFG, 0)
    want = float(sum(np.abs(v.astype(np.float64)).sum() for v in ref.values()))
    r = [np.load(tmp_path / f"r{k}.npy") for k in range(world)]
    for k in range(world):
        assert abs(r[k][0] - want) <= 1e-9 * want          # every rank holds rank 0's weights
        assert r[k][3] == 2.0                               # max over ranks
    assert (r[0][1], r[0][2], r[1][1], r[1][2]) == (0, 6, 6, 11)


def test_shard_range_partitions():
    for n in (1, 7, 64, 512):
        for world in (1, 2, 4, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def _grad_worker(rank, world, port, out_dir):
    """Data-parallel step on two CPU ranks: each rank's gradients = rank-dependent values; after allreduce_gradients every
    rank holds the mean, with the blob-resident gradients moved by one collective and the loose ones by a second."""
    from unitspeech_amd.sharding import allreduce_gradients
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    shapes = [(4, 3), (7,), (2, 2, 2), (5,)]
    params = [torch.nn.Parameter(torch.zeros(s)) for s in shapes]
    blob = torch.zeros(64 * 3)
    offs = [0, 64, 128]
    for i, (p, o) in enumerate(zip(params[:3], offs)):                 # three gradients are views into the blob ...
        p.grad = blob[o:o + p.numel()].view(p.shape)
        p.grad.fill_(float((rank + 1) * (i + 1)))
    params[3].grad = torch.full((5,), float(10 * (rank + 1)))         # ... one lives on its own
    params.append(torch.nn.Parameter(torch.zeros(3)))                   # and one has no gradient at all
    n = allreduce_gradients(params, world, blob=blob)
    got = [p.grad.clone() for p in params[:4]]
    np.save(os.path.join(out_dir, f"g{rank}.npy"), np.concatenate([g.reshape(-1).numpy() for g in got] + [np.array([n], np.float32)]))
    dist.destroy_process_group()


def test_two_rank_gradient_allreduce(tmp_path):
    world = 2
    mp.spawn(_grad_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r = [np.load(tmp_path / f"g{k}.npy") for k in range(world)]
    want = np.concatenate([np.full(12, 1.5), np.full(7, 3.0), np.full(8, 4.5), np.full(5, 15.0), [2.0]]).astype(np.float32)
    for k in range(world):
        np.testing.assert_allclose(r[k], want, rtol=0, atol=0)           # mean of ranks 1x and 2x; 2 collectives


# ---- bench.py's N>1 path: rank function end to end on 2 gloo ranks with the engine replaced at the model call ----------------
class _FakeWorkload:
    """Stand-in for bench.HipWorkload (same four methods): checks the broadcast weights arrived and burns a rank-dependent time."""

    def __init__(self, cfg, sd, device, a, rank):
        self.rank, self.a = rank, a
        self.checksum = float(sum(v.double().abs().sum() for v in sd.values()))
        self.calls = 0

    def step(self):
        import time
        self.calls += 1
        time.sleep(0.02 * (1 + self.rank))                 # rank 1 is the slow one: the job time must be ITS time
        return torch.zeros(self.a.batch, 80, self.a.frames)

    def sync(self):
        pass

    def profile_begin(self):
        pass

    def profile_end(self):
        return {"conv_ms": 1.0, "conv_flops": 1e9, "conv_launches": 10, "eval_ms": 2.0, "evals": 1, "flops_eval_item": 1e9}


def _bench_worker(rank, world, port, out_dir):
    import json
    import bench
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    a = bench.parse(["--gpus", str(world), "--batch", "3", "--frames", "64", "--steps", "4", "--warmup", "1", "--no-cpu-baseline"])
    a.cfg = CFG
    res = bench.run_rank(a, rank, rank, world, backend="gloo", workload_cls=_FakeWorkload, device=torch.device("cpu"))
    with open(os.path.join(out_dir, f"b{rank}.json"), "w") as f:
        json.dump(res, f)
    dist.destroy_process_group()


def test_bench_rank_function_two_gloo_ranks(tmp_path):
    import json
    world = 2
    mp.spawn(_bench_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r0 = json.load(open(tmp_path / "b0.json"))
    assert json.load(open(tmp_path / "b1.json")) is None             # only rank 0 reports
    assert r0["n_gpus"] == 2 and r0["steps"] == 4 and r0["scaling"] == "weak"
    # value = frames of ALL ranks / max-over-ranks time: rank 1 sleeps 40 ms per step
    assert r0["ms_per_step"] >= 40.0
    assert abs(r0["value"] - 2 * 3 * 64 * 4 / (r0["ms_per_step"] * 4e-3)) <= 1e-6 * r0["value"]
    assert "roofline" in r0 and "cpu_baseline" not in r0
    # what the process group says about itself (not WORLD_SIZE): both ranks, their devices, the broadcast that fed them
    rr = r0["rccl_ranks"]
    assert rr["world_size"] == 2 and rr["backend"] == "gloo" and sorted(x["rank"] for x in rr["ranks"]) == [0, 1]
    assert len({x["pid"] for x in rr["ranks"]}) == 2 and all(x["device"] == "cpu" for x in rr["ranks"])
    assert r0["broadcast_ms"] is not None and r0["broadcast_ms"] > 0 and r0["broadcast_bytes"] == 4 * sum(int(np.prod(v.shape)) for v in synthetic_state_dict(CFG, 0).values())
    # rank 0 sleeps 20 ms per step, rank 1 40 ms: the straggler is visible, and the median sits next to the mean
    assert r0["rank_busy_s"]["min"] < 0.75 * r0["rank_busy_s"]["max"] <= 0.75 * 1e-3 * r0["ms_per_step"] * 4 * 1.01
    assert r0["ms_per_step_median"] >= 40.0 and abs(r0["ms_per_step_median"] - r0["ms_per_step"]) <= 0.25 * r0["ms_per_step"]
    assert r0["roofline"]["algorithmic_bytes_per_eval"] > 0


def test_bench_spawns_its_own_ranks_and_refuses_mismatch(monkeypatch):
    import subprocess
    import sys
    import bench
    import pytest
    seen = {}
    monkeypatch.setattr(subprocess, "call", lambda cmd, env=None: seen.update(cmd=cmd, env=env) or 0)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--config", "64x8", "--steps", "2"])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 0
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=8" in cmd and "127.0.0.1" in cmd
    assert cmd[cmd.index("--gpus") + 1] == "8" and "--config" in cmd
    # a launcher's WORLD_SIZE that disagrees with --gpus is refused, also when it is 1
    monkeypatch.setenv("WORLD_SIZE", "1")
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2"])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert "disagrees" in str(e.value.code)
