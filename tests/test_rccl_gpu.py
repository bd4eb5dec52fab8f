"""RCCL on the box: a ONE-rank `nccl` process group on cuda:0 running the collectives the multi-GPU paths issue (unitspeech_amd/sharding.py),
at their real sizes.  With one rank the ring is degenerate -- this does not measure xGMI -- but it is the only RCCL the 1-GPU test box can
run: the communicator is created on the device, the calls the N-GPU bench makes are accepted and complete, and the data comes back
unchanged.  The N > 1 logic itself is covered by the 2-rank gloo tests (tests/test_sharding_gloo.py)."""
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

SCRIPT = textwrap.dedent("""
    import os, sys, time, json
    sys.path.insert(0, %r)
    import torch, torch.distributed as dist
    from unitspeech_amd import DecoderConfig, synthetic_state_dict
    from unitspeech_amd import sharding
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)        # RCCL
    cfg = DecoderConfig()
    sd = synthetic_state_dict(cfg, 0)
    timing = {}
    # world = 1 short-circuits inside the helpers, so the collectives are issued here the way the helpers issue them
    flat = sharding.pack_state_dict(cfg, sd, dev)
    ref = flat.clone()
    torch.cuda.synchronize(); dist.barrier(); t0 = time.perf_counter()
    dist.broadcast(flat, src=0)
    torch.cuda.synchronize(); t_b = time.perf_counter() - t0
    assert torch.equal(flat, ref)
    blob = torch.randn(119_000_000, device=dev)                                  # the gradient blob of the full-size decoder (476 MB)
    keep = blob.clone()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    dist.all_reduce(blob)
    blob.div_(1)
    torch.cuda.synchronize(); t_a = time.perf_counter() - t0
    assert torch.equal(blob, keep)
    v = torch.tensor([3.25], device=dev)
    dist.all_reduce(v, op=dist.ReduceOp.MAX)
    assert float(v) == 3.25
    print(json.dumps({"backend": dist.get_backend(), "world": dist.get_world_size(), "broadcast_ms": 1e3 * t_b, "allreduce_ms": 1e3 * t_a,
                      "bytes": flat.numel() * 4}))
    dist.destroy_process_group()
""") % ROOT


def test_rccl_one_rank_group_runs_the_collectives_of_the_multi_gpu_paths(tmp_path):
    import json
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = tmp_path / "rccl_one_rank.py"
    p.write_text(SCRIPT)
    r = subprocess.run([sys.executable, str(p)], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    print(d)
    assert d["backend"] == "nccl" and d["world"] == 1 and d["bytes"] > 400e6
