"""Developer probe: where does an optimiser step spend its time (host vs device)?"""
import faulthandler
import sys
import time

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from unitspeech_amd import FusedAdam  # noqa: E402

faulthandler.dump_traceback_later(60, exit=True)
dev = torch.device("cuda", 0)
big = int(sys.argv[2]) if len(sys.argv) > 2 else 100
sizes = [1 << 20] * big + [1000] * 130
which = [sys.argv[1]] if len(sys.argv) > 1 else ["torch", "fused"]
print("start", flush=True)
for name in which:
    ps = [torch.nn.Parameter(torch.randn(n, device=dev)) for n in sizes]
    opt = (FusedAdam if name == "fused" else torch.optim.Adam)(ps, lr=2e-5)
    for it in range(6):
        for p in ps:
            p.grad = torch.randn_like(p)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if name == "fused":
            opt.step(max_norm=1)
        else:
            torch.nn.utils.clip_grad_norm_(ps, 1)
            opt.step()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(name, it, f"host {1e3 * (t1 - t0):.2f} ms, device tail {1e3 * (t2 - t1):.2f} ms", flush=True)
        faulthandler.dump_traceback_later(60, exit=True)
