"""Fused gradient clipping + Adam for the decoder's fine-tune / training loop.

The reference's inner loop (finetune.py:163-165) is

    loss.backward()
    torch.nn.utils.clip_grad_norm_(decoder.parameters(), max_norm=1)
    optimizer.step()                      # torch.optim.Adam(decoder.parameters(), lr=2e-5)

`FusedAdam` keeps torch.optim.Adam's constructor and state layout (`state[p] = {"step", "exp_avg", "exp_avg_sq"}`) and runs
clip + update for all tensors in three HIP launches (`us_clip_adam_step`, csrc/optim.hip):

    optimizer.step(max_norm=1)            # == clip_grad_norm_(params, 1) followed by Adam.step()
    optimizer.last_grad_norm              # device scalar: the total norm clip_grad_norm_ would have returned

Only what the reference uses is supported: weight_decay=0, amsgrad=False, maximize=False, fp32 parameters on one ROCm device.
There is no CPU fallback.
"""
import ctypes as C

import torch

from . import _lib

_CHUNK = 4096


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, amsgrad=False):
        if weight_decay != 0.0 or amsgrad:
            raise ValueError("FusedAdam implements the reference's configuration only: weight_decay=0, amsgrad=False")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=amsgrad))
        self.lib = _lib.load()
        self.last_grad_norm = None
        self._tables = {}          # per group: cached device tables keyed by the pointer tuple
        self._pinned = {}          # per group: pinned staging ring for the pointer table

    def _state(self, p):
        st = self.state[p]
        if len(st) == 0:
            st["step"] = torch.zeros((), dtype=torch.float32)
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        return st

    @torch.no_grad()
    def step(self, closure=None, max_norm=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for gi, group in enumerate(self.param_groups):
            ps = [p for p in group["params"] if p.grad is not None]
            if not ps:
                continue
            dev = ps[0].device
            if dev.type != "cuda":
                raise RuntimeError("FusedAdam needs parameters on a ROCm device (no CPU fallback); got " + str(dev))
            for p in ps:
                if p.dtype != torch.float32 or p.device != dev or not p.is_contiguous():
                    raise RuntimeError("FusedAdam: parameters must be contiguous fp32 tensors on one device")
                if p.grad.dtype != torch.float32 or not p.grad.is_contiguous():
                    p.grad = p.grad.float().contiguous()
            sts = [self._state(p) for p in ps]
            key = tuple((p.data_ptr(), p.grad.data_ptr(), p.numel()) for p in ps)
            tab = self._tables.get(gi)
            if tab is None or tab["key"] != key:
                ptrs = torch.tensor([[p.data_ptr() for p in ps], [p.grad.data_ptr() for p in ps],
                                     [s["exp_avg"].data_ptr() for s in sts], [s["exp_avg_sq"].data_ptr() for s in sts],
                                     [p.numel() for p in ps]], dtype=torch.int64)
                # gradients usually live in fresh storage every iteration: upload the table without a host sync, through ONE pinned
                # staging allocation used as a ring (pinning costs seconds on some hosts; the event guards a slot's reuse)
                ring = self._pinned.get(gi)
                if ring is None or ring["buf"].shape[1:] != ptrs.shape:
                    ring = {"buf": torch.empty((4,) + tuple(ptrs.shape), dtype=torch.int64).pin_memory(), "ev": [None] * 4, "i": 0}
                    self._pinned[gi] = ring
                slot = ring["i"] % 4
                ring["i"] += 1
                if ring["ev"][slot] is not None:
                    ring["ev"][slot].synchronize()
                stage = ring["buf"][slot]
                stage.copy_(ptrs)
                if tab is None or tab["sizes"] != tuple(p.numel() for p in ps):
                    bt, bo = [], []
                    for i, p in enumerate(ps):
                        for off in range(0, p.numel(), _CHUNK):
                            bt.append(i)
                            bo.append(off)
                    tab = {"sizes": tuple(p.numel() for p in ps),
                           "blk_tensor": torch.tensor(bt, dtype=torch.int32).to(dev), "blk_off": torch.tensor(bo, dtype=torch.int64).to(dev),
                           "partial": torch.empty(len(bt) + 2, dtype=torch.float32, device=dev), "n_blocks": len(bt)}
                tab["key"] = key
                if "ptrs" not in tab or tab["ptrs"].shape != ptrs.shape:
                    tab["ptrs"] = torch.empty(ptrs.shape, dtype=torch.int64, device=dev)
                tab["ptrs"].copy_(stage, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(dev))
                ring["ev"][slot] = ev
                self._tables[gi] = tab
            step_t = sts[0]["step"]
            step = int(step_t.item()) + 1          # CPU scalar, as in torch's default (non-capturable) Adam
            for s in sts:
                s["step"] += 1
            b1, b2 = group["betas"]
            ptrs = tab["ptrs"]
            with torch.cuda.device(dev):
                rc = self.lib.us_clip_adam_step(ptrs[0].data_ptr(), ptrs[1].data_ptr(), ptrs[2].data_ptr(), ptrs[3].data_ptr(),
                                                ptrs[4].data_ptr(), tab["blk_tensor"].data_ptr(), tab["blk_off"].data_ptr(), len(ps),
                                                tab["n_blocks"], float(group["lr"]), float(b1), float(b2), float(group["eps"]), step,
                                                float(max_norm) if max_norm is not None else 0.0, tab["partial"].data_ptr(),
                                                C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
            _lib.check(rc, None, "us_clip_adam_step")
            if max_norm is not None:
                self.last_grad_norm = tab["partial"][tab["n_blocks"]]
            torch._C._increment_version(ps)        # the kernel wrote the parameters in place: let version-keyed caches (the HIP
                                                   # engine's weight sync) see it; takes an ITERABLE of tensors
        return loss
