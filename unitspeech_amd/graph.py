"""The fine-tune iteration as a HIP graph (forward, or forward + backward).

A speaker-adaptation iteration (finetune.py:131-165) is ~900 kernel launches of a few microseconds each on a 176-frame crop: the
GPU work is a few milliseconds, the launch stream 13-14.  Shapes, parameter storage and workspaces never change across the 500
iterations, so the whole `compute_loss` forward + `loss.backward()` (weight re-pack, score network forward with its tape, loss,
backward, gradient blob) is captured once into a HIP graph (`torch.cuda.CUDAGraph`, i.e. hipGraph; the library's launches go to
torch's current stream, which is the capture stream) and replayed per iteration.  Outside the graph stay only what depends on the
host: the random crop (Python's `random`, as in the reference) and the optimiser step (its bias corrections are host scalars).

    g = FineTuneGraph(decoder, spk_emb, batch=1, segment_size=176)
    for _ in range(n_iters):
        loss = g.step(cond_x, y, y_lengths, attn)       # == decoder.fine_tune(...) + loss.backward()
        optimizer.step(max_norm=1)

The gaussian draws inside (`torch.rand` for t, `torch.randn` for z) use torch's graph-safe Philox bookkeeping, so the stream of
random numbers, hence the loss trajectory, is the eager one.

`backward="eager"` (the default; `UNITSPEECH_FT_BACKWARD=graph` or `backward="graph"` captures everything as described above): only the forward (re-pack, score network with its tape, objective) is captured; `loss.backward(retain_graph=True)`
then runs per iteration as ordinary launches on the retained autograd graph, whose saved tensors are the capture's static buffers
(the library keeps the tape live: US_BACKWARD_KEEP_TAPE).  Outside a capture the library's backward runs its weight-gradient chains
on a second stream, which a replayed graph cannot do at a profit, and the host enqueues the backward while the GPU is
still replaying the forward.
"""
from __future__ import annotations

import os

import torch


class FineTuneGraph:
    def __init__(self, decoder, spk_emb: torch.Tensor, batch: int, segment_size: int, n_feats: int = None, warmup: int = 2,
                 backward: str = None):
        dev = next(decoder.parameters()).device
        if backward is None:
            backward = os.environ.get("UNITSPEECH_FT_BACKWARD", "eager")      # measured: 10.5 vs 11.1 ms per iteration
        if backward not in ("graph", "eager"):
            raise ValueError(f"FineTuneGraph: backward must be 'graph' or 'eager', got {backward!r}")
        self.backward = backward
        if dev.type != "cuda":
            raise RuntimeError("FineTuneGraph needs the decoder on a ROCm device")
        self.decoder = decoder
        self.n_feats = n_feats if n_feats is not None else decoder.n_feats
        self.segment_size = int(segment_size)
        self.spk = spk_emb.detach().to(dev, torch.float32).contiguous()
        B, F, S = int(batch), self.n_feats, self.segment_size
        # static inputs of the graph: `fine_tune_segment` writes the crop straight into them
        self.y = torch.zeros(B, F, S, device=dev)
        self.mask = torch.ones(B, 1, S, device=dev)
        self.cond = torch.zeros(B, F, S, device=dev)
        self.params = [p for p in decoder.parameters() if p.requires_grad]
        # warm-up on a side stream, as stream capture requires (allocator pools, lazy initialisation, kernel module loads)
        gen_state = torch.cuda.get_rng_state(dev)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(max(int(warmup), 1)):
                for p in self.params:
                    p.grad = None
                loss, _ = decoder.compute_loss(self.y, self.mask, self.cond, spk_emb=self.spk)
                loss.backward()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        torch.cuda.set_rng_state(gen_state, dev)          # the warm-up must not consume the training run's random numbers
        for p in self.params:
            p.grad = None
        decoder.invalidate_weights()                       # the captured graph re-packs EVERY weight: parameters change every replay
        self.graph = torch.cuda.CUDAGraph()
        # capture on the warm-up stream: the AccumulateGrad nodes created there stay bound to it, and a capture on another stream would
        # only work through autograd's cross-stream waits happening to be captured too
        self._engine = decoder._get_engine()
        self._side = side
        self._dev = dev
        if backward == "graph":
            with torch.cuda.graph(self.graph, stream=side):
                loss, _ = decoder.compute_loss(self.y, self.mask, self.cond, spk_emb=self.spk)
                loss.backward()
            self._loss_live = None
        else:
            self._engine.keep_tapes = True            # the forward's record must survive its backward calls (_EstimatorFn)
            try:
                with torch.cuda.graph(self.graph, stream=side):
                    loss, _ = decoder.compute_loss(self.y, self.mask, self.cond, spk_emb=self.spk)
            finally:
                self._engine.keep_tapes = False
            self._loss_live = loss                    # holds the autograd graph; its saved tensors are the capture's static buffers
        self.loss = loss.detach()
        # the graph holds raw pointers: parameter and gradient storage, the engine's packed weights and staging tables
        self._handle = self._engine.handle.value
        self._ptrs = [(p.data_ptr(), p.grad.data_ptr() if p.grad is not None else 0) for p in self.params]

    def _check_alive(self):
        eng = self.decoder._get_engine()
        if eng is not self._engine or eng.handle.value != self._handle:
            raise RuntimeError("FineTuneGraph: the decoder's engine was re-created (device move, `exact` switched) after capture; "
                               "build a new FineTuneGraph")
        for p, (dp, gp) in zip(self.params, self._ptrs):
            if p.data_ptr() != dp:
                raise RuntimeError("FineTuneGraph: parameter storage changed after capture (decoder.to(), load_state_dict(assign=True), "
                                   "...); build a new FineTuneGraph")
            if self.backward == "eager":
                continue                              # gradients are fresh tensors every iteration
            if (p.grad is not None and p.grad.data_ptr() != gp) or (p.grad is None and gp != 0):
                raise RuntimeError("FineTuneGraph: parameter or gradient storage changed after capture (decoder.to(), "
                                   "load_state_dict(assign=True), zero_grad(set_to_none=True), ...); build a new FineTuneGraph")

    def step(self, cond_x, y, y_lengths, attn) -> torch.Tensor:
        """One `fine_tune` call plus `loss.backward()`: gradients land in `p.grad` (static tensors with backward="graph", fresh views of one
        blob per iteration otherwise), returns the loss (a static tensor)."""
        self._check_alive()
        # the f16x3 range word: the captured forward cannot poll (no host read inside a capture), so the check lives here, host-side and
        # outside the replay -- an overflow of iteration i raises RangeError at iteration i + 1, as in the eager loop (_EstimatorFn)
        self._engine.range_poll()
        self.decoder.fine_tune_segment(cond_x, y, y_lengths, attn, self.segment_size, self.n_feats, out=(self.y, self.mask, self.cond))
        if self.backward == "graph":
            self.graph.replay()
            self._engine.range_post()             # (the eager backward posts it itself: _EstimatorFn.backward)
            return self.loss
        # forward graph and eager backward on the capture stream (the autograd nodes are bound to it), fenced against the caller's stream on
        # both sides: the crop above and the optimiser step after this call run there
        cur = torch.cuda.current_stream(self._dev)
        self._side.wait_stream(cur)
        with torch.cuda.stream(self._side):
            self.graph.replay()
            for p in self.params:
                p.grad = None
            self._loss_live.backward(retain_graph=True)
        cur.wait_stream(self._side)
        return self.loss
