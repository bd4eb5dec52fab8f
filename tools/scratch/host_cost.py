"""Host time of the library's training entry points on an EMPTY queue (one 176-frame crop): is the eager backward launch-bound?"""
import time, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from unitspeech_amd import DecoderConfig, UnitSpeech, synthetic_state_dict
cfg = DecoderConfig()
dev = torch.device("cuda:0")
m = UnitSpeech(cfg.n_feats, cfg.dim, list(cfg.dim_mults), cfg.beta_min, cfg.beta_max, cfg.pe_scale, cfg.spk_emb_dim)
m.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic_state_dict(cfg, 0).items()})
m = m.to(dev).train()
g = np.random.Generator(np.random.Philox(key=5))
T = 176
x0 = torch.from_numpy(g.standard_normal((1, 80, T), dtype=np.float32)).clamp(-1, 1).to(dev)
cond = torch.from_numpy(g.standard_normal((1, 80, T), dtype=np.float32) * .5).to(dev)
mask = torch.ones(1, 1, T, device=dev)
spk = torch.from_numpy(g.standard_normal((1, 1, cfg.spk_emb_dim), dtype=np.float32)).to(dev); spk = spk / spk.norm()
eng = m._get_engine()
times = {}
def wrap(name):
    f = getattr(eng.lib, name)
    def w(*a):
        torch.cuda.synchronize()
        t0 = time.perf_counter(); r = f(*a); t1 = time.perf_counter()
        torch.cuda.synchronize(); t2 = time.perf_counter()
        times.setdefault(name, []).append((t1 - t0, t2 - t0))
        return r
    setattr(eng.lib, name, w)
for n in ("us_estimator_forward_train", "us_estimator_backward"):
    wrap(n)
for i in range(6):
    for p in m.parameters(): p.grad = None
    loss, _ = m.compute_loss(x0, mask, cond, spk_emb=spk)
    loss.backward()
torch.cuda.synchronize()
for n, v in times.items():
    v = v[2:]
    print(f"{n}: host enqueue {1e3*sum(a for a, _ in v)/len(v):.3f} ms, enqueue+GPU (empty queue at start) {1e3*sum(b for _, b in v)/len(v):.3f} ms")
