"""Would level 0 run faster one item at a time (126 MB tensors against the 256 MB Infinity Cache) than as the CFG batch of 3 (377 MB)?
One score-network evaluation at 80 x 1024: batch 3 against three evaluations at batch 1, same inputs."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from unitspeech_amd import DecoderConfig, UnitSpeech, synthetic_state_dict
B = int(sys.argv[1]) if len(sys.argv) > 1 else 3
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
cfg = DecoderConfig()
dev = torch.device("cuda:0")
m = UnitSpeech(cfg.n_feats, cfg.dim, list(cfg.dim_mults), cfg.beta_min, cfg.beta_max, cfg.pe_scale, cfg.spk_emb_dim)
m.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic_state_dict(cfg, 0).items()})
m = m.to(dev).eval()
g = np.random.Generator(np.random.Philox(key=9))
T = 1024
x = torch.from_numpy(g.standard_normal((3, 80, T), dtype=np.float32)).to(dev)
mu = torch.from_numpy(g.standard_normal((3, 80, T), dtype=np.float32) * .5).to(dev)
mask = torch.ones(3, 1, T, device=dev)
t = torch.full((3,), 0.5, device=dev)
spk = torch.from_numpy(g.standard_normal((3, 1, cfg.spk_emb_dim), dtype=np.float32)).to(dev)
def run():
    with torch.no_grad():
        if B == 3:
            return [m.estimator(x, mask, mu, t, spk)]
        return [m.estimator(x[i:i+1], mask[i:i+1], mu[i:i+1], t[i:i+1], spk[i:i+1]) for i in range(3)]
for _ in range(3): run()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps): run()
torch.cuda.synchronize()
print(f"batch {B}: {(time.perf_counter() - t0) / reps * 1e3:.3f} ms per 3 items")
