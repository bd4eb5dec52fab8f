"""500 fine-tune iterations (default FineTuneGraph: forward graph + eager two-stream backward) then 10 decodes: finite losses, the loss
goes down, device memory does not grow."""
import sys, os, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from unitspeech_amd import DecoderConfig, UnitSpeech, synthetic_state_dict, FusedAdam
from unitspeech_amd.graph import FineTuneGraph
from unitspeech_amd.util import generate_path, sequence_mask
cfg = DecoderConfig(); dev = torch.device("cuda:0")
m = UnitSpeech(cfg.n_feats, cfg.dim, list(cfg.dim_mults), cfg.beta_min, cfg.beta_max, cfg.pe_scale, cfg.spk_emb_dim)
m.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic_state_dict(cfg, 0).items()})
m = m.to(dev).train()
opt = FusedAdam(m.parameters(), lr=2e-5)
g = np.random.Generator(np.random.Philox(key=11))
L, Lu = 600, 200
y = torch.from_numpy(g.standard_normal((1, 80, L), dtype=np.float32)).clamp(-1, 1).to(dev)
cond_x = torch.from_numpy(g.standard_normal((1, 80, Lu), dtype=np.float32) * .5).to(dev)
y_len = torch.LongTensor([L]).to(dev)
y_mask = sequence_mask(y_len, L).unsqueeze(1).float()
attn = generate_path(torch.full((1, Lu), 3.0, device=dev), (torch.ones(1, 1, Lu, device=dev).unsqueeze(-1) * y_mask.unsqueeze(2)).squeeze(1))
spk = torch.from_numpy(g.standard_normal((1, 1, cfg.spk_emb_dim), dtype=np.float32)).to(dev); spk = spk / spk.norm()
random.seed(0); torch.manual_seed(0)
graph = FineTuneGraph(m, spk, 1, 176, 80)
losses, mem = [], []
for i in range(500):
    loss = graph.step(cond_x, y, y_len, attn)
    opt.step(max_norm=1)
    if i % 50 == 49:
        losses.append(float(loss)); mem.append(torch.cuda.memory_allocated(dev) >> 20)
print("losses every 50:", [round(v, 4) for v in losses])
print("allocated MiB every 50:", mem)
assert all(np.isfinite(losses)) and mem[-1] <= mem[1] + 64
assert m.range_status() == 0
m.eval()
with torch.no_grad():
    for i in range(10):
        out = m.reverse_diffusion(torch.randn(1, 80, 256, device=dev), torch.ones(1, 1, 256, device=dev), torch.randn(1, 80, 256, device=dev) * .5,
                                  spk, 10) if hasattr(m, "reverse_diffusion") else None
print("decode ok", None if out is None else bool(torch.isfinite(out).all()), "MiB", torch.cuda.memory_allocated(dev) >> 20)
