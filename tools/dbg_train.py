import sys, numpy as np, torch
sys.path.insert(0, '.')
from unitspeech_amd import DecoderConfig, UnitSpeech, synthetic_state_dict, synthetic_inputs
from oracle import decoder_oracle as O
DEV='cuda:0'
for dim in [int(a) for a in sys.argv[1:]] or [16, 32, 64, 128]:
    cfg = DecoderConfig(dim=dim)
    T = 16
    sdn = synthetic_state_dict(cfg,0)
    m = UnitSpeech(cfg.n_feats, cfg.dim, list(cfg.dim_mults), cfg.beta_min, cfg.beta_max, cfg.pe_scale, cfg.spk_emb_dim)
    m.load_state_dict({k: torch.from_numpy(v) for k,v in sdn.items()})
    m = m.to(DEV).train()
    inp = {k: torch.from_numpy(v) for k,v in synthetic_inputs(cfg, 2, T, seed=3, lengths=[T, T-5]).items()}
    t = torch.tensor([0.3, 0.7])
    go = torch.from_numpy(np.random.default_rng(0).standard_normal((2,80,T)).astype(np.float32))
    out = m.estimator(inp['z'].to(DEV), inp['mask'].to(DEV), inp['cond'].to(DEV), t.to(DEV), inp['spk_emb'].to(DEV))
    (out * go.to(DEV)).sum().backward(); torch.cuda.synchronize()
    sd = {k: torch.from_numpy(v).clone().requires_grad_(True) for k,v in sdn.items()}
    ref = O.estimator_forward(sd, inp['z'], inp['mask'], inp['cond'], t, inp['spk_emb'])
    (ref*go).sum().backward()
    P = dict(m.named_parameters())
    bad = []
    for k in sd:
        if sd[k].grad is None: continue
        g1 = P[k].grad.cpu(); g0 = sd[k].grad
        err = (g1-g0).abs().max().item()/(g0.abs().max().item()+1e-12)
        if err > 1e-3: bad.append((k, round(err,4), g0.abs().max().item(), g1.abs().max().item()))
    print('dim', dim, 'fwd err', (out.detach().cpu()-ref.detach()).abs().max().item(), 'n bad', len(bad))
    print('   BAD:', ' '.join(sorted(set(k.replace('estimator.','').rsplit('.',1)[0] for k,_,_,_ in bad))))
    good=[k for k in sd if sd[k].grad is not None and k not in [b[0] for b in bad]]
    print('   GOOD:', ' '.join(sorted(set(k.replace('estimator.','').rsplit('.',1)[0] for k in good))))
