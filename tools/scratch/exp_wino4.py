#!/usr/bin/env python3
"""A/B of the 4-wide Winograd forms (US_WINO4, csrc/wino4.hip): parity numbers and decode time per configuration, one process.
usage: python tools/scratch/exp_wino4.py "0,0,0,0" "0,44,44,24" ..."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from unitspeech_amd import DecoderConfig, UnitSpeech, synthetic_inputs, synthetic_state_dict  # noqa: E402

DEV = "cuda:0"
FULL = DecoderConfig()
G = lambda d: {k: torch.from_numpy(np.asarray(v)) for k, v in d.items()}
l1 = lambda a, b: float((a.double().cpu() - b.double().cpu()).abs().mean())


def build(sd, exact=False):
    m = UnitSpeech(FULL.n_feats, FULL.dim, list(FULL.dim_mults), FULL.beta_min, FULL.beta_max, FULL.pe_scale, FULL.spk_emb_dim)
    m.load_state_dict(sd, strict=True)
    m.exact = exact
    return m.to(DEV).eval()


def main():
    cfgs = sys.argv[1:] or ["0,0,0,0", "0,44,44,24"]
    sd = {k: torch.from_numpy(v) for k, v in synthetic_state_dict(FULL, 0).items()}
    gold_e = G(np.load(os.path.join(ROOT, "tests/golden/estimator_full.npz")))
    gold_l = G(np.load(os.path.join(ROOT, "tests/golden/loop_full_N50_T1024.npz")))
    T = 1024
    inp = G(synthetic_inputs(FULL, 1, T, seed=21, lengths=[T - 40]))
    x3, mask3 = inp["z"].repeat(3, 1, 1).to(DEV), inp["mask"].repeat(3, 1, 1).to(DEV)
    mu3 = torch.cat([sd["text_uncon"].repeat(1, 1, T), inp["cond"], inp["cond"]], 0).to(DEV)
    spk3 = torch.cat([inp["spk_emb"], sd["spk_uncon"] / sd["spk_uncon"].norm(), inp["spk_emb"]], 0).to(DEV)
    t3 = torch.full((3,), 0.63, device=DEV)
    linp = {k: v.to(DEV) for k, v in G(synthetic_inputs(FULL, 1, T, seed=int(gold_l["seed"]), n_steps=50, lengths=[int(gold_l["lengths"][0])])).items()}
    NB = int(os.environ.get("EXP_BATCH", "1"))
    binp = {k: v.to(DEV) for k, v in G(synthetic_inputs(FULL, NB, T, seed=1000)).items()}
    ex = build(sd, exact=True)
    with torch.no_grad():
        ref_ex = ex.estimator(x3, mask3, mu3, t3, spk3)
    del ex
    for cfg in cfgs:
        os.environ["US_WINO4"] = cfg
        m = build(sd)
        with torch.no_grad():
            out = m.estimator(gold_e["x"].to(DEV), gold_e["mask"].to(DEV), gold_e["mu"].to(DEV), gold_e["t"].to(DEV), gold_e["spk_emb"].to(DEV))
            e_g, e_64 = l1(out, gold_e["out"]), l1(out, gold_e["out_fp64"])
            out = m.estimator(x3, mask3, mu3, t3, spk3)
            e_ex = l1(out, ref_ex)
            lo = m(linp["z"], linp["mask"], linp["cond"], linp["spk_emb"], 50, 1.0, 1.0, noise=linp["noise"])
            e_l32, e_l64 = l1(lo, gold_l["out"]), l1(lo, gold_l["out_fp64"])
            for _ in range(2):
                m(binp["z"], binp["mask"], binp["cond"], binp["spk_emb"], 50, 1.0, 1.0, rng="philox", seed=1234)
            torch.cuda.synchronize()
            ts = []
            for _ in range(6 if NB == 1 else 3):
                t0 = time.perf_counter()
                m(binp["z"], binp["mask"], binp["cond"], binp["spk_emb"], 50, 1.0, 1.0, rng="philox", seed=1234)
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t0)
        print(f"US_WINO4={cfg:14s} eval T=64 vs golden {e_g:.2e} vs fp64 {e_64:.2e} | eval T=1024 vs exact-fp32 engine {e_ex:.2e} | "
              f"50-step T=1024 vs ref fp32 {e_l32:.2e} fp64 {e_l64:.2e} | decode B={NB} {1e3 * float(np.median(ts)):.1f} ms = {NB * T / float(np.median(ts)):.0f} frames/s",
              flush=True)
        del m
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
