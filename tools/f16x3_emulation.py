#!/usr/bin/env python3
"""NumPy emulation of the f16x3 GEMM: operands as two fp16 planes (hi = fp16(x), lo = fp16((x - hi) * 2^11)), three
products with fp32 accumulation, C = C_hh + 2^-11 C_x, against an fp32 sgemm and the fp64 truth, on operands of mixed magnitude.

    python tools/f16x3_emulation.py            # prints mean |error| / mean |C| per K"""
import numpy as np


def split16(x):
    c = np.clip(x, -65504.0, 65504.0).astype(np.float32)
    hi = c.astype(np.float16)
    lo = np.clip((c - hi.astype(np.float32)) * np.float32(2048.0), -65504.0, 65504.0).astype(np.float16)
    return hi.astype(np.float32), lo.astype(np.float32)


def f16x3_matmul(a, b):
    """a [M, K], b [K, N] fp32 -> fp32, the way conv_igemm_kernel<.., F16> forms it (fp32 accumulation by numpy's sgemm here)."""
    a1, a2 = split16(a)
    b1, b2 = split16(b)
    hh = a1 @ b1
    x = a1 @ b2 + a2 @ b1
    return hh + x * np.float32(2.0 ** -11)


def errors(K, M=384, N=192, seed=0):
    rng = np.random.default_rng(seed)
    a = (rng.standard_normal((M, K)) * rng.choice([0.01, 1.0, 30.0], size=(M, K))).astype(np.float32)
    b = (rng.standard_normal((K, N)) / np.sqrt(K)).astype(np.float32)
    ref = a.astype(np.float64) @ b.astype(np.float64)
    scale = np.abs(ref).mean()
    return (np.abs(a @ b - ref).mean() / scale, np.abs(f16x3_matmul(a, b) - ref).mean() / scale)


if __name__ == "__main__":
    for K in (128, 1152, 4608):
        e32, e16 = errors(K)
        print(f"K = {K:5d}: fp32 sgemm {e32:.3e}   f16x3 {e16:.3e}   ratio {e16 / e32:.3f}")
