#!/usr/bin/env python3
"""Cook-Toom F(m, 3) transforms for a set of interpolation points (+ infinity) and an fp32 emulation of the Winograd pipeline
(input transform fp32, operands as f16x3-like 22-bit values, GEMM accumulate fp32, output transform fp32) against an fp64 direct sum."""
import itertools
import sys
from fractions import Fraction as Fr

import numpy as np


def cook_toom(points, m, r=3):
    """Returns (AT [m x n], G [n x r], BT [n x n]) as fp64 arrays, n = m + r - 1, points: n - 1 finite points (Fractions), last = infinity."""
    n = m + r - 1
    assert len(points) == n - 1
    pts = [Fr(p) for p in points]
    # polynomial evaluation matrices: V_k[i][j] = p_i^j (k columns), infinity row picks the leading coefficient
    def vander(k):
        V = [[p ** j for j in range(k)] for p in pts]
        V.append([Fr(0)] * (k - 1) + [Fr(1)])
        return V
    # Y = AT [ (G g) * (BT d) ]:  AT = V_m^T (transposed evaluation for outputs), G = scaled V_r, BT = inverse-transposed of V_n
    Vm, Vr, Vn = vander(m), vander(r), vander(n)
    # Lagrange scaling: f_i = prod_{j != i} (p_i - p_j)
    f = []
    for i, p in enumerate(pts):
        v = Fr(1)
        for j, q in enumerate(pts):
            if i != j:
                v *= (p - q)
        f.append(v)
    f.append(Fr(1))
    G = [[Vr[i][j] / f[i] for j in range(r)] for i in range(n)]
    AT = [[Vm[i][j] for i in range(n)] for j in range(m)]
    # BT = (Vn^-1 scaled)^T such that the algorithm is exact: solve via matrix inverse of Vn with scaling f
    A = np.array([[float(x) for x in row] for row in Vn], dtype=object)
    # exact inverse with Fractions
    M = [[Vn[i][j] for j in range(n)] + [Fr(int(i == j)) for j in range(n)] for i in range(n)]
    for c in range(n):
        piv = next(i for i in range(c, n) if M[i][c] != 0)
        M[c], M[piv] = M[piv], M[c]
        pv = M[c][c]
        M[c] = [x / pv for x in M[c]]
        for i in range(n):
            if i != c and M[i][c] != 0:
                fac = M[i][c]
                M[i] = [x - fac * y for x, y in zip(M[i], M[c])]
    Vinv = [row[n:] for row in M]            # Vn^-1
    # BT[i][j] = f[i] * Vinv[j][i]   (transpose of the inverse, rows scaled by f)
    BT = [[f[i] * Vinv[j][i] for j in range(n)] for i in range(n)]
    tof = lambda X: np.array([[float(x) for x in row] for row in X], dtype=np.float64)
    return tof(AT), tof(G), tof(BT)


def check(AT, G, BT, m):
    rng = np.random.default_rng(0)
    n = m + 2
    d = rng.standard_normal(n); g = rng.standard_normal(3)
    y = AT @ ((G @ g) * (BT @ d))
    ref = np.array([sum(d[i + k] * g[k] for k in range(3)) for i in range(m)])
    return np.abs(y - ref).max()


def split22(x):
    """value as hi + lo with hi = fp16(x), lo = fp16((x - hi) * 2^11) / 2^11: the f16x3 operand representation"""
    hi = x.astype(np.float16).astype(np.float32)
    lo = ((x - hi) * np.float32(2048)).astype(np.float16).astype(np.float32) / np.float32(2048)
    return (hi + lo).astype(np.float32)


def emulate(mh, mw, ptsh, ptsw, C=256, Co=64, tiles=48, seed=1, bias_mean=0.0):
    ATh, Gh, BTh = cook_toom(ptsh, mh) if mh > 1 else (np.eye(1), None, None)
    ATw, Gw, BTw = cook_toom(ptsw, mw)
    nh, nw = mh + 2, mw + 2
    rng = np.random.default_rng(seed)
    d = (rng.standard_normal((tiles, C, nh, nw)) + bias_mean).astype(np.float32)           # activations after Mish: not zero-mean
    g = (rng.standard_normal((Co, C, 3, 3)) / np.sqrt(9 * C)).astype(np.float32)
    # reference: fp64 direct
    ref = np.zeros((tiles, Co, mh, mw))
    d64, g64 = d.astype(np.float64), g.astype(np.float64)
    for i in range(mh):
        for j in range(mw):
            ref[:, :, i, j] = np.einsum("tcuv,ocuv->to", d64[:, :, i:i + 3, j:j + 3], g64)
    # weights: U = G g G^T in fp64, rounded to the 22-bit operand form
    U = np.einsum("ia,ocab,jb->ocij", Gh, g64, Gw)
    U = split22(U.astype(np.float32))
    # input transform in fp32 (two separable passes, each a short fp32 sum)
    f32 = np.float32
    V = np.einsum("ia,tcab->tcib", BTh.astype(f32), d).astype(f32)
    V = np.einsum("jb,tcib->tcij", BTw.astype(f32), V).astype(f32)
    V = split22(V)
    # GEMM per frequency: products exact, accumulation in fp32 (emulated: chunks of 32 channels summed in fp32)
    M = np.zeros((tiles, Co, nh, nw), dtype=f32)
    for c0 in range(0, C, 32):
        part = np.einsum("tcij,ocij->toij", V[:, c0:c0 + 32].astype(np.float64), U[:, c0:c0 + 32].astype(np.float64))
        M = (M + part.astype(f32)).astype(f32)
    Y = np.einsum("ia,toab->toib", ATh.astype(f32), M).astype(f32)
    Y = np.einsum("jb,toib->toij", ATw.astype(f32), Y).astype(f32)
    err = np.abs(Y.astype(np.float64) - ref)
    return err.mean() / np.abs(ref).mean(), err.max() / np.abs(ref).mean()


if __name__ == "__main__":
    std4 = [0, 1, -1, 2, -2]
    std2 = [0, 1, -1]
    for name, p in (("std {0,1,-1,2,-2}", std4), ("{0,1,-1,1/2,-1/2}", [0, 1, -1, Fr(1, 2), Fr(-1, 2)]),
                    ("{0,1,-1,1/2,-2}", [0, 1, -1, Fr(1, 2), -2]), ("{0,1,-1,2,-1/2}", [0, 1, -1, 2, Fr(-1, 2)]),
                    ("{0,1/2,-1/2,3/2,-3/2}", [0, Fr(1, 2), Fr(-1, 2), Fr(3, 2), Fr(-3, 2)]),
                    ("{0,1,-1,1/2,-3}", [0, 1, -1, Fr(1, 2), -3]),
                    ("{0,1/2,-1/2,1,-2}", [0, Fr(1, 2), Fr(-1, 2), 1, -2]),
                    ("{0,3/4,-3/4,3/2,-3/2}", [0, Fr(3, 4), Fr(-3, 4), Fr(3, 2), Fr(-3, 2)])):
        AT, G, BT = cook_toom(p, 4)
        assert check(AT, G, BT, 4) < 1e-9, name
        for bm in (0.0, 0.5):
            e44 = emulate(4, 4, p, p, bias_mean=bm)
            e24 = emulate(2, 4, std2, p, bias_mean=bm)
            print(f"{name:26s} mean-shift {bm}: F(4x4) mean {e44[0]:.2e} max {e44[1]:.2e} | F(2x4) mean {e24[0]:.2e} max {e24[1]:.2e}")
    for bm in (0.0, 0.5):
        e22 = emulate(2, 2, std2, std2, bias_mean=bm)
        print(f"F(2x2) std mean-shift {bm}: mean {e22[0]:.2e} max {e22[1]:.2e}")
