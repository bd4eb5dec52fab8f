#!/usr/bin/env python3
"""Join a rocprofv3 kernel trace of bench.py with the decoder's conv launch schedule: per-launch TFLOP/s.

usage: tools/analyze_trace.py <kernel_trace.csv> [--frames 1024] [--bp 3]
The launch order below mirrors estimator_eval() in unitspeech_amd/csrc/decoder.hip (one conv_igemm launch per convolution,
whichever instantiation executes it)."""
import argparse
import collections
import csv
import sys

ap = argparse.ArgumentParser()
ap.add_argument("trace")
ap.add_argument("--frames", type=int, default=1024)
ap.add_argument("--bp", type=int, default=3)
ap.add_argument("--dim", type=int, default=128)
ap.add_argument("--wino-min-level", type=int, default=1, help="levels >= this run their 3x3 convs as Winograd (US_WINO_MIN_LEVEL)")
ap.add_argument("--wino4", default=__import__("os").environ.get("US_WINO4", "0,44,44,24"),
                help="per-level 4-wide Winograd form of the inference path (US_WINO4): 0 = F(2x2), 44 = F(4x4), 24 = F(2x4)")
ap.add_argument("--wino-narrow", action="store_true", help="US_WINO_NARROW=1: Winograd also where cout <= dim (the last up level)")
ap.add_argument("--no-wtotal", action="store_true", help="US_ATTN_WTOTAL=0: q is computed and stored at every level")
ap.add_argument("--no-split-copy", action="store_true", help="US_SPLIT_COPY=0: res_conv stays the last launch of every ResnetBlock")
a = ap.parse_args()

F, T, BP = 80, a.frames, a.bp
C = [a.dim * m for m in (1, 2, 4, 8)]
L = 4
WMIN = a.wino_min_level


def npx(l):
    return (F >> l) * (T >> l)


seq = []


W4 = [int(v or 0) for v in a.wino4.split(",")] + [0] * 8
# multiplications per output pixel: direct 9; F(2x2) 16 / 4; F(2x4) 24 / 8; F(4x4) 36 / 16 (tile counts rounded up per axis)
WINO_DIV = {0: 2.25, 24: 3.0, 44: 4.0}
WINO_TAG = {0: "[W]", 24: "[W24]", 44: "[W44]"}


def conv(name, l_out_pixels, cin, cout, taps, wino=False, level=0):
    # wino: executed as Winograd F(2x2,3x3) / F(2x4,3x3) / F(4x4,3x3): 4 / 3 / 2.25 multiplications per output pixel instead of 9.
    # GFLOP = what the MFMA units execute for whole tiles (the launch's own count, tile padding included)
    fl = 2.0 * BP * l_out_pixels * cin * cout * taps
    tag = ""
    if wino:
        form = W4[level]
        h, w = F >> level, T >> level
        mh, mw = (form // 10, form % 10) if form else (2, 2)
        tiles = -(-h // mh) * -(-w // mw)
        fl = 2.0 * BP * tiles * (mh + 2) * (mw + 2) * cin * cout
        tag = " " + WINO_TAG[form]
    seq.append((name + tag, fl, l_out_pixels, cin, cout, taps))


def resnet(name, l, cin, cout, first=False):
    wino = l >= WMIN and (cout > a.dim or a.wino_narrow)
    # a block whose successor takes a split copy of its output (direct convolutions: the narrow last up level) runs res_conv FIRST and
    # lets block2's GroupNorm pass add it (decoder.hip, resnet(): split_out)
    res_first = cin != cout and not first and not wino and not a.no_split_copy
    if res_first:
        conv(f"{name}.res 1x1 {cin}->{cout} L{l}", npx(l), cin, cout, 1)
    if not first:
        conv(f"{name}.c1 3x3 {cin}->{cout} L{l}", npx(l), cin, cout, 9, wino=wino, level=l)
    conv(f"{name}.c2 3x3 {cout}->{cout} L{l}", npx(l), cout, cout, 9, wino=wino, level=l)
    if cin != cout and not first and not res_first:
        conv(f"{name}.res 1x1 {cin}->{cout} L{l}", npx(l), cin, cout, 1)


def attn(name, l, c):
    if l == 0 and c <= 128 and not a.no_wtotal:
        # q folded away (decoder.hip, attention(): W_total): to_qkv computes k | v only, the output projection is C x C on x itself
        conv(f"{name}.kv 1x1 {c}->256 L{l}", npx(l), c, 256, 1)
        conv(f"{name}.out 1x1 {c}->{c} L{l} (W_total)", npx(l), c, c, 1)
        return
    conv(f"{name}.qkv 1x1 {c}->384 L{l}", npx(l), c, 384, 1)
    conv(f"{name}.out 1x1 128->{c} L{l}", npx(l), 128, c, 1)


for l in range(L):
    cin = 2 if l == 0 else C[l - 1]
    resnet(f"downs.{l}.r1", l, cin, C[l], first=(l == 0))
    resnet(f"downs.{l}.r2", l, C[l], C[l])
    attn(f"downs.{l}.attn", l, C[l])
    if l < L - 1:
        conv(f"downs.{l}.down 3x3s2 {C[l]} L{l}->L{l+1}", npx(l + 1), C[l], C[l], 9)
resnet("mid1", L - 1, C[-1], C[-1])
attn("mid.attn", L - 1, C[-1])
resnet("mid2", L - 1, C[-1], C[-1])
for u in range(L - 1):
    l = L - 1 - u
    co = C[l - 1]
    resnet(f"ups.{u}.r1", l, 2 * C[l], co)
    resnet(f"ups.{u}.r2", l, co, co)
    attn(f"ups.{u}.attn", l, co)
    conv(f"ups.{u}.up 4x4s2T {co} L{l}->L{l-1} (4 phases, one launch)", npx(l), co, co, 16)
conv("final 3x3 128->128 L0", npx(0), C[0], C[0], 9, wino=0 >= WMIN)

rows = [r for r in csv.DictReader(open(a.trace)) if "conv_igemm" in r["Kernel_Name"]]
n = len(seq)
assert len(rows) % n == 0, (len(rows), n)
evals = len(rows) // n
dur = collections.defaultdict(list)
for i, r in enumerate(rows):
    dur[i % n].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
print(f"{evals} evaluations x {n} conv launches")
tot_t = tot_f = 0.0
print("[W] / [W24] / [W44] = Winograd F(2x2,3x3) / F(2x4,3x3) / F(4x4,3x3): GFLOP is what the MFMA units execute (2*M*N*K over the launch's\n"
      "whole tiles: 4 / 3 / 2.25 multiplications per pixel instead of 9), fp32-equivalent (the f16x3 kernels execute three fp16 products per\n"
      "count); time is the GEMM launch only (input / output transforms are their own kernels)")
print(f"{'launch':48s} {'GFLOP':>8s} {'us':>8s} {'TF/s':>7s} {'share':>6s}")
total_time = sum(sum(v) / len(v) for v in dur.values())
for i, (name, fl, px, cin, cout, taps) in enumerate(seq):
    t = sum(dur[i]) / len(dur[i]) * 1e-9
    tot_t += t
    tot_f += fl
    print(f"{name:48s} {fl/1e9:8.2f} {t*1e6:8.1f} {fl/t/1e12:7.1f} {t*1e9/total_time:6.3f}")
print(f"TOTAL conv: {tot_f/1e9:.1f} GFLOP in {tot_t*1e3:.3f} ms = {tot_f/tot_t/1e12:.1f} TFLOP/s")
