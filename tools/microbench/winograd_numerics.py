#!/usr/bin/env python
"""Go/no-go, part 1 (numerics, CPU only): Winograd / Cook-Toom minimal filtering for TIMED's 3x3x3 'same' convolutions,
emulated in float32 NumPy inside the oracle's forward pass, against the float64 oracle.

    python tools/microbench/winograd_numerics.py [--frames 32] [--classes 20]

A 1-D 'same' convolution over n outputs is cut into segments F(m, 3) (m outputs from m + 2 inputs with m + 2 products); the
3-D transform is the Kronecker cube of the per-dimension composite matrices.  For a 5-wide axis:
    direct            15 products per (ci, co) and axis  ->  3375 per 5^3 volume
    [2, 2, 1]         4 + 4 + 3 = 11                     ->  1331
    [3, 2]            5 + 4 = 9                          ->   729
    [5]               7                                  ->   343
The experiment replaces the convolutions of the chosen layers, keeps everything else (ELU, BatchNorm, pooling, GAP) as the
oracle has it, and reports max |logit - logit_fp64| over the frames, beside the same figure for the direct fp32 oracle."""
from __future__ import annotations

import argparse
import os
import sys
from fractions import Fraction

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "timed-design_amd"))

POINTS = [Fraction(0), Fraction(1), Fraction(-1), Fraction(1, 2), Fraction(-1, 2), Fraction(2), Fraction(-2), Fraction(1, 4), Fraction(-1, 4)]


def _inv(M):
    """exact inverse of a square matrix of Fractions (Gauss-Jordan)"""
    n = len(M)
    A = [list(r) + [Fraction(int(i == j)) for j in range(n)] for i, r in enumerate(M)]
    for c in range(n):
        p = next(r for r in range(c, n) if A[r][c] != 0)
        A[c], A[p] = A[p], A[c]
        d = A[c][c]
        A[c] = [v / d for v in A[c]]
        for r in range(n):
            if r != c and A[r][c] != 0:
                f = A[r][c]
                A[r] = [a - f * b for a, b in zip(A[r], A[c])]
    return [r[n:] for r in A]


def cook_toom(m: int, r: int = 3, points=None):
    """F(m, r): y = AT [(G g) * (BT d)], y[i] = sum_k d[i + k] g[k].  a = m + r - 1 evaluation points, the last one at infinity.
    Transposition of the Toom-Cook linear convolution: AT = Em^T, G = Er, BT = V^-T (E: evaluation, V: a x a Vandermonde)."""
    a = m + r - 1
    pts = list(points or POINTS[: a - 1])
    assert len(pts) == a - 1

    def evalm(cols):
        return [[p ** j for j in range(cols)] for p in pts] + [[Fraction(int(j == cols - 1)) for j in range(cols)]]
    V = evalm(a)
    Vi = _inv(V)
    AT = [[evalm(m)[i][j] for i in range(a)] for j in range(m)]
    G = evalm(r)
    BT = [[Vi[j][i] for j in range(a)] for i in range(a)]
    # balance: row i of BT times s_i, row i of G by 1 / s_i, with s_i the smallest factor that makes the BT row integer: the data
    # transform then multiplies by small integers only (exact products), and the fractions go into the weights, which are
    # transformed once on the host in double precision
    from math import gcd
    for i in range(a):
        den = 1
        for v in BT[i]:
            den = den * v.denominator // gcd(den, v.denominator)
        num = 0
        for v in BT[i]:
            num = gcd(num, abs(int(v * den)))
        sc = Fraction(den, num or 1)
        BT[i] = [v * sc for v in BT[i]]
        G[i] = [v / sc for v in G[i]]
    f = lambda M: np.array([[float(v) for v in row] for row in M], dtype=np.float64)
    return f(AT), f(G), f(BT)


def composite(n: int, segments):
    """per-axis matrices of a 'same' 3-tap convolution over n outputs cut into F(m, 3) segments:
    BTc [P, n + 2] on the zero-padded input, Gc [P, 3], ATc [n, P]"""
    assert sum(segments) == n
    P = sum(m + 2 for m in segments)
    BTc, Gc, ATc = np.zeros((P, n + 2)), np.zeros((P, 3)), np.zeros((n, P))
    o = p = 0
    for m in segments:
        if m == 1:                                   # F(1, 3): three products, nothing to transform
            AT, G, BT = np.ones((1, 3)), np.eye(3), np.eye(3)
        else:
            AT, G, BT = cook_toom(m, 3)
        a = m + 2
        BTc[p:p + a, o:o + a] = BT
        Gc[p:p + a] = G
        ATc[o:o + m, p:p + a] = AT
        o += m
        p += a
    return BTc, Gc, ATc


def winograd_conv3d_same(x, kernel, bias, segments, dtype=np.float32):
    """x [N, n, n, n, Cin] -> [N, n, n, n, Cout], every product and sum in `dtype`"""
    n = x.shape[1]
    BTc, Gc, ATc = composite(n, segments)
    U = np.einsum("ai,bj,ck,ijkmo->abcmo", Gc, Gc, Gc, kernel.astype(np.float64)).astype(dtype)       # host, once per model
    xp = np.pad(x.astype(dtype), [(0, 0), (1, 1), (1, 1), (1, 1), (0, 0)])
    B = BTc.astype(dtype)
    A = ATc.astype(dtype)
    v = np.einsum("ai,nijkm->najkm", B, xp)                  # three separable passes, as a kernel would run them
    v = np.einsum("bj,najkm->nabkm", B, v)
    v = np.einsum("ck,nabkm->nabcm", B, v)
    m_ = np.einsum("nabcm,abcmo->nabco", v, U)               # P^3 GEMMs [N, Cin] x [Cin, Cout]
    y = np.einsum("ia,nabco->nibco", A, m_)
    y = np.einsum("jb,nibco->nijco", A, y)
    y = np.einsum("kc,nijco->nijko", A, y)
    if bias is not None:
        y = y + bias.astype(dtype)
    return y.astype(dtype)


def winograd_inplane_conv3d_same(x, kernel, bias, segments, dtype=np.float32):
    """what csrc/conv_wino.hip computes: Cook-Toom in the two in-plane axes, the three z taps summed directly —
    M[z][a][b] = sum_dz V[z + dz - 1][a][b] . U[dz][a][b] (P^2 positions x 13 valid (z, dz) pairs for n = 5)"""
    n = x.shape[1]
    BTc, Gc, ATc = composite(n, segments)
    U = np.einsum("bj,ck,ijkmo->ibcmo", Gc, Gc, kernel.astype(np.float64)).astype(dtype)              # [dz][a][b][ci][co]
    xr = x.astype(dtype)
    B = BTc[:, 1:n + 1].astype(dtype)
    A = ATc.astype(dtype)
    v = np.einsum("bj,nzjkm->nzbkm", B, xr)
    v = np.einsum("ck,nzbkm->nzbcm", B, v)                   # V[n][z][a][b][ci]
    vp = np.pad(v, [(0, 0), (1, 1), (0, 0), (0, 0), (0, 0)])
    m_ = np.zeros(v.shape[:4] + (kernel.shape[-1],), dtype=dtype)
    for dz in range(3):
        m_ = m_ + np.einsum("nzbcm,bcmo->nzbco", vp[:, dz:dz + n], U[dz]).astype(dtype)
    y = np.einsum("jb,nzbco->nzjco", A, m_)
    y = np.einsum("kc,nzjco->nzjko", A, y)
    if bias is not None:
        y = y + bias.astype(dtype)
    return y.astype(dtype)


def emit_header(path, n=5):
    """csrc/wino_tables.h: the per-axis composite matrices of a 5-wide 'same' axis for the HIP kernels and the host-side weight
    transform — scheme 9 = F(3, 3) + F(2, 3) (9 points, the default: as accurate as the direct form) and scheme 7 = F(5, 3)
    (7 points, the minimum: 1.65x fewer products and bytes, ~4x the rounding error; opt-in)"""
    def arr(name, M, typ):
        rows = ",\n    ".join("{" + ", ".join(repr(float(v)) if typ == "double" else (f"{v:.10g}f" if v != int(v) else f"{int(v)}.f") for v in r) + "}" for r in M)
        dev = "__device__ " if typ == "float" else ""      # the float tables are read by the kernels (folded after unrolling)
        return f"static {dev}constexpr {typ} {name}[{M.shape[0]}][{M.shape[1]}] = {{\n    {rows}}};\n"
    txt = ("// GENERATED by tools/microbench/winograd_numerics.py --emit-header — do not edit.\n"
           f"// Cook-Toom minimal filtering F(m, 3) for a 'same' 3-tap axis of {n} outputs: y = AT [(G g) * (BT d)], d = the axis WITHOUT\n"
           "// its zero halo (the two halo columns of BT multiply zeros and are dropped).  BT is integer (exact products); the fractions\n"
           "// live in G, applied to the weights once on the host in double precision.\n"
           "//   scheme 9: segments [3, 2] = F(3, 3) + F(2, 3), 9 products per axis instead of 15\n"
           "//   scheme 7: segments [5]    = F(5, 3), 7 products per axis (points 0, 1, -1, 1/2, -1/2, 2, inf)\n#pragma once\n"
           f"#define WINO_N {n}\n")
    for segments in ([3, 2], [5]):
        BTc, Gc, ATc = composite(n, list(segments))
        P = BTc.shape[0]
        txt += arr(f"kWinoBT{P}", BTc[:, 1:n + 1], "float") + arr(f"kWinoAT{P}", ATc, "float") + arr(f"kWinoG{P}", Gc, "double")
    open(path, "w").write(txt)
    print("wrote", path)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=32)
    ap.add_argument("--classes", type=int, default=20)
    ap.add_argument("--only", default="", help="substring of the plan names to run")
    ap.add_argument("--emit-header", action="store_true", help="write timed-design_amd/csrc/wino_tables.h and exit")
    args = ap.parse_args()
    if args.emit_header:
        emit_header(os.path.join(ROOT, "timed-design_amd", "csrc", "wino_tables.h"))
        return
    from oracle import cnn_oracle
    from timed_hip import synth

    # self-check of the transforms in float64: equal to the direct convolution
    rng = np.random.default_rng(0)
    xs, ks = rng.standard_normal((2, 5, 5, 5, 3)), rng.standard_normal((3, 3, 3, 3, 4))
    want = cnn_oracle.conv3d(xs, ks, None, 1, 1, "same", np.float64)
    for seg in ([5], [3, 2], [2, 2, 1], [2, 3], [4, 1]):
        got = winograd_conv3d_same(xs, ks, None, seg, np.float64)
        assert np.abs(got - want).max() < 1e-9, (seg, np.abs(got - want).max())
    x10 = rng.standard_normal((1, 10, 10, 10, 2))
    k10 = rng.standard_normal((3, 3, 3, 2, 2))
    w10 = cnn_oracle.conv3d(x10, k10, None, 1, 1, "same", np.float64)
    for seg in ([2] * 5, [4, 4, 2], [3, 3, 2, 2], [5, 5]):
        assert np.abs(winograd_conv3d_same(x10, k10, None, seg, np.float64) - w10).max() < 1e-8, seg
    print("transforms exact in float64: ok")

    cfg, weights = synth.timed_synth(args.classes)
    frames = synth.synthetic_frames(args.frames, seed=77)
    ref64 = cnn_oracle.forward(cfg, weights, frames, dtype=np.float64, return_all=True)
    logit_layer = [k for k in ref64 if "global_average" in k][-1]
    direct = cnn_oracle.forward(cfg, weights, frames, dtype=np.float32, return_all=True)
    scale = np.abs(ref64[logit_layer]).max()
    print(f"logits: max |x| = {scale:.3f}; direct fp32 oracle: max |dlogit| = {np.abs(direct[logit_layer] - ref64[logit_layer]).max():.3e}")

    real_conv = cnn_oracle.conv3d
    got = winograd_inplane_conv3d_same(xs, ks, None, [3, 2], np.float64)
    assert np.abs(got - want).max() < 1e-9
    plans = {
        "5^3 in-plane [3,2], z direct": {5: ("inplane", [3, 2])},
        "5^3 in-plane [5], z direct": {5: ("inplane", [5])},
        "5^3 in-plane [4,1], z direct": {5: ("inplane", [4, 1])},
        "5^3 layers [2,2,1]": {5: [2, 2, 1]},
        "5^3 layers [3,2]": {5: [3, 2]},
        "5^3 layers [5]": {5: [5]},
        "5^3 [3,2] + 10^3 [2]*5": {5: [3, 2], 10: [2] * 5},
        "5^3 [3,2] + 10^3 [3,3,2,2]": {5: [3, 2], 10: [3, 3, 2, 2]},
        "5^3 [3,2] + 10^3 [4,4,2]": {5: [3, 2], 10: [4, 4, 2]},
        "5^3 [5] + 10^3 [5,5]": {5: [5], 10: [5, 5]},
    }
    for name, plan in plans.items():
        if args.only and args.only not in name:
            continue
        def conv(x, kernel, bias, strides, dilation, padding, acc_dtype):
            n = x.shape[1]
            if (acc_dtype == np.float32 and kernel.shape[:3] == (3, 3, 3) and padding == "same" and n in plan
                    and cnn_oracle._t3(strides) == (1, 1, 1) and x.shape[1:4] == (n, n, n)):
                if isinstance(plan[n], tuple):
                    return winograd_inplane_conv3d_same(x, kernel, bias, plan[n][1])
                return winograd_conv3d_same(x, kernel, bias, plan[n])
            return real_conv(x, kernel, bias, strides, dilation, padding, acc_dtype)
        cnn_oracle.conv3d = conv
        try:
            got = cnn_oracle.forward(cfg, weights, frames, dtype=np.float32, return_all=True)
        finally:
            cnn_oracle.conv3d = real_conv
        dl = np.abs(got[logit_layer] - ref64[logit_layer]).max()
        last = list(got)[-1]
        dp = np.abs(got[last] - ref64[last]).max()
        same = np.array_equal(got[last].argmax(1), ref64[last].argmax(1))
        print(f"{name:32s} max |dlogit| = {dl:.3e}  ({dl / scale:.1e} of max |logit|)   max |dp| = {dp:.3e}   argmax equal: {same}")


if __name__ == "__main__":
    main()
