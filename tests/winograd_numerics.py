#!/usr/bin/env python
"""Go/no-go, part 1 (numerics, CPU only; test infrastructure — it uses the CPU oracle): Cook-Toom / Winograd minimal filtering for
TIMED's 3x3x3 'same' convolutions, emulated in float32 NumPy inside the oracle's forward pass, against the float64 oracle.

    python tests/winograd_numerics.py [--frames 32] [--classes 20] [--only SUBSTRING]

A 1-D 'same' convolution over n outputs is cut into segments F(m, 3) (m outputs from m + 2 inputs with m + 2 products); the
3-D transform is the Kronecker cube of the per-dimension composite matrices.  For a 5-wide axis:
    direct            15 products per (ci, co) and axis  ->  3375 per 5^3 volume
    [2, 2, 1]         4 + 4 + 3 = 11                     ->  1331
    [3, 2]            5 + 4 = 9                          ->   729
    [5]               7                                  ->   343
and "in-plane": the two in-plane axes transformed, the three z taps summed directly (what csrc/conv_wino.hip does).  The
experiment replaces the convolutions of the chosen layers, keeps everything else (ELU, BatchNorm, pooling, GAP) as the oracle
has it, and reports max |logit - logit_fp64| over the frames, beside the same figure for the direct fp32 oracle.  The matrices
come from tools/gen_wino_tables.py (the generator of csrc/wino_tables.h).  Not collected by pytest (no test_ prefix)."""
from __future__ import annotations

import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "timed-design_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from gen_wino_tables import composite, cook_toom  # noqa: E402,F401


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=32)
    ap.add_argument("--classes", type=int, default=20)
    ap.add_argument("--only", default="", help="substring of the plan names to run")
    args = ap.parse_args()
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
