"""The numerics gates of the minimal-filtering and split-operand layers as COLLECTED CPU tests (the sweeps stay in
tests/winograd_numerics.py, a script): the transforms are exact in float64; TIMED-synth with its 5^3 layers as F(3,3)+F(2,3) in-plane
/ z direct and its 10^3 layer as F(2,3)^2 in-plane / z direct — what csrc/conv_wino.hip, conv_wfused.hip and conv_wfsplit.hip compute —
emulated in float32 NumPy inside the oracle stays within 2e-6 of the float64 oracle's logits; and with every product of those layers
replaced by the six bf16 piece products of exactly split operands (fp32 accumulation: k_wino_gemm_b3, k_conv_wfs) the distance does
not grow beyond 1.5x + 2e-7.  Serves reference predict.py:142; uses the CPU oracle (test infrastructure)."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import winograd_numerics as wn  # noqa: E402
from oracle import cnn_oracle  # noqa: E402
from timed_hip import synth  # noqa: E402


def _bf16(x):
    """round to nearest even to bfloat16, returned as float32"""
    u = np.ascontiguousarray(x, np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7fff + ((u >> 16) & 1)) & 0xffff0000
    return u.astype(np.uint32).view(np.float32)


def _split3(x):
    h = _bf16(x)
    m = _bf16(x - h)
    return h, m, _bf16(x - h - m)


def test_three_bf16_pieces_hold_every_float32_exactly():
    rng = np.random.default_rng(0)
    x = (rng.standard_normal(200000) * 10.0 ** rng.integers(-20, 20, 200000)).astype(np.float32)
    h, m, l = _split3(x)
    assert np.array_equal(h.astype(np.float64) + m.astype(np.float64) + l.astype(np.float64), x.astype(np.float64))
    for p in (h, m, l):
        assert np.array_equal(_bf16(p), p)


def _inplane_split(x, kernel, bias, segments):
    """winograd_inplane_conv3d_same with every product as the six kept piece products, accumulated in float32"""
    n = x.shape[1]
    BTc, Gc, ATc = wn.composite(n, segments)
    U = np.einsum("bj,ck,ijkmo->ibcmo", Gc, Gc, kernel.astype(np.float64))
    Uh = _bf16(U.astype(np.float32)); Um = _bf16((U - Uh).astype(np.float32)); Ul = _bf16((U - Uh - Um).astype(np.float32))
    B = BTc[:, 1:n + 1].astype(np.float32)
    A = ATc.astype(np.float32)
    v = np.einsum("bj,nzjkm->nzbkm", B, x.astype(np.float32))
    v = np.einsum("ck,nzbkm->nzbcm", B, v)
    vh, vm, vl = _split3(v)
    pad = [(0, 0), (1, 1), (0, 0), (0, 0), (0, 0)]
    vh, vm, vl = (np.pad(a, pad) for a in (vh, vm, vl))
    m_ = np.zeros(v.shape[:4] + (kernel.shape[-1],), np.float32)
    for dz in range(3):
        for a, u in ((vl, Uh), (vm, Uh), (vh, Uh), (vm, Um), (vh, Um), (vh, Ul)):
            m_ = m_ + np.einsum("nzbcm,bcmo->nzbco", a[:, dz:dz + n], u[dz]).astype(np.float32)
    y = np.einsum("jb,nzbco->nzjco", A, m_)
    y = np.einsum("kc,nzjco->nzjko", A, y)
    if bias is not None:
        y = y + bias.astype(np.float32)
    return y.astype(np.float32)


def test_transforms_are_exact_in_float64():
    rng = np.random.default_rng(0)
    xs, ks = rng.standard_normal((2, 5, 5, 5, 3)), rng.standard_normal((3, 3, 3, 3, 4))
    want = cnn_oracle.conv3d(xs, ks, None, 1, 1, "same", np.float64)
    for seg in ([5], [3, 2], [2, 2, 1]):
        assert np.abs(wn.winograd_conv3d_same(xs, ks, None, seg, np.float64) - want).max() < 1e-9
    assert np.abs(wn.winograd_inplane_conv3d_same(xs, ks, None, [3, 2], np.float64) - want).max() < 1e-9
    x10, k10 = rng.standard_normal((1, 10, 10, 10, 2)), rng.standard_normal((3, 3, 3, 2, 2))
    w10 = cnn_oracle.conv3d(x10, k10, None, 1, 1, "same", np.float64)
    assert np.abs(wn.winograd_inplane_conv3d_same(x10, k10, None, [2] * 5, np.float64) - w10).max() < 1e-9


@pytest.mark.parametrize("classes", [20, 338])
def test_the_default_plan_emulated_on_the_cpu_stays_at_fp32_rounding_of_the_float64_logits(classes):
    cfg, weights = synth.timed_synth(classes)
    frames = synth.synthetic_frames(4, seed=77)
    ref = cnn_oracle.forward(cfg, weights, frames, dtype=np.float64, return_all=True)
    logit = [k for k in ref if "global_average" in k][-1]
    real = cnn_oracle.conv3d
    seg = {5: [3, 2], 10: [2] * 5}

    def run(fn):
        def conv(x, kernel, bias, strides, dilation, padding, acc_dtype):
            n = x.shape[1]
            if (acc_dtype == np.float32 and kernel.shape[:3] == (3, 3, 3) and padding == "same" and n in seg
                    and cnn_oracle._t3(strides) == (1, 1, 1) and x.shape[1:4] == (n, n, n)):
                return fn(x, kernel, bias, seg[n])
            return real(x, kernel, bias, strides, dilation, padding, acc_dtype)
        cnn_oracle.conv3d = conv
        try:
            got = cnn_oracle.forward(cfg, weights, frames, dtype=np.float32, return_all=True)
        finally:
            cnn_oracle.conv3d = real
        return float(np.abs(got[logit] - ref[logit]).max()), got

    e_fp32, _ = run(lambda x, k, b, s: wn.winograd_inplane_conv3d_same(x, k, b, s))
    e_split, got = run(_inplane_split)
    assert e_fp32 <= 2e-6, e_fp32                               # the gate of round 4
    assert e_split <= 1.5 * e_fp32 + 2e-7, (e_split, e_fp32)    # the gate of rounds 5 and 6: the split costs nothing measurable
    last = list(ref)[-1]
    assert np.array_equal(got[last].argmax(1), ref[last].argmax(1))
