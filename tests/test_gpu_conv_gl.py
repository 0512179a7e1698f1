"""Strided and small-volume convolutions as an implicit GEMM with rows across the batch and operands straight from L2
(csrc/conv_gl.hip, v_mfma_f32_16x16x4_f32) against the CPU oracle and the kernels it replaces (TH_CONV_GL=0): stride 2 'same'
(Keras puts the odd padding row behind), 'valid', anisotropic kernels and strides, dilation, 1 .. 128 output channels, row tiles
that straddle frames and a batch whose last tile is ragged, fused bias / activation / BatchNorm.  Serves reference predict.py:142
(ProDCoNN's strided layer and the 'valid' one behind it)."""
import numpy as np
import pytest

from oracle import cnn_oracle
from timed_hip import engine, synth

pytestmark = pytest.mark.gpu


def _run(cfg, w, x, chunk=None):
    m = engine.HipFrameModel.from_keras(cfg, w)
    if chunk:
        m.set_chunk(chunk)
    got = m.predict(x)
    labels = [s["label"] for s in m.steps()]
    m.close()
    return got, labels


def _net(shape, build, seed):
    b = synth.KerasGraphBuilder(shape, seed=seed, bias_std=0.3)
    return b.finish(b.flatten(build(b, b.input_name)))


# (input shape, conv kwargs, chain, frames)
CASES = [
    ((10, 10, 10, 32), dict(filters=48, kernel_size=3, strides=2, padding="same"), "elu", 7),        # ProDCoNN conv3d_2
    ((5, 5, 5, 48), dict(filters=64, kernel_size=3, padding="valid"), "leaky", 9),                     # ProDCoNN conv3d_3: 27 rows per frame
    ((7, 6, 5, 16), dict(filters=1, kernel_size=(1, 3, 5), strides=(1, 2, 2), padding="same"), "none", 5),
    ((9, 9, 9, 16), dict(filters=100, kernel_size=3, strides=3, padding="valid"), "relu_bn", 3),
    ((8, 8, 8, 32), dict(filters=128, kernel_size=2, strides=2, padding="same"), "tanh", 4),
    ((6, 6, 6, 64), dict(filters=20, kernel_size=3, dilation_rate=2, strides=1, padding="same"), "none", 2),
]


def _build(kw, chain):
    def build(b, x):
        kw2 = dict(kw)
        filters, k = kw2.pop("filters"), kw2.pop("kernel_size")
        x = b.conv3d(x, filters, k, activation="relu" if chain == "relu_bn" else None, **kw2)
        if chain == "relu_bn":
            x = b.batchnorm(x)
        elif chain == "elu":
            x = b.activation(x, "elu")
        elif chain == "tanh":
            x = b.activation(x, "tanh")
        elif chain == "leaky":
            x = b.leaky_relu(x, 0.1)
        return x
    return build


@pytest.mark.parametrize("shape,kw,chain,n", CASES)
def test_conv_gl_per_element(gpu, monkeypatch, shape, kw, chain, n):
    monkeypatch.setenv("TH_CONV_GL", "2")                               # every eligible layer (the default takes strided / <= 64-output ones)
    cfg, w = _net(shape, _build(kw, chain), seed=sum(shape) + kw["filters"])
    x = np.random.default_rng(n).standard_normal((n, *shape)).astype(np.float32)
    want = cnn_oracle.forward(cfg, w, x, np.float64)
    got, labels = _run(cfg, w, x)
    assert any("k_conv_gl" in l for l in labels), labels
    scale = max(1.0, float(np.abs(want).max()))
    assert got.shape == want.shape and float(np.abs(got - want).max()) <= 5e-6 * scale
    got2, _ = _run(cfg, w, x, chunk=2)                                  # other tiles over the same rows: bit-identical
    assert np.array_equal(got, got2)
    monkeypatch.setenv("TH_CONV_GL", "0")
    ref, rl = _run(cfg, w, x)
    assert not any("k_conv_gl" in l for l in rl), rl
    assert float(np.abs(got - ref).max()) <= 3e-6 * scale
    assert float(np.abs(got - want).max()) <= 1.5 * float(np.abs(ref - want).max()) + 2e-7 * scale


@pytest.mark.parametrize("act", ["relu", "elu", None])
def test_conv_gl_input_prologue(gpu, monkeypatch, act):
    """BN -> activation in FRONT of the convolution (DenseCPD's growth layers; here on the model input, where no producer can absorb
    it): applied to the loaded voxels in registers, and the 'same' padding pads the ACTIVATED tensor — zeros, not act(shift)"""
    monkeypatch.setenv("TH_CONV_GL", "2")

    def build(b, x):
        x = b.batchnorm(x)
        if act:
            x = b.activation(x, act)
        return b.conv3d(x, 24, 3, strides=2, padding="same")

    cfg, w = _net((6, 5, 4, 32), build, seed=12)
    x = np.random.default_rng(3).standard_normal((5, 6, 5, 4, 32)).astype(np.float32)
    want = cnn_oracle.forward(cfg, w, x, np.float64)
    got, labels = _run(cfg, w, x)
    assert any("k_conv_gl" in l for l in labels), labels
    assert float(np.abs(got - want).max()) <= 5e-6 * max(1.0, float(np.abs(want).max()))
    monkeypatch.setenv("TH_CONV_GL", "0")
    ref, _ = _run(cfg, w, x)
    assert float(np.abs(got - ref).max()) <= 3e-6 * max(1.0, float(np.abs(ref).max()))


def test_default_rule_and_layers_it_leaves_alone(gpu):
    """by default: strided layers and those with at most 64 outputs per frame; a stride-1 'same' layer on 10^3 keeps its kernel, and
    so does one whose Cin is not a multiple of 16"""
    cfg, w = synth.prodconn_synth(20, seed=3)
    m = engine.HipFrameModel.from_keras(cfg, w)
    labels = [s["label"] for s in m.steps()]
    m.close()
    assert sum("k_conv_gl" in l for l in labels) == 2, labels           # conv3d_2 (stride 2) and conv3d_3 (27 outputs per frame)
    for shape, kw in (((10, 10, 10, 32), dict(filters=48, kernel_size=3, padding="same")),
                      ((6, 6, 6, 20), dict(filters=16, kernel_size=3, strides=2, padding="same"))):
        cfg, w = _net(shape, _build(kw, "none"), seed=1)
        x = np.random.default_rng(0).standard_normal((3, *shape)).astype(np.float32)
        got, labels = _run(cfg, w, x)
        assert not any("k_conv_gl" in l for l in labels), labels
        want = cnn_oracle.forward(cfg, w, x, np.float64)
        assert float(np.abs(got - want).max()) <= 2e-5 * max(1.0, float(np.abs(want).max()))


def test_conv_gl_random_geometries(gpu, monkeypatch):
    """thirty seeded draws of (extent, kernel, stride, dilation, padding, Cin in {16, 32, 48}, 1..128 filters, 1..9 frames): every one
    within 5e-6 x scale of the float64 oracle, on the kernel whenever the layer is eligible"""
    monkeypatch.setenv("TH_CONV_GL", "2")
    rng = np.random.default_rng(2024)
    ran = 0
    for case in range(30):
        shape = tuple(int(v) for v in rng.integers(3, 10, 3))
        k = tuple(int(v) for v in rng.integers(1, 5, 3))
        st = tuple(int(v) for v in rng.integers(1, 4, 3))
        dl = tuple(int(v) for v in rng.integers(1, 3, 3)) if case % 3 == 0 else (1, 1, 1)
        if any(d > 1 for d in dl):
            st = (1, 1, 1)                                              # Keras: strides and dilation do not combine
        padding = "same" if case % 2 else "valid"
        if padding == "valid" and any((kk - 1) * dd + 1 > n for kk, dd, n in zip(k, dl, shape)):
            padding = "same"
        cin, cout, n = int(rng.choice([16, 32, 48])), int(rng.integers(1, 129)), int(rng.integers(1, 10))
        kw = dict(filters=cout, kernel_size=k, strides=st, dilation_rate=dl, padding=padding)
        cfg, w = _net((*shape, cin), _build(kw, ("none", "elu", "relu_bn")[case % 3]), seed=case)
        x = rng.standard_normal((n, *shape, cin)).astype(np.float32)
        want = cnn_oracle.forward(cfg, w, x, np.float64)
        got, labels = _run(cfg, w, x)
        ran += any("k_conv_gl" in l for l in labels)
        assert got.shape == want.shape, (case, kw, shape)
        assert float(np.abs(got - want).max()) <= 5e-6 * max(1.0, float(np.abs(want).max())), (case, kw, shape, cin, n, labels)
    assert ran >= 20, ran
