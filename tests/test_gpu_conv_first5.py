"""ProDCoNN's 5x5x5 stem on the bf16 pipe (csrc/conv_first5.hip: direct form, input split exactly into three bf16 pieces once at
staging, six piece products on v_mfma_f32_16x16x32_bf16, 2^3 max-pool in registers) against the CPU oracle: 5 / 6 / 3 input
channels, 16 / 9 / 1 filters, float32 / float64 / uint8 frames, a monotone chain (pool first) and one that is not, bias, several
frames per persistent workgroup (the plane ring carries on across units, halves and frames), ragged chunks bit-identical, and the
fp32 kernel it replaces (TH_FIRST_SPLIT=0).  Serves reference predict.py:142 (north_star names ProDCoNN)."""
import numpy as np
import pytest

from oracle import cnn_oracle
from timed_hip import engine, synth

pytestmark = pytest.mark.gpu


def _net(cin, cout, chain, seed, bias=True):
    b = synth.KerasGraphBuilder((21, 21, 21, cin), seed=seed, bias_std=0.3 if bias else 0.0)
    x = b.conv3d(b.input_name, cout, 5, padding="same", use_bias=bias, activation="relu" if chain == "relu_bn" else None)
    if chain == "relu_bn":
        x = b.batchnorm(x)
    elif chain == "tanh":
        x = b.activation(x, "tanh")
    x = b.maxpool(x, 2)
    return b.finish(b.flatten(x))


def _frames(n, cin, seed):
    rng = np.random.default_rng(seed)
    return (rng.standard_normal((n, 21, 21, 21, cin)) * (rng.random((n, 21, 21, 21, cin)) < 0.4)).astype(np.float32)


def _run(cfg, w, x, chunk=None):
    m = engine.HipFrameModel.from_keras(cfg, w)
    if chunk:
        m.set_chunk(chunk)
    got = m.predict(x)
    labels = [s["label"] for s in m.steps()]
    m.close()
    return got, labels


@pytest.mark.parametrize("cin,cout,chain,n", [(6, 16, "relu_bn", 3), (5, 9, "tanh", 2), (3, 1, "none", 1), (6, 16, "none", 5)])
def test_stem_per_element(gpu, monkeypatch, cin, cout, chain, n):
    cfg, w = _net(cin, cout, chain, seed=cin * 17 + cout)
    x = _frames(n, cin, n)
    want = cnn_oracle.forward(cfg, w, x, np.float64)
    scale = max(1.0, float(np.abs(want).max()))
    got, labels = _run(cfg, w, x)
    assert any("k_conv_first5" in l and "bf16x3" in l for l in labels), labels
    assert got.shape == want.shape and float(np.abs(got - want).max()) <= 2e-5 * scale
    got2, _ = _run(cfg, w, x, chunk=2)
    assert np.array_equal(got, got2)
    monkeypatch.setenv("TH_WF_RESIDENT", "2")                       # two workgroups: several frames each, back to back
    got3, _ = _run(cfg, w, x)
    assert np.array_equal(got, got3)
    monkeypatch.delenv("TH_WF_RESIDENT")
    for dt in (np.float64,):
        g64, _ = _run(cfg, w, x.astype(dt))
        assert np.array_equal(g64, got)
    monkeypatch.setenv("TH_FIRST_SPLIT", "0")
    ref, rl = _run(cfg, w, x)
    assert not any("k_conv_first5" in l for l in rl), rl
    assert float(np.abs(got - ref).max()) <= 4e-6 * scale
    # against float64 the split form is in the fp32 kernel's class (K = 125 taps x Cin products per output; measured 1-2.4x)
    assert float(np.abs(got - want).max()) <= 3.0 * float(np.abs(ref - want).max()) + 1e-6 * scale


def test_stem_on_uint8_frames_and_inside_prodconn(gpu):
    cfg, w = _net(6, 16, "relu_bn", seed=5)
    u8 = (np.random.default_rng(1).integers(0, 256, (3, 21, 21, 21, 6)) * (np.random.default_rng(2).random((3, 21, 21, 21, 6)) < 0.2)).astype(np.uint8)
    a, _ = _run(cfg, w, u8)
    b, _ = _run(cfg, w, u8.astype(np.float32))
    assert np.array_equal(a, b)
    cfg, w = synth.prodconn_synth(20)
    x = synth.synthetic_frames(4, seed=3)
    want = cnn_oracle.forward(cfg, w, x, np.float64)
    got, labels = _run(cfg, w, x)
    assert any("k_conv_first5" in l for l in labels) and any("k_conv_first_b3" in l for l in labels), labels
    assert float(np.abs(got - want).max()) <= 5e-6
