"""Dense behind Flatten as one batch GEMM on the fp32 matrix pipe (csrc/dense_gemm.hip: 16 frames x all outputs per workgroup,
the features in four quarters, v_mfma_f32_16x16x4_f32) against the CPU oracle and against the kernel it replaces (k_dense,
TH_DENSE_GEMM=0): output counts that are not multiples of 16, feature counts that are not multiples of 16 or 64 (ragged quarters, a last block partly beyond the end), batches that are not multiples of 16 frames, fused activations behind it, and the
layers it does not take.  Serves reference predict.py:142 (ProDCoNN's Flatten -> Dense(relu) -> Dense(softmax))."""
import numpy as np
import pytest

from oracle import cnn_oracle
from timed_hip import engine, synth

pytestmark = pytest.mark.gpu


def _net(shape, outs, act, seed):
    b = synth.KerasGraphBuilder(shape, seed=seed, bias_std=0.3)
    x = b.flatten(b.input_name)
    for i, o in enumerate(outs):
        x = b.dense(x, o, activation=act if i + 1 < len(outs) else "linear")
    return b.finish(x)


def _run(cfg, w, x, chunk=None):
    m = engine.HipFrameModel.from_keras(cfg, w)
    if chunk:
        m.set_chunk(chunk)
    got = m.predict(x)
    labels = [s["label"] for s in m.steps()]
    m.close()
    return got, labels


# (input shape, Dense widths, activation, frames)
CASES = [
    ((3, 3, 3, 64), (96, 20), "relu", 37),          # ProDCoNN's head: 1728 -> 96 (-> 20 on k_dense)
    ((5, 5, 5, 8), (33, 20), "elu", 16),            # 1000 features (62.5 blocks of 16), 33 outputs (three tiles, one column in the last)
    ((4, 4, 4, 8), (128, 8), "tanh", 5),            # 512 features (8 blocks per quarter), 128 outputs
    ((1, 4, 8, 17), (100, 338), "relu", 49),        # 544 features (34 blocks: ragged quarters); the 338-way layer stays on k_dense
]


@pytest.mark.parametrize("shape,outs,act,n", CASES)
def test_dense_gemm_per_element(gpu, monkeypatch, shape, outs, act, n):
    cfg, w = _net(shape, outs, act, seed=sum(shape) + outs[0])
    x = np.random.default_rng(n).standard_normal((n, *shape)).astype(np.float32)
    want = cnn_oracle.forward(cfg, w, x, np.float64)
    got, labels = _run(cfg, w, x)
    assert any("k_dense_gemm" in l for l in labels), labels
    scale = max(1.0, float(np.abs(want).max()))
    assert got.shape == want.shape and float(np.abs(got - want).max()) <= 5e-6 * scale
    got2, _ = _run(cfg, w, x, chunk=7)                               # ragged chunks: a frame's row does not depend on its tile
    assert np.array_equal(got, got2)
    monkeypatch.setenv("TH_DENSE_GEMM", "0")
    ref, rl = _run(cfg, w, x)
    assert not any("k_dense_gemm" in l for l in rl), rl
    assert float(np.abs(got - ref).max()) <= 2e-6 * scale
    assert float(np.abs(got - want).max()) <= 1.5 * float(np.abs(ref - want).max()) + 2e-7 * scale


def test_layers_the_gemm_does_not_take(gpu):
    """fewer than 512 features, fewer than 8 or more than 128 outputs, a feature count that is not a multiple of 4: k_dense"""
    for shape, outs in (((1, 1, 4, 64), (32, 20)), ((4, 4, 4, 8), (4, 20)), ((1, 1, 9, 65), (16, 20)), ((4, 4, 4, 8), (200, 20))):
        cfg, w = _net(shape, outs, "relu", seed=3)
        x = np.random.default_rng(1).standard_normal((6, *shape)).astype(np.float32)
        got, labels = _run(cfg, w, x)
        first = [l for l in labels if "dense" in l][0]
        assert "k_dense_gemm" not in first, labels
        want = cnn_oracle.forward(cfg, w, x, np.float64)
        assert float(np.abs(got - want).max()) <= 5e-6 * max(1.0, float(np.abs(want).max()))
