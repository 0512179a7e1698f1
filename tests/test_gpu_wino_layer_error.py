"""Single-layer error of the direct and the Cook-Toom / split kernels against the float64 oracle on dense standard-normal inputs —
the gate that tests/wino_layer_error.py (a script) only printed, as a collected GPU test: the minimal-filtering layer with its
GEMM on split bf16 operands stays within 5x the direct fp32-MFMA kernel's own distance from float64 (+ 1e-6 of the tensor's scale),
and both within the per-element bound of the other kernel tests.  Serves reference predict.py:142."""
import numpy as np
import pytest

from oracle import cnn_oracle
from timed_hip import engine

import test_gpu_wino as T

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("cin,cout,n", [(256, 338, 5), (128, 256, 8), (64, 128, 8)])
def test_split_winograd_layer_error_against_float64(gpu, monkeypatch, cin, cout, n):
    cfg, w, layer = T._one_layer(cin, cout, seed=cin + cout)
    rng = np.random.default_rng(n)
    frames = (rng.standard_normal((n, 5, 5, 5, cin)) * (rng.random((n, 5, 5, 5, cin)) < 0.5)).astype(np.float32)
    ref = cnn_oracle.forward(cfg, w, frames, np.float64, return_all=True)[layer]
    scale = float(np.abs(ref).max())
    err = {}
    monkeypatch.setenv("TH_NO_TAIL_FUSE", "1")     # the layer's tensor is fetched below: keep it (no pooling output transform)
    for wg in ("0", "1"):
        monkeypatch.setenv("TH_WINOGRAD", wg)
        m = engine.HipFrameModel.from_keras(cfg, w)
        m.predict(frames)
        got = m.fetch(layer, n, (5, 5, 5, cout))
        assert any("conv_wino" in s["label"] for s in m.steps()) == (wg == "1")
        m.close()
        err[wg] = float(np.abs(got - ref).max())
    assert err["0"] <= 2e-5 * scale and err["1"] <= 2e-5 * scale, (err, scale)
    assert err["1"] <= 5.0 * err["0"] + 1e-6 * scale, (err, scale)      # measured: 1.3x, 1.9x, 3.8x (K = 64, 128, 256 channels x 27 taps)
