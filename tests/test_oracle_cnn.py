"""The CNN oracle against the torch-generated golden vectors (tests/golden/make_cnn_golden.py).
TensorFlow is unavailable (parity unpinned vs Keras itself); torch CPU is the independent arbiter."""
import numpy as np
import pytest

from oracle import cnn_oracle
from timed_hip import synth


def _case(meta, name):
    return next(m for m in meta if m["name"] == name)


@pytest.mark.parametrize("name", ["timed20", "timed338", "timed20_c5_bias", "timed20_bool", "densecpd20", "prodconn20",
                                  "timed_small"])
def test_oracle_matches_torch(cnn_golden, name):
    z, meta = cnn_golden
    m = _case(meta, name)
    cfg, weights = getattr(synth, m["builder"])(**m["kwargs"])
    frames = synth.synthetic_frames(m["n"], **m["frame_kwargs"])
    got = cnn_oracle.forward(cfg, weights, frames, np.float32)
    assert got.dtype == np.float32 and got.shape == z[f"{name}__torch32"].shape
    np.testing.assert_allclose(got.sum(axis=1), 1.0, atol=1e-5)
    # fp32 oracle vs the fp64 torch result: accumulation-order noise only
    np.testing.assert_allclose(got, z[f"{name}__torch64"], atol=2e-6, rtol=0)
    assert np.array_equal(got.argmax(1), z[f"{name}__torch64"].argmax(1))
    if name in ("timed_small", "densecpd20"):
        got64 = cnn_oracle.forward(cfg, weights, frames, np.float64)
        np.testing.assert_allclose(got64, z[f"{name}__torch64"], atol=1e-12, rtol=0)


def test_oracle_input_dtypes_equivalent():
    cfg, weights = synth.timed_synth(20, widths=(8, 8), side=7, in_channels=3)
    fb = synth.synthetic_frames(2, side=7, channels=3, gaussian=False, atoms=20, seed=2)
    a = cnn_oracle.forward(cfg, weights, fb.astype(bool))
    b = cnn_oracle.forward(cfg, weights, fb.astype(np.float64))
    assert np.array_equal(a, b)
