"""The CNN oracle against the torch-generated golden vectors (tests/golden/make_cnn_golden.py).
TensorFlow is unavailable (parity unpinned vs Keras itself); torch CPU is the independent arbiter."""
import os

import numpy as np
import pytest

from oracle import cnn_oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
from timed_hip import synth


def _case(meta, name):
    return next(m for m in meta if m["name"] == name)


@pytest.mark.parametrize("name", ["timed20", "timed338", "timed20_c5_bias", "timed20_bool", "densecpd20", "prodconn20",
                                  "timed_small", "padding_zoo"])
def test_oracle_matches_torch(cnn_golden, name):
    """padding_zoo is the case whose torch evaluation shares no padding arithmetic with the oracle: torch's own
    padding='same', hand-worked pad literals for stride-2 / even kernels, count_include_pad=False 'same' pooling,
    F.leaky_relu(0.2) (tests/golden/make_cnn_golden.py torch_forward_zoo)."""
    z, meta = cnn_golden
    m = _case(meta, name)
    cfg, weights = getattr(synth, m["builder"])(**m["kwargs"])
    n = m["n"] if name in ("timed_small", "padding_zoo") else 2        # the fp32 oracle needs ~2 s per full-size frame pair
    frames = synth.synthetic_frames(m["n"], **m["frame_kwargs"])[:n]
    vals = cnn_oracle.forward(cfg, weights, frames, np.float32, return_all=True)
    got = vals[cfg["config"]["output_layers"][0][0]]
    z = {k: (z[k][:n] if k.startswith(name + "__") and "__layer__" not in k else z[k]) for k in z.files if k.startswith(name + "__")}
    # logits and intermediate tensors, not only the (near-uniform) probabilities
    if m["logits_layer"]:
        np.testing.assert_allclose(vals[m["logits_layer"]], z[f"{name}__logits64"], atol=5e-6, rtol=0)
    for pn in m["probes"]:
        want = z[f"{name}__layer__{pn}"]
        np.testing.assert_allclose(vals[pn][:1], want, atol=5e-6 * max(1.0, float(np.abs(want).max())), rtol=0)
    assert got.dtype == np.float32 and got.shape == z[f"{name}__torch32"].shape
    np.testing.assert_allclose(got.sum(axis=1), 1.0, atol=1e-5)
    # fp32 oracle vs the fp64 torch result: accumulation-order noise only
    np.testing.assert_allclose(got, z[f"{name}__torch64"], atol=2e-6, rtol=0)
    assert np.array_equal(got.argmax(1), z[f"{name}__torch64"].argmax(1))
    if name in ("timed_small", "padding_zoo"):
        v64 = cnn_oracle.forward(cfg, weights, frames, np.float64, return_all=True)
        np.testing.assert_allclose(v64[cfg["config"]["output_layers"][0][0]], z[f"{name}__torch64"], atol=1e-14, rtol=0)
        np.testing.assert_allclose(v64[m["logits_layer"]], z[f"{name}__logits64"], atol=1e-13, rtol=0)


def test_activation_parameters_follow_keras_defaults():
    """Conv3D(activation='leaky_relu') is slope 0.2 (keras.activations.leaky_relu), a serialized activation
    object carries its own slope, ELU's alpha is 1.0 — never a silent identity."""
    x = np.array([-2.0, 0.5], np.float32)
    assert np.allclose(cnn_oracle._activation(x, "leaky_relu"), [-0.4, 0.5])
    assert np.allclose(cnn_oracle._activation(x, {"class_name": "LeakyReLU", "config": {"negative_slope": 0.1}}), [-0.2, 0.5])
    assert np.allclose(cnn_oracle._activation(x, {"class_name": "LeakyReLU", "config": {"alpha": 0.05}}), [-0.1, 0.5])
    from timed_hip import keras_config as kc
    assert kc._act_code("leaky_relu") == (kc.ACT_LEAKY, 0.2)
    assert kc._act_code({"class_name": "LeakyReLU", "config": {"negative_slope": 0.1}}) == (kc.ACT_LEAKY, 0.1)
    assert kc._act_code("elu") == (kc.ACT_ELU, 1.0) and kc._act_code(None) == (kc.ACT_LINEAR, 1.0)
    with pytest.raises(kc.UnsupportedLayer):
        kc._act_code("LeakyReLU")          # a bare class name carries no slope


def _keras_real_fixtures():
    import glob
    import os
    g = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    return sorted(p for p in glob.glob(os.path.join(g, "keras_real_*.npz")) if os.path.exists(p[:-4] + ".h5"))


@pytest.mark.parametrize("path", _keras_real_fixtures() or [None])
def test_oracle_matches_real_keras_fixture(path):
    """Pins the oracle to TensorFlow itself as soon as someone with TF + a released .h5 has run
    tools/validate_against_keras.py --emit-fixture (not possible in the build image: parity stays 'unpinned' until then)."""
    if path is None:
        pytest.skip("no tests/golden/keras_real_*.npz fixture (needs TensorFlow 2.13 and a released model)")
    from timed_hip import h5model
    z = np.load(path)
    cfg, weights = h5model.read_keras_h5(path[:-4] + ".h5")
    got = cnn_oracle.forward(cfg, weights, z["frames"], np.float32)
    np.testing.assert_allclose(got, z["keras_probs"], atol=1e-4, rtol=0)


def test_oracle_input_dtypes_equivalent():
    cfg, weights = synth.timed_synth(20, widths=(8, 8), side=7, in_channels=3)
    fb = synth.synthetic_frames(2, side=7, channels=3, gaussian=False, atoms=20, seed=2)
    a = cnn_oracle.forward(cfg, weights, fb.astype(bool))
    b = cnn_oracle.forward(cfg, weights, fb.astype(np.float64))
    assert np.array_equal(a, b)


def test_c_direct_convolution_equals_the_numpy_convolution():
    """oracle/conv3d_omp.c (the cpu_baseline configuration of bench.py: blocked direct convolution, OpenMP) against the NumPy
    im2col form it replaces there and against float64: 'same' / 'valid', strides, anisotropic and even kernels, 1x1x1, Cout that is
    not a multiple of the 32-channel register tile, widths that are not a multiple of the 6-voxel tile; then a whole forward pass"""
    import subprocess
    if not cnn_oracle.use_c_conv(2):
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle")], check=True)
        assert cnn_oracle.use_c_conv(2)
    try:
        rng = np.random.default_rng(0)
        cases = [((5, 6, 7), 3, 5, (3, 3, 3), (1, 1, 1), "same"), ((9, 8, 7), 4, 33, (3, 2, 5), (2, 1, 2), "same"),
                 ((7, 7, 7), 6, 16, (3, 3, 3), (1, 1, 1), "valid"), ((6, 5, 4), 2, 70, (1, 1, 1), (1, 1, 1), "same"),
                 ((8, 8, 8), 5, 20, (5, 5, 5), (2, 2, 2), "valid"), ((4, 4, 13), 1, 1, (4, 4, 4), (1, 1, 1), "same")]
        for shape, cin, cout, k, s, pad in cases:
            x = rng.standard_normal((3, *shape, cin)).astype(np.float32)
            w = rng.standard_normal((*k, cin, cout)).astype(np.float32)
            b = rng.standard_normal(cout).astype(np.float32)
            got = cnn_oracle.conv3d(x, w, b, s, (1, 1, 1), pad, np.float32)
            cnn_oracle.use_numpy_conv()
            ref32 = cnn_oracle.conv3d(x, w, b, s, (1, 1, 1), pad, np.float32)
            ref64 = cnn_oracle.conv3d(x.astype(np.float64), w.astype(np.float64), b.astype(np.float64), s, (1, 1, 1), pad, np.float64)
            assert cnn_oracle.use_c_conv(2)
            assert got.shape == ref32.shape and got.dtype == np.float32
            scale = max(1.0, float(np.abs(ref64).max()))
            assert float(np.abs(got - ref64).max()) <= 4e-6 * scale, (shape, k, s, pad)
            # no bias, and a dilated kernel (not the C path's business): both fall through correctly
            assert np.allclose(cnn_oracle.conv3d(x, w, None, s, (1, 1, 1), pad, np.float32), ref32 - b, atol=1e-4 * scale)
        cfg, weights = synth.timed_synth(20)
        frames = synth.synthetic_frames(3, seed=3)
        fast = cnn_oracle.forward(cfg, weights, frames)
        cnn_oracle.use_numpy_conv()
        assert np.abs(fast - cnn_oracle.forward(cfg, weights, frames)).max() < 2e-6
    finally:
        cnn_oracle.use_numpy_conv()
