"""Host logic without a GPU: Keras-config normalisation, pack round trip, C-ABI surface."""
import json
import os
import re

import numpy as np
import pytest

from timed_hip import keras_config as kc
from timed_hip import pack, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_timed_synth_flops_match_survey():
    for n_cls, want in ((20, 628.2e6), (338, 1177.7e6)):
        cfg, w = synth.timed_synth(n_cls)
        layers = kc.parse_keras_model(cfg, w)
        assert abs(kc.flops_per_frame(layers) - want) / want < 1e-3
    cfg, w = synth.densecpd_synth(20)
    assert abs(kc.flops_per_frame(kc.parse_keras_model(cfg, w)) - 391.0e6) / 391.0e6 < 2e-3


def test_parse_drops_dropout_and_orders_topologically():
    cfg, w = synth.timed_synth(20)
    layers = kc.parse_keras_model(cfg, w)
    names = [l.name for l in layers]
    assert not any("dropout" in n for n in names)
    idx = {n: i for i, n in enumerate(names)}
    for l in layers:
        assert all(idx[i] < idx[l.name] for i in l.inputs)
    assert layers[-1].op == kc.OP_ACT and layers[-1].ip["act"] == kc.ACT_SOFTMAX
    assert layers[-1].out_shape == (20,)


def test_pack_roundtrip():
    cfg, w = synth.densecpd_synth(20)
    layers = kc.parse_keras_model(cfg, w)
    recs = pack.read_pack(pack.layers_to_pack(layers))
    assert len(recs) == len(layers)
    for l, r in zip(layers, recs):
        assert r["op"] == l.op and r["name"] == l.name[:55] and r["out_shape"] == tuple(l.out_shape)
        for j, key in enumerate(pack.W_SLOTS.get(l.op, [])):
            if key in l.weights:
                assert np.array_equal(r["weights"][j], l.weights[key].ravel())
            else:
                assert r["weights"][j] is None


def test_sequential_model_config():
    cfg, w = synth.timed_synth(20, widths=(4,), side=5, in_channels=2)
    seq = dict(class_name="Sequential",
               config=dict(name="seq", layers=[dict(class_name=l["class_name"], config=l["config"]) for l in cfg["config"]["layers"]]))
    a = kc.parse_keras_model(cfg, w)
    b = kc.parse_keras_model(seq, w)
    assert [x.op for x in a] == [x.op for x in b] and [x.out_shape for x in a] == [x.out_shape for x in b]


def test_unsupported_layer_is_loud():
    cfg, w = synth.timed_synth(20, widths=(4,), side=5, in_channels=2)
    cfg["config"]["layers"][1]["class_name"] = "Conv3DTranspose"
    with pytest.raises(kc.UnsupportedLayer):
        kc.parse_keras_model(cfg, w)


def test_abi_exports_every_declared_symbol(lib):
    header = open(os.path.join(ROOT, "include", "timed_hip.h")).read()
    declared = set(re.findall(r"\b(th_[a-z0-9_]+)\s*\(", header))
    declared -= {"th_model", "th_comm"}
    from timed_hip import _lib
    assert declared == set(_lib.PROTOTYPES), (declared ^ set(_lib.PROTOTYPES))
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.th_version() == 1


def test_no_gpu_is_an_error_not_a_fallback(lib):
    """Without a device the product must fail loudly (there is no CPU path)."""
    from timed_hip import _lib, engine
    if _lib.device_count() > 0:
        pytest.skip("a GPU is present")
    cfg, w = synth.timed_synth(20, widths=(4,), side=5, in_channels=2)
    with pytest.raises(_lib.TimedHipError):
        engine.HipFrameModel.from_keras(cfg, w)


def test_per_channel_ops_and_pooling_are_pushed_through_branch_concats():
    """keras_config.push_through_concat: BN / activation / pooling that is the only consumer of a Concatenate of
    convolution branches is applied per branch with the BatchNorm vectors sliced (exact rewrite, lets the engine
    fuse them into each branch's convolution); DenseNet-style concatenations are left alone."""
    from timed_hip import keras_config as kc, synth
    cfg, w = synth.TOPOLOGIES["prodconn"]()
    layers = kc.parse_keras_model(cfg, w)
    names = [l.name for l in layers]
    assert "batch_normalization__b0" in names and "batch_normalization__b1" in names
    assert "max_pooling3d__b0" in names and "max_pooling3d__b1" in names
    by = {l.name: l for l in layers}
    bn_all = [np.asarray(a, np.float32) for a in w["batch_normalization"]]          # gamma, beta, mean, var over 32 channels
    for k, (lo, hi) in enumerate(((0, 16), (16, 32))):
        b = by[f"batch_normalization__b{k}"]
        assert b.ip["c"] == 16 and b.out_shape[-1] == 16
        for key, full in zip(("gamma", "beta", "mean", "var"), bn_all):
            assert np.array_equal(b.weights[key], full[lo:hi])
    merged = by["max_pooling3d"]                       # the concat now carries the name downstream layers refer to
    assert merged.op == kc.OP_CONCAT and merged.inputs == ["max_pooling3d__b0", "max_pooling3d__b1"]
    assert merged.out_shape == (10, 10, 10, 32)
    assert not any(l.op == kc.OP_CONCAT and l.name == "concatenate" for l in layers)
    # DenseCPD: every concat has several consumers or a concat branch -> untouched
    cfg, w = synth.TOPOLOGIES["densecpd"]()
    layers = kc.parse_keras_model(cfg, w)
    assert not any("__b" in l.name for l in layers)
    assert sum(l.op == kc.OP_CONCAT for l in layers) == 12


# ---- Keras 2.13-style serialisation details the synthetic builder does not emit -------------------
def _k213(class_name, name, inbound, **cfg):
    """a layer record as tf.keras 2.13 writes it into a legacy .h5 model_config"""
    base = dict(name=name, trainable=True, dtype="float32")
    base.update(cfg)
    return {"class_name": class_name, "config": base, "name": name, "inbound_nodes": [[[n, 0, 0, {}] for n in inbound]] if inbound else []}


def _keras213_config(policy="float32"):
    init = {"class_name": "GlorotUniform", "config": {"seed": None}}
    zeros = {"class_name": "Zeros", "config": {}}
    conv_common = dict(strides=[1, 1, 1], padding="same", data_format="channels_last", dilation_rate=[1, 1, 1], groups=1, use_bias=True,
                       kernel_initializer=init, bias_initializer=zeros, kernel_regularizer=None, bias_regularizer=None,
                       activity_regularizer=None, kernel_constraint=None, bias_constraint=None)
    layers = [
        {"class_name": "InputLayer", "config": {"batch_input_shape": [None, 5, 5, 5, 2], "dtype": "float32", "sparse": False,
                                                  "ragged": False, "name": "input_1"}, "name": "input_1", "inbound_nodes": []},
        _k213("Conv3D", "conv3d", ["input_1"], filters=4, kernel_size=[3, 3, 3], activation="linear",
              **dict(conv_common, dtype={"class_name": "Policy", "config": {"name": policy}})),
        _k213("ELU", "elu", ["conv3d"], alpha=1.0),
        _k213("BatchNormalization", "batch_normalization", ["elu"], axis=[4], momentum=0.99, epsilon=0.001, center=True, scale=True,
              beta_initializer=zeros, gamma_initializer={"class_name": "Ones", "config": {}},
              moving_mean_initializer=zeros, moving_variance_initializer={"class_name": "Ones", "config": {}},
              beta_regularizer=None, gamma_regularizer=None, beta_constraint=None, gamma_constraint=None),
        _k213("SpatialDropout3D", "spatial_dropout3d", ["batch_normalization"], rate=0.2, noise_shape=None, seed=None),
        # activation given as a serialized layer object (Conv3D(..., activation=tf.keras.layers.LeakyReLU(0.1)))
        _k213("Conv3D", "conv3d_1", ["spatial_dropout3d"], filters=3, kernel_size=[1, 1, 1],
              activation={"class_name": "LeakyReLU", "config": {"name": "leaky_re_lu", "trainable": True, "dtype": "float32", "alpha": 0.1}},
              **conv_common),
        _k213("Activation", "activation", ["conv3d_1"], activation="relu"),
        _k213("GlobalAveragePooling3D", "global_average_pooling3d", ["activation"], data_format="channels_last", keepdims=False),
        _k213("Softmax", "softmax", ["global_average_pooling3d"], axis=-1),
    ]
    return {"class_name": "Functional",
            "config": {"name": "model", "trainable": True, "layers": layers, "input_layers": [["input_1", 0, 0]],
                       "output_layers": [["softmax", 0, 0]]},
            "keras_version": "2.13.1", "backend": "tensorflow"}


def _keras213_weights(rng):
    return {"conv3d": [rng.normal(size=(3, 3, 3, 2, 4)).astype(np.float32), rng.normal(size=4).astype(np.float32)],
            "batch_normalization": [rng.uniform(0.5, 1.5, 4).astype(np.float32), rng.normal(size=4).astype(np.float32),
                                    rng.normal(size=4).astype(np.float32), rng.uniform(0.5, 1.5, 4).astype(np.float32)],
            "conv3d_1": [rng.normal(size=(1, 1, 1, 4, 3)).astype(np.float32), rng.normal(size=3).astype(np.float32)]}


def test_keras_2_13_style_config_roundtrips():
    """dtype policy dicts, groups: 1, keras_version/backend keys, initializer/regularizer objects, an activation given as
    a serialized layer, list-valued BN axis: parsed, packed and read back; the oracle (which walks the same JSON on its own)
    evaluates it too"""
    from oracle import cnn_oracle
    cfg = _keras213_config()
    w = _keras213_weights(np.random.default_rng(0))
    layers = kc.parse_keras_model(json.dumps(cfg), w)
    assert [l.name for l in layers] == ["input_1", "conv3d", "elu", "batch_normalization", "conv3d_1", "activation",
                                        "global_average_pooling3d", "softmax"]
    c1 = next(l for l in layers if l.name == "conv3d_1")
    assert c1.ip["act"] == kc.ACT_LEAKY and abs(c1.fp["alpha"] - 0.1) < 1e-7 and c1.inputs == ["batch_normalization"]
    blob = pack.keras_to_pack(cfg, w)
    assert blob[:8] == pack.MAGIC
    probs = cnn_oracle.forward(cfg, w, np.random.default_rng(1).random((2, 5, 5, 5, 2)).astype(np.float32))
    assert probs.shape == (2, 3) and np.allclose(probs.sum(1), 1, atol=1e-6)


@pytest.mark.parametrize("policy", ["mixed_float16", "float16", "bfloat16", "float64"])
def test_non_float32_policy_is_refused(policy):
    with pytest.raises(kc.UnsupportedLayer, match="float32"):
        kc.parse_keras_model(_keras213_config(policy), _keras213_weights(np.random.default_rng(0)))
    cfg = _keras213_config()
    cfg["config"]["layers"][1]["config"]["groups"] = 2
    with pytest.raises(kc.UnsupportedLayer, match="grouped"):
        kc.parse_keras_model(cfg, _keras213_weights(np.random.default_rng(0)))


def test_nested_functional_and_sequential_models_are_inlined():
    """a Model / Sequential used as a layer is spliced into the graph: same ops, same weights as the flat model"""
    flat = _keras213_config()
    w = _keras213_weights(np.random.default_rng(0))
    L = flat["config"]["layers"]
    inner_in = {"class_name": "InputLayer", "config": {"batch_input_shape": [None, 5, 5, 5, 2], "dtype": "float32", "name": "input_2"},
                "name": "input_2", "inbound_nodes": []}
    body = [dict(l) for l in L[1:5]]                                   # conv3d, elu, batch_normalization, spatial_dropout3d
    body[0] = dict(body[0], inbound_nodes=[[["input_2", 0, 0, {}]]])
    nested = {"class_name": "Functional", "name": "trunk", "inbound_nodes": [[["input_1", 0, 0, {}]]],
              "config": {"name": "trunk", "trainable": True, "layers": [inner_in] + body, "input_layers": [["input_2", 0, 0]],
                         "output_layers": [["spatial_dropout3d", 0, 0]]}}
    head = {"class_name": "Sequential", "name": "head", "inbound_nodes": [[["trunk", 0, 0, {}]]],
            "config": {"name": "head", "layers": [dict(l, inbound_nodes=[]) for l in L[5:8]]}}
    sm = dict(L[8], inbound_nodes=[[["head", 0, 0, {}]]])
    outer = {"class_name": "Functional", "keras_version": "2.13.1", "backend": "tensorflow",
             "config": {"name": "outer", "layers": [L[0], nested, head, sm], "input_layers": [["input_1", 0, 0]],
                        "output_layers": [["softmax", 0, 0]]}}
    # weights as timed_hip.h5model hands them out for nested groups: under the inner layer names
    a = kc.parse_keras_model(flat, w)
    b = kc.parse_keras_model(outer, w)
    assert [l.op for l in a] == [l.op for l in b] and [l.out_shape for l in a] == [l.out_shape for l in b]
    assert [l.name for l in b] == ["input_1", "trunk/conv3d", "trunk/elu", "trunk/batch_normalization", "head/conv3d_1", "head/activation",
                                   "head/global_average_pooling3d", "softmax"]
    for la, lb in zip(a, b):
        assert la.ip == lb.ip and la.fp == lb.fp and set(la.weights) == set(lb.weights)
        for k in la.weights:
            assert np.array_equal(la.weights[k], lb.weights[k])
    assert b[4].inputs == ["trunk/batch_normalization"] and b[-1].inputs == ["head/global_average_pooling3d"]
    assert pack.keras_to_pack(outer, w)[:8] == pack.MAGIC


def test_keras_pinning_tool_dry_runs_without_tensorflow():
    """tools/validate_against_keras.py (the hook that pins the CNN oracle to TensorFlow the day it is available): --help and
    the TensorFlow-free part of `--synth NAME --emit-fixture` work in this image"""
    import subprocess
    import sys
    tool = os.path.join(ROOT, "tools", "validate_against_keras.py")
    r = subprocess.run([sys.executable, tool, "--help"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "--synth" in r.stdout and "--emit-fixture" in r.stdout
    r = subprocess.run([sys.executable, tool, "--synth", "densecpd", "--dry-run", "--emit-fixture"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-1000:]
    assert "391.0 MFLOP/frame" in r.stdout and "keras_real_" in r.stdout
    r = subprocess.run([sys.executable, tool, "--dry-run"], capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "exactly one of" in r.stderr


def test_nested_model_weight_aliases_never_shadow_a_top_level_layer(tmp_path):
    """timed_hip.h5model: a nested model's layers are exposed as "<outer>/<inner>"; the bare "<inner>" alias only where it is
    unambiguous — a top-level layer of the same name keeps its own 2 arrays (they once grew to 4), and two nested models that both
    hold a "conv3d" get no bare alias at all"""
    from timed_hip import h5model, h5write
    rng = np.random.default_rng(3)
    arr = lambda *s: rng.normal(size=s).astype(np.float32)
    p = tmp_path / "nested.h5"
    with h5write.File(p) as f:
        f.attrs["model_config"] = json.dumps({"class_name": "Functional", "config": {"layers": []}})
        g = f.create_group("model_weights")
        g.attrs["layer_names"] = ["conv3d", "trunk", "head"]
        top = g.create_group("conv3d")
        top.attrs["weight_names"] = ["conv3d/kernel:0", "conv3d/bias:0"]
        tc = top.create_group("conv3d")
        tc.create_dataset("kernel:0", arr(3, 3, 3, 2, 4)); tc.create_dataset("bias:0", arr(4))
        for outer, inner in (("trunk", ["conv3d", "dense"]), ("head", ["conv3d", "batch_normalization"])):
            og = g.create_group(outer)
            names = []
            for lname in inner:
                lg = og.create_group(lname)
                for wn, shape in (("kernel:0", (1, 1, 1, 4, 4)), ("bias:0", (4,))):
                    lg.create_dataset(wn, arr(*shape))
                    names.append(f"{lname}/{wn}")
            og.attrs["weight_names"] = names
    _cfg, w = h5model.read_keras_h5(str(p))
    assert len(w["conv3d"]) == 2 and w["conv3d"][0].shape == (3, 3, 3, 2, 4)          # the top-level layer, untouched
    assert len(w["trunk/conv3d"]) == 2 and len(w["head/conv3d"]) == 2 and w["trunk/conv3d"][0].shape == (1, 1, 1, 4, 4)
    assert len(w["dense"]) == 2 and len(w["batch_normalization"]) == 2                 # unambiguous inner names keep their alias
    assert not np.array_equal(w["trunk/conv3d"][0], w["head/conv3d"][0])
