"""Host logic without a GPU: Keras-config normalisation, pack round trip, C-ABI surface."""
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
