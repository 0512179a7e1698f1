"""Error behaviour of the C ABI on a real device: every misuse returns an error code + message (raised
as TimedHipError / ValueError by the Python wrappers); nothing crashes, nothing silently falls back."""
import ctypes as C

import numpy as np
import pytest

from timed_hip import _lib, engine, pack, sampler, synth

pytestmark = pytest.mark.gpu


def test_model_load_rejects_garbage_and_truncation(gpu, lib):
    cfg, w = synth.timed_synth(20, widths=(4,), side=5, in_channels=2)
    good = pack.keras_to_pack(cfg, w)
    for bad in (b"", b"THPK0001", good[:100], good[: len(good) // 2], b"XXXX" + good[4:]):
        with pytest.raises(_lib.TimedHipError) as e:
            engine.HipFrameModel(bad or b"\0")
        assert e.value.code in (-1, -2)
    # corrupt an input index so the graph is no longer topological
    arr = bytearray(good)
    arr[24 + 256 + 8: 24 + 256 + 12] = (99).to_bytes(4, "little")
    with pytest.raises(_lib.TimedHipError):
        engine.HipFrameModel(bytes(arr))
    h = C.c_void_p()
    assert lib.th_model_load(b"/nonexistent/file.pack", gpu, 0, C.byref(h)) == -2
    assert b"cannot open" in lib.th_last_error()
    assert lib.th_model_load(None, gpu, 0, C.byref(h)) == -1


def test_predict_argument_checks(gpu, lib):
    cfg, w = synth.timed_synth(20, widths=(4,), side=5, in_channels=2)
    m = engine.HipFrameModel.from_keras(cfg, w)
    assert m.predict(np.zeros((0, 5, 5, 5, 2), np.float32)).shape == (0, 20)      # empty batch is fine
    x = np.zeros((1, 5, 5, 5, 2), np.float32)
    out = np.zeros((1, 20), np.float32)
    assert lib.th_predict(m._h, x.ctypes.data, 99, 1, out.ctypes.data, 0) == -1    # unknown dtype
    assert lib.th_predict(m._h, None, 0, 1, out.ctypes.data, 0) == -1              # null frames
    assert lib.th_predict(None, x.ctypes.data, 0, 1, out.ctypes.data, 0) == -1     # null model
    assert lib.th_predict(m._h, x.ctypes.data, 0, -3, out.ctypes.data, 0) == -1    # negative count
    assert lib.th_model_set_chunk(m._h, 0) == -1
    with pytest.raises(ValueError):
        m.predict(np.zeros((1, 5, 5, 5), np.float32))
    with pytest.raises(_lib.TimedHipError):
        m.fetch("no_such_layer", 1, (1,))
    m.predict(x)
    with pytest.raises(_lib.TimedHipError):
        m.fetch("conv3d", 1, (5, 5, 5, 4))      # fused away without TH_LOAD_KEEP_ALL
    # a model that does not end in Softmax has no logits to return
    b = synth.KerasGraphBuilder((5, 5, 5, 2))
    cfg2, w2 = b.finish(b.flatten(b.conv3d(b.input_name, 3, 3)))
    m2 = engine.HipFrameModel.from_keras(cfg2, w2)
    with pytest.raises(_lib.TimedHipError):
        m2.predict(x, logits=True)


def test_unsupported_layers_fail_at_conversion_or_load(gpu):
    from timed_hip import keras_config as kc
    cfg, w = synth.timed_synth(20, widths=(4,), side=5, in_channels=2)
    cfg["config"]["layers"][1]["config"]["groups"] = 2
    with pytest.raises(kc.UnsupportedLayer):
        engine.HipFrameModel.from_keras(cfg, w)
    cfg, w = synth.timed_synth(20, widths=(4,), side=5, in_channels=2)
    w["conv3d"][0] = w["conv3d"][0][..., :-1]      # kernel with the wrong filter count
    with pytest.raises(ValueError):
        engine.HipFrameModel.from_keras(cfg, w)


def test_sampler_argument_checks(gpu, lib):
    p = np.full((3, 20), 0.05)
    with pytest.raises(ZeroDivisionError):                             # what the reference raises: 1 / t (sampling_utils.py:159)
        sampler.sample_indices(p, 2, temperature=0.0, rng="philox")
    idx0 = np.zeros((2, 3), np.int32)
    assert lib.th_sample(p.ctypes.data, 3, 20, 2, 0.0, 1, 0, None, idx0.ctypes.data) == -1   # the C ABI: TH_EINVAL
    assert b"temperature 0" in lib.th_last_error()
    with pytest.raises(ValueError):
        sampler.sample_indices(p, 2, rng="host")                        # no uniforms
    with pytest.raises(ValueError):
        sampler.sample_indices(p, 2, uniforms=np.zeros((3, 3)))         # wrong shape
    with pytest.raises(ValueError):
        sampler.sample_indices(p, 2, rng="philox", letters="ABC")       # letters/categories mismatch
    with pytest.raises(_lib.TimedHipError):
        sampler.sample_indices(p, 2, rng="mt19937", seed=2 ** 40)       # legacy seeding takes 32 bits
    idx = np.zeros((2, 3), np.int32)
    assert lib.th_sample(p.ctypes.data, 3, 20, 2, 1.0, 7, 0, None, idx.ctypes.data) == -1   # unknown rng mode
    assert sampler.sample_indices(p, 0, rng="philox").shape == (0, 3)


def test_two_models_and_reload_do_not_interfere(gpu):
    a_cfg, a_w = synth.timed_synth(20, widths=(8, 8), side=7, in_channels=3, seed=1)
    b_cfg, b_w = synth.timed_synth(338, widths=(8,), side=7, in_channels=3, seed=2)
    x = synth.synthetic_frames(5, side=7, channels=3, atoms=20, seed=3)
    a, b = engine.HipFrameModel.from_keras(a_cfg, a_w), engine.HipFrameModel.from_keras(b_cfg, b_w)
    pa, pb = a.predict(x), b.predict(x)
    for _ in range(3):
        assert np.array_equal(b.predict(x), pb) and np.array_equal(a.predict(x), pa)
    a.close()
    assert np.array_equal(engine.HipFrameModel.from_keras(a_cfg, a_w).predict(x), pa)
    assert pb.shape == (5, 338)
