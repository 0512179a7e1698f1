"""Cook-Toom / Winograd path of the 3x3x3 'same' convolutions on 5^3 volumes (csrc/conv_wino.hip; the default, TH_WINOGRAD=0
switches it off) against the
float64 oracle and the torch-fp64 fixtures, through the C ABI.  Same bounds as the direct kernels (tests/test_gpu_cnn.py): the
transforms are integer / power-of-two matrices and the products run on the fp32 matrix pipe, so the error is accumulation-order
noise (tests/winograd_numerics.py: 6.8e-7 on the logits of TIMED-synth against 7.7e-7 for the direct form)."""
import numpy as np
import pytest

from oracle import cnn_oracle
from timed_hip import _lib, engine, synth

pytestmark = pytest.mark.gpu
TIGHT = 5e-6
# single layers on DENSE standard-normal inputs (sums of up to 27 x 256 products of magnitude ~1: values around 50 before the
# BatchNorm): the direct kernel's own worst element is 1.0e-5 x max|y| there (a serial fp32 chain over K = 6912), the Winograd
# path's 3.3e-5 x (rms 1.6e-6 against 5.9e-7; tests/wino_layer_error.py) — both accumulation noise at that scale
LAYER = 1e-5


def _one_layer(cin, cout, seed, pre=False, bias=True):
    b = synth.KerasGraphBuilder((5, 5, 5, cin), seed=seed)
    x = b.input_name
    if pre:                                   # DenseNet-style pre-activation: BN -> ReLU -> Conv
        x = b.batchnorm(x)
        x = b.relu(x)
    x = b.conv3d(x, cout, 3, padding="same", use_bias=bias)
    x = b.elu(x)
    x = b.batchnorm(x)
    name = x
    x = b.gap(x)
    x = b.softmax(x)
    cfg, w = b.finish(x)
    if bias:                                  # the builder's biases are zero: make them count
        rng = np.random.default_rng(seed + 1)
        for k, arrs in w.items():
            if k.startswith("conv3d") and len(arrs) == 2:
                arrs[1] = rng.normal(0, 0.2, arrs[1].shape).astype(np.float32)
    return cfg, w, name


@pytest.mark.parametrize("cin,cout,n,pre", [(32, 64, 1, False), (64, 128, 3, False), (128, 128, 64, False), (128, 256, 65, False),
                                            (256, 338, 5, False), (32, 96, 130, True), (64, 64, 7, True),
                                            (256, 20, 37, False), (128, 32, 70, True), (160, 7, 3, False),       # narrow: k_wino_gemm_n32
                                            (40, 72, 5, False), (72, 136, 66, True), (136, 264, 3, False),       # Cin padded to 64 / 96 / 160
                                            (36, 64, 4, False), (200, 24, 9, False)])                            # 36 -> 64; narrow with 200 -> 224
def test_single_layer_matches_the_float64_oracle(gpu, monkeypatch, cin, cout, n, pre):
    """one Conv -> ELU -> BN block (with and without a BN -> ReLU prologue), frame counts around the 64-frame GEMM row block,
    Cout that is not a multiple of the 128-column block (96, 338): the layer's tensor against the oracle in float64"""
    monkeypatch.setenv("TH_WINOGRAD", "1")
    monkeypatch.setenv("TH_NO_TAIL_FUSE", "1")     # the layer's tensor is fetched below: keep it (no pooling output transform)
    cfg, w, layer = _one_layer(cin, cout, seed=cin + cout, pre=pre)
    rng = np.random.default_rng(n)
    frames = (rng.standard_normal((n, 5, 5, 5, cin)) * (rng.random((n, 5, 5, 5, cin)) < 0.5)).astype(np.float32)
    model = engine.HipFrameModel.from_keras(cfg, w, device=gpu)       # (the block's last tensor is materialised: fetch works)
    assert any("conv_wino" in s["label"] for s in model.steps()), [s["label"] for s in model.steps()]
    probs = model.predict(frames)
    ref = cnn_oracle.forward(cfg, w, frames[: min(n, 8)], np.float64, return_all=True)
    got = model.fetch(layer, min(n, 8), (5, 5, 5, cout))
    want = ref[layer]
    np.testing.assert_allclose(got, want, atol=LAYER * max(1.0, float(np.abs(want).max())), rtol=0)
    assert float(np.sqrt(np.mean((got - want) ** 2))) < 3e-6
    last = list(ref)[-1]
    np.testing.assert_allclose(probs[: min(n, 8)], ref[last], atol=TIGHT, rtol=0)
    if n > 8:                                 # every row block and the ragged last one: against the direct kernels
        model.close()
        monkeypatch.setenv("TH_WINOGRAD", "0")
        direct = engine.HipFrameModel.from_keras(cfg, w, device=gpu)
        assert not any("conv_wino" in s["label"] for s in direct.steps())
        np.testing.assert_allclose(probs, direct.predict(frames), atol=TIGHT, rtol=0)
        direct.close()
    else:
        model.close()


@pytest.mark.parametrize("name", ["timed20", "timed338", "timed20_c5_bias", "timed20_bool"])
def test_timed_fixtures_with_winograd(gpu, cnn_golden, monkeypatch, name):
    """the torch-fp64 fixtures of tests/test_gpu_cnn.py with the 5^3 layers on the Winograd path: same 5e-6 bound on
    probabilities and logits, same argmax; small and large chunks (one launch, and launches of 1-3 frames)"""
    monkeypatch.setenv("TH_WINOGRAD", "1")
    z, meta = cnn_golden
    m = next(x for x in meta if x["name"] == name)
    cfg, weights = getattr(synth, m["builder"])(**m["kwargs"])
    frames = synth.synthetic_frames(m["n"], **m["frame_kwargs"])
    model = engine.HipFrameModel.from_keras(cfg, weights, device=gpu)
    assert sum("conv_wino" in s["label"] for s in model.steps()) >= 3
    for chunk in (1024, 3):
        model.set_chunk(chunk)
        probs = model.predict(frames)
        np.testing.assert_allclose(probs, z[f"{name}__torch64"], atol=TIGHT, rtol=0)
        assert np.array_equal(probs.argmax(1), z[f"{name}__torch64"].argmax(1))
        logits = model.predict(frames, logits=True)
        np.testing.assert_allclose(logits, z[f"{name}__logits64"], atol=TIGHT, rtol=0)
    model.close()


def test_winograd_is_not_taken_where_it_does_not_apply(gpu, monkeypatch):
    """valid padding, 10^3 volumes, strides, narrow heads and 1x1x1 layers stay on the direct kernels"""
    monkeypatch.setenv("TH_WINOGRAD", "1")
    cfg, w = synth.densecpd_synth(20)
    model = engine.HipFrameModel.from_keras(cfg, w, device=gpu)
    labels = [s["label"] for s in model.steps()]
    assert not any("conv_wino" in l for l in labels)         # growth convs are 16 wide, bottlenecks 1x1x1
    model.close()
    cfg, w = synth.timed_synth(20)
    model = engine.HipFrameModel.from_keras(cfg, w, device=gpu)
    labels = [s["label"] for s in model.steps()]
    conv = [l for l in labels if l.startswith("conv3d") and not any(k in l for k in ("wino_in", "wino_out", "wino_mid"))]
    # the 20-class head (256 -> 20) is a Winograd layer too since round 5: the narrow split GEMM (k_wino_gemm_n32)
    assert [("k_wino_gemm" in l) for l in conv] == [False, False, True, True, True, True]
    assert "k_wino_gemm_n32" in conv[-1] and "bf16x3" in conv[-1]
    # conv3d_2 .. conv3d_5 are consecutive Winograd layers: one input transform, three fused mid transforms, one (pooling) output
    assert [sum(k in l for l in labels) for k in ("k_wino_in", "k_wino_mid", "k_wino_out")] == [1, 3, 1]
    model.close()
    monkeypatch.setenv("TH_WINO_SPLIT", "0")           # fp32-input MFMA GEMMs: the narrow head stays on the direct kernel
    model = engine.HipFrameModel.from_keras(cfg, w, device=gpu)
    labels = [s["label"] for s in model.steps()]
    conv = [l for l in labels if l.startswith("conv3d") and not any(k in l for k in ("wino_in", "wino_out", "wino_mid"))]
    assert [("k_wino_gemm" in l) for l in conv] == [False, False, True, True, True, False]
    assert [sum(k in l for l in labels) for k in ("k_wino_in", "k_wino_mid", "k_wino_out")] == [1, 2, 1]
    model.close()


FAST = 2e-5          # F(5, 3): asserted bound of the opt-in 7-point scheme (north star: 1e-4 on the logits)


@pytest.mark.parametrize("name", ["timed20", "timed338"])
def test_timed_fixtures_with_the_seven_point_scheme(gpu, cnn_golden, monkeypatch, name):
    """TH_WINOGRAD=2: F(5, 3) in both in-plane axes (49 positions per plane instead of 81).  Its transforms carry entries up
    to 16, so the rounding error is ~4x that of the default scheme (tests/winograd_numerics.py: 2.4e-6 on the
    logits of TIMED-synth against 6.8e-7): asserted at 2e-5 — five times inside the north-star bound — with equal argmax;
    never the default and never the headline number."""
    monkeypatch.setenv("TH_WINOGRAD", "2")
    z, meta = cnn_golden
    m = next(x for x in meta if x["name"] == name)
    cfg, weights = getattr(synth, m["builder"])(**m["kwargs"])
    frames = synth.synthetic_frames(m["n"], **m["frame_kwargs"])
    model = engine.HipFrameModel.from_keras(cfg, weights, device=gpu)
    assert sum("F(5,3)" in s["label"] for s in model.steps()) >= 3
    probs = model.predict(frames)
    np.testing.assert_allclose(probs, z[f"{name}__torch64"], atol=FAST, rtol=0)
    assert np.array_equal(probs.argmax(1), z[f"{name}__torch64"].argmax(1))
    logits = model.predict(frames, logits=True)
    np.testing.assert_allclose(logits, z[f"{name}__logits64"], atol=FAST, rtol=0)
    err = float(np.abs(logits - z[f"{name}__logits64"]).max())
    print(f"{name}: F(5,3) max |dlogit| = {err:.2e}")
    model.close()


@pytest.mark.parametrize("cin,cout,n", [(64, 128, 70), (128, 256, 3)])
def test_single_layer_seven_point_scheme(gpu, monkeypatch, cin, cout, n):
    monkeypatch.setenv("TH_WINOGRAD", "2")
    monkeypatch.setenv("TH_NO_TAIL_FUSE", "1")     # the layer's tensor is fetched below
    cfg, w, layer = _one_layer(cin, cout, seed=cin + cout)
    rng = np.random.default_rng(n)
    frames = (rng.standard_normal((n, 5, 5, 5, cin)) * (rng.random((n, 5, 5, 5, cin)) < 0.5)).astype(np.float32)
    model = engine.HipFrameModel.from_keras(cfg, w, device=gpu)
    assert any("F(5,3)" in s["label"] for s in model.steps())
    model.predict(frames)
    k = min(n, 6)
    want = cnn_oracle.forward(cfg, w, frames[:k], np.float64, return_all=True)[layer]
    got = model.fetch(layer, k, (5, 5, 5, cout))
    np.testing.assert_allclose(got, want, atol=1e-4 * max(1.0, float(np.abs(want).max())), rtol=0)
    assert float(np.sqrt(np.mean((got - want) ** 2))) < 1.5e-5
    model.close()


def test_winograd_layers_on_concat_arenas(gpu):
    """a Winograd layer that WRITES into a channel slice of a zero-copy concat arena (voxel stride 96, first channel 32), one
    that READS the whole arena (Cin = 96) and one that reads only the slice its sibling wrote: strided views on both sides of
    the transform kernels, against the float64 oracle"""
    b = synth.KerasGraphBuilder((5, 5, 5, 32), seed=11)
    x = b.input_name
    a = b.batchnorm(b.elu(b.conv3d(x, 64, 3, padding="same")))
    c = b.concat([x, a])
    y1 = b.batchnorm(b.elu(b.conv3d(c, 64, 3, padding="same")))
    y2 = b.relu(b.conv3d(a, 96, 3, padding="same"))
    z = b.concat([y1, y2])
    out = b.softmax(b.gap(b.conv3d(z, 20, 1, padding="same")))
    cfg, w = b.finish(out)
    rng = np.random.default_rng(4)
    frames = (rng.standard_normal((9, 5, 5, 5, 32)) * (rng.random((9, 5, 5, 5, 32)) < 0.4)).astype(np.float32)
    model = engine.HipFrameModel.from_keras(cfg, w, device=gpu)
    assert sum("k_wino_gemm" in s["label"] for s in model.steps()) == 3, [s["label"] for s in model.steps()]
    probs = model.predict(frames)
    ref = cnn_oracle.forward(cfg, w, frames, np.float64, return_all=True)
    np.testing.assert_allclose(probs, ref[list(ref)[-1]], atol=TIGHT, rtol=0)
    for name in (y1, y2):
        want = ref[name]
        got = model.fetch(name, 9, want.shape[1:])
        np.testing.assert_allclose(got, want, atol=LAYER * max(1.0, float(np.abs(want).max())), rtol=0, err_msg=name)
    model.close()


def test_transform_launches_in_pieces(gpu):
    """the transform kernels address with 32-bit offsets; a launch beyond that is issued in pieces (TH_WINO_PIECE forces 50-frame
    pieces here — the library reads it once per process, hence the child): same bits as the one-piece run"""
    import os
    import subprocess
    import sys
    code = (
        "import sys, numpy as np\n"
        "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "from timed_hip import engine\nimport test_gpu_wino as T\n"
        "cfg, w, layer = T._one_layer(64, 128, seed=5)\n"
        "rng = np.random.default_rng(0)\n"
        "x = rng.standard_normal((130, 5, 5, 5, 64)).astype(np.float32)\n"
        "m = engine.HipFrameModel.from_keras(cfg, w)\n"
        "np.save(sys.argv[1], m.predict(x, logits=True))\n"
    ) % (os.path.dirname(os.path.abspath(engine.__file__)) + "/..", os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for piece in ("", "50"):
        out = f"/tmp/wino_piece_{piece or 'one'}_{os.getpid()}.npy"
        env = dict(os.environ)
        env.pop("TH_WINO_PIECE", None)
        if piece:
            env["TH_WINO_PIECE"] = piece
        subprocess.run([sys.executable, "-c", code, out], check=True, env=env)
        outs.append(np.load(out))
        os.remove(out)
    assert np.isfinite(outs[0]).all() and np.array_equal(outs[0], outs[1])


@pytest.mark.parametrize("name", ["timed20", "timed338"])
def test_fused_mid_transform_is_bit_identical(gpu, cnn_golden, monkeypatch, name):
    """k_wino_mid (output transform of one Winograd layer + input transform of the next, the tensor between them never written) against
    the separate k_wino_out / k_wino_in launches (TH_WINO_NOMID=1): same arithmetic in the same order, so the same bits"""
    z, meta = cnn_golden
    m = next(x for x in meta if x["name"] == name)
    cfg, weights = getattr(synth, m["builder"])(**m["kwargs"])
    frames = synth.synthetic_frames(m["n"], **m["frame_kwargs"])
    fused = engine.HipFrameModel.from_keras(cfg, weights, device=gpu)
    assert sum("k_wino_mid" in s["label"] for s in fused.steps()) >= 2
    with pytest.raises(_lib.TimedHipError, match="fused away"):
        fused.predict(frames[:1]); fused.fetch("batch_normalization_2", 1, (5, 5, 5, 128))
    monkeypatch.setenv("TH_WINO_NOMID", "1")
    plain = engine.HipFrameModel.from_keras(cfg, weights, device=gpu)
    assert not any("k_wino_mid" in s["label"] for s in plain.steps())
    assert np.array_equal(fused.predict(frames, logits=True), plain.predict(frames, logits=True))
    assert np.array_equal(fused.predict(frames), plain.predict(frames))
    fused.close(); plain.close()


@pytest.mark.parametrize("cin,cout,n", [(64, 338, 9), (32, 64, 70)])
def test_output_transform_pools_for_a_global_average_tail(gpu, monkeypatch, cin, cout, n):
    """Conv -> ELU -> BN -> GlobalAveragePooling3D -> Softmax with the convolution on conv_wino.hip: the output transform adds up
    the 125 values of a (frame, channel) itself (k_wino_out<P, true>) and the 5^3 activation is never written.  Same summation
    order as the pooling kernels: probabilities AND logits equal the unfused tail's bit for bit; against the oracle as usual."""
    monkeypatch.setenv("TH_WINOGRAD", "1")
    cfg, w, layer = _one_layer(cin, cout, seed=cin * 3 + cout, pre=False)
    rng = np.random.default_rng(n)
    frames = (rng.standard_normal((n, 5, 5, 5, cin)) * (rng.random((n, 5, 5, 5, cin)) < 0.5)).astype(np.float32)
    fused = engine.HipFrameModel.from_keras(cfg, w, device=gpu)
    labels = [s["label"] for s in fused.steps()]
    assert any("wino_out + global_avg_pool" in l for l in labels) and not any("k_gap_softmax" in l for l in labels), labels
    with pytest.raises(_lib.TimedHipError):
        fused.predict(frames[:1]); fused.fetch(layer, 1, (5, 5, 5, cout))          # that tensor does not exist
    monkeypatch.setenv("TH_NO_TAIL_FUSE", "1")
    plain = engine.HipFrameModel.from_keras(cfg, w, device=gpu)
    assert not any("global_avg_pool (" in s["label"] for s in plain.steps())
    for k in (1, n):
        assert np.array_equal(fused.predict(frames[:k]), plain.predict(frames[:k]))
        assert np.array_equal(fused.predict(frames[:k], logits=True), plain.predict(frames[:k], logits=True))
    ref = cnn_oracle.forward(cfg, w, frames[:8], np.float64)
    np.testing.assert_allclose(fused.predict(frames[:8]), ref, atol=TIGHT, rtol=0)
    fused.close(); plain.close()


def test_consecutive_layers_with_padded_channel_counts(gpu):
    """widths that are no multiple of the GEMM's 32-channel chunk through the fused mid transform: 40 -> 72 -> 136 -> 20 at 5^3
    (V rows 64, 96, 160 and 160 channels wide; k_wino_mid writes zeros into the padding channels of the next layer's V, whose
    scratch arena was last used by a layer of another width) against the float64 oracle; twice, so that the second run finds the
    scratch of the first"""
    b = synth.KerasGraphBuilder((5, 5, 5, 40), seed=77)
    x = b.input_name
    for c in (72, 136, 136, 20):
        x = b.conv3d(x, c, 3, padding="same")
        x = b.elu(x)
        x = b.batchnorm(x)
    x = b.gap(x)
    x = b.softmax(x)
    cfg, w = b.finish(x)
    rng = np.random.default_rng(5)
    frames = (rng.standard_normal((70, 5, 5, 5, 40)) * (rng.random((70, 5, 5, 5, 40)) < 0.5)).astype(np.float32)
    model = engine.HipFrameModel.from_keras(cfg, w, device=gpu)
    labels = [s["label"] for s in model.steps()]
    assert sum("k_wino_gemm" in l for l in labels) == 4 and sum("k_wino_mid" in l for l in labels) == 3, labels
    assert any("K64 (40 real)" in l for l in labels) and any("K160 (136 real)" in l for l in labels), labels
    ref = cnn_oracle.forward(cfg, w, frames[:6], np.float64, return_all=True)
    last = list(ref)[-1]
    logit_layer = [k for k in ref if "global_average" in k][-1]
    for _ in range(2):
        probs = model.predict(frames)
        np.testing.assert_allclose(probs[:6], ref[last], atol=TIGHT, rtol=0)
        logits = model.predict(frames, logits=True)
        np.testing.assert_allclose(logits[:6], ref[logit_layer], atol=TIGHT * max(1.0, float(np.abs(ref[logit_layer]).max())), rtol=0)
    model.set_chunk(64)                                       # the ragged second launch of 6 frames
    np.testing.assert_allclose(model.predict(frames), probs, atol=1e-6, rtol=0)
    model.close()
