"""Per-element parity of the fused convolution kernels over a sweep of shapes: non-cubic volumes,
odd channel counts, anisotropic / 1x1x1 / 5x5x5 kernels, 'same' vs 'valid', pooling on odd extents,
pre-activation (BN -> ReLU -> Conv) and post-activation chains, channel-offset writes into concat
buffers, batch sizes that do not divide the frames-per-workgroup.  The block output is Flatten-ed so
every voxel/channel is compared against the CPU oracle (tolerance 2e-5 relative to the tensor's scale;
fp32 MFMA vs BLAS differ only in accumulation order)."""
import numpy as np
import pytest

from oracle import cnn_oracle
from timed_hip import _lib, engine, synth

pytestmark = pytest.mark.gpu


def _net(shape, cin, build, seed=0, bias_std=0.3):
    b = synth.KerasGraphBuilder((*shape, cin), seed=seed, bias_std=bias_std)
    x = build(b, b.input_name)
    x = b.flatten(x)
    return b.finish(x)


def _frames(n, shape, cin, seed):
    rng = np.random.default_rng(seed)
    return (rng.standard_normal((n, *shape, cin)) * (rng.random((n, *shape, cin)) < 0.5)).astype(np.float32)


def _check(cfg, weights, frames, flags=0, chunk=None):
    want = cnn_oracle.forward(cfg, weights, frames, np.float32)
    m = engine.HipFrameModel.from_keras(cfg, weights, flags=flags)
    if chunk:
        m.set_chunk(chunk)
    got = m.predict(frames)
    labels = [s["label"] for s in m.steps()]
    m.close()
    scale = max(1.0, float(np.abs(want).max()))
    err = float(np.abs(got - want).max())
    assert got.shape == want.shape and err <= 2e-5 * scale, (err, scale, labels)
    return labels


# (shape, cin, cout, kernel, padding, pool, post, n_frames)
SWEEP = [
    ((21, 21, 21), 6, 32, 3, "same", "max", "elu_bn", 3),      # TIMED block 1 -> conv_first
    ((21, 21, 21), 5, 24, 3, "same", "max", "elu_bn", 2),      # 5-channel codec (CNOCBCA), Cout not a multiple of 32
    ((9, 7, 5), 3, 7, 3, "same", None, "relu", 5),              # non-cubic, tiny channel counts
    ((9, 7, 5), 8, 32, 3, "valid", "avg", "bn_relu", 5),        # valid padding + average pool on odd extents
    ((10, 10, 10), 32, 64, 3, "same", "max", "elu_bn", 3),      # TIMED block 2 shape
    ((5, 5, 5), 64, 128, 3, "same", None, "elu_bn", 7),         # TIMED block 3, n not a multiple of frames/workgroup
    ((5, 5, 5), 40, 130, 3, "same", None, "leaky", 3),          # Cin not a multiple of 16, Cout spans two N blocks
    ((5, 5, 5), 128, 20, 3, "same", None, "elu_bn", 5),         # last TIMED conv (Cout 20 -> padded tile)
    ((6, 6, 6), 48, 16, 3, "same", None, "none", 5),            # narrow output -> conv_n16
    ((10, 10, 10), 64, 16, 3, "same", None, "bn_relu", 2),      # DenseCPD growth conv shape
    ((4, 4, 4), 24, 12, 3, "same", "max", "relu", 9),           # conv_n16 + pool, Cout < 16
    ((8, 8, 8), 20, 64, 1, "same", None, "bn_relu", 3),         # 1x1x1 bottleneck
    ((7, 5, 9), 18, 100, 1, "same", "max", "elu_bn", 3),        # 1x1x1: Cin % 4 != 0, 4 output tiles, max pool on odd extents
    ((5, 5, 5), 136, 48, 1, "valid", "avg", "none", 5),         # 1x1x1: K > 128 (two K passes) + avg pool (transition layer)
    ((3, 3, 3), 12, 130, 1, "same", None, "relu", 4),           # 1x1x1 with Cout > 128 -> generic MFMA kernel
    ((8, 8, 8), 32, 64, 1, "same", None, "bn_relu", 3),         # 1x1x1, Cin % 8 == 0 -> pipelined k_conv_pw2<4,2,0>
    ((6, 6, 6), 96, 48, 1, "valid", "avg", "none", 5),          # k_conv_pw2<16,2,2> (transition: avg pool), ragged last tile
    ((4, 4, 4), 64, 100, 1, "same", "max", "elu_bn", 9),        # k_conv_pw2<8,4,1>, four output tiles
    ((5, 5, 5), 56, 20, 1, "same", None, "relu", 4),            # k_conv_pw2<8,1,0>: K8 = 7 of 8 slots
    ((7, 7, 7), 6, 16, 5, "same", None, "relu", 2),             # 5x5x5 kernel (125 taps)
    ((8, 6, 7), 12, 40, (3, 1, 3), "same", None, "elu", 3),     # anisotropic kernel
    ((2, 2, 2), 96, 16, 3, "same", None, "none", 17),           # tiny volume (DenseCPD block 3)
]


@pytest.mark.parametrize("shape,cin,cout,k,padding,pool,post,n", SWEEP)
def test_fused_conv_block_per_element(gpu, shape, cin, cout, k, padding, pool, post, n):
    def build(b, x):
        x = b.conv3d(x, cout, k, padding=padding)
        if post == "elu_bn":
            x = b.batchnorm(b.elu(x))
        elif post == "bn_relu":
            x = b.relu(b.batchnorm(x))
        elif post == "relu":
            x = b.relu(x)
        elif post == "elu":
            x = b.elu(x, 0.7)
        elif post == "leaky":
            x = b.leaky_relu(x, 0.2)
        if pool == "max":
            x = b.maxpool(x, 2)
        elif pool == "avg":
            x = b.avgpool(x, 2)
        return x

    cfg, weights = _net(shape, cin, build, seed=hash((shape, cin, cout)) % 1000)
    frames = _frames(n, shape, cin, seed=n)
    labels = _check(cfg, weights, frames)
    assert any("conv_" in l for l in labels), labels               # an MFMA kernel ran, not the direct fallback
    _check(cfg, weights, frames, flags=_lib.TH_LOAD_NO_MFMA)        # and the generic path agrees too
    _check(cfg, weights, frames, chunk=2)                           # ragged chunks


def test_preactivation_dense_layer_with_concat(gpu):
    """BN -> ReLU -> Conv1x1 -> BN -> ReLU -> Conv3x3x3 -> Concat, twice, written at channel offsets."""
    def build(b, x):
        x = b.conv3d(x, 24, 3, padding="same")
        for _ in range(2):
            y = b.relu(b.batchnorm(x))
            y = b.conv3d(y, 32, 1, padding="same", use_bias=False)
            y = b.relu(b.batchnorm(y))
            y = b.conv3d(y, 16, 3, padding="same", use_bias=False)
            x = b.concat([x, y])
        return b.relu(b.batchnorm(x))

    cfg, weights = _net((6, 6, 6), 6, build, seed=3)
    labels = _check(cfg, weights, _frames(5, (6, 6, 6), 6, 1))
    assert sum("concat(copy" in l for l in labels) == 0, labels      # zero-copy concat
    assert any("conv_n16" in l for l in labels), labels


def test_dense_block_transition_pointwise_kernel(gpu):
    """DenseCPD transition: concat buffer -> BN -> ReLU -> Conv1x1x1 (no bias) -> AvgPool(2), plus a
    bottleneck that reads a channel slice at a non-16-byte-aligned offset (scalar load path)."""
    def build(b, x):
        x = b.conv3d(x, 18, 3, padding="same")                       # 18 channels: the next concat slice starts at offset 18
        y = b.relu(b.batchnorm(x))
        y = b.conv3d(y, 32, 1, padding="same", use_bias=False)
        y = b.relu(b.batchnorm(y))
        y = b.conv3d(y, 14, 3, padding="same", use_bias=False)
        x = b.concat([x, y])                                         # 32 channels
        z = b.relu(b.batchnorm(y))                                   # reads the slice at channel offset 18
        z = b.conv3d(z, 40, 1, padding="same")
        t = b.relu(b.batchnorm(x))
        t = b.avgpool(b.conv3d(t, 16, 1, padding="same", use_bias=False), 2)
        return b.concat([b.avgpool(z, 2), t])

    cfg, weights = _net((7, 6, 6), 5, build, seed=11)
    labels = _check(cfg, weights, _frames(5, (7, 6, 6), 5, 6))
    assert sum("conv_pw" in l for l in labels) == 3, labels
    assert any("conv_pw" in l and "pool2" in l for l in labels), labels
    _check(cfg, weights, _frames(3, (7, 6, 6), 5, 7), flags=_lib.TH_LOAD_NO_MFMA)


@pytest.mark.parametrize("shape,cin,cout,pool,n", [((6, 6, 6), 48, 16, None, 11), ((4, 4, 4), 40, 12, "max", 23),
                                                    ((5, 5, 5), 16, 16, None, 9)])
def test_narrow_conv_persistent_workgroups_walk_several_groups(gpu, monkeypatch, shape, cin, cout, pool, n):
    """k_conv_n16 workgroups are persistent and prefetch the next frame group's first chunk: force 3
    resident workgroups so each walks several (incl. a ragged last) groups."""
    monkeypatch.setenv("TH_N16_RESIDENT", "3")

    def build(b, x):
        x = b.relu(b.batchnorm(x))
        x = b.conv3d(x, cout, 3, padding="same", use_bias=False)
        x = b.elu(x)
        return b.maxpool(x, 2) if pool else x

    cfg, weights = _net(shape, cin, build, seed=21)
    labels = _check(cfg, weights, _frames(n, shape, cin, 13))
    assert any("conv_n16" in l for l in labels), labels


@pytest.mark.parametrize("negative_gamma", [False, True])
@pytest.mark.parametrize("act,cin,cout", [("elu", 6, 32), ("relu", 6, 32), ("leaky", 6, 32), ("elu", 24, 64), ("relu", 40, 128)])
def test_first_block_pools_before_a_monotone_epilogue_only(gpu, monkeypatch, negative_gamma, act, cin, cout):
    """Conv -> act -> BN -> MaxPool (first-layer kernel for cin=6, brick kernel otherwise): when every BN scale is
    >= 0 the kernel max-pools the raw sums and runs act+BN on the pooled values (max commutes with a non-decreasing chain); a negative gamma
    on any channel must keep the original order.  Both orders are checked against the oracle, and against each
    other with the rewrite switched off."""
    def build(b, x):
        x = b.conv3d(x, cout, 3, padding="same")
        x = {"elu": b.elu, "relu": b.relu, "leaky": lambda t: b.leaky_relu(t, 0.2)}[act](x)
        return b.maxpool(b.batchnorm(x), 2)

    cfg, weights = _net((9, 8, 7), cin, build, seed=31)
    bn = [k for k in weights if k.startswith("batch_normalization")][0]
    if negative_gamma:
        g = weights[bn][0].copy()
        g[::5] *= -1.0
        weights[bn] = [g] + list(weights[bn][1:])
    frames = _frames(6, (9, 8, 7), cin, 5)
    labels = _check(cfg, weights, frames)
    assert any(("conv_first" if cin == 6 else "conv_mfma") in l and "pool1" in l for l in labels), labels
    got = engine.HipFrameModel.from_keras(cfg, weights).predict(frames)
    monkeypatch.setenv("TH_NO_POOL_FIRST", "1")
    ref = engine.HipFrameModel.from_keras(cfg, weights).predict(frames)
    if negative_gamma:
        assert np.array_equal(got, ref)          # the rewrite must not have been applied
    else:
        np.testing.assert_allclose(got, ref, rtol=0, atol=2e-6 * max(1.0, float(np.abs(ref).max())))


@pytest.mark.parametrize("shape,cin,cout,k,stride,padding", [
    ((9, 9, 9), 16, 20, 3, 2, "same"), ((10, 9, 8), 24, 48, 3, 2, "valid"), ((11, 11, 11), 6, 32, 3, 2, "same"),
    ((12, 10, 9), 5, 40, 5, 2, "same"), ((9, 9, 9), 32, 64, 3, 3, "same"), ((8, 8, 8), 48, 16, 1, 2, "same"),
])
def test_strided_convolutions_on_the_brick_kernel(gpu, monkeypatch, shape, cin, cout, k, stride, padding):
    """stride > 1 (ProDCoNN-style down-sampling convolutions): the brick kernel's row table carries the stride, the
    staged brick covers (n-1)*s + k input planes; Keras 'same' padding with a stride is asymmetric.  Since round 6 the strided
    layers with Cin % 16 == 0 go to conv_gl.hip by default: both kernels are held to the oracle here."""
    def build(b, x):
        return b.elu(b.conv3d(x, cout, k, strides=stride, padding=padding))

    cfg, weights = _net(shape, cin, build, seed=41)
    frames = _frames(5, shape, cin, 9)
    labels = _check(cfg, weights, frames)
    assert any(("conv_gl" if cin % 16 == 0 else "conv_mfma") in l for l in labels), labels
    monkeypatch.setenv("TH_CONV_GL", "0")
    labels = _check(cfg, weights, frames)
    assert any("conv_mfma" in l for l in labels), labels
    _check(cfg, weights, frames, chunk=2)


def test_branches_add_and_strided_fallback(gpu):
    def build(b, x):
        a = b.conv3d(x, 16, 3, padding="same", activation="relu")
        c = b.conv3d(x, 16, (1, 3, 3), padding="same", activation="elu")
        s = b.add([a, c])
        t = b.conv3d(s, 20, 3, strides=2, padding="same")            # stride 2 -> direct kernel
        return b.batchnorm(t)

    cfg, weights = _net((9, 9, 9), 4, build, seed=5)
    _check(cfg, weights, _frames(4, (9, 9, 9), 4, 2))


def test_nan_and_extreme_inputs_propagate_like_numpy(gpu):
    cfg, weights = _net((5, 5, 5), 6, lambda b, x: b.elu(b.conv3d(x, 8, 3, padding="same")), seed=9)
    fr = _frames(3, (5, 5, 5), 6, 4)
    fr[1, 2, 2, 2, 3] = np.nan
    fr[2, 0, 0, 0, 0] = 1e30
    want = cnn_oracle.forward(cfg, weights, fr, np.float32)
    got = engine.HipFrameModel.from_keras(cfg, weights).predict(fr)
    assert np.array_equal(np.isnan(got), np.isnan(want))
    ok = np.isfinite(want)
    np.testing.assert_allclose(got[ok], want[ok], rtol=2e-5, atol=2e-5 * max(1.0, float(np.abs(want[ok]).max())))
    assert np.array_equal(got[0], engine.HipFrameModel.from_keras(cfg, weights).predict(fr[:1])[0])  # frame independence


@pytest.mark.parametrize("side,cmid,cout,pool,n,tag", [
    (5, 64, 128, None, 5, ",7>]"),      # k_conv_mfma<4,4,2,4,16,2,0,7>: two frames per workgroup, ragged last group
    (10, 32, 64, "max", 3, ",12>]"),    # k_conv_mfma<8,4,2,2,16,2,1,12>
    (10, 48, 16, None, 3, ",12>]"),     # k_conv_n16<8,8,0,0,12>
    (5, 48, 16, None, 5, ",7>]"),       # k_conv_n16<4,4,0,0,7>
    (5, 64, 20, None, 3, ",7>]"),       # k_conv_n16<4,4,0,4,7> (20-class head)
    (2, 32, 16, None, 19, ",4>]"),      # k_conv_n16<4,4,0,0,4>, 16 frames per workgroup + 3
])
def test_compile_time_geometry_kernels_with_preactivation(gpu, monkeypatch, side, cmid, cout, pool, n, tag):
    """The instantiations with the staged row geometry fixed at compile time (tap offsets as ds_read immediates; chunks
    after the first re-stage real voxels only): BN -> ReLU in front of the convolution, several Cin chunks, ragged last
    workgroup; the plan label must name the specialised kernel.  (TH_WINOGRAD=0: these are the DIRECT kernels — the wide
    5^3 layers go to conv_wino.hip by default, tests/test_gpu_wino.py; TH_WFUSED=0: the 10^3 layers go to conv_wfused.hip by
    default, tests/test_gpu_conv_wfused.py.)"""
    monkeypatch.setenv("TH_WINOGRAD", "0")
    monkeypatch.setenv("TH_WFUSED", "0")
    def build(b, x):
        x = b.conv3d(x, cmid, 1, padding="same")
        y = b.relu(b.batchnorm(x))
        y = b.conv3d(y, cout, 3, padding="same")
        y = b.batchnorm(b.elu(y))
        return b.maxpool(y, 2) if pool == "max" else y

    cfg, weights = _net((side,) * 3, 8, build, seed=side * 100 + cout)
    frames = _frames(n, (side,) * 3, 8, seed=n)
    labels = _check(cfg, weights, frames)
    assert any(l.endswith(tag) and "conv_" in l for l in labels), labels
    _check(cfg, weights, frames, chunk=3)


# (shape, cin, cout, pool, post, n, kernels the label must name)
TAIL = [
    ((5, 5, 5), 64, 338, None, "elu_bn", 5, ("k_conv_mfma<4,4,2,4,16,2,0,7>", "k_conv_mfma<4,2,3,3,16,2,0,7>")),   # the rotamer head: 128 + 128 + 96
    ((5, 5, 5), 48, 300, None, "bn_relu", 3, ("k_conv_mfma<4,4,2,4,16,2,0,7>", "k_conv_mfma<4,4,2,2,16,2,0,7>")),  # 128 + 128 + 64
    ((5, 5, 5), 32, 280, None, "leaky", 4, ("k_conv_mfma<4,4,2,4,16,2,0,7>", "k_conv_mfma<4,2,1,1,16,2,0")),       # 128 + 128 + 32
    ((5, 5, 5), 40, 200, None, "elu_bn", 3, ("k_conv_mfma<4,4,2,4,16,2,0,7>", "k_conv_mfma<4,2,3,3,16,2,0,7>")),   # 128 + 96, Cin % 16 != 0
    ((4, 4, 4), 32, 210, "max", "elu_bn", 5, ("k_conv_mfma<4,4,2,4,16,2,1", "k_conv_mfma<4,2,3,3,16,2,1")),        # pooled, runtime geometry
]


@pytest.mark.parametrize("shape,cin,cout,pool,post,n,kernels", TAIL)
def test_heterogeneous_cout_blocks(gpu, shape, cin, cout, pool, post, n, kernels, monkeypatch):
    """the last Cout block of a wide layer on a narrower instantiation (conv_mfma_plan_tail): weights, bias and the
    per-channel BatchNorm vectors of the tail start at its first channel; per-element against the oracle, and
    bit-identical to the run with the tail switched off (same fmaf chains per output).  Direct kernels: TH_WINOGRAD=0."""
    monkeypatch.setenv("TH_WINOGRAD", "0")
    def build(b, x):
        x = b.conv3d(x, cout, 3, padding="same")
        if post == "elu_bn":
            x = b.batchnorm(b.elu(x))
        elif post == "bn_relu":
            x = b.relu(b.batchnorm(x))
        else:
            x = b.leaky_relu(x, 0.2)
        return b.maxpool(x, 2) if pool == "max" else x

    cfg, weights = _net(shape, cin, build, seed=cout)
    frames = _frames(n, shape, cin, seed=n)
    labels = _check(cfg, weights, frames)
    conv = next(l for l in labels if "conv_mfma" in l)
    assert all(k in conv for k in kernels) and " + " in conv, conv
    _check(cfg, weights, frames, chunk=2)
    with_tail = engine.HipFrameModel.from_keras(cfg, weights).predict(frames)
    monkeypatch.setenv("TH_CONV_NOTAIL", "1")
    m = engine.HipFrameModel.from_keras(cfg, weights)
    assert not any(" + " in s["label"] for s in m.steps())
    assert np.array_equal(m.predict(frames), with_tail)
    m.close()


@pytest.mark.parametrize("shape,cin,cout,pool,post,n", [
    ((21, 21, 21), 6, 32, "max", "elu_bn", 3),       # TIMED block 1: pool-first epilogue, compile-time geometry
    ((21, 21, 21), 6, 32, "max", "tanh", 2),         # non-monotone chain: the epilogue runs on all 2 x 16 values before the max
    ((12, 10, 8), 5, 24, "avg", "relu", 5),          # run-time geometry, average pool, 5 channels
    ((9, 8, 6), 8, 32, None, "leaky", 4),            # no pool: both outputs of a pair stored, 4 k-steps
    ((7, 6, 10), 3, 9, "max", "none", 7),            # 2 k-steps, odd depth (last plane dropped by the pool), Cout < 32
    ((6, 6, 4), 1, 16, None, "elu_bn", 3),           # one input channel
    ((21, 21, 21), 6, 64, "max", "elu_bn", 2),       # a 64-filter first layer: two passes of 32 columns over the caller's frames
    ((10, 8, 6), 5, 40, None, "elu_bn", 3),          # 32 + 8 columns, per-channel BatchNorm vectors of the second pass start at channel 32
    ((8, 8, 8), 6, 97, "avg", "leaky", 2),           # four passes, the last one a single column
    ((7, 6, 6), 4, 33, "max", "tanh", 3),            # non-monotone chain, 32 + 1
])
def test_first_layer_winograd_along_x(gpu, monkeypatch, shape, cin, cout, pool, post, n):
    """k_conv_first_w (F(2,3) along x: rows are x pairs, four transform points per (dz, dy) tap formed in registers) against the
    oracle and against the direct first-layer kernel (TH_FIRST_WINO=0), ragged chunks included."""
    def build(b, x):
        x = b.conv3d(x, cout, 3, padding="same")
        if post == "elu_bn":
            x = b.batchnorm(b.elu(x))
        elif post == "relu":
            x = b.relu(x)
        elif post == "leaky":
            x = b.leaky_relu(x, 0.2)
        elif post == "tanh":
            x = b.activation(x, "tanh")
        if pool == "max":
            x = b.maxpool(x, 2)
        elif pool == "avg":
            x = b.avgpool(x, 2)
        return x

    cfg, weights = _net(shape, cin, build, seed=cin * 7 + cout)
    frames = _frames(n, shape, cin, seed=n + 3)
    labels = _check(cfg, weights, frames)
    assert any("k_conv_first_w" in l or "k_conv_first_b3" in l for l in labels), labels
    if cout > 32:
        assert any(f"x{(cout + 31) // 32} passes" in l for l in labels), labels
    _check(cfg, weights, frames, chunk=2)
    got = engine.HipFrameModel.from_keras(cfg, weights).predict(frames)
    monkeypatch.setenv("TH_FIRST_WINO", "0")
    m = engine.HipFrameModel.from_keras(cfg, weights)
    ref = m.predict(frames)
    assert any("k_conv_first<" in s["label"] for s in m.steps()), [s["label"] for s in m.steps()]
    np.testing.assert_allclose(got, ref, rtol=0, atol=4e-6 * max(1.0, float(np.abs(ref).max())))


@pytest.mark.parametrize("cin,cout,post,n,resident,dtype", [
    (6, 32, "elu_bn", 3, 0, np.float32),        # TIMED block 1
    (6, 32, "elu_bn", 11, 2, np.float32),       # two workgroups: 6 and 5 frames each stream through the plane ring back to back
    (6, 32, "bn_relu", 7, 3, np.uint8),         # DenseCPD's opening chain, uint8 frames (the generic element loads), trips 3 / 2 / 2
    (5, 24, "relu", 4, 1, np.float32),          # 5-channel codec, Cout < 32, ONE workgroup walks all four frames
    (6, 64, "elu_bn", 5, 2, np.float32),        # two passes of 32 columns
    (6, 20, "relu", 1, 0, np.float64),          # a single frame, float64 frames
])
def test_first_layer_on_the_bf16_pipe_with_split_operands(gpu, monkeypatch, cin, cout, post, n, resident, dtype):
    """k_conv_first_b3 (conv_first_b3.hip: the aposteriori first layer as F(2,3) along x on bf16 MFMA, operands split exactly
    into three bf16 pieces, a persistent workgroup streaming frames through a ring of planes) against the oracle, against the
    fp32-input kernel it replaces (TH_FIRST_SPLIT=0), with several frames per workgroup and ragged chunks."""
    def build(b, x):
        x = b.conv3d(x, cout, 3, padding="same")
        if post == "elu_bn":
            x = b.batchnorm(b.elu(x))
        elif post == "bn_relu":
            x = b.relu(b.batchnorm(x))
        elif post == "relu":
            x = b.relu(x)
        return b.maxpool(x, 2)

    shape = (21, 21, 21)
    cfg, weights = _net(shape, cin, build, seed=cin * 11 + cout)
    if dtype == np.uint8:
        frames = np.random.default_rng(n).integers(0, 256, (n, *shape, cin)).astype(np.uint8) * (np.random.default_rng(n + 1).random((n, *shape, cin)) < 0.3)
        frames = frames.astype(np.uint8)
    else:
        frames = _frames(n, shape, cin, seed=n + 5).astype(dtype)
    if resident:
        monkeypatch.setenv("TH_WF_RESIDENT", str(resident))
    want = cnn_oracle.forward(cfg, weights, frames.astype(np.float32), np.float32)
    scale = max(1.0, float(np.abs(want).max()))
    m = engine.HipFrameModel.from_keras(cfg, weights)
    labels = [s["label"] for s in m.steps()]
    assert any("k_conv_first_b3" in l and "bf16x3" in l for l in labels), labels
    if cout > 32:
        assert any("x2 passes" in l for l in labels), labels
    got = m.predict(frames)
    m.set_chunk(3)
    got3 = m.predict(frames)
    m.close()
    assert float(np.abs(got - want).max()) <= 2e-5 * scale
    np.testing.assert_array_equal(got, got3)                    # a frame's result does not depend on its place in the ring
    monkeypatch.setenv("TH_FIRST_SPLIT", "0")
    m = engine.HipFrameModel.from_keras(cfg, weights)
    assert any("k_conv_first_w" in s["label"] for s in m.steps()), [s["label"] for s in m.steps()]
    ref = m.predict(frames)
    m.close()
    np.testing.assert_allclose(got, ref, rtol=0, atol=4e-6 * scale)


def test_split_first_layer_over_twenty_decades_of_input_magnitude(gpu, monkeypatch):
    """x = h + m + l with bf16 pieces is exact over the whole fp32 exponent range (bf16 has fp32's exponent): frames whose voxels
    span 1e-14 .. 1e6 (per-frame scale, both signs, a third of the voxels exactly zero) through k_conv_first_b3 agree with the
    fp32-input kernel to 4e-6 of each FRAME's own output scale — an absolute bound on the whole batch would hide the small frames."""
    def build(b, x):
        return b.maxpool(b.batchnorm(b.elu(b.conv3d(x, 32, 3, padding="same"))), 2)

    shape = (21, 21, 21)
    cfg, weights = _net(shape, 6, build, seed=5, bias_std=0.0)      # no bias: the output scales with the input
    rng = np.random.default_rng(11)
    scales = np.array([1e-14, 1e-9, 1e-4, 1.0, 3e2, 1e6], dtype=np.float64)
    frames = rng.standard_normal((len(scales), *shape, 6)) * (rng.random((len(scales), *shape, 6)) < 0.66)
    frames = (frames * scales[:, None, None, None, None]).astype(np.float32)
    m = engine.HipFrameModel.from_keras(cfg, weights)
    assert any("k_conv_first_b3" in s["label"] for s in m.steps())
    got = m.predict(frames)
    m.close()
    monkeypatch.setenv("TH_FIRST_SPLIT", "0")
    m = engine.HipFrameModel.from_keras(cfg, weights)
    ref = m.predict(frames)
    m.close()
    want = cnn_oracle.forward(cfg, weights, frames, np.float32)
    for i in range(len(scales)):
        # ELU -> BatchNorm of a zero-bias convolution: subtract the frame-independent offset f(0) before scaling
        s = max(float(np.abs(want[i] - want[i].mean()).max()), 1e-30)
        assert float(np.abs(got[i] - ref[i]).max()) <= 4e-6 * max(s, float(np.abs(want[i]).max()) * 1e-3), (i, scales[i])
        assert float(np.abs(got[i] - want[i]).max()) <= 2e-5 * max(s, float(np.abs(want[i]).max()) * 1e-3), (i, scales[i])
