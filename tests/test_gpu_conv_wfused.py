"""Per-element parity of the fused Winograd kernel (csrc/conv_wfused.hip: F(2,3) x F(2,3) in-plane, z taps direct, transform
domain in LDS) against the CPU oracle: every prologue / epilogue / pooling instantiation, output-channel counts that are not
multiples of 16, channel-slice inputs and outputs inside concat arenas, frame counts that are not multiples of the 8-frame XCD
deal, persistent workgroups that walk several (frame, column block) units, and bit-equality of ragged chunking.  Same bound as
the direct kernels (2e-5 of the tensor's scale): BT / AT hold 0 and +-1 only, the halves of G are folded into the weights."""
import numpy as np
import pytest

from oracle import cnn_oracle
from timed_hip import _lib, engine, synth

pytestmark = pytest.mark.gpu

SHAPE = (10, 10, 10)


def _net(cin, build, seed=0, bias_std=0.3):
    b = synth.KerasGraphBuilder((*SHAPE, cin), seed=seed, bias_std=bias_std)
    x = build(b, b.input_name)
    x = b.flatten(x)
    return b.finish(x)


def _frames(n, cin, seed):
    rng = np.random.default_rng(seed)
    return (rng.standard_normal((n, *SHAPE, cin)) * (rng.random((n, *SHAPE, cin)) < 0.5)).astype(np.float32)


def _run(cfg, weights, frames, chunk=None, flags=0):
    m = engine.HipFrameModel.from_keras(cfg, weights, flags=flags)
    if chunk:
        m.set_chunk(chunk)
    got = m.predict(frames)
    labels = [s["label"] for s in m.steps()]
    m.close()
    return got, labels


def _check(cfg, weights, frames, chunk=None, expect_wf=1):
    want = cnn_oracle.forward(cfg, weights, frames, np.float32)
    got, labels = _run(cfg, weights, frames, chunk)
    scale = max(1.0, float(np.abs(want).max()))
    err = float(np.abs(got - want).max())
    assert got.shape == want.shape and err <= 2e-5 * scale, (err, scale, labels)
    assert sum("k_conv_wf" in l for l in labels) == expect_wf, labels
    return got, labels


POSTS = {
    "none": lambda b, x: x,
    "relu": lambda b, x: b.relu(x),
    "elu_bn": lambda b, x: b.batchnorm(b.elu(x)),
    "bn_relu": lambda b, x: b.relu(b.batchnorm(x)),
    "leaky": lambda b, x: b.leaky_relu(x, 0.2),
    "tanh": lambda b, x: b.activation(x, "tanh"),
}
PRES = {
    "none": lambda b, x: x,
    "bn_relu": lambda b, x: b.relu(b.batchnorm(x)),       # DenseNet pre-activation: the PRE = 1 instantiation
    "relu": lambda b, x: b.relu(x),                       # no affine: generic prologue
    "bn_elu": lambda b, x: b.elu(b.batchnorm(x)),         # generic prologue with an exp
}

# (cin, cout, pre, post, pool, n_frames)
CASES = [
    (64, 16, "bn_relu", "none", None, 3),        # DenseCPD growth convolution
    (32, 64, "none", "elu_bn", "max", 3),        # TIMED conv3d_1 (pool-first epilogue: the BN scales of the builder are positive)
    (16, 16, "none", "none", None, 1),           # smallest K (4 stages), one frame: 7 of the 8 dealt slots are empty
    (20, 7, "none", "relu", None, 9),            # Cin = 5 chunks, Cout < 16, 9 frames: a second, ragged deal
    (24, 20, "relu", "leaky", None, 5),          # two column blocks, the second with 4 real channels
    (48, 33, "bn_elu", "tanh", "avg", 4),        # three column blocks, generic prologue, non-monotone epilogue before the average pool
    (32, 17, "bn_relu", "bn_relu", "max", 6),    # max pool, monotone chain
    (16, 48, "none", "elu_bn", "avg", 10),
    (128, 16, "bn_relu", "elu_bn", None, 2),     # 32 stages per unit
]


@pytest.mark.parametrize("cin,cout,pre,post,pool,n", CASES)
def test_wfused_block_per_element(gpu, cin, cout, pre, post, pool, n):
    def build(b, x):
        if pre != "none":
            x = b.conv3d(x, cin, 1, padding="same")          # something for the prologue to be fused onto
            x = PRES[pre](b, x)
        x = b.conv3d(x, cout, 3, padding="same", use_bias=(post != "none"))
        x = POSTS[post](b, x)
        if pool == "max":
            x = b.maxpool(x, 2)
        elif pool == "avg":
            x = b.avgpool(x, 2)
        return x

    cfg, weights = _net(cin, build, seed=(cin * 131 + cout) % 997)
    frames = _frames(n, cin, seed=n)
    got, labels = _check(cfg, weights, frames)
    got2, _ = _run(cfg, weights, frames, chunk=2)                  # ragged chunks: bit-identical
    assert np.array_equal(got, got2)
    ref, rl = _run(cfg, weights, frames, flags=_lib.TH_LOAD_NO_MFMA)   # generic path (no MFMA kernels): within the same bound
    assert not any("k_conv_wf" in l for l in rl), rl
    assert float(np.abs(got - ref).max()) <= 4e-5 * max(1.0, float(np.abs(ref).max()))


def test_wfused_negative_gamma_keeps_the_epilogue_before_the_max_pool(gpu):
    def build(b, x):
        return b.maxpool(b.batchnorm(b.elu(b.conv3d(x, 32, 3, padding="same"))), 2)

    cfg, weights = _net(16, build, seed=5)
    bn = [k for k in weights if k.startswith("batch_normalization")][0]
    g = weights[bn][0].copy()
    g[::3] *= -1.0
    weights[bn] = [g] + list(weights[bn][1:])
    _check(cfg, weights, _frames(5, 16, 2))


def test_wfused_persistent_workgroups_walk_several_units(gpu, monkeypatch):
    """8 resident workgroups, 21 frames x 3 column blocks = 72 slots: every workgroup runs 9 units back to back (the software
    pipeline crosses unit boundaries: the first chunk of the next frame is loaded and transformed under the last of this one)."""
    monkeypatch.setenv("TH_WF_RESIDENT", "8")

    def build(b, x):
        x = b.relu(b.batchnorm(x))
        return b.elu(b.conv3d(x, 40, 3, padding="same"))

    cfg, weights = _net(16, build, seed=21)
    _check(cfg, weights, _frames(21, 16, 13))


def test_wfused_reads_and_writes_channel_slices_of_concat_arenas(gpu):
    """A two-layer dense block at 10^3: the growth convolutions read the 32-channel bottleneck, write 16-channel slices at
    channel offsets 24 and 40 of the concat arena; the bottlenecks read the arena."""
    def build(b, x):
        x = b.conv3d(x, 24, 3, padding="same")
        for _ in range(2):
            y = b.relu(b.batchnorm(x))
            y = b.conv3d(y, 32, 1, padding="same", use_bias=False)
            y = b.relu(b.batchnorm(y))
            y = b.conv3d(y, 16, 3, padding="same", use_bias=False)
            x = b.concat([x, y])
        return b.relu(b.batchnorm(x))

    cfg, weights = _net(16, build, seed=3)
    got, labels = _check(cfg, weights, _frames(5, 16, 1), expect_wf=3)
    assert sum("concat(copy" in l for l in labels) == 0, labels


def test_wfused_input_slice_of_an_arena_and_switch(gpu, monkeypatch):
    """The convolution reads channels 16..31 of a concat arena (a slice with cs = 48, coff = 16); TH_WFUSED=0 plans the direct
    kernels and both agree."""
    def build(b, x):
        a = b.conv3d(x, 16, 1, padding="same")
        c = b.conv3d(x, 16, 1, padding="same")
        d = b.conv3d(x, 16, 1, padding="same")
        cat = b.concat([a, c, d])
        y = b.elu(b.conv3d(c, 24, 3, padding="same"))
        return b.concat([b.conv3d(cat, 8, 1, padding="same"), y])

    cfg, weights = _net(16, build, seed=9)
    frames = _frames(4, 16, 4)
    got, labels = _check(cfg, weights, frames)
    monkeypatch.setenv("TH_WFUSED", "0")
    ref, rl = _run(cfg, weights, frames)
    assert not any("k_conv_wf" in l for l in rl), rl
    assert float(np.abs(got - ref).max()) <= 4e-5 * max(1.0, float(np.abs(ref).max()))


@pytest.mark.parametrize("first", [False, True])
def test_wfused_chunk_blocked_input_is_bit_identical_to_channels_last(gpu, monkeypatch, first):
    """A tensor written by one pointwise (or first-layer) step and read only by a conv_wfused step is stored chunk-blocked
    ([C/4][voxel][4], TView::blk): the producer's stores and the consumer's slice loads change, no arithmetic does — the
    outputs equal the channels-last plan's (TH_WF_NOBLK=1) bit for bit, ragged chunks included."""
    if first:
        shape, cin = (21, 21, 21), 6

        def build(b, x):                                             # TIMED's first two blocks
            x = b.maxpool(b.batchnorm(b.elu(b.conv3d(x, 32, 3, padding="same"))), 2)
            return b.maxpool(b.batchnorm(b.elu(b.conv3d(x, 48, 3, padding="same"))), 2)
    else:
        shape, cin = SHAPE, 24

        def build(b, x):                                             # DenseCPD bottleneck -> growth convolution
            y = b.conv3d(b.relu(b.batchnorm(x)), 64, 1, padding="same", use_bias=False)
            return b.conv3d(b.relu(b.batchnorm(y)), 16, 3, padding="same", use_bias=False)

    b = synth.KerasGraphBuilder((*shape, cin), seed=77, bias_std=0.3)
    cfg, weights = b.finish(b.flatten(build(b, b.input_name)))
    rng = np.random.default_rng(5)
    frames = (rng.standard_normal((11, *shape, cin)) * (rng.random((11, *shape, cin)) < 0.5)).astype(np.float32)
    want = cnn_oracle.forward(cfg, weights, frames[:3], np.float32)
    got, labels = _run(cfg, weights, frames)
    assert sum("output chunk-blocked" in l for l in labels) == 1 and sum("input chunk-blocked" in l for l in labels) == 1, labels
    assert float(np.abs(got[:3] - want).max()) <= 2e-5 * max(1.0, float(np.abs(want).max()))
    got3, _ = _run(cfg, weights, frames, chunk=3)
    assert np.array_equal(got, got3)
    monkeypatch.setenv("TH_WF_NOBLK", "1")
    ref, rl = _run(cfg, weights, frames)
    assert not any("chunk-blocked" in l for l in rl), rl
    assert np.array_equal(got, ref)


def _random_block_net(seed):
    """a random two- or three-convolution net on 10^3 volumes around conv_wfused: random widths (multiples of 4 in, anything
    out), pointwise or 3x3x3 producers, pre-activation or post-activation chains, optional pool, optional concat"""
    rng = np.random.default_rng(seed)
    cin = int(rng.choice([16, 20, 24, 32, 48]))
    b = synth.KerasGraphBuilder((*SHAPE, cin), seed=seed, bias_std=0.2)
    x = b.input_name
    act = lambda t: [b.relu, b.elu, lambda u: b.leaky_relu(u, 0.2)][int(rng.integers(0, 3))](t)
    for _ in range(int(rng.integers(1, 3))):
        mid = int(rng.choice([16, 32, 40, 64]))
        if rng.random() < 0.5:                                       # DenseNet-style: BN -> act -> 1x1x1 -> BN -> act -> 3x3x3, concat
            y = b.conv3d(act(b.batchnorm(x)), mid, 1, padding="same", use_bias=False)
            y = b.conv3d(act(b.batchnorm(y)), int(rng.choice([8, 16, 20])), 3, padding="same", use_bias=bool(rng.integers(0, 2)))
            x = b.concat([x, y])
        else:                                                        # TIMED-style: 3x3x3 -> act -> BN
            x = b.batchnorm(act(b.conv3d(x, mid, 3, padding="same")))
    if rng.random() < 0.5:
        x = (b.maxpool if rng.random() < 0.5 else b.avgpool)(b.batchnorm(act(b.conv3d(x, int(rng.choice([12, 32, 36])), 3, padding="same"))), 2)
    cfg, w = b.finish(b.flatten(x))
    n = int(rng.integers(1, 12))
    frames = (rng.standard_normal((n, *SHAPE, cin)) * (rng.random((n, *SHAPE, cin)) < 0.5)).astype(np.float32)
    return cfg, w, frames


@pytest.mark.parametrize("seed", range(10))
def test_wfused_random_blocks(gpu, seed):
    cfg, w, frames = _random_block_net(seed)
    want = cnn_oracle.forward(cfg, w, frames, np.float32)
    got, labels = _run(cfg, w, frames, chunk=1 + seed % 4)
    assert any("k_conv_wf" in l for l in labels), labels
    assert float(np.abs(got - want).max()) <= 2e-5 * max(1.0, float(np.abs(want).max())), labels
