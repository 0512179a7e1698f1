"""Per-element parity of the split-operand form of the fused Winograd layer (csrc/conv_wfsplit.hip: the algorithm of
conv_wfused.hip — F(2,3) x F(2,3) in-plane, z taps direct — with every product as six bf16 piece products of exactly split
operands on v_mfma_f32_32x32x16_bf16, fp32 accumulation) against the CPU oracle and against the fp32-input kernel it replaces
(TH_WF_SPLIT=0): every pooling mode, one and two passes of 64 output channels, 1..3 phases of 16 input channels, column counts
that are not multiples of 32, persistent workgroups that walk several frames (the slice of the next phase is loaded under the
last step of this one), chunk-blocked and channels-last inputs, twenty decades of input magnitude.  Serves reference
predict.py:142 (TIMED's conv3d_1)."""
import numpy as np
import pytest

from oracle import cnn_oracle
from timed_hip import engine, synth

pytestmark = pytest.mark.gpu

SHAPE = (10, 10, 10)


def _net(cin, build, seed=0, bias_std=0.3):
    b = synth.KerasGraphBuilder((*SHAPE, cin), seed=seed, bias_std=bias_std)
    x = build(b, b.input_name)
    x = b.flatten(x)
    return b.finish(x)


def _frames(n, cin, seed, scale=1.0):
    rng = np.random.default_rng(seed)
    return (scale * rng.standard_normal((n, *SHAPE, cin)) * (rng.random((n, *SHAPE, cin)) < 0.5)).astype(np.float32)


def _run(cfg, weights, frames, chunk=None):
    m = engine.HipFrameModel.from_keras(cfg, weights)
    if chunk:
        m.set_chunk(chunk)
    got = m.predict(frames)
    labels = [s["label"] for s in m.steps()]
    m.close()
    return got, labels


POSTS = {
    "none": lambda b, x: x,
    "relu": lambda b, x: b.relu(x),
    "elu_bn": lambda b, x: b.batchnorm(b.elu(x)),
    "tanh": lambda b, x: b.activation(x, "tanh"),
}

# (cin, cout, post, pool, n_frames)
CASES = [
    (32, 64, "elu_bn", "max", 3),       # TIMED conv3d_1: two phases, pool before the monotone chain
    (16, 33, "none", None, 1),          # one phase, the second column tile holds ONE real channel, no pool
    (48, 40, "relu", "avg", 9),         # three phases, 9 frames: a second, ragged deal over the XCDs
    (32, 100, "tanh", "max", 4),        # two passes of 64 columns, non-monotone chain before the max pool
    (16, 64, "elu_bn", "avg", 10),
    (64, 48, "relu", None, 2),          # four phases, unpooled stores
]


def _build(cin, cout, post, pool):
    def build(b, x):
        x = b.conv3d(x, cout, 3, padding="same", use_bias=(post != "none"))
        x = POSTS[post](b, x)
        if pool == "max":
            x = b.maxpool(x, 2)
        elif pool == "avg":
            x = b.avgpool(x, 2)
        return x
    return build


@pytest.mark.parametrize("cin,cout,post,pool,n", CASES)
def test_split_layer_per_element(gpu, monkeypatch, cin, cout, post, pool, n):
    cfg, weights = _net(cin, _build(cin, cout, post, pool), seed=(cin * 131 + cout) % 997)
    frames = _frames(n, cin, seed=n)
    want = cnn_oracle.forward(cfg, weights, frames, np.float64)
    got, labels = _run(cfg, weights, frames)
    assert sum("k_conv_wfs<" in l and "bf16x3" in l for l in labels) == 1, labels
    scale = max(1.0, float(np.abs(want).max()))
    err = float(np.abs(got - want).max())
    assert got.shape == want.shape and err <= 2e-5 * scale, (err, scale)
    got2, _ = _run(cfg, weights, frames, chunk=2)                   # ragged chunks: bit-identical
    assert np.array_equal(got, got2)
    monkeypatch.setenv("TH_WF_SPLIT", "0")                          # the fp32-input kernel of the same algorithm
    ref, rl = _run(cfg, weights, frames)
    assert not any("k_conv_wfs<" in l for l in rl) and any("k_conv_wf<" in l for l in rl), rl
    assert float(np.abs(got - ref).max()) <= 4e-6 * scale
    # the split costs no accuracy against float64: within 1.5x of the fp32-product kernel's own distance (+ one fp32 ulp of the scale)
    err_ref = float(np.abs(ref - want).max())
    assert err <= 1.5 * err_ref + 2e-7 * scale, (err, err_ref)


def test_split_layer_persistent_workgroups_walk_several_frames(gpu, monkeypatch):
    """8 resident workgroups, 37 frames: workgroups run 4 or 5 frames back to back; results equal the one-frame-per-workgroup run's
    bit for bit (no state leaks from a frame into the next: accumulators, the V ring, the slice that is loaded a phase ahead)"""
    cfg, weights = _net(32, _build(32, 64, "elu_bn", "max"), seed=11)
    frames = _frames(37, 32, 5)
    base, labels = _run(cfg, weights, frames)
    assert any("k_conv_wfs<" in l for l in labels)
    monkeypatch.setenv("TH_WF_RESIDENT", "8")
    got, _ = _run(cfg, weights, frames)
    assert np.array_equal(got, base)
    want = cnn_oracle.forward(cfg, weights, frames, np.float32)
    assert float(np.abs(got - want).max()) <= 2e-5 * max(1.0, float(np.abs(want).max()))
    # every frame on its own and all in another order: a frame's result does not depend on its neighbours
    perm = np.random.default_rng(0).permutation(37)
    got_p, _ = _run(cfg, weights, frames[perm])
    assert np.array_equal(got_p, base[perm])


def test_split_layer_behind_the_first_layer_blocked_and_channels_last(gpu, monkeypatch):
    """The TIMED opening — first layer (21^3 x 6 -> 32, pooled to 10^3) into the 32 -> 64 layer: the tensor between them is
    chunk-blocked by default and channels-last under TH_WF_NOBLK=1; the split layer reads either, same bits"""
    b = synth.KerasGraphBuilder((21, 21, 21, 6), seed=3, bias_std=0.2)
    x = b.maxpool(b.batchnorm(b.elu(b.conv3d(b.input_name, 32, 3, padding="same"))), 2)
    x = b.maxpool(b.batchnorm(b.elu(b.conv3d(x, 64, 3, padding="same"))), 2)
    cfg, weights = b.finish(b.flatten(x))
    frames = synth.synthetic_frames(5, seed=8)
    want = cnn_oracle.forward(cfg, weights, frames, np.float64)
    got, labels = _run(cfg, weights, frames)
    assert any("k_conv_wfs<" in l and "chunk-blocked" in l for l in labels), labels
    scale = max(1.0, float(np.abs(want).max()))
    assert float(np.abs(got - want).max()) <= 2e-5 * scale
    monkeypatch.setenv("TH_WF_NOBLK", "1")
    got2, labels2 = _run(cfg, weights, frames)
    assert any("k_conv_wfs<" in l for l in labels2) and not any("chunk-blocked" in l for l in labels2), labels2
    assert np.array_equal(got, got2)


def test_split_layer_over_twenty_decades_of_input_magnitude(gpu):
    """bf16 has fp32's exponent range: the exact three-piece split needs no scaling — inputs of magnitude 1e-10 .. 1e10 keep the
    relative bound (a linear layer: no epilogue, no bias)"""
    cfg, weights = _net(32, _build(32, 64, "none", None), seed=2, bias_std=0.0)
    for mag in (1e-10, 1e-3, 1.0, 1e4, 1e10):
        frames = _frames(2, 32, 9, scale=mag)
        want = cnn_oracle.forward(cfg, weights, frames, np.float64)
        got, _ = _run(cfg, weights, frames)
        assert float(np.abs(got - want).max()) <= 2e-5 * float(np.abs(want).max()), mag


def test_layers_the_split_form_does_not_serve_stay_on_the_fp32_kernel(gpu):
    """an input prologue (DenseCPD's BN -> ReLU in front of the growth convolution), Cin not a multiple of 16, or at most 32 output
    channels: k_conv_wf as before"""
    def dense(b, x):                       # BN -> ReLU on the model INPUT: no producer to fold it into, it stays this layer's prologue
        return b.conv3d(b.relu(b.batchnorm(x)), 48, 3, padding="same")
    for cin, build in ((16, dense), (20, _build(20, 40, "relu", None)), (32, _build(32, 32, "relu", None))):
        cfg, weights = _net(cin, build, seed=4)
        frames = _frames(2, cin, 3)
        want = cnn_oracle.forward(cfg, weights, frames, np.float32)
        got, labels = _run(cfg, weights, frames)
        assert any("k_conv_wf<" in l for l in labels) and not any("k_conv_wfs<" in l for l in labels), labels
        assert float(np.abs(got - want).max()) <= 2e-5 * max(1.0, float(np.abs(want).max()))
