"""Seeded random-graph parity: small random Functional graphs (conv kernels 1/3/anisotropic, same/valid,
pre- and post-activation chains, BatchNorm, pools, Concatenate/Add branches, narrow and 17..20-channel
convolutions, GAP/Flatten+Dense heads) through the HIP engine vs the CPU oracle.  Every case is derived
from its seed only, so a failure names a reproducible graph.  Tolerance as in test_gpu_conv_sweep.py."""
import numpy as np
import pytest

from oracle import cnn_oracle
from timed_hip import _lib, engine, synth

pytestmark = pytest.mark.gpu


def _random_net(seed, dense_tail=False):
    rng = np.random.default_rng(1000 + seed)
    shape = tuple(int(rng.integers(3, 9)) for _ in range(3))
    cin = int(rng.choice([1, 3, 5, 6, 8]))
    b = synth.KerasGraphBuilder((*shape, cin), seed=seed, bias_std=0.2)
    x = b.input_name

    def spatial(t):
        return b.shapes[t][:3]

    def post(t):
        kind = rng.integers(0, 5)
        if kind == 0:
            return b.batchnorm(b.elu(t))
        if kind == 1:
            return b.relu(b.batchnorm(t))
        if kind == 2:
            return b.leaky_relu(t, float(rng.choice([0.1, 0.3])))
        if kind == 3:
            return b.relu(t)
        return t

    def conv(t, cout=None, k=None):
        cout = cout or int(rng.choice([4, 7, 12, 16, 18, 20, 24, 32, 40, 64]))
        k = k or [1, 3, 3, 3, (3, 1, 3), (1, 3, 3)][int(rng.integers(0, 6))]
        s = spatial(t)
        ks = (k, k, k) if isinstance(k, int) else k
        padding = "valid" if (rng.random() < 0.25 and all(s[i] - ks[i] + 1 >= 2 for i in range(3))) else "same"
        if rng.random() < 0.3:   # pre-activation form
            t = b.relu(b.batchnorm(t))
        return b.conv3d(t, cout, k, padding=padding, use_bias=bool(rng.random() < 0.7))

    n_blocks = int(rng.integers(2, 5))
    for _ in range(n_blocks):
        form = rng.integers(0, 4)
        if form == 0:      # plain conv block, maybe pooled
            x = post(conv(x))
            if min(spatial(x)) >= 4 and rng.random() < 0.5:
                x = b.maxpool(x, 2) if rng.random() < 0.5 else b.avgpool(x, 2)
        elif form == 1:    # dense-layer step: bottleneck 1x1 -> 3x3 growth -> concat
            y = b.conv3d(b.relu(b.batchnorm(x)), int(rng.choice([16, 32])), 1, padding="same", use_bias=False)
            y = b.conv3d(b.relu(b.batchnorm(y)), int(rng.choice([8, 12, 16])), 3, padding="same", use_bias=False)
            x = b.concat([x, y])
        elif form == 2:    # two branches added
            c = int(rng.choice([8, 16, 20]))
            a1 = post(b.conv3d(x, c, 3, padding="same"))
            a2 = b.conv3d(x, c, 1, padding="same", activation="relu")
            x = b.add([a1, a2])
        elif form == 3 and rng.random() < 0.5:   # Inception/ProDCoNN-style: parallel branches, concat, then BN (+ pool)
            c1, c2 = int(rng.choice([8, 16, 24])), int(rng.choice([8, 16]))
            a1 = b.conv3d(x, c1, 3, padding="same", activation="relu")
            a2 = b.conv3d(x, c2, int(rng.choice([1, 3, 5])) if min(spatial(x)) >= 5 else 3, padding="same", activation="relu")
            x = b.batchnorm(b.concat([a1, a2]))
            if min(spatial(x)) >= 4 and rng.random() < 0.6:
                x = b.maxpool(x, 2)
            if min(spatial(x)) >= 3 and rng.random() < 0.5:   # strided down-sampling convolution
                x = b.conv3d(x, int(rng.choice([16, 32, 48])), 3, strides=2, padding=str(rng.choice(["same", "valid"])),
                             activation="elu")
        else:              # transition: 1x1 conv then pool
            x = b.conv3d(b.relu(b.batchnorm(x)), max(4, b.shapes[x][3] // 2), 1, padding="same", use_bias=False)
            if min(spatial(x)) >= 2:
                x = b.avgpool(x, 2)
    head = rng.integers(0, 3)
    ncls = int(rng.choice([20, 338, 5]))
    if dense_tail:         # DenseCPD's head: [BN / activation]* -> GlobalAveragePooling3D -> Dense -> Softmax (the random body's draws are unchanged)
        for c in ("", "b", "br", "eb", "rb", "blrb")[seed % 6]:
            x = {"b": b.batchnorm, "r": b.relu, "e": b.elu, "l": lambda v: b.leaky_relu(v, 0.1)}[c](x)
        x = b.softmax(b.dense(b.gap(x), ncls, use_bias=bool(seed % 2)))
    elif head == 0:
        x = b.softmax(b.gap(b.conv3d(x, ncls, 3, padding="same")))
    elif head == 1:
        x = b.dense(b.flatten(x), ncls, activation="softmax")
    else:
        x = b.flatten(x)
    cfg, weights = b.finish(x)
    n = int(rng.integers(1, 12))
    frames = (rng.standard_normal((n, *shape, cin)) * (rng.random((n, *shape, cin)) < 0.6)).astype(np.float32)
    return cfg, weights, frames


@pytest.mark.parametrize("seed", range(48))
def test_random_graph_matches_oracle(gpu, seed):
    cfg, weights, frames = _random_net(seed)
    want = cnn_oracle.forward(cfg, weights, frames, np.float32)
    m = engine.HipFrameModel.from_keras(cfg, weights)
    m.set_chunk(int(1 + seed % 5))           # ragged chunks too
    got = m.predict(frames)
    labels = [s["label"] for s in m.steps()]
    m.close()
    scale = max(1.0, float(np.abs(want).max()))
    err = float(np.abs(got - want).max())
    assert got.shape == want.shape and err <= 2e-5 * scale, (seed, err, scale, labels)
    if want.ndim == 2 and np.allclose(want.sum(1), 1.0, atol=1e-5):   # classifier heads: identical argmax unless tied
        top2 = np.sort(want, axis=1)[:, -2:]
        clear = (top2[:, 1] - top2[:, 0]) > 1e-4
        assert np.array_equal(got.argmax(1)[clear], want.argmax(1)[clear]), (seed, labels)


@pytest.mark.parametrize("seed", range(100, 118))
def test_random_graph_with_a_dense_tail(gpu, seed):
    """random bodies (dense blocks, strided / valid convolutions, transitions: whatever tensor they end in — a concat arena, a pooled
    tensor, a plain one) under DenseCPD's head; the tail runs as ONE launch (k_tail_dense) and agrees with the oracle like every
    other plan"""
    cfg, weights, frames = _random_net(seed, dense_tail=True)
    want = cnn_oracle.forward(cfg, weights, frames, np.float32)
    m = engine.HipFrameModel.from_keras(cfg, weights)
    m.set_chunk(int(1 + seed % 5))
    got = m.predict(frames)
    labels = [s["label"] for s in m.steps()]
    m.close()
    # (a Winograd layer in front of the pooling pools in its output transform instead: then Dense and Softmax keep their steps)
    assert sum("k_tail_dense" in l for l in labels) == 1 or any("wino_out + global_avg_pool" in l for l in labels), (seed, labels)
    err = float(np.abs(got - want).max())
    assert got.shape == want.shape and err <= 2e-5, (seed, err, labels)
    top2 = np.sort(want, axis=1)[:, -2:]
    clear = (top2[:, 1] - top2[:, 0]) > 1e-4
    assert np.array_equal(got.argmax(1)[clear], want.argmax(1)[clear]), (seed, labels)


@pytest.mark.parametrize("seed", [3, 11, 19, 27])
def test_random_graph_generic_path_agrees(gpu, seed):
    """the same graphs with the MFMA kernels disabled (direct convolution, unfused) — bisecting aid"""
    cfg, weights, frames = _random_net(seed)
    want = cnn_oracle.forward(cfg, weights, frames, np.float32)
    got = engine.HipFrameModel.from_keras(cfg, weights, flags=_lib.TH_LOAD_NO_MFMA | _lib.TH_LOAD_NO_FUSE).predict(frames)
    assert float(np.abs(got - want).max()) <= 2e-5 * max(1.0, float(np.abs(want).max()))
