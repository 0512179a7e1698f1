"""Synthetic benchmark topologies + weights, emitted in Keras ``model_config`` form.

No released ``.h5`` ships with the reference (reference README.md:100-104 points at the GitHub
release page) and there is no network, so BASELINE.json's configs run on the stand-in topologies
SURVEY.md §8(d) defines from reference README.md:252-258 + img/timed_architecture.png.  They are
produced as ordinary Keras Functional ``model_config`` dicts so that they travel through exactly
the converter a real ``.h5`` would (keras_config.parse_keras_model -> pack -> HIP runtime).

Weights: He-normal conv/dense kernels, zero bias unless ``bias_std`` is given,
BN gamma~U(0.5,1.5), beta~N(0,0.1), mean~N(0,0.1), var~U(0.5,1.5); numpy PCG64, default seed 4321.
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import numpy as np


class KerasGraphBuilder:
    """Tiny helper that writes Keras-Functional-style layer records and random weights."""

    def __init__(self, input_shape, seed: int = 4321, bias_std: float = 0.0, name: str = "model"):
        self.rng = np.random.default_rng(seed)
        self.bias_std = bias_std
        self.name = name
        self.layers: List[dict] = []
        self.weights: Dict[str, List[np.ndarray]] = {}
        self.shapes: Dict[str, tuple] = {}
        self._count: Dict[str, int] = {}
        self.input_name = self._emit("InputLayer", "input",
                                     dict(batch_input_shape=[None, *input_shape], dtype="float32", sparse=False,
                                          ragged=False), [], tuple(input_shape))

    # -- plumbing ---------------------------------------------------------------------------
    def _uname(self, base: str) -> str:
        n = self._count.get(base, 0)
        self._count[base] = n + 1
        return base if n == 0 else f"{base}_{n}"

    def _emit(self, cls: str, base: str, cfg: dict, inputs: List[str], shape: tuple) -> str:
        name = self._uname(base)
        cfg = dict(cfg, name=name)
        self.layers.append(dict(class_name=cls, config=cfg, name=name,
                                inbound_nodes=[[[i, 0, 0, {}] for i in inputs]] if inputs else []))
        self.shapes[name] = shape
        return name

    @staticmethod
    def _out(n, k, s, same):
        return -(-n // s) if same else (n - k) // s + 1

    # -- layers -----------------------------------------------------------------------------
    def conv3d(self, x, filters, k=3, strides=1, padding="same", activation="linear", use_bias=True, dilation_rate=1):
        d, h, w, cin = self.shapes[x]
        same = padding == "same"
        ks = (k, k, k) if isinstance(k, int) else tuple(k)   # anisotropic kernels allowed: k=(3, 1, 3)
        st = (strides,) * 3 if isinstance(strides, int) else tuple(strides)
        dl = (dilation_rate,) * 3 if isinstance(dilation_rate, int) else tuple(dilation_rate)
        fan_in = ks[0] * ks[1] * ks[2] * cin
        kern = (self.rng.standard_normal((*ks, cin, filters)) * np.sqrt(2.0 / fan_in)).astype(np.float32)
        ws = [kern]
        if use_bias:
            ws.append((self.rng.standard_normal(filters) * self.bias_std).astype(np.float32))
        shape = tuple(self._out(n, (kk - 1) * dd + 1, ss, same) for n, kk, ss, dd in zip((d, h, w), ks, st, dl)) + (filters,)
        name = self._emit("Conv3D", "conv3d",
                          dict(filters=filters, kernel_size=list(ks), strides=list(st), padding=padding,
                               data_format="channels_last", dilation_rate=list(dl), groups=1,
                               activation=activation or "linear", use_bias=use_bias, trainable=True, dtype="float32"),
                          [x], shape)
        self.weights[name] = ws
        return name

    def dense(self, x, units, activation="linear", use_bias=True):
        (fin,) = self.shapes[x]
        kern = (self.rng.standard_normal((fin, units)) * np.sqrt(2.0 / fin)).astype(np.float32)
        ws = [kern]
        if use_bias:
            ws.append((self.rng.standard_normal(units) * self.bias_std).astype(np.float32))
        name = self._emit("Dense", "dense", dict(units=units, activation=activation, use_bias=use_bias), [x], (units,))
        self.weights[name] = ws
        return name

    def batchnorm(self, x, epsilon=1e-3, center=True, scale=True):
        shp = self.shapes[x]
        c = shp[-1]
        ws = []
        if scale:
            ws.append(self.rng.uniform(0.5, 1.5, c).astype(np.float32))
        if center:
            ws.append((self.rng.standard_normal(c) * 0.1).astype(np.float32))
        ws.append((self.rng.standard_normal(c) * 0.1).astype(np.float32))
        ws.append(self.rng.uniform(0.5, 1.5, c).astype(np.float32))
        name = self._emit("BatchNormalization", "batch_normalization",
                          dict(axis=[len(shp)], momentum=0.99, epsilon=epsilon, center=center, scale=scale), [x], shp)
        self.weights[name] = ws
        return name

    def elu(self, x, alpha=1.0):
        return self._emit("ELU", "elu", dict(alpha=alpha), [x], self.shapes[x])

    def relu(self, x):
        return self._emit("ReLU", "re_lu", dict(max_value=None, negative_slope=0.0, threshold=0.0), [x], self.shapes[x])

    def leaky_relu(self, x, alpha=0.3):
        return self._emit("LeakyReLU", "leaky_re_lu", dict(alpha=alpha), [x], self.shapes[x])

    def activation(self, x, act):
        return self._emit("Activation", "activation", dict(activation=act), [x], self.shapes[x])

    def softmax(self, x):
        return self._emit("Softmax", "softmax", dict(axis=-1), [x], self.shapes[x])

    def _pool(self, cls, base, x, size, strides, padding):
        d, h, w, c = self.shapes[x]
        strides = size if strides is None else strides
        same = padding == "same"
        shape = (self._out(d, size, strides, same), self._out(h, size, strides, same), self._out(w, size, strides, same), c)
        return self._emit(cls, base, dict(pool_size=[size] * 3, strides=[strides] * 3, padding=padding,
                                          data_format="channels_last"), [x], shape)

    def maxpool(self, x, size=2, strides=None, padding="valid"):
        return self._pool("MaxPooling3D", "max_pooling3d", x, size, strides, padding)

    def avgpool(self, x, size=2, strides=None, padding="valid"):
        return self._pool("AveragePooling3D", "average_pooling3d", x, size, strides, padding)

    def spatial_dropout(self, x, rate=0.2):
        return self._emit("SpatialDropout3D", "spatial_dropout3d", dict(rate=rate), [x], self.shapes[x])

    def dropout(self, x, rate=0.2):
        return self._emit("Dropout", "dropout", dict(rate=rate), [x], self.shapes[x])

    def gap(self, x):
        return self._emit("GlobalAveragePooling3D", "global_average_pooling3d",
                          dict(data_format="channels_last", keepdims=False), [x], (self.shapes[x][-1],))

    def gmp(self, x):
        return self._emit("GlobalMaxPooling3D", "global_max_pooling3d",
                          dict(data_format="channels_last", keepdims=False), [x], (self.shapes[x][-1],))

    def flatten(self, x):
        return self._emit("Flatten", "flatten", dict(data_format="channels_last"), [x],
                          (int(np.prod(self.shapes[x])),))

    def concat(self, xs):
        s0 = self.shapes[xs[0]]
        return self._emit("Concatenate", "concatenate", dict(axis=-1), list(xs),
                          s0[:-1] + (sum(self.shapes[x][-1] for x in xs),))

    def add(self, xs):
        return self._emit("Add", "add", dict(), list(xs), self.shapes[xs[0]])

    def finish(self, out) -> Tuple[dict, Dict[str, List[np.ndarray]]]:
        cfg = dict(class_name="Functional",
                   config=dict(name=self.name, layers=self.layers,
                               input_layers=[[self.input_name, 0, 0]], output_layers=[[out, 0, 0]]),
                   keras_version="2.13.1", backend="tensorflow")
        return cfg, self.weights


# ---- the three benchmark topologies ----------------------------------------
def timed_synth(n_classes: int = 20, in_channels: int = 6, side: int = 21, seed: int = 4321,
                widths=(32, 64, 128, 128, 256), bias_std: float = 0.0):
    """TIMED-synth: blocks of Conv3D(3,'same') -> ELU -> BN (reference README.md:254); MaxPool after
    blocks 1 and 2; SpatialDropout after block 1 and the last block; last conv has n_classes filters;
    GlobalAveragePooling3D -> Softmax.  628.2 MFLOP/frame at the default widths, 20 classes."""
    b = KerasGraphBuilder((side, side, side, in_channels), seed=seed, bias_std=bias_std,
                          name=f"timed_synth_{n_classes}")
    x = b.input_name
    chans = list(widths) + [n_classes]
    for i, c in enumerate(chans):
        x = b.conv3d(x, c, 3, padding="same")
        x = b.elu(x)
        x = b.batchnorm(x)
        if i in (0, 1):
            x = b.maxpool(x, 2)
        if i == 0 or i == len(chans) - 1:
            x = b.spatial_dropout(x)
    x = b.gap(x)
    x = b.softmax(x)
    return b.finish(x)


def densecpd_synth(n_classes: int = 20, in_channels: int = 6, side: int = 21, seed: int = 4321,
                   growth: int = 16, n_blocks: int = 3, layers_per_block: int = 4, bottleneck: int = 64,
                   stem: int = 32, bias_std: float = 0.0):
    """DenseCPD-synth: stem Conv3D(3,'same')+BN+ReLU, MaxPool(2); dense blocks of
    BN->ReLU->Conv1^3(->bottleneck)->BN->ReLU->Conv3^3('same',->growth)->Concat; transitions
    BN->ReLU->Conv1^3(C->C/2)->AvgPool(2); BN->ReLU->GAP->Dense(n_classes)->Softmax."""
    b = KerasGraphBuilder((side, side, side, in_channels), seed=seed, bias_std=bias_std,
                          name=f"densecpd_synth_{n_classes}")
    x = b.conv3d(b.input_name, stem, 3, padding="same")
    x = b.batchnorm(x)
    x = b.relu(x)
    x = b.maxpool(x, 2)
    for blk in range(n_blocks):
        for _ in range(layers_per_block):
            y = b.batchnorm(x)
            y = b.relu(y)
            y = b.conv3d(y, bottleneck, 1, padding="same", use_bias=False)
            y = b.batchnorm(y)
            y = b.relu(y)
            y = b.conv3d(y, growth, 3, padding="same", use_bias=False)
            x = b.concat([x, y])
        if blk != n_blocks - 1:
            c = b.shapes[x][-1]
            y = b.batchnorm(x)
            y = b.relu(y)
            y = b.conv3d(y, c // 2, 1, padding="same", use_bias=False)
            x = b.avgpool(y, 2)
    x = b.batchnorm(x)
    x = b.relu(x)
    x = b.gap(x)
    x = b.dense(x, n_classes)
    x = b.softmax(x)
    return b.finish(x)


def prodconn_synth(n_classes: int = 20, in_channels: int = 6, side: int = 21, seed: int = 4321,
                   bias_std: float = 0.0):
    """ProDCoNN-like stand-in: parallel 3^3 / 5^3 / valid-padded branches concatenated, strided conv,
    Flatten -> Dense(relu) -> Dense(softmax).  Exercises Concatenate of branches, stride 2, 'valid'
    padding, k=5, Flatten ordering and fused Dense activations."""
    b = KerasGraphBuilder((side, side, side, in_channels), seed=seed, bias_std=bias_std,
                          name=f"prodconn_synth_{n_classes}")
    a = b.conv3d(b.input_name, 16, 3, padding="same", activation="relu")
    c = b.conv3d(b.input_name, 16, 5, padding="same", activation="relu")
    x = b.concat([a, c])
    x = b.batchnorm(x)
    x = b.maxpool(x, 2)
    x = b.conv3d(x, 48, 3, strides=2, padding="same", activation="elu")
    x = b.conv3d(x, 64, 3, padding="valid")
    x = b.leaky_relu(x, 0.1)
    x = b.dropout(x)
    x = b.flatten(x)
    x = b.dense(x, 96, activation="relu")
    x = b.dense(x, n_classes, activation="softmax")
    return b.finish(x)


def padding_zoo_synth(n_classes: int = 20, in_channels: int = 4, side: int = 11, seed: int = 4321, bias_std: float = 0.1):
    """Every padding rule of SURVEY Appendix A in one small net (a parity-test case, not a benchmark): 'same' with odd,
    anisotropic and EVEN kernels, stride-2 'same' (asymmetric: the extra row goes after), 'same' average / max
    pooling (padding excluded from the average's divisor), a Conv3D with activation='leaky_relu' (slope 0.2) and a
    LeakyReLU layer with its own alpha."""
    b = KerasGraphBuilder((side, side, side, in_channels), seed=seed, bias_std=bias_std, name="padding_zoo")
    x = b.conv3d(b.input_name, 8, 3, padding="same", activation="leaky_relu")
    x = b.conv3d(x, 8, (1, 3, 5), padding="same")
    x = b.leaky_relu(x, 0.1)
    x = b.conv3d(x, 12, 4, padding="same", activation="relu")
    x = b.avgpool(x, 2, 2, padding="same")
    x = b.conv3d(x, 12, 3, strides=2, padding="same", activation="elu")
    x = b.avgpool(x, 3, 1, padding="same")
    x = b.maxpool(x, 3, 2, padding="same")
    x = b.conv3d(x, 16, 2, padding="same")
    x = b.flatten(x)
    x = b.dense(x, n_classes)
    x = b.softmax(x)
    return b.finish(x)


TOPOLOGIES = {
    "timed": lambda **kw: timed_synth(20, **kw),
    "timed_rotamer": lambda **kw: timed_synth(338, **kw),
    "densecpd": lambda **kw: densecpd_synth(20, **kw),
    "prodconn": lambda **kw: prodconn_synth(20, **kw),
}


def synthetic_frames(n: int, side: int = 21, channels: int = 6, seed: int = 1234, gaussian: bool = True,
                     atoms: int = 200) -> np.ndarray:
    """Synthetic residue frames: per frame ``atoms`` centres uniformly in the cube, a
    uniform channel each, splat a 3x3x3 Gaussian (sigma 0.6 voxel, peak 1.0), clip to [0,1]  —
    mimics aposteriori ``voxels_as_gaussian=True``.  ``gaussian=False`` gives the boolean variant
    (one voxel per atom, dtype uint8) that models ``voxels_as_gaussian=False`` (reference
    design_utils/utils.py:518-521)."""
    rng = np.random.default_rng(seed)
    if gaussian:
        x = np.zeros((n, side, side, side, channels), dtype=np.float32)
        g = np.exp(-0.5 * (np.arange(-1, 2) / 0.6) ** 2)
        k3 = (g[:, None, None] * g[None, :, None] * g[None, None, :]).astype(np.float32)
    else:
        x = np.zeros((n, side, side, side, channels), dtype=np.uint8)
    for f in range(n):
        pos = rng.integers(0, side, size=(atoms, 3))
        ch = rng.integers(0, channels, size=atoms)
        if not gaussian:
            x[f, pos[:, 0], pos[:, 1], pos[:, 2], ch] = 1
            continue
        for (z, y, w), c in zip(pos, ch):
            z0, z1 = max(z - 1, 0), min(z + 2, side)
            y0, y1 = max(y - 1, 0), min(y + 2, side)
            w0, w1 = max(w - 1, 0), min(w + 2, side)
            x[f, z0:z1, y0:y1, w0:w1, c] += k3[z0 - z + 1:z1 - z + 1, y0 - y + 1:y1 - y + 1, w0 - w + 1:w1 - w + 1]
    if gaussian:
        np.clip(x, 0.0, 1.0, out=x)
    return x
