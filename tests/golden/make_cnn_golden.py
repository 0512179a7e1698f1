#!/usr/bin/env python
"""Generate tests/golden/cnn_golden.npz — runs ONLY in the build container (needs torch CPU).

TensorFlow/Keras cannot be imported anywhere in this project, so the CNN oracle
(oracle/cnn_oracle.py) is pinned against an INDEPENDENT implementation instead: the same Keras
``model_config`` graphs are evaluated here with torch.nn.functional CPU ops (conv3d, max_pool3d,
avg_pool3d, batch_norm, elu, ...) written from the Keras layer documentation, not from the oracle.
The fixture stores, per case: the topology name/kwargs, frame seed, and the torch fp32 and fp64
outputs.  Inputs/weights are regenerated from their seeds at test time (numpy PCG64 is
deterministic), so the fixture stays a few KB.

Usage:  python tests/golden/make_cnn_golden.py
"""
import json
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "timed-design_amd"))
sys.path.insert(0, ROOT)

from timed_hip import synth  # noqa: E402

CASES = [
    # name, builder, kwargs, n_frames, frame kwargs
    ("timed20", "timed_synth", dict(n_classes=20), 2, dict(seed=1234)),
    ("timed338", "timed_synth", dict(n_classes=338), 2, dict(seed=1235)),
    ("timed20_c5_bias", "timed_synth", dict(n_classes=20, in_channels=5, bias_std=0.1, seed=7), 2,
     dict(seed=11, channels=5)),
    ("timed20_bool", "timed_synth", dict(n_classes=20, seed=99), 2, dict(seed=12, gaussian=False)),
    ("densecpd20", "densecpd_synth", dict(n_classes=20), 2, dict(seed=1236)),
    ("prodconn20", "prodconn_synth", dict(n_classes=20, bias_std=0.05), 2, dict(seed=1237)),
    ("timed_small", "timed_synth", dict(n_classes=20, widths=(8, 16, 16), side=9, in_channels=4, bias_std=0.2), 3,
     dict(seed=5, side=9, channels=4, atoms=30)),
]


def same_pad(n, k, s, d=1):
    ke = (k - 1) * d + 1
    out = -(-n // s)
    tot = max((out - 1) * s + ke - n, 0)
    return tot // 2, tot - tot // 2


def torch_forward(cfg, weights, frames, dt):
    vals = {}
    W = {k: [torch.from_numpy(np.asarray(a)).to(dt) for a in v] for k, v in weights.items()}

    def act(x, name, alpha=1.0):
        if name in (None, "linear"):
            return x
        return {"relu": F.relu, "elu": lambda t: F.elu(t, alpha), "softmax": lambda t: F.softmax(t, dim=1),
                "sigmoid": torch.sigmoid, "tanh": torch.tanh}[name](x)

    for lc in cfg["config"]["layers"]:
        cn, c, name = lc["class_name"], lc["config"], lc["name"]
        xs = [vals[t[0]] for t in lc["inbound_nodes"][0]] if lc["inbound_nodes"] else []
        w = W.get(name, [])
        if cn == "InputLayer":  # NDHWC -> NCDHW
            y = torch.from_numpy(np.asarray(frames)).to(dt).permute(0, 4, 1, 2, 3).contiguous()
        elif cn == "Conv3D":
            x = xs[0]
            k, s = c["kernel_size"], c["strides"]
            if c["padding"] == "same":
                p = [same_pad(x.shape[2 + i], k[i], s[i]) for i in range(3)]
                x = F.pad(x, (p[2][0], p[2][1], p[1][0], p[1][1], p[0][0], p[0][1]))
            kern = w[0].permute(4, 3, 0, 1, 2).contiguous()   # [kd,kh,kw,Ci,Co] -> [Co,Ci,kd,kh,kw]
            y = F.conv3d(x, kern, w[1] if c["use_bias"] else None, stride=s)
            y = act(y, c["activation"])
        elif cn == "Dense":
            y = xs[0] @ w[0]
            if c["use_bias"]:
                y = y + w[1]
            y = act(y, c["activation"])
        elif cn == "BatchNormalization":
            wi = iter(w)
            g = next(wi) if c["scale"] else None
            b = next(wi) if c["center"] else None
            m, v = next(wi), next(wi)
            y = F.batch_norm(xs[0], m, v, g, b, training=False, eps=c["epsilon"])
        elif cn == "ELU":
            y = F.elu(xs[0], c["alpha"])
        elif cn == "ReLU":
            y = F.relu(xs[0])
        elif cn == "LeakyReLU":
            y = F.leaky_relu(xs[0], c["alpha"])
        elif cn == "Softmax":
            y = F.softmax(xs[0], dim=1)
        elif cn == "Activation":
            y = act(xs[0], c["activation"])
        elif cn == "MaxPooling3D":
            assert c["padding"] == "valid"
            y = F.max_pool3d(xs[0], c["pool_size"], c["strides"])
        elif cn == "AveragePooling3D":
            assert c["padding"] == "valid"
            y = F.avg_pool3d(xs[0], c["pool_size"], c["strides"])
        elif cn == "GlobalAveragePooling3D":
            y = xs[0].mean(dim=(2, 3, 4))
        elif cn == "GlobalMaxPooling3D":
            y = xs[0].amax(dim=(2, 3, 4))
        elif cn == "Flatten":  # Keras flattens channels_last: (D,H,W,C) row-major
            y = xs[0].permute(0, 2, 3, 4, 1).reshape(xs[0].shape[0], -1)
        elif cn == "Concatenate":
            y = torch.cat(xs, dim=1)
        elif cn == "Add":
            y = sum(xs[1:], xs[0])
        elif cn in ("Dropout", "SpatialDropout3D"):
            y = xs[0]
        else:
            raise ValueError(cn)
        vals[name] = y
    return vals[cfg["config"]["output_layers"][0][0]].numpy()


def main():
    torch.set_num_threads(8)
    out = {}
    meta = []
    from oracle import cnn_oracle
    for name, builder, kw, n, fkw in CASES:
        cfg, weights = getattr(synth, builder)(**kw)
        frames = synth.synthetic_frames(n, **fkw)
        y32 = torch_forward(cfg, weights, frames, torch.float32)
        y64 = torch_forward(cfg, weights, frames, torch.float64)
        o32 = cnn_oracle.forward(cfg, weights, frames, np.float32)
        o64 = cnn_oracle.forward(cfg, weights, frames, np.float64)
        print(f"{name:18s} torch32-vs-64 {np.abs(y32 - y64).max():.2e}  oracle32-vs-torch64 "
              f"{np.abs(o32 - y64).max():.2e}  oracle64-vs-torch64 {np.abs(o64 - y64).max():.2e}  "
              f"argmax equal {np.array_equal(o32.argmax(1), y64.argmax(1))}  pmax {y64.max():.3f}")
        out[f"{name}__torch32"] = y32.astype(np.float32)
        out[f"{name}__torch64"] = y64.astype(np.float64)
        meta.append(dict(name=name, builder=builder, kwargs=kw, n=n, frame_kwargs=fkw))
    out["meta"] = np.array(json.dumps(meta))
    path = os.path.join(ROOT, "tests", "golden", "cnn_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
