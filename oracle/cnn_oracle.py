"""CPU ORACLE (test infrastructure only) — NumPy restatement of the CNN forward pass.

    *** PARITY UNPINNED ***  The arithmetic of this path lives in the third-party dependency
    tensorflow==2.13.0 (reference requirements.txt:8); TensorFlow, Keras and every released .h5 are
    absent from /root/reference and from this image, and the reference has no test or golden vector
    for `Model.predict`.  This file therefore restates the *published* Keras inference semantics
     and is anchored on the reference's call sites
    (predict.py:121 load_model, predict.py:142 frame_model.predict(X_batch)) and on an independent
    implementation (torch CPU functional ops, see tests/golden/make_cnn_golden.py) — not on TensorFlow.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product (timed-design_amd/) never does.

It deliberately shares NO code with the product: it walks the Keras ``model_config`` dict and the
``{layer: [arrays]}`` weights directly (the product goes config -> pack -> C++ planner -> HIP), so
it also checks the converter and the planner's fusion decisions.

What Keras does, op by op (predict => inference mode):
  InputLayer            cast to float32 (design_utils/utils.py:518-521 builds float64 or bool X)
  Conv3D                cross-correlation, channels_last, kernel [kd,kh,kw,Cin,Cout];
                        'same': out=ceil(in/s), pad_total=max((out-1)*s+(k-1)*d+1-in,0), before=floor(/2)
  BatchNormalization    gamma*(x-mean)/sqrt(var+eps)+beta, eps default 1e-3
  ELU/ReLU/LeakyReLU    x>0 ? x : alpha*(exp(x)-1) | max(x,0) | x>0 ? x : alpha*x
  Max/AveragePooling3D  'valid' => floor; 'same' average divides by the in-bounds count
  Dropout family        identity
  GlobalAverage/MaxPool mean/max over D,H,W
  Flatten               row-major over (D,H,W,C)
  Dense                 x @ W[in,out] + b
  Softmax               exp(x-max)/sum over the last axis
  Concatenate / Add     channel concat / elementwise sum
"""
from __future__ import annotations

import json
from typing import Dict, Sequence

import numpy as np
from numpy.lib.stride_tricks import sliding_window_view

_IDENTITY = {"Dropout", "SpatialDropout3D", "SpatialDropout2D", "SpatialDropout1D", "GaussianNoise",
             "GaussianDropout", "AlphaDropout", "ActivityRegularization"}


def _t3(v):
    return (v, v, v) if isinstance(v, int) else tuple(int(a) for a in v)


def _same_pads(n, k, s, d):
    ke = (k - 1) * d + 1
    out = -(-n // s)
    total = max((out - 1) * s + ke - n, 0)
    return total // 2, total - total // 2


def _activation(x, name, alpha=None):
    # `name` is what Keras serialises for a layer's `activation` argument: a function name (Keras applies that
    # function's default parameters: elu alpha=1.0, leaky_relu negative_slope=0.2) or a serialized object whose
    # config carries alpha / negative_slope.  An explicit `alpha` argument (ELU / LeakyReLU / ReLU layers) wins.
    if isinstance(name, dict):
        conf = name.get("config") if isinstance(name.get("config"), dict) else {}
        if alpha is None:
            alpha = conf.get("negative_slope", conf.get("alpha"))
        cls = name.get("class_name")
        name = {"LeakyReLU": "leaky_relu", "ELU": "elu", "ReLU": "relu", "Softmax": "softmax"}.get(cls) or conf.get("name", cls)
        if name == "leaky_relu" and alpha is None:
            alpha = 0.3 if cls == "LeakyReLU" else 0.2
    if alpha is None:
        alpha = {"leaky_relu": 0.2}.get(name, 1.0)
    if name in (None, "linear"):
        return x
    if name == "relu":
        return np.maximum(x, 0)
    if name == "elu":
        return np.where(x > 0, x, (alpha * np.expm1(np.minimum(x, 0))).astype(x.dtype))
    if name == "leaky_relu":
        return np.where(x > 0, x, (alpha * x).astype(x.dtype))
    if name == "sigmoid":
        return (1.0 / (1.0 + np.exp(-x))).astype(x.dtype)
    if name == "tanh":
        return np.tanh(x)
    if name == "softmax":
        e = np.exp(x - x.max(axis=-1, keepdims=True))
        return (e / e.sum(axis=-1, keepdims=True)).astype(x.dtype)
    raise ValueError(f"activation {name}")


_CONV_C = None        # ctypes handle of oracle/_build/liboracle_conv.so once use_c_conv() found it


def use_c_conv(threads: int = 0) -> bool:
    """Route float32, dilation-1 convolutions of forward() through the blocked direct convolution of oracle/conv3d_omp.c (OpenMP
    over `threads` host cores; 0 = OpenMP's default) — the `cpu_baseline` configuration of bench.py.  Returns
    False when the library has not been built (__graft_entry__.build() / make -C oracle): the NumPy im2col + sgemm form stays.
    tests/test_oracle_cnn.py holds the two forms together."""
    global _CONV_C
    import ctypes as C
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_build", "liboracle_conv.so")
    if not os.path.exists(path):
        return False
    lib = C.CDLL(path)
    lib.oracle_conv3d_f32.restype = C.c_int
    lib.oracle_conv3d_f32.argtypes = [C.c_void_p] * 4 + [C.c_long] + [C.c_int] * 18
    _CONV_C = (lib, int(threads))
    return True


def use_numpy_conv() -> None:
    global _CONV_C
    _CONV_C = None


def _conv3d_c(x, kernel, bias, s, padding):
    lib, threads = _CONV_C
    n, d, h, w, cin = x.shape
    kd, kh, kw, _cin, cout = kernel.shape
    pads = [_same_pads(x.shape[1 + i], kernel.shape[i], s[i], 1) if padding == "same" else (0, 0) for i in range(3)]
    outs = [(x.shape[1 + i] + sum(pads[i]) - kernel.shape[i]) // s[i] + 1 for i in range(3)]
    xc = np.ascontiguousarray(x, dtype=np.float32)
    kc = np.ascontiguousarray(kernel, dtype=np.float32)
    bc = None if bias is None else np.ascontiguousarray(bias, dtype=np.float32)
    y = np.empty((n, *outs, cout), dtype=np.float32)
    rc = lib.oracle_conv3d_f32(xc.ctypes.data, kc.ctypes.data, bc.ctypes.data if bc is not None else None, y.ctypes.data, n, d, h, w, cin,
                               cout, kd, kh, kw, s[0], s[1], s[2], pads[0][0], pads[1][0], pads[2][0], outs[0], outs[1], outs[2], threads)
    if rc != 0:
        raise MemoryError("oracle_conv3d_f32 could not allocate its scratch")
    return y


def conv3d(x, kernel, bias, strides, dilation, padding, acc_dtype):
    """x [N,D,H,W,Cin]; kernel [kd,kh,kw,Cin,Cout]."""
    kd, kh, kw, cin, cout = kernel.shape
    s, d = _t3(strides), _t3(dilation)
    if _CONV_C is not None and acc_dtype == np.float32 and d == (1, 1, 1) and x.ndim == 5 and min(x.shape) > 0:
        return _conv3d_c(x, kernel, bias, s, padding)
    if padding == "same":
        pads = [_same_pads(x.shape[1 + i], kernel.shape[i], s[i], d[i]) for i in range(3)]
        x = np.pad(x, [(0, 0), *pads, (0, 0)])
    ke = [(kernel.shape[i] - 1) * d[i] + 1 for i in range(3)]
    win = sliding_window_view(x, ke, axis=(1, 2, 3))            # [N,Do,Ho,Wo,C,ked,keh,kew]
    win = win[:, ::s[0], ::s[1], ::s[2], :, ::d[0], ::d[1], ::d[2]]
    n, do, ho, wo = win.shape[:4]
    cols = np.ascontiguousarray(win.transpose(0, 1, 2, 3, 5, 6, 7, 4)).reshape(n * do * ho * wo, kd * kh * kw * cin)
    y = cols.astype(acc_dtype, copy=False) @ kernel.reshape(-1, cout).astype(acc_dtype, copy=False)
    if bias is not None:
        y = y + bias.astype(acc_dtype)
    return y.reshape(n, do, ho, wo, cout)


def pool3d(x, size, strides, padding, mode):
    p, s = _t3(size), _t3(strides)
    if padding == "same":
        pads = [_same_pads(x.shape[1 + i], p[i], s[i], 1) for i in range(3)]
        fill = -np.inf if mode == "max" else 0.0
        cnt = np.pad(np.ones(x.shape[1:4], dtype=x.dtype), pads)
        x = np.pad(x, [(0, 0), *pads, (0, 0)], constant_values=fill)
    else:
        cnt = None
    win = sliding_window_view(x, p, axis=(1, 2, 3))[:, ::s[0], ::s[1], ::s[2]]
    if mode == "max":
        return win.max(axis=(-3, -2, -1))
    tot = win.sum(axis=(-3, -2, -1), dtype=x.dtype)
    if cnt is None:
        return (tot / x.dtype.type(p[0] * p[1] * p[2])).astype(x.dtype)
    cw = sliding_window_view(cnt, p)[::s[0], ::s[1], ::s[2]].sum(axis=(-3, -2, -1))
    return (tot / cw[None, ..., None]).astype(x.dtype)


def forward(model_config, weights: Dict[str, Sequence[np.ndarray]], frames: np.ndarray,
            dtype=np.float32, return_all: bool = False):
    """Run the Keras graph described by ``model_config`` on ``frames`` [N,D,H,W,C] (any real/bool dtype).

    ``dtype=np.float32`` mirrors Keras (fp32 storage, fp32 BLAS accumulation); ``np.float64`` is the
    high-precision arbiter used to size tolerances.
    """
    if isinstance(model_config, (str, bytes)):
        model_config = json.loads(model_config)
    cls = model_config["class_name"]
    cfg = model_config["config"]
    layers = cfg["layers"]
    vals: Dict[str, np.ndarray] = {}
    prev = None
    W = {k: [np.asarray(a) for a in v] for k, v in weights.items()}
    for lc in layers:
        cname, c = lc["class_name"], lc["config"]
        name = lc.get("name") or c["name"]
        if cls == "Sequential":
            if prev is None and cname != "InputLayer":
                vals["__in__"] = np.asarray(frames).astype(dtype)
                prev = "__in__"
            ins = [prev] if prev is not None else []
        else:
            ins = [t[0] for t in lc["inbound_nodes"][0]] if lc.get("inbound_nodes") else []
        xs = [vals[i] for i in ins]
        w = W.get(name, [])
        if cname == "InputLayer":
            y = np.asarray(frames).astype(dtype)
            want = tuple(c["batch_input_shape"][1:])
            assert y.shape[1:] == want, f"frames {y.shape[1:]} != model input {want}"
        elif cname == "Conv3D":
            b = w[1] if c.get("use_bias", True) else None
            y = conv3d(xs[0], w[0], b, c.get("strides", 1), c.get("dilation_rate", 1), c.get("padding", "valid"), dtype)
            y = _activation(y.astype(dtype), c.get("activation"))
        elif cname == "Dense":
            y = xs[0].astype(dtype) @ w[0].astype(dtype)
            if c.get("use_bias", True):
                y = y + w[1].astype(dtype)
            y = _activation(y.astype(dtype), c.get("activation"))
        elif cname == "BatchNormalization":
            wi = iter(w)
            gamma = next(wi) if c.get("scale", True) else None
            beta = next(wi) if c.get("center", True) else None
            mean, var = next(wi), next(wi)
            eps = dtype(c.get("epsilon", 1e-3))
            inv = (1.0 / np.sqrt(var.astype(dtype) + eps)).astype(dtype)
            if gamma is not None:
                inv = inv * gamma.astype(dtype)
            y = (xs[0] - mean.astype(dtype)) * inv
            if beta is not None:
                y = y + beta.astype(dtype)
        elif cname == "Activation":
            y = _activation(xs[0], c["activation"])
        elif cname == "ELU":
            y = _activation(xs[0], "elu", c.get("alpha", 1.0))
        elif cname == "ReLU":
            ns = float(c.get("negative_slope", 0.0))
            y = _activation(xs[0], "leaky_relu", ns) if ns else _activation(xs[0], "relu")
        elif cname == "LeakyReLU":
            y = _activation(xs[0], "leaky_relu", float(c.get("alpha", c.get("negative_slope", 0.3))))
        elif cname == "Softmax":
            y = _activation(xs[0], "softmax")
        elif cname in ("MaxPooling3D", "AveragePooling3D"):
            p = c.get("pool_size", 2)
            s = c["strides"] if c.get("strides") is not None else p
            y = pool3d(xs[0], p, s, c.get("padding", "valid"), "max" if cname.startswith("Max") else "avg")
        elif cname == "GlobalAveragePooling3D":
            y = xs[0].mean(axis=(1, 2, 3), dtype=dtype)
        elif cname == "GlobalMaxPooling3D":
            y = xs[0].max(axis=(1, 2, 3))
        elif cname == "Flatten":
            y = xs[0].reshape(xs[0].shape[0], -1)
        elif cname == "Concatenate":
            y = np.concatenate(xs, axis=-1)
        elif cname == "Add":
            y = xs[0]
            for o in xs[1:]:
                y = y + o
        elif cname in _IDENTITY:
            y = xs[0]
        else:
            raise ValueError(f"oracle: unsupported Keras layer {cname}")
        vals[name] = y.astype(dtype, copy=False)
        prev = name
    if return_all:
        return vals
    if cls == "Sequential":
        return vals[prev]
    return vals[cfg["output_layers"][0][0]]
