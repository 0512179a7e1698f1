"""Keras ``model_config`` (the JSON stored in a legacy ``.h5`` model file) -> normalised layer list.

The reference delegates the whole network to ``tf.keras.models.load_model(path)``
(reference predict.py:121) and never states a topology in code, so the engine is a
layer-graph interpreter: this module turns the Keras description into a flat, topologically
ordered list of :class:`Layer` records that ``pack.py`` serialises for the HIP runtime.

Only inference semantics matter (reference predict.py:142 calls ``Model.predict``):
Dropout / SpatialDropout3D / GaussianNoise are identities, BatchNormalization uses its moving
statistics.  The closed op set is SURVEY.md Appendix A.

Nothing here touches the GPU, TensorFlow or h5py; weights arrive as a
``{layer_name: [ndarray, ...]}`` dict in Keras' own per-layer order
(Conv3D/Dense: kernel, bias; BatchNormalization: gamma, beta, moving_mean, moving_variance,
with gamma/beta dropped when ``scale``/``center`` is False).
"""
from __future__ import annotations

import json
from dataclasses import dataclass, field
from typing import Dict, List, Sequence

import numpy as np

# ---- op / activation codes shared with csrc/pack.h (keep in sync) -------------------------
OP_INPUT, OP_CONV3D, OP_DENSE, OP_BN, OP_ACT, OP_MAXPOOL, OP_AVGPOOL = 0, 1, 2, 3, 4, 5, 6
OP_GAP, OP_GMP, OP_FLATTEN, OP_CONCAT, OP_ADD, OP_IDENTITY = 7, 8, 9, 10, 11, 12
OP_NAMES = {
    OP_INPUT: "input", OP_CONV3D: "conv3d", OP_DENSE: "dense", OP_BN: "batchnorm",
    OP_ACT: "activation", OP_MAXPOOL: "maxpool3d", OP_AVGPOOL: "avgpool3d", OP_GAP: "gap3d",
    OP_GMP: "gmp3d", OP_FLATTEN: "flatten", OP_CONCAT: "concat", OP_ADD: "add",
    OP_IDENTITY: "identity",
}
ACT_LINEAR, ACT_RELU, ACT_ELU, ACT_SOFTMAX, ACT_SIGMOID, ACT_TANH, ACT_LEAKY = 0, 1, 2, 3, 4, 5, 6
_ACT_BY_NAME = {
    None: ACT_LINEAR, "linear": ACT_LINEAR, "relu": ACT_RELU, "elu": ACT_ELU,
    "softmax": ACT_SOFTMAX, "sigmoid": ACT_SIGMOID, "tanh": ACT_TANH,
    "leaky_relu": ACT_LEAKY, "LeakyReLU": ACT_LEAKY,
}
_IDENTITY_CLASSES = {
    "Dropout", "SpatialDropout3D", "SpatialDropout2D", "SpatialDropout1D", "GaussianNoise",
    "GaussianDropout", "AlphaDropout", "ActivityRegularization",
}


class UnsupportedLayer(ValueError):
    """Raised for a Keras layer / option outside the closed inference op set."""


@dataclass
class Layer:
    name: str
    op: int
    inputs: List[str] = field(default_factory=list)
    ip: Dict[str, int] = field(default_factory=dict)     # integer parameters
    fp: Dict[str, float] = field(default_factory=dict)   # float parameters
    weights: Dict[str, np.ndarray] = field(default_factory=dict)
    out_shape: tuple = ()                                 # (D,H,W,C) or (F,)


def _act_code(name):
    """activation spec -> (code, alpha).  A string names the Keras function with its default parameters
    ('elu' alpha 1.0; 'leaky_relu' negative_slope 0.2 — keras.activations.leaky_relu's default); a serialized
    object ({"class_name": ..., "config": {...}}) carries its own alpha / negative_slope.  A leaky slope that
    cannot be determined is an error, never a silent identity."""
    cfg = {}
    if isinstance(name, dict):  # serialized activation object
        cfg = name.get("config") or {}
        if not isinstance(cfg, dict):       # {"class_name": "function", "config": "relu"}
            cfg = {"name": cfg}
        cls = name.get("class_name")
        name = cls if cls in _ACT_BY_NAME else cfg.get("name", cls)
    if name not in _ACT_BY_NAME:
        raise UnsupportedLayer(f"activation {name!r} is not supported")
    code = _ACT_BY_NAME[name]
    alpha = 1.0
    if code == ACT_ELU:
        alpha = float(cfg.get("alpha", 1.0))
    elif code == ACT_LEAKY:
        if "negative_slope" in cfg or "alpha" in cfg:
            alpha = float(cfg.get("negative_slope", cfg.get("alpha")))
        elif name == "leaky_relu":
            alpha = 0.2
        elif name == "LeakyReLU" and cfg:   # the layer class used as an activation: its own default
            alpha = 0.3
        else:
            raise UnsupportedLayer(f"activation {name!r}: negative slope not given")
    return code, alpha


def _check_compute_dtype(name: str, dtype) -> None:
    """A layer's ``dtype`` entry is a string ("float32") or, since TF 2.4, a serialized policy
    ({"class_name": "Policy", "config": {"name": "mixed_float16"}}).  The engine evaluates in float32 — what every
    model of the reference does; a float16 / bfloat16 / mixed-precision / float64 layer would give other numbers than
    Keras, so it is refused instead of being computed silently at another precision."""
    if dtype is None:
        return
    if isinstance(dtype, dict):
        dtype = (dtype.get("config") or {}).get("name", dtype.get("class_name"))
    if str(dtype) != "float32":
        raise UnsupportedLayer(f"{name}: dtype policy {dtype!r} — the engine computes in float32 only")


def _triple(v) -> tuple:
    if isinstance(v, int):
        return (v, v, v)
    v = tuple(int(x) for x in v)
    if len(v) != 3:
        raise UnsupportedLayer(f"expected a 3-tuple, got {v}")
    return v


def _conv_out(n: int, k: int, s: int, d: int, same: bool) -> int:
    ke = (k - 1) * d + 1
    return -(-n // s) if same else (n - ke) // s + 1


def _inbound_names(layer_cfg: dict) -> List[str]:
    nodes = layer_cfg.get("inbound_nodes") or []
    if not nodes:
        return []
    if len(nodes) != 1:
        raise UnsupportedLayer(f"layer {layer_cfg.get('name')} is shared ({len(nodes)} call sites)")
    node = nodes[0]
    # TF<=2.15 format: [[name, node_idx, tensor_idx, kwargs], ...]
    if isinstance(node, list):
        return [str(t[0]) for t in node]
    # Keras 3 format: {"args": [...], "kwargs": {}} with __keras_tensor__ entries
    names: List[str] = []

    def walk(o):
        if isinstance(o, dict):
            if o.get("class_name") == "__keras_tensor__":
                names.append(str(o["config"]["keras_history"][0]))
            else:
                for v in o.values():
                    walk(v)
        elif isinstance(o, (list, tuple)):
            for v in o:
                walk(v)

    walk(node.get("args", []))
    return names


def parse_keras_model(model_config, weights: Dict[str, Sequence[np.ndarray]]) -> List[Layer]:
    """Normalise a Keras Sequential/Functional ``model_config`` into a topologically ordered list.

    ``model_config`` may be the JSON string or the decoded dict.  The last element of the
    returned list is the model output.
    """
    if isinstance(model_config, (str, bytes)):
        model_config = json.loads(model_config)
    cls = model_config.get("class_name")
    cfg = model_config["config"]
    raw_layers = cfg["layers"] if isinstance(cfg, dict) else cfg  # very old Sequential: bare list
    if cls not in ("Sequential", "Functional", "Model"):
        raise UnsupportedLayer(f"model class {cls!r} is not supported")
    if cls != "Sequential":
        raw_layers, nested_alias = _inline_nested_models(raw_layers)
    else:
        nested_alias = {}

    layers: List[Layer] = []
    shapes: Dict[str, tuple] = {}
    alias: Dict[str, str] = dict(nested_alias)
    prev_name = None

    def resolve(n: str) -> str:
        while n in alias:
            n = alias[n]
        return n

    def add(layer: Layer):
        layers.append(layer)
        shapes[layer.name] = layer.out_shape

    for lc in raw_layers:
        cname = lc["class_name"]
        c = lc["config"]
        name = lc.get("name") or c["name"]
        if cls == "Sequential":
            if cname != "InputLayer" and prev_name is None:
                # legacy Sequential without an explicit InputLayer
                bis = c.get("batch_input_shape") or c.get("batch_shape")
                if bis is None:
                    raise UnsupportedLayer("Sequential model without an input shape")
                inp = Layer(name=name + "_input", op=OP_INPUT, out_shape=tuple(int(x) for x in bis[1:]))
                add(inp)
                prev_name = inp.name
            ins = [prev_name] if prev_name is not None else []
        else:
            ins = [resolve(n) for n in _inbound_names(lc)]
        if c.get("data_format", "channels_last") != "channels_last":
            raise UnsupportedLayer(f"{name}: only channels_last is supported")
        _check_compute_dtype(name, c.get("dtype"))
        # layers of an inlined nested model are called "<outer>/<inner>"; a Keras .h5 stores their weights under the
        # inner name inside the outer layer's group (timed_hip.h5model exposes both spellings)
        w = list(weights.get(name, weights.get(name.rsplit("/", 1)[-1], [])))

        if cname == "InputLayer":
            bis = c.get("batch_input_shape") or c.get("batch_shape")
            add(Layer(name=name, op=OP_INPUT, out_shape=tuple(int(x) for x in bis[1:])))
        elif cname == "Conv3D":
            (d, h, wd, cin) = shapes[ins[0]]
            k = _triple(c["kernel_size"]); s = _triple(c.get("strides", 1)); dl = _triple(c.get("dilation_rate", 1))
            if c.get("groups", 1) != 1:
                raise UnsupportedLayer(f"{name}: grouped convolution is not supported")
            pad = c.get("padding", "valid")
            if pad not in ("valid", "same"):
                raise UnsupportedLayer(f"{name}: padding {pad!r}")
            same = pad == "same"
            cout = int(c["filters"])
            use_bias = bool(c.get("use_bias", True))
            kern = np.asarray(w[0], dtype=np.float32)
            if kern.shape != (*k, cin, cout):
                raise ValueError(f"{name}: kernel shape {kern.shape} != {(*k, cin, cout)}")
            ws = {"kernel": kern}
            if use_bias:
                ws["bias"] = np.asarray(w[1], dtype=np.float32).reshape(cout)
            out = (_conv_out(d, k[0], s[0], dl[0], same), _conv_out(h, k[1], s[1], dl[1], same),
                   _conv_out(wd, k[2], s[2], dl[2], same), cout)
            act, alpha = _act_code(c.get("activation"))
            add(Layer(name=name, op=OP_CONV3D, inputs=ins,
                      ip=dict(kd=k[0], kh=k[1], kw=k[2], sd=s[0], sh=s[1], sw=s[2], dd=dl[0], dh=dl[1],
                              dw=dl[2], same=int(same), cin=cin, cout=cout, use_bias=int(use_bias),
                              act=act),
                      fp=dict(alpha=alpha), weights=ws, out_shape=out))
        elif cname == "Dense":
            shp = shapes[ins[0]]
            if len(shp) != 1:
                raise UnsupportedLayer(f"{name}: Dense on a rank-{len(shp)+1} tensor (Flatten/pool first)")
            fin, fout = shp[0], int(c["units"])
            use_bias = bool(c.get("use_bias", True))
            kern = np.asarray(w[0], dtype=np.float32)
            if kern.shape != (fin, fout):
                raise ValueError(f"{name}: kernel shape {kern.shape} != {(fin, fout)}")
            ws = {"kernel": kern}
            if use_bias:
                ws["bias"] = np.asarray(w[1], dtype=np.float32).reshape(fout)
            act, alpha = _act_code(c.get("activation"))
            add(Layer(name=name, op=OP_DENSE, inputs=ins,
                      ip=dict(fin=fin, fout=fout, use_bias=int(use_bias), act=act),
                      fp=dict(alpha=alpha), weights=ws, out_shape=(fout,)))
        elif cname == "BatchNormalization":
            shp = shapes[ins[0]]
            axis = c.get("axis", -1)
            if isinstance(axis, (list, tuple)):
                axis = axis[0]
            if axis not in (-1, len(shp)):
                raise UnsupportedLayer(f"{name}: BatchNormalization axis {axis}")
            ch = shp[-1]
            wi = iter(w)
            ws = {}
            if c.get("scale", True):
                ws["gamma"] = np.asarray(next(wi), dtype=np.float32).reshape(ch)
            if c.get("center", True):
                ws["beta"] = np.asarray(next(wi), dtype=np.float32).reshape(ch)
            ws["mean"] = np.asarray(next(wi), dtype=np.float32).reshape(ch)
            ws["var"] = np.asarray(next(wi), dtype=np.float32).reshape(ch)
            add(Layer(name=name, op=OP_BN, inputs=ins, ip=dict(c=ch),
                      fp=dict(eps=float(c.get("epsilon", 1e-3))), weights=ws, out_shape=shp))
        elif cname in ("Activation", "ELU", "ReLU", "LeakyReLU", "Softmax"):
            shp = shapes[ins[0]]
            alpha = 1.0
            if cname == "Activation":
                act, alpha = _act_code(c["activation"])
            elif cname == "ELU":
                act, alpha = ACT_ELU, float(c.get("alpha", 1.0))
            elif cname == "ReLU":
                if c.get("max_value") is not None or float(c.get("threshold", 0.0)) != 0.0:
                    raise UnsupportedLayer(f"{name}: ReLU max_value/threshold")
                ns = float(c.get("negative_slope", 0.0))
                act, alpha = (ACT_LEAKY, ns) if ns != 0.0 else (ACT_RELU, 0.0)
            elif cname == "LeakyReLU":
                act = ACT_LEAKY
                alpha = float(c.get("alpha", c.get("negative_slope", 0.3)))
            else:  # Softmax layer
                ax = c.get("axis", -1)
                if ax not in (-1, len(shp)):
                    raise UnsupportedLayer(f"{name}: Softmax axis {ax}")
                act = ACT_SOFTMAX
            add(Layer(name=name, op=OP_ACT, inputs=ins, ip=dict(act=act), fp=dict(alpha=alpha), out_shape=shp))
        elif cname in ("MaxPooling3D", "AveragePooling3D"):
            (d, h, wd, ch) = shapes[ins[0]]
            p = _triple(c.get("pool_size", 2))
            s = _triple(c["strides"]) if c.get("strides") is not None else p
            same = c.get("padding", "valid") == "same"
            out = tuple(_conv_out(n, pk, sk, 1, same) for n, pk, sk in zip((d, h, wd), p, s)) + (ch,)
            add(Layer(name=name, op=OP_MAXPOOL if cname.startswith("Max") else OP_AVGPOOL, inputs=ins,
                      ip=dict(pd=p[0], ph=p[1], pw=p[2], sd=s[0], sh=s[1], sw=s[2], same=int(same)),
                      out_shape=out))
        elif cname in ("GlobalAveragePooling3D", "GlobalMaxPooling3D"):
            shp = shapes[ins[0]]
            if c.get("keepdims", False):
                raise UnsupportedLayer(f"{name}: keepdims=True")
            add(Layer(name=name, op=OP_GAP if "Average" in cname else OP_GMP, inputs=ins, out_shape=(shp[-1],)))
        elif cname == "Flatten":
            shp = shapes[ins[0]]
            add(Layer(name=name, op=OP_FLATTEN, inputs=ins, out_shape=(int(np.prod(shp)),)))
        elif cname == "Concatenate":
            shp0 = shapes[ins[0]]
            ax = c.get("axis", -1)
            if ax not in (-1, len(shp0)):
                raise UnsupportedLayer(f"{name}: Concatenate axis {ax} (channel axis only)")
            for n in ins:
                if shapes[n][:-1] != shp0[:-1]:
                    raise ValueError(f"{name}: mismatched concat inputs")
            add(Layer(name=name, op=OP_CONCAT, inputs=ins,
                      out_shape=shp0[:-1] + (sum(shapes[n][-1] for n in ins),)))
        elif cname == "Add":
            shp0 = shapes[ins[0]]
            for n in ins:
                if shapes[n] != shp0:
                    raise ValueError(f"{name}: mismatched Add inputs")
            add(Layer(name=name, op=OP_ADD, inputs=ins, out_shape=shp0))
        elif cname in _IDENTITY_CLASSES:
            # inference-mode identity: do not emit a node, just alias the tensor name
            alias[name] = ins[0]
            shapes[name] = shapes[ins[0]]
            prev_name = ins[0]
            continue
        else:
            raise UnsupportedLayer(f"Keras layer class {cname!r} ({name}) is not in the supported op set")
        prev_name = name

    if cls == "Sequential":
        out_name = prev_name
    else:
        outs = cfg.get("output_layers")
        if not outs or len(outs) != 1 and not isinstance(outs[0], str):
            raise UnsupportedLayer("exactly one model output is supported")
        o = outs[0] if not isinstance(outs[0], str) else outs
        out_name = resolve(str(o[0]))
    # make the output the last element (drop anything after it that nobody needs)
    idx = next(i for i, l in enumerate(layers) if l.name == out_name)
    layers = layers[: idx + 1]
    if sum(1 for l in layers if l.op == OP_INPUT) != 1:
        raise UnsupportedLayer("exactly one model input is supported")
    return push_through_concat(layers)


def _inline_nested_models(raw_layers: list):
    """A Functional model may use another Model / Sequential as a layer (``class_name`` "Functional" / "Model" /
    "Sequential" with its own ``config.layers``).  At inference that is just a sub-graph: its layers are spliced into the
    outer list as "<outer>/<inner>", its InputLayer becomes an alias of the tensor the outer graph feeds it, and the outer
    layer's name an alias of the sub-graph's output.  Returns (flat layer list, alias map); nesting may be deeper than one."""
    flat, alias = [], {}
    for lc in raw_layers:
        if lc.get("class_name") not in ("Functional", "Model", "Sequential") or not isinstance(lc.get("config"), dict) \
                or "layers" not in lc["config"]:
            flat.append(lc)
            continue
        outer = lc.get("name") or lc["config"]["name"]
        fed = _inbound_names(lc)
        if len(fed) != 1:
            raise UnsupportedLayer(f"{outer}: a nested model with {len(fed)} inputs is not supported")
        inner_cfg = lc["config"]
        sequential = lc["class_name"] == "Sequential"
        inner_layers, inner_alias = (inner_cfg["layers"], {}) if sequential else _inline_nested_models(inner_cfg["layers"])
        prev = fed[0]
        last = None
        for il in inner_layers:
            iname = il.get("name") or il["config"]["name"]
            full = f"{outer}/{iname}"
            if il["class_name"] == "InputLayer":
                alias[full] = fed[0]
                prev = full
                continue
            new = dict(il)
            new["name"] = full
            new["config"] = dict(il["config"], name=full)
            if sequential:
                new["inbound_nodes"] = [[[prev, 0, 0, {}]]]
            else:
                new["inbound_nodes"] = [[[f"{outer}/{n}", 0, 0, {}] for n in _inbound_names(il)]]
            flat.append(new)
            prev = last = full
        for k, v in inner_alias.items():
            alias[f"{outer}/{k}"] = f"{outer}/{v}"
        if sequential:
            out_name = last
        else:
            outs = inner_cfg.get("output_layers")
            if not outs or len(outs) != 1:
                raise UnsupportedLayer(f"{outer}: a nested model needs exactly one output")
            out_name = f"{outer}/{outs[0][0]}"
        if out_name is None:
            raise UnsupportedLayer(f"{outer}: empty nested model")
        alias[outer] = out_name
    return flat, alias


def push_through_concat(layers: List[Layer]) -> List[Layer]:
    """Graph rewrite (exact): a per-channel op or a pooling layer that is the ONLY consumer of a channel
    Concatenate is applied to each branch instead — ``BN(concat(a, b)) == concat(BN_a(a), BN_b(b))`` with the
    BatchNorm vectors sliced, likewise activations and Max/AveragePooling.  The engine can then fuse the op into
    each branch's convolution (epilogue / pooled store at a channel offset) instead of running separate
    elementwise and pooling kernels over the concatenated tensor (Inception/ProDCoNN-style parallel branches).
    Only done when every branch ends in a convolution chain, so DenseNet-style concatenations (a branch is itself
    a concatenation, and the concat has several consumers) are left alone."""
    by_name = {l.name: l for l in layers}

    def ends_in_conv(name: str) -> bool:
        l = by_name[name]
        while l.op in (OP_BN, OP_ACT) and len(l.inputs) == 1:
            l = by_name[l.inputs[0]]
        return l.op == OP_CONV3D

    changed = True
    while changed:
        changed = False
        for ci, c in enumerate(layers):
            if c.op != OP_CONCAT or ci == len(layers) - 1:
                continue
            users = [l for l in layers if c.name in l.inputs]
            if len(users) != 1:
                continue
            x = users[0]
            if x.inputs != [c.name] or x.op not in (OP_BN, OP_ACT, OP_MAXPOOL, OP_AVGPOOL):
                continue
            if x.op == OP_ACT and x.ip.get("act") == ACT_SOFTMAX:
                continue
            if not all(ends_in_conv(b) for b in c.inputs) or len(set(c.inputs)) != len(c.inputs):
                continue
            xi_list, off = [], 0
            for k, b in enumerate(c.inputs):
                cb = by_name[b].out_shape[-1]
                ws = {kk: np.ascontiguousarray(v[off:off + cb]) for kk, v in x.weights.items()}
                ip = dict(x.ip)
                if x.op == OP_BN:
                    ip["c"] = cb
                xi_list.append(Layer(name=f"{x.name}__b{k}", op=x.op, inputs=[b], ip=ip, fp=dict(x.fp), weights=ws,
                                     out_shape=tuple(x.out_shape[:-1]) + (cb,)))
                off += cb
            merged = Layer(name=x.name, op=OP_CONCAT, inputs=[l.name for l in xi_list], out_shape=tuple(x.out_shape))
            xpos = layers.index(x)
            layers = layers[:ci] + layers[ci + 1:xpos] + xi_list + [merged] + layers[xpos + 1:]
            by_name = {l.name: l for l in layers}
            changed = True
            break
    return layers


def flops_per_frame(layers: List[Layer]) -> float:
    """Algorithmic FLOPs: 2*V_out*k^3*Cin*Cout per conv + 2*F*out per Dense."""
    total = 0.0
    for l in layers:
        if l.op == OP_CONV3D:
            d, h, w, co = l.out_shape
            total += 2.0 * d * h * w * l.ip["kd"] * l.ip["kh"] * l.ip["kw"] * l.ip["cin"] * co
        elif l.op == OP_DENSE:
            total += 2.0 * l.ip["fin"] * l.ip["fout"]
    return total
