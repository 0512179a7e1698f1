"""Keras legacy ``.h5`` model file -> (model_config dict, {layer: [weights]}).

This is the file half of ``tf.keras.models.load_model(path)`` (reference predict.py:121): root
attribute ``model_config`` (JSON) and group ``model_weights`` whose attr ``layer_names`` lists the
layers, each layer group carrying ``weight_names`` and the datasets (``conv3d/kernel:0`` ...).
``training_config`` (which references the custom metric ``top_3_cat_acc``, reference
predict.py:24-25,88) is irrelevant for inference and ignored.  Uses h5py when importable, else the
bundled pure-Python reader.
"""
from __future__ import annotations

import json
from typing import Dict, List, Tuple

import numpy as np


def _open(path):
    try:
        import h5py  # type: ignore
        return h5py.File(str(path), "r")
    except ImportError:
        from . import h5lite
        return h5lite.File(path)


def _text(v) -> str:
    if isinstance(v, bytes):
        return v.decode("utf-8")
    if isinstance(v, np.ndarray) and v.shape == ():
        return _text(v[()])
    if isinstance(v, (np.bytes_, np.str_)):
        return v.decode("utf-8") if isinstance(v, np.bytes_) else str(v)
    return str(v)


def _names(v) -> List[str]:
    return [_text(x).rstrip("\0") for x in np.asarray(v).ravel()]


def read_keras_h5(path) -> Tuple[dict, Dict[str, List[np.ndarray]]]:
    with _open(path) as f:
        if "model_config" not in f.attrs:
            raise ValueError(f"{path}: no model_config attribute (weights-only file?)")
        cfg = json.loads(_text(f.attrs["model_config"]))
        if "model_weights" not in f:
            raise ValueError(f"{path}: no model_weights group")
        g = f["model_weights"]
        weights: Dict[str, List[np.ndarray]] = {}
        top = set(_names(g.attrs["layer_names"]))
        bare: Dict[str, Dict[str, List[np.ndarray]]] = {}       # inner layer name -> {outer model: its arrays}
        for lname in _names(g.attrs["layer_names"]):
            lg = g[lname]
            wn = _names(lg.attrs["weight_names"]) if "weight_names" in lg.attrs else []
            weights[lname] = [np.asarray(lg[w][()], dtype=np.float32) for w in wn]
            # a nested model used as a layer keeps its layers' weights in ITS group as "<inner layer>/kernel:0" ... in
            # `layer.weights` order (trainable first): expose them under "<outer>/<inner>" (and "<inner>" where that is unambiguous) as well, each in
            # Keras' per-layer order (kernel, bias / gamma, beta, moving_mean, moving_variance — which that order keeps)
            for w, arr in zip(wn, weights[lname]):
                parts = w.split("/")
                if len(parts) >= 2 and parts[0] != lname:
                    inner = "/".join(parts[:-1])
                    weights.setdefault(f"{lname}/{inner}", []).append(arr)
                    bare.setdefault(parts[-2], {}).setdefault(lname, []).append(arr)
        # the bare "<inner>" alias exists only where it is unambiguous: not when a top-level layer has that name (its own list
        # would grow to 4 arrays), not when two nested models both hold a layer of that name — the qualified key always works
        for name, owners in bare.items():
            if name not in top and len(owners) == 1:
                weights[name] = next(iter(owners.values()))
    return cfg, weights
