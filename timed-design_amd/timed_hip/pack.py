"""Model pack: the flat little-endian file the HIP runtime loads instead of a Keras ``.h5``.

Replaces the deserialisation half of ``tf.keras.models.load_model`` (reference predict.py:121):
``model_config`` + ``model_weights`` are converted ONCE (here, on the host, by
keras_config.parse_keras_model) into a list of fixed-size node records plus one fp32 blob, which
``th_model_load`` (include/timed_hip.h) reads without any JSON / HDF5 code on the device side.

Layout (all little-endian; mirrored by csrc/pack.h):

    header  : char magic[8]="THPK0001"; u32 n_nodes; u32 n_blobs; u32 output_node; u32 reserved
    nodes   : n_nodes x 256 B  { u32 op; u32 n_in; i32 in[8]; i32 ip[24]; f32 fp[8]; i32 w[8]; char name[56] }
    blobs   : n_blobs x { u64 offset (in floats, from start of data); u64 count }
    (pad to 16 B)
    data    : fp32 weights

Integer parameter slots per op (IP_SLOTS) and blob slots (W_SLOTS) are fixed below.
"""
from __future__ import annotations

import struct
from typing import Dict, List

import numpy as np

from . import keras_config as kc

MAGIC = b"THPK0001"
NODE_BYTES = 256
MAX_IN, N_IP, N_FP, N_W, NAME_BYTES = 8, 24, 8, 8, 56

IP_SLOTS = {
    kc.OP_INPUT: ["d", "h", "w", "c"],
    kc.OP_CONV3D: ["kd", "kh", "kw", "sd", "sh", "sw", "dd", "dh", "dw", "same", "cin", "cout", "use_bias", "act"],
    kc.OP_DENSE: ["fin", "fout", "use_bias", "act"],
    kc.OP_BN: ["c"],
    kc.OP_ACT: ["act"],
    kc.OP_MAXPOOL: ["pd", "ph", "pw", "sd", "sh", "sw", "same"],
    kc.OP_AVGPOOL: ["pd", "ph", "pw", "sd", "sh", "sw", "same"],
}
FP_SLOTS = {kc.OP_CONV3D: ["alpha"], kc.OP_DENSE: ["alpha"], kc.OP_BN: ["eps"], kc.OP_ACT: ["alpha"]}
W_SLOTS = {
    kc.OP_CONV3D: ["kernel", "bias"],
    kc.OP_DENSE: ["kernel", "bias"],
    kc.OP_BN: ["gamma", "beta", "mean", "var"],
}


def layers_to_pack(layers: List[kc.Layer]) -> bytes:
    index = {l.name: i for i, l in enumerate(layers)}
    blobs: List[np.ndarray] = []
    nodes = bytearray()
    for l in layers:
        if len(l.inputs) > MAX_IN:
            raise kc.UnsupportedLayer(f"{l.name}: more than {MAX_IN} inputs")
        ins = [index[n] for n in l.inputs] + [-1] * (MAX_IN - len(l.inputs))
        ip = [0] * N_IP
        if l.op == kc.OP_INPUT:
            shp = list(l.out_shape)
            if len(shp) != 4:
                raise kc.UnsupportedLayer(f"input must be (D,H,W,C), got {l.out_shape}")
            ip[:4] = shp
        else:
            for j, key in enumerate(IP_SLOTS.get(l.op, [])):
                ip[j] = int(l.ip[key])
        fp = [0.0] * N_FP
        for j, key in enumerate(FP_SLOTS.get(l.op, [])):
            fp[j] = float(l.fp.get(key, 0.0))
        w = [-1] * N_W
        for j, key in enumerate(W_SLOTS.get(l.op, [])):
            if key in l.weights:
                w[j] = len(blobs)
                blobs.append(np.ascontiguousarray(l.weights[key], dtype="<f4").ravel())
        # out_shape in the tail of ip so the runtime can cross-check its own shape inference
        shp = list(l.out_shape)
        ip[N_IP - 5] = len(shp)
        ip[N_IP - 4: N_IP - 4 + len(shp)] = shp
        name = l.name.encode()[: NAME_BYTES - 1]
        rec = struct.pack(f"<II{MAX_IN}i{N_IP}i{N_FP}f{N_W}i{NAME_BYTES}s", l.op, len(l.inputs), *ins, *ip, *fp, *w, name)
        assert len(rec) == NODE_BYTES
        nodes += rec
    table = bytearray()
    off = 0
    for b in blobs:
        table += struct.pack("<QQ", off, b.size)
        off += (b.size + 3) // 4 * 4  # keep every blob 16-byte aligned
    head = struct.pack("<8sIIII", MAGIC, len(layers), len(blobs), len(layers) - 1, 0)
    out = bytearray(head + nodes + table)
    out += b"\0" * (-len(out) % 16)
    for b in blobs:
        out += b.tobytes()
        out += b"\0" * (-(b.size * 4) % 16)
    return bytes(out)


def keras_to_pack(model_config, weights: Dict[str, list]) -> bytes:
    return layers_to_pack(kc.parse_keras_model(model_config, weights))


def read_pack(buf: bytes) -> List[dict]:
    """Decode a pack back into dicts (host-side inspection / tests only)."""
    magic, n_nodes, n_blobs, out_node, _ = struct.unpack_from("<8sIIII", buf, 0)
    if magic != MAGIC:
        raise ValueError("not a THPK0001 pack")
    pos = 24
    recs = []
    for _ in range(n_nodes):
        f = struct.unpack_from(f"<II{MAX_IN}i{N_IP}i{N_FP}f{N_W}i{NAME_BYTES}s", buf, pos)
        pos += NODE_BYTES
        op, n_in = f[0], f[1]
        ins = list(f[2:2 + n_in])
        ip = list(f[2 + MAX_IN: 2 + MAX_IN + N_IP])
        fp = list(f[2 + MAX_IN + N_IP: 2 + MAX_IN + N_IP + N_FP])
        w = list(f[2 + MAX_IN + N_IP + N_FP: 2 + MAX_IN + N_IP + N_FP + N_W])
        name = f[-1].split(b"\0")[0].decode()
        recs.append(dict(op=op, inputs=ins, ip=ip, fp=fp, w=w, name=name,
                         out_shape=tuple(ip[N_IP - 4: N_IP - 4 + ip[N_IP - 5]])))
    table = [struct.unpack_from("<QQ", buf, pos + 16 * i) for i in range(n_blobs)]
    pos += 16 * n_blobs
    pos += -pos % 16
    data = np.frombuffer(buf, dtype="<f4", offset=pos)
    for r in recs:
        r["weights"] = [None if i < 0 else data[table[i][0]: table[i][0] + table[i][1]] for i in r["w"]]
    return recs
