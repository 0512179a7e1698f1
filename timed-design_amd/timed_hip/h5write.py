"""h5write — a minimal HDF5 WRITER for the one file the voxeliser has to produce: an aposteriori-style frame dataset
(layout documented in the reference at design_utils/utils.py:238-251; written by ``make-frame-dataset ... -o .``,
README.md:83-97), so that frames voxelised here can be read by anything that reads aposteriori's output — h5py, the
reference's own predict.py, and this repo's h5lite.

What it writes is the conservative subset the HDF5 library itself writes with libver='earliest':
  superblock v0; object headers v1; old-style groups (symbol-table message -> B-tree v1 of SNOD nodes + local heap);
  datasets with contiguous or single-chunk deflate storage (layout v3, chunk B-tree v1, filter pipeline v1);
  attribute messages v1 with variable-length UTF-8 strings (global heap), IEEE floats, integers and h5py's bool enum.
Tested by reading the files back with real h5py and with h5lite (tests/test_h5write.py).
"""
from __future__ import annotations

import struct
import zlib
from typing import Dict, List, Optional, Tuple

import numpy as np

UNDEF = 0xFFFFFFFFFFFFFFFF
LEAF_K, INTERNAL_K = 4, 16          # group B-tree: SNODs hold <= 2*LEAF_K symbols, nodes <= 2*INTERNAL_K children


def _pad8(b: bytes) -> bytes:
    return b + b"\0" * (-len(b) % 8)


# ---- datatype / dataspace messages -----------------------------------------------------------------------------------
def _dtype_msg(dt: np.dtype) -> bytes:
    dt = np.dtype(dt)
    if dt == np.bool_:          # h5py: ENUM {FALSE=0, TRUE=1} over int8
        base = _dtype_msg(np.dtype("i1"))
        names = _pad8(b"FALSE\0") + _pad8(b"TRUE\0")
        return struct.pack("<BBBBI", 0x18, 2, 0, 0, 1) + base + names + bytes([0, 1])
    if dt.kind == "f":
        size = dt.itemsize
        exp_bits, mant_bits, bias = {2: (5, 10, 15), 4: (8, 23, 127), 8: (11, 52, 1023)}[size]
        head = struct.pack("<BBBBI", 0x11, 0x20, size * 8 - 1, 0, size)
        return head + struct.pack("<HHBBBBI", 0, size * 8, mant_bits, exp_bits, 0, mant_bits, bias)
    if dt.kind in "iu":
        head = struct.pack("<BBBBI", 0x10, 0x08 if dt.kind == "i" else 0x00, 0, 0, dt.itemsize)
        return head + struct.pack("<HH", 0, dt.itemsize * 8)
    if dt.kind == "S":
        return struct.pack("<BBBBI", 0x13, 0x00, 0, 0, dt.itemsize)       # null-terminated ASCII
    raise TypeError(f"h5write: dtype {dt}")


def _space_msg(shape: Tuple[int, ...]) -> bytes:
    return struct.pack("<BBBxxxxx", 1, len(shape), 0) + b"".join(struct.pack("<Q", int(d)) for d in shape)


# variable-length UTF-8 string (what h5py writes for a Python str, and what the reference compares with str values:
# `residue_label not in standard_residues`, design_utils/utils.py:376): class 9, type string, charset UTF-8, over a
# 1-byte integer base type; the characters live in a global heap collection
_VLEN_STR = struct.pack("<BBBBI", 0x19, 0x01, 0x01, 0x00, 16) + struct.pack("<BBBBIHH", 0x10, 0, 0, 0, 1, 0, 8)


def _strings(value):
    """list of str (and its shape) when ``value`` is a string or an array/list of strings, else None"""
    if isinstance(value, str):
        return [value], ()
    if isinstance(value, (list, tuple)) and value and all(isinstance(v, str) for v in value):
        return list(value), (len(value),)
    if isinstance(value, np.ndarray) and value.dtype.kind == "U":
        return [str(v) for v in value.ravel()], value.shape
    return None


def _attr_msg(w: "_Writer", name: str, value) -> bytes:
    nm = name.encode("utf-8") + b"\0"
    st = _strings(value)
    if st is not None:
        strs, shape = st
        dt, sp = _VLEN_STR, _space_msg(shape)
        data = b""
        for v in strs:
            b = v.encode("utf-8")
            addr, idx = w.gheap_put(b)
            data += struct.pack("<IQI", len(b), addr, idx)
    else:
        a = np.asarray(value)
        if a.dtype == object:
            raise TypeError("h5write: object arrays are not supported")
        dt, sp = _dtype_msg(a.dtype), _space_msg(a.shape)
        data = np.ascontiguousarray(a).tobytes()
    body = struct.pack("<BxHHH", 1, len(nm), len(dt), len(sp)) + _pad8(nm) + _pad8(dt) + _pad8(sp)
    return body + data


def _header(messages: List[Tuple[int, bytes]]) -> bytes:
    """object header v1: 16-byte prefix + messages (each 8-byte header, body padded to 8)"""
    body = b""
    for mtype, data in messages:
        data = _pad8(data)
        body += struct.pack("<HHBxxx", mtype, len(data), 0) + data
    return struct.pack("<BxHII", 1, len(messages), 1, len(body)) + b"\0" * 4 + body


GCOL_BYTES = 4096


class _Writer:
    def __init__(self, path):
        self.f = open(path, "wb")
        self.pos = 0
        self._gcol_addr = None      # current global heap collection: reserved in the file, filled in at close
        self._gcol_objs: List[bytes] = []
        self._gcol_used = 16
        self._gcols: List[Tuple[int, List[bytes]]] = []

    def gheap_put(self, data: bytes) -> Tuple[int, int]:
        """store ``data`` as a global heap object; returns (collection address, object index)"""
        need = 16 + len(data) + (-len(data) % 8)
        if need + 32 > GCOL_BYTES:
            raise ValueError("h5write: string too long for a global heap collection")
        if self._gcol_addr is None or self._gcol_used + need + 16 > GCOL_BYTES:
            self._gcol_addr = self.alloc(b"\0" * GCOL_BYTES)
            self._gcol_objs = []
            self._gcol_used = 16
            self._gcols.append((self._gcol_addr, self._gcol_objs))
        self._gcol_objs.append(data)
        self._gcol_used += need
        return self._gcol_addr, len(self._gcol_objs)

    def flush_gheaps(self):
        for addr, objs in self._gcols:
            body = b"GCOL" + struct.pack("<BxxxQ", 1, GCOL_BYTES)
            for i, d in enumerate(objs, start=1):
                body += struct.pack("<HHIQ", i, 1, 0, len(d)) + _pad8(d)
            free = GCOL_BYTES - len(body)
            body += struct.pack("<HHIQ", 0, 0, 0, free) + b"\0" * (free - 16)
            self.f.seek(addr)
            self.f.write(body)

    def alloc(self, data: bytes, align: int = 8) -> int:
        pad = -self.pos % align
        if pad:
            self.f.write(b"\0" * pad)
            self.pos += pad
        addr = self.pos
        self.f.write(data)
        self.pos += len(data)
        return addr


class Group:
    def __init__(self):
        self.children: Dict[str, object] = {}
        self.attrs: Dict[str, object] = {}

    def create_group(self, name: str) -> "Group":
        g = Group()
        self.children[name] = g
        return g

    def create_dataset(self, name: str, data: np.ndarray, compression: Optional[str] = None, attrs: Optional[dict] = None):
        self.children[name] = _Dataset(np.ascontiguousarray(data), compression, dict(attrs or {}))


class _Dataset:
    def __init__(self, data, compression, attrs):
        self.data, self.compression, self.attrs = data, compression, attrs


def _write_dataset(w: _Writer, d: _Dataset) -> int:
    a = d.data
    raw = a.tobytes()
    msgs = [(0x01, _space_msg(a.shape)), (0x03, _dtype_msg(a.dtype))]
    # fill value message v2: allocate late, write on allocation, no user-defined value (zeros)
    msgs.append((0x05, struct.pack("<BBBB", 2, 2, 0, 0)))
    if d.compression == "gzip" and a.ndim >= 1 and a.size:
        comp = zlib.compress(raw, 4)
        caddr = w.alloc(comp)
        rank = a.ndim
        # chunk B-tree v1, one leaf node, one chunk = the whole dataset
        key0 = struct.pack("<II", len(comp), 0) + b"".join(struct.pack("<Q", 0) for _ in range(rank + 1))
        key1 = struct.pack("<II", 0, 0) + b"".join(struct.pack("<Q", int(s)) for s in a.shape) + struct.pack("<Q", 0)
        node = b"TREE" + struct.pack("<BBHQQ", 1, 0, 1, UNDEF, UNDEF) + key0 + struct.pack("<Q", caddr) + key1
        baddr = w.alloc(node)
        msgs.append((0x0B, struct.pack("<BBxxxxxx", 1, 1) + struct.pack("<HHHH", 1, 0, 1, 1) + struct.pack("<I", 4) + b"\0" * 4))
        layout = (struct.pack("<BBB", 3, 2, rank + 1) + struct.pack("<Q", baddr) +
                  b"".join(struct.pack("<I", int(s)) for s in a.shape) + struct.pack("<I", a.dtype.itemsize))
    else:
        daddr = w.alloc(raw) if raw else UNDEF
        layout = struct.pack("<BBQQ", 3, 1, daddr, len(raw))
    msgs.append((0x08, layout))
    for k, v in d.attrs.items():
        msgs.append((0x0C, _attr_msg(w, k, v)))
    return w.alloc(_header(msgs))


def _write_group(w: _Writer, g: Group) -> Tuple[int, int, int]:
    """returns (object header address, B-tree address, local heap address)"""
    addrs = {}
    for name, child in g.children.items():
        addrs[name] = _write_group(w, child)[0] if isinstance(child, Group) else _write_dataset(w, child)
    names = sorted(addrs, key=lambda s: s.encode("utf-8"))          # the library orders symbols by strcmp
    # local heap: offset 0 holds the empty string; names are NUL-terminated, 8-byte aligned
    heap = bytearray(b"\0" * 8)
    offs = {}
    for n in names:
        offs[n] = len(heap)
        heap += _pad8(n.encode("utf-8") + b"\0")
    free_off = len(heap)
    heap += struct.pack("<QQ", 1, 16)                                  # one free block (next = 1: none, size 16)
    heap_data = w.alloc(bytes(heap))
    heap_addr = w.alloc(b"HEAP" + struct.pack("<BxxxQQQ", 0, len(heap), free_off, heap_data))
    # symbol-table nodes of <= 2*LEAF_K entries
    cap = 2 * LEAF_K
    level = []                                                         # (address, offset of the LAST name in the node)
    for lo in range(0, len(names), cap):
        part = names[lo:lo + cap]
        body = b"SNOD" + struct.pack("<BxH", 1, len(part))
        for n in part:
            body += struct.pack("<QQII", offs[n], addrs[n], 0, 0) + b"\0" * 16
        body += b"\0" * (40 * (cap - len(part)))
        level.append((w.alloc(body), offs[part[-1]]))
    if not level:                                                      # empty group: one empty SNOD
        level.append((w.alloc(b"SNOD" + struct.pack("<BxH", 1, 0) + b"\0" * (40 * cap)), 0))
    depth = 0
    while True:                                                        # B-tree levels of <= 2*INTERNAL_K children
        nodes = []
        width = 2 * INTERNAL_K
        groups = [level[i:i + width] for i in range(0, len(level), width)]
        for gi, part in enumerate(groups):
            first_key = 0 if gi == 0 else groups[gi - 1][-1][1]
            body = b"TREE" + struct.pack("<BBHQQ", 0, depth, len(part), UNDEF, UNDEF) + struct.pack("<Q", first_key)
            for addr, last in part:
                body += struct.pack("<QQ", addr, last)
            body += b"\0" * (16 * (width - len(part)))
            nodes.append((w.alloc(body), part[-1][1]))
        level = nodes
        depth += 1
        if len(level) == 1:
            break
    btree = level[0][0]
    msgs = [(0x11, struct.pack("<QQ", btree, heap_addr))]
    for k, v in g.attrs.items():
        msgs.append((0x0C, _attr_msg(w, k, v)))
    return w.alloc(_header(msgs)), btree, heap_addr


class File(Group):
    """with h5write.File(path) as f: f.attrs[...] = ...; g = f.create_group(name); g.create_dataset(name, data, ...)"""

    def __init__(self, path):
        super().__init__()
        self.path = path

    def __enter__(self):
        return self

    def __exit__(self, exc_type, *_):
        if exc_type is None:
            self.close()

    def close(self):
        w = _Writer(self.path)
        w.alloc(b"\0" * 96)                                            # superblock placeholder (56 + 40-byte root entry)
        root, btree, heap = _write_group(w, self)
        w.alloc(b"", 8)
        eof = w.pos
        w.flush_gheaps()
        sb = b"\x89HDF\r\n\x1a\n" + struct.pack("<BBBBBBBxHHI", 0, 0, 0, 0, 0, 8, 8, LEAF_K, INTERNAL_K, 0)
        sb += struct.pack("<QQQQ", 0, UNDEF, eof, UNDEF)
        sb += struct.pack("<QQII", 0, root, 1, 0) + struct.pack("<QQ", btree, heap)
        assert len(sb) == 96, len(sb)
        w.f.seek(0)
        w.f.write(sb)
        w.f.close()
