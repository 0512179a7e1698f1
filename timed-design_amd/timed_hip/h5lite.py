"""h5lite — a minimal, read-only, pure-Python/NumPy HDF5 reader.

Why: the two files either side of the hot path are HDF5 — the Keras legacy ``.h5`` model
(reference predict.py:121) and the aposteriori frame dataset (reference
design_utils/utils.py:238-251, read at :359-393 and :514-529) — and h5py is not installable on the
target interpreter.  Scope is exactly what h5py/HDF5-1.10 writes for those two producers with default
settings (``libver='earliest'``):

  superblock v0 (and v2/v3), object headers v1 (and v2), old-style groups (symbol table: B-tree v1 +
  local heap + SNOD) and compact new-style link messages, datasets with compact / contiguous /
  chunked (B-tree v1) layout, deflate + shuffle + fletcher32 filters, fixed-point / IEEE float /
  fixed string / variable-length string / enum(bool) / array datatypes, attributes v1-v3.

Anything else (dense link/attribute storage, layout v4, compound types, external links, ...) raises
``H5Unsupported`` rather than guessing.  The interface mirrors the slice of h5py the reference uses:
``File(path)``, ``obj.attrs`` (dict), ``group.keys()``, ``group[name]``, iteration, ``dataset[()]``,
``dataset.shape/dtype``.
"""
from __future__ import annotations

import mmap
import struct
import zlib
from typing import Dict, Iterator, List, Optional, Tuple

import numpy as np

UNDEF = 0xFFFFFFFFFFFFFFFF
SIGNATURE = b"\x89HDF\r\n\x1a\n"


class H5Unsupported(NotImplementedError):
    pass


class H5FormatError(ValueError):
    pass


# ---- datatype ---------------------------------------------------------------------------------------
class _DType:
    """Decoded datatype message."""
    __slots__ = ("cls", "size", "np", "vlen_str", "enum_bool", "str_pad", "base", "dims")

    def __init__(self):
        self.cls = -1; self.size = 0; self.np = None; self.vlen_str = False; self.enum_bool = False
        self.str_pad = 0; self.base = None; self.dims = ()


def _parse_dtype(buf: bytes, off: int) -> Tuple[_DType, int]:
    """Returns (dtype, bytes consumed)."""
    b0, bf0, bf1, bf2, size = struct.unpack_from("<BBBBI", buf, off)
    cls, ver = b0 & 0x0F, b0 >> 4
    dt = _DType(); dt.cls = cls; dt.size = size
    p = off + 8
    if cls == 0:  # fixed point
        order = ">" if bf0 & 1 else "<"
        signed = bool(bf0 & 8)
        dt.np = np.dtype(f"{order}{'i' if signed else 'u'}{size}")
        p += 4
    elif cls == 1:  # float
        order = ">" if bf0 & 1 else "<"
        if size not in (2, 4, 8):
            raise H5Unsupported(f"float of {size} bytes")
        dt.np = np.dtype(f"{order}f{size}")
        p += 12
    elif cls == 3:  # fixed-length string
        dt.str_pad = bf0 & 0x0F
        dt.np = np.dtype(f"S{size}")
    elif cls == 9:  # variable length
        vtype = bf0 & 0x0F
        base, used = _parse_dtype(buf, p)
        p += used
        if vtype == 1:
            dt.vlen_str = True
        else:
            raise H5Unsupported("variable-length sequences (non-string)")
        dt.base = base
    elif cls == 8:  # enum: h5py stores numpy bool as ENUM{FALSE=0,TRUE=1} over int8
        nmemb = bf0 | (bf1 << 8)
        base, used = _parse_dtype(buf, p)
        p += used
        names = []
        for _ in range(nmemb):
            e = buf.index(b"\0", p)
            names.append(buf[p:e])
            n = e - p + 1
            p += n if ver >= 3 else (n + 7) // 8 * 8
        p += nmemb * base.size
        dt.base = base
        dt.np = base.np
        dt.enum_bool = sorted(names) == [b"FALSE", b"TRUE"]
    elif cls == 10:  # array
        if ver < 3:
            rank = buf[p]; p += 4
            dims = struct.unpack_from(f"<{rank}I", buf, p); p += 4 * rank
            p += 4 * rank  # permutation indices
        else:
            rank = buf[p]; p += 1
            dims = struct.unpack_from(f"<{rank}I", buf, p); p += 4 * rank
        base, used = _parse_dtype(buf, p)
        p += used
        dt.base = base; dt.dims = tuple(dims)
        dt.np = np.dtype((base.np, tuple(dims)))
    elif cls == 6:
        raise H5Unsupported("compound datatypes")
    elif cls == 7:
        raise H5Unsupported("reference datatypes")
    else:
        raise H5Unsupported(f"datatype class {cls}")
    return dt, p - off


def _parse_dataspace(buf: bytes, off: int) -> Tuple[Optional[Tuple[int, ...]], int]:
    ver = buf[off]
    if ver == 1:
        rank, flags = buf[off + 1], buf[off + 2]
        p = off + 8
    elif ver == 2:
        rank, flags, stype = buf[off + 1], buf[off + 2], buf[off + 3]
        p = off + 4
        if stype == 2:
            return None, 4  # null dataspace
    else:
        raise H5Unsupported(f"dataspace version {ver}")
    dims = struct.unpack_from(f"<{rank}Q", buf, p)
    p += 8 * rank
    if flags & 1:
        p += 8 * rank
    return tuple(dims), p - off


# ---- file ---------------------------------------------------------------------------------------------
class File:
    def __init__(self, path, mode: str = "r"):
        if mode != "r":
            raise H5Unsupported("h5lite is read-only")
        self._f = open(path, "rb")
        try:
            self._m = mmap.mmap(self._f.fileno(), 0, access=mmap.ACCESS_READ)
        except ValueError:
            self._f.close()
            raise H5FormatError(f"{path}: empty file")
        self.filename = str(path)
        self._gheap: Dict[int, Dict[int, bytes]] = {}
        self._groups: Dict[int, "Group"] = {}
        base = self._find_superblock()
        self._base = base
        m = self._m
        ver = m[base + 8]
        if ver in (0, 1):
            so, sl = m[base + 13], m[base + 14]
            if (so, sl) != (8, 8):
                raise H5Unsupported("offset/length sizes other than 8")
            p = base + 24 + (4 if ver == 1 else 0)
            p += 32  # base addr, free-space addr, eof addr, driver info addr
            # root group symbol table entry
            _link_off, ohdr, cache, _r = struct.unpack_from("<QQII", m, p)
            self._root_addr = ohdr
        elif ver in (2, 3):
            so, sl = m[base + 9], m[base + 10]
            if (so, sl) != (8, 8):
                raise H5Unsupported("offset/length sizes other than 8")
            _baseaddr, _ext, _eof, root = struct.unpack_from("<QQQQ", m, base + 12)
            self._root_addr = root
        else:
            raise H5Unsupported(f"superblock version {ver}")
        self._root = Group(self, self._root_addr, "/")

    def _find_superblock(self) -> int:
        off = 0
        while off + 8 <= len(self._m):
            if self._m[off:off + 8] == SIGNATURE:
                return off
            off = 512 if off == 0 else off * 2
        raise H5FormatError("not an HDF5 file")

    # h5py-like surface, delegated to the root group
    @property
    def attrs(self): return self._root.attrs
    def keys(self): return self._root.keys()
    def __iter__(self): return iter(self._root)
    def __getitem__(self, k): return self._root[k]
    def __contains__(self, k): return k in self._root
    def __len__(self): return len(self._root)
    def __enter__(self): return self
    def __exit__(self, *a): self.close()

    def close(self):
        if self._m is not None:
            self._m.close(); self._f.close(); self._m = None

    # ---- low level ----------------------------------------------------------------------------------
    def _messages(self, addr: int) -> List[Tuple[int, int, bytes]]:
        """All header messages of the object at addr as (type, flags, payload)."""
        m = self._m
        a = self._base + addr
        out: List[Tuple[int, int, bytes]] = []
        if m[a:a + 4] == b"OHDR":  # version 2
            if m[a + 4] != 2:
                raise H5Unsupported("object header version")
            flags = m[a + 5]
            p = a + 6
            if flags & 0x20:
                p += 16
            if flags & 0x10:
                p += 4
            szbytes = 1 << (flags & 3)
            chunk0 = int.from_bytes(m[p:p + szbytes], "little")
            p += szbytes
            track_order = bool(flags & 4)
            blocks = [(p, chunk0)]
            while blocks:
                bp, bl = blocks.pop(0)
                end = bp + bl
                q = bp
                while q + 4 <= end:
                    mtype = m[q]; msize = struct.unpack_from("<H", m, q + 1)[0]; mflags = m[q + 3]
                    q += 4 + (2 if track_order else 0)
                    data = bytes(m[q:q + msize])
                    q += msize
                    if mtype == 0x10:
                        co, cl = struct.unpack_from("<QQ", data, 0)
                        blocks.append((self._base + co + 4, cl - 8))  # skip "OCHK", drop checksum
                    elif mtype != 0:
                        out.append((mtype, mflags, data))
            return out
        ver = m[a]
        if ver != 1:
            raise H5FormatError(f"bad object header at {addr:#x}")
        nmsg, _ref, hsize = struct.unpack_from("<HII", m, a + 2)
        blocks = [(a + 16, hsize)]
        while blocks and len(out) < 100000:
            bp, bl = blocks.pop(0)
            q, end = bp, bp + bl
            while q + 8 <= end:
                mtype, msize, mflags = struct.unpack_from("<HHB", m, q)
                q += 8
                data = bytes(m[q:q + msize])
                q += msize
                if mtype == 0x10:
                    co, cl = struct.unpack_from("<QQ", data, 0)
                    blocks.append((self._base + co, cl))
                elif mtype != 0:
                    out.append((mtype, mflags, data))
        return out

    def _global_heap_object(self, addr: int, index: int) -> bytes:
        col = self._gheap.get(addr)
        if col is None:
            m = self._m
            a = self._base + addr
            if m[a:a + 4] != b"GCOL":
                raise H5FormatError("bad global heap collection")
            size = struct.unpack_from("<Q", m, a + 8)[0]
            col = {}
            p, end = a + 16, a + size
            while p + 16 <= end:
                idx, _rc, _r, osz = struct.unpack_from("<HHIQ", m, p)
                if idx == 0:
                    break
                col[idx] = bytes(m[p + 16:p + 16 + osz])
                p += 16 + (osz + 7) // 8 * 8
            self._gheap[addr] = col
        return col[index]

    def _decode(self, dt: _DType, shape, raw: bytes):
        """raw bytes of `shape` elements -> numpy array / python value, h5py-style."""
        n = int(np.prod(shape)) if shape else 1
        if dt.vlen_str:
            vals = []
            for i in range(n):
                ln, addr, idx = struct.unpack_from("<IQI", raw, 16 * i)
                vals.append(self._global_heap_object(addr, idx)[:ln].decode("utf-8", "replace") if ln or addr else "")
            if not shape:
                return vals[0]
            return np.array(vals, dtype=object).reshape(shape)
        if dt.cls == 3:
            arr = np.frombuffer(raw, dtype=dt.np, count=n)
            if not shape:
                return bytes(arr[0]).rstrip(b"\0 ").decode("utf-8", "replace") if dt.str_pad != 2 else bytes(arr[0]).decode()
            return arr.reshape(shape).copy()
        arr = np.frombuffer(raw, dtype=dt.np, count=n)
        if dt.enum_bool:
            arr = arr.astype(bool)
        arr = arr.reshape(tuple(shape) + tuple(dt.dims) if dt.cls == 10 else shape)
        if not shape and dt.cls != 10:
            return arr[()] if arr.dtype != bool else bool(arr[()])
        return arr.copy()


class _Object:
    def __init__(self, f: File, addr: int, name: str):
        self._f, self._addr, self.name = f, addr, name
        self._msgs = None
        self._attrs = None

    def _messages(self):
        if self._msgs is None:
            self._msgs = self._f._messages(self._addr)
        return self._msgs

    @property
    def attrs(self) -> dict:
        if self._attrs is None:
            out = {}
            for mtype, _fl, d in self._messages():
                if mtype == 0x15:
                    fheap = struct.unpack_from("<Q", d, 2 + (2 if d[1] & 1 else 0))[0]
                    if fheap != UNDEF:
                        raise H5Unsupported(f"{self.name}: dense attribute storage")
                if mtype != 0x0C:
                    continue
                ver = d[0]
                if ver == 1:
                    nsz, dsz, ssz = struct.unpack_from("<HHH", d, 2)
                    p = 8
                    name = d[p:p + nsz].split(b"\0")[0].decode(); p += (nsz + 7) // 8 * 8
                    dt, _ = _parse_dtype(d, p); p += (dsz + 7) // 8 * 8
                    shape, _ = _parse_dataspace(d, p); p += (ssz + 7) // 8 * 8
                elif ver in (2, 3):
                    nsz, dsz, ssz = struct.unpack_from("<HHH", d, 2)
                    p = 8 + (1 if ver == 3 else 0)
                    name = d[p:p + nsz].split(b"\0")[0].decode(); p += nsz
                    dt, _ = _parse_dtype(d, p); p += dsz
                    shape, _ = _parse_dataspace(d, p); p += ssz
                else:
                    raise H5Unsupported(f"attribute message version {ver}")
                if shape is None:
                    out[name] = None
                    continue
                out[name] = self._f._decode(dt, shape, d[p:])
            self._attrs = out
        return self._attrs


def fletcher32(data: bytes) -> Tuple[int, int]:
    """The checksum of HDF5's fletcher32 filter over ``data`` — two 16-bit one's-complement style sums over big-endian 16-bit
    words, an odd last byte taken as the high byte of a word — and the variant over byte-swapped words that libraries before
    1.6.3 wrote; the library accepts either on read (and fails the read otherwise), so do the readers here."""
    def run(words: np.ndarray, tail: Optional[int]) -> int:
        s1 = s2 = 0
        w = words.astype(np.uint64)
        for lo in range(0, len(w), 360):                       # the library folds every 360 words; folding is exact mod 65535,
            blk = w[lo:lo + 360]                                # so the blocks only have to be small enough not to overflow
            n = len(blk)
            s2 += n * s1 + int((blk * np.arange(n, 0, -1, dtype=np.uint64)).sum())
            s1 += int(blk.sum())
            s1 = (s1 & 0xffff) + (s1 >> 16)
            s2 = (s2 & 0xffff) + (s2 >> 16)
        if tail is not None:
            s1 += tail << 8
            s2 += s1
            s1 = (s1 & 0xffff) + (s1 >> 16)
            s2 = (s2 & 0xffff) + (s2 >> 16)
        s1 = (s1 & 0xffff) + (s1 >> 16)
        s2 = (s2 & 0xffff) + (s2 >> 16)
        return ((s2 & 0xffff) << 16) | (s1 & 0xffff)
    buf = np.frombuffer(data, dtype=np.uint8)
    n2 = len(buf) // 2
    pairs = buf[:2 * n2].reshape(n2, 2).astype(np.uint16)
    tail = int(buf[-1]) if len(buf) % 2 else None
    return run((pairs[:, 0] << 8) | pairs[:, 1], tail), run((pairs[:, 1] << 8) | pairs[:, 0], tail)


class Group(_Object):
    def __init__(self, f, addr, name):
        super().__init__(f, addr, name)
        self._links: Optional[Dict[str, int]] = None

    def _load(self) -> Dict[str, int]:
        if self._links is not None:
            return self._links
        links: Dict[str, int] = {}
        f = self._f
        for mtype, _fl, d in self._messages():
            if mtype == 0x11:  # symbol table
                btree, heap = struct.unpack_from("<QQ", d, 0)
                heap_data = self._local_heap(heap)
                if not self._links_native(btree, heap_data, links):
                    self._walk_btree(btree, heap_data, links)
            elif mtype == 0x06:  # link message
                ver, flags = d[0], d[1]
                p = 2
                ltype = 0
                if flags & 8:
                    ltype = d[p]; p += 1
                if flags & 4:
                    p += 8
                if flags & 16:
                    p += 1
                lsz = 1 << (flags & 3)
                nlen = int.from_bytes(d[p:p + lsz], "little"); p += lsz
                name = d[p:p + nlen].decode(); p += nlen
                if ltype != 0:
                    continue  # soft/external links are not followed
                links[name] = struct.unpack_from("<Q", d, p)[0]
            elif mtype == 0x02:  # link info
                flags = d[1]
                p = 2 + (8 if flags & 1 else 0)
                fheap = struct.unpack_from("<Q", d, p)[0]
                if fheap != UNDEF:
                    raise H5Unsupported(f"{self.name}: dense link storage (file written with libver='latest')")
        self._links = links
        return links

    def _links_native(self, btree: int, heap: Tuple[int, int], links: Dict[str, int]) -> bool:
        """the whole symbol table in one native call (libtimedhip th_h5_group_links); False when the library is not there
        or the call declines (the interpreter walk below then runs and raises its own format errors)"""
        import ctypes as C
        try:
            from . import _lib
            lib = _lib.load()
        except Exception:
            return False
        f = self._f
        cap = max(int(heap[1]), 16)
        names = np.empty(cap + 8, dtype=np.uint8)
        addrs = np.empty(cap // 2 + 8, dtype=np.int64)       # a name takes at least 2 heap bytes ("x\0")
        n, used = C.c_int64(0), C.c_int64(0)
        whole = np.frombuffer(f._m, dtype=np.uint8)
        try:
            rc = lib.th_h5_group_links(whole.ctypes.data_as(C.c_void_p), whole.size, f._base, int(btree), int(heap[0]), int(heap[1]),
                                       names.ctypes.data_as(C.c_void_p), names.size, addrs.ctypes.data_as(C.POINTER(C.c_int64)), addrs.size,
                                       C.byref(n), C.byref(used))
        finally:
            del whole
        if rc != 0:
            return False
        if n.value:
            try:
                keys = names[:used.value - 1].tobytes().decode().split("\0")
            except UnicodeDecodeError:
                return False
            if len(keys) != n.value:
                return False
            links.update(zip(keys, addrs[:n.value].tolist()))
        return True

    def _local_heap(self, addr: int) -> Tuple[int, int]:
        m = self._f._m
        a = self._f._base + addr
        if m[a:a + 4] != b"HEAP":
            raise H5FormatError("bad local heap")
        dsize, _free, daddr = struct.unpack_from("<QQQ", m, a + 8)
        return self._f._base + daddr, dsize

    def _walk_btree(self, addr: int, heap: Tuple[int, int], links: Dict[str, int]):
        m = self._f._m
        a = self._f._base + addr
        if m[a:a + 4] != b"TREE":
            raise H5FormatError("bad B-tree node")
        ntype, level, used = struct.unpack_from("<BBH", m, a + 4)
        if ntype != 0:
            raise H5FormatError("expected a group B-tree")
        p = a + 24
        for i in range(used):
            child = struct.unpack_from("<Q", m, p + 8)[0]  # key_i (8), child_i (8)
            p += 16
            if level > 0:
                self._walk_btree(child, heap, links)
            else:
                s = self._f._base + child
                if m[s:s + 4] != b"SNOD":
                    raise H5FormatError("bad symbol table node")
                nsym = struct.unpack_from("<H", m, s + 6)[0]
                q = s + 8
                for _ in range(nsym):
                    noff, ohdr = struct.unpack_from("<QQ", m, q)
                    q += 40
                    e = m.find(b"\0", heap[0] + noff)
                    links[bytes(m[heap[0] + noff:e]).decode()] = ohdr

    # ---- h5py-like surface --------------------------------------------------------------------------
    def keys(self):
        return sorted(self._load().keys())  # h5py iterates old-style groups in name order

    def __iter__(self) -> Iterator[str]:
        return iter(self.keys())

    def __len__(self):
        return len(self._load())

    def __contains__(self, name):
        try:
            self[name]
            return True
        except KeyError:
            return False

    def __getitem__(self, name: str):
        obj = self
        for part in str(name).strip("/").split("/"):
            if not isinstance(obj, Group):
                raise KeyError(name)
            links = obj._load()
            if part not in links:
                raise KeyError(f"{name!r} not found in {self.name}")
            addr = links[part]
            cached = self._f._groups.get(addr)     # groups are reused: their link tables are parsed once per file
            if cached is not None:
                obj = cached
                continue
            child_name = (obj.name.rstrip("/") + "/" + part)
            kinds = {t for t, _f, _d in self._f._messages(addr)}
            if 0x08 in kinds:
                obj = Dataset(self._f, addr, child_name)
            else:
                obj = Group(self._f, addr, child_name)
                self._f._groups[addr] = obj
        return obj

    def items(self):
        return [(k, self[k]) for k in self.keys()]


class Dataset(_Object):
    def __init__(self, f, addr, name):
        super().__init__(f, addr, name)
        self._dt = None; self._shape = None; self._layout = None; self._filters = []
        self._fill = None          # bytes of a user-defined fill value (what never-written elements read as), else None = zeros
        for mtype, _fl, d in self._messages():
            if mtype == 0x03:
                self._dt, _ = _parse_dtype(d, 0)
            elif mtype == 0x01:
                self._shape, _ = _parse_dataspace(d, 0)
            elif mtype == 0x08:
                self._layout = d
            elif mtype == 0x0B:
                self._filters = self._parse_filters(d)
            elif mtype in (0x04, 0x05):
                fv = self._parse_fill(mtype, d)
                if fv is not None:
                    self._fill = fv
        if self._dt is None or self._layout is None:
            raise H5FormatError(f"{name}: incomplete dataset header")

    @staticmethod
    def _parse_fill(mtype: int, d: bytes):
        """fill value bytes of a fill-value message (0x05, versions 1-3) or an old fill-value message (0x04); None when
        no value is defined (the library default: zeros)"""
        try:
            if mtype == 0x04:
                size = struct.unpack_from("<I", d, 0)[0]
                return bytes(d[4:4 + size]) if size else None
            ver = d[0]
            if ver in (1, 2):
                defined = d[3]
                if ver == 2 and not defined:
                    return None
                size = struct.unpack_from("<I", d, 4)[0]
                return bytes(d[8:8 + size]) if size else None
            if ver == 3:
                if not (d[1] & 0x20):
                    return None
                size = struct.unpack_from("<I", d, 2)[0]
                return bytes(d[6:6 + size]) if size else None
        except (struct.error, IndexError):
            pass
        return None

    @property
    def _nonzero_fill(self) -> bool:
        return self._fill is not None and any(self._fill)

    def _filled(self, shape, esz) -> np.ndarray:
        """an array of `shape` void elements holding the dataset's fill value"""
        out = np.zeros(shape, dtype=np.dtype(f"V{esz}"))
        if self._nonzero_fill and len(self._fill) == esz:
            out[...] = np.frombuffer(self._fill, dtype=out.dtype)[0]
        return out

    @staticmethod
    def _parse_filters(d: bytes):
        ver, n = d[0], d[1]
        out = []
        p = 8 if ver == 1 else 2
        for _ in range(n):
            fid = struct.unpack_from("<H", d, p)[0]
            if ver == 1 or fid >= 256:
                nlen = struct.unpack_from("<H", d, p + 2)[0]
                _flags, ncd = struct.unpack_from("<HH", d, p + 4)
                p += 8
                p += (nlen + 7) // 8 * 8 if ver == 1 else nlen
            else:
                _flags, ncd = struct.unpack_from("<HH", d, p + 2)
                p += 6
            cd = struct.unpack_from(f"<{ncd}I", d, p)
            p += 4 * ncd
            if ver == 1 and ncd % 2:
                p += 4
            out.append((fid, cd))
        return out

    @property
    def shape(self): return self._shape
    @property
    def dtype(self): return np.dtype(bool) if self._dt.enum_bool else self._dt.np
    @property
    def ndim(self): return len(self._shape or ())

    def _unfilter(self, raw: bytes, mask: int, elsize: int) -> bytes:
        for i in reversed(range(len(self._filters))):
            if mask & (1 << i):
                continue
            fid, cd = self._filters[i]
            if fid == 1:
                raw = zlib.decompress(raw)
            elif fid == 2:
                n = len(raw) // elsize
                raw = np.frombuffer(raw[:n * elsize], dtype=np.uint8).reshape(elsize, n).T.tobytes() + raw[n * elsize:]
            elif fid == 3:
                if len(raw) < 4:
                    raise H5FormatError(f"{self.name}: fletcher32 chunk shorter than its checksum")
                stored = int.from_bytes(raw[-4:], "little")
                raw = raw[:-4]
                if stored not in fletcher32(raw):
                    raise H5FormatError(f"{self.name}: fletcher32 checksum of a chunk does not match its data")
            else:
                raise H5Unsupported(f"{self.name}: filter id {fid}")
        return raw

    def _read_raw(self) -> bytes:
        f, m, d = self._f, self._f._m, self._layout
        ver = d[0]
        shape = self._shape or ()
        n = int(np.prod(shape)) if shape else 1
        esz = self._dt.size if not self._dt.vlen_str else 16
        if ver != 3:
            raise H5Unsupported(f"{self.name}: data layout version {ver}")
        cls = d[1]
        if cls == 0:
            size = struct.unpack_from("<H", d, 2)[0]
            return d[4:4 + size]
        if cls == 1:
            addr, size = struct.unpack_from("<QQ", d, 2)
            if addr == UNDEF:
                return self._filled(shape, esz).tobytes() if shape else (self._fill if self._nonzero_fill else bytes(esz))
            return bytes(m[f._base + addr:f._base + addr + size])
        if cls == 2:
            rank = d[2]
            btree = struct.unpack_from("<Q", d, 3)[0]
            cdims = struct.unpack_from(f"<{rank}I", d, 11)
            chunk = cdims[:-1]
            if cdims[-1] != esz:
                raise H5FormatError(f"{self.name}: chunk element size")
            out = self._filled(shape, esz)          # never-allocated chunks read as the fill value
            if btree != UNDEF:
                self._walk_chunks(btree, rank, chunk, esz, out)
            return out.tobytes()
        raise H5Unsupported(f"{self.name}: layout class {cls}")

    def _walk_chunks(self, addr, rank, chunk, esz, out):
        f, m = self._f, self._f._m
        a = f._base + addr
        if m[a:a + 4] != b"TREE":
            raise H5FormatError("bad chunk B-tree")
        ntype, level, used = struct.unpack_from("<BBH", m, a + 4)
        if ntype != 1:
            raise H5FormatError("expected a chunk B-tree")
        ksz = 8 + 8 * rank
        p = a + 24
        nel = 1
        for c in chunk:
            nel *= c
        written = 0
        for _ in range(used):
            csize, mask = struct.unpack_from("<II", m, p)
            offs = struct.unpack_from(f"<{rank}Q", m, p + 8)
            child = struct.unpack_from("<Q", m, p + ksz)[0]
            p += ksz + 8
            if level > 0:
                written += self._walk_chunks(child, rank, chunk, esz, out)
                continue
            raw = self._unfilter(m[f._base + child:f._base + child + csize], mask, esz)
            block = np.frombuffer(raw, dtype=out.dtype, count=nel).reshape(chunk)
            sl_out, sl_in = [], []
            for o, c, s in zip(offs[:-1], chunk, out.shape):
                hi = min(o + c, s)
                sl_out.append(slice(o, hi)); sl_in.append(slice(0, hi - o))
            out[tuple(sl_out)] = block[tuple(sl_in)]
            written += 1
        return written

    def chunked_geometry(self):
        """(btree address, shape, chunk dims, element size, filter ids) of a plain numeric chunked dataset whose
        bytes can be placed directly (layout v3, B-tree v1, little-endian), else None.  Input of
        ``read_many_direct`` / the native th_h5_read_chunked."""
        d = self._layout
        if self._shape is None or not self._shape or self._dt.vlen_str or d is None or d[0] != 3 or d[1] != 2:
            return None
        if self._nonzero_fill:          # the native reader zero-fills what was never written: only valid for a zero fill value
            return None
        npdt = np.dtype(bool) if self._dt.enum_bool else self._dt.np
        if npdt is None or npdt.byteorder == ">" or npdt.kind not in "fiub":
            return None
        rank = d[2]
        btree = struct.unpack_from("<Q", d, 3)[0]
        cdims = struct.unpack_from(f"<{rank}I", d, 11)
        if rank - 1 != len(self._shape) or cdims[-1] != self._dt.size:
            return None
        return btree, tuple(self._shape), tuple(cdims[:-1]), self._dt.size, tuple(fid for fid, _cd in self._filters)

    def read_direct(self, dest: np.ndarray) -> bool:
        """Fill the C-contiguous array ``dest`` (same shape and item size as the dataset) straight from the file:
        chunks are decompressed into their place, no intermediate whole-dataset copies.  Returns False when the
        dataset needs the general path (strings, byte-swapped or compact storage, shape mismatch); thread-safe
        (read-only mmap; zlib and the NumPy copies release the GIL), which is what load_batch uses to inflate
        many residues' frames in parallel."""
        shape = self._shape
        if shape is None or self._dt.vlen_str or tuple(dest.shape) != tuple(shape) or not dest.flags.c_contiguous:
            return False
        if self._nonzero_fill:
            return False
        esz = self._dt.size
        npdt = np.dtype(bool) if self._dt.enum_bool else self._dt.np
        if dest.dtype.itemsize != esz or npdt.byteorder == ">" or (dest.dtype != npdt and dest.dtype.kind != npdt.kind):
            return False
        d = self._layout
        if d[0] != 3:
            return False
        f, m = self._f, self._f._m
        raw_view = dest.view(np.dtype(f"V{esz}")) if esz > 1 else dest.view(np.uint8)
        if d[1] == 1:                                   # contiguous
            addr, size = struct.unpack_from("<QQ", d, 2)
            if addr == UNDEF:
                dest.view(np.uint8).fill(0)
            else:
                np.copyto(dest.view(np.uint8).reshape(-1), np.frombuffer(m, dtype=np.uint8, count=size, offset=f._base + addr))
            return True
        if d[1] != 2:
            return False
        rank = d[2]
        btree = struct.unpack_from("<Q", d, 3)[0]
        cdims = struct.unpack_from(f"<{rank}I", d, 11)
        if cdims[-1] != esz:
            raise H5FormatError(f"{self.name}: chunk element size")
        chunk = cdims[:-1]
        n_chunks = 1
        for s, c in zip(shape, chunk):
            n_chunks *= -(-s // c)
        if btree == UNDEF:
            dest.view(np.uint8).fill(0)
            return True
        out = raw_view if esz > 1 else dest.view(np.dtype("V1"))
        written = self._walk_chunks(btree, rank, chunk, esz, out)
        if written != n_chunks:                         # unallocated chunks hold the fill value: start from zeros
            dest.view(np.uint8).fill(0)
            self._walk_chunks(btree, rank, chunk, esz, out)
        return True

    def __getitem__(self, key):
        if self._shape is None:
            return None
        arr = self._f._decode(self._dt, self._shape, self._read_raw())
        if key == () or key is Ellipsis:
            return arr
        return arr[key]

    def __array__(self, dtype=None, copy=None):
        a = self[()]
        return a.astype(dtype) if dtype is not None else a

    def __len__(self):
        return self._shape[0]


def read_many_direct(datasets, dests) -> List[bool]:
    """Fill ``dests[i]`` (C-contiguous arrays) from ``datasets[i]`` for a whole batch.  Datasets that are plain
    chunked numeric arrays of one file and one geometry go to libtimedhip's ``th_h5_read_chunked`` in a single call
    (B-tree walk, inflate and placement on host threads, no interpreter work per chunk); the rest use
    ``Dataset.read_direct``.  Returns, per dataset, whether it was filled (False: caller must use ``ds[()]``)."""
    import ctypes as C
    done = [False] * len(datasets)
    groups: Dict[tuple, List[int]] = {}
    for i, (ds, dest) in enumerate(zip(datasets, dests)):
        geo = ds.chunked_geometry() if isinstance(ds, Dataset) else None
        kind_ok = geo is not None and (dest.dtype.kind == ds.dtype.kind or (ds.dtype.kind == "b" and dest.dtype == np.bool_))
        if (geo is not None and kind_ok and tuple(dest.shape) == geo[1] and dest.flags.c_contiguous and dest.dtype.itemsize == geo[3]
                and dest.flags.writeable):
            groups.setdefault((id(ds._f),) + geo[1:], []).append(i)
    if groups:
        try:
            from . import _lib
            lib = _lib.load()
        except Exception:          # no native library: the pure-Python chunk walk below still works
            lib = None
        for key, idx in groups.items():
            if lib is None:
                break
            f = datasets[idx[0]]._f
            _fid, shape, chunk, esz, filters = key
            whole = np.frombuffer(f._m, dtype=np.uint8)
            try:
                n, rank = len(idx), len(shape)
                addrs = (C.c_int64 * n)(*[datasets[i].chunked_geometry()[0] if datasets[i].chunked_geometry()[0] != UNDEF else -1
                                          for i in idx])
                ptrs = (C.c_void_p * n)(*[dests[i].ctypes.data for i in idx])
                rc = lib.th_h5_read_chunked(whole.ctypes.data_as(C.c_void_p), whole.size, f._base, n, addrs, ptrs, rank,
                                            (C.c_int64 * rank)(*shape), (C.c_int64 * rank)(*chunk), esz, len(filters),
                                            (C.c_int * max(1, len(filters)))(*filters), 0)
            finally:
                del whole                      # release the exported buffer so the file can be closed
            if rc == 0:
                for i in idx:
                    done[i] = True
    for i, (ds, dest) in enumerate(zip(datasets, dests)):
        if not done[i] and isinstance(ds, Dataset):
            done[i] = ds.read_direct(dest)
    return done


def resolve_many(f: File, addrs, num_attr: Optional[str] = None, num_len: int = 0, str_attr: Optional[str] = None,
                 str_len: int = 16):
    """Native bulk resolution of dataset object headers (libtimedhip th_h5_resolve): one call for a whole batch instead
    of a Python header parse + two attribute decodes per residue.  Returns a dict of arrays
    (status, btree, geom, num [n, num_len] float64, strs list) or None when the native library is unavailable."""
    import ctypes as C
    try:
        from . import _lib
        lib = _lib.load()
    except Exception:
        return None
    n = len(addrs)
    a = np.ascontiguousarray(np.asarray(addrs, dtype=np.int64))
    status = np.zeros(n, dtype=np.int32)
    btree = np.zeros(n, dtype=np.int64)
    geom = np.zeros(40, dtype=np.int64)
    num = np.zeros((n, max(num_len, 1)), dtype=np.float64) if num_attr else None
    sbuf = np.zeros((n, str_len), dtype=np.uint8) if str_attr else None
    whole = np.frombuffer(f._m, dtype=np.uint8)
    try:
        rc = lib.th_h5_resolve(whole.ctypes.data_as(C.c_void_p), whole.size, f._base, n, a.ctypes.data_as(C.POINTER(C.c_int64)),
                               num_attr.encode() if num_attr else None, num.ctypes.data if num is not None else None, num_len,
                               str_attr.encode() if str_attr else None, sbuf.ctypes.data if sbuf is not None else None, str_len,
                               btree.ctypes.data_as(C.POINTER(C.c_int64)), geom.ctypes.data_as(C.POINTER(C.c_int64)),
                               status.ctypes.data_as(C.POINTER(C.c_int)), 0)
    finally:
        del whole
    if rc != 0:
        return None
    strs = None
    if sbuf is not None:
        # fixed-width rows -> str: "S<n>" drops the trailing NULs; a NUL inside a row ends the string as the C side wrote it
        if str_len > 1 and not np.any((sbuf[:, :-1] == 0) & (sbuf[:, 1:] != 0)):
            try:
                strs = np.char.rstrip(sbuf.view(f"S{str_len}").ravel(), b" ").astype(str).tolist()
            except UnicodeDecodeError:
                strs = None
        if strs is None:
            strs = [bytes(row).split(b"\0", 1)[0].rstrip(b" ").decode("utf-8", "replace") for row in sbuf]
    return dict(status=status, btree=btree, geom=geom, num=num, strs=strs)


def read_resolved(f: File, resolved: dict, rows, dests, as_float32: bool = False) -> bool:
    """Inflate and place the datasets ``rows`` (indices into a resolve_many result whose status bit 1 is set) straight
    into ``dests`` with one native call (th_h5_read_chunked_as).  ``as_float32`` converts float64 data to float32 on
    the fly.  False when the geometry is not a plain numeric one (the caller then reads dataset by dataset)."""
    import ctypes as C
    from . import _lib
    lib = _lib.load()
    g = resolved["geom"]
    rank, esz, cls, nf = int(g[0]), int(g[15]), int(g[16]), int(g[18])
    if rank < 1 or cls not in (0, 1, 8) or (as_float32 and not (cls == 1 and esz == 8)):
        return False
    shape, chunk = [int(x) for x in g[1:1 + rank]], [int(x) for x in g[8:8 + rank]]
    filters = [int(x) for x in g[19:19 + nf]]
    n = len(rows)
    addrs = (C.c_int64 * n)(*[int(resolved["btree"][i]) for i in rows])
    ptrs = (C.c_void_p * n)(*[d.ctypes.data for d in dests])
    whole = np.frombuffer(f._m, dtype=np.uint8)
    if int(g[27]) == 1:                       # contiguous storage
        try:
            rc = lib.th_h5_read_contiguous_as(whole.ctypes.data_as(C.c_void_p), whole.size, f._base, n, addrs, ptrs,
                                              int(np.prod(shape)), esz, 1 if as_float32 else 0)
        finally:
            del whole
        return rc == 0
    try:
        rc = lib.th_h5_read_chunked_as(whole.ctypes.data_as(C.c_void_p), whole.size, f._base, n, addrs, ptrs, rank,
                                       (C.c_int64 * rank)(*shape), (C.c_int64 * rank)(*chunk), esz, len(filters),
                                       (C.c_int * max(1, len(filters)))(*filters), 0, 1 if as_float32 else 0)
    finally:
        del whole
    return rc == 0


MIN_DECODE = 64          # datasets per th_h5_decode_device call below which an out-of-memory batch is not split further


def decode_resolved_device(f: File, resolved: dict, d_out: int, device: int, as_float32: bool = False) -> bool:
    """Every dataset of a resolve_many result (all with status bit 1: one shared chunked geometry) inflated ON THE GPU
    straight into device memory at ``d_out`` — [n, *shape] float32 when ``as_float32`` (float64 data) or the stored element
    type — by libtimedhip th_h5_decode_device: only the compressed bytes cross PCIe.  False when the file uses anything but
    the deflate-only pipeline (the caller then reads through the host path)."""
    import ctypes as C
    from . import _lib
    lib = _lib.load()
    g = resolved["geom"]
    rank, esz, cls, nf = int(g[0]), int(g[15]), int(g[16]), int(g[18])
    if int(g[27]) != 2 or rank < 1 or cls not in (0, 1, 8) or (as_float32 and not (cls == 1 and esz == 8)):
        return False
    filters = [int(x) for x in g[19:19 + nf]]
    if filters not in ([1], [2, 1]):            # deflate, or shuffle + deflate
        return False
    shape, chunk = [int(x) for x in g[1:1 + rank]], [int(x) for x in g[8:8 + rank]]
    n = len(resolved["btree"])
    addrs = np.ascontiguousarray(resolved["btree"], dtype=np.int64)
    frame_bytes = int(np.prod(shape)) * (4 if as_float32 else esz)
    whole = np.frombuffer(f._m, dtype=np.uint8)

    def decode(lo: int, hi: int) -> int:
        part = addrs[lo:hi]
        return lib.th_h5_decode_device(whole.ctypes.data_as(C.c_void_p), whole.size, f._base, hi - lo, part.ctypes.data_as(C.POINTER(C.c_int64)),
                                       rank, (C.c_int64 * rank)(*shape), (C.c_int64 * rank)(*chunk), esz, len(filters),
                                       (C.c_int * len(filters))(*filters), 1 if as_float32 else 0, int(device),
                                       C.c_void_p(int(d_out) + lo * frame_bytes))
    try:
        # the decoder's token arena is ~5 bytes per uncompressed byte of the batch (9 GB for 4096 float64 frames): when the device
        # has no room for it (TH_ENOMEM; its scratch is freed then) the batch is decoded in halves, quarters, ... — never below
        # MIN_DECODE datasets per call, where the host reader is the better path
        pieces = [(0, n)]
        while pieces:
            lo, hi = pieces.pop()
            rc = decode(lo, hi)
            if rc == _lib.TH_ENOMEM and hi - lo >= 2 * MIN_DECODE:
                mid = (lo + hi) // 2
                pieces += [(mid, hi), (lo, mid)]
                continue
            if rc != 0:
                break
    finally:
        del whole
    if rc == -4:            # TH_EUNSUP: not an error, just not this path
        return False
    _lib.check(rc)
    return True
