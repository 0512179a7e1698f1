"""Native text output for the prediction writers: byte-identical to
``np.savetxt(f, matrix, delimiter=",")`` (reference design_utils/utils.py:768-771, predict.py:145-146),
formatted by libtimedhip's host-side ``th_format_csv`` instead of Python's per-row ``%`` operator."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib

_DTYPES = {np.dtype(np.float16): _lib.TH_F16, np.dtype(np.float32): _lib.TH_F32, np.dtype(np.float64): _lib.TH_F64}


_BLOCK_VALUES = 1 << 19      # values formatted per native call: ~13 MB of text (a 1000-row group of the 338-class matrix is one call)


def _check(matrix) -> np.ndarray:
    a = np.asarray(matrix)
    if a.ndim == 1:
        a = a.reshape(-1, 1)
    if a.ndim != 2 or a.dtype not in _DTYPES:
        raise TypeError("format_csv takes a 1-D or 2-D float16/float32/float64 array")
    return np.ascontiguousarray(a)


class TextScratch:
    """A text buffer kept between calls (a fresh 10 MB array per group costs its page faults every time), page-locked once a device
    formats into it (the text then comes back at the PCIe rate).  One writer at a time; ``close()`` unlocks it.  The memoryviews
    _blocks yields out of it are only valid until the next call — every caller writes or copies them at once."""

    def __init__(self):
        self._buf = None
        self._pinned = False
        self._pin_failed = False      # th_host_register refused THIS buffer: not tried again until it is reallocated

    def get(self, cap: int, lib, pin: bool) -> np.ndarray:
        if self._buf is None or self._buf.size < cap:
            self.close()
            self._buf = np.empty(max(cap, 1 << 20), dtype=np.uint8)                # not zero-filled
        if pin and not self._pinned and not self._pin_failed:
            self._buf.fill(0)                                                      # resident before it is locked
            self._pinned = lib.th_host_register(C.c_void_p(self._buf.ctypes.data), self._buf.nbytes) == 0
            self._pin_failed = not self._pinned                                   # (a failing hipHostRegister + a 13 MB fill per group otherwise)
        return self._buf

    def close(self) -> None:
        if self._buf is not None and self._pinned:
            _lib.load().th_host_unregister(C.c_void_p(self._buf.ctypes.data))
        self._buf, self._pinned, self._pin_failed = None, False, False

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _blocks(a: np.ndarray, device=None, scratch: "TextScratch | None" = None):
    """memoryviews of the text of consecutive row blocks; the scratch buffer is reused between blocks.  With ``device`` (a HIP
    device index) float32 blocks are formatted on that GPU (th_format_csv_device); a block holding a value without the fixed
    24-character form — or any failure of the device path — is formatted by the host threads instead: same bytes either way."""
    n, k = a.shape
    if n == 0:
        return
    if k == 0:
        yield memoryview(b"\n" * n)
        return
    rows = max(1, min(n, _BLOCK_VALUES // k))
    cap = rows * k * 28 + 32
    lib = _lib.load()
    buf = (scratch.get(cap, lib, pin=device is not None and a.dtype == np.float32) if scratch is not None
           else np.empty(cap, dtype=np.uint8))          # not zero-filled
    on_device = device is not None and a.dtype == np.float32
    for lo in range(0, n, rows):
        blk = a[lo:lo + rows]
        if on_device:
            got = lib.th_format_csv_device(int(device), blk.ctypes.data_as(C.c_void_p), blk.shape[0], k,
                                           buf.ctypes.data_as(C.c_void_p), cap)
            if got >= 0:
                yield memoryview(buf)[:got]
                continue
            if got != _lib.TH_EUNSUP:
                on_device = False           # no device / out of memory: the host threads take over for the rest of the matrix
        got = lib.th_format_csv(blk.ctypes.data_as(C.c_void_p), _DTYPES[a.dtype], blk.shape[0], k,
                                buf.ctypes.data_as(C.c_void_p), cap)
        if got < 0:
            raise _lib.TimedHipError(int(got), lib.th_last_error().decode(errors="replace"))
        yield memoryview(buf)[:got]


def format_csv(matrix: np.ndarray, device=None) -> bytes:
    """'%.18e' / ',' / '\\n' text of a 2-D float16/32/64 matrix (a 1-D array is one value per line, as np.savetxt)."""
    return b"".join(bytes(m) for m in _blocks(_check(matrix), device))


def savetxt_csv(f, matrix: np.ndarray, device=None, scratch: "TextScratch | None" = None) -> None:
    """Drop-in for ``np.savetxt(f, matrix, delimiter=",")`` on a file opened in text or binary mode.  ``device``: format float32
    matrices on that GPU (see ``_blocks``); ``scratch``: a TextScratch the caller keeps between calls."""
    for m in _blocks(_check(matrix), device, scratch):
        try:
            f.write(m)
        except TypeError:       # text-mode handle
            f.write(bytes(m).decode("ascii"))


def loadtxt_f16(path) -> np.ndarray:
    """What ``np.genfromtxt(path, delimiter=",", dtype=np.float16)`` returns for a probability CSV (reference
    predict.py:163), read with NumPy's C parser: text -> float64 -> float16 is the same double rounding."""
    try:
        return np.loadtxt(path, delimiter=",", dtype=np.float64, ndmin=2).astype(np.float16)
    except ValueError:      # ragged / missing fields: let genfromtxt apply its own rules
        return np.atleast_2d(np.genfromtxt(path, delimiter=",", dtype=np.float16))


def read_string_table(path, delimiter: str = ",") -> "np.ndarray | None":
    """The rows of a plain delimiter-separated text file as a 2-D NumPy string array — what
    ``np.atleast_2d(np.genfromtxt(path, delimiter=delimiter, dtype=str))`` returns (reference predict.py:99 reads
    datasetmap.txt that way) — parsed natively (th_csv_shape / th_csv_fill).  None when the file is anything but plain
    (comments, quotes, blank or ragged lines, non-ASCII …): the caller then lets NumPy apply its own rules."""
    with open(path, "rb") as f:
        text = f.read()
    lib = _lib.load()
    rows, cols, width = C.c_int64(), C.c_int(), C.c_int()
    d = delimiter.encode("ascii")
    if lib.th_csv_shape(text, len(text), d, C.byref(rows), C.byref(cols), C.byref(width)) != 0:
        return None
    out = np.empty((rows.value, cols.value), dtype=f"<U{width.value}")
    if lib.th_csv_fill(text, len(text), d, rows.value, cols.value, width.value, out.ctypes.data_as(C.c_void_p)) != 0:
        return None
    return out


def argmax_letters(matrix: np.ndarray, column_letters) -> np.ndarray:
    """'S1' array: the one-letter code of each row's arg-max column (np.argmax rules: first maximum, first NaN wins) —
    reference design_utils/utils.py:659,689-692 — in one native pass over the float16/32/64 matrix."""
    a = np.asarray(matrix)
    if a.ndim != 2 or a.dtype not in _DTYPES:
        raise TypeError("argmax_letters takes a 2-D float16/float32/float64 matrix")
    a = np.ascontiguousarray(a)
    letters = "".join(str(c) for c in column_letters).encode("ascii")
    if len(letters) != a.shape[1]:
        raise ValueError(f"need one letter per column ({a.shape[1]}), got {len(letters)}")
    out = np.empty(a.shape[0], dtype="S1")
    if a.shape[0]:
        _lib.check(_lib.load().th_argmax_letters(a.ctypes.data_as(C.c_void_p), _DTYPES[a.dtype], a.shape[0], a.shape[1], letters,
                                                 out.ctypes.data_as(C.c_void_p), None))
    return out
