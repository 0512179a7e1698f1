"""Thin host wrappers over the sampler entry points of the C ABI (th_apply_temp / th_sample_ex).

These are the numeric halves of reference design_utils/sampling_utils.py:
``apply_temp_to_probs`` (:139-161) and the inverse-CDF draw of ``random_choice_prob_index``
(:81-82).  The reference-named functions that call them live in design_utils/sampling_utils.py.
"""
from __future__ import annotations

import ctypes as C
import threading
from typing import Optional, Sequence

import numpy as np

from . import _lib

RNG_HOST, RNG_PHILOX, RNG_MT19937, RNG_MT_WORDS = _lib.TH_RNG_HOST, _lib.TH_RNG_PHILOX, _lib.TH_RNG_MT19937, _lib.TH_RNG_MT_WORDS


def _as_probs(probs) -> np.ndarray:
    p = np.ascontiguousarray(np.asarray(probs, dtype=np.float64))
    if p.ndim != 2:
        raise ValueError(f"probs must be 2-D (n_residues, n_categories), got shape {p.shape}")
    return p


_EXACT_EXPONENTS = (1.0, 2.0, 0.5)     # NumPy's `**` fast paths (copy, square, sqrt): exact IEEE operations on the device too
_CUM = {np.dtype(np.float64): _lib.TH_F64, np.dtype(np.float32): _lib.TH_F32, np.dtype(np.float16): _lib.TH_F16}


def _power_args(p: np.ndarray, t: float):
    """(rows, temper_mode) for q = p**(1/t).  Exact exponents are raised on the device; for any other exponent the
    rows are raised HERE with NumPy's own ``**`` — the very ufunc the reference executes (sampling_utils.py:159), so q
    is bit-identical to the reference on whatever host this runs on (NumPy's float64 power is libm pow or an AVX-512
    SVML routine depending on the CPU, and the two differ in the last bit)."""
    if t == 0:
        raise ZeroDivisionError("temperature 0 (the reference divides by it: sampling_utils.py:159)")
    e = 1 / t
    if e in _EXACT_EXPONENTS:
        return p, _lib.TH_TEMPER_POW
    with np.errstate(all="ignore"):
        return np.ascontiguousarray(p ** e), _lib.TH_TEMPER_PREPOWERED


class Sampler:
    """Resident sampler (th_sampler_*): load the probability rows of every key once, then draw for all keys in one
    launch sequence.  ``load`` may be called again with new rows."""

    def __init__(self, device: int = 0):
        self._lib = _lib.load()
        self._h = C.c_void_p()
        self.device = device
        self.n_rows = self.n_cls = 0
        # `load` then `draw` is ONE logical operation on a resident sampler: callers that share a Sampler between
        # threads (the process-wide default_sampler) hold this re-entrant lock across the pair; the C mutex only
        # protects each call on its own
        self.lock = threading.RLock()
        _lib.check(self._lib.th_sampler_create(device, C.byref(self._h)))

    def load(self, probs, temperature: float = 1.0, apply_temperature: Optional[bool] = None, cum_dtype=np.float64,
             return_q: bool = False) -> Optional[np.ndarray]:
        """probs [n_rows, n_cls].  ``apply_temperature`` None: temper iff temperature != 1 (what sample.py:40 does)."""
        p = _as_probs(probs)
        apply = (temperature != 1) if apply_temperature is None else apply_temperature
        rows, mode = _power_args(p, float(temperature)) if apply else (p, _lib.TH_TEMPER_NONE)
        q = np.empty_like(p) if return_q else None
        if p.shape[0]:
            _lib.check(self._lib.th_sampler_load(self._h, rows.ctypes.data, p.shape[0], p.shape[1], float(temperature), mode,
                                                 _CUM[np.dtype(cum_dtype)], q.ctypes.data if q is not None else None))
        self.n_rows, self.n_cls = p.shape
        return q

    def draw(self, row_off: Sequence[int], n_samples: int, uniforms: Optional[np.ndarray] = None, rng: str = "auto",
             seed: int = 0, rng_offset: int = 0, letters: Optional[str] = None, want_idx: bool = True,
             want_uniforms: bool = False, want_metrics: bool = False) -> dict:
        """Draw ``n_samples`` sequences for every key; key k owns rows [row_off[k], row_off[k+1]).  Results are flat
        arrays in the reference's draw order (for key: for sample: for residue); ``split`` cuts them per key."""
        off = np.ascontiguousarray(np.asarray(row_off, dtype=np.int64))
        n_keys = off.size - 1
        total = int(n_samples) * int(off[-1] - off[0]) if n_keys > 0 else 0
        mode = {"host": RNG_HOST, "philox": RNG_PHILOX, "mt19937": RNG_MT19937,
                "auto": RNG_HOST if uniforms is not None else RNG_PHILOX}[rng]
        u_ptr = None
        if mode == RNG_HOST:
            if uniforms is None:
                raise ValueError("rng='host' needs uniforms")
            u = np.ascontiguousarray(np.asarray(uniforms, dtype=np.float64)).ravel()
            if u.size != total:
                raise ValueError(f"uniforms must hold {total} values, got {u.size}")
            u_ptr = u.ctypes.data
        out = {}
        idx = np.empty(total, dtype=np.int32) if want_idx else None
        r_out = np.empty(total, dtype=np.float64) if want_uniforms else None
        cat = let = met = None
        if letters is not None:
            if len(letters) != self.n_cls:
                raise ValueError(f"letters must have one character per category ({self.n_cls}), got {len(letters)}")
            cat = letters.encode("ascii")
            let = np.empty(total, dtype="S1")
            if want_metrics:
                met = np.empty((n_keys * int(n_samples), 4), dtype=np.float64)
        elif want_metrics:
            raise ValueError("metrics need letters")
        if total:
            _lib.check(self._lib.th_sampler_draw(
                self._h, n_keys, off.ctypes.data_as(C.POINTER(C.c_int64)), int(n_samples), mode, int(seed), int(rng_offset), u_ptr,
                cat, idx.ctypes.data if idx is not None else None, r_out.ctypes.data if r_out is not None else None,
                let.ctypes.data if let is not None else None, met.ctypes.data if met is not None else None))
        out.update(idx=idx, uniforms=r_out, letters=let, metrics=met, row_off=off, n_samples=int(n_samples))
        return out

    def uniform_buffer(self, count: int, dtype) -> np.ndarray:
        """`count` elements of page-locked memory owned by the sampler (th_sampler_uniform_buffer): uniforms / raw generator words
        written here are uploaded by direct DMA.  Valid until the next call of this method."""
        nbytes = int(count) * np.dtype(dtype).itemsize
        ptr = C.c_void_p()
        _lib.check(self._lib.th_sampler_uniform_buffer(self._h, max(nbytes, 16), C.byref(ptr)))
        return np.frombuffer((C.c_char * nbytes).from_address(ptr.value), dtype=dtype, count=int(count))

    def run(self, probs, row_off: Sequence[int], n_samples: int, uniforms: Optional[np.ndarray] = None, rng: str = "auto", seed: int = 0,
            rng_offset: int = 0, letters: Optional[str] = None, want_idx: bool = True, want_letters: bool = True,
            want_metrics: bool = False, cum_dtype=np.float64) -> dict:
        """load + draw as ONE submission (th_sampler_run): the rows of every key (used as they are: temper first), their running
        sums, all draws, letters and metrics between one upload and one download.  The arrays returned are VIEWS of a page-locked
        block owned by the sampler: valid until its next call (callers hold ``lock`` while they consume them)."""
        p = _as_probs(probs)
        off = np.ascontiguousarray(np.asarray(row_off, dtype=np.int64))
        n_keys = off.size - 1
        total = int(n_samples) * p.shape[0]
        mode = {"host": RNG_HOST, "philox": RNG_PHILOX, "mt19937": RNG_MT19937, "mt_words": RNG_MT_WORDS,
                "auto": RNG_HOST if uniforms is not None else RNG_PHILOX}[rng]
        u_ptr = None
        if mode == RNG_HOST:
            if uniforms is None:
                raise ValueError("rng='host' needs uniforms")
            u = np.ascontiguousarray(np.asarray(uniforms, dtype=np.float64)).ravel()
            if u.size != total:
                raise ValueError(f"uniforms must hold {total} values, got {u.size}")
            u_ptr = u.ctypes.data
        elif mode == RNG_MT_WORDS:      # raw MT19937 state words (th_mt19937_words), two per draw: tempered and converted by the kernel
            u = np.ascontiguousarray(np.asarray(uniforms, dtype=np.uint32)).ravel()
            if u.size != 2 * total:
                raise ValueError(f"rng='mt_words' needs {2 * total} state words, got {u.size}")
            u_ptr = u.ctypes.data
        cat = None
        if letters is not None:
            if len(letters) != p.shape[1]:
                raise ValueError(f"letters must have one character per category ({p.shape[1]}), got {len(letters)}")
            cat = letters.encode("ascii")
        elif want_letters or want_metrics:
            raise ValueError("letters / metrics need the category letters")
        want = (1 if want_idx else 0) | (2 if want_letters else 0) | (4 if want_metrics else 0)
        block = C.c_void_p()
        offs = (C.c_int64 * 3)()
        _lib.check(self._lib.th_sampler_run(self._h, p.ctypes.data, p.shape[0], p.shape[1], _CUM[np.dtype(cum_dtype)], n_keys,
                                            off.ctypes.data_as(C.POINTER(C.c_int64)), int(n_samples), mode, int(seed), int(rng_offset), u_ptr,
                                            cat, want, C.byref(block), offs))
        def view(o, dtype, count):
            if o < 0:
                return None
            nbytes = count * np.dtype(dtype).itemsize
            return np.frombuffer((C.c_char * nbytes).from_address(block.value + o), dtype=dtype, count=count)
        return dict(idx=view(offs[0], np.int32, total), letters=view(offs[1], "S1", total),
                    metrics=(lambda m: None if m is None else m.reshape(-1, 4))(view(offs[2], np.float64, n_keys * int(n_samples) * 4)),
                    uniforms=None, row_off=off, n_samples=int(n_samples))

    @staticmethod
    def split(flat: np.ndarray, row_off: np.ndarray, n_samples: int):
        """per-key [n_samples, n_res_k] views of a flat draw-ordered array"""
        res, base = [], int(row_off[0])
        for k in range(len(row_off) - 1):
            lo, hi = int(row_off[k]) - base, int(row_off[k + 1]) - base
            res.append(flat[n_samples * lo: n_samples * hi].reshape(n_samples, hi - lo))
        return res

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._lib.th_sampler_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_DEFAULT: dict = {}
_DEFAULT_LOCK = threading.Lock()


def default_sampler(device: int = 0) -> Sampler:
    """Process-wide sampler of a device.  Shared between threads: take ``sampler.lock`` around a load + draw pair."""
    with _DEFAULT_LOCK:
        if device not in _DEFAULT:
            _DEFAULT[device] = Sampler(device)
        return _DEFAULT[device]


def apply_temperature(probs, t: float = 1.0, device: int = 0) -> np.ndarray:
    """q = p**(1/t), rows renormalised (fp64): exact exponents entirely on the GPU; generic exponents are raised
    with NumPy's ``**`` first (see _power_args), the pairwise-order normaliser runs on the GPU."""
    p = _as_probs(probs)
    if p.shape[0] == 0:
        return np.empty_like(p)
    sm = default_sampler(device)
    with sm.lock:
        return sm.load(p, t, apply_temperature=True, return_q=True)


def sample_indices(probs, n_samples: int, temperature: float = 1.0, uniforms: Optional[np.ndarray] = None,
                   seed: int = 0, rng: str = "auto", rng_offset: int = 0, return_uniforms: bool = False,
                   letters: Optional[str] = None, device: int = 0, cum_dtype=np.float64):
    """Draw ``n_samples`` residue indices per row of ``probs`` in one fused launch.

    idx[s, i] = first j with cumsum_j(q[i]) > r[s, i], else 0   (reference sampling_utils.py:82)

    rng: "host" (use ``uniforms`` [n_samples, n_res]), "philox" (rocRAND on device),
    "mt19937" (device replay of np.random.seed(seed); np.random.rand), "auto" = host if uniforms given
    else philox.  Returns int32 [n_samples, n_res] (plus uniforms / letter matrix when requested).
    """
    p = _as_probs(probs)
    n_res, n_cls = p.shape
    if uniforms is not None and np.shape(uniforms) != (n_samples, n_res):
        raise ValueError(f"uniforms must have shape {(n_samples, n_res)}, got {np.shape(uniforms)}")
    if letters is not None and len(letters) != n_cls:
        raise ValueError(f"letters must have one character per category ({n_cls}), got {len(letters)}")
    if not (n_samples and n_res):
        res = [np.empty((n_samples, n_res), np.int32)]
        if return_uniforms:
            res.append(np.empty((n_samples, n_res)))
        if letters is not None:
            res.append(np.empty((n_samples, n_res), "S1"))
        return res[0] if len(res) == 1 else tuple(res)
    sm = default_sampler(device)
    with sm.lock:       # another thread's load must not slip in between this load and its draw
        sm.load(p, temperature, cum_dtype=cum_dtype)
        d = sm.draw([0, n_res], n_samples, uniforms=uniforms, rng=rng, seed=seed, rng_offset=rng_offset, letters=letters,
                    want_uniforms=return_uniforms)
    res = [d["idx"].reshape(n_samples, n_res)]
    if return_uniforms:
        res.append(d["uniforms"].reshape(n_samples, n_res))
    if letters is not None:
        res.append(d["letters"].reshape(n_samples, n_res))
    return res[0] if len(res) == 1 else tuple(res)
