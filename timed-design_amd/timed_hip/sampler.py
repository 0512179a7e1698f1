"""Thin host wrappers over the sampler entry points of the C ABI (th_apply_temp / th_sample_ex).

These are the numeric halves of reference design_utils/sampling_utils.py:
``apply_temp_to_probs`` (:139-161) and the inverse-CDF draw of ``random_choice_prob_index``
(:81-82).  The reference-named functions that call them live in design_utils/sampling_utils.py.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np

from . import _lib

RNG_HOST, RNG_PHILOX, RNG_MT19937 = _lib.TH_RNG_HOST, _lib.TH_RNG_PHILOX, _lib.TH_RNG_MT19937


def _as_probs(probs) -> np.ndarray:
    p = np.ascontiguousarray(np.asarray(probs, dtype=np.float64))
    if p.ndim != 2:
        raise ValueError(f"probs must be 2-D (n_residues, n_categories), got shape {p.shape}")
    return p


def apply_temperature(probs, t: float = 1.0, device: int = 0) -> np.ndarray:
    """q = p**(1/t), rows renormalised (fp64) — computed on the GPU."""
    p = _as_probs(probs)
    out = np.empty_like(p)
    if p.shape[0] == 0:
        return out
    lib = _lib.load()
    _lib.check(lib.th_sample_ex(p.ctypes.data, p.shape[0], p.shape[1], 0, float(t), RNG_PHILOX, 0, 0, None, None, None,
                                None, None, out.ctypes.data, device))
    return out


def sample_indices(probs, n_samples: int, temperature: float = 1.0, uniforms: Optional[np.ndarray] = None,
                   seed: int = 0, rng: str = "auto", rng_offset: int = 0, return_uniforms: bool = False,
                   letters: Optional[str] = None, device: int = 0):
    """Draw ``n_samples`` residue indices per row of ``probs`` in one fused launch.

    idx[s, i] = first j with cumsum_j(q[i]) > r[s, i], else 0   (reference sampling_utils.py:82)

    rng: "host" (use ``uniforms`` [n_samples, n_res]), "philox" (rocRAND on device),
    "mt19937" (device replay of np.random.seed(seed); np.random.rand), "auto" = host if uniforms given
    else philox.  Returns int32 [n_samples, n_res] (plus uniforms / letter matrix when requested).
    """
    p = _as_probs(probs)
    n_res, n_cls = p.shape
    mode = {"host": RNG_HOST, "philox": RNG_PHILOX, "mt19937": RNG_MT19937,
            "auto": RNG_HOST if uniforms is not None else RNG_PHILOX}[rng]
    u_ptr = None
    if mode == RNG_HOST:
        if uniforms is None:
            raise ValueError("rng='host' needs uniforms")
        u = np.ascontiguousarray(np.asarray(uniforms, dtype=np.float64))
        if u.shape != (n_samples, n_res):
            raise ValueError(f"uniforms must have shape {(n_samples, n_res)}, got {u.shape}")
        u_ptr = u.ctypes.data
    idx = np.empty((n_samples, n_res), dtype=np.int32)
    r_out = np.empty((n_samples, n_res), dtype=np.float64) if return_uniforms else None
    let_out = None
    cat = None
    if letters is not None:
        if len(letters) != n_cls:
            raise ValueError(f"letters must have one character per category ({n_cls}), got {len(letters)}")
        cat = letters.encode("ascii")
        let_out = np.empty((n_samples, n_res), dtype="S1")
    if n_samples and n_res:
        lib = _lib.load()
        _lib.check(lib.th_sample_ex(p.ctypes.data, n_res, n_cls, n_samples, float(temperature), mode, int(seed),
                                    int(rng_offset), u_ptr, idx.ctypes.data,
                                    r_out.ctypes.data if r_out is not None else None, cat,
                                    let_out.ctypes.data if let_out is not None else None, None, device))
    res = [idx]
    if return_uniforms:
        res.append(r_out)
    if letters is not None:
        res.append(let_out)
    return res[0] if len(res) == 1 else tuple(res)
