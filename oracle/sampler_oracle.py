"""CPU ORACLE (test infrastructure only) — NumPy restatement of the reference Monte-Carlo sampler.

Pinned: tests/golden/sampler_golden.npz was produced by IMPORTING the reference's own
design_utils/sampling_utils.py in the build container (tests/golden/make_sampler_golden.py) and
tests/test_oracle_sampler.py checks every function here against those vectors.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

  apply_temp            reference design_utils/sampling_utils.py:159-161
  choice_indices        reference design_utils/sampling_utils.py:81-82 with r made explicit
  legacy_uniforms       the stream np.random.rand draws from after np.random.seed(seed)
                        (sampling_utils.py:81 uses the global legacy MT19937 RandomState)
  philox_uniforms       restatement of rocRAND's Philox4x32-10 device generator as used by
                        csrc/sampler.hip: rocrand_init(seed, subsequence=d, offset=0) then
                        rocrand_uniform_double (rocrand_philox4x32_10.h, rocrand_uniform.h:102-109)
"""
from __future__ import annotations

import numpy as np


def apply_temp(probs: np.ndarray, t: float = 1.0) -> np.ndarray:
    probs = np.array(probs) ** (1 / t)
    p_sum = np.sum(probs, axis=1)
    return probs / p_sum[:, None]


def choice_indices(probs: np.ndarray, r: np.ndarray) -> np.ndarray:
    """probs [n_res, n_cls]; r [n_res] or [n_samples, n_res] uniforms -> first index with cumsum > r (0 if none)."""
    probs = np.asarray(probs, dtype=np.float64)
    c = probs.cumsum(axis=1)
    r = np.asarray(r, dtype=np.float64)
    if r.ndim == 1:
        return (c > r[:, None]).argmax(axis=1)
    return (c[None, :, :] > r[:, :, None]).argmax(axis=2)


def legacy_uniforms(seed: int, n: int, skip: int = 0) -> np.ndarray:
    """What ``np.random.seed(seed); np.random.rand(...)`` yields, as one flat stream."""
    rs = np.random.RandomState(seed)
    if skip:
        rs.random_sample(skip)
    return rs.random_sample(n)


# ---- Philox4x32-10 exactly as rocRAND's device engine ------------------------------------------------
_M0, _M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
_W0, _W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)
_MASK = np.uint64(0xFFFFFFFF)


def _philox4x32_10(c0, c1, c2, c3, k0, k1):
    c0, c1, c2, c3 = (np.asarray(x, dtype=np.uint32) for x in (c0, c1, c2, c3))
    k0 = np.uint32(k0); k1 = np.uint32(k1)
    with np.errstate(over="ignore"):
        for _ in range(10):
            m0 = _M0 * c0.astype(np.uint64)
            m1 = _M1 * c2.astype(np.uint64)
            hi0, lo0 = (m0 >> np.uint64(32)).astype(np.uint32), (m0 & _MASK).astype(np.uint32)
            hi1, lo1 = (m1 >> np.uint64(32)).astype(np.uint32), (m1 & _MASK).astype(np.uint32)
            c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
            k0 = np.uint32((int(k0) + int(_W0)) & 0xFFFFFFFF)
            k1 = np.uint32((int(k1) + int(_W1)) & 0xFFFFFFFF)
    return c0, c1, c2, c3


def philox_uniforms(seed: int, n: int, offset: int = 0) -> np.ndarray:
    """r[d] for draw d = offset..offset+n-1: counter = (0, 0, d_lo, d_hi), key = (seed_lo, seed_hi);
    the first two 32-bit outputs x, y give 2^-53 + (x | (y>>11)<<32) * 2^-53  (range (0, 1])."""
    d = np.arange(offset, offset + n, dtype=np.uint64)
    zeros = np.zeros(n, dtype=np.uint32)
    x, y, _, _ = _philox4x32_10(zeros, zeros, (d & _MASK).astype(np.uint32), (d >> np.uint64(32)).astype(np.uint32),
                                seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    v = x.astype(np.uint64) | ((y >> np.uint32(11)).astype(np.uint64) << np.uint64(32))
    two53 = 1.1102230246251565e-16
    return two53 + v.astype(np.float64) * two53
