"""calculate_seq_metrics — the one function of the reference's analyse_utils on the sampler path
(reference design_utils/analyse_utils.py:351-371; called per drawn sequence at
sampling_utils.py:132, where it dominates wall time — SURVEY.md §8 row f-2).

    *** PARITY UNPINNED ***  The reference delegates to ampal==1.5.1
    (sequence_charge, sequence_isoelectric_point, sequence_molecular_weight,
    sequence_molar_extinction_280), which is absent from /root/reference and from this image, and no
    reference test covers it.  When ampal is importable its functions are used verbatim.  Otherwise
    the restatement below (ampal's published algorithm: Henderson-Hasselbalch partial charges at
    pH 7.4 incl. termini, pI = pH of minimum |charge| on a 0.1 grid over [1,13), average residue
    masses + one water, Trp/Tyr/Cys extinction at 280 nm) is used with tabulated constants;
    ``METRICS_SOURCE`` says which.  Vectorised over a batch of sequences via residue histograms.
"""
from __future__ import annotations

import math
import typing as t

import numpy as np

_AA = "ACDEFGHIKLMNPQRSTVWY"
_MWT = dict(A=71.0779, C=103.1429, D=115.0874, E=129.114, F=147.1739, G=57.0513, H=137.1393, I=113.1576, K=128.1723,
            L=113.1576, M=131.1961, N=114.1026, P=97.1152, Q=128.1292, R=156.1857, S=87.0773, T=101.1039, V=99.1311,
            W=186.2099, Y=163.1733)
_WATER = 18.01528
_EXT280 = dict(W=5690, Y=1280, C=120)
_CHARGE = dict(C=-1, D=-1, E=-1, H=+1, K=+1, R=+1, Y=-1)
_CHARGE_TERM = {"N-term": +1, "C-term": -1}
_PKA = dict(C=8.3, D=3.65, E=4.25, H=6.1, K=10.53, R=12.48, Y=10.1)
_PKA_TERM = {"N-term": 8.0, "C-term": 3.1}

try:  # pragma: no cover - ampal is not installed in the build image
    from ampal.analyse_protein import (sequence_charge, sequence_isoelectric_point, sequence_molar_extinction_280,
                                       sequence_molecular_weight)
    METRICS_SOURCE = "ampal"
except Exception:  # ImportError or a broken optional dependency
    METRICS_SOURCE = "restatement (unpinned)"
    sequence_charge = None


def _partial(pka: float, sign: int, ph) -> np.ndarray:
    """ampal's partial_charge: 10**d / (1 + 10**d), d = pH - pKa (negated for basic groups).  ampal evaluates it with
    Python floats, i.e. libm pow — so does this (NumPy's vectorised power may round differently) and so does the table
    the device kernel uses (csrc/sampler.hip build_metric_tables)."""
    out = []
    for p in np.atleast_1d(np.asarray(ph, dtype=float)):
        diff = float(p) - pka
        if sign > 0:
            diff = -diff
        r = math.pow(10.0, diff)
        out.append(r / (1.0 + r))
    return np.array(out)


def _charge_table(ph) -> t.Tuple[np.ndarray, np.ndarray]:
    """signed partial charge of one residue of each class (alphabetical order) and of the two termini at each pH"""
    ph = np.atleast_1d(np.asarray(ph, dtype=float))
    per_res = np.zeros((20, ph.size))
    for aa, sign in _CHARGE.items():
        per_res[_AA.index(aa)] = _partial(_PKA[aa], sign, ph) * sign
    term = _partial(_PKA_TERM["N-term"], +1, ph) * (+1) + _partial(_PKA_TERM["C-term"], -1, ph) * (-1)
    return per_res, term


def _dot_in_class_order(counts: np.ndarray, table: np.ndarray) -> np.ndarray:
    """sum_c counts[:, c] * table[c] accumulated strictly in class order (c = 0..19) — the order the device kernel
    (csrc/sampler.hip k_seq_metrics) uses, so that the two agree bit for bit (a BLAS dot would not)."""
    acc = np.zeros((counts.shape[0],) + table.shape[1:])
    for c in range(20):
        acc = acc + counts[:, c].reshape((-1,) + (1,) * (table.ndim - 1)) * table[c]
    return acc


def _charge_from_counts(counts: np.ndarray, ph) -> np.ndarray:
    """counts [n_seq, 20] -> net charge [n_seq, len(ph)]"""
    per_res, term = _charge_table(ph)
    return _dot_in_class_order(counts, per_res) + term[None, :]


def residue_counts(seqs: t.Sequence[str]) -> np.ndarray:
    lut = np.full(256, -1, dtype=np.int64)
    for i, a in enumerate(_AA):
        lut[ord(a)] = i
    counts = np.zeros((len(seqs), 20))
    for k, s in enumerate(seqs):
        idx = lut[np.frombuffer(s.encode("ascii"), dtype=np.uint8)]
        counts[k] = np.bincount(idx[idx >= 0], minlength=20)
    return counts


def seq_metrics_batch(seqs: t.Sequence[str]) -> np.ndarray:
    """[n_seq, 4] = (charge at pH 7.4, isoelectric point, molecular weight, molar extinction at 280)."""
    if sequence_charge is not None:  # pragma: no cover
        return np.array([[sequence_charge(s), sequence_isoelectric_point(s), sequence_molecular_weight(s),
                          sequence_molar_extinction_280(s)] for s in seqs], dtype=float)
    counts = residue_counts(seqs)
    grid = np.arange(1, 13, 0.1)                # NumPy fills first + i*((1 + 0.1) - 1): 120 points
    charge = _charge_from_counts(counts, [7.4])[:, 0]
    pi = grid[np.abs(_charge_from_counts(counts, grid)).argmin(axis=1)]
    mw = _dot_in_class_order(counts, np.array([_MWT[a] for a in _AA])) + _WATER
    ext = _dot_in_class_order(counts, np.array([float(_EXT280.get(a, 0)) for a in _AA]))
    return np.stack([charge, pi, mw, ext], axis=1)


def calculate_seq_metrics(seq: str) -> t.Tuple[float, float, float, float]:
    """reference analyse_utils.py:351-371 -> (charge, iso_ph, mw, me)."""
    c, p, m, e = seq_metrics_batch([seq])[0]
    return float(c), float(p), float(m), int(e) if float(e).is_integer() else float(e)
