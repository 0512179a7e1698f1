"""design_utils.sampling_utils — Monte-Carlo sequence sampling on the GPU.

Same five functions and signatures as the reference module (reference
design_utils/sampling_utils.py); the two numeric kernels run in libtimedhip.so:

    apply_temp_to_probs        :139-161  -> th_apply_temp   (fp64, NumPy-order normaliser)
    random_choice_prob_index   :53-90    -> th_sample_ex    (fused cumsum / compare / first-true)

Randomness: like the reference, the uniforms come from NumPy's GLOBAL legacy generator
(``np.random.rand``, reference :81), drawn in the reference's order, so after ``np.random.seed(s)``
every function here returns what the reference returns — bit-exact residue indices.  (The
reference never seeds it — its ``--seed`` is inert; our sample.py does.)
``sample_with_multiprocessing`` draws every sequence of every PDB key in ONE launch sequence on a resident
sampler (th_sampler_load / th_sampler_draw), including the per-sequence metrics of calculate_seq_metrics.
"""
from __future__ import annotations

import ctypes as C
import json
import threading
import typing as t

import numpy as np

from timed_hip import sampler as _sampler

from .amino_acids import standard_amino_acids
from .analyse_utils import METRICS_SOURCE, seq_metrics_batch

_LETTERS20 = "".join(standard_amino_acids.keys())
_CUM_DTYPES = (np.dtype(np.float16), np.dtype(np.float32), np.dtype(np.float64))


# ---- writers ------------------------------------------------------------------------------------------------
# Output formats of reference sampling_utils.py:12-50, as data: (suffix, selected-by, row writer).
def _emit_json(fh, pdb_to_sampled):
    json.dump(pdb_to_sampled, fh)


def _emit_fasta(fh, pdb_to_sampled):
    fh.writelines(f">{pdb}_{n}\n{rec[0]}\n" for pdb, recs in pdb_to_sampled.items() for n, rec in enumerate(recs))


_METRIC_COLUMNS = ("pdb", "sequence", "charge", "isoelectric_point", "molecular_weight", "molar_extinction")


def _emit_metrics(fh, pdb_to_sampled):
    fh.write(",".join(_METRIC_COLUMNS) + "\n")
    fh.writelines(",".join(str(v) for v in (pdb, *rec[:5])) + "\n" for pdb, recs in pdb_to_sampled.items() for rec in recs)


# a sequence file is written unless the mode names the OTHER sequence format ("all", or anything else, selects both)
_SEQUENCE_OUTPUTS = ((".json", "fasta", _emit_json), (".fasta", "json", _emit_fasta))


def save_as(pdb_to_sampled: dict, filename: str, mode: str):
    """reference sampling_utils.py:12-50: ``<filename>.json`` unless mode == "fasta", ``<filename>.fasta`` unless
    mode == "json", and always ``<filename>_metrics.csv``; returns the paths in that order."""
    jobs = [(filename + suffix, emit) for suffix, skipped_by, emit in _SEQUENCE_OUTPUTS if mode != skipped_by]
    jobs.append((filename + "_metrics.csv", _emit_metrics))
    print(f"Writing {len(jobs)} output file(s) for {len(pdb_to_sampled)} structure(s) (mode={mode})")
    for path, emit in jobs:
        with open(path, "w") as fh:
            emit(fh, pdb_to_sampled)
    return [path for path, _ in jobs]


# ---- sampling -----------------------------------------------------------------------------------------------
def _category_letters(rotamer_categories, n_cls: int) -> np.ndarray:
    if rotamer_categories is not None and len(rotamer_categories):
        res = np.array(rotamer_categories)
    else:
        res = np.array(list(standard_amino_acids.keys()))
    if len(res) < n_cls:
        raise ValueError(f"{len(res)} category names for {n_cls} probability columns")
    return res


def _cum_dtype(a: np.ndarray):
    """np.cumsum accumulates in the array's dtype (reference :82): float16/float32 rows keep their running sum in
    that type on the GPU too; everything else is float64 like NumPy's promotion of ints/bools would make it."""
    return a.dtype if a.dtype in _CUM_DTYPES else np.dtype(np.float64)


def random_choice_prob_index(
    probs: np.ndarray,
    axis: int = 1,
    return_seq: bool = True,
    rotamer_categories: t.Optional[np.ndarray] = None,
) -> np.ndarray:
    """reference sampling_utils.py:53-90: one uniform per residue from np.random.rand, inverse CDF
    (first index whose running sum exceeds r, 0 if none) — computed on the GPU."""
    probs = np.asarray(probs)
    p = probs if axis == 1 else probs.T
    r = np.random.rand(p.shape[0])
    idxs = _sampler.sample_indices(p, 1, uniforms=r[None, :], cum_dtype=_cum_dtype(p))[0].astype(np.int64)
    if return_seq:
        return _category_letters(rotamer_categories, p.shape[1])[idxs]
    return idxs


_RAND_LOCK = threading.Lock()


def _legacy_rand(n: int) -> np.ndarray:
    """``np.random.rand(n)`` — the reference's uniforms (:81), from NumPy's GLOBAL legacy generator — with the same values AND the
    same generator state afterwards, produced by th_mt19937_rand: the state (MT19937 key + position) is taken out with
    np.random.get_state(), advanced natively (vectorised block regeneration, bulk tempering) and put back.  NumPy's own path costs
    0.70 ms for the 300 000 uniforms of a config-5 call, two thirds of the call.  Short draws, a generator that is not MT19937 and
    any native failure go through np.random.rand itself.  Like the reference's own use of the global generator this is for one
    drawing thread at a time: calls through this function are serialised, but a thread that calls np.random directly between the
    get_state and the set_state here would see its draws replayed."""
    n = int(n)
    if n < 4096:
        return np.random.rand(n)
    with _RAND_LOCK:
        st = np.random.get_state()
        if st[0] != "MT19937":
            return np.random.rand(n)
        from timed_hip import _lib
        key = np.ascontiguousarray(st[1], dtype=np.uint32).copy()
        pos = C.c_int(int(st[2]))
        out = np.empty(n, dtype=np.float64)
        if _lib.load().th_mt19937_rand(key.ctypes.data_as(C.c_void_p), C.byref(pos), n, out.ctypes.data_as(C.c_void_p)) != 0:
            return np.random.rand(n)
        np.random.set_state((st[0], key, int(pos.value), st[3], st[4]))
        return out


def _legacy_words(n: int, out=None):
    """The RAW state words behind ``np.random.rand(n)`` (two per double), the global generator advanced exactly as rand(n) would
    have advanced it: th_mt19937_words walks the recurrence on the host — it is sequential — and leaves tempering and the 53-bit
    conversion (two thirds of _legacy_rand's time) to the draw kernel (TH_RNG_MT_WORDS).  None when the generator is not MT19937
    or the native call fails: the caller falls back to _legacy_rand."""
    n = int(n)
    with _RAND_LOCK:
        st = np.random.get_state()
        if st[0] != "MT19937":
            return None
        from timed_hip import _lib
        key = np.ascontiguousarray(st[1], dtype=np.uint32).copy()
        pos = C.c_int(int(st[2]))
        if out is None:
            out = np.empty(2 * n, dtype=np.uint32)
        if _lib.load().th_mt19937_words(key.ctypes.data_as(C.c_void_p), C.byref(pos), n, out.ctypes.data_as(C.c_void_p)) != 0:
            return None
        np.random.set_state((st[0], key, int(pos.value), st[3], st[4]))
        return out


RNG_CHOICES = ("numpy", "philox", "mt19937")

# Draws already taken from the device generators in this process, per (rng, seed): like the global NumPy stream, which advances
# between calls, two sample_with_multiprocessing calls with the same --rng / --seed continue ONE stream instead of returning the
# same draws twice.  reset_device_rng() starts the streams over (a fresh process starts at 0: a run is reproducible).
_DEVICE_RNG_DRAWS: dict = {}


def reset_device_rng(rng: t.Optional[str] = None, seed: t.Optional[int] = None) -> None:
    for k in [k for k in _DEVICE_RNG_DRAWS if (rng is None or k[0] == rng) and (seed is None or k[1] == int(seed))]:
        del _DEVICE_RNG_DRAWS[k]


def _sample_keys(keys, sample_n: int, pdb_to_probability: dict, rotamer_categories, device: int = 0, rng: str = "numpy",
                 seed: int = 0) -> dict:
    """All ``keys`` in ONE submission (th_sampler_run): rows of every key, offsets and letters uploaded together, the running
    sums, every draw over (key, sample, residue), the letters and the per-sequence metrics computed by two kernels, one
    page-locked block copied back.  ``rng="numpy"`` (the default): uniforms from the global legacy generator in the reference's
    order — for key: for sample: rand(n_res) — which one rand() call of the total length reproduces exactly (the stream is
    consumed value by value).  ``"philox"`` (rocRAND Philox4x32-10, one subsequence per draw) and ``"mt19937"`` (MT19937 seeded
    with ``seed`` like np.random.seed, generated ON the device) draw their uniforms on the GPU: nothing is generated or uploaded
    by the host, the result depends on ``seed`` only and is NOT the reference's stream."""
    if rng not in RNG_CHOICES:
        raise ValueError(f"rng must be one of {RNG_CHOICES}, got {rng!r}")
    keys = list(keys)
    if not keys:
        return {}
    # what the reference builds per sample (:125); a LazyProbabilities mapping hands the rows out as an array directly
    as_matrix = getattr(pdb_to_probability, "matrix", None)
    mats = [np.array(as_matrix(k) if as_matrix else pdb_to_probability[k]) for k in keys]
    for k, m in zip(keys, mats):
        if m.ndim != 2 or m.shape[0] == 0:
            raise ValueError(f"{k}: expected a non-empty (n_residues, n_categories) probability matrix, got shape {m.shape}")
    n_cls = mats[0].shape[1]
    if any(m.shape[1] != n_cls for m in mats):
        raise ValueError("all keys must have the same number of categories")
    cum = _cum_dtype(mats[0]) if all(m.dtype == mats[0].dtype for m in mats) else np.dtype(np.float64)
    row_off = np.concatenate([[0], np.cumsum([m.shape[0] for m in mats])]).astype(np.int64)
    cats = _category_letters(rotamer_categories, n_cls)
    one_letter = all(len(c) == 1 for c in cats[:n_cls])
    sm = _sampler.default_sampler(device)
    on_device = one_letter and METRICS_SOURCE != "ampal"
    out = {}
    with sm.lock:       # the process-wide sampler is shared between threads; the arrays of run() are views of ITS result block
        r, mode, offset = None, rng, 0
        if rng != "numpy":
            offset = _DEVICE_RNG_DRAWS.get((rng, int(seed)), 0)
            _DEVICE_RNG_DRAWS[(rng, int(seed))] = offset + int(sample_n) * int(row_off[-1])
        if rng == "numpy":
            # the recurrence of the global generator is walked on the host straight into the sampler's page-locked buffer; the kernel
            # tempers the words and forms the doubles (short draws and other generators: np.random.rand's own doubles)
            n_draws = int(sample_n) * int(row_off[-1])
            r = _legacy_words(n_draws, out=sm.uniform_buffer(2 * n_draws, np.uint32)) if n_draws >= 4096 else None
            mode = "mt_words"
            if r is None:
                r, mode = _legacy_rand(n_draws), "host"
        d = sm.run(np.concatenate(mats).astype(np.float64), row_off, sample_n, uniforms=r, rng=mode, seed=seed, rng_offset=offset,
                   letters="".join(cats[:n_cls]) if one_letter else None, want_idx=not one_letter, want_letters=one_letter,
                   want_metrics=on_device, cum_dtype=cum)
        out = _collect(keys, d, row_off, sample_n, cats, one_letter, on_device)
    return out


def _collect(keys, d, row_off, sample_n, cats, one_letter, on_device) -> dict:
    out = {}
    for k, key in enumerate(keys):
        lo, hi = int(row_off[k]), int(row_off[k + 1])
        if one_letter:
            n_res = hi - lo            # one decode for the key's sample_n sequences, then slices (a bytes object per row costs 3x as much)
            text = d["letters"][sample_n * lo: sample_n * hi].tobytes().decode("ascii")
            seqs = [text[i * n_res:(i + 1) * n_res] for i in range(sample_n)]
        else:   # multi-character category names (full rotamer labels): join on the host like the reference
            block = d["idx"][sample_n * lo: sample_n * hi].reshape(sample_n, hi - lo)
            seqs = ["".join(cats[row]) for row in block]
        if on_device:
            met = d["metrics"][k * sample_n: (k + 1) * sample_n]
        else:
            met = seq_metrics_batch(seqs) if seqs else np.empty((0, 4))
        out[key] = _result_tuples(seqs, met)
    return out


def _result_tuples(seqs, met) -> list:
    """[(sequence, charge, pI, MW, eps280), ...] as the reference returns them (:131-135): Python floats, the extinction
    coefficient an int when it is integral (ampal sums integer constants; half a disulphide bridge makes it x.5).  Built column-wise —
    tolist() per column and one zip — instead of converting 5 000 NumPy scalars one by one (a third of an API call at config 5)."""
    met = np.asarray(met, dtype=np.float64).reshape(-1, 4)
    c, p, w, e = met.T.tolist()
    e_col = met[:, 3]
    if bool(np.all(np.isfinite(e_col) & (np.floor(e_col) == e_col) & (np.abs(e_col) < 2.0 ** 53))):
        e = e_col.astype(np.int64).tolist()
    else:
        e = [int(x) if x.is_integer() else x for x in e]
    return list(zip(seqs, c, p, w, e))


def sample_from_sequences(
    pdb: str,
    sample_n: int,
    pdb_to_probability: dict,
    rotamer_categories: t.Optional[np.ndarray],
) -> dict:
    """reference sampling_utils.py:93-136 -> {pdb: [(seq, charge, pI, MW, eps280), ...]}."""
    return _sample_keys([pdb], sample_n, pdb_to_probability, rotamer_categories)


def apply_temp_to_probs(probs: np.ndarray, t: float = 1.0):
    """reference sampling_utils.py:139-161: probs**(1/t), rows renormalised, fp64.  The power is NumPy's own ``**``
    for generic exponents (bit-identical to the reference on any host), square/sqrt/copy on the GPU for t in
    {0.5, 2, 1}; the pairwise-order normaliser runs on the GPU."""
    return _sampler.apply_temperature(np.array(probs, dtype=np.float64), t)


def sample_with_multiprocessing(workers, pdb_codes, sample_n, pdb_to_probability, flat_categories, rng: str = "numpy", seed: int = 0):
    """reference sampling_utils.py:164-197.  The reference fans PDB keys over a process Pool (every forked worker
    inherits the SAME generator state, so different PDBs can receive identical uniform streams — Appendix C-1).  Here
    all keys are drawn together on the GPU from one continuous stream in key order; ``workers`` is accepted and
    ignored.  Opt-in ``rng`` / ``seed`` (not in the reference): "numpy" replays np.random.rand bit for bit (default);
    "philox" / "mt19937" draw the uniforms on the device from ``seed`` (see _sample_keys); successive calls in one process
    continue the (rng, seed) stream where the previous call stopped, as the global NumPy stream does (reset_device_rng)."""
    return _sample_keys(pdb_codes, sample_n, pdb_to_probability, flat_categories, rng=rng, seed=seed)
