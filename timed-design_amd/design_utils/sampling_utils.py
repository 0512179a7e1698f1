"""design_utils.sampling_utils — Monte-Carlo sequence sampling on the GPU.

Same five functions and signatures as the reference module (reference
design_utils/sampling_utils.py); the two numeric kernels run in libtimedhip.so:

    apply_temp_to_probs        :139-161  -> th_apply_temp   (fp64, NumPy-order normaliser)
    random_choice_prob_index   :53-90    -> th_sample_ex    (fused cumsum / compare / first-true)

Randomness: like the reference, the uniforms come from NumPy's GLOBAL legacy generator
(``np.random.rand``, reference :81), drawn in the reference's order, so after ``np.random.seed(s)``
every function here returns what the reference returns — bit-exact residue indices.  (The
reference never seeds it — its ``--seed`` is inert; our sample.py does.)
``sample_from_sequences`` draws all ``sample_n`` sequences of a PDB in ONE launch
([sample_n, n_res] uniforms — the same stream the reference's per-sample loop consumes).
"""
from __future__ import annotations

import json
import typing as t

import numpy as np

from timed_hip import sampler as _sampler

from .amino_acids import standard_amino_acids
from .analyse_utils import seq_metrics_batch

_LETTERS20 = "".join(standard_amino_acids.keys())


def save_as(pdb_to_sampled: dict, filename: str, mode: str):
    """reference sampling_utils.py:12-50: .json (unless mode=='fasta'), .fasta (unless 'json'), and
    always ``_metrics.csv``.  Returns the written paths in that order."""
    output_paths = []
    print(f"Saving sampled sequences in mode {mode}")
    if mode != "fasta":
        outfile_path = f"{filename}.json"
        output_paths.append(outfile_path)
        with open(outfile_path, "w") as outfile:
            json.dump(pdb_to_sampled, outfile)
    if mode != "json":
        outfile_path = f"{filename}.fasta"
        output_paths.append(outfile_path)
        with open(outfile_path, "w") as outfile:
            for pdb, seq_list in pdb_to_sampled.items():
                for i, seq in enumerate(seq_list):
                    outfile.write(f">{pdb}_{i}\n{seq[0]}\n")
    print("Saving Metrics")
    outfile_path = f"{filename}_metrics.csv"
    output_paths.append(outfile_path)
    with open(outfile_path, "w") as outfile:
        outfile.write("pdb,sequence,charge,isoelectric_point,molecular_weight,molar_extinction\n")
        for pdb, seq_list in pdb_to_sampled.items():
            for seq in seq_list:
                outfile.write(f"{pdb},{seq[0]},{seq[1]},{seq[2]},{seq[3]},{seq[4]}\n")
    return output_paths


def _category_letters(rotamer_categories, n_cls: int) -> np.ndarray:
    if rotamer_categories is not None and len(rotamer_categories):
        res = np.array(rotamer_categories)
    else:
        res = np.array(list(standard_amino_acids.keys()))
    if len(res) < n_cls:
        raise ValueError(f"{len(res)} category names for {n_cls} probability columns")
    return res


def random_choice_prob_index(
    probs: np.ndarray,
    axis: int = 1,
    return_seq: bool = True,
    rotamer_categories: t.Optional[np.ndarray] = None,
) -> np.ndarray:
    """reference sampling_utils.py:53-90: one uniform per residue from np.random.rand, inverse CDF
    (first index whose running sum exceeds r, 0 if none) — computed on the GPU."""
    probs = np.asarray(probs, dtype=np.float64)
    p = probs if axis == 1 else probs.T
    r = np.random.rand(p.shape[0])
    idxs = _sampler.sample_indices(p, 1, uniforms=r[None, :])[0].astype(np.int64)
    if return_seq:
        return _category_letters(rotamer_categories, p.shape[1])[idxs]
    return idxs


def sample_from_sequences(
    pdb: str,
    sample_n: int,
    pdb_to_probability: dict,
    rotamer_categories: t.Optional[np.ndarray],
) -> dict:
    """reference sampling_utils.py:93-136 -> {pdb: [(seq, charge, pI, MW, eps280), ...]}."""
    probs = np.array(pdb_to_probability[pdb], dtype=np.float64)
    n_res, n_cls = probs.shape
    cats = _category_letters(rotamer_categories, n_cls)
    r = np.random.rand(sample_n, n_res) if sample_n else np.empty((0, n_res))
    if all(len(c) == 1 for c in cats[:n_cls]):
        _, letters = _sampler.sample_indices(probs, sample_n, uniforms=r, letters="".join(cats[:n_cls]))
        seqs = [row.tobytes().decode("ascii") for row in letters]
    else:  # multi-character category names (full rotamer labels): join on the host like the reference
        idx = _sampler.sample_indices(probs, sample_n, uniforms=r)
        seqs = ["".join(cats[row]) for row in idx]
    metrics = seq_metrics_batch(seqs) if seqs else np.empty((0, 4))
    return {pdb: [(s, *(float(x) for x in m)) for s, m in zip(seqs, metrics)]}


def apply_temp_to_probs(probs: np.ndarray, t: float = 1.0):
    """reference sampling_utils.py:139-161: probs**(1/t), rows renormalised — on the GPU, fp64."""
    return _sampler.apply_temperature(np.array(probs, dtype=np.float64), t)


def sample_with_multiprocessing(workers, pdb_codes, sample_n, pdb_to_probability, flat_categories):
    """reference sampling_utils.py:164-197.  The reference fans PDB keys over a process Pool (every
    forked worker inherits the SAME generator state, so different PDBs can receive identical uniform
    streams — Appendix C-1).  Here the keys run in order in this process, one fused launch each, and
    consume one continuous stream; ``workers`` is accepted and ignored."""
    pdb_to_sample = {}
    for pdb in pdb_codes:
        pdb_to_sample.update(sample_from_sequences(pdb, sample_n, pdb_to_probability, flat_categories))
    return pdb_to_sample
