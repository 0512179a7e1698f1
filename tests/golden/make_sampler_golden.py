#!/usr/bin/env python
"""Generate tests/golden/sampler_golden.npz by IMPORTING the reference sampler — build container only.

/root/reference is importable here once its heavyweight module-level imports (ampal, aposteriori,
h5py, logomaker, ... — none needed by the functions we call) are stubbed in sys.modules
(SURVEY.md §8c).  Nothing of the reference travels: only inputs and the outputs it computed are
stored.  Also stores rocRAND Philox4x32-10 reference draws produced by compiling rocRAND's own
header-only (host+device) generator for the HOST with hipcc (philox_ref.cpp, below).

Usage:  python tests/golden/make_sampler_golden.py
"""
import os
import subprocess
import sys
import tempfile
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = "/root/reference"


def stub_modules():
    aa = {"A": "ALA", "C": "CYS", "D": "ASP", "E": "GLU", "F": "PHE", "G": "GLY", "H": "HIS", "I": "ILE", "K": "LYS",
          "L": "LEU", "M": "MET", "N": "ASN", "P": "PRO", "Q": "GLN", "R": "ARG", "S": "SER", "T": "THR", "V": "VAL",
          "W": "TRP", "Y": "TYR"}
    chi = {"ARG": 4, "ASN": 2, "ASP": 2, "CYS": 1, "GLN": 3, "GLU": 3, "HIS": 2, "ILE": 2, "LEU": 2, "LYS": 4, "MET": 3,
           "PHE": 2, "PRO": 2, "SER": 1, "THR": 1, "TRP": 2, "TYR": 2, "VAL": 1}

    class _Any(types.ModuleType):
        def __getattr__(self, k):
            if k.startswith("__"):
                raise AttributeError(k)
            return _Any(k)

        def __call__(self, *a, **k):
            return None

    for name in ["ampal", "ampal.amino_acids", "ampal.analyse_protein", "aposteriori", "aposteriori.config",
                 "aposteriori.data_prep", "aposteriori.data_prep.create_frame_data_set", "h5py", "logomaker",
                 "matplotlib", "matplotlib.pyplot", "seaborn", "sklearn", "sklearn.metrics", "sklearn.preprocessing",
                 "scipy.stats", "tqdm", "millify", "Bio", "Bio.PDB"]:
        if name not in sys.modules:
            try:
                __import__(name)
            except Exception:
                sys.modules[name] = _Any(name)
    am = sys.modules["ampal.amino_acids"]
    am.standard_amino_acids = aa
    am.side_chain_dihedrals = {k: [None] * v for k, v in chi.items()}
    am.polarity_Zimmerman = {}
    am.residue_charge = {}
    sys.modules["ampal"].amino_acids = am
    sys.modules["aposteriori.config"].MAKE_FRAME_DATASET_VER = "0.0.0"
    sys.modules["aposteriori.config"].UNCOMMON_RESIDUE_DICT = {}
    sys.modules["aposteriori.data_prep.create_frame_data_set"].DatasetMetadata = object


PHILOX_CPP = r"""
#include <cstdio>
#include <rocrand/rocrand_kernel.h>
int main(int argc, char** argv) {
    unsigned long long seed = strtoull(argv[1], 0, 10), off = strtoull(argv[2], 0, 10);
    int n = atoi(argv[3]);
    for (int d = 0; d < n; ++d) {
        rocrand_state_philox4x32_10 st;
        rocrand_init(seed, off + d, 0, &st);
        printf("%.17g\n", rocrand_uniform_double(&st));
    }
    return 0;
}
"""


def philox_reference(seed, offset, n):
    with tempfile.TemporaryDirectory() as td:
        src = os.path.join(td, "philox_ref.cpp")
        exe = os.path.join(td, "philox_ref")
        open(src, "w").write(PHILOX_CPP)
        subprocess.run(["/opt/rocm/bin/hipcc", "-O1", "--offload-arch=gfx950", "-I/opt/rocm/include", src, "-o", exe],
                       check=True, capture_output=True)
        out = subprocess.run([exe, str(seed), str(offset), str(n)], check=True, capture_output=True, text=True).stdout
    return np.array([float(x) for x in out.split()], dtype=np.float64)


def main():
    stub_modules()
    sys.path.insert(0, REF)
    from design_utils import sampling_utils as ref_su
    from design_utils import utils as ref_utils

    out = {}
    # --- the distribution of the reference's own test (tests/test_sampling_utils.py:5-28) ---
    theo = np.array([[0.01] * 5 + [0.20] + [0.01] * 5 + [0.50, 0.10] + [0.01] * 6 + [0.04]])
    out["theoretical_prob"] = theo
    for t in (1, 0.01, 100, 0.1, 0.5, 2.0):
        out[f"temp_theo_t{t}"] = ref_su.apply_temp_to_probs(theo, t=t)

    rng = np.random.default_rng(7)
    cases = {}
    for n_cls in (20, 338):
        p = rng.dirichlet(np.full(n_cls, 0.3), size=96 if n_cls == 20 else 10)
        cases[f"dir{n_cls}_f64"] = p
        cases[f"dir{n_cls}_f16"] = p.astype(np.float16).astype(np.float64)  # what the CSV round trip yields
    bad = cases["dir20_f16"][:6].copy()
    bad[0] *= 0.5          # row summing to 0.5: falls through to index 0 when r >= 0.5
    bad[1] = 0.0           # all-zero row
    bad[2, 3] = np.nan     # NaN poisons the cumsum from index 3 on
    cases["edge20"] = bad
    for name, p in cases.items():
        out[f"probs_{name}"] = p
        for t in (0.1, 0.5, 1.0, 2.0):
            with np.errstate(all="ignore"):
                out[f"temp_{name}_t{t}"] = ref_su.apply_temp_to_probs(p, t=t)
        for seed in (0, 42):
            np.random.seed(seed)
            n_samp = 5
            rs, idxs, seqs = [], [], []
            for _ in range(n_samp):
                state = np.random.get_state()
                r = np.random.rand(p.shape[0])          # what random_choice_prob_index is about to draw
                np.random.set_state(state)
                idx = ref_su.random_choice_prob_index(p, return_seq=False)
                rs.append(r); idxs.append(idx)
                if p.shape[1] == 20:
                    np.random.set_state(state)
                    seqs.append("".join(ref_su.random_choice_prob_index(p, return_seq=True)))
            out[f"r_{name}_s{seed}"] = np.array(rs)
            out[f"idx_{name}_s{seed}"] = np.array(idxs, dtype=np.int64)
            if seqs:
                out[f"seq_{name}_s{seed}"] = np.array(seqs)
    # legacy MT19937 stream heads (np.random.seed(s); np.random.rand(n))
    for seed in (0, 42, 123456789):
        np.random.seed(seed)
        out[f"mt_s{seed}"] = np.random.rand(2000)
    # rotamer codec tables from the reference
    codec, cats, guide = ref_utils.get_rotamer_codec(return_reduction_guide=True)
    out["rot_categories"] = np.array(cats)
    out["rot_reduction_guide"] = np.array(guide, dtype=np.int64)
    out["rot_codec_argmax"] = np.array([int(np.argmax(codec[i])) for i in range(338)], dtype=np.int64)
    m = rng.random((7, 338))
    out["rot_compress_in"] = m
    out["rot_compress_out"] = ref_utils.compress_rotamer_predictions_to_20(m)
    # rocRAND Philox reference draws
    for seed, off in ((42, 0), (0xDEADBEEFCAFE, 5_000_000_000)):
        out[f"philox_s{seed}_o{off}"] = philox_reference(seed, off, 64)
    path = os.path.join(ROOT, "tests", "golden", "sampler_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes;", len(out), "arrays")


if __name__ == "__main__":
    main()
