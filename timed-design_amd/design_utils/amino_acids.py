"""Amino-acid tables the hot path needs (the reference takes them from ampal.amino_acids,
reference design_utils/utils.py:10-15, design_utils/sampling_utils.py:7).

``standard_amino_acids`` is ordered by one-letter code — the order that fixes the column meaning of
every probability matrix (`residue_encoder`, argmax → letter at reference utils.py:655-659) — and
``side_chain_dihedrals`` carries the number of chi angles per residue, which fixes the 338 rotamer
categories (3**n_chi per residue; reference utils.py:430-462).  With these tables
``get_rotamer_codec`` reproduces the reduction guide printed in the reference docstring
(utils.py:425) and the reference-generated fixture tests/golden/sampler_golden.npz.
"""

standard_amino_acids = {
    "A": "ALA", "C": "CYS", "D": "ASP", "E": "GLU", "F": "PHE", "G": "GLY", "H": "HIS", "I": "ILE", "K": "LYS",
    "L": "LEU", "M": "MET", "N": "ASN", "P": "PRO", "Q": "GLN", "R": "ARG", "S": "SER", "T": "THR", "V": "VAL",
    "W": "TRP", "Y": "TYR",
}

# residue -> number of side-chain dihedrals (ALA and GLY have none and get a single "_0" category)
_N_CHI = {"ARG": 4, "ASN": 2, "ASP": 2, "CYS": 1, "GLN": 3, "GLU": 3, "HIS": 2, "ILE": 2, "LEU": 2, "LYS": 4, "MET": 3,
          "PHE": 2, "PRO": 2, "SER": 1, "THR": 1, "TRP": 2, "TYR": 2, "VAL": 1}
side_chain_dihedrals = {res: list(range(n)) for res, n in _N_CHI.items()}

# Stand-in for aposteriori.config.UNCOMMON_RESIDUE_DICT (used at reference utils.py:381-385 to map a
# non-standard residue label onto its parent amino acid while the dataset map is built).  aposteriori
# 2.4.0 is not in the reference tree nor in this image, so its exact table is UNPINNED; this one holds
# the common modified residues of the PDB and can be extended/replaced by the caller
# (design_utils.utils.create_flat_dataset_map(..., uncommon_residue_dict=...)).
UNCOMMON_RESIDUE_DICT = {
    "MSE": "MET", "SEP": "SER", "TPO": "THR", "PTR": "TYR", "HYP": "PRO", "CSO": "CYS", "CME": "CYS", "OCS": "CYS",
    "CSD": "CYS", "CSS": "CYS", "KCX": "LYS", "LLP": "LYS", "MLY": "LYS", "M3L": "LYS", "MLZ": "LYS", "ALY": "LYS",
    "PCA": "GLU", "SEC": "CYS", "PYL": "LYS", "HIC": "HIS", "FME": "MET", "DAL": "ALA", "DAR": "ARG", "DAS": "ASP",
    "DCY": "CYS", "DGL": "GLU", "DGN": "GLN", "DHI": "HIS", "DIL": "ILE", "DLE": "LEU", "DLY": "LYS", "DPN": "PHE",
    "DPR": "PRO", "DSN": "SER", "DSG": "ASN", "DTH": "THR", "DTR": "TRP", "DTY": "TYR", "DVA": "VAL",
}
