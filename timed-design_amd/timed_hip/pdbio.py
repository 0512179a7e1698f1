"""Minimal PDB reader for the voxeliser: ATOM/HETATM records of fixed-column PDB files, plain or
gzipped (the reference ships tests/testing_files/1ubq.pdb1.gz and passes ``is_pdb_gzipped`` to aposteriori, ui.py:81).
Only what voxelisation needs: coordinates, atom / residue names, chain, residue number (+ insertion code), element,
model number.  Alternate locations: the first one seen for an atom name within a residue wins (blank or 'A').
"""
from __future__ import annotations

import gzip
from dataclasses import dataclass, field
from typing import Dict, List, Tuple

import numpy as np


@dataclass
class Residue:
    chain: str
    number: str            # residue number + insertion code as written ("12", "12A")
    name: str              # three-letter code
    atoms: Dict[str, np.ndarray] = field(default_factory=dict)   # atom name -> xyz (float64)
    elements: Dict[str, str] = field(default_factory=dict)
    hetero: bool = False


@dataclass
class Model:
    number: int
    residues: List[Residue]


def _open(path):
    path = str(path)
    if path.endswith(".gz"):
        return gzip.open(path, "rt", encoding="ascii", errors="replace")
    return open(path, "r", encoding="ascii", errors="replace")


def read_pdb(path) -> List[Model]:
    """All models of the file (a file without MODEL records is one model, number 1)."""
    models: List[Model] = []
    residues: List[Residue] = []
    index: Dict[Tuple[str, str, str], Residue] = {}
    number = 1

    def close_model():
        nonlocal residues, index
        if residues:
            models.append(Model(number, residues))
        residues, index = [], {}

    with _open(path) as f:
        for line in f:
            rec = line[:6]
            if rec == "MODEL ":
                close_model()
                try:
                    number = int(line[10:14])
                except ValueError:
                    number = len(models) + 1
            elif rec == "ENDMDL":
                close_model()
                number += 1
            elif rec in ("ATOM  ", "HETATM") and len(line) >= 54:
                name = line[12:16].strip()
                resname = line[17:20].strip()
                chain = line[21].strip() or "A"
                resnum = (line[22:26].strip() + line[26].strip())
                try:
                    xyz = np.array([float(line[30:38]), float(line[38:46]), float(line[46:54])])
                except ValueError:
                    continue
                key = (chain, resnum, resname)
                res = index.get(key)
                if res is None:
                    res = index[key] = Residue(chain, resnum, resname, hetero=rec == "HETATM")
                    residues.append(res)
                if name in res.atoms:            # another alternate location of an atom we already have
                    continue
                res.atoms[name] = xyz
                el = line[76:78].strip() if len(line) >= 78 else ""
                res.elements[name] = el or name[:1]
    close_model()
    return models
