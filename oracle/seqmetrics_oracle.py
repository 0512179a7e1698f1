"""CPU ORACLE (test infrastructure only) — scalar restatement of the four ampal sequence metrics the reference's
sampler computes per drawn sequence (reference design_utils/analyse_utils.py:351-371 calls
ampal.analyse_protein.sequence_charge / sequence_isoelectric_point / sequence_molecular_weight /
sequence_molar_extinction_280; call site design_utils/sampling_utils.py:132).

    *** PARITY UNPINNED ***  ampal==1.5.1 (requirements.txt) is a third-party package absent from /root/reference and
    from this image, and the reference holds no test or golden value for these four numbers.  What follows restates
    ampal's published algorithm the way ampal itself evaluates it — plain Python floats, one residue CLASS at a time in
    first-occurrence order of a collections.Counter, `10 ** x` on Python floats — with the constant tables restated in
    this file.  It deliberately shares NO code, table object or accumulation order with the product's
    design_utils/analyse_utils.py / csrc/sampler.hip (histogram dot products in alphabetical class order), so a
    transcription slip in either shows up as a disagreement; tests/test_oracle_seqmetrics.py also holds three
    sequences worked out with 50-digit decimal arithmetic.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

from collections import Counter

# average residue masses (Da, residue = amino acid minus water), side-chain pKa / charge sign, molar extinction at 280 nm
RESIDUE_MWT = {
    "A": 71.0779, "R": 156.1857, "N": 114.1026, "D": 115.0874, "C": 103.1429, "E": 129.114, "Q": 128.1292, "G": 57.0513,
    "H": 137.1393, "I": 113.1576, "L": 113.1576, "K": 128.1723, "M": 131.1961, "F": 147.1739, "P": 97.1152, "S": 87.0773,
    "T": 101.1039, "W": 186.2099, "Y": 163.1733, "V": 99.1311,
}
WATER_MASS = 18.01528
RESIDUE_EXT_280 = {"W": 5690, "Y": 1280, "C": 120}
RESIDUE_PKA = {"D": 3.65, "E": 4.25, "H": 6.1, "C": 8.3, "Y": 10.1, "K": 10.53, "R": 12.48, "N-term": 8.0, "C-term": 3.1}
RESIDUE_CHARGE = {"D": -1, "E": -1, "H": +1, "C": -1, "Y": -1, "K": +1, "R": +1, "N-term": +1, "C-term": -1}


def partial_charge(aa: str, ph: float) -> float:
    """fraction of the group that is charged at `ph` (Henderson-Hasselbalch); 0 for groups without a pKa"""
    if aa not in RESIDUE_PKA:
        return 0.0
    difference = ph - RESIDUE_PKA[aa]
    if RESIDUE_CHARGE[aa] > 0:
        difference *= -1
    ratio = (10 ** difference) / (1 + 10 ** difference)
    return ratio


def sequence_charge(seq: str, ph: float = 7.4) -> float:
    total = sum(partial_charge(aa, ph) * RESIDUE_CHARGE.get(aa, 0) * n for aa, n in Counter(seq).items())
    total += partial_charge("N-term", ph) * RESIDUE_CHARGE["N-term"]
    total += partial_charge("C-term", ph) * RESIDUE_CHARGE["C-term"]
    return total


def charge_series(seq: str, granularity: float = 0.1):
    import numpy
    ph_range = numpy.arange(1, 13, granularity)
    return ph_range, [sequence_charge(seq, ph) for ph in ph_range]


def sequence_isoelectric_point(seq: str, granularity: float = 0.1) -> float:
    ph_range, charge_at_ph = charge_series(seq, granularity)
    abs_charge = [abs(c) for c in charge_at_ph]
    pi_index = min(enumerate(abs_charge), key=lambda x: x[1])[0]      # first minimum
    return float(ph_range[pi_index])


def sequence_molecular_weight(seq: str) -> float:
    return sum(RESIDUE_MWT.get(aa, 0.0) * n for aa, n in Counter(seq).items()) + WATER_MASS


def sequence_molar_extinction_280(seq: str) -> float:
    return sum(RESIDUE_EXT_280.get(aa, 0) * n for aa, n in Counter(seq).items())


def seq_metrics(seq: str):
    """(charge at pH 7.4, isoelectric point, molecular weight, molar extinction at 280 nm) — the tuple
    reference analyse_utils.py:367-371 returns"""
    return (sequence_charge(seq), sequence_isoelectric_point(seq), sequence_molecular_weight(seq),
            sequence_molar_extinction_280(seq))
