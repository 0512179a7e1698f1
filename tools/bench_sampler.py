#!/usr/bin/env python
"""BASELINE config 5: MC sampling, 1k sequences x 300 residues at T in {0.1, 0.5, 1.0} — GPU (fused
kernel, one launch per call) vs the NumPy oracle loop the reference runs (sampling_utils.py:123-128)."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "timed-design_amd"))
from oracle import sampler_oracle as so
from timed_hip import sampler

rng = np.random.default_rng(7)
p = rng.dirichlet(np.full(20, 0.3), size=300).astype(np.float16).astype(np.float64)
out = {}
for t in (0.1, 0.5, 1.0):
    for mode in ("philox", "mt19937"):
        sampler.sample_indices(p, 1000, temperature=t, rng=mode, seed=42)  # warm
        t0 = time.perf_counter()
        reps = 20
        for _ in range(reps):
            idx = sampler.sample_indices(p, 1000, temperature=t, rng=mode, seed=42)
        dt = (time.perf_counter() - t0) / reps
        out[f"gpu_{mode}_T{t}"] = dict(ms=dt * 1e3, draws_per_s=300e3 / dt, seqs_per_s=1e3 / dt)
    # CPU: the reference's own loop shape (one call per sample, array rebuilt from list-of-lists)
    q = so.apply_temp(p, t) if t != 1.0 else p
    ql = [list(r) for r in q]
    np.random.seed(42)
    t0 = time.perf_counter()
    for _ in range(1000):
        r = np.random.rand(300)
        so.choice_indices(np.array(ql), r)
    dt = time.perf_counter() - t0
    out[f"cpu_numpy_T{t}"] = dict(ms=dt * 1e3, draws_per_s=300e3 / dt, seqs_per_s=1e3 / dt)
print(json.dumps(out, indent=1))
