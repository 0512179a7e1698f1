#!/usr/bin/env python
"""Run the seeded random-graph parity check of tests/test_gpu_fuzz.py over many more seeds (GPU box):
    python tools/fuzz_soak.py 48 600"""
import sys, importlib.util, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "timed-design_amd"))
spec = importlib.util.spec_from_file_location("fz", os.path.join(ROOT, "tests", "test_gpu_fuzz.py")); fz = importlib.util.module_from_spec(spec); spec.loader.exec_module(fz)
from oracle import cnn_oracle
from timed_hip import engine
lo, hi = int(sys.argv[1]), int(sys.argv[2])
bad = []
worst = 0.0
for seed in range(lo, hi):
    cfg, w, fr = fz._random_net(seed)
    want = cnn_oracle.forward(cfg, w, fr, np.float32)
    try:
        m = engine.HipFrameModel.from_keras(cfg, w); m.set_chunk(1 + seed % 5); got = m.predict(fr); m.close()
    except Exception as e:
        bad.append((seed, repr(e)[:200])); continue
    rel = float(np.abs(got - want).max()) / max(1.0, float(np.abs(want).max()))
    worst = max(worst, rel)
    if got.shape != want.shape or not rel <= 2e-5:
        bad.append((seed, rel))
print("seeds", lo, hi, "worst rel err %.3g" % worst, "failures", bad)
