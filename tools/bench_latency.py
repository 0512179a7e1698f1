#!/usr/bin/env python
"""Latency of HipFrameModel.predict (host buffers in, host probabilities out) for small batches:
the CLI default batch_size=12 (reference predict.py:255) and one-protein UI requests."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "timed-design_amd"))
from timed_hip import engine, synth
cfg, w = synth.timed_synth(20)
m = engine.HipFrameModel.from_keras(cfg, w)
X = synth.synthetic_frames(512, seed=1)
out = {}
for n in (1, 12, 76, 300, 500):
    x = X[:n].astype(np.float64)   # what load_batch hands over for Gaussian datasets
    for _ in range(3): m.predict(x)
    t0 = time.perf_counter(); reps = 20
    for _ in range(reps): m.predict(x)
    dt = (time.perf_counter() - t0) / reps
    x32 = X[:n]
    for _ in range(3): m.predict(x32)
    t0 = time.perf_counter()
    for _ in range(reps): m.predict(x32)
    dt32 = (time.perf_counter() - t0) / reps
    out[n] = dict(ms_f64=round(dt * 1e3, 3), frames_per_s_f64=round(n / dt), ms_f32=round(dt32 * 1e3, 3), frames_per_s_f32=round(n / dt32))
print(json.dumps(out))
