#!/usr/bin/env python
"""Do host->device copies from a memory-mapped .npy (what frame packs hand out) overlap with the kernels like copies from
ordinary NumPy memory do?  Streams 16 batches of 1024 frames through th_predict_async (two tickets in flight)."""
import json, os, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "timed-design_amd"))
from timed_hip import engine, synth
cfg, w = synth.timed_synth(20)
m = engine.HipFrameModel.from_keras(cfg, w); m.set_chunk(1024)
n, B = 16384, 1024
base = synth.synthetic_frames(256, seed=1)
ram = np.concatenate([base] * (n // 256))


def stream(x, prep=lambda a: a):
    pend, outs = [], []
    t0 = time.perf_counter()
    for lo in range(0, n, B):
        pend.append(m.predict_async(prep(x[lo:lo + B])))
        if len(pend) > 1:
            outs.append(pend.pop(0).result())
    outs += [p.result() for p in pend]
    return n / (time.perf_counter() - t0)


res = {}
with tempfile.TemporaryDirectory() as td:
    path = os.path.join(td, "f.npy")
    np.save(path, ram)
    mm = np.load(path, mmap_mode="r")
    for name, x, prep in (("ram", ram, lambda a: a), ("memmap", mm, lambda a: a), ("memmap_copy_to_ram", mm, lambda a: np.array(a)),
                          ("memmap_private", np.load(path, mmap_mode="c"), lambda a: a)):
        stream(x, prep)
        res[name] = max(stream(x, prep) for _ in range(2))
print(json.dumps(res))
