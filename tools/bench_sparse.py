#!/usr/bin/env python
"""Sparse vs dense transport through the C ABI, no predict.py around it: frames/s of back-to-back th_predict_async calls on
page-locked batches of `group` frames (3 tickets in flight), and of the same frames resident on the device.
    python tools/bench_sparse.py [group] [groups]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "timed-design_amd"))
from timed_hip import _lib, engine, synth
group = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
groups = int(sys.argv[2]) if len(sys.argv) > 2 else 40
cfg, w = synth.timed_synth(20)
m = engine.HipFrameModel.from_keras(cfg, w)
x = synth.synthetic_frames(group, seed=5)
sf = engine.SparseFrames.from_dense(x)
ring = engine.BlobRing(4, 4)
dense_pinned, owner = engine.pinned_empty(x.shape, np.float32)
dense_pinned[:] = x
want = m.predict(x)


def pump(make):
    pend = []
    t0 = time.perf_counter()
    for k in range(groups):
        pend.append(make())
        if len(pend) == 3:
            t, slot = pend.pop(0)
            r = t.result()
            if slot is not None:
                ring.release(slot)
    for t, slot in pend:
        r = t.result()
        if slot is not None:
            ring.release(slot)
    dt = time.perf_counter() - t0
    assert r.tobytes() == want.tobytes()
    return groups * group / dt


def sparse_call():
    staged, slot = ring.stage(sf)
    return m.predict_async(staged), slot


for name, fn in (("dense pinned", lambda: (m.predict_async(dense_pinned), None)), ("sparse staged", sparse_call),
                 ("sparse blob ready", None)):
    if fn is None:
        staged, slot = ring.stage(sf)
        fn = lambda: (m.predict_async(staged), None)
    pump(fn)
    print(f"{name:18s} group {group}: {pump(fn):10.0f} frames/s   (blob {sf.blob_bytes / group:.0f} B/frame, dense {x.nbytes / group:.0f})")
d_in = engine.DeviceBuffer(x.nbytes); d_in.upload(x)
d_out = engine.DeviceBuffer(group * 20 * 4)
m.predict_device(d_in.ptr, group, d_out.ptr)
t0 = time.perf_counter()
for _ in range(groups):
    m.predict_device(d_in.ptr, group, d_out.ptr)
print(f"device resident    group {group}: {groups * group / (time.perf_counter() - t0):10.0f} frames/s")
