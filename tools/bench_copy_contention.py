#!/usr/bin/env python
"""Does a concurrent host->device copy slow the convolution kernels down?  Device-resident forward passes, alone and
with another thread uploading 227 MB buffers (pageable and page-locked) to the same GPU the whole time."""
import json, os, sys, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "timed-design_amd"))
from timed_hip import _lib, engine, synth
import ctypes as C
lib = _lib.load()
cfg, w = synth.timed_synth(20)
m = engine.HipFrameModel.from_keras(cfg, w); m.set_chunk(1024)
n = 16384
d_in = engine.DeviceBuffer(n * 21 ** 3 * 6 * 4); d_out = engine.DeviceBuffer(n * 80)
_lib.check(lib.th_dev_synth_frames(0, C.c_void_p(d_in.ptr), n, 21, 6, 200, 1))
side = engine.DeviceBuffer(1024 * 21 ** 3 * 6 * 4)
host = np.ones(1024 * 21 ** 3 * 6, np.float32)
pin, own = engine.pinned_empty(host.shape, np.float32); pin[...] = 1


def rate(reps=4):
    m.predict_device(d_in.ptr, n, d_out.ptr)
    t0 = time.perf_counter()
    for _ in range(reps):
        m.predict_device(d_in.ptr, n, d_out.ptr)
    return reps * n / (time.perf_counter() - t0)


res = {"alone": rate()}
for name, buf in (("with_pageable_h2d", host), ("with_pinned_h2d", pin)):
    stop = threading.Event(); cnt = [0]
    def pump():
        while not stop.is_set():
            side.upload(buf); cnt[0] += 1
    t = threading.Thread(target=pump); t.start()
    t0 = time.perf_counter(); res[name] = rate(); dt = time.perf_counter() - t0
    stop.set(); t.join()
    res[name + "_copy_GBps"] = cnt[0] * buf.nbytes / dt / 1e9
print(json.dumps(res))
