#!/usr/bin/env python
"""Host->device copy rate from pageable NumPy memory: one hipMemcpy vs the same range split over threads."""
import ctypes as C, os, sys, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "timed-design_amd"))
from timed_hip import _lib, engine
lib = _lib.load()
nbytes = 1 << 30
host = np.ones(nbytes, np.uint8)
dev = engine.DeviceBuffer(nbytes, 0)
def up(lo, hi):
    _lib.check(lib.th_dev_upload(0, C.c_void_p(dev.ptr + lo), C.c_void_p(host.ctypes.data + lo), hi - lo))
up(0, nbytes)
for nt in (1, 2, 4, 8):
    best = 0
    for rep in range(3):
        ths = [threading.Thread(target=up, args=(nbytes * i // nt, nbytes * (i + 1) // nt)) for i in range(nt)]
        t0 = time.perf_counter(); [t.start() for t in ths]; [t.join() for t in ths]; dt = time.perf_counter() - t0
        best = max(best, nbytes / dt / 1e9)
    print(f"{nt} thread(s): {best:.1f} GB/s")
