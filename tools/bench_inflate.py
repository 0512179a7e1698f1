#!/usr/bin/env python
"""th_inflate_many microbenchmark: n identical-size streams of one kind (zeros / random / sparse float64), wall time incl. copies."""
import ctypes as C, os, sys, time, zlib
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "timed-design_amd"))
from timed_hip import _lib
lib = _lib.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
size = int(sys.argv[2]) if len(sys.argv) > 2 else 14000
rng = np.random.default_rng(0)
def sparse():
    g = np.zeros(size // 8); idx = rng.integers(0, g.size, g.size // 12); g[idx] = rng.random(len(idx)); return g.tobytes()
kinds = {"zeros": lambda: bytes(size), "random": lambda: rng.integers(0, 256, size, dtype=np.uint8).tobytes(), "sparse_f64": sparse,
         "text": lambda: (b"the quick brown fox " * (size // 20 + 1))[:size]}
for name, gen in kinds.items():
    base = [zlib.compress(gen(), 4) for _ in range(16)]
    streams = [base[i % 16] for i in range(n)]
    comp = b"".join(streams)
    src_len = np.array([len(s) for s in streams], dtype=np.int64)
    src_off = np.concatenate([[0], np.cumsum(src_len)[:-1]]).astype(np.int64)
    sz8 = (size + 7) // 8 * 8
    dst_off = (np.arange(n, dtype=np.int64) * sz8)
    dst_len = np.full(n, size, dtype=np.int64)
    out = np.empty(n * sz8, dtype=np.uint8)
    cbuf = np.frombuffer(comp, dtype=np.uint8)
    p64 = lambda a: a.ctypes.data_as(C.POINTER(C.c_int64))
    for rep in range(2):
        t0 = time.perf_counter()
        rc = lib.th_inflate_many(0, cbuf.ctypes.data_as(C.c_void_p), len(comp), n, p64(src_off), p64(src_len), p64(dst_off), p64(dst_len),
                                 out.ctypes.data_as(C.c_void_p), out.size, 1, None)
        dt = time.perf_counter() - t0
    assert rc == 0, lib.th_last_error()
    print(f"{name:12s} n={n} size={size} comp={len(base[0])} B/stream: {dt*1e3:8.1f} ms  ({n*size/dt/1e9:6.2f} GB/s out)")
