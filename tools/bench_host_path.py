#!/usr/bin/env python
"""Host-resident frames through the C ABI vs the device-resident rate.

    python tools/bench_host_path.py [--frames 16384] [--topology timed]

Legs (frames/s; every leg computes the same frames):
  device         th_predict_device, frames resident in HBM (what bench.py's `value` measures)
  sync_pageable  one th_predict call over all frames in pageable NumPy memory
  sync_pinned    the same from page-locked memory (th_host_alloc)
  async_*        a loop of batches of B frames through th_predict_async/_wait, two tickets in flight
  u8             boolean (1 B/voxel) frames
"""
import argparse, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "timed-design_amd"))
from timed_hip import _lib, engine, synth  # noqa: E402
import ctypes as C  # noqa: E402


def timed(fn, reps=3):
    fn()
    best = 1e30
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); best = min(best, time.perf_counter() - t0)
    return best


def run(a):
    lib = _lib.load()
    cfg, weights = synth.TOPOLOGIES[a.topology]()
    model = engine.HipFrameModel.from_keras(cfg, weights, device=0, name=a.topology)
    model.set_chunk(a.chunk)
    n = a.frames
    D, H, W, Cc = model.input_shape
    ff = D * H * W * Cc
    d_frames = engine.DeviceBuffer(n * ff * 4, 0)
    d_probs = engine.DeviceBuffer(n * model.n_classes * 4, 0)
    _lib.check(lib.th_dev_synth_frames(0, C.c_void_p(d_frames.ptr), n, D, Cc, 200, 1234))
    host = d_frames.download((n, D, H, W, Cc), np.float32)
    pinned, own = engine.pinned_empty(host.shape, np.float32)
    pinned[...] = host
    res = dict(frames=n, chunk=a.chunk, topology=a.topology, host_cores=os.cpu_count())

    def dev():
        model.predict_device(d_frames.ptr, n, d_probs.ptr)
        _lib.check(lib.th_dev_sync(0))
    res["device"] = n / timed(dev)
    ref = d_probs.download((n, model.n_classes), np.float32)
    out = {}

    def sync(x, key):
        def f():
            out[key] = model.predict(x)
        return f
    res["sync_pageable"] = n / timed(sync(host, "a"))
    res["sync_pinned"] = n / timed(sync(pinned, "b"))
    assert np.array_equal(out["a"], ref) and np.array_equal(out["b"], ref)

    def stream(x, B, depth):
        def f():
            pend, outs = [], []
            for lo in range(0, n, B):
                pend.append(model.predict_async(x[lo:lo + B]))
                if len(pend) > depth - 1:
                    outs.append(pend.pop(0).result())
            outs += [p.result() for p in pend]
            out["s"] = np.concatenate(outs)
        return f
    for B in a.batches:
        for name, x in (("pageable", host), ("pinned", pinned)):
            for depth in (1, 2, 3):
                res[f"async_{name}_B{B}_d{depth}"] = n / timed(stream(x, B, depth))
                assert np.array_equal(out["s"], ref)
    # boolean frames: 1 byte per voxel over PCIe, expanded by the first-layer kernel
    hb = (host > 0.05).astype(np.uint8)
    pb, ownb = engine.pinned_empty(hb.shape, np.uint8)
    pb[...] = hb
    res["sync_pageable_u8"] = n / timed(sync(hb, "u"))
    res["sync_pinned_u8"] = n / timed(sync(pb, "v"))
    assert np.array_equal(out["u"], out["v"])
    d_u8 = engine.DeviceBuffer(hb.nbytes, 0); d_u8.upload(hb)

    def dev8():
        model.predict_device(d_u8.ptr, n, d_probs.ptr, dtype=_lib.TH_U8)
        _lib.check(lib.th_dev_sync(0))
    res["device_u8"] = n / timed(dev8)
    for B in a.batches:
        res[f"async_pinned_u8_B{B}_d2"] = n / timed(stream(pb, B, 2))
    print(json.dumps(res))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=16384)
    ap.add_argument("--chunk", type=int, default=1024)
    ap.add_argument("--topology", default="timed")
    ap.add_argument("--batches", type=int, nargs="*", default=[512, 1024, 4096])
    run(ap.parse_args())
