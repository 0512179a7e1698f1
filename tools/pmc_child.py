#!/usr/bin/env python
"""Workload of bench.py's in-run PMC passes (tools/bench_legs.py: pmc_traffic_inrun): ONE chunk of every BASELINE topology
through th_predict_device, run under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` and again under `--pmc WRITE_SIZE`
(separate passes, never combined with another trace domain).  Per topology: a warm-up pass, then a k_synth_frames
marker launch and the measured pass, in which every plan step dispatches exactly once, in plan order (the parent splits
the trace on the markers and reads the segment after each topology's second one).
Prints one JSON line: per topology the plan-step labels in order."""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "timed-design_amd"))


def main():
    from timed_hip import _lib, engine, synth
    chunk = int(sys.argv[1])
    names = sys.argv[2:]
    lib = _lib.load()
    out = {}
    for name in names:
        cfg, weights = synth.TOPOLOGIES[name]()
        model = engine.HipFrameModel.from_keras(cfg, weights, device=0, name=name)
        model.set_chunk(chunk)
        D, H, W, Cc = model.input_shape
        d_frames = engine.DeviceBuffer(chunk * D * H * W * Cc * 4, 0)
        d_probs = engine.DeviceBuffer(chunk * model.n_classes * 4, 0)
        # warm-up pass first (arena allocation + memset fills, lazy function attributes), then the marker launch and the
        # measured pass: the parent reads the segment after each topology's SECOND k_synth_frames
        _lib.check(lib.th_dev_synth_frames(0, C.c_void_p(d_frames.ptr), chunk, D, Cc, 200, 1234))
        model.predict_device(d_frames.ptr, chunk, d_probs.ptr)
        _lib.check(lib.th_dev_sync(0))
        for _ in range(int(os.environ.get("PMC_CHILD_REPS", "1"))):      # (tools/roofline_table.py averages several passes)
            _lib.check(lib.th_dev_synth_frames(0, C.c_void_p(d_frames.ptr), chunk, D, Cc, 200, 1234))    # the marker launch
            model.predict_device(d_frames.ptr, chunk, d_probs.ptr)
            _lib.check(lib.th_dev_sync(0))
        out[name] = [s["label"] for s in model.steps()]
        if os.environ.get("PMC_CHILD_COSTS"):
            out[name + "/costs"] = [[s["flops"], s["exec_flops"], s["bytes"]] for s in model.steps()]
        model.close()
        d_frames.free()
        d_probs.free()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
