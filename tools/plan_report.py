#!/usr/bin/env python
"""What would a given model run?  For every plan step of a Keras `model_config` (a synthetic topology, a width / padding /
input-channel variant of one, or a real `.h5`): the kernel the planner chose, the step's own and direct-form FLOPs per frame,
the reason a 3x3x3 layer stayed on a direct kernel, and — with --measure — its time per 4096 frames and the fraction of its
matrix pipe's peak (fp32-input MFMA 157.3 TFLOP/s; bf16x3-split steps 6 products on the 2500 TFLOP/s bf16 pipe).

    python tools/plan_report.py [--measure] [--chunk 4096] [NAME | model.h5 | model.pack ...]      (GPU box: plans are made at load)

NAME: timed, timed_rotamer, densecpd, prodconn, or one of the variants below (default: all of them).  The fast paths are
shape-specific (conv_wino: 5^3 'same', Cin % 32 == 0, Cout >= 64; conv_wf: 10^3 'same', Cin % 4 == 0; first layer: Cin <= 8,
Cout <= 32): this table is the answer to "which kernel, at what rate" for shapes other than the benchmark's.
"""
from __future__ import annotations

import argparse
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "timed-design_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def _timed_valid(n_classes=20, in_channels=6, widths=(32, 64, 128, 128, 256)):
    """TIMED with `valid` convolutions: 21 -> 19 -> (pool) 9 -> 7 -> (pool) 3 -> 1: every layer off the 'same' fast paths"""
    from timed_hip import synth
    b = synth.KerasGraphBuilder((21, 21, 21, in_channels), seed=4321, name="timed_valid")
    x = b.input_name
    for i, c in enumerate(widths[:3]):                                # 21 -> 19 -> pool 9 -> 7 -> pool 3 -> 1
        x = b.conv3d(x, c, 3, padding="valid")
        x = b.elu(x)
        x = b.batchnorm(x)
        if i in (0, 1):
            x = b.maxpool(x, 2)
    x = b.conv3d(x, n_classes, 1, padding="valid")
    x = b.gap(x)
    x = b.softmax(x)
    return b.finish(x)


def _timed_valid_first(n_classes=20):
    """TIMED whose FIRST block is `valid` (21 -> 19 -> pool 9), the rest 'same': the 9^3 / 4^3 volumes of the VERDICT's question"""
    from timed_hip import synth
    b = synth.KerasGraphBuilder((21, 21, 21, 6), seed=4321, name="timed_valid_first")
    x = b.conv3d(b.input_name, 32, 3, padding="valid")
    x = b.elu(x); x = b.batchnorm(x); x = b.maxpool(x, 2)             # 19 -> 9
    x = b.conv3d(x, 64, 3, padding="same")
    x = b.elu(x); x = b.batchnorm(x); x = b.maxpool(x, 2)             # 9 -> 4
    for c in (128, 128, 256, n_classes):
        x = b.conv3d(x, c, 3, padding="same")
        x = b.elu(x); x = b.batchnorm(x)
    x = b.gap(x)
    x = b.softmax(x)
    return b.finish(x)


def variants():
    from timed_hip import synth
    v = dict(synth.TOPOLOGIES)
    v["timed_c5"] = lambda: synth.timed_synth(20, in_channels=5)                                   # plain CNOCBCA: 5 atom channels
    v["timed_w48"] = lambda: synth.timed_synth(20, widths=(24, 48, 96, 96, 192))                   # widths that are not multiples of 32
    v["timed_w40"] = lambda: synth.timed_synth(20, widths=(40, 72, 136, 136, 264))                 # nor of 16
    v["timed_deep"] = lambda: synth.timed_synth(20, widths=(32, 64, 128, 128, 256, 256, 256))      # TIMED_Deep-like: two more 5^3 blocks
    v["timed_valid"] = _timed_valid
    v["timed_valid_first"] = _timed_valid_first
    v["densecpd_c5"] = lambda: synth.densecpd_synth(20, in_channels=5)
    return v


def load(name, device):
    from timed_hip import engine
    if os.path.exists(name):
        return engine.load_model(name, device=device), os.path.basename(name)
    cfg, w = variants()[name]()
    return engine.HipFrameModel.from_keras(cfg, w, device=device, name=name), name


def short(label):
    layer = label.split(":", 1)[0]
    m = re.search(r"\[([^\]]+)\]\s*$", label)
    kern = m.group(1) if m else label.split(":", 1)[-1].strip()
    why = re.search(r"\(direct form: (.*)\) \[", label)
    return layer, kern, (why.group(1) if why else "")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--measure", action="store_true", help="time every step on device-resident synthetic frames")
    ap.add_argument("--chunk", type=int, default=4096)
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("names", nargs="*")
    args = ap.parse_args()
    import ctypes as C
    import bench_legs
    from timed_hip import _lib, engine
    lib = _lib.load()
    for name in (args.names or list(variants())):
        model, shown = load(name, args.device)
        model.set_chunk(args.chunk)
        D, H, W, Cc = model.input_shape
        cost = model.cost()
        fps = None
        if args.measure:
            n = args.chunk
            d_in = engine.DeviceBuffer(n * D * H * W * Cc * 4, args.device)
            d_out = engine.DeviceBuffer(n * model.n_classes * 4, args.device)
            _lib.check(lib.th_dev_synth_frames(args.device, C.c_void_p(d_in.ptr), n, D, Cc, 200, 1234))
            model.predict_device(d_in.ptr, n, d_out.ptr)
            model.profile(1)
            reps = 3
            for _ in range(reps):
                model.predict_device(d_in.ptr, n, d_out.ptr)
        steps = model.steps()
        g = model.guard()
        own = sum(s["flops"] for s in steps)
        print(f"# {shown}: input {D}x{H}x{W}x{Cc}, {model.n_classes} classes; direct-form {cost['algo_flops'] / 1e6:.1f} MFLOP per frame, the kernels' own "
              f"{own / 1e6:.1f}; guard state {g['state']} (max |dlogit| {g['max_dlogit']:.2e} of scale {g['logit_scale']:.2f}); knobs '{model.knobs()}'")
        if args.measure:
            tot = sum(s["ms"] for s in steps) / reps
            fps = args.chunk / (tot * 1e-3) if tot else 0.0
            print(f"#   measured: {tot:.3f} ms per {args.chunk} frames over the plan steps = {fps:,.0f} frames/s; matrix pipes busy at peak for "
                  f"{bench_legs.pipe_time_frac(steps, fps):.3f} of that time")
        hdr = f"{'layer':28s} {'kernel':46s} {'own MFLOP':>10s} {'direct MFLOP':>12s}"
        if args.measure:
            hdr += f" {'ms/chunk':>9s} {'share':>6s} {'pipe':>5s} {'of peak':>8s} {'GB/s':>7s}"
        print(hdr + "  why not a minimal-filtering form")
        for s in steps:
            layer, kern, why = short(s["label"])
            direct = s.get("direct_flops")
            row = f"{layer[:28]:28s} {kern[:46]:46s} {s['flops'] / 1e6:10.2f} {'' if direct is None else format(direct / 1e6, '12.2f'):>12s}"
            if args.measure:
                ms = s["ms"] / reps
                if ms and s["flops"]:
                    pipe, peak, pf = bench_legs.step_pipe(s)
                    frac = pf * args.chunk / (ms * 1e-3) / 1e12 / peak
                    row += f" {ms:9.3f} {100 * ms / tot:5.1f}% {pipe:>5s} {frac:8.3f} {s['bytes'] * args.chunk / (ms * 1e-3) / 1e9:7.0f}"
                elif ms:
                    row += f" {ms:9.3f} {100 * ms / tot:5.1f}% {'':>5s} {'':>8s} {s['bytes'] * args.chunk / (ms * 1e-3) / 1e9:7.0f}"
                else:
                    row += " " * 40
            print(row + ("  " + why if why else ""))
        print()
        model.close()
        if args.measure:
            d_in.free(); d_out.free()


if __name__ == "__main__":
    main()
