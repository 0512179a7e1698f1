#!/usr/bin/env python
"""GPU-box diagnostic: per-layer error of every execution mode vs the oracle, plus step labels.
Run via gpurun; prints compact tables (not a test)."""
import os
import sys
import time
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "timed-design_amd"))
from oracle import cnn_oracle  # noqa: E402
from timed_hip import _lib, engine, synth  # noqa: E402


def main():
    print("devices:", _lib.device_count(), _lib.device_info(0))
    for name, builder, kw, fkw in [
        ("timed_small", synth.timed_synth, dict(n_classes=20, widths=(8, 16, 16), side=9, in_channels=4, bias_std=0.2),
         dict(side=9, channels=4, atoms=30, seed=5)),
        ("timed20", synth.timed_synth, dict(n_classes=20), dict(seed=1234)),
        ("densecpd20", synth.densecpd_synth, dict(n_classes=20), dict(seed=1236)),
        ("prodconn20", synth.prodconn_synth, dict(n_classes=20, bias_std=0.05), dict(seed=1237)),
    ]:
        cfg, weights = builder(**kw)
        frames = synth.synthetic_frames(3, **fkw)
        vals = cnn_oracle.forward(cfg, weights, frames, np.float32, return_all=True)
        out_name = cfg["config"]["output_layers"][0][0]
        for mode, flags in [("fused_mfma", 0), ("fused_direct", _lib.TH_LOAD_NO_MFMA),
                            ("keepall_mfma", _lib.TH_LOAD_KEEP_ALL), ("keepall_direct", _lib.TH_LOAD_KEEP_ALL | _lib.TH_LOAD_NO_MFMA)]:
            try:
                t0 = time.time()
                m = engine.HipFrameModel.from_keras(cfg, weights, flags=flags)
                p = m.predict(frames)
                err = np.abs(p - vals[out_name]).max()
                print(f"[{name}/{mode}] max|dprob|={err:.3e} argmax_ok={np.array_equal(p.argmax(1), vals[out_name].argmax(1))} "
                      f"({time.time()-t0:.2f}s) steps={m.cost()['n_steps']}")
                if flags == 0:
                    for s in m.steps():
                        print("     step:", s["label"])
                if flags & _lib.TH_LOAD_KEEP_ALL:
                    for l in cfg["config"]["layers"]:
                        if l["class_name"] in ("InputLayer", "SpatialDropout3D", "Dropout"):
                            continue
                        want = vals[l["name"]]
                        try:
                            got = m.fetch(l["name"], 3, want.shape[1:])
                            e = np.abs(got - want).max()
                            flag = "" if e <= 2e-5 * max(1.0, np.abs(want).max()) else "   <<<<<< MISMATCH"
                            print(f"     {l['name']:28s} {str(want.shape[1:]):20s} max|d|={e:.3e} max|ref|={np.abs(want).max():.3e}{flag}")
                        except Exception as ex:
                            print(f"     {l['name']:28s} fetch failed: {ex}")
                m.close()
            except Exception:
                print(f"[{name}/{mode}] FAILED")
                traceback.print_exc()


if __name__ == "__main__":
    main()
