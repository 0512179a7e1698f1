"""Where the wall clock of `predict.py --predict_rotamers` goes (BASELINE config 4's per-GPU share: 125 k uint8 frames, the 338-class
model, a 1 GB _rot.csv): the bench's rotamer leg (tools/bench_legs.predict_py_e2e with n_rotamer) twice with TIMED_PIPELINE_TRACE on.

    python tools/trace_rotamer_e2e.py [frames]      (TRACE_PROFILE=1: cProfile of every load_dataset_and_predict call)
"""
import os, sys, json
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "timed-design_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
os.environ["TIMED_PIPELINE_TRACE"] = "1"
from timed_hip import synth
import bench_legs as b
import predict

if os.environ.get("TRACE_PROFILE"):
    import cProfile, pstats, io
    inner = predict.load_dataset_and_predict

    def profiled(*a, **kw):
        pr = cProfile.Profile()
        try:
            return pr.runcall(inner, *a, **kw)
        finally:
            s = io.StringIO()
            pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(25)
            print("\n".join(l for l in s.getvalue().splitlines() if l.strip())[:6000], file=sys.stderr)
    predict.load_dataset_and_predict = profiled

cfg, w = synth.timed_synth(20)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 125000
for _ in range(2):
    r = b.predict_py_e2e(cfg, w, n_pack=0, n_hdf5=0, n_rotamer=n)
    print(json.dumps({k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items() if "rotamer" in k}))
