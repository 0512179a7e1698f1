"""Where the wall clock of predict.py goes: runs the bench's predict.py legs (tools/bench_legs.predict_py_e2e) twice with the
pipeline trace on (TIMED_PIPELINE_TRACE; add TH_H5_TRACE=1 for the decoder's own breakdown).  With TRACE_PROFILE=1 every
load_dataset_and_predict call also runs under cProfile and prints its 18 most expensive entries by cumulative time.

    python tools/trace_predict_e2e.py [frames in the packs] [frames in the gzip .hdf5] [frames of the rotamer leg]
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
            pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(18)
            print("\n".join(l for l in s.getvalue().splitlines() if l.strip())[:4000], file=sys.stderr)
    predict.load_dataset_and_predict = profiled

cfg, w = synth.timed_synth(20)
n_pack = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
n_hdf5 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
n_rot = int(sys.argv[3]) if len(sys.argv) > 3 else 0          # bench.py's order: packs, rotamer leg, config 1, then the .hdf5 calls
for _ in range(int(os.environ.get("TRACE_REPEATS", "2"))):
    r = b.predict_py_e2e(cfg, w, n_pack=n_pack, n_hdf5=n_hdf5, batch_size=500, n_rotamer=n_rot)
    print(json.dumps({k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items()}))
