import sys, time, numpy as np
sys.path.insert(0, "/root/repo/timed-design_amd")
from design_utils import sampling_utils as su
from timed_hip import sampler as _s
rng = np.random.default_rng(7)
p = rng.dirichlet(np.full(20, 0.3), size=300).astype(np.float16).astype(np.float64)
def T(f, n=30):
    f(); t0 = time.perf_counter()
    for _ in range(n): f()
    return (time.perf_counter() - t0) / n * 1e3
print("np.random.rand(300000)          %.3f ms" % T(lambda: np.random.rand(300000)))
q = su.apply_temp_to_probs(p, 0.5)
print("apply_temp_to_probs             %.3f ms" % T(lambda: su.apply_temp_to_probs(p, 0.5)))
print("sample_with_multiprocessing     %.3f ms" % T(lambda: su.sample_with_multiprocessing(8, ["k"], 1000, {"k": q}, None)))
sm = _s.default_sampler(0)
r = np.random.rand(300000)
print("sm.load                         %.3f ms" % T(lambda: sm.load(q, cum_dtype=np.dtype(np.float64))))
let = "ACDEFGHIKLMNPQRSTVWY"
print("sm.draw host uniforms+let+met   %.3f ms" % T(lambda: sm.draw([0, 300], 1000, uniforms=r, letters=let, want_idx=False, want_metrics=True)))
print("sm.draw philox idx only         %.3f ms" % T(lambda: sm.draw([0, 300], 1000, rng="philox", seed=1)))
print("sm.draw philox let+met          %.3f ms" % T(lambda: sm.draw([0, 300], 1000, rng="philox", seed=1, letters=let, want_idx=False, want_metrics=True)))
print("sm.draw mt19937 let+met         %.3f ms" % T(lambda: sm.draw([0, 300], 1000, rng="mt19937", seed=1, letters=let, want_idx=False, want_metrics=True)))
d = sm.draw([0, 300], 1000, uniforms=r, letters=let, want_idx=False, want_metrics=True)
def tuples():
    text = d["letters"][:300000].tobytes().decode("ascii"); seqs = [text[i * 300:(i + 1) * 300] for i in range(1000)]
    return su._result_tuples(seqs, d["metrics"])
print("python tuples                   %.3f ms" % T(tuples))
