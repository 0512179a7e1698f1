"""Where a config-5 sampler call (1000 sequences x 300 residues) spends its time: python tools/profile_sampler_api.py   (GPU box)"""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "timed-design_amd"))
from design_utils import sampling_utils as su
from timed_hip import sampler as _s
rng = np.random.default_rng(7)
p = rng.dirichlet(np.full(20, 0.3), size=300).astype(np.float16).astype(np.float64)
def T(f, n=50):
    f(); f(); t0 = time.perf_counter()
    for _ in range(n): f()
    return (time.perf_counter() - t0) / n * 1e3
print("np.random.rand(300000)                 %.3f ms" % T(lambda: np.random.rand(300000)))
print("_legacy_rand(300000) (native replay)   %.3f ms" % T(lambda: su._legacy_rand(300000)))
q = su.apply_temp_to_probs(p, 0.5)
print("apply_temp_to_probs                    %.3f ms" % T(lambda: su.apply_temp_to_probs(p, 0.5)))
for mode in su.RNG_CHOICES:
    print("sample_with_multiprocessing rng=%-7s %.3f ms" % (mode, T(lambda: su.sample_with_multiprocessing(8, ["k"], 1000, {"k": q}, None, rng=mode, seed=1))))
sm = _s.default_sampler(0)
r = np.random.rand(300000)
let = "ACDEFGHIKLMNPQRSTVWY"
print("sm.load + sm.draw host uniforms let+met %.3f ms" % T(lambda: (sm.load(q), sm.draw([0, 300], 1000, uniforms=r, letters=let, want_idx=False, want_metrics=True))))
print("sm.run  host uniforms   let+met        %.3f ms" % T(lambda: sm.run(q, [0, 300], 1000, uniforms=r, letters=let, want_idx=False, want_metrics=True)))
print("sm.run  philox          let+met        %.3f ms" % T(lambda: sm.run(q, [0, 300], 1000, rng="philox", seed=1, letters=let, want_idx=False, want_metrics=True)))
print("sm.run  mt19937 (device) let+met       %.3f ms" % T(lambda: sm.run(q, [0, 300], 1000, rng="mt19937", seed=1, letters=let, want_idx=False, want_metrics=True)))
d = sm.run(q, [0, 300], 1000, rng="philox", seed=1, letters=let, want_idx=False, want_metrics=True)
def tuples():
    text = d["letters"][:300000].tobytes().decode("ascii"); seqs = [text[i * 300:(i + 1) * 300] for i in range(1000)]
    return su._result_tuples(seqs, d["metrics"])
print("python: decode + slices + tuples       %.3f ms" % T(tuples))
