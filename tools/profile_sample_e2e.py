#!/usr/bin/env python
"""sample.py end to end on BASELINE config 5 (1 000 sequences x 300 residues): wall time and where it goes."""
import argparse, cProfile, io, os, pstats, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "timed-design_amd"))
import sample
rng = np.random.default_rng(7)
p = rng.dirichlet(np.full(20, 0.3), size=300).astype(np.float16)
with tempfile.TemporaryDirectory() as td:
    os.chdir(td)
    np.savetxt("TIMED.csv", p, delimiter=",")
    open("TIMED.txt", "w").write("ignore_uncommon False\ninclude_pdbs\n##########\n1abcA 300\n")
    for T in (1.0, 0.5, 0.1):
        args = argparse.Namespace(path_to_pred_matrix="TIMED.csv", path_to_datasetmap="TIMED.txt", predict_rotamers=False,
                                  sample_n=1000, save_as="all", workers=8, temperature=T, support_old_datasetmap=False, seed=42)
        sample.main_sample(args)                      # warm (library load, table builds)
        t0 = time.perf_counter(); sample.main_sample(args); dt = time.perf_counter() - t0
        print(f"T={T}: main_sample {dt * 1e3:.1f} ms for 1000 sequences x 300 residues ({300e3 / dt / 1e6:.2f} M draws/s end to end)")
    pr = cProfile.Profile(); pr.enable(); sample.main_sample(args); pr.disable()
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(16); print(s.getvalue()[-2800:])
