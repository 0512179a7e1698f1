#!/usr/bin/env python
"""the NumPy restatement of the reference sampler loop under multiprocessing.Pool, on config 5's inputs
(Dirichlet(0.3) rows, seed 7, float16-rounded).  The reference's Pool runs ONE task per PDB key (sampling_utils.py:181-190)
and config 5 is a single 300-residue key — one task however many workers — so two shapes are timed: that one, and the same
samples cut into one task per usable core (the most a Pool can give).  Run as a process of its own by tools/bench_legs.py.

    python tools/sampler_pool_baseline.py N_RES N_SAMPLES SEED WORKERS   -> one JSON line
"""
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "timed-design_amd"))


def task(args):
    """`count` samples of the reference's per-sample loop (sampling_utils.py:123-128: array rebuilt from the list of lists)"""
    rows, count, seed, with_metrics = args
    from oracle import sampler_oracle as so
    from design_utils import analyse_utils
    letters = np.array(list("ACDEFGHIKLMNPQRSTVWY"))
    np.random.seed(seed)
    n_res = len(rows)
    done = 0
    for _ in range(count):
        idx = so.choice_indices(np.array(rows), np.random.rand(n_res))
        s = "".join(letters[idx])
        if with_metrics:
            analyse_utils.calculate_seq_metrics(s)
        done += 1
    return done


def main():
    n_res, n_samples, seed, workers = (int(x) for x in sys.argv[1:5])
    p = np.random.default_rng(7).dirichlet(np.full(20, 0.3), size=n_res).astype(np.float16).astype(np.float64)
    rows = [list(r) for r in p]
    res = {"workers": workers, "host_cores_visible": os.cpu_count(), "n_residues": n_res, "n_samples": n_samples}
    with mp.get_context("fork").Pool(workers) as pool:
        pool.map(task, [(rows[:4], 1, 0, True)] * workers)            # workers started, imports done
        for metrics in (False, True):
            tag = "_with_metrics" if metrics else ""
            t0 = time.perf_counter()
            pool.map(task, [(rows, n_samples, seed, metrics)])
            dt = time.perf_counter() - t0
            res[f"one_task_per_key_ms{tag}"] = dt * 1e3
            res[f"one_task_per_key_sequences_per_s{tag}"] = n_samples / dt
            per = [n_samples // workers + (1 if k < n_samples % workers else 0) for k in range(workers)]
            t0 = time.perf_counter()
            pool.map(task, [(rows, c, seed + k, metrics) for k, c in enumerate(per) if c])
            dt = time.perf_counter() - t0
            res[f"split_over_workers_ms{tag}"] = dt * 1e3
            res[f"split_over_workers_sequences_per_s{tag}"] = n_samples / dt
    print(json.dumps(res))


if __name__ == "__main__":
    main()
