"""Secondary legs of bench.py (rank 0, N=1, outside the timed region): host-resident / end-to-end frame rates (SURVEY.md
§8 f-1), BASELINE config 5 (the sampler, §8d) and the other BASELINE topologies.  Everything here is measurement
plumbing around the product's own entry points; the oracle appears only as the CPU baseline / checker."""
from __future__ import annotations

import json
import os
import subprocess
import sys
import tempfile
import time
from pathlib import Path

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _best(fn, reps=3):
    fn()
    best = 1e30
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        best = min(best, time.perf_counter() - t0)
    return best


# ---- host-resident frames through th_predict / th_predict_async ----------------------------------------------------
def host_resident(model, d_frames_ptr, n_total, device_fps, n=16384, batch=1024):
    """n of the bench's own frames copied to host memory, then (a) one synchronous th_predict call over all of them,
    (b) a loop of `batch`-frame calls through th_predict_async/_wait with two tickets in flight (what predict.py does);
    pageable NumPy memory and page-locked memory (th_host_alloc).  Results are checked bit for bit against the
    device-resident run of the same frames."""
    from timed_hip import _lib, engine
    import ctypes as C
    n = int(min(n, n_total))
    D, H, W, Cc = model.input_shape
    lib = _lib.load()
    host = np.empty((n, D, H, W, Cc), np.float32)
    _lib.check(lib.th_dev_download(model.device, host.ctypes.data, C.c_void_p(d_frames_ptr), host.nbytes))
    d_ref = engine.DeviceBuffer(n * model.n_classes * 4, model.device)
    old_chunk = model.chunk
    model.set_chunk(batch)
    model.predict_device(d_frames_ptr, n, d_ref.ptr)
    ref = d_ref.download((n, model.n_classes), np.float32)
    pinned, owner = engine.pinned_empty(host.shape, np.float32)
    pinned[...] = host
    out = {}

    def sync(x):
        def f():
            out["y"] = model.predict(x)
        return f

    def stream(x):
        def f():
            pend, ys = [], []
            for lo in range(0, n, batch):
                pend.append(model.predict_async(x[lo:lo + batch]))
                if len(pend) > 1:
                    ys.append(pend.pop(0).result())
            ys += [p.result() for p in pend]
            out["y"] = np.concatenate(ys)
        return f

    res = {"frames": n, "batch": batch, "dtype": "f32", "device_resident_fps": device_fps}
    for name, fn in (("th_predict_sync_pageable", sync(host)), ("th_predict_sync_pinned", sync(pinned)),
                     ("th_predict_async_pageable", stream(host)), ("th_predict_async_pinned", stream(pinned))):
        res[name + "_fps"] = n / _best(fn)
        assert np.array_equal(out["y"], ref), name + ": host-resident result differs from the device-resident one"
    res["ratio_sync_pageable"] = res["th_predict_sync_pageable_fps"] / device_fps
    res["ratio_async_pageable"] = res["th_predict_async_pageable_fps"] / device_fps
    model.set_chunk(old_chunk)
    del pinned
    owner.free()
    return res


# ---- predict.py end to end -------------------------------------------------------------------------------------
THREE = ["ALA", "CYS", "ASP", "GLU", "PHE", "GLY", "HIS", "ILE", "LYS", "LEU", "MET", "ASN", "PRO", "GLN", "ARG", "SER", "THR",
         "VAL", "TRP", "TYR"]


def make_frame_pack(stem, n, gaussian=True, seed=1):
    """synthetic frame pack (timed_hip/framepack.py layout): n frames tiled from 256 synthetic ones"""
    from timed_hip import synth
    side, c = 21, 6
    base = synth.synthetic_frames(256, seed=seed, gaussian=gaussian)
    if not gaussian:
        base = base.astype(np.uint8)
    frames = np.lib.format.open_memmap(stem + ".frames.npy", mode="w+", dtype=base.dtype, shape=(n, side, side, side, c))
    for lo in range(0, n, 256):
        frames[lo:lo + 256] = base[: min(256, n - lo)]
    frames.flush()
    del frames
    labels = np.zeros((n, 20), np.uint8)
    labels[np.arange(n), np.arange(n) % 20] = 1
    np.save(stem + ".labels.npy", labels)
    rows = [(f"p{i // 300:04d}", "A", str(i % 300 + 1), THREE[i % 20]) for i in range(n)]
    np.savetxt(stem + ".map.txt", np.array(rows), delimiter=",", fmt="%s")
    json.dump(dict(frame_dims=[side, side, side, c], voxels_as_gaussian=gaussian, n_frames=n, source="synthetic",
                   make_frame_dataset_ver=""), open(stem + ".meta.json", "w"))


def predict_py_e2e(cfg, weights, n_pack=20000, n_hdf5=2000, batch_size=500, workdir=None):
    """predict.load_dataset_and_predict wall clock (model load, dataset map, load_batch, H2D, kernels, every output file,
    FASTA extraction) on (a) a float32 frame pack, (b) a uint8 (boolean) frame pack, (c) an aposteriori-style gzip
    .hdf5 written by real h5py when the image's conda interpreter is present."""
    import warnings
    sys.path.insert(0, os.path.join(ROOT, "timed-design_amd"))
    import predict
    from timed_hip import pack
    res = {"batch_size": batch_size, "frames_per_call": 1024}
    with tempfile.TemporaryDirectory(dir=workdir) as td:
        mp = Path(td) / "TIMED.pack"
        mp.write_bytes(pack.keras_to_pack(cfg, weights))

        def run(dataset, tag, n):
            out = Path(td) / f"out_{tag}"
            out.mkdir()
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                t0 = time.perf_counter()
                predict.load_dataset_and_predict([mp], dataset, batch_size=batch_size, dataset_map_path=out / "datasetmap.txt",
                                                 path_to_output=out)
                dt = time.perf_counter() - t0
            rows = sum(1 for _ in open(out / "TIMED.csv"))
            assert rows == n, f"{tag}: {rows} rows written for {n} frames"
            res[f"{tag}_frames"] = n
            res[f"{tag}_fps"] = n / dt
            res[f"{tag}_s"] = dt

        if n_pack > 0:
            stem = os.path.join(td, "synth_f32")
            make_frame_pack(stem, n_pack, gaussian=True)
            run(stem + ".framepack", "predict_py_framepack_f32", n_pack)
            for s in (".frames.npy",):
                os.remove(stem + s)
            stem = os.path.join(td, "synth_u8")
            make_frame_pack(stem, n_pack, gaussian=False)
            run(stem + ".framepack", "predict_py_framepack_u8", n_pack)
            os.remove(stem + ".frames.npy")
        # BASELINE config 1 (the reference's own CPU-runnable case: predict.py on the 1ubq structure of its tests
        # directory, tests/testing_files/1ubq.pdb1.gz): here the structure file is voxelised on the GPU (row f-4, parity
        # unpinned against aposteriori) and predicted with a 5-channel TIMED-synth; beside it the CPU oracle on the very
        # same 76 frames (the "plumbing baseline": NumPy port, not TensorFlow)
        ubq = os.path.join(ROOT, "tests", "golden", "1ubq.pdb1.gz")
        if os.path.exists(ubq):
            from timed_hip import synth, voxeliser
            cfg5, w5 = synth.timed_synth(20, in_channels=5)
            mp5 = Path(td) / "TIMED5.pack"
            mp5.write_bytes(pack.keras_to_pack(cfg5, w5))
            out = Path(td) / "out_1ubq"
            out.mkdir()
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                t0 = time.perf_counter()
                predict.load_dataset_and_predict([mp5], ubq, batch_size=12, dataset_map_path=out / "datasetmap.txt", path_to_output=out)
                dt = time.perf_counter() - t0
            frames, _labels, _flat = voxeliser.voxelise_pdb(ubq)
            from oracle import cnn_oracle
            t0 = time.perf_counter()
            ref = cnn_oracle.forward(cfg5, w5, frames)
            dt_cpu = time.perf_counter() - t0
            got = np.loadtxt(out / "TIMED5.csv", delimiter=",")
            assert got.shape == ref.shape and np.abs(got - ref.astype(np.float16).astype(np.float64)).max() < 2e-3
            res["config1_1ubq_pdb_frames"] = int(frames.shape[0])
            res["config1_predict_py_from_pdb_s"] = dt
            res["config1_cpu_oracle_forward_s"] = dt_cpu
        conda = "/opt/conda/bin/python3.9"
        if n_hdf5 > 0 and os.path.exists(conda):
            h5 = os.path.join(td, "frames.hdf5")
            n_pdb = max(1, n_hdf5 // 100)
            r = subprocess.run([conda, os.path.join(ROOT, "tools", "make_synthetic_hdf5.py"), h5, str(n_pdb), "100"],
                               capture_output=True, text=True)
            if r.returncode == 0:
                res["hdf5_file_MB"] = os.path.getsize(h5) / 1e6
                run(h5, "predict_py_hdf5_gzip_f64", n_pdb * 100)
            else:
                res["predict_py_hdf5_gzip_f64_fps"] = None
                res["hdf5_note"] = "h5py writer failed: " + r.stderr[-200:]
        elif n_hdf5 > 0:
            res["predict_py_hdf5_gzip_f64_fps"] = None
            res["hdf5_note"] = "no h5py in this image to write the synthetic .hdf5"
    return res


# ---- BASELINE config 5: the sampler ----------------------------------------------------------------------------------
def sampler_config5(device=0, n_res=300, n_samples=1000, temps=(0.1, 0.5, 1.0), seed=42):
    """1 000 sequences x 300 residues per temperature from float16-rounded Dirichlet(0.3) rows:
      api     design_utils.sampling_utils: apply_temp_to_probs + sample_with_multiprocessing (uniforms from NumPy's legacy
              generator, letters AND the four sequence metrics per sequence, Python tuples built) — what sample.py runs;
      kernel  resident sampler, device Philox, indices only (th_sampler_load + th_sampler_draw);
      cpu     the reference's loop restated with NumPy on this host (oracle/sampler_oracle.py: one rand + cumsum + argmax
              per sample, array rebuilt from the list of lists — sampling_utils.py:123-128), without and with the
              per-sequence metrics.
    Indices are checked bit for bit against the oracle for the same uniforms."""
    from oracle import sampler_oracle as so
    sys.path.insert(0, os.path.join(ROOT, "timed-design_amd"))
    from design_utils import analyse_utils, sampling_utils as su
    from timed_hip import sampler
    rng = np.random.default_rng(7)
    p = rng.dirichlet(np.full(20, 0.3), size=n_res).astype(np.float16).astype(np.float64)
    letters = np.array(list("ACDEFGHIKLMNPQRSTVWY"))
    draws = n_res * n_samples
    out = {"n_residues": n_res, "n_samples": n_samples, "draws_per_run": draws, "metrics_source": analyse_utils.METRICS_SOURCE,
           "temperatures": {}}
    sm = sampler.Sampler(device)
    for t in temps:
        q_ref = so.apply_temp(p, t) if t != 1 else p

        def api():
            np.random.seed(seed)
            q = su.apply_temp_to_probs(p, t) if t != 1 else p
            api.out = su.sample_with_multiprocessing(8, ["k"], n_samples, {"k": q}, None)

        def kernel():
            sm.load(p, t)
            kernel.out = sm.draw([0, n_res], n_samples, rng="philox", seed=seed)

        t_api, t_kernel = _best(api, 5), _best(kernel, 10)
        r = so.legacy_uniforms(seed, draws).reshape(n_samples, n_res)
        want = ["".join(letters[i]) for i in so.choice_indices(q_ref, r)]
        exact = [s[0] for s in api.out["k"]] == want
        ql = [list(row) for row in q_ref]

        def cpu(with_metrics):
            np.random.seed(seed)
            seqs = []
            for _ in range(n_samples):
                idx = so.choice_indices(np.array(ql), np.random.rand(n_res))
                s = "".join(letters[idx])
                seqs.append((s, *analyse_utils.calculate_seq_metrics(s)) if with_metrics else s)
            return seqs
        t0 = time.perf_counter(); cpu(False); t_cpu = time.perf_counter() - t0
        t0 = time.perf_counter(); cpu(True); t_cpu_m = time.perf_counter() - t0
        out["temperatures"][str(t)] = {
            "api_ms": t_api * 1e3, "api_sequences_per_s": n_samples / t_api, "api_draws_per_s": draws / t_api,
            "kernel_ms": t_kernel * 1e3, "kernel_draws_per_s": draws / t_kernel,
            "cpu_numpy_ms": t_cpu * 1e3, "cpu_numpy_sequences_per_s": n_samples / t_cpu,
            "cpu_numpy_with_metrics_ms": t_cpu_m * 1e3, "indices_bit_exact_vs_oracle": bool(exact)}
        assert exact, f"sampler indices differ from the oracle at T={t}"
    sm.close()
    out["cpu_cores"] = 1
    out["note"] = "launch-latency bound (a few microseconds of device work per run): no roofline fraction is meaningful"
    return out


# ---- other BASELINE topologies ---------------------------------------------------------------------------------------
def topology_rate(name, device, d_frames_ptr, n, chunk, steps=2):
    """device-resident frames/s of another BASELINE topology on the same frames (config 3: densecpd, config 4's model:
    timed_rotamer), with the per-kernel table from HIP events"""
    from timed_hip import _lib, engine, synth
    cfg, weights = synth.TOPOLOGIES[name]()
    model = engine.HipFrameModel.from_keras(cfg, weights, device=device, name=name)
    model.set_chunk(chunk)
    d_probs = engine.DeviceBuffer(n * model.n_classes * 4, device)
    lib = _lib.load()
    model.profile(1)
    model.predict_device(d_frames_ptr, n, d_probs.ptr)
    _lib.check(lib.th_dev_sync(device))
    table = [s for s in model.steps() if s["launches"]]
    model.profile(0)
    t0 = time.perf_counter()
    for _ in range(steps):
        model.predict_device(d_frames_ptr, n, d_probs.ptr)
    _lib.check(lib.th_dev_sync(device))
    dt = (time.perf_counter() - t0) / steps
    cost = model.cost()
    probe = d_probs.download((min(n, 64), model.n_classes), np.float32)
    assert np.all(np.isfinite(probe)) and np.allclose(probe.sum(1), 1.0, atol=1e-4)
    tot = sum(s["ms"] for s in table)
    res = {"topology": name, "frames": n, "n_classes": model.n_classes, "frames_per_s": n / dt,
           "algo_mflop_per_frame": cost["algo_flops"] / 1e6, "model_tflops": n / dt * cost["algo_flops"] / 1e12,
           "kernels": [{"label": s["label"], "share": s["ms"] / tot,
                        "tflops_algo": (s["flops"] * n / (s["ms"] * 1e-3) / 1e12) if s["ms"] and s["flops"] else 0.0,
                        "GBps_algo": (s["bytes"] * n / (s["ms"] * 1e-3) / 1e9) if s["ms"] else 0.0} for s in table]}
    model.close()
    d_probs.free()
    return res
