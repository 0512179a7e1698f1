"""Secondary legs of bench.py (rank 0, N=1, outside the timed region): host-resident / end-to-end frame rates (SURVEY.md
§8 f-1), BASELINE config 5 (the sampler, §8d) and the other BASELINE topologies.  Everything here is measurement
plumbing around the product's own entry points; the oracle appears only as the CPU baseline / checker."""
from __future__ import annotations

import json
import os
import shutil
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
    # the same model on DEVICE-RESIDENT uint8 frames (boolean aposteriori datasets, voxels_as_gaussian=False: reference
    # utils.py:518-521): the first-layer kernel reads the caller's bytes directly (a quarter of the float32 frames' HBM traffic);
    # 4096 distinct frames tiled to 32 768 on the device, the chunk the headline run uses
    try:
        from timed_hip import synth
        base = synth.synthetic_frames(4096, side=D, channels=Cc, seed=77, gaussian=False)
        n8 = 8 * len(base)
        d8 = engine.DeviceBuffer(n8 * base[0].size, model.device)
        for k in range(8):
            _lib.check(lib.th_dev_upload(model.device, C.c_void_p(d8.ptr + k * base.nbytes), base.ctypes.data, base.nbytes))
        d_o8 = engine.DeviceBuffer(n8 * model.n_classes * 4, model.device)
        model.predict_device(d8.ptr, n8, d_o8.ptr, dtype=_lib.TH_U8)

        def run8():
            model.predict_device(d8.ptr, n8, d_o8.ptr, dtype=_lib.TH_U8)
        res["device_resident_u8_frames"] = n8
        res["device_resident_u8_fps"] = n8 / _best(run8, 3)
        y8 = d_o8.download((n8, model.n_classes), np.float32)
        assert np.isfinite(y8).all() and np.abs(y8.sum(1) - 1).max() < 1e-4 and np.array_equal(y8[:4096], y8[4096:8192])
        d8.free(); d_o8.free()
    except Exception as e:            # a leg never takes the bench line down
        res["device_resident_u8_fps"] = None
        res["device_resident_u8_note"] = repr(e)[:200]
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


def predict_py_e2e(cfg, weights, n_pack=20000, n_hdf5=2000, batch_size=500, workdir=None, n_rotamer=0):
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
            res["predict_py_framepack_f32_pcie_bytes_per_frame"] = os.path.getsize(stem + ".frames.npy") / n_pack
            # the same pack with the sparse transport files (framepack.sparsify): bitmap + stored values cross PCIe, the device
            # rebuilds the dense frames (th_predict_sparse_async); the output files are the dense run's, byte for byte (tests)
            from timed_hip import framepack
            t0 = time.perf_counter()
            dense_b, sparse_b = framepack.sparsify(stem)
            res["sparsify_s"] = time.perf_counter() - t0
            run(stem + ".framepack", "predict_py_framepack_sparse_f32", n_pack)
            res["predict_py_framepack_sparse_f32_pcie_bytes_per_frame"] = sparse_b / n_pack
            for s in (".frames.npy",) + tuple(framepack.SPARSE_SUFFIXES):
                os.remove(stem + s)
            stem = os.path.join(td, "synth_u8")
            make_frame_pack(stem, n_pack, gaussian=False)
            run(stem + ".framepack", "predict_py_framepack_u8", n_pack)
            os.remove(stem + ".frames.npy")
        if n_rotamer > 0:
            # BASELINE config 4's per-GPU share (1 M frames / 8 GPUs = 125 k) through `predict.py --predict_rotamers` on
            # ONE GPU, every file written: the 338-class model, the full-precision _rot.csv (125 k x 338 '%.18e' values,
            # ~1 GB of text), the codec one-hot <model>.csv, labels, map, FASTA.  uint8 (boolean) frames keep the
            # synthetic pack at 7 GB; in the 8-GPU run every rank formats and writes its own shard like this.
            from timed_hip import synth
            cfg_r, w_r = synth.TOPOLOGIES["timed_rotamer"]()
            mpr = Path(td) / "TIMED_rotamer.pack"
            mpr.write_bytes(pack.keras_to_pack(cfg_r, w_r))
            stem = os.path.join(td, "synth_rot_u8")
            t0 = time.perf_counter()
            make_frame_pack(stem, n_rotamer, gaussian=False)
            res["predict_py_rotamer_pack_write_s"] = time.perf_counter() - t0
            out = Path(td) / "out_rotamer"
            out.mkdir()
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                t0 = time.perf_counter()
                predict.load_dataset_and_predict([mpr], stem + ".framepack", batch_size=batch_size, dataset_map_path=out / "datasetmap.txt",
                                                 predict_rotamers=True, path_to_output=out)
                dt = time.perf_counter() - t0
            assert sum(1 for _ in open(out / "TIMED_rotamer.csv")) == n_rotamer
            res["predict_py_rotamer_frames"] = n_rotamer
            res["predict_py_rotamer_fps"] = n_rotamer / dt
            res["predict_py_rotamer_s"] = dt
            res["predict_py_rotamer_rot_csv_MB"] = os.path.getsize(out / "TIMED_rotamer_rot.csv") / 1e6
            os.remove(stem + ".frames.npy")
            for f in out.iterdir():
                f.unlink()
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
                # twice: the first call of a process also allocates the GPU inflater's scratch (a 9 GB token arena for 4 096-frame
                # calls) and parses the file's group tables; both numbers are reported, `_fps` is the second call
                run(h5, "predict_py_hdf5_gzip_f64_first_call", n_pdb * 100)
                shutil.rmtree(Path(td) / "out_predict_py_hdf5_gzip_f64_first_call")
                run(h5, "predict_py_hdf5_gzip_f64", n_pdb * 100)
                # and as the first thing a fresh process does — what every `python predict.py` invocation is: HIP start-up and
                # code-object load, model load + guard, decoder scratch, first parse of the group tables inside the timed region
                try:
                    rc = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_h5_cold.py"), h5, str(n_pdb * 100), str(batch_size)],
                                        capture_output=True, text=True, timeout=300)
                    cold = json.loads(rc.stdout.strip().splitlines()[-1])
                    res["predict_py_hdf5_cold_process_fps"] = cold["cold_fps"]
                    res["predict_py_hdf5_cold_process_s"] = cold["cold_s"]
                    res["predict_py_hdf5_cold_process_second_call_fps"] = cold["warm_fps"]
                except Exception as e:            # a leg never takes the bench line down
                    res["predict_py_hdf5_cold_process_fps"] = None
                    res["hdf5_cold_note"] = repr(e)[:200]
            else:
                res["predict_py_hdf5_gzip_f64_fps"] = None
                res["hdf5_note"] = "h5py writer failed: " + r.stderr[-200:]
        elif n_hdf5 > 0:
            res["predict_py_hdf5_gzip_f64_fps"] = None
            res["hdf5_note"] = "no h5py in this image to write the synthetic .hdf5"
    from design_utils import utils as du
    du.release_device_memory()          # what predict.py's CLI does at its end: pooled batch buffers, decoder scratch, staging rings
    return res


def _sampler_pool_baseline(n_res, n_samples, seed, usable):
    """BASELINE.md B3 under multiprocessing.Pool: tools/sampler_pool_baseline.py in a child process of its own (no HIP
    context to fork, bounded by a timeout)."""
    try:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "sampler_pool_baseline.py"), str(n_res), str(n_samples),
                            str(seed), str(usable)], capture_output=True, text=True, timeout=180)
        if r.returncode != 0:
            return {"error": r.stderr[-300:]}
        return json.loads(r.stdout.strip().splitlines()[-1])
    except Exception as e:        # a baseline leg never takes the bench line down
        return {"error": repr(e)}


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

        def api_philox():       # sample.py --rng philox: uniforms drawn on the device (rocRAND Philox4x32-10), nothing generated by the host
            q = su.apply_temp_to_probs(p, t) if t != 1 else p
            api_philox.out = su.sample_with_multiprocessing(8, ["k"], n_samples, {"k": q}, None, rng="philox", seed=seed)

        def kernel():
            sm.load(p, t)
            kernel.out = sm.draw([0, n_res], n_samples, rng="philox", seed=seed)

        t_api, t_kernel, t_api_ph = _best(api, 5), _best(kernel, 10), _best(api_philox, 5)
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
            "api_philox_ms": t_api_ph * 1e3, "api_philox_sequences_per_s": n_samples / t_api_ph,
            "kernel_ms": t_kernel * 1e3, "kernel_draws_per_s": draws / t_kernel,
            "cpu_numpy_ms": t_cpu * 1e3, "cpu_numpy_sequences_per_s": n_samples / t_cpu,
            "cpu_numpy_with_metrics_ms": t_cpu_m * 1e3, "indices_bit_exact_vs_oracle": bool(exact)}
        assert exact, f"sampler indices differ from the oracle at T={t}"
    # where an API call's time goes (T = 0.5): the reference's contract fixes two host-side costs — the uniforms come from NumPy's
    # global legacy generator (np.random.rand: MT19937 on one core) and the result is a Python list of (str, float, ...) tuples;
    # what is left is the GPU part (upload of rows + uniforms, temper / draw / metrics kernels, download of letters + metrics)
    q = su.apply_temp_to_probs(p, 0.5)
    r = np.random.rand(draws)
    let = "".join(letters)
    t_rand = _best(lambda: su._legacy_rand(draws), 5)        # np.random.rand's values and state, replayed natively (th_mt19937_rand)
    t_words = _best(lambda: su._legacy_words(draws, out=sm.uniform_buffer(2 * draws, np.uint32)), 5)   # the recurrence only, into page-locked memory
    t_np = _best(lambda: np.random.rand(draws), 5)
    w = su._legacy_words(draws)
    t_gpu = _best(lambda: sm.run(q, [0, n_res], n_samples, uniforms=w, rng="mt_words", letters=let, want_idx=False, want_metrics=True), 10)
    t_gpu_ph = _best(lambda: sm.run(q, [0, n_res], n_samples, rng="philox", seed=seed, letters=let, want_idx=False, want_metrics=True), 10)
    d = sm.draw([0, n_res], n_samples, uniforms=r, letters=let, want_idx=False, want_metrics=True)

    def tuples():          # as design_utils/sampling_utils._sample_keys builds them
        text = d["letters"][:n_samples * n_res].tobytes().decode("ascii")
        seqs = [text[i * n_res:(i + 1) * n_res] for i in range(n_samples)]
        return su._result_tuples(seqs, d["metrics"])
    t_py = _best(tuples, 5)
    out["api_breakdown_ms"] = {"mt19937_words_on_host": t_words * 1e3, "legacy_rand_replay_full": t_rand * 1e3, "np_random_rand_itself": t_np * 1e3,
                               "gpu_one_submission_host_words": t_gpu * 1e3, "gpu_one_submission_philox": t_gpu_ph * 1e3, "python_result_tuples": t_py * 1e3}
    sm.close()
    out["cpu_cores"] = 1
    from timed_hip import _lib
    out["cpu_pool"] = _sampler_pool_baseline(n_res, n_samples, seed, max(1, int(_lib.load().th_host_cpus())))
    out["note"] = "launch-latency bound (a few microseconds of device work per run): no roofline fraction is meaningful"
    return out


# ---- HBM traffic of every kernel, measured in THIS run -------------------------------------------------------------
def pmc_traffic_inrun(topologies, chunk, timeout=240):
    """Two rocprofv3 child runs of tools/pmc_child.py (`--kernel-trace --pmc FETCH_SIZE`, then `--pmc WRITE_SIZE`: separate
    passes, as /opt/skills/guides/MI355X_MICROARCH.md prescribes) on the library this process has loaded: one chunk of every
    topology after a warm-up pass, every plan step dispatched once in plan order.  FETCH_SIZE x2 (the guide's gfx950 correction; calibrated on
    k_convert_frames), WRITE_SIZE x1, both counted in KiB.  Returns {topology: {"steps": {label: bytes}, "model": bytes}}
    per launch of `chunk` frames, or {"error": ...} — nothing is looked up by kernel name or duration in committed files."""
    import re
    import shutil
    import sqlite3
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        return {"error": "rocprofv3 not found"}
    child = os.path.join(ROOT, "tools", "pmc_child.py")
    env = dict(os.environ, TMPDIR="/tmp")
    result = {}
    labels = None
    with tempfile.TemporaryDirectory(dir="/tmp") as td:
        for counter, corr in (("FETCH_SIZE", 2.0), ("WRITE_SIZE", 1.0)):
            out_dir = os.path.join(td, counter)
            cmd = [rocprof, "--kernel-trace", "--pmc", counter, "-d", out_dir, "-o", counter, "--",
                   sys.executable, child, str(chunk), *topologies]
            try:
                r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, cwd="/tmp", env=env)
            except subprocess.TimeoutExpired:
                return {"error": f"rocprofv3 --pmc {counter} timed out"}
            line = next((l for l in r.stdout.splitlines() if l.startswith("{")), None)
            dbs = [os.path.join(dp, f) for dp, _dn, fn in os.walk(out_dir) for f in fn if f.endswith(".db")]
            if r.returncode != 0 or line is None or not dbs:
                return {"error": f"rocprofv3 --pmc {counter} failed (rc {r.returncode}): " + (r.stderr or r.stdout)[-300:]}
            labels = json.loads(line)
            c = sqlite3.connect(dbs[0])
            disp = c.execute("select d.event_id, coalesce(s.display_name, s.kernel_name) from rocpd_kernel_dispatch d "
                             "join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start").fetchall()
            vals = dict(c.execute("select e.event_id, sum(e.value) from rocpd_pmc_event e join rocpd_info_pmc p on e.pmc_id = p.id "
                                  "where p.name = ? group by e.event_id", (counter,)).fetchall())
            c.close()
            # split the dispatch stream on the k_synth_frames markers: segment t belongs to topologies[t]
            segments, cur = [], None
            for ev, name in disp:
                if "k_synth_frames" in name:
                    cur = []
                    segments.append(cur)
                elif cur is not None and "k_clip01" not in name:        # (k_clip01 is the second half of the marker itself)
                    cur.append((re.sub(r"\(anonymous namespace\)::", "", name), vals.get(ev, 0.0) * 1024.0 * corr))
            if len(segments) != 2 * len(topologies):
                return {"error": f"expected {2 * len(topologies)} marker launches in the {counter} trace, found {len(segments)}"}
            for topo, seg in zip(topologies, segments[1::2]):          # [warm-up, measured] per topology
                rec = result.setdefault(topo, {"steps": {}, "model": 0.0})
                rec["model"] += sum(b for _n, b in seg)
                pos = 0
                for label in labels[topo]:                     # greedy in-order match of the tagged steps
                    m = re.search(r"\[(k_[a-z0-9_]+)(<[^>]*>)?\]", label)
                    if not m:
                        continue
                    want = m.group(1) + (m.group(2) or "")
                    for j in range(pos, len(seg)):
                        if want.replace(" ", "") in seg[j][0].replace(" ", ""):
                            rec["steps"][label] = rec["steps"].get(label, 0.0) + seg[j][1]
                            pos = j + 1
                            break
    for topo in result:
        result[topo]["chunk"] = chunk
    result["source"] = ("measured in this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE child passes of "
                        "tools/pmc_child.py on the loaded library (FETCH x2 gfx950 correction, WRITE x1), one chunk per topology")
    return result


PEAK_FP32_MFMA_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md: dense fp32-input MFMA = the fp32 vector peak
PEAK_BF16_MFMA_TFLOPS = 2500.0     # same guide: dense bf16 MFMA (the 5 PF headline figure includes 2:1 sparsity)


def step_pipe(step):
    """(matrix pipe, its dense peak in TFLOP/s, FLOPs per frame that pipe executes for the step's OWN arithmetic).  A step whose
    label says bf16x3 runs every fp32 multiply-add as six bf16 piece products (csrc/conv_wino.hip, k_wino_gemm_b3): it is priced
    on the bf16 pipe with six times its fp32-equivalent FLOPs — never with the fp32 count against the fp32 peak (which it exceeds)."""
    if "bf16x3" in step["label"]:
        # k_conv_first_b3 runs 8 of its 9 (dz, dy) taps as six bf16 products and tap 8 as fp32-input MFMAs, which cost 16 bf16-pipe
        # FLOPs per FLOP (the fp32 form runs at 1/16 of the bf16 rate on the same pipe): 8/9 x 6 + 1/9 x 16
        mult = (8.0 * 6.0 + 16.0) / 9.0 if "k_conv_first_b3" in step["label"] else 6.0
        return "bf16", PEAK_BF16_MFMA_TFLOPS, mult * step["flops"]
    return "fp32", PEAK_FP32_MFMA_TFLOPS, step["flops"]


def pipe_time_frac(steps, fps):
    """fraction of the wall time the matrix pipes would need at their dense peaks for what the plan's kernels compute:
    frames/s x sum over steps of (own FLOPs on the step's pipe / that pipe's peak)"""
    return fps * sum(step_pipe(s)[2] / (step_pipe(s)[1] * 1e12) for s in steps if s["flops"])


def step_roofline(step, frames, traffic_bytes=None, chunk=None):
    """roofline record of one plan step from its HIP-event time over `frames` frames: bound 'mfma' for convolutions
    whose arithmetic intensity is above the ridge (157.3 TFLOP/s / 8 TB/s = 19.7 FLOP/B), 'hbm' otherwise"""
    ms = step["ms"]
    if not ms:
        return None
    pipe, peak, pflops = step_pipe(step)
    tf = pflops * frames / (ms * 1e-3) / 1e12
    gbs = step["bytes"] * frames / (ms * 1e-3) / 1e9
    intensity = step["flops"] / step["bytes"] if step["bytes"] else float("inf")
    if intensity >= 157.3e12 / 8.0e12 and step["flops"]:
        rec = {"bound": "mfma", "achieved": tf, "peak": peak, "unit": "TFLOP/s", "frac": tf / peak, "pipe": pipe}
        if pipe != "fp32":
            rec["fp32_equiv_tflops"] = step["flops"] * frames / (ms * 1e-3) / 1e12
    else:
        rec = {"bound": "hbm", "achieved": gbs, "peak": 8000.0, "unit": "GB/s", "frac": gbs / 8000.0}
    rec.update(kernel=step["label"], flop_per_byte=intensity if step["bytes"] else None,
               avg_launch_ms=ms / max(1, step["launches"]), launches=step["launches"], traffic=traffic_bytes)
    if traffic_bytes is not None and chunk:
        rec["traffic_over_algorithmic"] = traffic_bytes / (step["bytes"] * chunk) if step["bytes"] else None
    return rec


# ---- other BASELINE topologies ---------------------------------------------------------------------------------------
def topology_rate(name, device, d_frames_ptr, n, chunk, steps=2, traffic=None, cpu_baseline=None, env=None):
    """device-resident frames/s of another BASELINE topology on the same frames (config 3: densecpd, config 4's model:
    timed_rotamer) with its own roofline records: the whole model against the fp32-MFMA peak, the dominant kernel
    (largest share of device time) and every kernel's bound / fraction / measured HBM traffic; all n output rows are
    checked.  ``traffic``: this topology's record from pmc_traffic_inrun; ``cpu_baseline``: callable(cfg, weights, name)."""
    from timed_hip import _lib, engine, synth
    cfg, weights = synth.TOPOLOGIES[name]()
    # ``env``: TH_* knobs of a non-default plan ({"TH_WINO_SPLIT": "0"}): the library reads them once, when the model is loaded
    keep = {k: os.environ.get(k) for k in (env or {})}
    os.environ.update(env or {})
    try:
        model = engine.HipFrameModel.from_keras(cfg, weights, device=device, name=name)
    finally:
        for k, v in keep.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
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
    rows = d_probs.download((n, model.n_classes), np.float32)
    bad = int(np.count_nonzero(~np.isfinite(rows).all(1) | (np.abs(rows.sum(1, dtype=np.float64) - 1.0) > 1e-4)))
    assert bad == 0, f"{name}: {bad} of {n} output rows are not probability vectors"
    tot = sum(s["ms"] for s in table)
    per_step = (traffic or {}).get("steps", {})
    D, H, W, Cc = model.input_shape
    algo_bytes = D * H * W * Cc * 4 + 4 * model.n_classes
    fps = n / dt
    dom = max(table, key=lambda s: s["ms"])
    kflops = sum(s["flops"] for s in model.steps())        # what the kernels compute (Winograd layers: their own, fewer, FLOPs)
    res = {"topology": name, "frames": n, "chunk": chunk, "n_classes": model.n_classes, "frames_per_s": fps, "rows_verified": n,
           "algo_mflop_per_frame": cost["algo_flops"] / 1e6, "kernel_mflop_per_frame": kflops / 1e6, "model_tflops": fps * kflops / 1e12,
           "model_direct_equiv_tflops": fps * cost["algo_flops"] / 1e12,
           # the whole model against its matrix pipes: the share of the wall time they would need at their dense peaks (fp32-input
           # MFMA 157.3 TFLOP/s; the bf16x3-split GEMMs on the bf16 pipe, 2500 TFLOP/s, with six products per multiply-add);
           # `achieved` / `peak` restate it in fp32-pipe terms for plans without split steps
           "model_roofline": {"bound": "mfma", "achieved": pipe_time_frac(model.steps(), fps) * 157.3, "peak": 157.3, "unit": "TFLOP/s",
                              "frac": pipe_time_frac(model.steps(), fps), "fp32_equiv_tflops": fps * kflops / 1e12,
                              "traffic": (traffic or {}).get("model"), "traffic_frames": (traffic or {}).get("chunk"),
                              "algorithmic_bytes": algo_bytes * ((traffic or {}).get("chunk") or 0) or None},
           "hbm": {"algo_bytes_per_frame": algo_bytes, "achieved_GBps": fps * algo_bytes / 1e9, "frac": fps * algo_bytes / 1e9 / 8000.0},
           "roofline": dict(step_roofline(dom, n, per_step.get(dom["label"]), (traffic or {}).get("chunk")),
                            share_of_device_time=dom["ms"] / tot),
           "kernels": [dict(step_roofline(s, n, per_step.get(s["label"]), (traffic or {}).get("chunk")) or {}, label=s["label"],
                            share=s["ms"] / tot,
                            tflops_algo=(s["flops"] * n / (s["ms"] * 1e-3) / 1e12) if s["ms"] and s["flops"] else 0.0,
                            GBps_algo=(s["bytes"] * n / (s["ms"] * 1e-3) / 1e9) if s["ms"] else 0.0) for s in table]}
    for k in res["kernels"]:
        k.pop("kernel", None)
    res["knobs"] = model.knobs()
    if env:
        # parity note of a non-default plan: its logits on the first frames against the DEFAULT plan's on the same frames
        res["topology"] = f"{name} ({' '.join(f'{k}={v}' for k, v in env.items())})"
        k = min(n, 512)
        ref = engine.HipFrameModel.from_keras(cfg, weights, device=device, name=name)
        d_a, d_b = engine.DeviceBuffer(k * model.n_classes * 4, device), engine.DeviceBuffer(k * model.n_classes * 4, device)
        try:
            model.predict_device(d_frames_ptr, k, d_a.ptr, logits=True)
            ref.predict_device(d_frames_ptr, k, d_b.ptr, logits=True)
            a, b = d_a.download((k, model.n_classes), np.float32), d_b.download((k, model.n_classes), np.float32)
            res["max_abs_dlogit_vs_default_plan"] = float(np.abs(a - b).max())
            res["argmax_equal_to_default_plan"] = bool(np.array_equal(a.argmax(1), b.argmax(1)))
        finally:
            ref.close(); d_a.free(); d_b.free()
    model.close()
    d_probs.free()
    if cpu_baseline is not None:
        res["cpu_baseline"] = cpu_baseline(cfg, weights, f"{name}-synth")
    return res
