#!/usr/bin/env python
"""Where load_batch's time goes on an aposteriori-style gzip .hdf5 (host only): open, link tables, native header
resolution, native inflate/placement at several thread counts.  python tools/profile_load_batch.py [n_pdb] [n_res] [batch]"""
import ctypes as C, json, os, subprocess, sys, tempfile, time, warnings
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "timed-design_amd"))
from design_utils import utils
from timed_hip import h5lite, _lib
n_pdb, n_res, bs = (int(sys.argv[i]) if len(sys.argv) > i else d for i, d in ((1, 40), (2, 100), (3, 1024)))
with tempfile.TemporaryDirectory() as td:
    h5 = os.path.join(td, "frames.hdf5")
    subprocess.run(["/opt/conda/bin/python3.9", os.path.join(ROOT, "tools", "make_synthetic_hdf5.py"), h5, str(n_pdb), str(n_res)], check=True)
    warnings.simplefilter("ignore")
    t0 = time.perf_counter(); flat, _ = utils.create_flat_dataset_map(h5); t_map = time.perf_counter() - t0
    batch = flat[:bs]
    res = dict(frames=len(flat), batch=len(batch), host_cores=os.cpu_count(), create_flat_dataset_map_s=t_map)
    for dt in (None, np.float32):
        utils.load_batch(h5, batch, dtype=dt)
        t0 = time.perf_counter()
        for _ in range(3):
            utils.load_batch(h5, batch, dtype=dt)
        res[f"load_batch_{'f64' if dt is None else 'f32'}_fps"] = 3 * len(batch) / (time.perf_counter() - t0)
    from timed_hip import engine
    if _lib.device_count() > 0:
        t0 = time.perf_counter(); buf, owner = engine.pinned_empty((len(batch), 21, 21, 21, 6), np.float32); res["pinned_alloc_s"] = time.perf_counter() - t0
        res["pinned_MB"] = buf.nbytes / 1e6
        for name, out in (("pinned_out", buf), ("pageable_out", np.empty(buf.shape, np.float32))):
            utils.load_batch(h5, batch, dtype=np.float32, out=out)
            t0 = time.perf_counter()
            for _ in range(3):
                utils.load_batch(h5, batch, dtype=np.float32, out=out)
            res[f"load_batch_f32_{name}_fps"] = 3 * len(batch) / (time.perf_counter() - t0)
        t0 = time.perf_counter(); del buf; owner.free(); res["pinned_free_s"] = time.perf_counter() - t0
    # stages
    t0 = time.perf_counter(); f = h5lite.File(h5); res["open_s"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    addrs = []
    links_cache = {}
    for p, c, r, _ in batch:
        k = (p, c)
        if k not in links_cache:
            links_cache[k] = f[p][c]._load()
        addrs.append(links_cache[k][r])
    res["links_s"] = time.perf_counter() - t0
    t0 = time.perf_counter(); rr = h5lite.resolve_many(f, addrs, num_attr="encoded_residue", num_len=20); res["resolve_s"] = time.perf_counter() - t0
    g = rr["geom"]; rank = int(g[0]); shape = [int(x) for x in g[1:1 + rank]]
    X = np.empty((len(batch), *shape), np.float32)
    lib = _lib.load()
    whole = np.frombuffer(f._m, dtype=np.uint8)
    n = len(batch)
    a = (C.c_int64 * n)(*[int(x) for x in rr["btree"]]); ptrs = (C.c_void_p * n)(*[X[i].ctypes.data for i in range(n)])
    chunk = [int(x) for x in g[8:8 + rank]]; nf = int(g[18]); filt = [int(x) for x in g[19:19 + nf]]
    try:
        res["cgroup_cpu_max"] = open("/sys/fs/cgroup/cpu.max").read().strip()
    except Exception as e:
        res["cgroup_cpu_max"] = str(e)
    res["affinity"] = len(os.sched_getaffinity(0))
    for nt in (1, 8, 16, 32, 48, 64, 96, 128, 256):
        ts = []
        for _ in range(6 if nt > 1 else 1):
            t0 = time.perf_counter()
            rc = lib.th_h5_read_chunked_as(whole.ctypes.data_as(C.c_void_p), whole.size, f._base, n, a, ptrs, rank, (C.c_int64 * rank)(*shape),
                                           (C.c_int64 * rank)(*chunk), int(g[15]), nf, (C.c_int * max(1, nf))(*filt), nt, 1)
            ts.append(time.perf_counter() - t0)
            assert rc == 0
        res[f"inflate_place_{nt}thr_fps"] = [round(n / t) for t in ts]
    del whole
    print(json.dumps(res))
