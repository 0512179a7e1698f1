"""Host-side text path of `predict.py --predict_rotamers` in isolation: '%.18e' formatting of a [n, 338] float32 matrix
(th_format_csv, threads) and the append of its ~25 bytes per value to a file, separately and together.  No GPU involved.

    python tools/bench_textio.py [rows] [directory for the scratch files]
"""
import os, sys, time, tempfile
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "timed-design_amd"))
from timed_hip import textio, _lib

lib = _lib.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 125000
where = sys.argv[2] if len(sys.argv) > 2 else None
print("usable cpus", lib.th_host_cpus())
rng = np.random.default_rng(0)
x = rng.random((n, 338), dtype=np.float32)
x /= x.sum(1, keepdims=True)
for rep in range(2):
    t0 = time.perf_counter(); tot = 0
    for m in textio._blocks(textio._check(x)):
        tot += len(m)
    dt = time.perf_counter() - t0
    print(f"format only        : {dt:.3f} s, {tot / 1e6:.0f} MB, {n * 338 / dt / 1e6:.1f} M values/s")
with tempfile.TemporaryDirectory(dir=where) as td:
    for rep in range(2):
        p = os.path.join(td, f"a{rep}.csv")
        t0 = time.perf_counter()
        with open(p, "ab") as f:
            textio.savetxt_csv(f, x)
        dt = time.perf_counter() - t0
        print(f"format + append    : {dt:.3f} s = {n / dt / 1e3:.0f} k rows/s")
        os.remove(p)
    buf = bytes(tot)
    for rep in range(2):
        p = os.path.join(td, f"b{rep}.csv")
        t0 = time.perf_counter()
        with open(p, "ab") as f:
            for i in range(0, tot, 7_000_000):
                f.write(buf[i:i + 7_000_000])
        dt = time.perf_counter() - t0
        print(f"append only        : {dt:.3f} s, {tot / 1e9 / dt:.2f} GB/s")
        os.remove(p)
    # pwrite from several threads into a preallocated range (what a parallel appender would do)
    from concurrent.futures import ThreadPoolExecutor
    for nt in (2, 4, 8):
        p = os.path.join(td, f"c{nt}.csv")
        fd = os.open(p, os.O_WRONLY | os.O_CREAT, 0o644)
        mv = memoryview(buf)
        cuts = [tot * i // nt for i in range(nt + 1)]
        def put(lo, hi):
            step = 8 << 20
            for o in range(lo, hi, step):
                os.pwrite(fd, mv[o:min(hi, o + step)], o)
        t0 = time.perf_counter()
        with ThreadPoolExecutor(nt) as ex:
            list(ex.map(lambda a: put(*a), zip(cuts[:-1], cuts[1:])))
        dt = time.perf_counter() - t0
        os.close(fd)
        print(f"pwrite, {nt} threads  : {dt:.3f} s, {tot / 1e9 / dt:.2f} GB/s")
        os.remove(p)
