"""N>1 path on CPU: contiguous sharding + reassembly over a real 2-process gloo group.
The GPU transport (RCCL, th_comm_*) is exercised with a 1-rank communicator in the -m gpu test below
and by the driver's multi-GPU bench."""
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np
import pytest

from timed_hip import distributed as td

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_bounds_cover_exactly():
    for n in (0, 1, 7, 8, 100_000, 1_000_003):
        for w in (1, 2, 3, 8):
            b = td.shard_bounds(n, w)
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1 and sizes == td.shard_counts(n, w)
    assert td.shard_bounds(1_000_000, 8)[3] == (375_000, 500_000)  # BASELINE config 4: 125k frames per GPU


WORKER = textwrap.dedent("""
    import os, sys
    import numpy as np
    sys.path.insert(0, os.path.join({root!r}, "timed-design_amd"))
    import torch.distributed as dist
    from timed_hip import distributed as td

    dist.init_process_group(backend="gloo")
    g = td.GlooGather()
    n_total, width = {n_total}, 20

    class FakeModel:  # stands in for HipFrameModel: row i of the map -> a row that encodes i
        n_classes = width
        def predict(self, X):
            return (X[:, :1] + np.arange(width, dtype=np.float32)[None, :] / 100.0).astype(np.float32)

    seen = []
    def frames_for_range(lo, hi):
        seen.append((lo, hi))
        return np.arange(lo, hi, dtype=np.float32)[:, None]

    out = td.predict_sharded(FakeModel(), frames_for_range, n_total, g, root=0)
    lo, hi = td.shard_bounds(n_total, g.world)[g.rank]
    assert seen == ([(lo, hi)] if hi > lo else []), seen
    if g.rank == 0:
        want = np.arange(n_total, dtype=np.float32)[:, None] + np.arange(width, dtype=np.float32)[None, :] / 100.0
        assert out.shape == (n_total, width) and np.array_equal(out, want.astype(np.float32)), "rows out of map order"
        print("ROOT_OK", n_total)
    else:
        assert out is None
    g.barrier()
    dist.destroy_process_group()
""")


@pytest.mark.parametrize("n_total", [11, 2, 1])
def test_two_rank_gloo_gather_restores_map_order(tmp_path, n_total):
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT, n_total=n_total))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(script)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert f"ROOT_OK {n_total}" in r.stdout


@pytest.mark.gpu
def test_rccl_transport_single_rank(gpu):
    """th_comm_* end to end with a 1-rank communicator: dlopen(librccl), init, gather (root's own
    block is a device copy), barrier."""
    from timed_hip import engine
    uid = td.RcclGather.new_unique_id()
    assert len(uid) == 128
    c = td.RcclGather(uid, 1, 0, gpu)
    rows = np.random.default_rng(0).random((37, 20)).astype(np.float32)
    src = engine.DeviceBuffer(rows.nbytes, gpu)
    dst = engine.DeviceBuffer(rows.nbytes, gpu)
    src.upload(rows)
    c.gather_rows_device(src.ptr, [37], 20, 0, dst.ptr)
    c.barrier()
    assert np.array_equal(dst.download((37, 20), np.float32), rows)
    c.close()
