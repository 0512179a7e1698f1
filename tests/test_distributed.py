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
    sys.path.insert(0, os.path.join({root!r}, "timed-design_amd")); sys.path.insert(0, os.path.join({root!r}, "tests"))
    import torch.distributed as dist
    from timed_hip import distributed as td
    from _gloo_transport import GlooGather

    dist.init_process_group(backend="gloo")
    g = GlooGather()
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


PREDICT_WORKER = textwrap.dedent("""
    import os, sys, warnings
    from pathlib import Path
    sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "timed-design_amd")); sys.path.insert(0, os.path.join({root!r}, "tests"))
    import torch.distributed as dist
    from timed_hip import distributed as td
    import predict, _oracle_model
    from _gloo_transport import GlooGather
    warnings.simplefilter("ignore")
    dist.init_process_group(backend="gloo")
    out = Path({out!r})
    res = predict.load_dataset_and_predict([Path({model!r})], {data!r}, batch_size={bs}, start_batch={start}, dataset_map_path=out / "datasetmap.txt",
                                           path_to_output=out, frames_per_call={fpc}, model_loader=_oracle_model.load_model,
                                           gather=GlooGather())
    rank = dist.get_rank()
    lo, hi = td.shard_bounds(26 - {start} * {bs}, 2)[rank]
    assert sum(n for _d, n in _oracle_model.OracleModel.calls) == hi - lo, _oracle_model.OracleModel.calls
    if rank == 0:
        assert set(res[1]) == {{"1ubqA", "2xyz_0A", "2xyz_0B"}}
        print("ROOT_WROTE", sorted(p.name for p in out.iterdir()))
    else:
        assert res[1] is None and len(res[0]) == 26
    dist.barrier()
    dist.destroy_process_group()
""")


@pytest.mark.parametrize("bs,fpc,start", [(7, 1024, 0), (3, 5, 0), (5, 10, 2)])
def test_two_rank_predict_writes_the_same_bytes_as_one_rank(tmp_path, bs, fpc, start):
    """The real predict.load_dataset_and_predict control flow on two gloo ranks (contiguous shards of the flat dataset
    map, gather to rank 0, rank 0 writes) against the single-process run — every output file byte for byte.  The model
    is the CPU oracle behind the HipFrameModel surface (tests/_oracle_model.py): no GPU here, and the product has no
    CPU path of its own."""
    import warnings
    from pathlib import Path
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import predict
    import _oracle_model
    G = os.path.join(ROOT, "tests", "golden")
    model, data = os.path.join(G, "keras_tiny.h5"), os.path.join(G, "frames_tiny.hdf5")
    one, two = tmp_path / "one", tmp_path / "two"
    one.mkdir(); two.mkdir()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        if start:      # a resumed run appends to what the first batches left behind: seed both directories alike
            for d in (one, two):
                predict.load_dataset_and_predict([Path(model)], data, batch_size=bs, dataset_map_path=d / "datasetmap.txt",
                                                 path_to_output=d, model_loader=_oracle_model.load_model)
                for fn in ("keras_tiny.csv", "encoded_labels.csv"):
                    lines = (d / fn).read_text().splitlines(True)
                    (d / fn).write_text("".join(lines[: start * bs]))
        predict.load_dataset_and_predict([Path(model)], data, batch_size=bs, start_batch=start, dataset_map_path=one / "datasetmap.txt",
                                         path_to_output=one, frames_per_call=fpc, model_loader=_oracle_model.load_model)
    script = tmp_path / "worker.py"
    script.write_text(PREDICT_WORKER.format(root=ROOT, out=str(two), model=model, data=data, bs=bs, fpc=fpc, start=start))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(script)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "ROOT_WROTE" in r.stdout
    names = sorted(p.name for p in one.iterdir())
    assert names == sorted(p.name for p in two.iterdir()) and "keras_tiny.csv" in names
    for fn in names:
        assert (one / fn).read_bytes() == (two / fn).read_bytes(), fn


@pytest.mark.parametrize("bs,start,expect", [(4, 0, [3, 3, 3, 4, 3, 3, 3, 4]), (5, 4, [0, 1, 1, 1, 0, 1, 1, 1])])
def test_eight_rank_predict_uneven_and_empty_shards(tmp_path, bs, start, expect):
    """BASELINE config 4's rank count on CPU: eight gloo ranks through the real predict.py control flow — uneven shard
    sizes (26 rows: 3/4), and a resumed run that leaves 6 rows for 8 ranks (two ranks hold NO row: empty shard, empty
    text part, zero-row gather block).  Files byte for byte as the 1-rank run."""
    import warnings
    from pathlib import Path
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import predict
    import _oracle_model
    assert td.shard_counts(26 - start * bs, 8) == expect
    G = os.path.join(ROOT, "tests", "golden")
    model, data = os.path.join(G, "keras_tiny.h5"), os.path.join(G, "frames_tiny.hdf5")
    one, two = tmp_path / "one", tmp_path / "two"
    one.mkdir(); two.mkdir()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        if start:
            for d in (one, two):
                predict.load_dataset_and_predict([Path(model)], data, batch_size=bs, dataset_map_path=d / "datasetmap.txt",
                                                 path_to_output=d, model_loader=_oracle_model.load_model)
                for fn in ("keras_tiny.csv", "encoded_labels.csv"):
                    lines = (d / fn).read_text().splitlines(True)
                    (d / fn).write_text("".join(lines[: start * bs]))
        predict.load_dataset_and_predict([Path(model)], data, batch_size=bs, start_batch=start, dataset_map_path=one / "datasetmap.txt",
                                         path_to_output=one, frames_per_call=2, model_loader=_oracle_model.load_model)
    script = tmp_path / "worker.py"
    worker = PREDICT_WORKER.replace("td.shard_bounds(26 - {start} * {bs}, 2)", "td.shard_bounds(26 - {start} * {bs}, 8)")
    script.write_text(worker.format(root=ROOT, out=str(two), model=model, data=data, bs=bs, fpc=2, start=start))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=8", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(script)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    names = sorted(p.name for p in one.iterdir())
    assert names == sorted(p.name for p in two.iterdir())
    for fn in names:
        assert (one / fn).read_bytes() == (two / fn).read_bytes(), fn


RDZV_WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, os.path.join({root!r}, "timed-design_amd"))
    from timed_hip.rendezvous import HostRendezvous
    assert "torch" not in sys.modules
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    r = HostRendezvous(rank, world, timeout=60)
    ident = r.broadcast(bytes(range(128)) if rank == 0 else None)
    assert ident == bytes(range(128))
    table = r.allgather_ints([rank, rank * rank, -rank])
    assert table == [[q, q * q, -q] for q in range(world)], table
    parts = r.allgather(b"x" * (rank * 1000))                  # ragged payloads, an empty one from rank 0
    assert [len(p) for p in parts] == [q * 1000 for q in range(world)]
    assert r.all_min(1 if rank != world - 1 else 0) == 0 and r.all_min(7) == 7
    assert r.all_max_float(0.5 + rank) == world - 0.5
    for _ in range(20):
        r.barrier()
    r.close()
    assert "torch" not in sys.modules, "the rendezvous must not pull PyTorch in"
    sys.stdout.write("RDZV_OK %d\\n" % rank)
""")


@pytest.mark.parametrize("world", [2, 8])
def test_tcp_rendezvous_under_torchrun_without_torch(tmp_path, world):
    """timed_hip.rendezvous (the product's N > 1 control plane: RCCL id broadcast, status reduce, text sizes) inside a
    torch.distributed.run job — whose agent owns MASTER_PORT itself — without importing torch in the workers."""
    script = tmp_path / "rdzv_worker.py"
    script.write_text(RDZV_WORKER.format(root=ROOT))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(script)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    import re       # (the ranks' lines may interleave on the shared pipe)
    assert sorted(int(x) for x in re.findall(r"RDZV_OK (\d+)", r.stdout)) == list(range(world))


def test_tcp_rendezvous_skips_a_foreign_listener(tmp_path):
    """a service that is not this job sits on the first candidate port: ranks recognise it by the failed handshake"""
    import threading
    from timed_hip.rendezvous import HostRendezvous
    with socket.socket() as probe:
        probe.bind(("127.0.0.1", 0))
        base = probe.getsockname()[1]
    foreign = socket.socket()
    foreign.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
    try:
        foreign.bind(("", base + 1))
    except OSError:
        pytest.skip("candidate port already taken")
    foreign.listen(4)
    stop = threading.Event()

    def chatter():          # accepts and answers garbage
        foreign.settimeout(0.2)
        while not stop.is_set():
            try:
                c, _ = foreign.accept()
                c.sendall(b"HTTP/1.1 400 Bad Request\r\n\r\n")
                c.close()
            except OSError:
                pass
    th = threading.Thread(target=chatter)
    th.start()
    old = {k: os.environ.get(k) for k in ("MASTER_ADDR", "MASTER_PORT")}
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(base))
    out = {}
    try:
        def run(rank):
            r = HostRendezvous(rank, 2, timeout=30)
            out[rank] = r.allgather_ints([rank + 10])
            r.close()
        ts = [threading.Thread(target=run, args=(k,)) for k in (0, 1)]
        [t.start() for t in ts]
        [t.join() for t in ts]
    finally:
        stop.set()
        th.join()
        foreign.close()
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    assert out == {0: [[10], [11]], 1: [[10], [11]]}


def test_in_process_multi_device_round_robin(tmp_path):
    """devices=[...]: one handle per device in this process, call groups dealt round-robin, files unchanged"""
    import warnings
    from pathlib import Path
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import predict
    import _oracle_model
    G = os.path.join(ROOT, "tests", "golden")
    model, data = os.path.join(G, "keras_tiny.h5"), os.path.join(G, "frames_tiny.hdf5")
    a, b = tmp_path / "a", tmp_path / "b"
    a.mkdir(); b.mkdir()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        predict.load_dataset_and_predict([Path(model)], data, batch_size=4, dataset_map_path=a / "datasetmap.txt", path_to_output=a,
                                         frames_per_call=4, model_loader=_oracle_model.load_model)
        _oracle_model.OracleModel.calls.clear()
        predict.load_dataset_and_predict([Path(model)], data, batch_size=4, dataset_map_path=b / "datasetmap.txt", path_to_output=b,
                                         frames_per_call=4, devices=[0, 1, 2], model_loader=_oracle_model.load_model)
    assert [d for d, _n in _oracle_model.OracleModel.calls] == [0, 1, 2, 0, 1, 2, 0]
    for fn in sorted(p.name for p in a.iterdir()):
        assert (a / fn).read_bytes() == (b / fn).read_bytes(), fn


@pytest.mark.gpu
def test_sharded_predict_through_rccl_equals_plain_run(gpu, tmp_path):
    """The shard + gather path of predict.py on the real engine with a 1-rank RCCL communicator: probabilities are
    written by the kernels into a device shard buffer (TH_PREDICT_OUT_DEVICE), gathered with th_comm_gather_rows,
    downloaded on rank 0 and written — same bytes as the plain run."""
    import warnings
    from pathlib import Path
    import predict
    G = os.path.join(ROOT, "tests", "golden")
    model, data = Path(os.path.join(G, "keras_tiny.h5")), os.path.join(G, "frames_tiny.hdf5")
    a, b = tmp_path / "a", tmp_path / "b"
    a.mkdir(); b.mkdir()
    comm = td.RcclGather(td.RcclGather.new_unique_id(), 1, 0, gpu)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        predict.load_dataset_and_predict([model], data, batch_size=6, dataset_map_path=a / "datasetmap.txt", path_to_output=a)
        predict.load_dataset_and_predict([model], data, batch_size=6, dataset_map_path=b / "datasetmap.txt", path_to_output=b,
                                         frames_per_call=8, gather=comm)
    comm.close()
    for fn in sorted(p.name for p in a.iterdir()):
        assert (a / fn).read_bytes() == (b / fn).read_bytes(), fn


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


SHARED_GPU_WORKER = textwrap.dedent("""
    import os, sys, warnings
    from pathlib import Path
    sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "timed-design_amd"))
    import predict
    from timed_hip import distributed as td
    warnings.simplefilter("ignore")
    out = Path({out!r})
    res = predict.load_dataset_and_predict([Path({model!r})], {data!r}, batch_size={bs}, start_batch={start}, dataset_map_path=out / "datasetmap.txt",
                                           path_to_output=out, frames_per_call={fpc}, devices=[0])
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    lo, hi = td.shard_bounds(26 - {start} * {bs}, world)[rank]
    print("RANK_DONE", rank, hi - lo, "root" if res[1] is not None else "peer")
""")


@pytest.mark.gpu
@pytest.mark.parametrize("world,bs,fpc,start", [(2, 7, 1024, 0), (3, 6, 4, 0), (3, 12, 5, 2)])
def test_processes_sharing_one_gpu_write_the_same_bytes_as_one_process(gpu, tmp_path, world, bs, fpc, start):
    """BASELINE config 4's control flow on REAL kernels with more than one process: `world` processes on the one GPU of the box run
    predict.py's shard + gather path — every rank predicts its contiguous shard on the device, formats its own text, rank 0
    assembles — over the explicit host transport (TIMED_GATHER=host: RCCL refuses two ranks on one device; the RCCL transfer
    itself is covered by the 1-rank tests above and by the driver's 8-GPU run).  Shards of 13 + 13, 8 + 9 + 9 and — a resumed run
    with 2 rows left for 3 ranks — 0 + 1 + 1 rows (the ROOT's shard is empty): every file equals the single-process run's."""
    import warnings
    from pathlib import Path
    import predict
    G = os.path.join(ROOT, "tests", "golden")
    model, data = os.path.join(G, "keras_tiny.h5"), os.path.join(G, "frames_tiny.hdf5")
    one, many = tmp_path / "one", tmp_path / "many"
    one.mkdir(); many.mkdir()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        if start:      # a resumed run appends to what the first batches left behind: seed both directories alike
            for d in (one, many):
                predict.load_dataset_and_predict([Path(model)], data, batch_size=bs, dataset_map_path=d / "datasetmap.txt", path_to_output=d)
                for fn in ("keras_tiny.csv", "encoded_labels.csv"):
                    lines = (d / fn).read_text().splitlines(True)
                    (d / fn).write_text("".join(lines[: start * bs]))
        predict.load_dataset_and_predict([Path(model)], data, batch_size=bs, start_batch=start, dataset_map_path=one / "datasetmap.txt",
                                         path_to_output=one, frames_per_call=fpc)
    script = tmp_path / "worker.py"
    script.write_text(SHARED_GPU_WORKER.format(root=ROOT, out=str(many), model=model, data=data, bs=bs, fpc=fpc, start=start))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), LOCAL_WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), TIMED_GATHER="host")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        try:
            o, e = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append((p.returncode, o, e))
    for r, (rc, o, e) in enumerate(outs):
        assert rc == 0, f"rank {r}: {o[-1500:]} {e[-1500:]}"
    counts = td.shard_counts(26 - start * bs, world)
    for r, (_rc, o, _e) in enumerate(outs):
        assert f"RANK_DONE {r} {counts[r]} {'root' if r == 0 else 'peer'}" in o, o[-500:]
    for fn in sorted(p.name for p in one.iterdir()):
        assert (one / fn).read_bytes() == (many / fn).read_bytes(), fn


HOST_GATHER_WORKER = textwrap.dedent("""
    import os, sys
    import numpy as np
    sys.path.insert(0, os.path.join({root!r}, "timed-design_amd"))
    from timed_hip import distributed as td
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    g = td.HostGather(rank, world)
    counts = {counts!r}
    base = sum(counts[:rank])
    local = (np.arange(base, base + counts[rank], dtype=np.float32)[:, None] * 10 + np.arange(7, dtype=np.float32)[None, :]).astype(np.float32)
    out = g.gather_rows(local.reshape(counts[rank], 7), counts, root={root_rank})
    assert g.allgather_ints([rank, counts[rank]]) == [[r, counts[r]] for r in range(world)]
    if rank == {root_rank}:
        n = sum(counts)
        want = np.arange(n, dtype=np.float32)[:, None] * 10 + np.arange(7, dtype=np.float32)[None, :]
        assert out.shape == (n, 7) and np.array_equal(out, want)
        print("ROOT_OK")
    else:
        assert out is None
    try:
        g.gather_rows(local[:0], counts, 0) if counts[rank] else g.gather_rows(np.zeros((1, 7), np.float32), counts, 0)
        raise SystemExit("a block that disagrees with counts must be refused")
    except ValueError:
        pass
    g.barrier()
    g.close()
""")


@pytest.mark.parametrize("counts,root_rank", [([3, 0, 5], 0), ([0, 4], 1), ([1, 1, 1, 0], 0)])
def test_host_gather_over_the_tcp_rendezvous(tmp_path, counts, root_rank):
    """timed_hip.distributed.HostGather (the explicit transport for ranks that share a GPU, TIMED_GATHER=host): uneven and empty
    blocks, a root other than rank 0, a block that disagrees with `counts` refused on the spot — no PyTorch, no GPU"""
    script = tmp_path / "worker.py"
    script.write_text(HOST_GATHER_WORKER.format(root=ROOT, counts=counts, root_rank=root_rank))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    world = len(counts)
    procs = [subprocess.Popen([sys.executable, str(script)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                              env=dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port)))
             for r in range(world)]
    outs = [p.communicate(timeout=180) + (p.returncode,) for p in procs]
    for r, (o, e, rc) in enumerate(outs):
        assert rc == 0, f"rank {r}: {o[-800:]} {e[-800:]}"
    assert "ROOT_OK" in outs[root_rank][0]
