"""BASELINE config 4 (TIMED-rotamer, frames sharded over the GPUs of one node, ONE RCCL gather to rank 0 — SURVEY.md §8e) on
hardware.  Two layers of evidence:

  * on ANY box, a 1-rank communicator created under TH_COMM_SELF_RCCL=1 pushes the root's own block through the grouped
    ncclSend/ncclRecv pair of th_comm_gather_rows (csrc/comm.hip) instead of a device copy: the RCCL transfer calls
    themselves run on the device and are counted (th_comm_stats);
  * on a box with >= 2 devices, min(devices, 8) processes — one per GPU — run the rotamer model on UNEVEN contiguous shards,
    gather through th_comm_gather_rows (real peer-to-peer ncclSend/ncclRecv over xGMI) and every gathered byte is compared
    with a 1-process run; predict.py's own sharded path is run the same way and its files byte-compared.  Skipped (not
    passed) on 1-GPU boxes.

Reference seam: predict.py:161-185 — rank 0 needs the whole [N, n_classes] matrix in map order."""
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SELF_PAIR_WORKER = textwrap.dedent("""
    import os, sys
    import numpy as np
    sys.path.insert(0, os.path.join({root!r}, "timed-design_amd"))
    from timed_hip import distributed as td, engine
    c = td.RcclGather(td.RcclGather.new_unique_id(), 1, 0, 0)
    rows = np.random.default_rng(5).random(({n}, {w})).astype(np.float32)
    src, dst = engine.DeviceBuffer(rows.nbytes, 0), engine.DeviceBuffer(rows.nbytes, 0)
    src.upload(rows)
    dst.upload(np.full_like(rows, -1.0))
    for _ in range(3):
        c.gather_rows_device(src.ptr, [{n}], {w}, 0, dst.ptr)
    c.barrier()
    assert np.array_equal(dst.download(({n}, {w}), np.float32), rows), "rows changed on the way through ncclSend/ncclRecv"
    s = c.stats()
    print("STATS", s["sends"], s["recvs"], s["bytes_sent"], s["bytes_received"], s["root_copies"])
    c.close()
""")


@pytest.mark.gpu
@pytest.mark.parametrize("n,w", [(37, 20), (4099, 338)])
def test_rccl_send_recv_pair_runs_on_the_device_with_one_rank(gpu, tmp_path, n, w):
    """th_comm_gather_rows' ncclSend/ncclRecv pair executes on hardware (to self) and moves the rows unchanged; without the
    knob the same gather issues no RCCL transfer at all (the root's block is a device copy)."""
    script = tmp_path / "w.py"
    script.write_text(SELF_PAIR_WORKER.format(root=ROOT, n=n, w=w))
    for knob, want in (("1", (3, 3, 3 * n * w * 4, 3 * n * w * 4, 0)), ("0", (0, 0, 0, 0, 3))):
        r = subprocess.run([sys.executable, str(script)], env=dict(os.environ, TH_COMM_SELF_RCCL=knob), capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
        got = tuple(int(x) for x in r.stdout.split("STATS")[1].split()[:5])
        assert got == want, (knob, got)


# one process per GPU: rotamer model, uneven shards, RCCL gather; the root saves what it gathered
ROTAMER_WORKER = textwrap.dedent("""
    import os, sys
    import numpy as np
    sys.path.insert(0, os.path.join({root!r}, "timed-design_amd"))
    from timed_hip import distributed as td, engine, synth
    rank, world, local = td.env_rank_world()
    n_total = {n_total}
    cfg, weights = synth.timed_synth(338)
    model = engine.HipFrameModel.from_keras(cfg, weights, device=local)
    comm = td.RcclGather.from_environment(rank, world, local)
    counts = {counts!r}
    lo = sum(counts[:rank]); hi = lo + counts[rank]
    frames = synth.synthetic_frames(n_total, seed=77)[lo:hi]
    local_rows = model.predict(frames) if hi > lo else np.empty((0, 338), np.float32)
    src = engine.DeviceBuffer(max(local_rows.nbytes, 4), local)
    if hi > lo:
        src.upload(local_rows)
    dst = engine.DeviceBuffer(n_total * 338 * 4, local) if rank == 0 else None
    comm.gather_rows_device(src.ptr, counts, 338, 0, dst.ptr if dst else 0)
    comm.barrier()
    s = comm.stats()
    print("STATS", rank, s["sends"], s["recvs"], s["bytes_sent"], s["bytes_received"], s["root_copies"])
    if rank == 0:
        np.save({out!r}, dst.download((n_total, 338), np.float32))
    comm.close()
""")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run_ranks(script, world, extra_env=None, timeout=900):
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), LOCAL_WORLD_SIZE=str(world),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", **(extra_env or {}))
        env.pop("TIMED_GATHER", None)
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        try:
            o, e = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append((p.returncode, o, e))
    for r, (rc, o, e) in enumerate(outs):
        assert rc == 0, f"rank {r}: {o[-1500:]} {e[-1500:]}"
    return outs


def _world(gpu):
    from timed_hip import _lib
    ndev = _lib.device_count()
    if ndev < 2:
        pytest.skip(f"{ndev} HIP device visible: the N > 1 RCCL gather needs one GPU per rank (runs on multi-GPU boxes)")
    return min(ndev, 8)


@pytest.mark.gpu
def test_rotamer_shards_one_process_per_gpu_rccl_gather_equals_one_process(gpu, tmp_path):
    world = _world(gpu)
    from timed_hip import engine, synth
    # uneven on purpose: rank r holds 5 + 3 r rows, the last rank none (an empty shard must not wedge the group)
    counts = [5 + 3 * r for r in range(world)]
    counts[-1] = 0
    n_total = sum(counts)
    out = tmp_path / "gathered.npy"
    script = tmp_path / "w.py"
    script.write_text(ROTAMER_WORKER.format(root=ROOT, n_total=n_total, counts=counts, out=str(out)))
    outs = _run_ranks(script, world)
    cfg, weights = synth.timed_synth(338)
    want = engine.HipFrameModel.from_keras(cfg, weights, device=gpu).predict(synth.synthetic_frames(n_total, seed=77))
    got = np.load(out)
    assert got.shape == want.shape and got.tobytes() == want.tobytes(), "the gathered matrix differs from the 1-process run"
    for r, (_rc, o, _e) in enumerate(outs):
        st = [int(x) for x in o.split("STATS")[1].split()[:6]]
        peers_with_rows = sum(1 for q in range(1, world) if counts[q] > 0)
        if r == 0:
            assert st == [0, 0, peers_with_rows, 0, sum(counts[1:]) * 338 * 4, 1], st
        else:
            assert st == [r, 1 if counts[r] else 0, 0, counts[r] * 338 * 4, 0, 0], st


PREDICT_WORKER = textwrap.dedent("""
    import os, sys, warnings
    from pathlib import Path
    sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "timed-design_amd"))
    import predict
    warnings.simplefilter("ignore")
    out = Path({out!r})
    predict.load_dataset_and_predict([Path({model!r})], {data!r}, batch_size=6, dataset_map_path=out / "datasetmap.txt",
                                     path_to_output=out, frames_per_call=4)
    print("RANK_DONE", os.environ["RANK"])
""")


@pytest.mark.gpu
def test_predict_py_one_process_per_gpu_writes_the_same_bytes_as_one_process(gpu, tmp_path):
    """predict.py's own N > 1 path (shard -> TH_PREDICT_OUT_DEVICE -> ONE th_comm_gather_rows -> every rank pwrites its text)
    over real RCCL: 26 frames over min(devices, 8) ranks, every output file equal to the single-process run's."""
    world = _world(gpu)
    import warnings
    from pathlib import Path
    import predict
    G = os.path.join(ROOT, "tests", "golden")
    model, data = os.path.join(G, "keras_tiny.h5"), os.path.join(G, "frames_tiny.hdf5")
    one, many = tmp_path / "one", tmp_path / "many"
    one.mkdir(); many.mkdir()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        predict.load_dataset_and_predict([Path(model)], data, batch_size=6, dataset_map_path=one / "datasetmap.txt", path_to_output=one)
    script = tmp_path / "w.py"
    script.write_text(PREDICT_WORKER.format(root=ROOT, out=str(many), model=model, data=data))
    _run_ranks(script, world)
    for fn in sorted(p.name for p in one.iterdir()):
        assert (one / fn).read_bytes() == (many / fn).read_bytes(), fn
