"""predict.py — the reference's command line, ``load_dataset_and_predict`` signature and output files, with the Keras
model object replaced by the HIP engine and the strictly sequential batch loop replaced by a pipeline:

    reference predict.py:121   frame_model = tf.keras.models.load_model(Path(m))
    reference predict.py:142   y_pred_batch = frame_model.predict(X_batch)
    here                       engine.load_model(m) ; frame_model.predict_async(X) ... .result()

    python3 predict.py --path_to_dataset data.hdf5 --path_to_model TIMED.h5 --path_to_output .

``--path_to_model`` takes a Keras legacy ``.h5`` (converted on the fly) or a ``.pack``; ``--path_to_dataset`` an
aposteriori ``.hdf5``, a frame pack, or a PDB file (``.pdb[.gz]`` / ``.pdb1[.gz]`` / ``.ent``: voxelised on the GPU by
timed_hip.voxeliser — parity unpinned against aposteriori).  Outputs (reference README.md:119-131): <model>.csv, <model>.fasta, <model>.txt,
dataset.fasta, datasetmap.txt, encoded_labels.csv, plus <model>_rot.csv in rotamer mode.

How a run is organised (all of it invisible in the files, which are byte-identical to a batch-by-batch run):
  * consecutive reference batches are grouped into GPU calls of about ``frames_per_call`` frames;
  * a loader thread builds group g+1 (reference load_batch, utils.py:487-530) while the GPU computes group g and the
    main thread formats and appends the outputs of group g-1 (th_predict_async / th_predict_wait);
  * ``devices=[0, 1, ...]`` (``--devices 0,1``): one model handle per GPU in this process, groups dealt round-robin;
  * one process per GPU (WORLD_SIZE > 1: ``torch.distributed.run`` or any launcher that sets RANK / WORLD_SIZE /
    MASTER_ADDR / MASTER_PORT; the product itself needs no PyTorch): the flat dataset map is cut into contiguous per-rank
    shards, every rank predicts its shard into device memory AND formats the text of its own rows, ONE gather (RCCL over
    xGMI, th_comm_gather_rows) assembles the [N, n_classes] matrix on rank 0 for the per-chain sequences, every rank
    writes its text at its byte offset of the shared files; other ranks return only the dataset map.

Deliberate differences from the reference: the raw rotamer probabilities go to
``<model_name>_rot.csv`` (the reference's missing f-string writes a file literally called "{model_name}_rot.csv",
predict.py:123 — the UI and scripts expect the real name); the consensus files honour --path_to_output.
"""
import argparse
import os
import sys
import threading
from collections import deque
from concurrent.futures import ThreadPoolExecutor
from math import ceil
from pathlib import Path

import numpy as np

from design_utils import utils as du
from timed_hip import engine, textio

N_RESIDUE_CLASSES, N_ROTAMER_CLASSES = 20, 338


# ---- one model over one dataset --------------------------------------------------------------------------------
class _TextSink:
    """Append-only files held in memory: what one rank of a sharded run formats for ITS rows (put into the real files,
    at this rank's byte offset, by _assemble_shard_files)."""

    class _Handle:
        def __init__(self, blocks):
            self._blocks = blocks

        def write(self, data):
            # textio hands out views of a scratch buffer it reuses: keep a copy
            self._blocks.append(data.encode("ascii") if isinstance(data, str) else bytes(data))

        def __enter__(self):
            return self

        def __exit__(self, *exc):
            return False

    def __init__(self):
        self.blocks = {}

    def opener(self, path):
        return self._Handle(self.blocks.setdefault(str(path), []))

    def size(self, path) -> int:
        return sum(len(b) for b in self.blocks.get(str(path), ()))


class _OutputFiles:
    """The per-model files of reference predict.py:123,145-155 / utils.save_outputs_to_file, appended group by group in
    row order, and the float16 matrix the reference re-reads from the CSV afterwards (predict.py:163).  With a ``sink``
    (sharded runs) the text of this rank's rows is collected in memory instead of being appended to the files."""

    def __init__(self, model_index, model_name, flat_dataset_map, path_to_output, predict_rotamers, codec, resume, sink=None,
                 device=None):
        self.model_index, self.model_name = model_index, model_name
        self.flat_dataset_map, self.path_to_output = flat_dataset_map, path_to_output
        self.codec = codec
        self.sink = sink
        # the GPU that formats the full-precision rotamer matrix (th_format_csv_device); None / TIMED_GPU_FORMAT=0: host threads
        self.device = device if os.environ.get("TIMED_GPU_FORMAT", "1") != "0" else None
        self.matrix_path = path_to_output / (f"{model_name}_rot.csv" if predict_rotamers else f"{model_name}.csv")
        # rows already in the file (an earlier, interrupted or repeated run) are part of what the reference reads back
        self.must_reread = resume or (self.matrix_path.exists() and self.matrix_path.stat().st_size > 0)
        self._f16_rows = []
        self._scratch = textio.TextScratch()        # the writer thread's text buffer (page-locked when the GPU formats into it)
        self._codec_matrix = (np.array([codec[k] for k in range(len(codec))], dtype=np.float16) if codec is not None else None)

    def paths(self):
        """the files ``append`` writes, in a fixed order (the dataset map is not one of them)"""
        out = [self.path_to_output / "encoded_labels.csv"] if self.model_index == 0 else []
        if self.codec is not None:
            out.append(self.matrix_path)
        return out + [self.path_to_output / f"{self.model_name}.csv"]

    def _open(self, path):
        return self.sink.opener(path) if self.sink is not None else open(path, "ab")

    def append(self, probs: np.ndarray, labels: np.ndarray):
        f16 = probs.astype(np.float16)
        if self.sink is None:
            self._f16_rows.append(f16)      # (sharded runs take the matrix from the gather instead)
        if self.codec is not None:
            with self._open(self.matrix_path) as f:
                # = np.savetxt(f, y_pred_batch, delimiter=","), full precision: 338 '%.18e' values per residue, formatted on the GPU
                textio.savetxt_csv(f, probs, device=self.device, scratch=self._scratch)     # (float32 rows only; anything else stays on the host threads)
            f16 = self._codec_matrix[np.argmax(probs, axis=1)]      # one-hot residue of the arg-max rotamer
        # the appends of utils.save_outputs_to_file, on arrays: no list-of-lists round trip under the GIL while the
        # submitter thread is waiting to launch the next group
        du.append_outputs(labels, f16, self.flat_dataset_map, self.model_index, self.model_name, self.path_to_output,
                          opener=self._open, write_map=self.sink is None)

    def close(self):
        self._scratch.close()

    def set_gathered(self, probs: np.ndarray):
        """sharded runs, rank 0: the [N, n_classes] float32 rows of every rank in map order (the RCCL gather)"""
        self._f16_rows = [probs.astype(np.float16)]

    def prediction_matrix(self) -> np.ndarray:
        """what np.genfromtxt(matrix_path, delimiter=",", dtype=np.float16) would return: '%.18e' text round-trips every
        float16/float32 exactly, so for a fresh file it is the float16 cast of what was written; a file that already
        held rows is read back."""
        if self.must_reread or not self._f16_rows:
            return textio.loadtxt_f16(self.matrix_path)
        return np.concatenate(self._f16_rows, axis=0)


def _assemble_shard_files(files, gather, rank, world):
    """Sharded runs: every rank formatted the text of its OWN rows (rows are independent, so the files of the 1-rank run
    are the concatenation of the ranks' texts in rank order).  The ranks tell each other how many bytes each file part
    has, and every rank writes its part at its byte offset — no text travels through rank 0 (config 4's _rot.csv is
    8.5 GB of '%.18e' text; formatted and written by one core it would idle eight GPUs).  One node, one file system."""
    paths = files.paths()
    mine = [files.sink.size(p) for p in paths]
    # rank 0 also reports what the files hold already (append semantics of utils.py:758,770: a resumed or repeated run)
    bases = [(p.stat().st_size if p.exists() else 0) if rank == 0 else 0 for p in paths]
    table = gather.allgather_ints(mine + bases)             # doubles as the barrier between "stat" and "write"
    if rank == 0 and not (files.path_to_output / "datasetmap.txt").exists():
        with open(files.path_to_output / "datasetmap.txt", "a") as f:
            du._savetxt_strings(f, files.flat_dataset_map)
    for j, p in enumerate(paths):
        offset = table[0][len(paths) + j] + sum(table[r][j] for r in range(rank))
        blocks = files.sink.blocks.get(str(p), [])
        if not blocks and rank != 0:
            continue                                        # (rank 0 always creates the file, even an empty one)
        fd = os.open(p, os.O_WRONLY | os.O_CREAT, 0o666)
        try:
            for b in blocks:
                view = memoryview(b)
                while len(view):
                    done = os.pwrite(fd, view, offset)
                    offset += done
                    view = view[done:]
        finally:
            os.close(fd)
    files.sink.blocks.clear()
    gather.barrier()                                        # rank 0 reads the assembled files from here on


def _row_groups(n_rows, batch_size, start_batch, frames_per_call):
    """[lo, hi) row ranges, each a whole number of reference batches (so resuming at ``start_batch`` lines up)"""
    per_call = max(1, int(frames_per_call) // max(1, batch_size)) * batch_size
    first = start_batch * batch_size
    groups, lo = [], first
    # a run of several large calls starts with a quarter- and a half-size one: the GPU gets its first frames after a quarter of
    # a group's load time instead of a whole one (the first load and the last forward pass are the two stages nothing overlaps)
    ramp = [per_call // 4, per_call // 2] if per_call >= 1024 and n_rows - first > per_call else []
    for size in ramp:
        size = size // batch_size * batch_size
        if size >= batch_size and lo + size < n_rows:
            groups.append((lo, lo + size))
            lo += size
    groups += [(a, min(a + per_call, n_rows)) for a in range(lo, n_rows, per_call)]
    return groups


def _is_sparse(X) -> bool:
    return type(X).__name__ == "SparseFrames"


def _models_take_sparse(models) -> bool:
    """every model of the round-robin is a HIP engine handle (or the sharded wrapper around one): only those read SparseFrames"""
    for m in models:
        inner = getattr(m, "sparse_ok", None)
        if inner is None:
            inner = hasattr(m, "_predict_async_sparse")
        if not inner:
            return False
    return True


def _run_groups(models, dataset_path, flat_dataset_map, groups, consume, gpu_decode=False):
    """Pipeline over the call groups, three stages on three threads:
         loader thread   load_batch of group g+1 (reference utils.py:487-530)
         this thread     th_predict_async of group g, round-robin over ``models`` (the host->device copy of pageable
                         frames blocks here while the kernels of group g-1 run)
         writer thread   ``consume(probs, labels)`` for group g-1, strictly in group order (formats and appends)
    An exception in any stage surfaces here."""
    if not groups:
        return
    depth = 2 * len(models)

    # Batches of an .hdf5 dataset are built in a small ring of reused buffers: a fresh 0.2-0.5 GB NumPy batch costs more
    # in first-touch page faults than its inflation does.  The first ``ring_size`` full-size batches simply become the
    # ring (no extra allocation); a buffer is reused only after the ticket that read it has completed: at most 3 groups
    # per model are outstanding (below) + the one being loaded + one spare.  Pageable memory on purpose: host->device
    # copies from it already run at the PCIe rate here, and page-locking 1 GB costs more than a short run takes
    # (th_host_alloc exists for callers that keep their buffers).  Frame packs hand out memory-mapped rows instead.
    ring = []
    from timed_hip import framepack
    plain_h5 = not framepack.is_pack(dataset_path) and not framepack.is_structure(dataset_path)
    # a float32 pack with sparse transport files (framepack.sparsify): the loader hands out SparseFrames batches — a tenth of the
    # bytes cross PCIe, the device rebuilds the dense frames (th_predict_sparse_async).  TIMED_SPARSE=0: the dense rows as before
    sparse_pack = None
    if framepack.is_pack(dataset_path) and os.environ.get("TIMED_SPARSE", "1") != "0" and getattr(models[0], "device", None) is not None:
        fp = du._frame_pack(dataset_path)
        if fp.sparse is not None and _models_take_sparse(models):
            sparse_pack = fp
    # loader threads: two on GPU-inflated datasets (see below), one elsewhere.  The ring is sized for ALL of them: when the device
    # decode falls back to the host reader mid-run (an unsupported file, TH_ENOMEM), both loaders keep filling host buffers, and
    # a ring sized for one would hand group k + 2 the slot of a group whose ticket may still be outstanding
    n_loaders = max(1, int(os.environ.get("TIMED_LOADERS", "2" if (gpu_decode and plain_h5) else "1")))
    ring_size = 3 * len(models) + 1 + n_loaders
    max_rows = max(hi - lo for lo, hi in groups)
    use_ring = len(groups) > ring_size and plain_h5
    full_seen = [0]                    # full-size groups loaded so far: they take the ring's slots in turn (ramp-up groups are smaller)

    load_seconds = [0.0]

    def load(k):
        import time
        t0 = time.perf_counter()
        try:
            return _load(k)
        finally:
            load_seconds[0] += time.perf_counter() - t0

    decode_on_gpu = [bool(gpu_decode) and plain_h5]

    host_lock = threading.Lock()       # the host reader's buffer ring is filled by one loader at a time

    # Memory-mapped rows of a frame pack go through a ring of page-locked buffers (engine.StagingRing): the copy to the device
    # then runs at the PCIe rate instead of the rate at which the driver pins pages it has never seen.  3 groups per model
    # outstanding + the one waiting to be submitted + the one being filled.  TIMED_STAGING=0 hands out the mapped rows as before.
    staging = [None]

    sparse_ring = [None]

    def _stage_sparse(X):
        # a sparse batch (views of the memory-mapped transport files) is assembled into ONE page-locked blob: rank table, bitmaps, values
        if sparse_ring[0] is False:
            return X, None
        if sparse_ring[0] is None:
            sparse_ring[0] = False
            try:
                from timed_hip import _lib, engine
                cpus = int(_lib.load().th_host_cpus())
                ranks_here = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1") or 1))
                threads = int(os.environ.get("TIMED_STAGING_THREADS", str(max(1, min(8, (cpus - 2) // ranks_here)))))
                if os.environ.get("TIMED_STAGING", "1") != "0":
                    sparse_ring[0] = engine.BlobRing(3 * len(models) + 2, threads)
            except Exception:
                sparse_ring[0] = False
            if sparse_ring[0] is False:
                return X, None
        got, slot = sparse_ring[0].stage(X)
        return got, (None if slot is None else ("sparse", slot))

    def _stage(X):
        if _is_sparse(X):
            return _stage_sparse(X)
        if staging[0] is False or not isinstance(X, np.memmap) or getattr(models[0], "device", None) is None:
            return X, None
        if staging[0] is None:
            staging[0] = False
            try:
                from timed_hip import _lib, engine
                cpus = int(_lib.load().th_host_cpus())
                ranks_here = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1") or 1))          # one process per GPU shares the host's cores
                threads = int(os.environ.get("TIMED_STAGING_THREADS", str(min(8, (cpus - 2) // ranks_here))))
                if os.environ.get("TIMED_STAGING", "1") != "0" and threads >= 2 and len(groups) > 2:
                    staging[0] = du.take_staging_ring(3 * len(models) + 2, max_rows, X.shape[1:], X.dtype, threads)
            except Exception:
                staging[0] = False
            if staging[0] is False:
                return X, None
        return staging[0].stage(X)

    def _load(k):
        X, y = _load_rows(k)
        X, slot = _stage(X)
        return X, y, slot

    def _load_rows(k):
        lo, hi = groups[k]
        if sparse_pack is not None:
            rows = sparse_pack.contiguous_rows(flat_dataset_map[lo:hi])
            if rows is not None:
                return sparse_pack.sparse_batch(*rows), sparse_pack.labels[rows[0]:rows[1]].astype(float)
        if decode_on_gpu[0]:
            # gzip .hdf5: the chunks of this group are inflated ON the GPU that will predict it (th_h5_decode_device) — the
            # frames never exist on the host; a dataset that cannot take the path (other filters, a layout h5lite does not read ...) says so
            # once and the host reader takes over
            got = du.load_batch_device(dataset_path, flat_dataset_map[lo:hi], device=models[k % len(models)].device)
            if got is not None:
                return got
            decode_on_gpu[0] = False
        with host_lock:
            # full-size groups take the ring's slots in turn, starting with the FIRST full-size group (the smaller ramp-up groups
            # in front of it, and the ragged last one, get buffers of their own)
            slot = None
            if use_ring and hi - lo == max_rows:
                slot = full_seen[0] % ring_size
                full_seen[0] += 1
            # float32 frames: the rounding Keras applies to load_batch's float64 anyway, done while the chunks are placed
            X, y = du.load_batch(dataset_path, flat_dataset_map[lo:hi], dtype=np.float32,
                                 out=ring[slot] if slot is not None and slot < len(ring) else None)
            if slot is not None and len(ring) == slot and isinstance(X, np.ndarray) and X.base is None and len(X) == max_rows:
                ring.append(X)
            return X, y

    write_seconds = [0.0, 0.0]

    def finish(ticket, labels, slot=None):
        import time
        t0 = time.perf_counter()
        probs = ticket.result()
        if isinstance(slot, tuple):
            sparse_ring[0].release(slot[1])
        elif slot is not None:
            staging[0].release(slot)
        t1 = time.perf_counter()
        consume(probs, labels)
        write_seconds[0] += t1 - t0
        write_seconds[1] += time.perf_counter() - t1

    pending = deque()          # tickets submitted to a GPU, oldest first
    writing = deque()          # futures of the writer thread, oldest first
    # three Python threads share the interpreter lock; with the default 5 ms switch interval the submitter can sit behind
    # the writer for longer than a whole group takes on the GPU (4.6 ms per 1024 frames)
    switch = sys.getswitchinterval()
    sys.setswitchinterval(2e-4)
    completed = False
    try:
        # Two loaders on GPU-inflated datasets: the host part of batch g+2 (object headers, B-trees, descriptors, ~6 ms of Python
        # per 4096 frames) and its pageable upload run while the inflate kernels of batch g+1 hold the decoder's scratch set.
        # Measured on 40 k gzip float64 frames (tools/trace_predict_e2e.py): the loop takes 0.229 s with one loader, 0.178 s with
        # two — 17.8 ms per 4096 frames, which is the inflate kernels (8 ms) plus the CNN (10.3 ms) one after the other on the
        # GPU — and 0.19 s with three.  (When the CNN took twice as long the second loader bought nothing.)  One loader elsewhere:
        # the host reader fills its ring under a lock anyway, and frame packs hand out mapped rows.
        _pump(models, groups, load, finish, pending, writing, depth, loaders=n_loaders)
        completed = True
    finally:
        sys.setswitchinterval(switch)
        if staging[0]:
            for ticket, *_rest in pending:           # after an error: nothing may still be copying out of a slot
                try:
                    ticket.result()
                except Exception:
                    pass
            if completed:
                du.give_back_staging_ring(staging[0])    # kept for the next call; du.release_device_memory() (the CLI, at its end) closes it
            else:
                staging[0].close()
        if sparse_ring[0]:
            if not completed:
                for ticket, *_rest in pending:
                    try:
                        ticket.result()
                    except Exception:
                        pass
            sparse_ring[0].close()
    if os.environ.get("TIMED_PIPELINE_TRACE"):
        print(f"[pipeline] load_batch calls took {load_seconds[0]:.3f} s in the loader thread(s); the writer thread waited {write_seconds[0]:.3f} s "
              f"for results and spent {write_seconds[1]:.3f} s formatting / appending", file=sys.stderr)
        if decode_on_gpu[0] and any(du._LOAD_TRACE):
            a = du._LOAD_TRACE
            print(f"[pipeline] load_batch_device: group links {a[0]:.3f} s, object headers (resolve_many) {a[1]:.3f} s, device buffer {a[2]:.3f} s, "
                  f"B-trees + upload + inflate {a[3]:.3f} s", file=sys.stderr)
            a[:] = [0.0] * 4
    del ring[:]


def _pump(models, groups, load, finish, pending, writing, depth, loaders=1):
    trace = os.environ.get("TIMED_PIPELINE_TRACE")
    if trace:
        import time
        t_load = t_wait = t_submit = 0.0
        clock = time.perf_counter
    with ThreadPoolExecutor(max_workers=loaders, thread_name_prefix="load_batch") as loader, \
            ThreadPoolExecutor(max_workers=1, thread_name_prefix="write_outputs") as writer:
        ahead = deque(loader.submit(load, j) for j in range(min(loaders, len(groups))))     # groups being loaded, in order
        for k in range(len(groups)):
            if trace:
                t0 = clock()
            X, y, *extra = ahead.popleft().result()
            if k + loaders < len(groups):
                ahead.append(loader.submit(load, k + loaders))
            if trace:
                t1 = clock(); t_load += t1 - t0
            # a model has 4 tickets (th_predict_async): at most 3 consecutive groups per model are outstanding here,
            # 2 on the GPU queue and 1 with the writer
            while len(pending) + len(writing) >= 3 * len(models):
                if not writing:
                    writing.append(writer.submit(finish, *pending.popleft()))
                writing.popleft().result()
            if trace:
                t2 = clock(); t_wait += t2 - t1
            pending.append((models[k % len(models)].predict_async(X), y, *extra))
            del X
            if trace:
                t_submit += clock() - t2
            if len(pending) >= depth:
                writing.append(writer.submit(finish, *pending.popleft()))
        while pending:
            writing.append(writer.submit(finish, *pending.popleft()))
        while writing:
            writing.popleft().result()
    if trace:
        print(f"[pipeline] {len(groups)} groups: waiting for the loader {t_load:.3f} s, for the writer/GPU {t_wait:.3f} s, "
              f"submitting (incl. pageable host->device copies) {t_submit:.3f} s", file=sys.stderr)


def _distributed_context(gather):
    """(rank, world, local_rank, gather): world > 1 when launched one process per GPU (torch.distributed.run)."""
    from timed_hip import distributed as td
    if gather is not None:
        return gather.rank, gather.world, int(os.environ.get("LOCAL_RANK", gather.rank)), gather
    rank, world, local = td.env_rank_world()
    return rank, world, local, None


def load_dataset_and_predict(
    models: list,
    dataset_path: Path,
    batch_size: int = 20,
    start_batch: int = 0,
    dataset_map_path: Path = "datasetmap.txt",
    blacklist: Path = None,
    predict_rotamers: bool = False,
    model_name_suffix: str = "",
    is_consensus: bool = False,
    path_to_output: Path = Path.cwd(),
    device: int = 0,
    frames_per_call: int = None,
    devices=None,
    model_loader=None,
    gather=None,
) -> (np.ndarray, np.ndarray, np.ndarray, np.ndarray, np.ndarray, np.ndarray):
    """reference predict.py:28-194 — same leading parameters and return tuple (flat_dataset_map, pdb_to_sequence,
    pdb_to_probability, pdb_to_real_sequence, pdb_to_consensus, pdb_to_consensus_prob).  ``start_batch`` keeps the
    reference's resume semantics (batches before it are skipped, outputs are appended).

    Opt-in extras: ``device`` / ``devices`` (HIP device indices for this process), ``frames_per_call`` (frames handed to
    a GPU per call), ``model_loader(path, device=...)`` (default ``timed_hip.engine.load_model``) and ``gather`` (a
    row-gather transport from ``timed_hip.distributed``; default: RCCL when WORLD_SIZE > 1)."""
    path_to_output = Path(path_to_output)
    n_classes = N_ROTAMER_CLASSES if predict_rotamers else N_RESIDUE_CLASSES
    rank, world, local_rank, gather = _distributed_context(gather)
    sharded = world > 1 or gather is not None      # an explicit transport selects the shard + gather path even for 1 rank
    if rank == 0:
        print(f"Predicting {n_classes} classes per residue ({'rotamer' if predict_rotamers else 'residue'} mode)")
    loader = model_loader or engine.load_model
    # gzip .hdf5 datasets are inflated on the GPU unless TIMED_GPU_INFLATE=0 (or a model double without a device is in use)
    gpu_decode = model_loader is None and os.environ.get("TIMED_GPU_INFLATE", "1") != "0"
    if frames_per_call is None:
        # frames handed to a GPU per call: 1024 keeps the host->device copies of a call hidden behind the previous call's
        # kernels; a batch that is inflated ON the GPU is better large — one stream's Huffman layer is a
        # 5 ms serial chain whatever the batch, so 4096 frames cost 31 ms of decode where 4 x 1024 cost 45
        from timed_hip import framepack as _fp
        on_gpu = gpu_decode and not _fp.is_pack(dataset_path) and not _fp.is_structure(dataset_path)
        frames_per_call = 4096 if on_gpu else 1024
    if world > 1:
        device_ids = [local_rank if devices is None else list(devices)[local_rank % len(devices)]]
        if model_loader is None:
            from timed_hip import _lib
            visible = _lib.device_count()
            if device_ids[0] >= visible:
                raise RuntimeError(f"rank {rank} (local rank {local_rank}) needs HIP device {device_ids[0]} but only {visible} "
                                   f"device(s) are visible: launch one process per GPU")
    elif sharded:
        device_ids = [device]
    else:
        device_ids = [device] if not devices else list(devices)
    # the first model is loaded (device allocations, weight upload: native code) while this thread builds the dataset map
    side = ThreadPoolExecutor(max_workers=1, thread_name_prefix="predict_side")
    pending_handles = side.submit(lambda m=models[0]: [loader(Path(m), device=d) for d in device_ids]) if models else None
    if Path(dataset_map_path).exists():
        flat_dataset_map = du.read_flat_dataset_map(dataset_map_path)
    else:
        excluded = du.get_pdb_keys_to_filter(blacklist) if blacklist else []
        flat_dataset_map = du.flat_dataset_map_array(dataset_path, excluded)
    # (the reference converts to an array after the first model; rows slice as arrays)
    # what the sequence extraction needs from the map alone is prepared on the side thread, under the GPU's work
    plan = side.submit(du.SequencePlan, flat_dataset_map) if rank == 0 else None
    old_datasetmap = len(flat_dataset_map[0]) == 4
    codec, flat_categories = du.get_rotamer_codec() if predict_rotamers else (None, None)
    outputs = (None,) * 5
    try:
        for index, model_path in enumerate(models):
            model_name = (model_path.stem if isinstance(model_path, Path) else str(model_path)) + model_name_suffix
            handles = pending_handles.result()
            pending_handles = None
            if index + 1 < len(models):      # the next model loads while this one runs
                pending_handles = side.submit(lambda m=models[index + 1]: [loader(Path(m), device=d) for d in device_ids])
            srb = None
            files = None
            try:
                for h in handles:
                    if h.n_classes != n_classes:
                        raise ValueError(f"{model_path}: the model has {h.n_classes} outputs, predict_rotamers="
                                         f"{predict_rotamers} needs {n_classes}")
                if rank == 0:
                    # <model>.txt depends on the map only: written on the side thread while the GPU works
                    srb = side.submit(du.convert_dataset_map_for_srb, flat_dataset_map, model_name, path_to_output)
                if sharded:
                    files = _OutputFiles(index, model_name, flat_dataset_map, path_to_output, predict_rotamers, codec,
                                         resume=start_batch > 0, sink=_TextSink(), device=getattr(handles[0], "device", None))
                    gathered = _predict_sharded(handles[0], gather, rank, world, dataset_path, flat_dataset_map, batch_size,
                                                start_batch, frames_per_call, files, gpu_decode=gpu_decode)
                    if rank != 0:
                        continue
                    files.set_gathered(gathered)
                else:
                    files = _OutputFiles(index, model_name, flat_dataset_map, path_to_output, predict_rotamers, codec,
                                         resume=start_batch > 0, device=getattr(handles[0], "device", None))
                    _run_groups(handles, dataset_path, flat_dataset_map,
                                _row_groups(len(flat_dataset_map), batch_size, start_batch, frames_per_call), files.append,
                                gpu_decode=gpu_decode)
            finally:
                if files is not None:
                    files.close()
                for h in handles:
                    h.close()
                if srb is not None:
                    srb.result()
            outputs = du.extract_sequence_from_pred_matrix(
                flat_dataset_map, files.prediction_matrix(), rotamers_categories=flat_categories if predict_rotamers else None,
                old_datasetmap=old_datasetmap, is_consensus=is_consensus, plan=plan.result())
            pdb_to_sequence, _prob, pdb_to_real_sequence, pdb_to_consensus, pdb_to_consensus_prob = outputs
            du.save_dict_to_fasta(pdb_to_sequence, model_name, path_to_output)
            du.save_dict_to_fasta(pdb_to_real_sequence, "dataset", path_to_output)
            if pdb_to_consensus:
                du.save_dict_to_fasta(pdb_to_consensus, model_name + "_consensus", path_to_output)
                du.save_consensus_probs(pdb_to_consensus_prob, model_name, path_to_output)
    finally:
        if pending_handles is not None:          # a model that was prefetched but never used (an error above)
            try:
                for h in pending_handles.result():
                    h.close()
            except Exception:
                pass
        side.shutdown(wait=True)
    return (flat_dataset_map, *outputs)


def _predict_sharded(model, gather, rank, world, dataset_path, flat_dataset_map, batch_size, start_batch, frames_per_call, files,
                     gpu_decode=False):
    """One process per GPU: rows [row0, N) are cut into ``world`` contiguous shards (timed_hip.distributed.shard_bounds);
    this rank predicts its shard, formats the text of its own rows (``files``, an in-memory sink) while the GPU works,
    the probability shards are gathered to rank 0 in rank order = map order and the file parts are put
    together by _assemble_shard_files.  With the default transport the probabilities stay on the GPUs for the gather:
    predict_async writes them into a device shard buffer, th_comm_gather_rows moves them over xGMI, and the writer
    thread fetches each group's rows from the shard buffer for formatting.  Returns the [n, n_classes] float32 matrix
    on rank 0, None elsewhere."""
    from timed_hip import distributed as td
    row0 = min(start_batch * batch_size, len(flat_dataset_map))
    n = len(flat_dataset_map) - row0
    counts = td.shard_counts(n, world)
    lo, hi = td.shard_bounds(n, world)[rank]
    shard = flat_dataset_map[row0 + lo: row0 + hi]
    groups = [(a, min(a + max(1, int(frames_per_call)), len(shard))) for a in range(0, len(shard), max(1, int(frames_per_call)))]
    own_transport = gather is None
    if own_transport:
        if os.environ.get("TIMED_GATHER", "rccl") == "host":
            # explicit opt-in for ranks that share a GPU (RCCL refuses two ranks on one device): rows travel over the TCP rendezvous
            gather = td.HostGather(rank, world)
        else:
            gather = td.RcclGather.from_environment(rank, world, model.device)     # raises when RCCL cannot be brought up
    try:
        if isinstance(gather, td.RcclGather):
            width = model.n_classes
            d_local = engine.DeviceBuffer(max(1, len(shard) * width * 4), model.device)

            class _RowsOnDevice:      # a ticket whose rows sit in d_local: result() brings them to the host for formatting
                def __init__(self, ticket, row, rows):
                    self.ticket, self.row, self.rows = ticket, row, rows

                def result(self):
                    self.ticket.result()
                    return d_local.download((self.rows, width), np.float32, offset=self.row * width * 4)

            class _ToDevice:          # the model facade _run_groups drives: outputs land in d_local at the shard row
                sparse_ok = True          # predict_async_device reads SparseFrames batches too

                def __init__(self):
                    self.row = 0
                    self.device = model.device

                def predict_async(self, X):
                    t = _RowsOnDevice(model.predict_async_device(X, d_local.ptr + self.row * width * 4), self.row, len(X))
                    self.row += len(X)
                    return t
            _run_groups([_ToDevice()], dataset_path, shard, groups, files.append, gpu_decode=gpu_decode)
            d_all = engine.DeviceBuffer(max(1, n * width * 4), model.device) if rank == 0 else None
            gather.gather_rows_device(d_local.ptr, counts, width, 0, d_all.ptr if d_all else 0)
            probs = d_all.download((n, width), np.float32) if rank == 0 else None
        else:                     # host transport (distributed.HostGather; tests/_gloo_transport.GlooGather): rows are on the host already
            local = np.zeros((len(shard), model.n_classes), dtype=np.float32)
            cursor = [0]

            def keep(p, y):
                local[cursor[0]:cursor[0] + len(y)] = p
                cursor[0] += len(y)
                files.append(p, y)
            _run_groups([model], dataset_path, shard, groups, keep, gpu_decode=gpu_decode)
            probs = gather.gather_rows(local, counts, 0)
        _assemble_shard_files(files, gather, rank, world)
    finally:
        if own_transport:
            gather.close()
    return probs if rank == 0 else None


# ---- command line ----------------------------------------------------------------------------------------------
# (flag, argparse keywords): names, types and defaults are the reference's (predict.py:251-296); --device/--devices and
# --frames_per_call are additions of this build.
CLI_FLAGS = (
    ("--batch_size", dict(type=int, default=12, help="frames per reference batch; also the unit --start_batch-style resumes count in")),
    ("--path_to_dataset", dict(type=str, help="aposteriori frame dataset (.hdf5) or frame pack")),
    ("--path_to_datasetmap", dict(type=str, default="datasetmap.txt", help="flat dataset map (.txt); created when missing")),
    ("--path_to_model", dict(type=str, help="Keras legacy .h5 model or converted .pack")),
    ("--path_to_blacklist", dict(type=str, default=None, help="directory of PDB lists to refuse (training-set structures)")),
    ("--path_to_output", dict(type=str, default=".", help="output directory (asked before it is created)")),
    ("--output_analysis", dict(action="store_true", help="accepted for compatibility; unused, as in the reference")),
    ("--predict_rotamers", dict(action="store_true", help="the model predicts 338 rotamer classes instead of 20 residues")),
    ("--is_structure_nmr", dict(action="store_true", help="merge the states of an NMR ensemble into a consensus")),
    ("--device", dict(type=int, default=0, help="HIP device index")),
    ("--devices", dict(type=str, default=None, help="comma-separated HIP device indices to spread the frames over")),
    ("--frames_per_call", dict(type=int, default=None, help="frames handed to a GPU per call (default 1024; 4096 for .hdf5 datasets inflated on the GPU)")),
)


def build_parser():
    parser = argparse.ArgumentParser(description="Residue / rotamer probabilities for every frame of a dataset (MI355X)")
    for flag, keywords in CLI_FLAGS:
        parser.add_argument(flag, **keywords)
    return parser


def _confirm_output_directory(directory: Path):
    if directory.exists():
        return
    print(f"{directory} does not exist yet - create it? (y/n)")
    if input() != "y":
        print("Nothing written.")
        sys.exit()
    directory.mkdir(parents=True, exist_ok=True)


def main(args):
    required = {"model": Path(args.path_to_model), "dataset": Path(args.path_to_dataset)}
    if args.path_to_blacklist:
        required["blacklist"] = Path(args.path_to_blacklist)
    _confirm_output_directory(Path(args.path_to_output))
    for what, path in required.items():
        assert path.exists(), f"No {what} at {path}"
    assert args.batch_size > 0, f"--batch_size must be positive, got {args.batch_size}"
    devices = [int(d) for d in args.devices.split(",")] if getattr(args, "devices", None) else None
    try:
        return _main_predict(args, required, devices)
    finally:
        du.release_device_memory()           # pooled batch buffers + the GPU decoder's scratch: a CLI run keeps nothing


def _main_predict(args, required, devices):
    return load_dataset_and_predict(
        [required["model"]], required["dataset"], batch_size=args.batch_size, start_batch=0,
        dataset_map_path=Path(args.path_to_datasetmap), blacklist=required.get("blacklist"),
        predict_rotamers=args.predict_rotamers, is_consensus=args.is_structure_nmr,
        path_to_output=Path(args.path_to_output), device=getattr(args, "device", 0), devices=devices,
        frames_per_call=getattr(args, "frames_per_call", None))


if __name__ == "__main__":
    main(build_parser().parse_args())
