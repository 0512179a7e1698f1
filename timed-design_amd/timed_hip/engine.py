"""HipFrameModel — drop-in for the Keras model object at the reference's CNN seam.

    reference predict.py:121   frame_model = tf.keras.models.load_model(Path(m))
    reference predict.py:142   y_pred_batch = frame_model.predict(X_batch)

``HipFrameModel.load(path)`` accepts a ``.pack`` (THPK0001) or, when an HDF5 reader is available,
a Keras ``.h5``; ``predict(X)`` takes the same ``X[B,D,H,W,C]`` ndarray ``load_batch`` builds
(float64 when voxels_as_gaussian, bool otherwise — reference design_utils/utils.py:518-521; any of
float32/float64/float16/uint8/bool is accepted) and returns a NEW float32 ``[B, n_classes]`` array,
exactly like Keras.  Errors surface as Python exceptions.  All arithmetic happens in the HIP
library; there is no CPU path.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Optional

import numpy as np

from . import _lib, keras_config, pack

_DTYPES = {
    np.dtype(np.float32): _lib.TH_F32, np.dtype(np.float64): _lib.TH_F64, np.dtype(np.uint8): _lib.TH_U8,
    np.dtype(np.bool_): _lib.TH_BOOL, np.dtype(np.float16): _lib.TH_F16,
}


class HipFrameModel:
    def __init__(self, pack_bytes: bytes, device: int = 0, flags: int = _lib.TH_LOAD_DEFAULT, name: str = "model"):
        self._lib = _lib.load()
        self._h = C.c_void_p()
        self.device = device
        self.name = name
        buf = (C.c_char * len(pack_bytes)).from_buffer_copy(pack_bytes)
        _lib.check(self._lib.th_model_load_mem(buf, len(pack_bytes), device, flags, C.byref(self._h)))
        dims = (C.c_int * 4)()
        ncls = C.c_int()
        _lib.check(self._lib.th_model_info(self._h, C.byref(dims), C.byref(ncls)))
        self.input_shape = tuple(dims)
        self.n_classes = ncls.value
        self.device_name, self.device_arch, self.device_cus = _lib.device_info(device)
        self.chunk = 1024           # frames per internal pass (th_model_set_chunk); the library's default

    # ---- constructors -----------------------------------------------------------------------
    @classmethod
    def from_keras(cls, model_config, weights: Dict[str, list], device: int = 0, flags: int = 0, name: str = "model"):
        return cls(pack.keras_to_pack(model_config, weights), device=device, flags=flags, name=name)

    @classmethod
    def load(cls, path, device: int = 0, flags: int = 0):
        path = os.fspath(path)
        stem = os.path.splitext(os.path.basename(path))[0]
        with open(path, "rb") as f:
            head = f.read(8)
        if head == pack.MAGIC:
            with open(path, "rb") as f:
                return cls(f.read(), device=device, flags=flags, name=stem)
        if head == b"\x89HDF\r\n\x1a\n":
            from . import h5model  # Keras .h5 -> (model_config, weights); pure-Python HDF5 reader
            cfg, weights = h5model.read_keras_h5(path)
            return cls.from_keras(cfg, weights, device=device, flags=flags, name=stem)
        raise ValueError(f"{path}: neither a THPK0001 pack nor an HDF5 (.h5) Keras model")

    # ---- Keras-compatible surface --------------------------------------------------------------
    def predict(self, X, batch_size: Optional[int] = None, verbose=0, logits: bool = False, **_ignored) -> np.ndarray:
        """X[B,D,H,W,C] -> float32 [B,n_classes].  ``batch_size``/``verbose`` are accepted for
        signature compatibility with ``Model.predict``; chunking is internal."""
        if isinstance(X, SparseFrames):
            return self.predict_async(X, logits=logits).result()
        X = np.asarray(X)
        if X.ndim != 5 or tuple(X.shape[1:]) != self.input_shape:
            raise ValueError(f"expected frames of shape (B, {', '.join(map(str, self.input_shape))}), got {X.shape}")
        dt = _DTYPES.get(X.dtype)
        if dt is None:
            X = X.astype(np.float32)
            dt = _lib.TH_F32
        X = np.ascontiguousarray(X)
        n = X.shape[0]
        width = self.logits_width if logits else self.n_classes
        out = np.empty((n, width), dtype=np.float32)
        if n:
            _lib.check(self._lib.th_predict(self._h, X.ctypes.data, dt, n, out.ctypes.data,
                                            _lib.TH_PREDICT_LOGITS if logits else 0))
        return out

    __call__ = predict

    def predict_async(self, X, logits: bool = False) -> "PendingPrediction":
        if isinstance(X, DeviceFrames):
            return self.predict_async_frames_on_device(X, logits=logits)
        if isinstance(X, SparseFrames):
            return self._predict_async_sparse(X, None, logits=logits)
        return self._predict_async_host(X, logits=logits)

    def _predict_async_sparse(self, X: "SparseFrames", d_out, logits: bool = False) -> "PendingPrediction":
        """A batch in the sparse transport form (SparseFrames: bitmap + stored values, timed_hip/framepack.py): a tenth of the
        bytes of dense Gaussian frames cross PCIe, the dense frames are rebuilt bit for bit on the device (th_predict_sparse_async)."""
        if tuple(X.shape[1:]) != self.input_shape:
            raise ValueError(f"expected frames of shape (B, {', '.join(map(str, self.input_shape))}), got {X.shape}")
        blob = X.blob()
        n = X.shape[0]
        flags = _lib.TH_PREDICT_LOGITS if logits else 0
        out = None
        if d_out is None:
            out = np.empty((n, self.logits_width if logits else self.n_classes), dtype=np.float32)
        else:
            flags |= _lib.TH_PREDICT_OUT_DEVICE
        ticket = C.c_int(-1)
        _lib.check(self._lib.th_predict_sparse_async(self._h, blob.ctypes.data, blob.nbytes,
                                                     C.c_void_p(d_out) if d_out is not None else out.ctypes.data, flags, C.byref(ticket)))
        return PendingPrediction(self, ticket.value, (X, blob), out)

    def _predict_async_host(self, X, logits: bool = False) -> "PendingPrediction":
        """Queue ``X`` (copy to the device, kernels, copy of the probabilities back) and return at once; the
        returned handle's ``result()`` blocks until that batch is done and gives the float32 [B, n_classes] array.
        Up to four batches may be in flight per model; they finish in submission order.  The caller's loop (load the
        next batch, format the previous one) then runs under the GPU's work — reference predict.py:125-155 is strictly
        sequential.  ``X`` is held by the handle until ``result()``; batches built in ``pinned_empty`` memory are
        fetched by the DMA engine without blocking this call."""
        X = np.asarray(X)
        if X.ndim != 5 or tuple(X.shape[1:]) != self.input_shape:
            raise ValueError(f"expected frames of shape (B, {', '.join(map(str, self.input_shape))}), got {X.shape}")
        dt = _DTYPES.get(X.dtype)
        if dt is None:
            X = X.astype(np.float32)
            dt = _lib.TH_F32
        X = np.ascontiguousarray(X)
        n = X.shape[0]
        out = np.empty((n, self.logits_width if logits else self.n_classes), dtype=np.float32)
        ticket = C.c_int(-1)
        _lib.check(self._lib.th_predict_async(self._h, X.ctypes.data, dt, n, out.ctypes.data,
                                              _lib.TH_PREDICT_LOGITS if logits else 0, C.byref(ticket)))
        return PendingPrediction(self, ticket.value, X, out)

    def predict_async_device(self, X, d_out: int) -> "PendingPrediction":
        """As predict_async, but the probability rows are left in device memory at address ``d_out`` (this model's
        device, room for len(X) * n_classes floats) — e.g. a shard buffer that th_comm_gather_rows sends over xGMI.
        ``result()`` returns None once the rows are there."""
        if isinstance(X, SparseFrames):
            return self._predict_async_sparse(X, d_out)
        if isinstance(X, DeviceFrames):          # frames on the device already (GPU-inflated .hdf5 batches): nothing is copied
            if tuple(X.shape[1:]) != self.input_shape or X.device != self.device:
                raise ValueError(f"device frames of shape {X.shape} on device {X.device} do not fit this model")
            ticket = C.c_int(-1)
            _lib.check(self._lib.th_predict_async(self._h, C.c_void_p(X.ptr), _DTYPES[np.dtype(X.dtype)], X.shape[0], C.c_void_p(d_out),
                                                  _lib.TH_PREDICT_OUT_DEVICE | _lib.TH_PREDICT_IN_DEVICE, C.byref(ticket)))
            return PendingPrediction(self, ticket.value, X, None)
        X = np.asarray(X)
        if X.ndim != 5 or tuple(X.shape[1:]) != self.input_shape:
            raise ValueError(f"expected frames of shape (B, {', '.join(map(str, self.input_shape))}), got {X.shape}")
        dt = _DTYPES.get(X.dtype)
        if dt is None:
            X = X.astype(np.float32)
            dt = _lib.TH_F32
        X = np.ascontiguousarray(X)
        ticket = C.c_int(-1)
        _lib.check(self._lib.th_predict_async(self._h, X.ctypes.data, dt, X.shape[0], C.c_void_p(d_out),
                                              _lib.TH_PREDICT_OUT_DEVICE, C.byref(ticket)))
        return PendingPrediction(self, ticket.value, X, None)

    def predict_async_frames_on_device(self, frames: "DeviceFrames", logits: bool = False) -> "PendingPrediction":
        """As predict_async for a batch that is in this device's memory already (DeviceFrames, e.g. from
        design_utils.utils.load_batch_device: frames inflated on the GPU): no host->device copy at all."""
        if tuple(frames.shape[1:]) != self.input_shape:
            raise ValueError(f"expected frames of shape (B, {', '.join(map(str, self.input_shape))}), got {frames.shape}")
        if frames.device != self.device:
            raise ValueError(f"frames live on device {frames.device}, the model on {self.device}")
        n = frames.shape[0]
        out = np.empty((n, self.logits_width if logits else self.n_classes), dtype=np.float32)
        ticket = C.c_int(-1)
        _lib.check(self._lib.th_predict_async(self._h, C.c_void_p(frames.ptr), _DTYPES[np.dtype(frames.dtype)], n, out.ctypes.data,
                                              _lib.TH_PREDICT_IN_DEVICE | (_lib.TH_PREDICT_LOGITS if logits else 0), C.byref(ticket)))
        return PendingPrediction(self, ticket.value, frames, out)

    @property
    def logits_width(self) -> int:
        return self.n_classes

    def predict_device(self, d_frames: int, n: int, d_probs: int, dtype: int = _lib.TH_F32, logits: bool = False):
        """Frames and outputs already resident in device memory (raw device addresses as ints)."""
        _lib.check(self._lib.th_predict_device(self._h, C.c_void_p(d_frames), dtype, n, C.c_void_p(d_probs),
                                               _lib.TH_PREDICT_LOGITS if logits else 0))

    # ---- introspection / tuning ----------------------------------------------------------------
    def set_chunk(self, frames: int):
        _lib.check(self._lib.th_model_set_chunk(self._h, int(frames)))
        self.chunk = int(frames)

    def cost(self):
        a, e, n = C.c_double(), C.c_double(), C.c_int()
        _lib.check(self._lib.th_model_cost(self._h, C.byref(a), C.byref(e), C.byref(n)))
        return dict(algo_flops=a.value, exec_flops=e.value, n_steps=n.value)

    def profile(self, mode=True):
        """0/False off; 1/True time every plan step; 2 time only the step with the most FLOPs (cheap)."""
        _lib.check(self._lib.th_model_profile(self._h, int(mode)))

    def steps(self) -> List[dict]:
        out = []
        for i in range(self.cost()["n_steps"]):
            label = C.create_string_buffer(640)
            ms, fl, ef, by = C.c_double(), C.c_double(), C.c_double(), C.c_double()
            ln = C.c_int64()
            _lib.check(self._lib.th_model_step_info(self._h, i, label, 640, C.byref(ms), C.byref(ln), C.byref(fl),
                                                    C.byref(ef), C.byref(by)))
            df = C.c_double()
            _lib.check(self._lib.th_model_step_direct_flops(self._h, i, C.byref(df)))
            out.append(dict(label=label.value.decode(), ms=ms.value, launches=ln.value, flops=fl.value,
                            exec_flops=ef.value, bytes=by.value, direct_flops=df.value if df.value >= 0 else None))
        return out

    def guard(self) -> dict:
        """what the load-time guard found: state 0 not run / 1 passed / 2 tripped, the kept plan's max |dlogit| against the
        direct fp32 plan on the guard frames, the logit scale, and a note when it tripped (include/timed_hip.h)"""
        st, d, sc = C.c_int(), C.c_double(), C.c_double()
        note = C.create_string_buffer(512)
        _lib.check(self._lib.th_model_guard_info(self._h, C.byref(st), C.byref(d), C.byref(sc), note, 512))
        return dict(state=st.value, max_dlogit=d.value, logit_scale=sc.value, note=note.value.decode())

    def knobs(self) -> str:
        """the non-default TH_* knobs this handle was loaded under ("" when none): read once at load, never again"""
        buf = C.create_string_buffer(1024)
        _lib.check(self._lib.th_model_knobs(self._h, buf, 1024))
        return buf.value.decode()

    def fetch(self, layer_name: str, n: int, shape) -> np.ndarray:
        out = np.empty((n, *shape), dtype=np.float32)
        _lib.check(self._lib.th_model_fetch(self._h, layer_name.encode(), n, out.ctypes.data, out.size))
        return out

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._lib.th_model_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class PendingPrediction:
    """A batch in flight (HipFrameModel.predict_async).  Keeps the input alive until the result is taken."""

    def __init__(self, model: HipFrameModel, ticket: int, frames: np.ndarray, out: np.ndarray):
        self._model, self._ticket, self._frames, self._out = model, ticket, frames, out

    def result(self) -> np.ndarray:
        if self._ticket is not None:
            t, self._ticket = self._ticket, None
            try:
                _lib.check(self._model._lib.th_predict_wait(self._model._h, t))
            finally:
                self._frames = None
        return self._out

    def __del__(self):          # a dropped handle must still return its ticket to the model
        try:
            if self._ticket is not None and self._model._h:
                self._model._lib.th_predict_wait(self._model._h, self._ticket)
        except Exception:
            pass


class PinnedBuffer:
    """Page-locked host memory from the C ABI (th_host_alloc) exposed as NumPy arrays: frame batches built here are
    copied to the device asynchronously (no staging pass through pageable memory)."""

    def __init__(self, nbytes: int):
        self._lib = _lib.load()
        self.nbytes = int(nbytes)
        p = C.c_void_p()
        _lib.check(self._lib.th_host_alloc(self.nbytes, C.byref(p)))
        self.ptr = p.value
        self._raw = (C.c_char * self.nbytes).from_address(self.ptr)

    def array(self, shape, dtype, offset: int = 0) -> np.ndarray:
        dtype = np.dtype(dtype)
        count = int(np.prod(shape))
        if offset + count * dtype.itemsize > self.nbytes:
            raise ValueError("PinnedBuffer too small for the requested array")
        a = np.frombuffer(self._raw, dtype=dtype, count=count, offset=offset).reshape(shape)
        return a          # holds a reference to self._raw; keep the PinnedBuffer alive while arrays are in use

    def free(self):
        if self.ptr:
            self._raw = None
            self._lib.th_host_free(C.c_void_p(self.ptr))
            self.ptr = 0

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


SPARSE_MAGIC = b"THSPF001"


def sparse_blob_layout(n: int, words: int, n_values: int):
    """byte offsets of the sections of a THSPF001 blob (include/timed_hip.h): (ranks, bitmaps, values, total)"""
    o_rank = 32
    o_bits = (o_rank + (n + 1) * 8 + 15) // 16 * 16
    o_val = o_bits + n * words * 4
    return o_rank, o_bits, o_val, o_val + n_values * 4


class SparseFrames:
    """A batch of float32 frames in the sparse transport form: ``bits`` [n, W] uint32 (bit k of word w set <=> element 32 w + k
    is stored), ``rank`` [n + 1] uint64 (stored elements in front of frame i, relative to any base) and ``values`` (the stored
    elements of the n frames in order).  Lossless: every element whose bit pattern is not +0.0 is stored.  ``blob()`` is the
    THSPF001 buffer th_predict_sparse_async reads — built into ``out`` (e.g. a page-locked slot of a BlobRing) when given."""

    def __init__(self, bits, rank, values, frame_shape):
        self.bits, self.rank, self.values = bits, rank, values
        self.frame_shape = tuple(int(d) for d in frame_shape)
        self.shape = (int(bits.shape[0]),) + self.frame_shape
        self.dtype = np.dtype(np.float32)
        self._blob = None

    def __len__(self):
        return self.shape[0]

    @property
    def n_values(self) -> int:
        return int(self.rank[-1] - self.rank[0]) if len(self.rank) else 0

    @property
    def blob_bytes(self) -> int:
        return sparse_blob_layout(self.shape[0], int(self.bits.shape[1]), self.n_values)[3]

    @property
    def dense_bytes(self) -> int:
        return int(np.prod(self.shape)) * 4

    def blob(self, out: Optional[np.ndarray] = None, pool=None, threads: int = 1) -> np.ndarray:
        if out is None and self._blob is not None:
            return self._blob
        n, W, nv = self.shape[0], int(self.bits.shape[1]), self.n_values
        o_rank, o_bits, o_val, total = sparse_blob_layout(n, W, nv)
        buf = np.empty(total + 16, np.uint8) if out is None else out
        if buf.nbytes < total or buf.ctypes.data % 16:
            raise ValueError("sparse blob: the target buffer is too small or not 16-byte aligned")
        buf[:8] = np.frombuffer(SPARSE_MAGIC, np.uint8)
        buf[8:24].view(np.uint32)[:] = (n, int(np.prod(self.frame_shape)), W, 4)
        buf[24:32].view(np.uint64)[0] = nv
        buf[o_rank:o_rank + (n + 1) * 8].view(np.uint64)[:] = self.rank
        dst_bits = buf[o_bits:o_bits + n * W * 4].view(np.uint32).reshape(n, W)
        dst_val = buf[o_val:o_val + nv * 4].view(np.float32)
        src_val = self.values[:nv]
        if pool is not None and threads > 1 and nv > (1 << 20):       # NumPy copies release the interpreter lock
            cuts = [nv * i // threads for i in range(threads + 1)]
            jobs = [pool.submit(np.copyto, dst_val[a:b], src_val[a:b]) for a, b in zip(cuts[:-1], cuts[1:]) if b > a]
            jobs.append(pool.submit(np.copyto, dst_bits, self.bits))
            for j in jobs:
                j.result()
        else:
            np.copyto(dst_bits, self.bits)
            np.copyto(dst_val, src_val)
        view = buf[:total]
        if out is None:
            self._blob = view
        return view

    def staged(self, blob: np.ndarray) -> "SparseFrames":
        """the same batch with its blob already built in (page-locked) memory owned by somebody else"""
        sf = SparseFrames(self.bits, self.rank, self.values, self.frame_shape)
        sf._blob = blob
        return sf

    def dense(self) -> np.ndarray:
        """the frames as an ordinary [n, *frame_shape] float32 array (host expansion: tests, CPU-side consumers)"""
        n, E = self.shape[0], int(np.prod(self.frame_shape))
        mask = np.unpackbits(np.ascontiguousarray(self.bits).view(np.uint8), axis=1, bitorder="little")[:, :E].astype(bool)
        out = np.zeros((n, E), np.float32)
        out[mask] = np.asarray(self.values[:self.n_values])
        return out.reshape(self.shape)

    @classmethod
    def from_dense(cls, frames: np.ndarray) -> "SparseFrames":
        frames = np.ascontiguousarray(frames, dtype=np.float32)
        n = frames.shape[0]
        E = int(np.prod(frames.shape[1:]))
        W = ((E + 31) // 32 + 3) // 4 * 4
        flat = frames.reshape(n, E)
        mask = flat.view(np.uint32) != 0
        padded = np.zeros((n, W * 32), bool)
        padded[:, :E] = mask
        bits = np.packbits(padded, axis=1, bitorder="little").view(np.uint32).reshape(n, W)
        rank = np.concatenate([[0], np.cumsum(mask.sum(1, dtype=np.int64))]).astype(np.uint64)
        return cls(bits, rank, flat[mask], frames.shape[1:])


class BlobRing:
    """Page-locked byte buffers for sparse batches between a memory-mapped pack and th_predict_sparse_async — StagingRing's
    counterpart for variable-size blobs: a slot grows (and is re-registered) when a batch does not fit."""

    def __init__(self, slots: int, threads: int = 8):
        import queue
        from concurrent.futures import ThreadPoolExecutor
        self._lib = _lib.load()
        self._arrays = [None] * slots
        self._pinned = [False] * slots
        self._free = queue.SimpleQueue()
        for i in range(slots):
            self._free.put(i)
        self._threads = max(1, int(threads))
        self._pool = ThreadPoolExecutor(max_workers=self._threads, thread_name_prefix="stage_sparse")
        self.enabled = True

    def stage(self, X: SparseFrames):
        if not self.enabled or len(X) == 0:
            return X, None
        need = X.blob_bytes
        slot = self._free.get()
        try:
            a = self._arrays[slot]
            if a is None or a.nbytes < need:
                if a is not None and self._pinned[slot]:
                    self._lib.th_host_unregister(C.c_void_p(a.ctypes.data))
                    self._pinned[slot] = False
                raw = np.empty(need + need // 8 + 4096 + 16, np.uint8)
                off = (-raw.ctypes.data) % 16
                a = self._arrays[slot] = raw[off:]
            blob = X.blob(out=a, pool=self._pool, threads=self._threads)
            if not self._pinned[slot]:
                if self._lib.th_host_register(C.c_void_p(a.ctypes.data), a.nbytes) != 0:
                    self.enabled = False
                    self._free.put(slot)
                    return X, None
                self._pinned[slot] = True
            return X.staged(blob), slot
        except BaseException:
            self._free.put(slot)
            raise

    def release(self, slot) -> None:
        if slot is not None:
            self._free.put(slot)

    def close(self) -> None:
        if self._pool is None:
            return
        self._pool.shutdown(wait=True)
        self._pool = None
        for i, a in enumerate(self._arrays):
            if a is not None and self._pinned[i]:
                self._lib.th_host_unregister(C.c_void_p(a.ctypes.data))
                self._pinned[i] = False
        self._arrays = [None] * len(self._arrays)
        self.enabled = False


class StagingRing:
    """A ring of page-locked host buffers between a memory-mapped dataset and th_predict_async (no reference counterpart: the
    reference hands Keras freshly built NumPy batches, utils.py:487-530).  ``stage(rows)`` copies the rows into the next free
    slot with a few threads (NumPy copies release the interpreter lock) and returns (staged array, slot); the caller gives the
    slot back with ``release(slot)`` once the prediction that read it has completed.

    Why: a host->device copy straight from never-touched mapped pages runs at 165-177 k float32 frames/s on this box (the
    driver pins every new page on its way), from the same pages a second time or from page-locked memory at 238-248 k =
    the PCIe rate (tools/exp_h2d_pack.py).  A slot is ordinary memory made resident by its first fill (the copy threads fault it
    in side by side) and page-locked afterwards with th_host_register, 5 ms per 227 MB — hipHostMalloc of the same slot takes
    45 ms.  If registration fails (no device, limits) the ring is switched off: ``stage`` returns (rows, None)."""

    def __init__(self, slots: int, rows: int, frame_shape, dtype, threads: int = 8):
        import queue
        from concurrent.futures import ThreadPoolExecutor
        self._lib = _lib.load()
        self._shape = (int(rows),) + tuple(int(d) for d in frame_shape)
        self._dtype = np.dtype(dtype)
        self._arrays = [None] * slots
        self._pinned = [False] * slots
        self._free = queue.SimpleQueue()
        for i in range(slots):
            self._free.put(i)
        self._threads = max(1, int(threads))
        self._pool = ThreadPoolExecutor(max_workers=self._threads, thread_name_prefix="stage_frames")
        self.enabled = True

    def stage(self, rows: np.ndarray):
        n = len(rows)
        if not self.enabled or n == 0 or n > self._shape[0] or rows.dtype != self._dtype or tuple(rows.shape[1:]) != self._shape[1:]:
            return rows, None
        slot = self._free.get()
        try:
            a = self._arrays[slot]
            if a is None:
                a = self._arrays[slot] = np.empty(self._shape, self._dtype)
            dst = a[:n]
            cuts = [n * i // self._threads for i in range(self._threads + 1)]
            jobs = [self._pool.submit(np.copyto, dst[lo:hi], rows[lo:hi]) for lo, hi in zip(cuts[:-1], cuts[1:]) if hi > lo]
            for j in jobs:
                j.result()
            if not self._pinned[slot]:
                if self._lib.th_host_register(C.c_void_p(a.ctypes.data), a.nbytes) != 0:
                    # staging through pageable memory would only add a copy: hand out the caller's rows from here on
                    self.enabled = False
                    self._free.put(slot)
                    return rows, None
                self._pinned[slot] = True
            return dst, slot
        except BaseException:
            self._free.put(slot)
            raise

    def release(self, slot) -> None:
        if slot is not None:
            self._free.put(slot)

    def close(self) -> None:
        """Unlock and drop the slots: only when no copy from them is in flight any more."""
        if self._pool is None:
            return
        self._pool.shutdown(wait=True)
        self._pool = None
        for i, a in enumerate(self._arrays):
            if a is not None and self._pinned[i]:
                self._lib.th_host_unregister(C.c_void_p(a.ctypes.data))
                self._pinned[i] = False
        self._arrays = [None] * len(self._arrays)
        self.enabled = False


def pinned_empty(shape, dtype=np.float32):
    """(array, owner): an uninitialised page-locked array; drop both to release it."""
    dtype = np.dtype(dtype)
    buf = PinnedBuffer(max(1, int(np.prod(shape)) * dtype.itemsize))
    return buf.array(shape, dtype), buf


class DeviceFrames:
    """A batch of frames resident in device memory: [n, D, H, W, C] of ``dtype`` at ``ptr`` inside ``buffer`` (kept alive by
    this object).  ``len()`` is the number of frames; a pool may hand the buffer out again once ``release`` was called."""

    def __init__(self, buffer: "DeviceBuffer", shape, dtype, on_release=None):
        self.buffer, self.shape, self.dtype = buffer, tuple(int(x) for x in shape), np.dtype(dtype)
        self.ptr, self.device = buffer.ptr, buffer.device
        self._on_release = on_release

    def __len__(self):
        return self.shape[0]

    def release(self):
        cb, self._on_release = self._on_release, None
        if cb is not None:
            cb(self.buffer)

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


class DeviceBuffer:
    """Raw device allocation through the C ABI (bench / multi-GPU plumbing; no torch needed)."""

    def __init__(self, nbytes: int, device: int = 0):
        self._lib = _lib.load()
        self.device, self.nbytes = device, int(nbytes)
        p = C.c_void_p()
        _lib.check(self._lib.th_dev_alloc(device, self.nbytes, C.byref(p)))
        self.ptr = p.value

    def upload(self, arr: np.ndarray, offset: int = 0):
        arr = np.ascontiguousarray(arr)
        assert offset + arr.nbytes <= self.nbytes
        _lib.check(self._lib.th_dev_upload(self.device, C.c_void_p(self.ptr + offset), arr.ctypes.data, arr.nbytes))

    def download(self, shape, dtype, offset: int = 0) -> np.ndarray:
        out = np.empty(shape, dtype=dtype)
        assert offset + out.nbytes <= self.nbytes
        _lib.check(self._lib.th_dev_download(self.device, out.ctypes.data, C.c_void_p(self.ptr + offset), out.nbytes))
        return out

    def free(self):
        if self.ptr:
            self._lib.th_dev_free(self.device, C.c_void_p(self.ptr))
            self.ptr = 0

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def load_model(path, device: int = 0) -> HipFrameModel:
    """Name-compatible with ``tf.keras.models.load_model`` (reference predict.py:121)."""
    return HipFrameModel.load(path, device=device)
