"""ctypes binding of libtimedhip.so (include/timed_hip.h).  No fallback: if the shared object is
missing or lacks a symbol this raises — the product never silently computes on the CPU."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TIMED_HIP_LIB", os.path.join(_HERE, "libtimedhip.so"))

TH_OK = 0
TH_EINVAL, TH_EIO, TH_EHIP, TH_EUNSUP, TH_ENOMEM, TH_ECOMM, TH_EBUSY = -1, -2, -3, -4, -5, -6, -7      # include/timed_hip.h
TH_F32, TH_F64, TH_U8, TH_BOOL, TH_F16 = 0, 1, 2, 3, 4
TH_LOAD_DEFAULT, TH_LOAD_NO_FUSE, TH_LOAD_NO_MFMA, TH_LOAD_KEEP_ALL = 0, 1, 2, 4
TH_PREDICT_DEFAULT, TH_PREDICT_LOGITS, TH_PREDICT_OUT_DEVICE, TH_PREDICT_IN_DEVICE = 0, 1, 2, 4
TH_RNG_HOST, TH_RNG_PHILOX, TH_RNG_MT19937, TH_RNG_MT_WORDS = 0, 1, 2, 3
TH_TEMPER_NONE, TH_TEMPER_POW, TH_TEMPER_PREPOWERED = 0, 1, 2
TH_COMM_ID_BYTES = 128

_vp, _i, _u, _i64, _u64, _d, _sz = C.c_void_p, C.c_int, C.c_uint, C.c_int64, C.c_uint64, C.c_double, C.c_size_t
_pi, _pd, _pi64 = C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_int64)

# name -> (restype, argtypes): the complete C ABI; tests check every entry resolves
PROTOTYPES = {
    "th_version": (_i, []),
    "th_last_error": (C.c_char_p, []),
    "th_device_count": (_i, [_pi]),
    "th_host_cpus": (_i, []),
    "th_device_info": (_i, [_i, C.c_char_p, _sz, C.c_char_p, _sz, _pi]),
    "th_model_load": (_i, [C.c_char_p, _i, _u, C.POINTER(_vp)]),
    "th_model_load_mem": (_i, [_vp, _sz, _i, _u, C.POINTER(_vp)]),
    "th_model_free": (None, [_vp]),
    "th_model_info": (_i, [_vp, C.POINTER(C.c_int * 4), _pi]),
    "th_model_cost": (_i, [_vp, _pd, _pd, _pi]),
    "th_model_set_chunk": (_i, [_vp, _i]),
    "th_predict": (_i, [_vp, _vp, _i, _i64, _vp, _u]),
    "th_predict_device": (_i, [_vp, _vp, _i, _i64, _vp, _u]),
    "th_predict_async": (_i, [_vp, _vp, _i, _i64, _vp, _u, _pi]),
    "th_predict_sparse_async": (_i, [_vp, _vp, _sz, _vp, _u, _pi]),
    "th_predict_wait": (_i, [_vp, _i]),
    "th_host_alloc": (_i, [_sz, C.POINTER(_vp)]),
    "th_host_free": (_i, [_vp]),
    "th_host_register": (_i, [_vp, _sz]),
    "th_host_unregister": (_i, [_vp]),
    "th_model_fetch": (_i, [_vp, C.c_char_p, _i64, _vp, _i64]),
    "th_model_profile": (_i, [_vp, _i]),
    "th_model_step_info": (_i, [_vp, _i, C.c_char_p, _sz, _pd, _pi64, _pd, _pd, _pd]),
    "th_model_knobs": (_i, [_vp, C.c_char_p, _sz]),
    "th_model_step_direct_flops": (_i, [_vp, _i, _pd]),
    "th_model_guard_info": (_i, [_vp, C.POINTER(C.c_int), _pd, _pd, C.c_char_p, _sz]),
    "th_dev_alloc": (_i, [_i, _sz, C.POINTER(_vp)]),
    "th_dev_free": (_i, [_i, _vp]),
    "th_dev_upload": (_i, [_i, _vp, _vp, _sz]),
    "th_dev_download": (_i, [_i, _vp, _vp, _sz]),
    "th_dev_sync": (_i, [_i]),
    "th_dev_synth_frames": (_i, [_i, _vp, _i64, _i, _i, _i, _u64]),
    "th_apply_temp": (_i, [_vp, _i64, _i, _d, _vp]),
    "th_apply_temp_on": (_i, [_i, _vp, _i64, _i, _d, _vp]),
    "th_sampler_create": (_i, [_i, C.POINTER(_vp)]),
    "th_sampler_free": (None, [_vp]),
    "th_sampler_load": (_i, [_vp, _vp, _i64, _i, _d, _i, _i, _vp]),
    "th_sampler_draw": (_i, [_vp, _i64, _pi64, _i64, _i, _u64, _u64, _vp, C.c_char_p, _vp, _vp, _vp, _vp]),
    "th_sampler_uniform_buffer": (_i, [_vp, _sz, C.POINTER(C.c_void_p)]),
    "th_sampler_run": (_i, [_vp, _vp, _i64, _i, _i, _i64, _pi64, _i64, _i, _u64, _u64, _vp, C.c_char_p, C.c_uint, C.POINTER(C.c_void_p), _pi64]),
    "th_sample": (_i, [_vp, _i64, _i, _i64, _d, _i, _u64, _vp, _vp]),
    "th_sample_ex": (_i, [_vp, _i64, _i, _i64, _d, _i, _u64, _u64, _vp, _vp, _vp, C.c_char_p, _vp, _vp, _i]),
    "th_comm_unique_id": (_i, [C.c_char_p]),
    "th_comm_init": (_i, [C.c_char_p, _i, _i, _i, C.POINTER(_vp)]),
    "th_comm_free": (None, [_vp]),
    "th_comm_gather_rows": (_i, [_vp, _vp, _pi64, _i, _i, _vp]),
    "th_comm_barrier": (_i, [_vp]),
    "th_comm_stats": (_i, [_vp, _pi64]),
    "th_voxelise": (_i, [_i, _vp, _vp, _vp, _i64, _vp, _i64, _i, C.c_float, _i, _i, _vp, _i]),
    "th_mt19937_rand": (_i, [_vp, _pi, _i64, _vp]),
    "th_mt19937_words": (_i, [_vp, _pi, _i64, _vp]),
    "th_format_csv": (_i64, [_vp, _i, _i64, _i64, _vp, _i64]),
    "th_format_csv_device": (_i64, [_i, _vp, _i64, _i64, _vp, _i64]),
    "th_format_csv_device_release": (_i, []),
    "th_csv_shape": (_i, [C.c_char_p, _i64, C.c_char, _pi64, _pi, _pi]),
    "th_csv_fill": (_i, [C.c_char_p, _i64, C.c_char, _i64, _i, _i, _vp]),
    "th_argmax_letters": (_i, [_vp, _i, _i64, _i64, C.c_char_p, _vp, _vp]),
    "th_h5_read_chunked": (_i, [_vp, _i64, _i64, _i64, _pi64, C.POINTER(_vp), _i, _pi64, _pi64, _i, _i, _pi, _i]),
    "th_h5_read_chunked_as": (_i, [_vp, _i64, _i64, _i64, _pi64, C.POINTER(_vp), _i, _pi64, _pi64, _i, _i, _pi, _i, _i]),
    "th_h5_read_contiguous_as": (_i, [_vp, _i64, _i64, _i64, _pi64, C.POINTER(_vp), _i64, _i, _i]),
    "th_h5_decode_device": (_i, [_vp, _i64, _i64, _i64, _pi64, _i, _pi64, _pi64, _i, _i, _pi, _i, _i, _vp]),
    "th_h5_release_scratch": (_i, [_i]),
    "th_dev_trim": (_i, [_i]),
    "th_dev_cache_info": (_i, [_i, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), _pi]),
    "th_inflate_many": (_i, [_i, _vp, _i64, _i64, _pi64, _pi64, _pi64, _pi64, _vp, _i64, _i, _pi]),
    "th_h5_group_links": (_i, [_vp, _i64, _i64, _i64, _i64, _i64, _vp, _i64, _pi64, _i64, _pi64, _pi64]),
    "th_h5_resolve": (_i, [_vp, _i64, _i64, _i64, _pi64, C.c_char_p, _vp, _i, C.c_char_p, _vp, _i, _pi64, _pi64, _pi, _i]),
}

_lib = None


class TimedHipError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"timed_hip error {code}: {msg}")
        self.code = code


def load() -> C.CDLL:
    """Load libtimedhip.so and bind every prototype.  Raises if the library or a symbol is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `python __graft_entry__.py` (hipcc --offload-arch=gfx950). "
            "There is no CPU fallback.")
    # the host driver only supports dmabuf IPC: RCCL (and any cross-process device-memory sharing) needs this set
    # before the HIP runtime initialises; an explicit setting in the environment wins
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export the symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int) -> None:
    if rc != TH_OK:
        raise TimedHipError(rc, load().th_last_error().decode(errors="replace"))


def device_count() -> int:
    n = C.c_int(0)
    rc = load().th_device_count(C.byref(n))
    return n.value if rc == TH_OK else 0


def device_info(device: int = 0):
    name = C.create_string_buffer(256)
    arch = C.create_string_buffer(256)
    cus = C.c_int(0)
    check(load().th_device_info(device, name, 256, arch, 256, C.byref(cus)))
    return name.value.decode(), arch.value.decode(), cus.value
