"""The per-process device block cache behind th_model_load / th_model_free (csrc/runtime.hip): blocks a closed model
gives back are parked for the next load (exact-size reuse), at most min(24 GB, an eighth of the device's memory) per device, the
blocks parked longest ago leave first, th_dev_trim empties it, and every allocator of the library trims it before it reports
TH_ENOMEM (th_malloc_retry)."""
import ctypes as C

import numpy as np
import pytest

from timed_hip import _lib, engine, synth

pytestmark = pytest.mark.gpu


def _info(lib, device):
    cached, cap, blocks = C.c_uint64(), C.c_uint64(), C.c_int()
    _lib.check(lib.th_dev_cache_info(device, C.byref(cached), C.byref(cap), C.byref(blocks)))
    return cached.value, cap.value, blocks.value


def test_cache_parks_reuses_evicts_oldest_and_trims(gpu, lib):
    lib.th_dev_trim(gpu)
    assert _info(lib, gpu)[0] == 0 and _info(lib, gpu)[2] == 0
    cap = _info(lib, gpu)[1]
    assert 0 < cap <= 24 << 30
    cfg, w = synth.timed_synth(20)
    frames = synth.synthetic_frames(3, seed=1)
    m = engine.HipFrameModel.from_keras(cfg, w, device=gpu)
    want = m.predict(frames)
    m.close()
    cached1, _, blocks1 = _info(lib, gpu)
    assert cached1 > 0 and blocks1 > 10                       # weights, arenas, rings of the closed handle are parked
    m = engine.HipFrameModel.from_keras(cfg, w, device=gpu)   # same sizes: taken back out of the cache
    assert _info(lib, gpu)[0] < cached1
    assert np.array_equal(m.predict(frames), want)
    m.close()
    assert _info(lib, gpu)[0] <= cap
    # models of other sizes do not accumulate for ever: whatever is parked stays under the cap, and the count is bounded
    for chunk in (64, 200, 333):
        m = engine.HipFrameModel.from_keras(cfg, w, device=gpu)
        m.set_chunk(chunk)
        assert np.array_equal(m.predict(frames), want)
        m.close()
        cached, cap2, blocks = _info(lib, gpu)
        assert cached <= cap2 and blocks <= 1024
    lib.th_dev_trim(gpu)
    assert _info(lib, gpu)[0] == 0 and _info(lib, gpu)[2] == 0
