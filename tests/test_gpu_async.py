"""th_predict_async / th_predict_wait and page-locked host memory (include/timed_hip.h): the pipelined form of
frame_model.predict(X_batch) (reference predict.py:142, called once per batch by the loop at :125-155).  The bar is
bit-identity with the synchronous call — the pipeline only reorders copies, never arithmetic."""
import ctypes as C

import numpy as np
import pytest

from oracle import cnn_oracle
from timed_hip import _lib, engine, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def small(gpu):
    cfg, weights = synth.timed_synth(20, widths=(8, 16, 16), side=9, in_channels=6)
    model = engine.HipFrameModel.from_keras(cfg, weights, device=gpu)
    yield cfg, weights, model
    model.close()


def _frames(n, seed, dtype=np.float32):
    f = synth.synthetic_frames(n, side=9, channels=6, atoms=40, seed=seed)
    return f.astype(dtype)


def test_async_equals_sync_and_oracle(small):
    cfg, weights, model = small
    batches = [_frames(n, 10 + i) for i, n in enumerate([7, 1, 33, 12])]
    want = [model.predict(b) for b in batches]
    pend = [model.predict_async(b) for b in batches]          # four tickets in flight at once
    got = [p.result() for p in pend]
    for g, w, b in zip(got, want, batches):
        assert np.array_equal(g, w)
        np.testing.assert_allclose(g, cnn_oracle.forward(cfg, weights, b), atol=5e-6, rtol=0)


def test_wait_out_of_order_and_ticket_reuse(small):
    _, _, model = small
    a, b = _frames(5, 1), _frames(9, 2)
    wa, wb = model.predict(a), model.predict(b)
    for _ in range(6):                                        # more rounds than there are tickets: slots are reused
        pa, pb = model.predict_async(a), model.predict_async(b)
        assert np.array_equal(pb.result(), wb)                # later ticket first
        assert np.array_equal(pa.result(), wa)
        assert np.array_equal(pa.result(), wa)                # result() is idempotent


def test_busy_after_four_tickets(small):
    _, _, model = small
    x = _frames(3, 3)
    pend = [model.predict_async(x) for _ in range(4)]
    with pytest.raises(_lib.TimedHipError) as e:
        model.predict_async(x)
    assert e.value.code == -7                                 # TH_EBUSY
    want = model_predict_after = None
    for p in pend:
        r = p.result()
        want = r if want is None else want
        assert np.array_equal(r, want)
    assert np.array_equal(model.predict(x), want)             # the model is usable again


def test_dropped_handle_returns_its_ticket(small):
    _, _, model = small
    x = _frames(4, 4)
    for _ in range(10):
        model.predict_async(x)                                # handle dropped immediately
    assert model.predict(x).shape == (4, 20)


def test_bad_ticket(small, lib):
    _, _, model = small
    assert lib.th_predict_wait(model._h, 99) == -1
    assert lib.th_predict_wait(model._h, 0) == -1             # not in flight
    assert b"not in flight" in lib.th_last_error()


@pytest.mark.parametrize("dtype", [np.float32, np.float64, np.uint8, np.float16])
def test_pinned_input_many_pieces(small, dtype):
    """page-locked frames (the truly asynchronous copy path), more pieces than ring buffers, every input dtype"""
    cfg, weights, model = small
    model.set_chunk(8)                                        # 50 frames -> 7 pieces through a ring of 3
    try:
        src = _frames(50, 5) if dtype != np.uint8 else synth.synthetic_frames(50, side=9, channels=6, atoms=40, seed=5, gaussian=False).astype(np.uint8)
        src = src.astype(dtype)
        X, owner = engine.pinned_empty(src.shape, dtype)
        X[...] = src
        p1 = model.predict_async(X)
        p2 = model.predict_async(src)                         # pageable copy of the same frames, queued behind it
        r1, r2 = p1.result(), p2.result()
        assert np.array_equal(r1, r2)
        np.testing.assert_allclose(r1, cnn_oracle.forward(cfg, weights, src), atol=5e-6, rtol=0)
        del X, p1
        owner.free()
    finally:
        model.set_chunk(1024)


def test_host_register_roundtrip(small, lib):
    _, _, model = small
    x = np.ascontiguousarray(_frames(16, 6))
    want = model.predict(x)
    _lib.check(lib.th_host_register(C.c_void_p(x.ctypes.data), x.nbytes))
    try:
        assert np.array_equal(model.predict(x), want)
    finally:
        _lib.check(lib.th_host_unregister(C.c_void_p(x.ctypes.data)))
    assert lib.th_host_register(None, 0) == -1


def test_empty_batch_async(small):
    _, _, model = small
    p = model.predict_async(np.empty((0, 9, 9, 9, 6), np.float32))
    assert p.result().shape == (0, 20)


def test_logits_async(small):
    _, _, model = small
    x = _frames(6, 7)
    assert np.array_equal(model.predict_async(x, logits=True).result(), model.predict(x, logits=True))


def test_submit_and_wait_on_different_threads_pinned(small):
    """predict.py waits on its writer thread while the main thread keeps submitting to the SAME handle, and
    with page-locked frames th_predict_async returns in microseconds — a slot released before its waiter has copied the
    rows out would be re-used underneath it.  Distinct batches (so a mixed-up slot cannot pass), pinned inputs, a
    waiter thread that lags behind the submitter; every batch must come back with ITS rows, bit for bit."""
    import queue
    import threading

    _, _, model = small
    n_batches, per = 64, 24
    src = _frames(n_batches * per, 21)
    want = model.predict(src).reshape(n_batches, per, 20)
    X, owner = engine.pinned_empty(src.shape, np.float32)
    X[...] = src
    q: "queue.Queue" = queue.Queue(maxsize=3)                  # <= 3 in the queue + 1 being waited on = 4 tickets
    got, errors = {}, []

    def waiter():
        try:
            while True:
                item = q.get()
                if item is None:
                    return
                i, pend = item
                got[i] = pend.result().copy()
        except Exception as e:                                 # pragma: no cover - reported by the assert below
            errors.append(e)

    th = threading.Thread(target=waiter)
    th.start()
    try:
        for i in range(n_batches):
            while True:
                try:
                    pend = model.predict_async(X[i * per:(i + 1) * per])
                    break
                except _lib.TimedHipError as e:                # all four slots still owned by the waiter: not an error
                    assert e.code == -7
            q.put((i, pend))
    finally:
        q.put(None)
        th.join()
    assert not errors, errors
    for i in range(n_batches):
        assert np.array_equal(got[i], want[i]), f"batch {i} came back with another batch's rows"
    del X
    owner.free()


def test_double_wait_on_one_ticket_is_refused(small, lib):
    _, _, model = small
    x = _frames(4, 8)
    out = np.empty((4, 20), np.float32)
    t = C.c_int(-1)
    _lib.check(lib.th_predict_async(model._h, x.ctypes.data, _lib.TH_F32, 4, out.ctypes.data, 0, C.byref(t)))
    _lib.check(lib.th_predict_wait(model._h, t.value))
    assert lib.th_predict_wait(model._h, t.value) == -1       # returned already: not in flight any more
    assert np.array_equal(out, model.predict(x))
