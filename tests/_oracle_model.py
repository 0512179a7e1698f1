"""Test double for timed_hip.engine.HipFrameModel on machines without a GPU: same surface (n_classes, predict,
predict_async().result(), close), arithmetic by the CPU oracle.  TEST INFRASTRUCTURE ONLY — lets the CPU suite drive
the real predict.py control flow (grouping, pipelining, sharding, writers); the product never imports this."""
import numpy as np

from oracle import cnn_oracle
from timed_hip import h5model


class _Done:
    def __init__(self, model, value):
        self._model, self._value, self._open = model, value, True
        model.in_flight += 1
        assert model.in_flight <= 4, "more than 4 tickets in flight on one model (th_predict_async would return TH_EBUSY)"

    def result(self):
        if self._open:
            self._open = False
            self._model.in_flight -= 1
        return self._value


class OracleModel:
    calls = []          # (device, n_frames) per predict call, for the tests

    def __init__(self, path, device=0):
        self.cfg, self.weights = h5model.read_keras_h5(str(path))
        self.device = device
        self.in_flight = 0
        layers = self.cfg["config"]["layers"]
        self.input_shape = tuple(layers[0]["config"]["batch_input_shape"][1:])
        probe = cnn_oracle.forward(self.cfg, self.weights, np.zeros((1, *self.input_shape), np.float32))
        self.n_classes = probe.shape[1]

    def predict(self, X):
        OracleModel.calls.append((self.device, len(X)))
        return cnn_oracle.forward(self.cfg, self.weights, np.asarray(X)).astype(np.float32)

    def predict_async(self, X):
        return _Done(self, self.predict(X))

    def close(self):
        pass


def load_model(path, device=0):
    return OracleModel(path, device=device)
