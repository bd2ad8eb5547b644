"""Drop-in mirror of the reference's ``utils.alpha_nnet`` (alpha_nnet.py).

``AlphaNNet`` keeps ``AlphaNNet(model_name=None, input_shape=None)``, ``v(X)``, ``train``,
``copy_and_compile``, ``save`` and ``v_net`` (alpha_nnet.py:8-109).  Inference (``v``) runs the
hand-written HIP kernels of csrc/net.hip (MFMA 3x3 convolutions) through ``snake_engine.net.QNet``;
``v_device`` is the same call for callers that already hold the observations in HBM (the MCTS).
Weights are kept in the Keras layout/order so Keras-format checkpoints map one to one.
"""
import numpy as np
import torch

from snake_engine.net import QNet, glorot_uniform_weights, n_blocks_of


def lr_schedule(learning_rate=0.0001):
    """alpha_nnet.py:79-84: PiecewiseConstantDecay([20, 40, 60, 80, 100], [lr, lr/4, lr/16, lr/64, lr/256, 0.0]) of the
    optimizer's step count -- (boundaries, values); the products are formed exactly as the reference forms them"""
    boundaries = [20, 40, 60, 80, 100]
    values = [0.0] * (len(boundaries) + 1)
    n = learning_rate
    for i in range(len(boundaries)):
        values[i] = n
        n *= 0.25
    return boundaries, values


class _VNet:
    """the slice of the Keras Model object the reference's scripts touch (test_model.py:12, test_weights.py:7)"""

    def __init__(self, owner):
        self._o = owner

    def get_weights(self):
        return self._o._qnet.get_weights()

    def set_weights(self, ws):
        self._o._qnet.set_weights(ws)

    def summary(self):
        ws = self.get_weights()
        print(f"AlphaNNet Q-net: input {self._o.input_shape}, {n_blocks_of(ws)} residual blocks x 2 conv3x3(128), "
              f"head conv1x1 -> Dense(128) -> Dense(3, tanh); {sum(w.size for w in ws)} parameters")

    def save(self, path):
        from utils import checkpoint
        checkpoint.save_h5(path, self.get_weights(), self._o.input_shape)


class AlphaNNet:
    def __init__(self, model_name=None, input_shape=None, _weights=None, blocks=4):
        self.learning_rate = None
        self.lr_schedule = None
        if model_name:
            from utils import checkpoint
            weights, input_shape = checkpoint.load_h5(model_name)      # raises OSError when missing (pit.py:58 relies on it)
        elif input_shape:
            weights = _weights if _weights is not None else glorot_uniform_weights(tuple(input_shape), blocks,
                                                                                  seed=int(np.random.randint(1 << 31)))
        else:
            self.v_net = None
            return
        self.input_shape = tuple(int(v) for v in input_shape)
        self._qnet = QNet(weights, self.input_shape)
        self.v_net = _VNet(self)

    # ---- inference ------------------------------------------------------------------------------------
    def v_device(self, planes, mask=None):
        """planes: cuda float32 [n, h, w, 3]; mask: optional cuda uint8 [n, 3] from the observe kernel.
        Without a mask the obstacle test of alpha_nnet.py:63-76 is evaluated on the planes."""
        if mask is None:
            cy, cx = planes.shape[1] // 2, planes.shape[2] // 2
            thr = torch.tensor(0.04, dtype=torch.float32, device=planes.device)
            mask = torch.stack([planes[:, cy, cx - 1, 1] >= thr, planes[:, cy - 1, cx, 1] >= thr,
                                planes[:, cy, cx + 1, 1] >= thr], dim=1).to(torch.uint8).contiguous()
        # the split-f16 convolutions clamp inputs beyond the f16 range and say so: forward_guarded reads the flags after the
        # batch, widens the scale of a layer that clamped and evaluates the batch again -- what leaves here is within the
        # 1e-5 contract or an EngineError, for host callers (v) and for the search (Agent.make_moves) alike
        return self._qnet.forward_guarded(planes, mask)

    @property
    def guard(self):
        """the Q-net when its evaluations carry a range guard word (the search gates its tick kernels on it), else None"""
        return self._qnet if self._qnet.guard_ptr else None

    def v_device_unguarded(self, planes, mask):
        """the search's leaf evaluation: the forward and the asynchronous copy of the guard word, no synchronisation -- the
        caller MUST gate what it does with the result on `guard.guard_ptr` and check `guard.guard_tripped()` at its next
        synchronisation (snake_engine.mcts.DeviceMCTS does)"""
        q = self._qnet.forward(planes, mask)
        self._qnet.guard_post()
        return q

    def v(self, X):
        """alpha_nnet.py:61-73: list/array of (h, w, 3) float32 observations -> (N, 3) float32"""
        planes = torch.as_tensor(np.ascontiguousarray(np.array(X, dtype=np.float32)), device=self._qnet.device)
        return self.v_device(planes).cpu().numpy()

    def calibrate(self, planes):
        if not self._qnet.calibrated:
            self._qnet.calibrate(planes)

    def is_obstacle(self, value):
        return value >= 0.04

    # ---- training half --------------------------------------------------------------
    def train(self, X, Y, epochs=32, batch_size=2048):
        from utils import trainer_torch
        ws = trainer_torch.fit(self._qnet.get_weights(), self.input_shape, np.array(X, np.float32), np.array(Y, np.float32),
                               epochs, batch_size, self.lr_schedule)
        self._qnet.set_weights(ws)

    def copy_and_compile(self, learning_rate=0.0001, TPU=None):
        """alpha_nnet.py:78-106: a copy with Adam + PiecewiseConstantDecay([20,40,60,80,100] steps,
        [lr, lr/4, lr/16, lr/64, lr/256, 0.0])"""
        cp = AlphaNNet(input_shape=self.input_shape, _weights=self._qnet.get_weights())
        cp.learning_rate = learning_rate
        cp.lr_schedule = lr_schedule(learning_rate)
        return cp

    def save(self, name):
        self.v_net.save('models/' + name + '.h5')
