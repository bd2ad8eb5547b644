"""CPU tests of the host-side logic of the Q-net wrapper (no GPU, no HIP library): weight list layout and the
power-of-two activation scales the split-f16 convolution kernel is given."""
import math
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "alphasnake-zero_amd"))


def _net():
    import importlib
    return importlib.import_module("snake_engine.net")


def test_glorot_weight_list_has_the_keras_layout():
    net = _net()
    ws = net.glorot_uniform_weights((21, 21, 3), blocks=4, seed=0)
    assert net.n_blocks_of(ws) == 4 and len(ws) == 14 + 10 * 4
    assert ws[0].shape == (3, 3, 3, 128) and ws[5].shape == (3, 3, 128, 128)
    assert ws[-4].shape == (441, 128) and ws[-2].shape == (128, 3)
    assert sum(int(np.prod(w.shape)) for w in ws) == 1244807      # parameters incl. BN statistics


def test_activation_scales_are_powers_of_two_with_headroom():
    net = _net()
    ws = net.glorot_uniform_weights((21, 21, 3), blocks=4, seed=0)
    sc = net.activation_scales(ws)
    assert len(sc) == 8
    # gen-0 net: gamma 1, beta 0 -> stem bound 8, shortcut bounds add 8 per block
    bounds = [8, 8, 16, 8, 24, 8, 32, 8]
    for s, b in zip(sc, bounds):
        m, e = math.frexp(s)
        assert m == 0.5, "not a power of two"
        assert b * s <= 2 ** 9 and b * s > 2 ** 8
    # a large gamma / beta in the stem's batch-norm lowers the first layer's scale accordingly
    ws[1] = np.full(128, 100.0, np.float32)
    ws[2] = np.full(128, -50.0, np.float32)
    sc2 = net.activation_scales(ws)
    assert sc2[0] * 850.0 <= 2 ** 9 < sc2[0] * 850.0 * 2
    assert sc2[1] == sc[1]                     # the second layer's input is bounded by the first block's own batch-norm
    assert sc2[2] < sc[2]                      # ... but the block output adds the shortcut's bound


def test_torch_net_oracle_agrees_with_the_numpy_restatement():
    """oracle/net_ref.py (PyTorch fp32, the checker of "Q within 1e-5") against oracle/train_ref.py::forward_eval (explicit
    float64 NumPy loops): two statements of the Keras graph that share no code agree to float32 rounding, on the gen-0 net
    and on a net with randomised batch-norm statistics, 11x11 observations from the reference's goldens"""
    import numpy as np
    from conftest import load_golden
    from oracle import net_ref, train_ref
    from snake_engine.net import glorot_uniform_weights
    X = load_golden("states_11x11x4.npz")["raw"][:24]
    rng = np.random.RandomState(3)
    ws = glorot_uniform_weights((21, 21, 3), blocks=4, seed=0)
    for randomise in (False, True):
        if randomise:
            k = 0
            while k < len(ws):
                if ws[k].ndim == 4:
                    n = ws[k].shape[3]
                    ws[k + 1] = (1.0 + 0.2 * rng.randn(n)).astype(np.float32); ws[k + 2] = (0.1 * rng.randn(n)).astype(np.float32)
                    ws[k + 3] = (0.05 * rng.randn(n)).astype(np.float32); ws[k + 4] = (0.5 + rng.rand(n)).astype(np.float32)
                    k += 5
                else:
                    k += 1
        a = net_ref.forward(ws, X, apply_mask=False)
        b = train_ref.forward_eval(ws, X)
        assert np.abs(a - b).max() <= 2e-6, np.abs(a - b).max()
        assert len(np.unique(np.round(b, 4))) > 20
