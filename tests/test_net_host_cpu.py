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
