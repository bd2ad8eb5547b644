"""CPU tests of the training half (utils/trainer_torch.py): Keras semantics of the
loss, the piecewise learning-rate schedule, BN moving statistics, and that fit() learns."""
import numpy as np
import torch


def test_lr_schedule_matches_keras_piecewise_constant_decay():
    from utils.trainer_torch import lr_at
    sched = ([20, 40, 60, 80, 100], [1e-4, 2.5e-5, 6.25e-6, 1.5625e-6, 3.90625e-7, 0.0])
    assert lr_at(0, sched) == 1e-4 and lr_at(20, sched) == 1e-4 and lr_at(21, sched) == 2.5e-5
    assert lr_at(100, sched) == 3.90625e-7 and lr_at(101, sched) == 0.0 and lr_at(10 ** 6, sched) == 0.0


def test_fit_reduces_loss_and_updates_bn_stats():
    from snake_engine.net import glorot_uniform_weights
    from utils import trainer_torch
    from oracle import net_ref
    torch.manual_seed(0)
    rng = np.random.RandomState(0)
    ws = glorot_uniform_weights((7, 7, 3), blocks=1, seed=0)           # small board, one block: fast on CPU
    X = rng.rand(96, 7, 7, 3).astype(np.float32)
    Y = np.tanh(rng.randn(96, 3)).astype(np.float32) * 0.5
    before = float(((net_ref.forward(ws, X, apply_mask=False) - Y) ** 2).mean())
    out = trainer_torch.fit(ws, (7, 7, 3), X, Y, epochs=6, batch_size=32, lr_schedule=([1000], [3e-3, 0.0]),
                            device=torch.device("cpu"), seed=1, verbose=False)
    assert len(out) == len(ws) and all(a.shape == b.shape for a, b in zip(ws, out))
    hist = trainer_torch.fit.last_history
    assert hist[-1] < hist[0]
    assert not np.array_equal(out[3], ws[3]) and not np.array_equal(out[4], ws[4])      # BN moving mean / variance moved
    after = float(((net_ref.forward(out, X, apply_mask=False) - Y) ** 2).mean())
    assert after < before
    # learning rate 0 after step 100 (alpha_nnet.py:79-84): further training leaves the kernels untouched
    frozen = trainer_torch.fit(out, (7, 7, 3), X, Y, epochs=1, batch_size=32, lr_schedule=([0], [0.0, 0.0]),
                               device=torch.device("cpu"), seed=2, verbose=False)
    assert np.array_equal(frozen[0], out[0]) and np.array_equal(frozen[-2], out[-2])


def test_tensorflow_shim_sends_train_py_down_the_cpu_branch():
    import importlib
    tf = importlib.import_module("tensorflow")
    TPU = "unset"
    try:                                             # train.py:15-27, verbatim control flow
        Resolver = tf.distribute.cluster_resolver.TPUClusterResolver(tpu="")
        tf.config.experimental_connect_to_cluster(Resolver)
        TPU = tf.distribute.experimental.TPUStrategy(Resolver)
    except:                                          # noqa: E722
        TPU = None
    assert TPU is None
