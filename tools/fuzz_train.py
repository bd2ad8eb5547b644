"""differential fuzz of the training step's round-5 forms (snake_engine/train_step.py): the same step on random shapes with every
switch on (deferred batch norms incl. the stem's, the shortcut's gradient through mask bytes, batched weight images, the head's 1x1
stage in the last batch-norm kernel) and with every switch off -- Q, the loss, every parameter gradient and the batch-norm moving
statistics must agree to float32 rounding (the two differ in summation order and in the power of two of some input ranges only).
Development aid: fuzz_train.py [seed 0] [trials 24]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "alphasnake-zero_amd")]
import numpy as np, torch
from snake_engine import net, train_step
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
trials = int(sys.argv[2]) if len(sys.argv) > 2 else 24
rs = np.random.RandomState(seed)
SW = ("_DEFER_BN", "_DEFER_STEM", "_RES_MASK", "_BATCH_PREP", "_HEAD_FUSED")
worst_g, worst_q, n_def, n_stem = 0.0, 0.0, 0, 0
for t in range(trials):
    hw = int(rs.choice([5, 7, 9, 11, 13, 15, 17, 21, 21, 25, 29, 33, 37, 37, 41]))
    if not train_step.supported((hw, hw, 3)):
        continue
    blocks = int(rs.choice([1, 1, 2, 3, 4]))
    n = int(rs.choice([1, 2, 3, 5, 8, 13, 24, 40]))
    n = max(1, min(n, 30000 // (hw * hw)))
    X = torch.as_tensor(rs.rand(n, hw, hw, 3).astype(np.float32), device="cuda")
    Y = torch.as_tensor(np.tanh(rs.randn(n, 3)).astype(np.float32), device="cuda")
    ws = net.glorot_uniform_weights((hw, hw, 3), blocks=blocks, seed=seed * 100 + t)
    for l in range(2 + 2 * blocks):                         # batch-norm parameters away from 1 / 0, some scales negative
        g = ws[5 * l + 1] * (0.5 + rs.rand(*ws[5 * l + 1].shape))
        g[rs.rand(*g.shape) < 0.15] *= -1.0
        ws[5 * l + 1] = g.astype(np.float32)
        ws[5 * l + 2] = (0.3 * rs.randn(*ws[5 * l + 2].shape)).astype(np.float32)
    out = {}
    for on in (True, False):
        for k in SW:
            setattr(train_step, k, on)
        ts = train_step.TrainStep(ws, (hw, hw, 3), n, "cuda")
        q = ts.forward(X, Y, n, want_bwd=True).clone()
        ts.backward(Y, n)
        out[on] = (q, float(ts.G[ts.n_params]), ts.gradients(), ts.weights(), ts.defer, ts.defer_stem)
    (q1, l1, g1, w1, d1, s1), (q0, l0, g0, w0, _, _) = out[True], out[False]
    n_def += d1; n_stem += s1
    dq = float((q1 - q0).abs().max())
    assert dq <= 5e-6 and abs(l1 - l0) <= 1e-5 * abs(l0), (t, hw, n, blocks, dq, l1, l0)
    for j in g1:
        e = float(np.abs(g1[j] - g0[j]).max() / max(np.abs(g0[j]).max(), 1e-12))
        worst_g = max(worst_g, e)
        assert e <= 1e-4, (t, hw, n, blocks, j, e)
    for j, (u, v) in enumerate(zip(w1, w0)):
        assert np.abs(u - v).max() <= 2e-6 * max(np.abs(v).max(), 1e-12), (t, hw, n, blocks, j)
    worst_q = max(worst_q, dq)
print(f"fuzz_train seed {seed}: {trials} random steps (widths 5 .. 41, 1 .. 40 images, 1 .. 4 blocks; {n_def} with deferred batch norms, {n_stem} with the "
      f"stem's too): round-5 forms against the written-activation forms -- worst |dQ| {worst_q:.1e}, worst gradient difference {worst_g:.1e} of the tensor's largest entry")
