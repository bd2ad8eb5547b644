"""Independent float64 NumPy restatement of one `AlphaNNet.train` optimizer step (TEST INFRASTRUCTURE, see
oracle/__init__.py): forward, hand-written backward, Keras Adam, batch-norm moving averages.

PARITY UNPINNED: the reference's training arithmetic lives in TensorFlow/Keras 2.1 (alpha_nnet.py:58-59, 78-106), which
is absent here and commits no training fixture; this file restates the published Keras / TF formulas
  Conv2D(padding="same", use_bias=False, kernel_regularizer=l2(1e-5))  alpha_nnet.py:21, 27, 36, 49
  BatchNormalization(axis=3): training mode, momentum 0.99, epsilon 1e-3; the moving variance takes the unbiased batch
                              variance (TF fused batch norm)                     alpha_nnet.py:22, 28, 37, 50
  Dense(128, relu), Dense(3, tanh), both with l2(1e-5) on the kernel            alpha_nnet.py:52-54
  loss = mean_squared_error + regularisation losses                             alpha_nnet.py:95, 105
  Adam(learning_rate = PiecewiseConstantDecay(...), epsilon = 1e-7)             alpha_nnet.py:79-84, 94, 104
with explicit loops and no autograd, so that utils/trainer_torch.py (autograd + its own Adam + synchronised batch norm)
can be cross-checked against something it shares no code with.  Small shapes only (it is O(slow)).
"""
import numpy as np

BN_EPS, BN_MOMENTUM, L2_C = 1e-3, 0.99, 1e-5
B1, B2, EPS = 0.9, 0.999, 1e-7


def _conv_fwd(x, w):
    kh = w.shape[0]
    p = kh // 2
    n, H, W, _ = x.shape
    xp = np.pad(x, ((0, 0), (p, p), (p, p), (0, 0)))
    y = np.zeros((n, H, W, w.shape[3]))
    for a in range(kh):
        for b in range(kh):
            y += xp[:, a:a + H, b:b + W, :] @ w[a, b]
    return y


def _conv_bwd(x, w, dy):
    kh = w.shape[0]
    p = kh // 2
    n, H, W, _ = x.shape
    xp = np.pad(x, ((0, 0), (p, p), (p, p), (0, 0)))
    dxp = np.zeros_like(xp)
    dw = np.zeros_like(w)
    for a in range(kh):
        for b in range(kh):
            patch = xp[:, a:a + H, b:b + W, :]
            dw[a, b] = np.einsum("nhwc,nhwd->cd", patch, dy)
            dxp[:, a:a + H, b:b + W, :] += dy @ w[a, b].T
    return dxp[:, p:p + H, p:p + W, :], dw


def _bn_fwd(y, g, b):
    m = y.mean(axis=(0, 1, 2))
    v = y.var(axis=(0, 1, 2))                  # biased
    inv = 1.0 / np.sqrt(v + BN_EPS)
    xhat = (y - m) * inv
    return xhat * g + b, (xhat, inv, m, v, y.shape[0] * y.shape[1] * y.shape[2])


def _bn_bwd(dz, g, cache):
    xhat, inv, _, _, _ = cache
    dg = (dz * xhat).sum(axis=(0, 1, 2))
    db = dz.sum(axis=(0, 1, 2))
    dy = g * inv * (dz - dz.mean(axis=(0, 1, 2)) - xhat * (dz * xhat).mean(axis=(0, 1, 2)))
    return dy, dg, db


def loss_and_grads(ws, X, Y):
    """ws: Keras-order weight list (float64 arrays).  Returns (loss, grads dict index -> array, bn caches index -> cache)."""
    blocks = (len(ws) - 14) // 10
    grads, caches = {}, {}
    tape = []

    def cbr(x, i, add=None):
        y = _conv_fwd(x, ws[i])
        z, c = _bn_fwd(y, ws[i + 1], ws[i + 2])
        caches[i] = c
        if add is not None:
            z = z + add
        out = np.maximum(z, 0.0)
        tape.append((i, x, z))
        return out
    h = cbr(X, 0)
    i = 5
    skips = []
    for _ in range(blocks):
        sc = h
        h1 = cbr(h, i)
        h = cbr(h1, i + 5, add=sc)
        skips.append(i)
        i += 10
    hh = cbr(h, i)
    flat = hh.reshape(hh.shape[0], -1)
    a1 = flat @ ws[i + 5] + ws[i + 6]
    r1 = np.maximum(a1, 0.0)
    a2 = r1 @ ws[i + 7] + ws[i + 8]
    pred = np.tanh(a2)
    kernel_idx = [5 * k for k in range(2 + 2 * blocks)] + [i + 5, i + 7]
    loss = ((pred - Y) ** 2).mean() + L2_C * sum((ws[k] ** 2).sum() for k in kernel_idx)
    # backward
    da2 = (2.0 * (pred - Y) / pred.size) * (1.0 - pred ** 2)
    grads[i + 7] = r1.T @ da2
    grads[i + 8] = da2.sum(axis=0)
    da1 = (da2 @ ws[i + 7].T) * (a1 > 0)
    grads[i + 5] = flat.T @ da1
    grads[i + 6] = da1.sum(axis=0)
    dh = (da1 @ ws[i + 5].T).reshape(hh.shape)
    pending_skip = {}
    for (k, x_in, z) in reversed(tape):
        dz = dh * (z > 0)
        dy, grads[k + 1], grads[k + 2] = _bn_bwd(dz, ws[k + 1], caches[k])
        dx, grads[k] = _conv_bwd(x_in, ws[k], dy)
        is_second_of_block = k >= 5 and k < 5 + 10 * blocks and (k - 5) % 10 == 5
        is_first_of_block = k >= 5 and k < 5 + 10 * blocks and (k - 5) % 10 == 0
        if is_second_of_block:
            pending_skip[k - 5] = dz            # the shortcut carries dz straight to the block's input
            dh = dx
        elif is_first_of_block:
            dh = dx + pending_skip.pop(k)
        else:
            dh = dx
    for k in kernel_idx:
        grads[k] = grads[k] + 2.0 * L2_C * ws[k]
    return loss, grads, caches


class KerasAdamRef:
    def __init__(self):
        self.m, self.v, self.t = {}, {}, 0

    def step(self, ws, grads, lr):
        self.t += 1
        lr_t = lr * np.sqrt(1.0 - B2 ** self.t) / (1.0 - B1 ** self.t)
        for k, g in grads.items():
            self.m[k] = B1 * self.m.get(k, 0.0) + (1 - B1) * g
            self.v[k] = B2 * self.v.get(k, 0.0) + (1 - B2) * g * g
            ws[k] = ws[k] - lr_t * self.m[k] / (np.sqrt(self.v[k]) + EPS)


def train_steps(weights, X, Y, lrs):
    """full-batch steps with the given learning rates; returns (weights after, losses)"""
    ws = [np.asarray(w, np.float64).copy() for w in weights]
    X, Y = np.asarray(X, np.float64), np.asarray(Y, np.float64)
    opt = KerasAdamRef()
    losses = []
    for lr in lrs:
        loss, grads, caches = loss_and_grads(ws, X, Y)
        losses.append(loss)
        for k, (_, _, m, v, n) in caches.items():     # moving averages (unbiased variance into the average)
            ws[k + 3] = ws[k + 3] * BN_MOMENTUM + m * (1 - BN_MOMENTUM)
            ws[k + 4] = ws[k + 4] * BN_MOMENTUM + v * (n / max(n - 1.0, 1.0)) * (1 - BN_MOMENTUM)
        opt.step(ws, grads, lr)
    return ws, losses


def forward_eval(weights, X):
    """AlphaNNet.v_net.predict in float64 NumPy (inference-mode batch norm, alpha_nnet.py:19-56): a second,
    code-independent statement of what oracle/net_ref.py (PyTorch) computes; tests/test_net_host_cpu.py compares the two.
    Returns the unmasked (N, 3) outputs."""
    ws = [np.asarray(w, np.float64) for w in weights]
    blocks = (len(ws) - 14) // 10

    def cbn(x, i):
        y = _conv_fwd(x, ws[i])
        return ws[i + 1] * (y - ws[i + 3]) / np.sqrt(ws[i + 4] + BN_EPS) + ws[i + 2]
    h = np.maximum(cbn(np.asarray(X, np.float64), 0), 0.0)
    i = 5
    for _ in range(blocks):
        sc = h
        h = np.maximum(cbn(h, i), 0.0)
        h = np.maximum(cbn(h, i + 5) + sc, 0.0)
        i += 10
    h = np.maximum(cbn(h, i), 0.0)
    flat = h.reshape(h.shape[0], -1)
    a1 = np.maximum(flat @ ws[i + 5] + ws[i + 6], 0.0)
    return np.tanh(a1 @ ws[i + 7] + ws[i + 8])
