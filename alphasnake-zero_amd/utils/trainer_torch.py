"""Training half of AlphaNNet (reference: AlphaNNet.train / copy_and_compile, alpha_nnet.py:58-59, 78-106) --
SURVEY.md section 8 row f-1.  On the GPU a step is sequenced by hand on this library's kernels
(snake_engine/train_step.py: convolutions, batch norms, head, Adam -- no autograd graph, no library convolution);
the autograd restatement `_Net` below is the float64 / CPU cross-check of the same formulas (tests/test_trainer_cpu.py)
and the `SNK_TRAIN_CONV=torch` A/B arm (every operator from PyTorch / MIOpen).

Keras 2.x / TF 2.1 semantics restated (the formulas, not the code; tests/test_trainer_cpu.py cross-checks them against
an independent float64 NumPy restatement with a hand-written backward pass):
  loss   mean squared error over batch and the 3 outputs + 1e-5 * sum(kernel^2) over every Conv2D / Dense kernel
         (kernel_regularizer=l2(c), alpha_nnet.py:15, 21...)
  Adam   m = b1 m + (1 - b1) g;  v = b2 v + (1 - b2) g^2;  lr_t = lr * sqrt(1 - b2^t) / (1 - b1^t);
         w -= lr_t * m / (sqrt(v) + 1e-7)        (epsilon is added to the UNCORRECTED sqrt(v): Keras' "epsilon hat",
         not torch.optim.Adam's placement), b1 0.9, b2 0.999
  lr     PiecewiseConstantDecay([20,40,60,80,100], [lr, lr/4, lr/16, lr/64, lr/256, 0.0]) of the optimizer's step count
         (alpha_nnet.py:79-84: the rate is 0 after step 100)
  BN     training mode: normalise with the batch mean and the BIASED batch variance, epsilon 1e-3; moving_mean and
         moving_variance move with momentum 0.99, the variance that enters the moving average is the UNBIASED one
         (TF's fused batch norm applies Bessel's correction to the variance it hands to the moving average)
  fit    shuffles every epoch.
Data parallel (torch.distributed initialised, one process per GPU, RCCL): every rank holds the same sample set (the
iteration-end all-gather, snake_engine/dist.py) and the same seed, draws the same global batches and takes every
world-th row of each; batch-norm statistics are all-reduced (forward: sum, sum of squares; backward: the two reductions
of the gradient), the gradients are summed in ONE flat bucket per step.  The result equals the one-rank full-batch
result up to float32 rounding, the ranks run the same number of steps by construction and end with identical weights
and moving statistics.
"""
import os

import numpy as np
import torch
import torch.nn.functional as F

_NATIVE = os.environ.get("SNK_TRAIN_CONV", "native") != "torch"           # `torch`: every operator through PyTorch / MIOpen (A/B runs)
# steps at learning rate 0 (alpha_nnet.py:79-84: all after the 100th) cannot move a weight; Adam's moments die with the
# optimizer (copy_and_compile re-creates it): only the forward half -- the batch-norm moving averages -- is run for them.
# `full` runs the dead backward passes anyway (the equivalence test, A/B timing)
_DEAD_STEPS_FULL = os.environ.get("SNK_TRAIN_DEAD_STEPS", "skip") == "full"

BN_EPS, BN_MOMENTUM, L2_C = 1e-3, 0.99, 1e-5
ADAM_B1, ADAM_B2, ADAM_EPS = 0.9, 0.999, 1e-7


def lr_at(step, schedule, default=1e-4):
    """tf.keras PiecewiseConstantDecay: values[0] while step <= boundaries[0], ..., values[-1] afterwards"""
    if schedule is None:
        return default
    boundaries, values = schedule
    for b, v in zip(boundaries, values):
        if step <= b:
            return v
    return values[-1]


def _dist():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        return dist
    return None


class _SyncBatchNormTrain(torch.autograd.Function):
    """y = gamma (x - mean) / sqrt(var + eps) + beta with mean / biased var over (N, H, W) of ALL ranks.
    Returns (y, mean, var, count) -- the last three for the moving averages."""

    @staticmethod
    def forward(ctx, x, gamma, beta):
        dist = _dist()
        c = x.shape[1]
        stat = torch.empty(2 * c + 1, dtype=x.dtype, device=x.device)
        stat[:c] = x.sum(dim=(0, 2, 3))
        stat[c:2 * c] = (x * x).sum(dim=(0, 2, 3))
        stat[2 * c] = float(x.numel() // c)
        if dist is not None:
            dist.all_reduce(stat)
        n = stat[2 * c]
        mean = stat[:c] / n
        var = (stat[c:2 * c] / n - mean * mean).clamp_min(0.0)
        inv = torch.rsqrt(var + BN_EPS)
        xhat = (x - mean[None, :, None, None]) * inv[None, :, None, None]
        ctx.save_for_backward(xhat, gamma, inv, n)
        ctx.mark_non_differentiable(mean, var, n)
        return xhat * gamma[None, :, None, None] + beta[None, :, None, None], mean, var, n

    @staticmethod
    def backward(ctx, dy, _dm, _dv, _dn):
        xhat, gamma, inv, n = ctx.saved_tensors
        dist = _dist()
        c = dy.shape[1]
        red = torch.empty(2 * c, dtype=dy.dtype, device=dy.device)
        red[:c] = dy.sum(dim=(0, 2, 3))                       # local d beta
        red[c:] = (dy * xhat).sum(dim=(0, 2, 3))              # local d gamma
        dbeta, dgamma = red[:c].clone(), red[c:].clone()
        if dist is not None:
            dist.all_reduce(red)                               # the input gradient needs the global reductions
        g = (gamma * inv)[None, :, None, None]
        dx = g * (dy - red[:c][None, :, None, None] / n - xhat * red[c:][None, :, None, None] / n)
        return dx, dgamma, dbeta


class _Net:
    """functional restatement of the graph on torch tensors kept in the Keras layout"""

    def __init__(self, weights, device, dtype=torch.float32):
        self.device = device
        self.blocks = (len(weights) - 14) // 10
        self.t = [torch.tensor(np.asarray(w), dtype=dtype, device=device) for w in weights]
        self.kernel_idx, self.bn_idx, self.param_idx = [], [], []
        i = 0
        for _ in range(2 + 2 * self.blocks):
            self.kernel_idx.append(i); self.bn_idx.append(i + 1); self.param_idx += [i, i + 1, i + 2]
            i += 5
        self.kernel_idx += [i, i + 2]
        self.param_idx += [i, i + 1, i + 2, i + 3]
        for j in self.param_idx:
            self.t[j].requires_grad_(True)

    def params(self):
        return [self.t[j] for j in self.param_idx]

    def _conv_bn(self, x, i, train, residual=None, relu=True):
        """act(batch_norm(conv(x)) (+ residual)), act = ReLU (every layer of the graph has one after its batch norm)"""
        k = self.t[i]
        y = F.conv2d(x, k.permute(3, 2, 0, 1), padding=k.shape[0] // 2)
        g, b, mean, var = self.t[i + 1:i + 5]
        if train:
            out, m, v, n = _SyncBatchNormTrain.apply(y, g, b)
            with torch.no_grad():
                unbiased = v * (n / (n - 1.0).clamp_min(1.0))
                mean.mul_(BN_MOMENTUM).add_(m * (1 - BN_MOMENTUM))
                var.mul_(BN_MOMENTUM).add_(unbiased * (1 - BN_MOMENTUM))
        else:
            out = (y - mean[None, :, None, None]) * (g / torch.sqrt(var + BN_EPS))[None, :, None, None] + b[None, :, None, None]
        if residual is not None:
            out = out + residual
        return F.relu(out) if relu else out

    def forward(self, x_nhwc, train):
        x = x_nhwc.permute(0, 3, 1, 2)
        h = self._conv_bn(x, 0, train)
        i = 5
        for _ in range(self.blocks):
            sc = h
            h = self._conv_bn(h, i, train)
            h = self._conv_bn(h, i + 5, train, residual=sc)
            i += 10
        h = self._conv_bn(h, i, train)
        h = h.permute(0, 2, 3, 1).reshape(h.shape[0], h.shape[1] * h.shape[2] * h.shape[3])      # (explicit: a rank's slice may be empty)
        h = F.relu(h @ self.t[i + 5] + self.t[i + 6])
        return torch.tanh(h @ self.t[i + 7] + self.t[i + 8])

    def l2(self):
        return L2_C * sum((self.t[j] ** 2).sum() for j in self.kernel_idx)

    def weights(self):
        return [w.detach().cpu().numpy().copy() for w in self.t]


class KerasAdam:
    """Adam exactly as tf.keras.optimizers.Adam applies it (see the module docstring), on one flat state"""

    def __init__(self, params):
        self.params = params
        n = sum(p.numel() for p in params)
        dev = params[0].device
        self.m = torch.zeros(n, dtype=params[0].dtype, device=dev)
        self.v = torch.zeros(n, dtype=params[0].dtype, device=dev)
        self.t = 0

    def step(self, flat_grad, lr):
        self.t += 1
        self.m.mul_(ADAM_B1).add_(flat_grad, alpha=1 - ADAM_B1)
        self.v.mul_(ADAM_B2).addcmul_(flat_grad, flat_grad, value=1 - ADAM_B2)
        lr_t = lr * np.sqrt(1.0 - ADAM_B2 ** self.t) / (1.0 - ADAM_B1 ** self.t)
        upd = self.m / (self.v.sqrt() + ADAM_EPS)
        o = 0
        with torch.no_grad():
            for p in self.params:
                k = p.numel()
                p.sub_(upd[o:o + k].view_as(p), alpha=float(lr_t))
                o += k


def fit(weights, input_shape, X, Y, epochs=32, batch_size=2048, lr_schedule=None, device=None, seed=None, verbose=True,
        shuffle=True, dtype=torch.float32):
    """Returns the trained weights (Keras order).  X: (N, h, w, 3) float32, Y: (N, 3) float32 -- on every rank the SAME
    arrays (and the same seed) when torch.distributed is initialised; rank r then works on rows r::world of every batch."""
    dist = _dist()
    world = dist.get_world_size() if dist is not None else 1
    rank = dist.get_rank() if dist is not None else 0
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
    device = torch.device(device)
    Xd = torch.as_tensor(np.ascontiguousarray(X), dtype=dtype, device=device)
    Yd = torch.as_tensor(np.ascontiguousarray(Y), dtype=dtype, device=device)
    n = Xd.shape[0]
    if seed is None:
        s = torch.tensor([int(np.random.randint(1 << 31))], dtype=torch.int64, device=device if dist is not None and dist.get_backend() == "nccl" else "cpu")
        if dist is not None:
            dist.broadcast(s, 0)                 # one shuffle order for all ranks
        seed = int(s.item())
    gen = torch.Generator(device="cpu")
    gen.manual_seed(seed)
    native = None
    if _NATIVE and device.type == "cuda" and dtype == torch.float32:
        from snake_engine import train_step
        if train_step.supported(input_shape, (len(weights) - 14) // 10):
            native = train_step.TrainStep(weights, input_shape, -(-min(batch_size, n) // world), device, dist)
        else:
            import warnings
            warnings.warn(f"fit: no weight-gradient kernel for {tuple(input_shape)} observations (square, width 3 .. 96, at least one "
                          "residual block): this fit runs on PyTorch's operators")
    if native is None:
        net = _Net(weights, device, dtype)        # float64 only for the cross-checks in tests/test_trainer_cpu.py
        params = net.params()
        opt = KerasAdam(params)
    step = 0
    history = []
    fit.last_mode = "kernels" if native is not None else "autograd"
    for ep in range(epochs):
        perm = (torch.randperm(n, generator=gen) if shuffle else torch.arange(n)).to(device)
        tot = torch.zeros((), dtype=torch.float64, device=device)      # the epoch's loss stays on the device: one read per epoch
        cnt = 0
        for s0 in range(0, n, batch_size):
            idx_all = perm[s0:s0 + batch_size]
            if len(idx_all) < world and native is not None:
                # the kernels of the native step take at least one row.  Every rank sees the same permutation, so every rank takes
                # this branch: nobody is left inside the batch-norm or gradient all-reduces of a batch that another rank refused.
                # (The trainer never gets here: snake_engine.dist.sample_plan yields whole batches.)
                raise RuntimeError(f"fit: a trailing batch of {len(idx_all)} rows cannot be split over {world} ranks")
            # Keras' fit trains on a short trailing batch like on any other; a rank whose slice of it is empty still joins every
            # all-reduce of the step with sums over no rows (zeros) and its share of the l2 term
            idx = idx_all[rank::world]
            lr = lr_at(step, lr_schedule)
            if native is not None:
                xb, yb = Xd[idx].contiguous(), Yd[idx].contiguous()
                if lr == 0.0 and not _DEAD_STEPS_FULL:
                    loss2 = native.forward_only(xb, yb, len(idx_all))
                else:
                    loss2 = native.step(xb, yb, len(idx_all), lr)
                tot += loss2.sum().double() * len(idx_all)
            else:
                pred = net.forward(Xd[idx], True)
                # this rank's share of the global batch loss; summed over ranks it is mse + l2 of the whole batch
                loss = ((pred - Yd[idx]) ** 2).sum() / (3.0 * len(idx_all)) + net.l2() / world
                grads = torch.autograd.grad(loss, params)
                flat = torch.cat([g.reshape(-1) for g in grads] + [loss.detach().reshape(1)])
                if dist is not None:
                    dist.all_reduce(flat)            # one bucket: every gradient + the loss value
                opt.step(flat[:-1], lr)
                tot += flat[-1].double() * len(idx_all)
            step += 1
            cnt += len(idx_all)
        history.append(float(tot.item()) / max(1, cnt))
        if verbose and rank == 0:
            print(f"Epoch {ep + 1}/{epochs} - loss: {history[-1]:.6f}")
    fit.last_history = history
    return native.weights() if native is not None else net.weights()
