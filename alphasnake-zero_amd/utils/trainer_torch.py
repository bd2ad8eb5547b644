"""Training half of AlphaNNet (reference: AlphaNNet.train / copy_and_compile, alpha_nnet.py:58-59, 78-106) --
SURVEY.md section 8 row f-1 ("next").  Plain PyTorch (autograd, MIOpen convolutions on the GPU): this is the
caller-side fit step that consumes the self-play samples, not part of the HIP hot path.

Keras semantics restated: loss = mean squared error + 1e-5 * sum(kernel^2) over every Conv2D/Dense kernel
(kernel_regularizer=l2(c), alpha_nnet.py:15,21...); Adam(beta 0.9/0.999, epsilon 1e-7) with
PiecewiseConstantDecay([20,40,60,80,100], [lr, lr/4, lr/16, lr/64, lr/256, 0.0]) on the optimizer step
(alpha_nnet.py:79-84: the rate is 0 after step 100); BatchNormalization in training mode (batch statistics,
moving averages with momentum 0.99 of the biased batch variance, epsilon 1e-3); `fit` shuffles every epoch.
With torch.distributed initialised the gradients are averaged across ranks (RCCL all-reduce) each step.
"""
import numpy as np
import torch
import torch.nn.functional as F

BN_EPS, BN_MOMENTUM, L2_C = 1e-3, 0.99, 1e-5


def lr_at(step, schedule, default=1e-4):
    """tf.keras PiecewiseConstantDecay: values[0] while step <= boundaries[0], ..., values[-1] afterwards"""
    if schedule is None:
        return default
    boundaries, values = schedule
    for b, v in zip(boundaries, values):
        if step <= b:
            return v
    return values[-1]


class _Net:
    """functional restatement of the graph on torch tensors kept in the Keras layout"""

    def __init__(self, weights, device):
        self.device = device
        self.blocks = (len(weights) - 14) // 10
        self.t = [torch.tensor(np.asarray(w, np.float32), device=device) for w in weights]
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

    def _conv_bn(self, x, i, train):
        k = self.t[i]
        y = F.conv2d(x, k.permute(3, 2, 0, 1), padding=k.shape[0] // 2)
        g, b, mean, var = self.t[i + 1:i + 5]
        if train:
            m = y.mean(dim=(0, 2, 3))
            v = y.var(dim=(0, 2, 3), unbiased=False)
            with torch.no_grad():
                mean.mul_(BN_MOMENTUM).add_(m.detach() * (1 - BN_MOMENTUM))
                var.mul_(BN_MOMENTUM).add_(v.detach() * (1 - BN_MOMENTUM))
        else:
            m, v = mean, var
        return (y - m[None, :, None, None]) * (g / torch.sqrt(v + BN_EPS))[None, :, None, None] + b[None, :, None, None]

    def forward(self, x_nhwc, train):
        x = x_nhwc.permute(0, 3, 1, 2)
        h = F.relu(self._conv_bn(x, 0, train))
        i = 5
        for _ in range(self.blocks):
            sc = h
            h = F.relu(self._conv_bn(h, i, train))
            h = F.relu(self._conv_bn(h, i + 5, train) + sc)
            i += 10
        h = F.relu(self._conv_bn(h, i, train))
        h = h.permute(0, 2, 3, 1).reshape(h.shape[0], -1)
        h = F.relu(h @ self.t[i + 5] + self.t[i + 6])
        return torch.tanh(h @ self.t[i + 7] + self.t[i + 8])

    def l2(self):
        return L2_C * sum((self.t[j] ** 2).sum() for j in self.kernel_idx)

    def weights(self):
        return [w.detach().cpu().numpy().copy() for w in self.t]


def fit(weights, input_shape, X, Y, epochs=32, batch_size=2048, lr_schedule=None, device=None, seed=None, verbose=True):
    """Returns the trained weights (Keras order).  X: (N, h, w, 3) float32, Y: (N, 3) float32."""
    import torch.distributed as dist
    if device is None:
        device = torch.device("cuda") if torch.cuda.is_available() else torch.device("cpu")
    net = _Net(weights, device)
    Xd = torch.as_tensor(np.ascontiguousarray(X, np.float32), device=device)
    Yd = torch.as_tensor(np.ascontiguousarray(Y, np.float32), device=device)
    n = Xd.shape[0]
    opt = torch.optim.Adam(net.params(), lr=1.0, betas=(0.9, 0.999), eps=1e-7)
    gen = torch.Generator(device="cpu")
    gen.manual_seed(int(np.random.randint(1 << 31)) if seed is None else seed)
    world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    step = 0
    history = []
    for ep in range(epochs):
        perm = torch.randperm(n, generator=gen).to(device)
        tot, cnt = 0.0, 0
        for s0 in range(0, n, batch_size):
            idx = perm[s0:s0 + batch_size]
            pred = net.forward(Xd[idx], True)
            mse = ((pred - Yd[idx]) ** 2).mean()
            loss = mse + net.l2()
            opt.zero_grad(set_to_none=True)
            loss.backward()
            if world > 1:
                for p in net.params():
                    dist.all_reduce(p.grad)
                    p.grad.div_(world)
            for gparam in opt.param_groups:
                gparam["lr"] = lr_at(step, lr_schedule)
            opt.step()
            step += 1
            tot += float(loss.item()) * len(idx); cnt += len(idx)
        history.append(tot / max(1, cnt))
        if verbose:
            print(f"Epoch {ep + 1}/{epochs} - loss: {history[-1]:.6f}")
    fit.last_history = history
    return net.weights()
