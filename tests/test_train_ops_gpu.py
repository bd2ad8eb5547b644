"""The training step on this library's kernels (snake_engine/train_step.py) against float64
PyTorch expressions of the same formulas (test infrastructure: autograd in float64 is the checker, never the product):
the three passes of the tower convolution, the stem convolution and its weight gradient, the batch-norm kernels (also on
channels whose mean is a hundred standard deviations from zero), the head, Adam, and one whole step -- every parameter
gradient to 1e-4 of its largest entry once the ReLU masks of the float64 run are imposed on the float32 run (a mask that
flips on round-off changes a gradient by far more than round-off; with equal masks the step is a smooth function)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_gpu():
    import torch
    assert torch.cuda.is_available()
    return torch


def rel(a, b):
    return float((a.detach().double() - b.detach().double()).abs().max() / (b.detach().double().abs().max() + 1e-300))


def _tail_of(torch, L, x):
    """{., ., scale, 1 / scale} of a tensor, as the element-wise kernels leave it for the convolution that reads it"""
    from snake_engine.net import F16S_TAIL_OFFSET, F16S_WEIGHT_BYTES
    from snake_engine._lib import check
    image = torch.empty(F16S_WEIGHT_BYTES, dtype=torch.uint8, device="cuda")
    part = torch.empty(L.snk_bn_train_partials(), device="cuda")
    check(L.snk_conv3x3_f16s_input_scale(x.data_ptr(), x.numel(), image.data_ptr(), part.data_ptr(), torch.cuda.current_stream().cuda_stream))
    return image[F16S_TAIL_OFFSET:F16S_TAIL_OFFSET + 16].view(torch.float32).clone()


@pytest.mark.parametrize("n,hw,xmag,gmag", [(64, 21, 1.0, 1.0), (48, 21, 30.0, 1e-6), (5, 13, 1e-3, 50.0), (3, 37, 1.0, 1e-4)])
def test_tower_convolution_three_passes_match_float64(torch_gpu, n, hw, xmag, gmag):
    """forward and input gradient on k_conv3x3_f16s (snk_conv3x3_prepare_weights_f16s_train: the mirrored kernel is laid out by
    the library), weight gradient on k_wgrad_f16s, with the power-of-two scales taken from the data on the device"""
    torch = torch_gpu
    import torch.nn.functional as F
    from snake_engine._lib import lib, check
    from snake_engine.net import F16S_WEIGHT_BYTES
    L, st = lib(), torch.cuda.current_stream().cuda_stream
    g = torch.Generator(device="cuda").manual_seed(n)
    x = torch.randn(n, hw, hw, 128, device="cuda", generator=g) * xmag
    k = torch.randn(3, 3, 128, 128, device="cuda", generator=g) * 0.05
    dy = torch.randn(n, hw, hw, 128, device="cuda", generator=g) * gmag
    dy = (dy * (torch.rand(n, hw, hw, 1, device="cuda", generator=g) ** 6)).contiguous()    # a wide spread of magnitudes, as real gradients have
    ones, zeros = torch.ones(128, device="cuda"), torch.zeros(128, device="cuda")
    image = torch.empty(F16S_WEIGHT_BYTES, dtype=torch.uint8, device="cuda")
    image_b = torch.empty_like(image)                       # the mirrored kernel of the input gradient reuses the first image's weight scale
    tx, tdy = _tail_of(torch, L, x), _tail_of(torch, L, dy)
    y, dx = torch.empty_like(x), torch.empty_like(x)
    check(L.snk_conv3x3_prepare_weights_f16s_train(k.data_ptr(), image.data_ptr(), tx.data_ptr(), 0, None, st))
    check(L.snk_conv3x3_bn_f16s(x.data_ptr(), image.data_ptr(), ones.data_ptr(), zeros.data_ptr(), None, y.data_ptr(), n, hw, hw, 0, st))
    check(L.snk_conv3x3_prepare_weights_f16s_train(k.data_ptr(), image_b.data_ptr(), tdy.data_ptr(), 1, image.data_ptr(), st))
    check(L.snk_conv3x3_bn_f16s(dy.data_ptr(), image_b.data_ptr(), ones.data_ptr(), zeros.data_ptr(), None, dx.data_ptr(), n, hw, hw, 0, st))
    need = L.snk_conv3x3_wgrad_partials(hw, hw)
    assert need > 0, "every square observation the engine supports has a weight-gradient kernel"
    part = torch.empty(need, device="cuda")
    dk = torch.empty(3, 3, 128, 128, device="cuda")
    check(L.snk_conv3x3_wgrad_f16s(x.data_ptr(), dy.data_ptr(), tx.data_ptr(), tdy.data_ptr(), part.data_ptr(), dk.data_ptr(), n, hw, hw, st))
    torch.cuda.synchronize()

    # the forward form the step uses: the same convolution with the batch-norm sums taken in its epilogue
    center = torch.randn(128, device="cuda", generator=g) * xmag
    cpart = torch.empty(L.snk_conv3x3_stats_partials(n, hw, hw), device="cuda")
    sums = torch.empty(256, dtype=torch.float64, device="cuda")
    y2 = torch.empty_like(x)
    check(L.snk_conv3x3_prepare_weights_f16s_train(k.data_ptr(), image.data_ptr(), tx.data_ptr(), 0, None, st))
    check(L.snk_conv3x3_f16s_stats(x.data_ptr(), image.data_ptr(), y2.data_ptr(), center.data_ptr(), cpart.data_ptr(), sums.data_ptr(), n, hw, hw, st))
    assert torch.equal(y2, y)
    e = y.double().reshape(-1, 128) - center.double()
    assert rel(sums[:128], e.sum(dim=0)) < 1e-5 and rel(sums[128:], (e * e).sum(dim=0)) < 1e-6

    x64 = x.double().permute(0, 3, 1, 2).requires_grad_(True)
    k64 = k.double().requires_grad_(True)
    y64 = F.conv2d(x64, k64.permute(3, 2, 0, 1), padding=1)
    dx64, dk64 = torch.autograd.grad(y64, (x64, k64), dy.double().permute(0, 3, 1, 2))
    assert rel(y, y64.permute(0, 2, 3, 1)) < 2e-6
    assert rel(dx, dx64.permute(0, 2, 3, 1)) < 2e-6
    assert rel(dk, dk64) < 3e-6, rel(dk, dk64)
    y32 = F.conv2d(x.permute(0, 3, 1, 2), k.permute(3, 2, 0, 1), padding=1).permute(0, 2, 3, 1)
    assert rel(y, y64.permute(0, 2, 3, 1)) <= 4 * rel(y32, y64.permute(0, 2, 3, 1)) + 1e-7      # no further from float64 than the library's float32


@pytest.mark.parametrize("n,hw", [(70, 21), (9, 13), (5, 37)])
def test_stem_convolution_and_its_weight_gradient_match_float64(torch_gpu, n, hw):
    torch = torch_gpu
    import torch.nn.functional as F
    from snake_engine._lib import lib, check
    L, st = lib(), torch.cuda.current_stream().cuda_stream
    g = torch.Generator(device="cuda").manual_seed(n)
    x = torch.rand(n, hw, hw, 3, device="cuda", generator=g) * 2 - 0.5
    k = torch.randn(3, 3, 3, 128, device="cuda", generator=g) * 0.2
    dy = (torch.randn(n, hw, hw, 128, device="cuda", generator=g) * (torch.rand(n, hw, hw, 1, device="cuda", generator=g) ** 4)).contiguous()
    y = torch.empty(n, hw, hw, 128, device="cuda")
    check(L.snk_stem_conv_f32(x.data_ptr(), k.data_ptr(), y.data_ptr(), n, hw, hw, st))
    part = torch.empty(L.snk_stem_wgrad_partials(n, hw, hw), device="cuda")
    dk = torch.full((3, 3, 3, 128), 7.0, device="cuda")
    check(L.snk_stem_wgrad_f32(x.data_ptr(), dy.data_ptr(), part.data_ptr(), dk.data_ptr(), n, hw, hw, st))
    x64, k64 = x.double().permute(0, 3, 1, 2), k.double().requires_grad_(True)
    y64 = F.conv2d(x64, k64.permute(3, 2, 0, 1), padding=1)
    dk64, = torch.autograd.grad(y64, (k64,), dy.double().permute(0, 3, 1, 2))
    assert rel(y, y64.permute(0, 2, 3, 1)) < 2e-6
    assert rel(dk, dk64) < 3e-6, rel(dk, dk64)


@pytest.mark.parametrize("with_res,n,mean,std", [(False, 33, 0.7, 3.0), (True, 64, 0.7, 3.0), (False, 40, 50.0, 0.5), (True, 17, -300.0, 2.0)])
def test_batch_norm_kernels_match_float64(torch_gpu, with_res, n, mean, std):
    """sums (centred on the moving mean, float64) -> finalize -> apply, and the way back, against float64 autograd; the
    channels of the third and fourth case sit 100 / 150 standard deviations from zero, where E[x^2] - mean^2 in float32
    loses every digit of the variance"""
    torch = torch_gpu
    from snake_engine._lib import lib, check
    L, st = lib(), torch.cuda.current_stream().cuda_stream
    g = torch.Generator(device="cuda").manual_seed(7 + n)
    hw, C = 21, 128
    rows = n * hw * hw
    chan_mean = mean * (1.0 + 0.1 * torch.randn(C, device="cuda", generator=g))
    y = (torch.randn(rows, C, device="cuda", generator=g) * std + chan_mean).contiguous()
    gamma = torch.rand(C, device="cuda", generator=g) + 0.5
    beta = torch.randn(C, device="cuda", generator=g) * 0.3
    res = torch.randn(rows, C, device="cuda", generator=g) if with_res else None
    dout = torch.randn(rows, C, device="cuda", generator=g)
    mm = (chan_mean + 0.3 * std * torch.randn(C, device="cuda", generator=g)).contiguous()      # a moving mean near the batch mean
    mv = torch.ones(C, device="cuda")
    mm0, mv0 = mm.clone(), mv.clone()
    f = lambda k, dt=torch.float32: torch.empty(k, dtype=dt, device="cuda")
    part, sums = f(L.snk_bn_train_partials()), f(2 * C, torch.float64)
    m_, inv, sc, sh, tail = f(C), f(C), f(C), f(C), f(4)
    out = torch.empty_like(y)
    check(L.snk_bn_train_sums_f64(y.data_ptr(), rows, mm.data_ptr(), part.data_ptr(), sums.data_ptr(), st))
    check(L.snk_bn_train_finalize(sums.data_ptr(), float(rows), mm.data_ptr(), gamma.data_ptr(), beta.data_ptr(), mm.data_ptr(), mv.data_ptr(),
                                  0.99, 1e-3, m_.data_ptr(), inv.data_ptr(), sc.data_ptr(), sh.data_ptr(), C, st))
    bits = torch.zeros(rows * 32, dtype=torch.uint8, device="cuda")
    check(L.snk_bn_train_apply(y.data_ptr(), sc.data_ptr(), sh.data_ptr(), res.data_ptr() if with_res else None, out.data_ptr(), rows, 1,
                               part.data_ptr(), tail.data_ptr(), bits.data_ptr(), st))
    want_bits = ((out > 0).reshape(rows, 32, 4).to(torch.uint8) * torch.tensor([1, 2, 4, 8], dtype=torch.uint8, device="cuda")).sum(dim=2)
    assert torch.equal(bits.reshape(rows, 32), want_bits.to(torch.uint8))             # one byte per quad of channels: out > 0
    a, b, c, dg, db, tail_dx = f(C), f(C), f(C), f(C), f(C), f(4)
    dx, gres = torch.empty_like(y), torch.empty_like(y)
    check(L.snk_bn_train_grad_sums_f64(dout.data_ptr(), None, bits.data_ptr(), y.data_ptr(), m_.data_ptr(), inv.data_ptr(), rows, 1, part.data_ptr(),
                                       sums.data_ptr(), st))
    sums_from_sign = torch.empty_like(sums)                    # the other way to give the mask: the sign of a tensor
    check(L.snk_bn_train_grad_sums_f64(dout.data_ptr(), out.data_ptr(), None, y.data_ptr(), m_.data_ptr(), inv.data_ptr(), rows, 1, part.data_ptr(),
                                       sums_from_sign.data_ptr(), st))
    assert torch.equal(sums, sums_from_sign)
    check(L.snk_bn_train_grad_finalize(sums.data_ptr(), sums.data_ptr(), float(rows), gamma.data_ptr(), inv.data_ptr(), a.data_ptr(),
                                       b.data_ptr(), c.data_ptr(), dg.data_ptr(), db.data_ptr(), C, st))
    check(L.snk_bn_train_grad_apply(dout.data_ptr(), None, bits.data_ptr(), y.data_ptr(), m_.data_ptr(), inv.data_ptr(), a.data_ptr(), b.data_ptr(),
                                    c.data_ptr(), dx.data_ptr(), gres.data_ptr() if with_res else None, rows, 1, part.data_ptr(), tail_dx.data_ptr(), st))
    dx_sign = torch.empty_like(dx)
    check(L.snk_bn_train_grad_apply(dout.data_ptr(), out.data_ptr(), None, y.data_ptr(), m_.data_ptr(), inv.data_ptr(), a.data_ptr(), b.data_ptr(),
                                    c.data_ptr(), dx_sign.data_ptr(), None, rows, 1, part.data_ptr(), tail_dx.data_ptr(), st))
    assert torch.equal(dx, dx_sign)

    y64, g64, b64 = (t.double().requires_grad_(True) for t in (y, gamma, beta))
    r64 = res.double().requires_grad_(True) if with_res else None
    m64, v64 = y64.mean(dim=0), y64.var(dim=0, unbiased=False)
    z = (y64 - m64) / torch.sqrt(v64 + 1e-3) * g64 + b64
    m64, v64 = m64.detach(), v64.detach()
    if with_res:
        z = z + r64
    o64 = torch.relu(z)
    # the float32 run's mask (out > 0) is imposed on the float64 gradient: identical functions are differentiated
    ins64 = (y64, g64, b64) + ((r64,) if with_res else ())
    grads64 = torch.autograd.grad(z, ins64, dout.double() * (out > 0).double())
    assert float((m_.double() - m64).abs().max()) < 1e-6 * max(1.0, abs(mean))
    assert float(((1.0 / inv.double() ** 2 - 1e-3) - v64).abs().max() / v64.max()) < 2e-5
    assert float((out.double() - o64.detach()).abs().max()) < 3e-5 * max(1.0, abs(mean) / std / 20)
    tol = 2e-5 * max(1.0, abs(mean) / std / 20)            # xhat = (y - mean) inv in float32: the mean's rounding error counts in units of std
    assert rel(dx, grads64[0]) < tol and rel(dg, grads64[1]) < tol and rel(db, grads64[2]) < tol
    if with_res:
        assert rel(gres, grads64[3]) < 1e-6
    unb = v64 * rows / (rows - 1)
    assert float((mm.double() - (mm0.double() * 0.99 + m64 * 0.01)).abs().max()) < 1e-5 * max(1.0, abs(mean))
    assert float((mv.double() - (mv0.double() * 0.99 + unb * 0.01)).abs().max()) < 1e-6 * float(unb.max() + 1)
    amax = float(out.abs().max())
    assert 2048.0 <= amax * float(tail[2]) < 4096.0 and float(tail[2] * tail[3]) == 1.0       # the range handed to the next convolution
    assert 2048.0 <= float(dx.abs().max()) * float(tail_dx[2]) < 4096.0


def _net64(torch, ws, X, Y):
    """float64 autograd restatement of the graph on the GPU, returning loss (without the regularizer), the per-tensor
    gradients and the ReLU inputs of every layer"""
    import torch.nn.functional as F
    t = [torch.tensor(np.asarray(w), dtype=torch.float64, device="cuda") for w in ws]
    blocks = (len(ws) - 14) // 10
    n_conv = 2 + 2 * blocks
    pidx = []
    for l in range(n_conv):
        pidx += [5 * l, 5 * l + 1, 5 * l + 2]
    i = 5 * n_conv
    pidx += [i, i + 1, i + 2, i + 3]
    for j in pidx:
        t[j].requires_grad_(True)
    pre = {}

    def cbr(x, l, res=None):
        k = t[5 * l]
        y = F.conv2d(x, k.permute(3, 2, 0, 1), padding=k.shape[0] // 2)
        m, v = y.mean(dim=(0, 2, 3)), y.var(dim=(0, 2, 3), unbiased=False)
        z = (y - m[None, :, None, None]) / torch.sqrt(v[None, :, None, None] + 1e-3) * t[5 * l + 1][None, :, None, None] + t[5 * l + 2][None, :, None, None]
        if res is not None:
            z = z + res
        pre[l] = z
        return torch.relu(z)
    x = torch.as_tensor(X, dtype=torch.float64, device="cuda").permute(0, 3, 1, 2)
    h = cbr(x, 0)
    for b in range(blocks):
        sc = h
        h = cbr(h, 2 * b + 1)
        h = cbr(h, 2 * b + 2, sc)
    h = cbr(h, n_conv - 1)
    flat = h.permute(0, 2, 3, 1).reshape(h.shape[0], -1)
    a1 = flat @ t[i] + t[i + 1]
    pre["d1"] = a1
    q = torch.tanh(torch.relu(a1) @ t[i + 2] + t[i + 3])
    y = torch.as_tensor(Y, dtype=torch.float64, device="cuda")
    loss = ((q - y) ** 2).sum() / (3.0 * len(X))
    grads = torch.autograd.grad(loss, [t[j] for j in pidx])
    return float(loss), dict(zip(pidx, grads)), pre, q.detach(), n_conv


@pytest.mark.parametrize("n,blocks,hw", [(40, 2, 21), (23, 4, 21), (5, 1, 37), (9, 1, 13), (3, 10, 37)])
def test_one_whole_step_matches_float64_with_the_same_relu_masks(torch_gpu, n, blocks, hw):
    """hw = 37: BASELINE configs[4]'s 19 x 19 board (its weight gradient runs in slabs of five image rows; the last case is that
    config's 10-block net), 13: a 7 x 7 board"""
    torch = torch_gpu
    from snake_engine import net
    from snake_engine.train_step import TrainStep
    rs = np.random.RandomState(5 + n)
    X = rs.rand(n, hw, hw, 3).astype(np.float32)
    Y = np.tanh(rs.randn(n, 3)).astype(np.float32)
    ws = net.glorot_uniform_weights((hw, hw, 3), blocks=blocks, seed=9)
    for l in range(2 + 2 * blocks):                         # batch-norm parameters away from their initial 1 / 0
        ws[5 * l + 1] = (ws[5 * l + 1] * (0.6 + 0.8 * rs.rand(*ws[5 * l + 1].shape))).astype(np.float32)
        ws[5 * l + 2] = (0.2 * rs.randn(*ws[5 * l + 2].shape)).astype(np.float32)
    loss64, g64, pre, q64, n_conv = _net64(torch, ws, X, Y)
    ts = TrainStep(ws, (hw, hw, 3), n, "cuda")
    x, y = torch.as_tensor(X, device="cuda"), torch.as_tensor(Y, device="cuda")
    q = ts.forward(x, y, n)
    assert float((q.double() - q64).abs().max()) < 1e-5
    assert abs(float(ts.G[ts.n_params]) - loss64) < 1e-5 * loss64
    sign = lambda z: torch.where(z > 0, 1.0, -1.0).float().contiguous()
    for l in range(n_conv - 1):                             # the 128-channel layers: [n, C, h, w] float64 -> channels-last rows
        ts.mask_override[l] = sign(pre[l].permute(0, 2, 3, 1)).reshape(-1)
    ts.mask_override["h"] = sign(pre[n_conv - 1].permute(0, 2, 3, 1)).reshape(-1)
    ts.mask_override["d1"] = sign(pre["d1"]).reshape(-1)
    ts.backward(y, n)
    got = ts.gradients()
    worst = {}
    for j, ref in g64.items():
        worst[j] = float((torch.as_tensor(got[j], device="cuda").double() - ref).abs().max() / (ref.abs().max() + 1e-300))
    assert max(worst.values()) < 1e-4, sorted(worst.items(), key=lambda kv: -kv[1])[:5]
    # without the imposed masks the step still agrees to what flipped masks allow
    ts.mask_override.clear()
    ts.forward(x, y, n)
    ts.backward(y, n)
    got = ts.gradients()
    loose = max(float((torch.as_tensor(got[j], device="cuda").double() - ref).abs().max() / (ref.abs().max() + 1e-300)) for j, ref in g64.items())
    assert loose < 5e-2, loose


def test_adam_l2_and_loss_follow_the_keras_formulas(torch_gpu):
    """three steps of TrainStep.step against the float64 formulas (oracle/train_ref.py::KerasAdamRef on the float64 gradients
    of the same batch is covered on the CPU; here: the flat-buffer kernel itself on given gradients)"""
    torch = torch_gpu
    from snake_engine._lib import lib, check
    L, st = lib(), torch.cuda.current_stream().cuda_stream
    g = torch.Generator(device="cuda").manual_seed(3)
    n = 100_003
    w = torch.randn(n, device="cuda", generator=g)
    decay = (torch.rand(n, device="cuda", generator=g) < 0.7).to(torch.uint8)
    m, v = torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    w64, m64, v64 = w.double(), m.double(), v.double()
    part, l2 = torch.empty(1024, device="cuda"), torch.empty(1, device="cuda")
    for t in range(1, 4):
        grad = torch.randn(n, device="cuda", generator=g) * 10.0 ** (-t)
        lr = 1e-3 / t
        check(L.snk_l2_sum(w.data_ptr(), decay.data_ptr(), n, 1e-5, part.data_ptr(), l2.data_ptr(), st))
        assert abs(float(l2) - 1e-5 * float((w64 ** 2 * decay.double()).sum())) < 1e-6 * float(l2)
        lr_t = lr * np.sqrt(1 - 0.999 ** t) / (1 - 0.9 ** t)
        check(L.snk_adam_l2_step(w.data_ptr(), grad.data_ptr(), m.data_ptr(), v.data_ptr(), decay.data_ptr(), n, lr_t, 0.9, 0.999, 1e-7, 1e-5, st))
        g64 = grad.double() + 2e-5 * w64 * decay.double()
        m64 = 0.9 * m64 + 0.1 * g64
        v64 = 0.999 * v64 + 0.001 * g64 * g64
        w64 = w64 - lr_t * m64 / (v64.sqrt() + 1e-7)
        assert float((w.double() - w64).abs().max()) < 2e-6
        assert rel(m, m64) < 1e-6 and rel(v, v64) < 1e-6
    frozen = w.clone()
    check(L.snk_adam_l2_step(w.data_ptr(), grad.data_ptr(), m.data_ptr(), v.data_ptr(), decay.data_ptr(), n, 0.0, 0.9, 0.999, 1e-7, 1e-5, st))
    assert torch.equal(w, frozen)                           # learning rate 0 leaves every bit of every weight


def test_steps_at_learning_rate_zero_are_forward_only_and_change_nothing_else(torch_gpu, monkeypatch):
    """alpha_nnet.py:79-84: the rate is 0 after optimizer step 100 -- 220 of a generation's 320 steps.  With the dead backward
    passes skipped, the weights AND the batch-norm moving averages after the fit are bit-identical to the full path's."""
    torch = torch_gpu
    from snake_engine import net
    from utils import trainer_torch
    rs = np.random.RandomState(3)
    X = rs.rand(96, 21, 21, 3).astype(np.float32)
    Y = np.tanh(rs.randn(96, 3)).astype(np.float32)
    ws = net.glorot_uniform_weights((21, 21, 3), blocks=2, seed=5)
    sched = ([2, 4], [1e-3, 2.5e-4, 0.0])                   # 3 epochs x 3 batches = 9 steps, the last 4 at rate 0
    outs = {}
    for full in (False, True):
        monkeypatch.setattr(trainer_torch, "_DEAD_STEPS_FULL", full)
        outs[full] = trainer_torch.fit(ws, (21, 21, 3), X, Y, 3, 32, sched, seed=11, verbose=False)
        assert trainer_torch.fit.last_mode == "kernels"
        hist = list(trainer_torch.fit.last_history)
        outs[full] = (outs[full], hist)
    (wa, ha), (wb, hb) = outs[False], outs[True]
    assert all(np.array_equal(a, b) for a, b in zip(wa, wb))
    assert ha == hb and ha[2] < ha[0]
    moved = [float(np.abs(a - np.asarray(b)).max()) for a, b in zip(wa, ws)]
    assert min(moved[j] for j in (3, 4)) > 0                # moving statistics did move


def test_a_net_without_residual_blocks_is_fitted_off_the_kernels_and_says_so(torch_gpu):
    """stem -> head (blocks = 0; not a reference shape, alpha_nnet.py:25 builds four blocks) used to reach the batched
    weight-image preparation with zero layers (EngineError) or, with that switched off, skip the head's sums unseen.  Such a net
    is not a TrainStep shape: fit warns, runs on autograd and moves the weights."""
    from snake_engine import net, train_step
    from utils import trainer_torch
    assert train_step.supported((21, 21, 3), 4) and not train_step.supported((21, 21, 3), 0)
    rs = np.random.RandomState(4)
    X = rs.rand(16, 21, 21, 3).astype(np.float32)
    Y = np.tanh(rs.randn(16, 3)).astype(np.float32)
    ws = net.glorot_uniform_weights((21, 21, 3), blocks=0, seed=5)
    with pytest.warns(UserWarning, match="at least one residual block"):
        out = trainer_torch.fit(ws, (21, 21, 3), X, Y, 2, 8, ([100], [1e-3, 0.0]), seed=11, verbose=False)
    assert trainer_torch.fit.last_mode == "autograd"
    assert any(np.abs(a - np.asarray(b)).max() > 0 for a, b in zip(out, ws))
    with pytest.raises(ValueError, match="0 residual blocks"):
        train_step.TrainStep(ws, (21, 21, 3), 8, "cuda")


@pytest.mark.parametrize("n,blocks,hw", [(24, 2, 21), (6, 1, 37), (10, 2, 13)])
def test_batch_norm_backward_sums_in_the_input_gradient_epilogue_change_nothing_but_rounding(torch_gpu, monkeypatch, n, blocks, hw):
    """round 5: the input-gradient convolution of layer l also takes the two sums layer l - 1's batch-norm backward starts with
    (snk_conv3x3_f16s_igrad_stats) instead of snk_bn_train_grad_sums_f64 reading the gradient, the pre-batch-norm tensor and the
    mask bits again.  The gradient tensor it writes is bit-identical to the plain launch's; the sums agree with the separate pass up
    to float32 summation order, and so does every parameter gradient of a whole step (alpha_nnet.py:58-59)"""
    torch = torch_gpu
    from snake_engine import net, train_step
    from snake_engine._lib import lib, check
    rs = np.random.RandomState(n)
    X = torch.as_tensor(rs.rand(n, hw, hw, 3).astype(np.float32), device="cuda")
    Y = torch.as_tensor(np.tanh(rs.randn(n, 3)).astype(np.float32), device="cuda")
    ws = net.glorot_uniform_weights((hw, hw, 3), blocks=blocks, seed=4)
    grads = {}
    monkeypatch.setattr(train_step, "_DEFER_BN", False)     # the mask bytes of layer 1 are read below (deferred: tests further down)
    for fused in (True, False):
        monkeypatch.setattr(train_step, "_IGRAD_STATS", fused)
        ts = train_step.TrainStep(ws, (hw, hw, 3), n, "cuda")
        ts.forward(X, Y, n)
        ts.backward(Y, n)
        grads[fused] = ts.gradients()
    for j in grads[True]:
        a, b = grads[True][j], grads[False][j]
        assert np.abs(a - b).max() <= 2e-5 * max(np.abs(b).max(), 1e-12), (j, np.abs(a - b).max(), np.abs(b).max())
    # the entry point by itself: same gradient tensor, same sums as the separate pass
    L, st = lib(), torch.cuda.current_stream().cuda_stream
    l = 2                                              # the launch that produces the gradient at out[1] (no shortcut gradient) ...
    from snake_engine.train_step import _p
    dA1, dA2 = torch.empty_like(ts.dA), torch.empty_like(ts.dA)
    sums = torch.zeros(256, dtype=torch.float64, device="cuda")
    ref = torch.zeros(256, dtype=torch.float64, device="cuda")
    for res in (None, ts.gres):                        # ... and with one
        check(L.snk_conv3x3_bn_f16s(_p(ts.dY), _p(ts.img_b), _p(ts.ones), _p(ts.zeros), _p(res), _p(dA1), n, hw, hw, 0, st))
        check(L.snk_conv3x3_f16s_igrad_stats(_p(ts.dY), _p(ts.img_b), _p(res), _p(dA2), _p(ts.y[l - 1]), _p(ts.relu_mask[l - 1]),
                                             _p(ts.mean[l - 1]), _p(ts.inv[l - 1]), _p(ts.cv_partials), _p(sums), n, hw, hw, st))
        check(L.snk_bn_train_grad_sums_f64(_p(dA1), None, _p(ts.relu_mask[l - 1]), _p(ts.y[l - 1]), _p(ts.mean[l - 1]), _p(ts.inv[l - 1]),
                                           n * hw * hw, 1, _p(ts.partials), _p(ref), st))
        assert torch.equal(dA1[:n * hw * hw * 128], dA2[:n * hw * hw * 128])
        err = (sums - ref).abs().max().item()
        assert err <= 1e-5 * max(ref.abs().max().item(), 1e-12), (err, ref.abs().max().item())


@pytest.mark.parametrize("n,hw", [(5, 21), (3, 13), (2, 37)])
def test_input_gradient_epilogue_reads_nothing_behind_the_last_image(torch_gpu, monkeypatch, n, hw):
    """ADVICE round 5, finding 1.  The input-gradient epilogue (conv_split.hip, MODE 5 / 8) steps from row to row through the buffer
    instructions' SGPR soffset and relies on the image's descriptor to return zero for rows past the image (hw * hw is no multiple
    of 32 here: the last block of every image has such rows, and behind the LAST image lies whatever the allocator put there).
    LLVM documents soffset as excluded from the range check; on gfx950 it is included (tools/micro/store_hazard.hip,
    profiles/r6_store_hazard_micro.json: 64 of 64 lanes read zero).  This pins it where it matters: every tensor the epilogue
    reads is followed by NaN (mask bytes: all ones), the output by a sentinel -- sums and gradient are bit-identical to the run
    on ordinary buffers, finite, and nothing is written behind the output."""
    torch = torch_gpu
    from snake_engine import net, train_step
    from snake_engine._lib import lib, check
    from snake_engine.train_step import _p
    assert (hw * hw) % 32 != 0
    rs = np.random.RandomState(7 * n + hw)
    X = torch.as_tensor(rs.rand(n, hw, hw, 3).astype(np.float32), device="cuda")
    Y = torch.as_tensor(np.tanh(rs.randn(n, 3)).astype(np.float32), device="cuda")
    ws = net.glorot_uniform_weights((hw, hw, 3), blocks=2, seed=4)
    monkeypatch.setattr(train_step, "_DEFER_BN", False)
    ts = train_step.TrainStep(ws, (hw, hw, 3), n, "cuda")          # n == max_rows: nothing of the step's own lies behind image n - 1
    ts.forward(X, Y, n)
    ts.backward(Y, n)
    L, st = lib(), torch.cuda.current_stream().cuda_stream
    act, l = n * hw * hw * 128, 2
    slack = 64 * 128                                                # 64 rows: more than any block reaches past its image

    def padded(t, count, fill):
        buf = torch.full((count + (slack if t.dtype != torch.uint8 else slack // 4),), fill, dtype=t.dtype, device="cuda")
        buf[:count] = t.reshape(-1)[:count]
        return buf
    nan = float("nan")
    dY, res = padded(ts.dY, act, nan), padded(ts.gres, act, nan)
    y, mask = padded(ts.y[l - 1], act, nan), padded(ts.relu_mask[l - 1], act // 4, 255)
    rmask = padded(ts.relu_mask[l], act // 4, 255)
    for masked in (False, True):
        outs = []
        for pad in (False, True):
            out = torch.full((act + slack,), 12345.0, device="cuda")
            sums = torch.zeros(256, dtype=torch.float64, device="cuda")
            a = (dY, res, y, mask, rmask) if pad else (ts.dY, ts.gres, ts.y[l - 1], ts.relu_mask[l - 1], ts.relu_mask[l])
            if masked:
                check(L.snk_conv3x3_f16s_igrad_stats_masked_res(_p(a[0]), _p(ts.img_b), _p(a[1]), _p(a[4]), _p(out), _p(a[2]), _p(a[3]),
                                                                _p(ts.mean[l - 1]), _p(ts.inv[l - 1]), _p(ts.cv_partials), _p(sums), n, hw, hw, st))
            else:
                check(L.snk_conv3x3_f16s_igrad_stats(_p(a[0]), _p(ts.img_b), _p(a[1]), _p(out), _p(a[2]), _p(a[3]), _p(ts.mean[l - 1]),
                                                     _p(ts.inv[l - 1]), _p(ts.cv_partials), _p(sums), n, hw, hw, st))
            torch.cuda.synchronize()
            outs.append((out.clone(), sums.clone()))
        (o0, s0), (o1, s1) = outs
        assert torch.isfinite(s1).all() and torch.isfinite(o1[:act]).all()
        assert torch.equal(o0[:act], o1[:act]) and torch.equal(s0, s1)
        assert (o1[act:] == 12345.0).all() and (o0[act:] == 12345.0).all(), "the epilogue stored behind the last image"
        assert float(s1.abs().max()) > 0


@pytest.mark.parametrize("n,blocks,hw", [(24, 2, 21), (6, 1, 37), (10, 2, 13)])
def test_deferred_batch_norm_reads_the_same_values_as_the_written_activation(torch_gpu, monkeypatch, n, blocks, hw):
    """round 5: the activation between the two convolutions of a residual block is never written -- its three readers (the
    block's second convolution, that layer's weight gradient, the first layer's batch-norm backward) evaluate relu(y * scale +
    shift) themselves (snake_engine/train_step.py; alpha_nnet.py:25-47 under Keras fit).  Entry point by entry point against the
    written activation with the SAME input scales: bit-identical outputs; then two whole steps, one with and one without the
    deferral (only the power-of-two input range differs: a bound instead of the measured maximum)"""
    torch = torch_gpu
    from snake_engine import net, train_step
    from snake_engine._lib import lib, check
    from snake_engine.train_step import _p
    rs = np.random.RandomState(100 + n)
    X = torch.as_tensor(rs.rand(n, hw, hw, 3).astype(np.float32), device="cuda")
    Y = torch.as_tensor(np.tanh(rs.randn(n, 3)).astype(np.float32), device="cuda")
    ws = net.glorot_uniform_weights((hw, hw, 3), blocks=blocks, seed=6)
    for l in range(2 + 2 * blocks):                         # batch-norm parameters away from 1 / 0, some scales NEGATIVE
        g = ws[5 * l + 1] * (0.6 + 0.8 * rs.rand(*ws[5 * l + 1].shape))
        g[::7] *= -1.0
        ws[5 * l + 1] = g.astype(np.float32)
        ws[5 * l + 2] = (0.3 * rs.randn(*ws[5 * l + 2].shape)).astype(np.float32)
    L, st = lib(), torch.cuda.current_stream().cuda_stream
    assert L.snk_train_deferred_bn_supported(hw, hw) == 1
    runs = {}
    for defer in (False, True):
        monkeypatch.setattr(train_step, "_DEFER_BN", defer)
        ts = train_step.TrainStep(ws, (hw, hw, 3), n, "cuda")
        assert ts.defer == defer and (ts.out[1] is None) == defer and (ts.relu_mask[1] is None) == defer and ts.out[2] is not None
        q = ts.forward(X, Y, n).clone()
        ts.backward(Y, n)
        runs[defer] = (ts, q, ts.gradients(), ts.weights())
    ts, q0, g0, w0 = runs[False]
    td, q1, g1, w1 = runs[True]
    rows, act = n * hw * hw, n * hw * hw * 128
    # ---- the range: a bound of the written activation's maximum, at most a few binary orders above it
    true_max = float(ts.out[1][:act].max())
    k_meas, k_bound = float(ts.tail_out[1][2]), float(td.tail_out[1][2])
    assert 2.0 ** 11 <= true_max * k_meas < 2.0 ** 12
    assert true_max * k_bound < 2.0 ** 12 and k_bound >= k_meas / 64.0, (true_max, k_meas, k_bound)
    # ---- forward convolution of layer 2 reading y_1 through layer 1's scale / shift == reading the written out_1 (same image, same scale)
    y2 = torch.empty_like(ts.y[2])
    sums, ref = torch.zeros(256, dtype=torch.float64, device="cuda"), torch.zeros(256, dtype=torch.float64, device="cuda")
    y2_ref = torch.empty_like(ts.y[2])
    check(L.snk_conv3x3_f16s_stats(_p(ts.out[1]), _p(ts.img_f[2]), _p(y2_ref), None, _p(ts.cv_partials), _p(ref), n, hw, hw, st))
    check(L.snk_conv3x3_f16s_stats_deferred(_p(ts.y[1]), _p(ts.img_f[2]), _p(y2), None, _p(ts.scale[1]), _p(ts.shift[1]), None,
                                            _p(ts.cv_partials), _p(sums), n, hw, hw, st))
    assert torch.equal(y2[:act], y2_ref[:act]) and torch.equal(sums, ref)
    # ---- the maxima next to the sums: exact
    amax = torch.zeros(128, device="cuda")
    check(L.snk_conv3x3_f16s_stats_deferred(_p(ts.out[1]), _p(ts.img_f[2]), _p(y2), None, None, None, _p(amax), _p(ts.cv_partials),
                                            _p(sums), n, hw, hw, st))
    assert torch.equal(y2[:act], y2_ref[:act]) and torch.equal(sums, ref)
    assert torch.equal(amax, y2_ref[:act].view(rows, 128).abs().max(dim=0).values)
    # ---- backward: the state of the written run after its backward pass holds dY of layer 1 last; rebuild layer 2's pieces
    dy = torch.randn(act, device="cuda") * 1e-3
    tail_dy = _tail_of(torch, L, dy)
    dw_a, dw_b = torch.zeros(9 * 128 * 128, device="cuda"), torch.zeros(9 * 128 * 128, device="cuda")
    check(L.snk_conv3x3_wgrad_f16s(_p(ts.out[1]), _p(dy), _p(ts.tail_out[1]), _p(tail_dy), _p(ts.wg_partials), _p(dw_a), n, hw, hw, st))
    check(L.snk_conv3x3_wgrad_f16s_deferred(_p(ts.y[1]), _p(ts.scale[1]), _p(ts.shift[1]), _p(dy), _p(ts.tail_out[1]), _p(tail_dy),
                                            _p(ts.wg_partials), _p(dw_b), n, hw, hw, st))
    assert torch.equal(dw_a, dw_b) and float(dw_a.abs().max()) > 0
    k = ts._k(2)
    check(L.snk_conv3x3_prepare_weights_f16s_train(_p(ts.view[k]), _p(ts.img_b), _p(tail_dy), 1, _p(ts.img_f[2]), st))
    dA_a, dA_b = torch.empty(act, device="cuda"), torch.empty(act, device="cuda")
    check(L.snk_conv3x3_f16s_igrad_stats(_p(dy), _p(ts.img_b), None, _p(dA_a), _p(ts.y[1]), _p(ts.relu_mask[1]), _p(ts.mean[1]),
                                         _p(ts.inv[1]), _p(ts.cv_partials), _p(ref), n, hw, hw, st))
    check(L.snk_conv3x3_f16s_igrad_stats_deferred(_p(dy), _p(ts.img_b), None, _p(dA_b), _p(ts.y[1]), _p(ts.scale[1]), _p(ts.shift[1]),
                                                  _p(ts.mean[1]), _p(ts.inv[1]), _p(ts.cv_partials), _p(sums), n, hw, hw, st))
    assert torch.equal(dA_a, dA_b) and torch.equal(sums, ref)
    check(L.snk_bn_train_grad_sums_f64(_p(dA_a), None, _p(ts.relu_mask[1]), _p(ts.y[1]), _p(ts.mean[1]), _p(ts.inv[1]), rows, 1,
                                       _p(ts.partials), _p(ref), st))
    check(L.snk_bn_train_grad_sums_f64_deferred(_p(dA_a), _p(ts.y[1]), _p(ts.scale[1]), _p(ts.shift[1]), _p(ts.mean[1]), _p(ts.inv[1]), rows,
                                                _p(ts.partials), _p(sums), st))
    assert torch.equal(sums, ref)
    a, b, c = ts.abc[:128], ts.abc[128:256], ts.abc[256:]
    dx_a, dx_b, g_a, g_b = (torch.empty(act, device="cuda") for _ in range(4))
    t_a, t_b = torch.zeros(4, device="cuda"), torch.zeros(4, device="cuda")
    check(L.snk_bn_train_grad_apply(_p(dA_a), None, _p(ts.relu_mask[1]), _p(ts.y[1]), _p(ts.mean[1]), _p(ts.inv[1]), _p(a), _p(b), _p(c),
                                    _p(dx_a), _p(g_a), rows, 1, _p(ts.partials), _p(t_a), st))
    check(L.snk_bn_train_grad_apply_deferred(_p(dA_a), _p(ts.y[1]), _p(ts.scale[1]), _p(ts.shift[1]), _p(ts.mean[1]), _p(ts.inv[1]), _p(a),
                                             _p(b), _p(c), _p(dx_b), _p(g_b), rows, _p(ts.partials), _p(t_b), st))
    assert torch.equal(dx_a, dx_b) and torch.equal(g_a, g_b) and torch.equal(t_a, t_b)
    # ---- whole steps: Q, every gradient, the moving statistics
    assert float((q0 - q1).abs().max()) <= 2e-6
    for j in g0:
        assert np.abs(g0[j] - g1[j]).max() <= 2e-5 * max(np.abs(g0[j]).max(), 1e-12), (j, np.abs(g0[j] - g1[j]).max(), np.abs(g0[j]).max())
    for j, (u, v) in enumerate(zip(w0, w1)):
        assert np.abs(u - v).max() <= 1e-6 * max(np.abs(u).max(), 1e-12), j


@pytest.mark.parametrize("n,blocks,hw", [(24, 2, 21), (6, 1, 37), (10, 3, 13)])
def test_shortcut_gradient_through_the_mask_bytes_equals_the_masked_copy(torch_gpu, monkeypatch, n, blocks, hw):
    """round 5: the batch-norm backward of a block's second layer no longer writes dout * mask for the shortcut; the
    input-gradient launch of the block's first layer reads the gradient the block RECEIVED through the output's mask bytes
    (snk_conv3x3_f16s_igrad_stats_masked_res), in place.  Same values added in the same order: every gradient of a whole
    step is bit-identical, and so are the entry point's tensor and sums (alpha_nnet.py:25-47 under Keras fit)"""
    torch = torch_gpu
    from snake_engine import net, train_step
    from snake_engine._lib import lib, check
    from snake_engine.train_step import _p
    rs = np.random.RandomState(200 + n)
    X = torch.as_tensor(rs.rand(n, hw, hw, 3).astype(np.float32), device="cuda")
    Y = torch.as_tensor(np.tanh(rs.randn(n, 3)).astype(np.float32), device="cuda")
    ws = net.glorot_uniform_weights((hw, hw, 3), blocks=blocks, seed=8)
    grads = {}
    monkeypatch.setattr(train_step, "_DEFER_STEM", False)   # (layer 0's mask bytes are read below; the deferred stem has a test of its own)
    for masked in (True, False):
        monkeypatch.setattr(train_step, "_RES_MASK", masked)
        ts = train_step.TrainStep(ws, (hw, hw, 3), n, "cuda")
        assert ts.res_mask == masked
        ts.forward(X, Y, n)
        ts.backward(Y, n)
        grads[masked] = ts.gradients()
    for j in grads[True]:
        assert np.array_equal(grads[True][j], grads[False][j]), j
    # the entry point: shortcut rows through mask bytes == the masked copy as shortcut, also when the output overwrites the rows
    L, st = lib(), torch.cuda.current_stream().cuda_stream
    act = n * hw * hw * 128
    g = torch.Generator(device="cuda").manual_seed(n)
    R = torch.randn(act, device="cuda", generator=g) * 1e-3
    M = ts.relu_mask[2][:act // 4]
    bits = torch.stack([(M >> b) & 1 for b in range(4)], dim=1).reshape(-1).bool()
    RM = torch.where(bits, R, torch.zeros_like(R))
    out_a, out_b = torch.empty(act, device="cuda"), R.clone()
    sums, ref = torch.zeros(256, dtype=torch.float64, device="cuda"), torch.zeros(256, dtype=torch.float64, device="cuda")
    l = 1                                                  # the launch below layer 1: sums for layer 0 (the stem's batch norm)
    check(L.snk_conv3x3_f16s_igrad_stats(_p(ts.dY), _p(ts.img_b), _p(RM), _p(out_a), _p(ts.y[l - 1]), _p(ts.relu_mask[l - 1]),
                                         _p(ts.mean[l - 1]), _p(ts.inv[l - 1]), _p(ts.cv_partials), _p(ref), n, hw, hw, st))
    check(L.snk_conv3x3_f16s_igrad_stats_masked_res(_p(ts.dY), _p(ts.img_b), _p(out_b), _p(M), _p(out_b), _p(ts.y[l - 1]),
                                                    _p(ts.relu_mask[l - 1]), _p(ts.mean[l - 1]), _p(ts.inv[l - 1]), _p(ts.cv_partials),
                                                    _p(sums), n, hw, hw, st))
    assert torch.equal(out_a, out_b) and torch.equal(sums, ref) and float(ref.abs().max()) > 0


@pytest.mark.parametrize("n,hw", [(70, 21), (9, 13), (5, 37), (1, 21)])
def test_sums_taken_in_the_stem_and_the_head_on_their_way_out(torch_gpu, n, hw):
    """round 5: the stem's convolution leaves with the sums its batch norm starts from (snk_stem_conv_f32_stats), and the last
    tower layer's batch-norm kernel with the head's 1x1 convolution and ITS sums (snk_bn_train_apply_head) -- two passes over a
    462 MB tensor less in every step.  Outputs bit-identical to the separate kernels', sums equal up to summation order
    (alpha_nnet.py:21-24, 46-50 under Keras fit)"""
    torch = torch_gpu
    from snake_engine._lib import lib, check
    L, st = lib(), torch.cuda.current_stream().cuda_stream
    g = torch.Generator(device="cuda").manual_seed(n * hw)
    rows = n * hw * hw
    x = torch.rand(n, hw, hw, 3, device="cuda", generator=g)
    w = torch.randn(3, 3, 3, 128, device="cuda", generator=g) * 0.2
    cen = torch.randn(128, device="cuda", generator=g) * 0.1
    part = torch.empty(L.snk_bn_train_partials(), device="cuda")
    y_a, y_b = torch.empty(rows * 128, device="cuda"), torch.empty(rows * 128, device="cuda")
    s_a, s_b = torch.zeros(256, dtype=torch.float64, device="cuda"), torch.zeros(256, dtype=torch.float64, device="cuda")
    check(L.snk_stem_conv_f32(x.data_ptr(), w.data_ptr(), y_a.data_ptr(), n, hw, hw, st))
    check(L.snk_bn_train_sums_f64(y_a.data_ptr(), rows, cen.data_ptr(), part.data_ptr(), s_a.data_ptr(), st))
    check(L.snk_stem_conv_f32_stats(x.data_ptr(), w.data_ptr(), y_b.data_ptr(), cen.data_ptr(), part.data_ptr(), s_b.data_ptr(), n, hw, hw, st))
    assert torch.equal(y_a, y_b)
    ref = torch.cat([(y_a.view(rows, 128).double() - cen.double()).sum(0), ((y_a.view(rows, 128).double() - cen.double()) ** 2).sum(0)])
    assert float((s_b - ref).abs().max()) <= 1e-6 * float(ref.abs().max()) and float((s_a - ref).abs().max()) <= 1e-6 * float(ref.abs().max())
    # ---- the last layer's batch norm + shortcut + ReLU with the head's 1x1 stage
    y = torch.randn(rows * 128, device="cuda", generator=g)
    res = torch.randn(rows * 128, device="cuda", generator=g)
    sc, sh = torch.rand(128, device="cuda", generator=g) + 0.5, torch.randn(128, device="cuda", generator=g) * 0.3
    w1 = torch.randn(128, device="cuda", generator=g) * 0.1
    c1 = torch.full((1,), 0.37, device="cuda")
    o_a, o_b = torch.empty(rows * 128, device="cuda"), torch.empty(rows * 128, device="cuda")
    m_a, m_b = torch.zeros(rows * 32, dtype=torch.uint8, device="cuda"), torch.zeros(rows * 32, dtype=torch.uint8, device="cuda")
    z_a, z_b = torch.empty(rows, device="cuda"), torch.empty(rows, device="cuda")
    h_a, h_b = torch.zeros(2, dtype=torch.float64, device="cuda"), torch.zeros(2, dtype=torch.float64, device="cuda")
    tail = torch.zeros(4, device="cuda")
    check(L.snk_bn_train_apply(y.data_ptr(), sc.data_ptr(), sh.data_ptr(), res.data_ptr(), o_a.data_ptr(), rows, 1, part.data_ptr(),
                               tail.data_ptr(), m_a.data_ptr(), st))
    check(L.snk_head_conv1x1_sums(o_a.data_ptr(), w1.data_ptr(), rows, c1.data_ptr(), z_a.data_ptr(), part.data_ptr(), h_a.data_ptr(), st))
    check(L.snk_bn_train_apply_head(y.data_ptr(), sc.data_ptr(), sh.data_ptr(), res.data_ptr(), o_b.data_ptr(), rows, part.data_ptr(),
                                    m_b.data_ptr(), w1.data_ptr(), c1.data_ptr(), z_b.data_ptr(), h_b.data_ptr(), st))
    assert torch.equal(o_a, o_b) and torch.equal(m_a, m_b) and torch.equal(z_a, z_b)
    assert float((h_a - h_b).abs().max()) <= 1e-9 * float(h_a.abs().max())
    zr = (o_a.view(rows, 128).double() * w1.double()).sum(1)
    assert float((z_b.double() - zr).abs().max()) <= 1e-5 * float(zr.abs().max())
    # ---- and back: the head's expansion kernel leaves with the two sums the last layer's batch-norm backward starts from
    gz = torch.randn(rows, device="cuda", generator=g) * 1e-2
    mean_inv = torch.tensor([0.1, 1.7], device="cuda")
    abc = torch.tensor([0.9, 0.01, 0.02], device="cuda")
    mu, iv = torch.randn(128, device="cuda", generator=g) * 0.1, torch.rand(128, device="cuda", generator=g) + 0.5
    da_a, da_b = torch.empty(rows * 128, device="cuda"), torch.empty(rows * 128, device="cuda")
    dw_a, dw_b = torch.zeros(128, device="cuda"), torch.zeros(128, device="cuda")
    part2 = torch.empty(L.snk_bn_train_partials(), device="cuda")
    gs_a, gs_b = torch.zeros(256, dtype=torch.float64, device="cuda"), torch.zeros(256, dtype=torch.float64, device="cuda")
    check(L.snk_head_conv1x1_bwd(gz.data_ptr(), z_a.data_ptr(), mean_inv.data_ptr(), abc.data_ptr(), o_a.data_ptr(), w1.data_ptr(),
                                 da_a.data_ptr(), dw_a.data_ptr(), part.data_ptr(), rows, st))
    check(L.snk_bn_train_grad_sums_f64(da_a.data_ptr(), None, m_a.data_ptr(), y.data_ptr(), mu.data_ptr(), iv.data_ptr(), rows, 1,
                                       part.data_ptr(), gs_a.data_ptr(), st))
    check(L.snk_head_conv1x1_bwd_stats(gz.data_ptr(), z_a.data_ptr(), mean_inv.data_ptr(), abc.data_ptr(), o_a.data_ptr(), w1.data_ptr(),
                                       da_b.data_ptr(), dw_b.data_ptr(), part.data_ptr(), y.data_ptr(), m_a.data_ptr(), mu.data_ptr(),
                                       iv.data_ptr(), part2.data_ptr(), gs_b.data_ptr(), rows, st))
    assert torch.equal(da_a, da_b) and torch.equal(dw_a, dw_b)
    assert float((gs_a - gs_b).abs().max()) <= 1e-9 * float(gs_a.abs().max()) and float(gs_a.abs().max()) > 0


@pytest.mark.parametrize("n,blocks,hw", [(24, 2, 21), (6, 1, 37), (10, 3, 13)])
def test_deferred_stem_reads_the_same_values_as_the_written_activation(torch_gpu, monkeypatch, n, blocks, hw):
    """round 5: the stem's batch norm + ReLU output deferred as well -- read from the stem's y by the first tower convolution (which
    also takes its own maxima), by that layer's weight gradient (covered above), by the first block's shortcut
    (snk_bn_train_apply_res_deferred) and, as a sign, by the launch that takes the stem's batch-norm backward sums
    (snk_conv3x3_f16s_igrad_stats_masked_res_deferred).  Entry points bit-identical to their written-activation forms; whole
    steps equal up to the input range's power of two (alpha_nnet.py:21-31 under Keras fit)"""
    torch = torch_gpu
    from snake_engine import net, train_step
    from snake_engine._lib import lib, check
    from snake_engine.train_step import _p
    rs = np.random.RandomState(300 + n)
    X = torch.as_tensor(rs.rand(n, hw, hw, 3).astype(np.float32), device="cuda")
    Y = torch.as_tensor(np.tanh(rs.randn(n, 3)).astype(np.float32), device="cuda")
    ws = net.glorot_uniform_weights((hw, hw, 3), blocks=blocks, seed=7)
    for l in range(2 + 2 * blocks):
        g_ = ws[5 * l + 1] * (0.6 + 0.8 * rs.rand(*ws[5 * l + 1].shape))
        g_[::5] *= -1.0
        ws[5 * l + 1] = g_.astype(np.float32)
        ws[5 * l + 2] = (0.3 * rs.randn(*ws[5 * l + 2].shape)).astype(np.float32)
    L, st = lib(), torch.cuda.current_stream().cuda_stream
    runs = {}
    for stem in (False, True):
        monkeypatch.setattr(train_step, "_DEFER_STEM", stem)
        ts = train_step.TrainStep(ws, (hw, hw, 3), n, "cuda")
        assert ts.defer and ts.defer_stem == stem and (ts.out[0] is None) == stem and (ts.relu_mask[0] is None) == stem
        q = ts.forward(X, Y, n).clone()
        ts.backward(Y, n)
        runs[stem] = (ts, q, ts.gradients(), ts.weights())
    ts, q0, g0, w0 = runs[False]
    td, q1, g1, w1 = runs[True]
    rows, act = n * hw * hw, n * hw * hw * 128
    true_max = float(ts.out[0][:act].max())
    k_meas, k_bound = float(ts.tail_out[0][2]), float(td.tail_out[0][2])
    assert true_max * k_bound < 2.0 ** 12 and k_bound >= k_meas / 64.0, (true_max, k_meas, k_bound)
    # ---- the stem's maxima: exact
    y0 = torch.empty(act, device="cuda")
    amax = torch.zeros(128, device="cuda")
    sums = torch.zeros(256, dtype=torch.float64, device="cuda")
    check(L.snk_stem_conv_f32_stats_deferred(_p(X), _p(ts.view[0]), _p(y0), None, _p(amax), _p(ts.partials), _p(sums), n, hw, hw, st))
    assert torch.equal(y0, ts.y[0][:act]) and torch.equal(amax, y0.view(rows, 128).abs().max(dim=0).values)
    # ---- layer 1's forward convolution reading y_0 through the stem's scale / shift AND taking its own maxima
    y1, y1_ref = torch.empty(act, device="cuda"), torch.empty(act, device="cuda")
    ref = torch.zeros(256, dtype=torch.float64, device="cuda")
    check(L.snk_conv3x3_f16s_stats(_p(ts.out[0]), _p(ts.img_f[1]), _p(y1_ref), None, _p(ts.cv_partials), _p(ref), n, hw, hw, st))
    check(L.snk_conv3x3_f16s_stats_deferred(_p(ts.y[0]), _p(ts.img_f[1]), _p(y1), None, _p(ts.scale[0]), _p(ts.shift[0]), _p(amax),
                                            _p(ts.cv_partials), _p(sums), n, hw, hw, st))
    assert torch.equal(y1, y1_ref) and torch.equal(sums, ref) and torch.equal(amax, y1_ref.view(rows, 128).abs().max(dim=0).values)
    # ---- the first block's shortcut read from y_0
    o_a, o_b = torch.empty(act, device="cuda"), torch.empty(act, device="cuda")
    m_a, m_b = torch.zeros(rows * 32, dtype=torch.uint8, device="cuda"), torch.zeros(rows * 32, dtype=torch.uint8, device="cuda")
    t_a, t_b = torch.zeros(4, device="cuda"), torch.zeros(4, device="cuda")
    check(L.snk_bn_train_apply(_p(ts.y[2]), _p(ts.scale[2]), _p(ts.shift[2]), _p(ts.out[0]), _p(o_a), rows, 1, _p(ts.partials), _p(t_a), _p(m_a), st))
    check(L.snk_bn_train_apply_res_deferred(_p(ts.y[2]), _p(ts.scale[2]), _p(ts.shift[2]), _p(ts.y[0]), _p(ts.scale[0]), _p(ts.shift[0]), _p(o_b),
                                            rows, _p(ts.partials), _p(t_b), _p(m_b), st))
    assert torch.equal(o_a, o_b) and torch.equal(m_a, m_b) and torch.equal(t_a, t_b) and torch.equal(o_a, ts.out[2][:act])
    # ---- the launch below layer 1: shortcut rows through layer 2's mask bytes, the stem's sums with the recomputed ReLU decision
    g = torch.Generator(device="cuda").manual_seed(n)
    R = torch.randn(act, device="cuda", generator=g) * 1e-3
    out_a, out_b = R.clone(), R.clone()
    check(L.snk_conv3x3_f16s_igrad_stats_masked_res(_p(ts.dY), _p(ts.img_b), _p(out_a), _p(ts.relu_mask[2]), _p(out_a), _p(ts.y[0]),
                                                    _p(ts.relu_mask[0]), _p(ts.mean[0]), _p(ts.inv[0]), _p(ts.cv_partials), _p(ref), n, hw, hw, st))
    check(L.snk_conv3x3_f16s_igrad_stats_masked_res_deferred(_p(ts.dY), _p(ts.img_b), _p(out_b), _p(ts.relu_mask[2]), _p(out_b), _p(ts.y[0]),
                                                             _p(ts.scale[0]), _p(ts.shift[0]), _p(ts.mean[0]), _p(ts.inv[0]), _p(ts.cv_partials),
                                                             _p(sums), n, hw, hw, st))
    assert torch.equal(out_a, out_b) and torch.equal(sums, ref) and float(ref.abs().max()) > 0
    # ---- whole steps
    assert float((q0 - q1).abs().max()) <= 2e-6
    for j in g0:
        assert np.abs(g0[j] - g1[j]).max() <= 2e-5 * max(np.abs(g0[j]).max(), 1e-12), (j, np.abs(g0[j] - g1[j]).max(), np.abs(g0[j]).max())
    for j, (u, v) in enumerate(zip(w0, w1)):
        assert np.abs(u - v).max() <= 1e-6 * max(np.abs(u).max(), 1e-12), j


def test_fit_on_the_kernels_tracks_fit_with_library_operators(torch_gpu):
    """utils.trainer_torch.fit on the same rows and shuffle order on this library's kernels and with every operator from
    PyTorch / MIOpen (SNK_TRAIN_CONV=torch).  Adam's first steps move every weight by about lr * sign(gradient), so weights
    whose gradient is near zero may part by 2 lr between two float32 implementations: what has to agree is the loss of
    every epoch and what the trained nets predict"""
    import os
    import subprocess
    import sys
    import tempfile
    from conftest import REPO
    code = r'''
import sys, numpy as np, torch
sys.path[:0] = [r"%s", r"%s/alphasnake-zero_amd"]
from snake_engine import net
from utils import trainer_torch
rs = np.random.RandomState(3)
X = rs.rand(512, 21, 21, 3).astype(np.float32); Y = np.tanh(rs.randn(512, 3)).astype(np.float32)
ws = net.glorot_uniform_weights((21, 21, 3), blocks=4, seed=5)
out = trainer_torch.fit(ws, (21, 21, 3), X, Y, 3, 128, ([4, 8], [1e-3, 2.5e-4, 0.0]), seed=11, verbose=False)
q = trainer_torch._Net(out, torch.device("cuda")).forward(torch.as_tensor(X[:96], device="cuda"), False).detach().cpu().numpy()
np.savez(sys.argv[1], hist=np.array(trainer_torch.fit.last_history), q=q, mode=trainer_torch.fit.last_mode)
''' % (REPO, REPO)
    res = {}
    with tempfile.TemporaryDirectory() as d:
        for mode in ("native", "torch"):
            path = os.path.join(d, mode + ".npz")
            r = subprocess.run([sys.executable, "-c", code, path], env=dict(os.environ, SNK_TRAIN_CONV=mode), capture_output=True,
                               text=True, timeout=600)
            assert r.returncode == 0, r.stderr[-2000:]
            z = np.load(path)
            res[mode] = (z["hist"], z["q"], str(z["mode"]))
    (h_a, q_a, m_a), (h_b, q_b, m_b) = res["native"], res["torch"]
    assert (m_a, m_b) == ("kernels", "autograd")
    assert len(h_a) == 3 and h_a[2] < h_a[0]                                    # it trains
    assert np.abs(h_a - h_b).max() / np.abs(h_b).max() < 1e-2, (h_a, h_b)
    assert np.abs(q_a - q_b).max() < 5e-2 and np.abs(q_a - q_b).mean() < 5e-3, (np.abs(q_a - q_b).max(), np.abs(q_a - q_b).mean())


def test_two_rank_fit_on_the_kernels_equals_the_one_rank_fit(torch_gpu, tmp_path):
    """data-parallel fit (rows r::2 of every batch per rank, float64 batch-norm sums and ONE gradient bucket all-reduced, here
    over gloo with both ranks on the box's GPU) against the same fit in one process: both ranks end bit-identical, and equal to
    the one-rank result up to float32 summation order -- 6 optimizer steps, the last one at rate 0 (forward only)"""
    import os
    import socket
    import subprocess
    import sys
    from conftest import REPO
    helper = os.path.join(REPO, "tests", "helpers", "fit_rank.py")
    env1 = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, helper, str(tmp_path)], env=env1, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = [subprocess.Popen([sys.executable, helper, str(tmp_path)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                              env=dict(env1, RANK=str(k), WORLD_SIZE="2", LOCAL_RANK=str(k), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port)))
             for k in range(2)]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs[0][-2000:] + outs[1][-2000:]
    one, a, b = (np.load(tmp_path / f) for f in ("fit_w1_r0.npz", "fit_w2_r0.npz", "fit_w2_r1.npz"))
    keys = [k for k in one.files if k not in ("hist", "q")]
    for k in keys + ["hist", "q"]:
        assert np.array_equal(a[k], b[k]), k                                         # the ranks agree bit for bit
    # Against the one-rank run: not "to rounding".  The two runs' batch-norm scales differ in the last bit (sums added in another
    # order), ONE ReLU mask of 3.6 M flips on that, the flipped element's gradient moves its channel's sum(g) -- a sum with
    # hundredfold cancellation -- by a percent, and that offset reaches every row of the layer's input gradient (measured with
    # tools/diag_ddp_step.py: all tensors of a step equal to 1e-9 up to that element, the first tower layer's weight gradient
    # then 1.5e-2 apart); Adam's first steps turn that into weights up to a few 1e-3 apart (each moved by up to 6e-3 here).
    # What has to agree is what training produced: the loss of every epoch and what the trained nets predict.
    assert np.abs(a["hist"] - one["hist"]).max() / np.abs(one["hist"]).max() < 1e-2, (a["hist"], one["hist"])
    dq = np.abs(a["q"] - one["q"])
    print(f"2 ranks vs 1: loss history {np.abs(a['hist'] - one['hist']).max() / np.abs(one['hist']).max():.2e}, |dq| max {dq.max():.2e} mean {dq.mean():.2e}, "
          f"weights {max(float(np.abs(a[k] - one[k]).max()) for k in keys):.2e}")
    assert dq.max() < 5e-2 and dq.mean() < 1e-2, (dq.max(), dq.mean())              # measured: 1.2e-2 / 2.6e-3
    assert max(float(np.abs(a[k] - one[k]).max()) for k in keys) < 2e-2              # measured: 4e-3
    from snake_engine.net import glorot_uniform_weights
    start = glorot_uniform_weights((21, 21, 3), blocks=2, seed=5)
    assert max(float(np.abs(one[k] - w).max()) for k, w in zip(keys, start)) > 1e-3  # it trained
