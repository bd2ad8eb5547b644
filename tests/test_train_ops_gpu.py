"""snake_engine.train_ops.SplitConv3x3 (the tower convolution of the training half on k_conv3x3_f16s, SURVEY.md section 8
row f-1) against float64 torch convolutions: forward, input gradient, weight gradient; float32-grade tolerances."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_gpu():
    import torch
    assert torch.cuda.is_available()
    return torch


@pytest.mark.parametrize("n,hw,xmag,gmag", [(64, 21, 1.0, 1.0), (48, 21, 30.0, 1e-6), (5, 13, 1e-3, 50.0), (3, 37, 1.0, 1e-4)])
def test_split_conv3x3_forward_and_gradients_match_float64(torch_gpu, n, hw, xmag, gmag):
    torch = torch_gpu
    import torch.nn.functional as F
    from snake_engine import train_ops
    g = torch.Generator(device="cuda").manual_seed(n)
    x = (torch.randn(n, hw, hw, 128, device="cuda", generator=g) * xmag).permute(0, 3, 1, 2).requires_grad_(True)   # channels-last
    k = (torch.randn(3, 3, 128, 128, device="cuda", generator=g) * 0.05).requires_grad_(True)
    dy = torch.randn(n, hw, hw, 128, device="cuda", generator=g).permute(0, 3, 1, 2) * gmag
    dy = dy * (torch.rand(n, 1, hw, hw, device="cuda", generator=g) ** 6)            # a wide spread of magnitudes, as real gradients have
    assert train_ops.usable(x, k)
    y = train_ops.SplitConv3x3.apply(x, k)
    dx, dk = torch.autograd.grad(y, (x, k), dy)
    x64, k64 = x.detach().double().requires_grad_(True), k.detach().double().requires_grad_(True)
    y64 = F.conv2d(x64, k64.permute(3, 2, 0, 1), padding=1)
    dx64, dk64 = torch.autograd.grad(y64, (x64, k64), dy.double())

    def rel(a, b):
        return float((a.detach().double() - b.detach()).abs().max() / b.detach().abs().max())
    assert y.shape == y64.shape and y.is_contiguous(memory_format=torch.channels_last)
    assert rel(y, y64) < 2e-6, rel(y, y64)
    assert rel(dx, dx64) < 2e-6, rel(dx, dx64)
    assert rel(dk, dk64) < 2e-5, rel(dk, dk64)                      # the library's float32 weight gradient (a sum over n*h*w terms)
    # the float32 library convolution is no closer to float64 than the split-f16 kernel
    y32 = F.conv2d(x.detach(), k.detach().permute(3, 2, 0, 1), padding=1)
    assert rel(y, y64) <= 4 * rel(y32, y64) + 1e-7


def test_zero_input_and_unusable_shapes(torch_gpu):
    torch = torch_gpu
    from snake_engine import train_ops
    x = torch.zeros(4, 128, 21, 21, device="cuda").contiguous(memory_format=torch.channels_last)
    k = torch.randn(3, 3, 128, 128, device="cuda")
    assert float(train_ops.SplitConv3x3.apply(x, k).abs().max()) == 0.0
    assert not train_ops.usable(x, torch.randn(3, 3, 3, 128, device="cuda"))          # the stem
    assert not train_ops.usable(x.double(), k.double()) and not train_ops.usable(x.cpu(), k.cpu())


def test_fit_with_native_convolutions_tracks_fit_with_library_convolutions(torch_gpu):
    """utils.trainer_torch.fit on the same rows and shuffle order with the tower convolutions on k_conv3x3_f16s and with
    every convolution through the library.  Adam's first steps move every weight by about lr * sign(gradient), so weights
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
np.savez(sys.argv[1], hist=np.array(trainer_torch.fit.last_history), q=q)
''' % (REPO, REPO)
    res = {}
    with tempfile.TemporaryDirectory() as d:
        for mode in ("native", "torch"):
            path = os.path.join(d, mode + ".npz")
            r = subprocess.run([sys.executable, "-c", code, path], env=dict(os.environ, SNK_TRAIN_CONV=mode), capture_output=True,
                               text=True, timeout=600)
            assert r.returncode == 0, r.stderr[-2000:]
            z = np.load(path)
            res[mode] = (z["hist"], z["q"])
    (h_a, q_a), (h_b, q_b) = res["native"], res["torch"]
    assert len(h_a) == 3 and h_a[2] < h_a[0]                                    # it trains
    assert np.abs(h_a - h_b).max() / np.abs(h_b).max() < 1e-2, (h_a, h_b)
    assert np.abs(q_a - q_b).max() < 5e-2 and np.abs(q_a - q_b).mean() < 5e-3, (np.abs(q_a - q_b).max(), np.abs(q_a - q_b).mean())


@pytest.mark.parametrize("with_res,n", [(False, 33), (True, 64)])
def test_fused_batch_norm_act_matches_float64(torch_gpu, with_res, n):
    """snake_engine.train_ops.FusedBatchNormAct (csrc/train.hip) against the same expression in float64 autograd:
    output, batch statistics, and the gradients with respect to the input, gamma, beta and the residual"""
    torch = torch_gpu
    from snake_engine import train_ops
    g = torch.Generator(device="cuda").manual_seed(7 + n)
    hw = 21
    y = (torch.randn(n, hw, hw, 128, device="cuda", generator=g) * 3.0 + 0.7).permute(0, 3, 1, 2).requires_grad_(True)
    gamma = (torch.rand(128, device="cuda", generator=g) + 0.5).requires_grad_(True)
    beta = (torch.randn(128, device="cuda", generator=g) * 0.3).requires_grad_(True)
    res = torch.randn(n, hw, hw, 128, device="cuda", generator=g).permute(0, 3, 1, 2).requires_grad_(True) if with_res else None
    dout = torch.randn(n, hw, hw, 128, device="cuda", generator=g).permute(0, 3, 1, 2)
    assert train_ops.bn_usable(y)
    out, mean, var, cnt = train_ops.FusedBatchNormAct.apply(y, gamma, beta, res, True, None)
    ins = (y, gamma, beta) + ((res,) if with_res else ())
    grads = torch.autograd.grad(out, ins, dout)

    y64, g64, b64 = (t.detach().double().requires_grad_(True) for t in (y, gamma, beta))
    r64 = res.detach().double().requires_grad_(True) if with_res else None
    m64 = y64.mean(dim=(0, 2, 3))
    v64 = y64.var(dim=(0, 2, 3), unbiased=False)
    z = (y64 - m64[None, :, None, None]) / torch.sqrt(v64[None, :, None, None] + 1e-3) * g64[None, :, None, None] + b64[None, :, None, None]
    if with_res:
        z = z + r64
    o64 = torch.relu(z)
    ins64 = (y64, g64, b64) + ((r64,) if with_res else ())
    grads64 = torch.autograd.grad(o64, ins64, dout.double())
    assert float(cnt) == n * hw * hw
    assert float((mean.detach().double() - m64.detach()).abs().max()) < 1e-5
    assert float((var.detach().double() - v64.detach()).abs().max() / v64.detach().max()) < 1e-5
    assert float((out.double() - o64).abs().max()) < 2e-5
    for a, b, name in zip(grads, grads64, ("dy", "dgamma", "dbeta", "dres")):
        err = float((a.double() - b).abs().max() / b.abs().max())
        assert err < 2e-5, (name, err)


def test_one_training_step_on_the_kernels_matches_float64_autograd(torch_gpu, monkeypatch):
    """loss and every parameter gradient of one batch: utils.trainer_torch._Net on the GPU in float32 with the tower
    convolutions on k_conv3x3_f16s and the batch norms on csrc/train.hip, against the same graph on the CPU in float64 --
    and against the GPU float32 run with the library's operators, which sets the scale of float32 round-off for this graph
    (a ReLU mask that flips on round-off changes a gradient by far more than round-off: 1e-3 .. 1e-2 of the largest entry
    in either float32 run; measured, the kernels are the closer of the two in 29 of 34 tensors)"""
    torch = torch_gpu
    from snake_engine import net
    from utils import trainer_torch
    rs = np.random.RandomState(5)
    X = rs.rand(96, 21, 21, 3).astype(np.float32)
    Y = np.tanh(rs.randn(96, 3)).astype(np.float32)
    ws = net.glorot_uniform_weights((21, 21, 3), blocks=4, seed=9)

    def run(dev, dt, native):
        monkeypatch.setattr(trainer_torch, "_NATIVE_CONV", native)
        m = trainer_torch._Net(ws, dev, dt)
        x, y = torch.as_tensor(X, dtype=dt, device=dev), torch.as_tensor(Y, dtype=dt, device=dev)
        pred = m.forward(x, True)
        loss = ((pred - y) ** 2).sum() / (3.0 * len(x)) + m.l2()
        grads = torch.autograd.grad(loss, m.params())
        return float(loss.detach()), [g.detach().double().cpu() for g in grads], [t.detach().double().cpu() for t in m.t]
    l_nat, g_nat, t_nat = run(torch.device("cuda"), torch.float32, True)
    l_lib, g_lib, t_lib = run(torch.device("cuda"), torch.float32, False)
    l_64, g_64, t_64 = run(torch.device("cpu"), torch.float64, False)
    assert abs(l_nat - l_64) < 1e-5 * abs(l_64)

    def errs(gs):
        return [float((a - b).abs().max() / (b.abs().max() + 1e-30)) for a, b in zip(gs, g_64)]
    e_nat, e_lib = errs(g_nat), errs(g_lib)
    assert max(e_nat) < 5e-2, max(e_nat)                        # ReLU masks flip on round-off: the library path shows 1e-3 .. 1e-2 too
    assert max(e_nat) <= 3.0 * max(e_lib) + 1e-6, (max(e_nat), max(e_lib))
    assert float(np.median(e_nat)) <= 3.0 * float(np.median(e_lib)) + 1e-7, (np.median(e_nat), np.median(e_lib))
    for a, b in zip(t_nat, t_64):                                # the moving averages moved the same way
        assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max()) + 1e-7
