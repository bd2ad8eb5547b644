"""one training batch: parameter-gradient errors of the GPU float32 paths (kernels / library) against CPU float64"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "alphasnake-zero_amd")]
import numpy as np, torch
from snake_engine import net
from utils import trainer_torch
rs = np.random.RandomState(5)
X = rs.rand(96, 21, 21, 3).astype(np.float32); Y = np.tanh(rs.randn(96, 3)).astype(np.float32)
ws = net.glorot_uniform_weights((21, 21, 3), blocks=4, seed=9)
def run(dev, dt, native):
    trainer_torch._NATIVE_CONV = native
    m = trainer_torch._Net(ws, dev, dt)
    x, y = torch.as_tensor(X, dtype=dt, device=dev), torch.as_tensor(Y, dtype=dt, device=dev)
    pred = m.forward(x, True)
    loss = ((pred - y) ** 2).sum() / (3.0 * len(x)) + m.l2()
    return float(loss.detach()), [g.detach().double().cpu() for g in torch.autograd.grad(loss, m.params())]
l_nat, g_nat = run(torch.device("cuda"), torch.float32, True)
l_lib, g_lib = run(torch.device("cuda"), torch.float32, False)
l_64, g_64 = run(torch.device("cpu"), torch.float64, False)
print("loss", l_nat, l_lib, l_64)
for i, (a, b, c) in enumerate(zip(g_nat, g_lib, g_64)):
    s = float(c.abs().max()) + 1e-30
    print(i, tuple(c.shape), f"native {float((a - c).abs().max()) / s:.2e}  library {float((b - c).abs().max()) / s:.2e}  max|g| {s:.2e}")
