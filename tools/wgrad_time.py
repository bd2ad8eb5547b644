"""k_wgrad_f16s alone: time and error against a float64 weight gradient.  Development tool: wgrad_time.py [images [hw]]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "alphasnake-zero_amd")]
import numpy as np, torch
import snake_engine._lib as _l
_l.LIB_PATH = os.environ.get("OBS_LIB", _l.LIB_PATH)      # a development build of the library
from snake_engine import train_ops
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
hw = int(sys.argv[2]) if len(sys.argv) > 2 else 21
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.relu(torch.randn(n, hw, hw, 128, device="cuda", generator=g))
dy = torch.randn(n, hw, hw, 128, device="cuda", generator=g) * 1e-4
xt, dt = train_ops._input_scale(x), train_ops._input_scale(dy)
dk = train_ops._wgrad(x, dy, xt, dt)
torch.cuda.synchronize()
ts = []
for _ in range(10):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); train_ops._wgrad(x, dy, xt, dt); b.record(); torch.cuda.synchronize()
    ts.append(a.elapsed_time(b))
t = float(np.median(ts)) * 1e-3
fl = 2.0 * n * hw * hw * 9 * 128 * 128
print(f"k_wgrad_f16s n={n} {hw}x{hw}: {t * 1e3:.3f} ms, {fl / t / 1e12:.1f} TFLOP/s algorithmic")
m = min(n, 64)
ref = torch.ops.aten.convolution_backward(dy[:m].double().permute(0, 3, 1, 2), x[:m].double().permute(0, 3, 1, 2),
                                          torch.zeros(128, 128, 3, 3, dtype=torch.float64, device="cuda"), None, [1, 1], [1, 1], [1, 1],
                                          False, [0, 0], 1, [False, True, False])[1].permute(2, 3, 1, 0)
got = train_ops._wgrad(x[:m].contiguous(), dy[:m].contiguous(), xt, dt)
print("max rel err vs float64 (first %d images): %.2e" % (m, float((got.double() - ref).abs().max() / ref.abs().max())))
