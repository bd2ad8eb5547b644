import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "alphasnake-zero_amd")]
import numpy as np, torch
from snake_engine import net, train_step
from snake_engine._lib import lib, check
from snake_engine.train_step import _p
n, blocks, hw = 24, 2, 21
rs = np.random.RandomState(n)
X = torch.as_tensor(rs.rand(n, hw, hw, 3).astype(np.float32), device="cuda")
Y = torch.as_tensor(np.tanh(rs.randn(n, 3)).astype(np.float32), device="cuda")
ws = net.glorot_uniform_weights((hw, hw, 3), blocks=blocks, seed=4)
train_step._DEFER_BN = False; train_step._RES_MASK = False; train_step._IGRAD_STATS = False
ts = train_step.TrainStep(ws, (hw, hw, 3), n, "cuda")
ts.forward(X, Y, n); ts.backward(Y, n)
L, st = lib(), torch.cuda.current_stream().cuda_stream
act = n * hw * hw * 128
l = 2
dA1, dA2 = torch.zeros(act, device="cuda"), torch.zeros(act, device="cuda")
sums = torch.zeros(256, dtype=torch.float64, device="cuda"); ref = torch.zeros(256, dtype=torch.float64, device="cuda")
for res in (None, ts.gres):
    check(L.snk_conv3x3_bn_f16s(_p(ts.dY), _p(ts.img_b), _p(ts.ones), _p(ts.zeros), _p(res), _p(dA1), n, hw, hw, 0, st))
    check(L.snk_conv3x3_f16s_igrad_stats(_p(ts.dY), _p(ts.img_b), _p(res), _p(dA2), _p(ts.y[l - 1]), _p(ts.relu_mask[l - 1]),
                                         _p(ts.mean[l - 1]), _p(ts.inv[l - 1]), _p(ts.cv_partials), _p(sums), n, hw, hw, st))
    check(L.snk_bn_train_grad_sums_f64(_p(dA1), None, _p(ts.relu_mask[l - 1]), _p(ts.y[l - 1]), _p(ts.mean[l - 1]), _p(ts.inv[l - 1]),
                                       n * hw * hw, 1, _p(ts.partials), _p(ref), st))
    torch.cuda.synchronize()
    d = (dA1 - dA2).abs().view(n, hw * hw, 128)
    bad = (d > 0).nonzero()
    print("res", res is not None, "dA max diff", float(d.max()), "n bad", len(bad), "first bad", bad[:5].tolist(), "sum err", float((sums - ref).abs().max()), float(ref.abs().max()))
    if len(bad):
        px = torch.unique(bad[:, 1]); print(" bad pixels", px[:40].tolist(), "count", len(px)); print(" bad images", torch.unique(bad[:, 0])[:10].tolist())
print("---- MODE 7 vs MODE 5 (no shortcut)")
dA3 = torch.zeros(act, device="cuda"); s7 = torch.zeros(256, dtype=torch.float64, device="cuda")
check(L.snk_conv3x3_f16s_igrad_stats(_p(ts.dY), _p(ts.img_b), None, _p(dA2), _p(ts.y[1]), _p(ts.relu_mask[1]), _p(ts.mean[1]), _p(ts.inv[1]),
                                     _p(ts.cv_partials), _p(sums), n, hw, hw, st))
check(L.snk_conv3x3_f16s_igrad_stats_deferred(_p(ts.dY), _p(ts.img_b), None, _p(dA3), _p(ts.y[1]), _p(ts.scale[1]), _p(ts.shift[1]),
                                              _p(ts.mean[1]), _p(ts.inv[1]), _p(ts.cv_partials), _p(s7), n, hw, hw, st))
torch.cuda.synchronize()
d = (dA2 - dA3).abs().view(n, hw * hw, 128); bad = (d > 0).nonzero()
print("dA max diff", float(d.max()), "n bad", len(bad), bad[:5].tolist(), "sums diff", float((sums - s7).abs().max()), float(sums.abs().max()))
ds = (sums - s7).abs(); print(" sums bad idx", (ds > 0).nonzero().view(-1)[:20].tolist())
if len(bad):
    px = torch.unique(bad[:, 1]); print(" bad pixels", px[:40].tolist(), "count", len(px)); print(" bad images", torch.unique(bad[:, 0])[:10].tolist())
print("---- MODE 8 vs MODE 5 with the masked copy")
g = torch.Generator(device="cuda").manual_seed(n)
R = torch.randn(act, device="cuda", generator=g) * 1e-3
M = ts.relu_mask[2][:act // 4]
bits = torch.stack([(M >> b) & 1 for b in range(4)], dim=1).reshape(-1).bool()
RM = torch.where(bits, R, torch.zeros_like(R))
for inplace in (False, True):
    out_b = R.clone() if inplace else torch.zeros(act, device="cuda")
    s8 = torch.zeros(256, dtype=torch.float64, device="cuda")
    check(L.snk_conv3x3_f16s_igrad_stats(_p(ts.dY), _p(ts.img_b), _p(RM), _p(dA2), _p(ts.y[0]), _p(ts.relu_mask[0]), _p(ts.mean[0]), _p(ts.inv[0]),
                                         _p(ts.cv_partials), _p(sums), n, hw, hw, st))
    check(L.snk_conv3x3_f16s_igrad_stats_masked_res(_p(ts.dY), _p(ts.img_b), _p(out_b if inplace else R), _p(M), _p(out_b), _p(ts.y[0]),
                                                    _p(ts.relu_mask[0]), _p(ts.mean[0]), _p(ts.inv[0]), _p(ts.cv_partials), _p(s8), n, hw, hw, st))
    torch.cuda.synchronize()
    d = (dA2 - out_b).abs().view(n, hw * hw, 128); bad = (d > 0).nonzero()
    print("inplace", inplace, "dA max diff", float(d.max()), "n bad", len(bad), bad[:5].tolist(), "sums diff", float((sums - s8).abs().max()))
    if len(bad):
        px = torch.unique(bad[:, 1]); print(" bad pixels", px[:40].tolist(), "count", len(px)); print(" bad images", torch.unique(bad[:, 0])[:10].tolist())
