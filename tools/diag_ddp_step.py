"""development tool: gradients of ONE TrainStep step, one process against two gloo ranks sharing GPU 0"""
import os, sys, socket, subprocess
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "alphasnake-zero_amd")]
import numpy as np
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    torch.cuda.set_device(0)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("gloo")
    from snake_engine import net
    from snake_engine.train_step import TrainStep
    rs = np.random.RandomState(3)
    X = rs.rand(64, 21, 21, 3).astype(np.float32); Y = np.tanh(rs.randn(64, 3)).astype(np.float32)
    ws = net.glorot_uniform_weights((21, 21, 3), blocks=2, seed=5)
    class TS(TrainStep):
        log = {}
        def _bn_backward(self, l, n, count, want_res, tail):
            dA_sum = float(self.dA[:n * self.hw * 128].double().sum()); dA_abs = float(self.dA[:n * self.hw * 128].double().abs().sum())
            if l in (1, 2):
                self.log[f"dAfull{l}"] = self.dA[:n * self.hw * 128].cpu().numpy().reshape(n, -1)
            super()._bn_backward(l, n, count, want_res, tail)
            if l in (1, 2):
                self.log[f"dYfull{l}"] = self.dY[:n * self.hw * 128].cpu().numpy().reshape(n, -1)
                self.log[f"taildy{l}"] = self.tail_dy.cpu().numpy()
            self.log[f"sums{l}"] = self.sums.cpu().numpy().copy()
            self.log[f"local{l}"] = (self.sums_local if self.dist is not None else self.sums).cpu().numpy().copy()
            self.log[f"abc{l}"] = self.abc.cpu().numpy().copy()
            self.log[f"dA{l}"] = np.array([dA_sum, dA_abs])
            self.log[f"dY{l}"] = np.array([float(self.dY[:n * self.hw * 128].double().sum()), float(self.dY[:n * self.hw * 128].double().abs().sum())])
    ts = TS(ws, (21, 21, 3), 64, "cuda", dist)
    x = torch.as_tensor(X[rank::world], device="cuda").contiguous(); y = torch.as_tensor(Y[rank::world], device="cuda").contiguous()
    ts.forward(x, y, 64); ts.backward(y, 64); ts._all_reduce(ts.G)
    g = ts.gradients()
    nn = x.shape[0] * 441 * 128
    for l in (0, 1, 2):
        ts.log[f"y{l}"] = ts.y[l][:nn].cpu().numpy().reshape(x.shape[0], -1)
        if ts.out[l] is not None:
            ts.log[f"out{l}"] = ts.out[l][:nn].cpu().numpy().reshape(x.shape[0], -1)
        ts.log[f"tail{l}"] = ts.tail_out[l].cpu().numpy()
    np.savez(sys.argv[2] + f"/log_w{world}_r{rank}.npz", **ts.log)
    np.savez(sys.argv[2] + f"/g_w{world}_r{rank}.npz", **{str(k): v for k, v in g.items()}, mse=ts.G[ts.n_params].cpu().numpy(),
             mm=np.concatenate([ts.moving[k].cpu().numpy() for k in sorted(ts.moving)]))
    if world > 1:
        dist.barrier(); dist.destroy_process_group()
    sys.exit(0)
import tempfile
d = tempfile.mkdtemp()
env1 = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
subprocess.run([sys.executable, __file__, "child", d], env=env1, check=True)
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
ps = [subprocess.Popen([sys.executable, __file__, "child", d], env=dict(env1, RANK=str(k), WORLD_SIZE="2", LOCAL_RANK=str(k), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))) for k in range(2)]
assert all(p.wait() == 0 for p in ps)
one, a, b = (np.load(f"{d}/{f}") for f in ("g_w1_r0.npz", "g_w2_r0.npz", "g_w2_r1.npz"))
l1, la, lb = (np.load(f"{d}/{f}") for f in ("log_w1_r0.npz", "log_w2_r0.npz", "log_w2_r1.npz"))
for k in sorted(l1.files):
    if k.startswith("sums") or k.startswith("abc"):
        sc = np.abs(l1[k]).max() + 1e-30
        h = len(l1[k]) // (2 if k.startswith("sums") else 3)
        print(k, "first part %.2e" % (np.abs(l1[k][:h] - la[k][:h]).max() / sc), "rest %.2e" % (np.abs(l1[k][h:] - la[k][h:]).max() / sc))
    else:
        print(k, "1 rank", l1[k], " 2 ranks (sum of both)", la[k] + lb[k])
for l in (0, 1, 2):
    for r, lg in ((0, la), (1, lb)):
        y1, y2 = l1[f"y{l}"][r::2], lg[f"y{l}"]
        o1, o2 = l1[f"out{l}"][r::2], lg[f"out{l}"]
        print(f"layer {l} rank {r}: max|dy|/max|y| {np.abs(y1 - y2).max() / np.abs(y1).max():.2e}  mask flips {int(((o1 > 0) != (o2 > 0)).sum())} of {o1.size}  max|dout| {np.abs(o1 - o2).max():.2e}  tails", l1[f"tail{l}"][2], lg[f"tail{l}"][2])
for name in ("dAfull2", "dYfull2", "dAfull1", "dYfull1"):
    for r, lg in ((0, la), (1, lb)):
        u, v = l1[name][r::2].astype(np.float64), lg[name].astype(np.float64)
        d = (v - u).reshape(u.shape[0], 441, 128)
        print(f"{name} rank {r}: max|d| {np.abs(d).max():.2e} of max {np.abs(u).max():.2e}; mean|u| {np.abs(u).mean():.2e}; per-channel mean diff max {np.abs(d.mean(axis=(0, 1))).max():.2e}; rms diff {np.sqrt((d ** 2).mean()):.2e}", lg.get("taildy" + name[-1], None))
k = "local1"
print("1 rank Σg[:6]", l1[k][:6]); print("rank0", la[k][:6]); print("rank1", lb[k][:6]); print("r0+r1", (la[k] + lb[k])[:6]); print("global r0", la["sums1"][:6])
dd = np.abs(l1[k][:128] - (la[k] + lb[k])[:128]); print("worst channels", np.argsort(-dd)[:5], dd.max(), "of", np.abs(l1[k][:128]).max())
import torch
sys.path.insert(0, os.path.join(REPO, "tests"))
from test_train_ops_gpu import _net64
from snake_engine import net
rs = np.random.RandomState(3)
X = rs.rand(64, 21, 21, 3).astype(np.float32); Y = np.tanh(rs.randn(64, 3)).astype(np.float32)
ws = net.glorot_uniform_weights((21, 21, 3), blocks=2, seed=5)
loss64, g64, pre, q64, n_conv = _net64(torch, ws, X, Y)
for k in one.files:
    if k.isdigit():
        ref = g64[int(k)].cpu().numpy()
        sc = np.abs(ref).max() + 1e-30
        print("vs float64", k, " 1 rank: %.2e" % (np.abs(one[k] - ref).max() / sc), " 2 ranks: %.2e" % (np.abs(a[k] - ref).max() / sc))
for k in one.files:
    ref = np.abs(one[k]).max() + 1e-30
    print(k, one[k].shape, "1 vs 2 ranks: %.2e" % (np.abs(one[k] - a[k]).max() / ref), " rank0 vs rank1: %.2e" % (np.abs(a[k] - b[k]).max() / ref), " |g|max %.2e" % ref)
