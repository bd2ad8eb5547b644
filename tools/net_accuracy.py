"""max |dQ| of the whole Q-net on the GPU, per conv algorithm, against a float64 PyTorch-CPU evaluation of the same
Keras graph (alpha_nnet.py:19-56) on recorded observations (tests/golden/states_11x11x4.npz)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "alphasnake-zero_amd")]
import numpy as np, torch, torch.nn.functional as F
from snake_engine import net

def forward64(ws, states):
    t = [torch.as_tensor(np.asarray(w)).double() for w in ws]
    blocks = (len(t) - 14) // 10
    x = torch.as_tensor(states).double().permute(0, 3, 1, 2)
    conv = lambda x, k: F.conv2d(x, k.permute(3, 2, 0, 1), padding=k.shape[0] // 2)
    bn = lambda x, g, b, m, v: F.batch_norm(x, m, v, g, b, training=False, eps=1e-3)
    h = F.relu(bn(conv(x, t[0]), *t[1:5]))
    for blk in range(blocks):
        b0 = 5 + 10 * blk; sc = h
        h = F.relu(bn(conv(h, t[b0]), *t[b0 + 1:b0 + 5]))
        h = F.relu(bn(conv(h, t[b0 + 5]), *t[b0 + 6:b0 + 10]) + sc)
    b0 = 5 + 10 * blocks
    h = F.relu(bn(conv(h, t[b0]), *t[b0 + 1:b0 + 5]))
    h = h.permute(0, 2, 3, 1).reshape(h.shape[0], -1)
    h = F.relu(h @ t[b0 + 5] + t[b0 + 6])
    return torch.tanh(h @ t[b0 + 7] + t[b0 + 8]).numpy()

s = np.load(os.path.join(REPO, "tests", "golden", "states_11x11x4.npz"))
states = s["raw"][:128]
rng = np.random.RandomState(5)
for name in ("gen-0 Glorot (bench net)", "randomised BN"):
    ws = net.glorot_uniform_weights((21, 21, 3), blocks=4, seed=0)
    if name != "gen-0 Glorot (bench net)":
        k = 0
        while k < len(ws):
            if ws[k].ndim == 4:
                n = ws[k].shape[3]
                gam = 1.0 + 0.2 * rng.randn(n)
                if "small" in name: gam *= 0.25
                ws[k + 1:k + 5] = [gam.astype(np.float32), (0.1 * rng.randn(n)).astype(np.float32) * (0.25 if "small" in name else 1.0),
                                   (0.05 * rng.randn(n)).astype(np.float32), (0.5 + rng.rand(n)).astype(np.float32)]
                k += 5
            else:
                k += 1
    ref = forward64(ws, states)
    line = f"{name:28s} max|Q| {np.abs(ref).max():.3f}:"
    for algo in ("direct", "winograd", "f16s"):
        os.environ["SNK_CONV_ALGO"] = algo
        q = net.QNet(ws, (21, 21, 3)).forward(torch.as_tensor(states, device="cuda")).cpu().numpy().astype(np.float64)
        line += f"  {algo} {np.abs(q - ref).max():.2e}"
    print(line, flush=True)
