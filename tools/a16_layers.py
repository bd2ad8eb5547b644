"""per-layer times of the tower with 16-bit activations (SNK_CONV_ALGO=bf16 | f16a) on mid-game observations, full form
against sub-rectangle form, from HIP events around every conv launch of QNet.forward:
    a16_layers.py [board 19] [games 600] [reps 5]          (SNK_LIB_PATH selects a variant build)"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "alphasnake-zero_amd")]
os.environ.setdefault("SNK_CONV_ALGO", "bf16")
import numpy as np
import torch
import snake_engine as se
from snake_engine import net

board = int(sys.argv[1]) if len(sys.argv) > 1 else 19
games = int(sys.argv[2]) if len(sys.argv) > 2 else 600
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
snakes, blocks = (8, 10) if board == 19 else (4, 4)
h = w = 2 * board - 1
eng = se.Engine(games, board, board, snakes, 1, 0.15, seed=7)
eng.reset()
g = torch.Generator(device="cuda").manual_seed(7)
for _ in range(14):
    pairs = torch.nonzero(eng.alive()).to(torch.int32).contiguous()
    _, mask, _ = eng.observe_all(pairs, want_planes=False, want_key=False)
    pick = torch.multinomial((mask == 0).to(torch.float32) + 1e-3, 1, generator=g).squeeze(1).to(torch.uint8)
    mv = torch.ones((games, snakes), dtype=torch.uint8, device="cuda")
    mv[pairs[:, 0].long(), pairs[:, 1].long()] = pick
    eng.step(mv)
pairs = torch.nonzero(eng.alive()).to(torch.int32).contiguous()
planes, mask, _ = eng.observe_all(pairs)
m = planes.shape[0]
qn = net.QNet(net.glorot_uniform_weights((h, w, 3), blocks=blocks, seed=0), (h, w, 3), max_chunk=1 << 20)
qn.backgrounds()
n_layers = 2 * blocks
flops_layer = 2.0 * m * h * w * 9 * 128 * 128


def run(rect):
    qn.rect_min = 48 if rect else 1 << 30
    out = None
    t = np.zeros((reps, n_layers))
    wall = np.zeros(reps)
    for r in range(reps):
        qn.conv_timing = []
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        out = qn.forward(planes, mask)
        b.record()
        torch.cuda.synchronize()
        t[r] = [e0.elapsed_time(e1) for e0, e1, _ in qn.conv_timing]
        wall[r] = a.elapsed_time(b)
    qn.conv_timing = None
    return np.median(t[1:], axis=0), np.median(wall[1:]), out


full, wall_full, q_full = run(False)
rect, wall_rect, q_rect = run(True)
print(f"{board}x{board} {qn.conv_algo}: {m} observations, {n_layers} tower layers, n_rect = {qn.n_rect}; Q equal in both forms: {bool(torch.equal(q_full, q_rect))}")
for i in range(n_layers):
    print(f"layer {i:2d}: full {full[i]:.3f} ms = {flops_layer / full[i] / 1e9:6.0f} TFLOP/s   "
          f"{'rect' if i < qn.n_rect else 'full'} {rect[i]:.3f} ms = {flops_layer / rect[i] / 1e9:6.0f} TFLOP/s-equivalent")
print(f"tower: full {full.sum():.3f} ms = {n_layers * flops_layer / full.sum() / 1e9:.0f} TFLOP/s; with the sub-rectangle layers "
      f"{rect.sum():.3f} ms = {n_layers * flops_layer / rect.sum() / 1e9:.0f} TFLOP/s-equivalent; forward wall {wall_full:.3f} / {wall_rect:.3f} ms")
