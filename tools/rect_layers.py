"""per-layer times of the tower on mid-game observations, full form against sub-rectangle form, with the GEMM tiles each
executes:   rect_layers.py [board 11] [games 2300] [n_rect]      (SNK_LIB_PATH selects a variant build)"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "alphasnake-zero_amd")]
import numpy as np
import torch
import snake_engine as se
from snake_engine import net
from snake_engine._lib import check

# rect_layers.py 11 2300 6 centre: the same work in both forms at layer 3 (what the form itself costs)
board = int(sys.argv[1]) if len(sys.argv) > 1 else 11
games = int(sys.argv[2]) if len(sys.argv) > 2 else 2300
snakes, blocks = (8, 10) if board == 19 else (4, 4)
h = w = 2 * board - 1
if len(sys.argv) > 3:
    os.environ["SNK_CONV_RECT_LAYERS"] = sys.argv[3]
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
planes, _, _ = eng.observe_all(pairs)
if len(sys.argv) > 4 and sys.argv[4] == "centre":     # every window at the canvas centre: layer 3's rectangle is the whole canvas
    planes = torch.tensor([0.0, 1.0, 0.0], device="cuda").repeat(planes.shape[0], h, w, 1).contiguous()
    lo = (h - board) // 2
    planes[:, lo:lo + board, lo:lo + board] = torch.rand((planes.shape[0], board, board, 3), device="cuda", generator=g)
m = planes.shape[0]
ws = net.glorot_uniform_weights((h, w, 3), blocks=blocks, seed=0)
qn = net.QNet(ws, (h, w, 3), max_chunk=1 << 20)
st = torch.cuda.current_stream().cuda_stream
bufs = [torch.empty((m, h, w, 128), device="cuda") for _ in range(3)]
qn.backgrounds()


def run(use_plan, reps=6):
    n_layers = 2 * blocks - 1
    times = np.zeros((reps, n_layers + 2))
    counts = None
    for r in range(reps):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(n_layers + 3)]
        ev[0].record()
        plan = qn._rect_plan(planes, m, 0, st) if use_plan else None
        ev[1].record()
        check(qn.L.snk_stem_conv_bn_relu_f32(planes.data_ptr(), qn.stem_w.data_ptr(), qn.stem_sc.data_ptr(), qn.stem_sh.data_ptr(),
                                             bufs[0].data_ptr(), m, h, w, st))
        ev[2].record()
        cur, t1, t2 = bufs
        for i in range(n_layers):
            if i % 2 == 0:
                qn._conv(i, cur, None, t1, m, st, plan=plan)
            else:
                qn._conv(i, t1, cur, t2, m, st, plan=plan)
                cur, t2 = t2, cur
            ev[3 + i].record()
        torch.cuda.synchronize()
        times[r] = [ev[k].elapsed_time(ev[k + 1]) for k in range(n_layers + 2)]
        if plan is not None:
            counts = plan[1].cpu().numpy()
    return np.median(times[1:], axis=0), counts


full, _ = run(False)
rect, counts = run(True)
T = (h * w + 31) // 32
print(f"{board}x{board}: {m} observations, n_rect = {qn.n_rect}; plan {rect[0]*1e3:.0f} us, stem {rect[1]*1e3:.0f} us")
for i in range(len(full) - 2):
    if i < qn.n_rect:
        print(f"layer {i}: full {full[2+i]:.3f} ms   rect {rect[2+i]:.3f} ms  = {rect[2+i]/full[2+i]:.3f}   tiles {counts[i,1]/(m*T):.3f} of full, "
              f"{counts[i,0]/m:.2f} blocks per image")
    else:
        print(f"layer {i}: full {full[2+i]:.3f} ms   (full) {rect[2+i]:.3f} ms")
print(f"tower (without the last layer): full {full[2:].sum():.3f} ms, rect {rect[2:].sum() + rect[0]:.3f} ms incl. plan  = {(rect[2:].sum() + rect[0]) / full[2:].sum():.3f}")
