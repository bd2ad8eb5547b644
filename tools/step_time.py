"""k_step on mid-game boards (32 warm-up ticks of uniform legal moves): time and HBM fraction with and without food spawning.
Development tool: step_time.py [games]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "alphasnake-zero_amd")]
import numpy as np, torch
import snake_engine as se
n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
eng = se.Engine(n, 11, 11, 4, 1, 0.15, seed=1234)
eng.reset()
g = torch.Generator(device="cuda").manual_seed(1234)
sub = torch.arange(n, dtype=torch.int32, device="cuda").repeat_interleave(4)
pairs = torch.stack([sub, torch.arange(4, dtype=torch.int32, device="cuda").repeat(n)], dim=1).contiguous()
blocked = torch.empty((4 * n, 3), dtype=torch.uint8, device="cuda")
def legal():
    eng.observe(pairs, 4 * n, None, blocked, None)
    r = torch.rand((4 * n, 3), device="cuda", generator=g) - 2.0 * blocked.float()
    mv = torch.where(blocked.bool().all(dim=1), torch.ones((), dtype=torch.int64, device="cuda"), r.argmax(dim=1))
    return mv.to(torch.uint8).reshape(n, 4).contiguous()
for _ in range(32):
    eng.step(legal())
snap = se.Engine(n, 11, 11, 4, 1, 0.15)
eng.clone_to(snap)
live = int((eng.alive().sum(dim=1) > 1).sum().item())
mv = legal()
G = eng.slot_bytes
for chance in (0.15, 0.0):
    eng.set_params(food_spawn_chance=chance)
    ts = []
    for _ in range(20):
        snap.clone_to(eng); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); eng.step(mv); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e-3)
    t = float(np.median(ts))
    print(f"{n} games ({live} unfinished), spawn chance {chance}: {t * 1e6:.1f} us, {(n + live) * G / t / 1e12:.2f} TB/s = {(n + live) * G / t / 8e12:.3f} of 8 TB/s")
