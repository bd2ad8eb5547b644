"""k_observe on mid-game boards (32 warm-up ticks of uniform legal moves): planes form and mask + key form, time and HBM
fraction.  Development tool: observe_time.py [games [planes|mask+key|planes+mask+key]]      (OBS_BOARD=19: 19x19 / 8 snakes)"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "alphasnake-zero_amd")]
import numpy as np, torch
import snake_engine as se
import snake_engine._lib as _l
_l.LIB_PATH = os.environ.get("OBS_LIB", _l.LIB_PATH)      # a development build of the library
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
only = sys.argv[2] if len(sys.argv) > 2 else None      # time one form only (for counter passes)
B = int(os.environ.get("OBS_BOARD", "11"))
S = 8 if B == 19 else 4
OBS = 2 * B - 1
eng = se.Engine(n, B, B, S, 1, 0.15, seed=1234)
eng.reset()
g = torch.Generator(device="cuda").manual_seed(1234)
sub = torch.arange(n, dtype=torch.int32, device="cuda").repeat_interleave(S)
allp = torch.stack([sub, torch.arange(S, dtype=torch.int32, device="cuda").repeat(n)], dim=1).contiguous()
blocked = torch.empty((S * n, 3), dtype=torch.uint8, device="cuda")
for _ in range(32):
    eng.observe(allp, S * n, None, blocked, None)
    r = torch.rand((S * n, 3), device="cuda", generator=g) - 2.0 * blocked.float()
    mv = torch.where(blocked.bool().all(dim=1), torch.ones((), dtype=torch.int64, device="cuda"), r.argmax(dim=1))
    eng.step(mv.to(torch.uint8).reshape(n, S).contiguous())
pairs = torch.nonzero(eng.alive()).to(torch.int32).contiguous()
m = pairs.shape[0]
planes = torch.empty((m, OBS, OBS, 3), device="cuda")
mask = torch.empty((m, 3), dtype=torch.uint8, device="cuda")
key = torch.empty((m, 2), dtype=torch.int64, device="cuda")
G = eng.slot_bytes
def timed(fn, iters=10):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e-3)
    return float(np.median(ts))
for name, fn, byts in (("planes", lambda: eng.observe(pairs, m, planes, None, None), m * (G + OBS * OBS * 12)),
                       ("mask+key", lambda: eng.observe(pairs, m, None, mask, key), m * (G + 19)),
                       ("planes+mask+key", lambda: eng.observe(pairs, m, planes, mask, key), m * (G + OBS * OBS * 12 + 19))):
    if only and name != only:
        continue
    t = timed(fn)
    print(f"{m} observations, {name}: {t * 1e6:.1f} us, {byts / t / 1e12:.2f} TB/s = {byts / t / 8e12:.3f} of 8 TB/s")
print("checksum", int(key.sum().item()) & 0xFFFFFFFF, float(planes.sum().item()), int(mask.sum().item()))
