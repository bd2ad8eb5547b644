"""runs only the HBM-bound engine kernels (for rocprofv3 counter passes): engine_only.py [games]
   k_step, k_clone, k_observe (planes) and k_observe (mask + key) on `games` 11x11 / 4-snake boards after 16 warm-up ticks."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "alphasnake-zero_amd")]
import torch
import snake_engine as se
n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
eng = se.Engine(n, 11, 11, 4, 1, 0.15, seed=1234)
eng.reset()
g = torch.Generator(device="cuda").manual_seed(1234)
for _ in range(16):
    eng.step(torch.randint(0, 3, (n, 4), device="cuda", generator=g, dtype=torch.int32).to(torch.uint8))
snap = se.Engine(n, 11, 11, 4, 1, 0.15)
eng.clone_to(snap)
pairs = torch.nonzero(eng.alive()).to(torch.int32).contiguous()
m = min(pairs.shape[0], 65536)
pairs = pairs[:m].contiguous()
planes = torch.empty((m, 21, 21, 3), device="cuda")
mask = torch.empty((m, 3), dtype=torch.uint8, device="cuda")
key = torch.empty((m, 2), dtype=torch.int64, device="cuda")
mv = torch.randint(0, 3, (n, 4), device="cuda", generator=g, dtype=torch.int32).to(torch.uint8)
torch.cuda.synchronize()
for _ in range(5):
    snap.clone_to(eng)
    eng.step(mv)
    eng.observe(pairs, m, planes, None, None)
    eng.observe(pairs, m, None, mask, key)
torch.cuda.synchronize()
print(f"games {n} slot_bytes {eng.slot_bytes} observations {m}")
