"""runs the whole Q-net forward on mid-game observations a few times (for rocprofv3 passes: --kernel-trace --stats, or one
--pmc counter per pass): tower_only.py [games 2300] [forwards 3] [board 11]     (SNK_CONV_RECT=0: the full form of every layer;
board 19 = 8 snakes, 10 blocks: BASELINE configs[4]'s shape; SNK_CONV_ALGO selects the tower)
prints the number of observations, so that counter sums can be divided into per-state-and-layer figures"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "alphasnake-zero_amd")]
import torch
import snake_engine as se
from snake_engine import net

games = int(sys.argv[1]) if len(sys.argv) > 1 else 2300
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
board = int(sys.argv[3]) if len(sys.argv) > 3 else 11
snakes, blocks = (8, 10) if board == 19 else (4, 4)
hw = 2 * board - 1
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
qn = net.QNet(net.glorot_uniform_weights((hw, hw, 3), blocks=blocks, seed=0), (hw, hw, 3), max_chunk=1 << 20)
qn.backgrounds()
torch.cuda.synchronize()
for _ in range(reps):
    q = qn.forward(planes, mask)
torch.cuda.synchronize()
print(f"observations {planes.shape[0]} forwards {reps} n_rect {qn.n_rect} layers {2 * qn.blocks} algo {qn.conv_algo} board {board}")
