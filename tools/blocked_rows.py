"""how many of the states the search evaluates have ALL three moves blocked (their Q is overwritten by -1 whatever the net says,
alpha_nnet.py:63-76): blocked_rows.py [games 1024] [turns 12]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "alphasnake-zero_amd")]
import numpy as np, torch
from snake_engine import net
from utils.agent import Agent
from utils.alpha_nnet import AlphaNNet
from utils.mp_game_runner import MPGameRunner
games = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
turns = int(sys.argv[2]) if len(sys.argv) > 2 else 12
MPGameRunner.verbose = False; MPGameRunner.init = "device"
nn_ = AlphaNNet(input_shape=(21, 21, 3), _weights=net.glorot_uniform_weights((21, 21, 3), 4, seed=0))
alice = Agent(nn_, 2, True, 8, 50, seed=1)
gr = MPGameRunner(11, 11, 4, 1, games, seed=2)
cnt = torch.zeros(4, dtype=torch.int64, device="cuda")
orig = nn_.v_device_unguarded
def spy(planes, mask):
    b = mask.sum(dim=1)
    cnt.add_(torch.bincount(b.long(), minlength=4)[:4])
    return orig(planes, mask)
nn_.v_device_unguarded = spy
for t in range(turns):
    gr.run(alice, max_turns=1)
    c = cnt.cpu().numpy()
    print(f"turn {t + 1}: evaluated rows by number of blocked moves 0/1/2/3: {c.tolist()}  all-blocked share {c[3] / max(1, c.sum()):.4f}", flush=True)
