"""soak: one complete self-play iteration with the real net, the way the trainer drives it (development aid)"""
import os, sys, time, random
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "alphasnake-zero_amd")]
import numpy as np, torch
from utils.agent import Agent
from utils.alpha_nnet import AlphaNNet
from utils.mp_game_runner import MPGameRunner
games, breadth = int(sys.argv[1]), int(sys.argv[2])
health_dec = int(sys.argv[3]) if len(sys.argv) > 3 else 9
random.seed(0); np.random.seed(0)
MPGameRunner.verbose = False
nnet = AlphaNNet(input_shape=(21, 21, 3)).copy_and_compile()
alice = Agent(nnet, 2, True, 8, breadth, seed=1, tt_capacity=int(sys.argv[4]) if len(sys.argv) > 4 else None)
gr = MPGameRunner(11, 11, 4, health_dec, games, seed=2)
t0 = time.time()
rewards = gr.run(alice)
dt = time.time() - t0
print(f"{games} games to completion in {dt:.1f} s, {gr.turns} root turns, env-steps {gr.env_steps} ({gr.env_steps/dt:.1f}/s)")
print("counters", [getattr(gr, k) for k in ("wall_collision", "body_collision", "head_collision", "starvation", "food_eaten", "game_length")])
print("records", len(alice.records), "cache", len(alice.cached_values), "tt", alice._mcts.tt.status(), "evals", alice._mcts.stats)
idx = random.sample(range(len(alice.records)), min(len(alice.records), 2048))
X = [alice.records[i] for i in idx[:64]]; V = [alice.values[i] for i in idx[:64]]
print("sample", np.array(X).shape, np.array(V).shape, float(np.abs(np.array(V)).max()))
assert all(r is not None for r in rewards)
