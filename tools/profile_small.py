"""cProfile of the host side of a small self-play run (development aid)"""
import cProfile, pstats, os, sys, io
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "alphasnake-zero_amd")]
import numpy as np, torch
from snake_engine import net
from utils.agent import Agent
from utils.alpha_nnet import AlphaNNet
from utils.mp_game_runner import MPGameRunner
MPGameRunner.verbose = False
MPGameRunner.init = "device"
nnet = AlphaNNet(input_shape=(21, 21, 3), _weights=net.glorot_uniform_weights((21, 21, 3), 4, 0))
alice = Agent(nnet, 2, True, 8, 25, seed=1)
gr = MPGameRunner(11, 11, 4, 1, 8, seed=2)
gr.run(alice, max_turns=2)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
gr.run(alice, max_turns=10)
torch.cuda.synchronize()
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(35); print(s.getvalue()[:6000])
