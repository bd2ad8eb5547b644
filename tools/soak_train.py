"""soak: the generation loop at the reference's settings (train.py:6-13: 256 games, depth 8, breadth 128, 11x11, 4 snakes) for a
few generations on one MI355X -- self-play, log.csv, sampling, fit on the library's kernels, copy_and_compile, .h5 -- with the
time of each part (development aid): soak_train.py [generations] [games]"""
import os, sys, tempfile, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "alphasnake-zero_amd")]
import numpy as np, random
import torch
random.seed(0); np.random.seed(0)
gens = int(sys.argv[1]) if len(sys.argv) > 1 else 3
games = int(sys.argv[2]) if len(sys.argv) > 2 else 256
import train
from utils import trainer_torch
from utils.alpha_snake_zero_trainer import AlphaSnakeZeroTrainer
from utils.mp_game_runner import MPGameRunner
MPGameRunner.verbose = False
marks = []
sp, co = AlphaSnakeZeroTrainer._self_play, AlphaSnakeZeroTrainer._collect


def timed_self_play(self, nnet, iteration):
    torch.cuda.synchronize(); t0 = time.time()
    alice, runner = sp(self, nnet, iteration)
    torch.cuda.synchronize()
    marks.append(("self-play", iteration, time.time() - t0, runner.env_steps, len(alice.records)))
    return alice, runner


def timed_collect(self, alice):
    t0 = time.time()
    out = co(self, alice)
    marks.append(("collect", len(out[0]), time.time() - t0))
    return out
AlphaSnakeZeroTrainer._self_play, AlphaSnakeZeroTrainer._collect = timed_self_play, timed_collect
fit = trainer_torch.fit


def timed_fit(*a, **k):
    torch.cuda.synchronize(); t0 = time.time()
    k["verbose"] = False
    out = fit(*a, **k)
    torch.cuda.synchronize()
    me = trainer_torch.fit                    # fit() leaves last_mode / last_history on the module's `fit`, which is this wrapper now
    marks.append(("fit", me.last_mode, len(a[2]), time.time() - t0, me.last_history[0], me.last_history[-1]))
    return out
trainer_torch.fit = timed_fit
d = tempfile.mkdtemp()
os.chdir(d)
t0 = time.time()
train.start("soak", 0, max_iterations=gens, self_play_games=games)
print(f"{gens} generations of {games} games in {time.time() - t0:.1f} s")
for m in marks:
    print(m)
print(open("log.csv").read())
print(sorted(os.listdir("models")))
