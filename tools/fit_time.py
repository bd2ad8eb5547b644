"""time of one generation's fit: 5 x 2 048 sampled rows + their mirror images = 20 480 rows,
32 epochs, batch 2 048 (trainer.py:63-83, alpha_nnet.py:58-59) on the 4-block 11x11 net: 320 optimizer steps, the last 220
at learning rate 0 (alpha_nnet.py:79-84).  Development tool: fit_time.py [epochs]
  SNK_TRAIN_CONV=torch        every operator from PyTorch / MIOpen (the A/B arm)
  SNK_TRAIN_DEAD_STEPS=full   run the backward passes of the rate-0 steps too"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "alphasnake-zero_amd")]
import numpy as np, torch
from snake_engine import net
from utils import trainer_torch
epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 32
rs = np.random.RandomState(0)
X = rs.rand(20480, 21, 21, 3).astype(np.float32)
Y = np.tanh(rs.randn(20480, 3)).astype(np.float32)
ws = net.glorot_uniform_weights((21, 21, 3), blocks=4, seed=0)
sched = ([20, 40, 60, 80, 100], [1e-4, 2.5e-5, 6.25e-6, 1.5625e-6, 3.90625e-7, 0.0])
trainer_torch.fit(ws, (21, 21, 3), X[:4096], Y[:4096], 1, 2048, sched, verbose=False)      # warm-up


def timed(ep):
    torch.cuda.synchronize()
    t0 = time.time()
    trainer_torch.fit(ws, (21, 21, 3), X, Y, ep, 2048, sched, verbose=False)
    torch.cuda.synchronize()
    return time.time() - t0
live = timed(min(10, epochs))          # steps 0 .. 99: every one at a non-zero rate
dt = timed(epochs)
steps = epochs * 10
n_live = min(steps, 100)
fl = 3 * 2 * 521.86e6 * 2048          # forward + input gradient + weight gradient, per live step
line = (f"mode {trainer_torch.fit.last_mode}: {steps} optimizer steps of 2048 rows in {dt:.2f} s; live steps {live / n_live * 1e3:.1f} ms each "
        f"(~{fl * n_live / live / 1e12:.1f} TFLOP/s counting 3 x the forward flops)")
if steps > 100:
    line += f", rate-0 steps {(dt - live) / (steps - 100) * 1e3:.1f} ms each"
print(line)
