"""|dQ| of the default (split-f16) net against the float32 CPU restatement and the kernel's range headroom on weights that went
through a few real generations of the trainer loop (self-play -> sample -> mirror -> fit), i.e. with moving batch-norm
statistics that only partly follow the data.  Development tool: trained_parity.py [generations] [games]"""
import os, sys, random, tempfile
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "alphasnake-zero_amd")]
import numpy as np, torch
from oracle import net_ref
from utils.alpha_nnet import AlphaNNet
from utils.alpha_snake_zero_trainer import AlphaSnakeZeroTrainer
from utils.mp_game_runner import MPGameRunner
from utils.agent import Agent
gens = int(sys.argv[1]) if len(sys.argv) > 1 else 3
games = int(sys.argv[2]) if len(sys.argv) > 2 else 96
random.seed(0); np.random.seed(0)
MPGameRunner.verbose = False
os.chdir(tempfile.mkdtemp()); os.mkdir("models")
net0 = AlphaNNet(input_shape=(21, 21, 3))
trainer = AlphaSnakeZeroTrainer(games, 4, 16, 1e-3, 0.98)
import io, contextlib
with contextlib.redirect_stdout(io.StringIO()):
    net = trainer.train(net0, "tp", 0, max_iterations=gens)
ws = net.v_net.get_weights()
alice = Agent(net, 2, True, 4, 8, seed=1)
gr = MPGameRunner(11, 11, 4, 3, 64, seed=2)
gr.run(alice, max_turns=6)
X = alice.records.fetch(range(min(512, len(alice.records))))
ref = net_ref.forward(ws, X)
got = net.v(list(X))
rep = net._qnet.activation_report(torch.as_tensor(X, device="cuda"))
moved = max(float(np.abs(a - b).max()) for a, b in zip(ws, net0.v_net.get_weights()))
print(f"{gens} generations x {games} games: weights moved by up to {moved:.3f}; {len(X)} fresh observations: max |dQ| = {np.abs(got - ref).max():.2e} "
      f"(max |Q| {np.abs(ref[ref > -1]).max():.3f}); smallest headroom to the f16 limit over the 8 tower layers: {min(h for _, _, h in rep):.0f}x; "
      f"range flags {net._qnet.range_flags()}")
print(open("log.csv").read())
