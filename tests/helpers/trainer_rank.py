"""one rank of the two-rank trainer test (tests/test_next_rows_gpu.py): torch.distributed over gloo, both ranks on GPU 0"""
import os
import random
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "alphasnake-zero_amd")]
import numpy as np
import torch
import torch.distributed as dist

out_dir = sys.argv[1]
rank = int(os.environ["RANK"])
torch.cuda.set_device(0)
dist.init_process_group("gloo")
from utils.alpha_nnet import AlphaNNet
from utils.alpha_snake_zero_trainer import AlphaSnakeZeroTrainer
from utils.mp_game_runner import MPGameRunner
from snake_engine.net import glorot_uniform_weights

random.seed(10 + rank); np.random.seed(10 + rank)          # different games per rank, same start weights
os.chdir(out_dir)
MPGameRunner.verbose = False
nnet = AlphaNNet(input_shape=(21, 21, 3), _weights=glorot_uniform_weights((21, 21, 3), 4, seed=0))
trainer = AlphaSnakeZeroTrainer(10, 4, 8, 1e-3, 0.98, 11, 11, 4, None)
collect = trainer._collect
seen = {}


def counting_collect(alice):
    seen["records"] = len(alice.records)
    X, V, batch = collect(alice)
    seen["rows"], seen["batch"] = len(X), batch
    return X, V, batch
trainer._collect = counting_collect
last = trainer.train(nnet, name="dp", iteration=0, max_iterations=1)
import json
json.dump(seen, open(os.path.join(out_dir, f"collect_r{rank}.json"), "w"))
np.savez(os.path.join(out_dir, f"weights_r{rank}.npz"), *last.v_net.get_weights())
dist.barrier()
dist.destroy_process_group()
