"""the RCCL calls of the multi-GPU path on a one-GPU box: a one-rank "nccl" process group (two ranks on one GPU are refused by
RCCL: "Duplicate GPU detected"), the same collectives with the same dtypes, shapes and devices bench.py and the trainer use"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "alphasnake-zero_amd")]
import torch
import torch.distributed as dist
from snake_engine import dist as sdist

torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
X = torch.rand((5120, 21, 21, 3), device="cuda")
V = torch.rand((5120, 3), device="cuda")
Xg, Vg = sdist.all_gather_samples(X, V, single_rank_collective=True)            # ncclAllGather x 2
assert torch.equal(Xg, X) and torch.equal(Vg, V) and Xg.data_ptr() != X.data_ptr()
avg, games = sdist.all_reduce_counters([4.0, 8.0, 12.0, 16.0, 20.0, 400.0], 4, "cuda", single_rank_collective=True)   # ncclAllReduce f64
assert games == 4 and avg == [1.0, 2.0, 3.0, 4.0, 5.0, 100.0]
t = torch.tensor([1.5, 7.0], dtype=torch.float64, device="cuda")
dist.all_reduce(t, op=dist.ReduceOp.MAX)                                          # bench.py's max-over-ranks time
flat = torch.rand(1_244_807 + 1, device="cuda")                                   # the trainer's one gradient bucket
ref = flat.clone()
dist.all_reduce(flat)
sums = torch.rand(256, dtype=torch.float64, device="cuda")                        # the training step's batch-norm sums (float64)
sums_ref = sums.clone()
dist.all_reduce(sums)
n_rec = torch.tensor([4711], dtype=torch.int64, device="cuda")                    # record counts of all ranks -> the sample plan
dist.all_reduce(n_rec)
counts, seed = sdist.gather_counts(4711, single_rank_collective=True)             # ncclAllGather int64 [records, seed]
assert counts == [4711] and sdist.share_counts(counts, 4096, seed) == [4096]
mine = torch.arange(8, dtype=torch.float64, device="cuda")                        # bench.py's per-rank row
rows = torch.empty(8, dtype=torch.float64, device="cuda")
dist.all_gather_into_tensor(rows, mine)
assert torch.equal(sums, sums_ref) and int(n_rec) == 4711 and torch.equal(rows, mine)
seed = torch.tensor([123], dtype=torch.int64, device="cuda")
dist.broadcast(seed, 0)
dist.barrier()
torch.cuda.synchronize()
assert torch.equal(flat, ref) and float(t[1]) == 7.0 and int(seed) == 123
dist.destroy_process_group()
print("rccl ok")
