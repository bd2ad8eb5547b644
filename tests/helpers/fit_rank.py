"""one rank of the data-parallel fit test (tests/test_train_ops_gpu.py): utils.trainer_torch.fit on the kernels under
torch.distributed over gloo, the ranks sharing GPU 0; world size 1 runs the same call without a process group"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "alphasnake-zero_amd")]
import numpy as np
import torch

out = sys.argv[1]
world = int(os.environ.get("WORLD_SIZE", "1"))
rank = int(os.environ.get("RANK", "0"))
torch.cuda.set_device(0)
if world > 1:
    import torch.distributed as dist
    dist.init_process_group("gloo")
from snake_engine import net
from utils import trainer_torch

rs = np.random.RandomState(3)
X = rs.rand(201, 21, 21, 3).astype(np.float32)          # three batches of 67 rows: two ranks get 34 and 33 of each
Y = np.tanh(rs.randn(201, 3)).astype(np.float32)
ws = net.glorot_uniform_weights((21, 21, 3), blocks=2, seed=5)
got = trainer_torch.fit(ws, (21, 21, 3), X, Y, 2, 67, ([3, 5], [1e-3, 2.5e-4, 0.0]), seed=11, verbose=False)
assert trainer_torch.fit.last_mode == "kernels"
q = trainer_torch._Net(got, torch.device("cuda")).forward(torch.as_tensor(X[:96], device="cuda"), False).detach().cpu().numpy()
np.savez(os.path.join(out, f"fit_w{world}_r{rank}.npz"), *got, hist=np.array(trainer_torch.fit.last_history), q=q)
if world > 1:
    dist.barrier()
    dist.destroy_process_group()
