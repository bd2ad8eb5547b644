"""Probe for sizing bench.py's cpu_baseline: usable host cpus on the box and the C MCTS baseline's rate at several
thread counts (oracle/mcts_cpu.c + PyTorch-CPU net).  Development tool, not part of the product."""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "alphasnake-zero_amd")]
import numpy as np
import torch


def usable():
    out = {"cpu_count": os.cpu_count(), "affinity": len(os.sched_getaffinity(0)), "torch_threads": torch.get_num_threads()}
    for p in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
        try:
            out[p] = open(p).read().strip()
        except OSError:
            pass
    return out


if __name__ == "__main__":
    print(usable(), flush=True)
    from oracle.mcts_cpu import CpuSelfPlay, seeded_games
    from oracle import net_ref
    from snake_engine import net
    ws = net.glorot_uniform_weights((21, 21, 3), 4, 0)
    X = np.random.RandomState(0).rand(2048, 21, 21, 3).astype(np.float32)
    for thr in [int(a) for a in sys.argv[1:]] or [16, 32, 64, 128]:
        torch.set_num_threads(thr)
        net_ref.forward(ws, X[:64])
        t = time.time(); net_ref.forward(ws, X); dt = time.time() - t
        print(f"net {thr} threads: {2048 / dt:.0f} states/s = {2048 * 1.0437 / dt:.0f} GFLOP/s", flush=True)
    for thr in (1, 16, 64):
        sp = CpuSelfPlay(seeded_games(256, seed=1), net=None, threads=thr, max_breadth=50, seed=1)
        t = time.time(); st = sp.run(max_turns=2); dt = time.time() - t
        print(f"stub net, {thr} threads: {st['env_steps'] / dt:.1f} env-steps/s ({dt:.2f} s)", flush=True)
        sp.close()
