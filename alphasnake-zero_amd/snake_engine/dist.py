"""Multi-GPU plumbing: games shard across ranks with no communication during self-play; at iteration
end the sampled training rows are all-gathered and the six log counters all-reduced (SURVEY.md 8e;
the reference has no collective at all -- trainer.py:56-75 is the single-process form of this step).
One process per GPU; backend "nccl" (= RCCL over xGMI) on GPUs, "gloo" in the CPU tests."""
import numpy as np
import torch
import torch.distributed as dist


def shard_range(total, rank, world):
    """contiguous block of game ids owned by `rank` (sizes differ by at most one)"""
    base, extra = divmod(total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


BATCH_ROWS = 2048          # trainer.py:63
MAX_BATCHES = 5            # trainer.py:65


def sample_plan(n_records_all_ranks, world=1):
    """(rows to sample in total, fit batch size, rows per rank) from the number of records of ALL ranks, as
    trainer.py:63-72 derives them from len(Alice.records): at most 5 batches of 2 048 rows, whole batches only.
    With fewer than one batch of records the reference draws nothing and fails in `flip` (its `samples > len(records)`
    branch cannot be taken); here all records form one batch, which is what that branch says."""
    batches = min(MAX_BATCHES, n_records_all_ranks // BATCH_ROWS)
    if batches:
        wanted, batch_size = BATCH_ROWS * batches, BATCH_ROWS
    else:
        wanted = batch_size = (n_records_all_ranks // world) * world
    return wanted, batch_size, wanted // world


def sample_share(n_local_records, total_samples, world, rng):
    """indices of this rank's share of the `total_samples` training rows (trainer.py:63-74 draws
    sample(range(len(records)), samples) from one pool; here every rank draws total/world from its own)"""
    share = total_samples // world
    if n_local_records == 0:
        return np.zeros(0, np.int64), share
    if n_local_records >= share:
        return rng.choice(n_local_records, size=share, replace=False), share
    return np.resize(rng.permutation(n_local_records), share), share


def all_gather_samples(X, V, group=None, single_rank_collective=False):
    """X [k, h, w, 3] float32, V [k, 3] float32 with the same k on every rank -> concatenation over ranks.
    With one rank there is nothing to exchange; `single_rank_collective` runs the collectives anyway (the RCCL calls
    themselves can then be exercised on a one-GPU box, tests/test_bench_gpu.py)."""
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size(group) == 1 and not single_rank_collective):
        return X, V
    world = dist.get_world_size(group)
    dev = X.device
    if dist.get_backend(group) == "gloo" and X.is_cuda:      # rehearsal backend: stage through host memory
        X, V = X.cpu(), V.cpu()
    Xo = torch.empty((world * X.shape[0],) + tuple(X.shape[1:]), dtype=X.dtype, device=X.device)
    Vo = torch.empty((world * V.shape[0],) + tuple(V.shape[1:]), dtype=V.dtype, device=V.device)
    dist.all_gather_into_tensor(Xo, X.contiguous(), group=group)
    dist.all_gather_into_tensor(Vo, V.contiguous(), group=group)
    return Xo.to(dev), Vo.to(dev)


def all_reduce_counters(counters, game_cnt, device, group=None, single_rank_collective=False):
    """sum of the six per-rank counter totals and of the game counts -> per-game averages (mp_game_runner.py:71-76)"""
    t = torch.tensor(list(counters) + [game_cnt], dtype=torch.float64, device=device)
    active = dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or single_rank_collective)
    if active and dist.get_backend(group) == "gloo":
        t = t.cpu()
    if active:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    t = t.cpu().numpy()
    return (t[:6] / t[6]).tolist(), int(t[6])
