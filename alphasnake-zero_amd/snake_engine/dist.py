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


# the seed the ranks share for share_counts comes from a stream of this module's own: a run over N ranks must leave the caller's
# global NumPy / random state exactly where a one-rank run leaves it (the reference draws nothing here, trainer.py:63-75)
_seed_stream = np.random.Generator(np.random.PCG64())

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


def gather_counts(n_local_records, group=None, single_rank_collective=False):
    """(every rank's record count, a seed all ranks share): one small all-gather.  The counts decide how many rows each rank
    contributes (share_counts); the seed is rank 0's draw, so that every rank computes the same split."""
    seed = int(_seed_stream.integers(1 << 31))
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size(group) == 1 and not single_rank_collective):
        return [int(n_local_records)], seed
    world = dist.get_world_size(group)
    dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
    mine = torch.tensor([int(n_local_records), seed], dtype=torch.int64, device=dev)
    out = torch.empty((world * 2,), dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(out, mine, group=group)
    out = out.cpu().numpy().reshape(world, 2)
    return [int(v) for v in out[:, 0]], int(out[0, 1])


def share_counts(counts, total_samples, seed):
    """rows every rank contributes to the `total_samples` training rows, the same list on every rank.  trainer.py:71 draws
    sample(range(len(records)), samples) from ONE pool, without replacement; here every rank draws from its own records:
    total // world each wherever a rank holds that many.  A rank that holds fewer gives all it has, and the shortfall is drawn
    from what the other ranks have left, as one sample without replacement from the union of those leftovers (the per-rank
    counts of such a sample are multivariate hypergeometric) -- no row is ever used twice.  (Until round 4 a short rank repeated
    its rows.)"""
    world = len(counts)
    counts = np.asarray(counts, np.int64)
    assert total_samples <= int(counts.sum()), (total_samples, counts)
    share = total_samples // world
    k = np.minimum(counts, share)
    short = int(total_samples - k.sum())
    if short:
        left = counts - k
        k = k + np.random.Generator(np.random.PCG64(seed)).multivariate_hypergeometric(left, short)
    return [int(v) for v in k]


def sample_share(n_local_records, rows, rng):
    """indices of `rows` of this rank's records, without replacement"""
    assert rows <= n_local_records, (rows, n_local_records)
    return rng.choice(n_local_records, size=rows, replace=False) if rows else np.zeros(0, np.int64)


def all_gather_samples(X, V, rows_per_rank=None, group=None, single_rank_collective=False):
    """X [k, h, w, 3] float32, V [k, 3] float32 -> concatenation over ranks.  `rows_per_rank` (share_counts): the ranks' k when
    they differ -- the exchange then moves max(k) rows per rank (short ranks pad) and the padding is dropped on arrival.
    With one rank there is nothing to exchange; `single_rank_collective` runs the collectives anyway (the RCCL calls
    themselves can then be exercised on a one-GPU box, tests/test_bench_gpu.py)."""
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size(group) == 1 and not single_rank_collective):
        return X, V
    world = dist.get_world_size(group)
    dev = X.device
    if dist.get_backend(group) == "gloo" and X.is_cuda:      # rehearsal backend: stage through host memory
        X, V = X.cpu(), V.cpu()
    kmax = X.shape[0] if rows_per_rank is None else max(rows_per_rank)
    if X.shape[0] < kmax:
        X = torch.cat([X, X.new_zeros((kmax - X.shape[0],) + tuple(X.shape[1:]))])
        V = torch.cat([V, V.new_zeros((kmax - V.shape[0],) + tuple(V.shape[1:]))])
    Xo = torch.empty((world * kmax,) + tuple(X.shape[1:]), dtype=X.dtype, device=X.device)
    Vo = torch.empty((world * kmax,) + tuple(V.shape[1:]), dtype=V.dtype, device=V.device)
    dist.all_gather_into_tensor(Xo, X.contiguous(), group=group)
    dist.all_gather_into_tensor(Vo, V.contiguous(), group=group)
    if rows_per_rank is not None and min(rows_per_rank) < kmax:
        keep = torch.cat([torch.arange(r * kmax, r * kmax + k) for r, k in enumerate(rows_per_rank)]).to(Xo.device)
        Xo, Vo = Xo[keep], Vo[keep]
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
