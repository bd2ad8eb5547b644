"""CPU tests (gloo, world_size 2) of the multi-GPU plumbing: games shard with no overlap, the iteration-end
exchange all-gathers every rank's share of the sampled rows and all-reduces the log counters."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [repo, os.path.join(repo, "alphasnake-zero_amd")]
    from snake_engine import dist as sdist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        total_games = 11
        lo, hi = sdist.shard_range(total_games, rank, world)
        # each rank "self-plays" its shard: records are tagged with (rank, index)
        n_local = 37 + 5 * rank
        rng = np.random.RandomState(rank)
        idx, share = sdist.sample_share(n_local, 64, world, rng)
        assert share == 32 and len(idx) == 32 and len(set(idx.tolist())) == 32 and idx.max() < n_local
        X = torch.zeros((share, 3, 3, 3))
        X[:, 0, 0, 0] = rank
        X[:, 0, 0, 1] = torch.as_tensor(idx, dtype=torch.float32)
        V = torch.full((share, 3), float(rank))
        Xg, Vg = sdist.all_gather_samples(X, V)
        counters = [1.0 * (rank + 1)] * 6
        avg, games = sdist.all_reduce_counters(counters, hi - lo, "cpu")
        torch.save(dict(lo=lo, hi=hi, Xg=Xg, Vg=Vg, avg=avg, games=games, idx=idx), os.path.join(out_dir, f"r{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_two_rank_exchange(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    res = [torch.load(tmp_path / f"r{r}.pt", weights_only=False) for r in range(world)]
    assert (res[0]["lo"], res[0]["hi"], res[1]["lo"], res[1]["hi"]) == (0, 6, 6, 11)
    for r in range(world):
        Xg, Vg = res[r]["Xg"], res[r]["Vg"]
        assert Xg.shape == (64, 3, 3, 3) and Vg.shape == (64, 3)
        for src in range(world):                      # rank-major concatenation, every rank sees the same rows
            blk = Xg[src * 32:(src + 1) * 32]
            assert (blk[:, 0, 0, 0] == src).all()
            assert blk[:, 0, 0, 1].tolist() == [float(i) for i in res[src]["idx"]]
            assert (Vg[src * 32:(src + 1) * 32] == src).all()
        assert res[r]["games"] == 11 and np.allclose(res[r]["avg"], [3.0 / 11] * 6)
    assert torch.equal(res[0]["Xg"], res[1]["Xg"])


def _collect_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [repo, os.path.join(repo, "alphasnake-zero_amd")]
    from utils.alpha_snake_zero_trainer import AlphaSnakeZeroTrainer
    import datetime
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))
    try:
        class Alice:                       # rank 1 played games that recorded nothing
            records = [0] * (40 if rank == 0 else 0)
            values = records
        tr = AlphaSnakeZeroTrainer(8, 8, 8, 1e-3, 0.98)
        try:
            tr._collect(Alice())
            what = "returned"
        except RuntimeError as e:
            what = str(e)
        open(os.path.join(out_dir, f"c{rank}.txt"), "w").write(what)
        # fit: a trailing batch smaller than the world is refused by EVERY rank (same permutation on all of them)
        from utils import trainer_torch
        from snake_engine.net import glorot_uniform_weights
        ws = glorot_uniform_weights((5, 5, 3), blocks=1, seed=0)
        X = np.zeros((5, 5, 5, 3), np.float32); Y = np.zeros((5, 3), np.float32)
        try:
            trainer_torch.fit(ws, (5, 5, 3), X, Y, epochs=1, batch_size=4, lr_schedule=None, device="cpu", seed=0, verbose=False)
            what = "returned"
        except RuntimeError as e:
            what = str(e)
        open(os.path.join(out_dir, f"f{rank}.txt"), "w").write(what)
    finally:
        dist.destroy_process_group()


def test_a_rank_without_records_and_an_unsplittable_batch_fail_on_every_rank(tmp_path):
    """a rank that cannot contribute must not raise alone while its peers wait inside a collective: both ranks raise the same
    error, decided from the all-reduced counts (iteration-end sampling) or from what every rank sees (the fit's trailing batch)"""
    world, port = 2, _free_port()
    mp.spawn(_collect_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert "1 of 2 ranks recorded no state" in open(tmp_path / f"c{r}.txt").read()
        assert "trailing batch of 1 rows cannot be split over 2 ranks" in open(tmp_path / f"f{r}.txt").read()


def test_shard_range_partitions_everything():
    from snake_engine.dist import shard_range, sample_share
    for total in (1, 7, 8, 4096, 262144):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1
    idx, share = sample_share(5, 64, 2, np.random.RandomState(0))     # fewer local records than the share: wraps
    assert share == 32 and len(idx) == 32 and set(idx.tolist()) == set(range(5))
    idx, share = sample_share(0, 64, 2, np.random.RandomState(0))
    assert share == 32 and len(idx) == 0


def test_bench_launcher_reports_a_failed_rank_and_does_not_hang():
    """`python bench.py --gpus 2` on a box without a GPU: the parent starts two rank processes without touching the GPU
    itself; they fail ("needs an MI355X"), the parent kills what is left, prints the exit codes and returns non-zero"""
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: tests/test_bench_gpu.py covers the working path")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--games", "8"], capture_output=True,
                         text=True, env=env, timeout=300)
    assert out.returncode != 0
    assert "rank exit codes" in out.stderr and "needs an MI355X" in out.stderr
    assert out.stdout.strip() == ""                      # no JSON line from a failed run
    # and a WORLD_SIZE that contradicts --gpus is refused before anything else happens
    out = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2"], capture_output=True, text=True,
                         env=dict(env, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0"), timeout=120)
    assert out.returncode != 0 and "WORLD_SIZE" in out.stderr
