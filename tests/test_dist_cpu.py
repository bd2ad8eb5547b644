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
        counts, seed = sdist.gather_counts(n_local)
        assert counts == [37, 42]
        rows = sdist.share_counts(counts, 64, seed)
        assert rows == [32, 32]
        share = rows[rank]
        idx = sdist.sample_share(n_local, share, rng)
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


def _uneven_worker(rank, world, port, out_dir):
    """rank 0 recorded 40 states, rank 1 recorded 4 000: the plan wants 2 048 rows (one batch of ALL records)"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [repo, os.path.join(repo, "alphasnake-zero_amd")]
    from snake_engine import dist as sdist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n_local = 40 if rank == 0 else 4000
        counts, seed = sdist.gather_counts(n_local)
        wanted, batch, _ = sdist.sample_plan(sum(counts), world)
        rows = sdist.share_counts(counts, wanted, seed)
        idx = sdist.sample_share(n_local, rows[rank], np.random.RandomState(100 + rank))
        X = torch.zeros((len(idx), 3, 3, 3))
        X[:, 0, 0, 0] = rank
        X[:, 0, 0, 1] = torch.as_tensor(idx, dtype=torch.float32)
        V = torch.full((len(idx), 3), float(rank))
        Xg, Vg = sdist.all_gather_samples(X, V, rows)
        torch.save(dict(counts=counts, wanted=wanted, batch=batch, rows=rows, Xg=Xg, Vg=Vg), os.path.join(out_dir, f"u{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_a_rank_short_of_its_share_is_topped_up_without_repeating_a_row(tmp_path):
    """trainer.py:71 samples WITHOUT replacement from all records.  With very uneven shards (40 / 4 000 records, 2 048 rows wanted)
    the short rank gives all it has and the other rank makes up the difference: 2 048 distinct rows arrive on every rank, none
    twice (until round 4 the short rank repeated its 40 rows 25 times)"""
    world, port = 2, _free_port()
    mp.spawn(_uneven_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    res = [torch.load(tmp_path / f"u{r}.pt", weights_only=False) for r in range(world)]
    for r in res:
        assert r["counts"] == [40, 4000] and (r["wanted"], r["batch"]) == (2048, 2048) and r["rows"] == [40, 2008]
        Xg, Vg = r["Xg"], r["Vg"]
        assert Xg.shape == (2048, 3, 3, 3) and Vg.shape == (2048, 3)
        tags = {(int(a), int(b)) for a, b in zip(Xg[:, 0, 0, 0].tolist(), Xg[:, 0, 0, 1].tolist())}
        assert len(tags) == 2048, "a row was used twice"
        assert sum(1 for a, _ in tags if a == 0) == 40 and (Vg[:40] == 0).all() and (Vg[40:] == 1).all()
    assert torch.equal(res[0]["Xg"], res[1]["Xg"])


def test_share_counts_split():
    """equal shares wherever every rank can afford them; otherwise the shortfall comes from the others' leftovers, the same split on
    every rank (same seed), never more than a rank holds"""
    from snake_engine.dist import share_counts
    assert share_counts([5000, 5000, 5000, 5000], 8192, 1) == [2048] * 4
    for seed in range(20):
        k = share_counts([10, 3000, 50, 9000], 8192, seed)
        assert sum(k) == 8192 and k[0] == 10 and k[2] == 50 and 2048 <= k[1] <= 3000 and 2048 <= k[3] <= 9000
        assert k == share_counts([10, 3000, 50, 9000], 8192, seed)
    assert share_counts([1, 1, 1, 8189], 8192, 7) == [1, 1, 1, 8189]
    ks = {tuple(share_counts([10, 3000, 50, 9000], 8192, s)) for s in range(20)}
    assert len(ks) > 1, "the top-up is a draw, not a fixed rule"


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
        # fit: a trailing batch smaller than the world (5 rows, batch 4, 2 ranks: one row) is trained on like any other batch
        # (Keras' fit does): the rank whose slice of it is empty joins the step's all-reduces with sums over no rows
        from utils import trainer_torch
        from snake_engine.net import glorot_uniform_weights
        ws = glorot_uniform_weights((5, 5, 3), blocks=1, seed=0)
        rs = np.random.RandomState(3)
        X = rs.rand(5, 5, 5, 3).astype(np.float32); Y = np.tanh(rs.randn(5, 3)).astype(np.float32)
        out = trainer_torch.fit(ws, (5, 5, 3), X, Y, epochs=2, batch_size=4, lr_schedule=([100], [1e-3, 0.0]), device="cpu", seed=0,
                                verbose=False, dtype=torch.float64)
        np.savez(os.path.join(out_dir, f"f{rank}.npz"), *out)
    finally:
        dist.destroy_process_group()


def test_a_rank_without_records_fails_on_every_rank_and_a_short_trailing_batch_is_trained_on(tmp_path):
    """a rank that cannot contribute must not raise alone while its peers wait inside a collective: both ranks raise the same
    error, decided from the gathered counts (iteration-end sampling).  The fit's trailing batch of fewer rows than ranks is an
    ordinary optimizer step (Keras trains on it): both ranks end with the same weights, and those equal the one-process fit"""
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [p for p in (repo, os.path.join(repo, "alphasnake-zero_amd")) if p not in sys.path]
    world, port = 2, _free_port()
    mp.spawn(_collect_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert "1 of 2 ranks recorded no state" in open(tmp_path / f"c{r}.txt").read()
    got = [[z[k] for k in z.files] for z in (np.load(tmp_path / f"f{r}.npz") for r in range(world))]
    from utils import trainer_torch
    from snake_engine.net import glorot_uniform_weights
    ws = glorot_uniform_weights((5, 5, 3), blocks=1, seed=0)
    rs = np.random.RandomState(3)
    X = rs.rand(5, 5, 5, 3).astype(np.float32); Y = np.tanh(rs.randn(5, 3)).astype(np.float32)
    one = trainer_torch.fit(ws, (5, 5, 3), X, Y, epochs=2, batch_size=4, lr_schedule=([100], [1e-3, 0.0]), device="cpu", seed=0,
                            verbose=False, dtype=torch.float64)
    assert any(np.abs(a - b).max() > 1e-6 for a, b in zip(one, ws)), "the fit moved nothing"
    for a, b, c in zip(got[0], got[1], one):
        assert np.array_equal(a, b) and np.abs(a - c).max() <= 1e-9 * max(1.0, np.abs(c).max())


EIGHT_COUNTS = [3000, 10, 2500, 700, 5000, 1, 2048, 4000]          # records per rank: three ranks hold less than their share of 1 280


def _eight_worker(rank, world, port, out_dir):
    """the whole iteration-end exchange and a data-parallel fit at the world size the headline is quoted on (8), over gloo"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import sys
    import datetime
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [repo, os.path.join(repo, "alphasnake-zero_amd")]
    torch.set_num_threads(1)
    from snake_engine import dist as sdist
    from utils.alpha_snake_zero_trainer import AlphaSnakeZeroTrainer
    from utils import trainer_torch
    from snake_engine.net import glorot_uniform_weights
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    try:
        np.random.seed(77)
        state0 = np.random.get_state()
        n_local = EIGHT_COUNTS[rank]
        counts, seed = sdist.gather_counts(n_local)
        wanted, batch, _ = sdist.sample_plan(sum(counts), world)
        rows = sdist.share_counts(counts, wanted, seed)
        idx = sdist.sample_share(n_local, rows[rank], np.random.RandomState(100 + rank))
        X = torch.zeros((len(idx), 3, 3, 3))
        X[:, 0, 0, 0] = rank
        X[:, 0, 0, 1] = torch.as_tensor(idx, dtype=torch.float32)
        V = torch.full((len(idx), 3), float(rank))
        Xg, Vg = sdist.all_gather_samples(X, V, rows)
        lo, hi = sdist.shard_range(262144, rank, world)
        avg, games = sdist.all_reduce_counters([float(rank + 1)] * 6, hi - lo, "cpu")
        state1 = np.random.get_state()
        untouched = state0[0] == state1[0] and np.array_equal(state0[1], state1[1]) and state0[2:] == state1[2:]
        torch.save(dict(counts=counts, seed=seed, wanted=wanted, batch=batch, rows=rows, Xg=Xg, Vg=Vg, avg=avg, games=games,
                        untouched=untouched), os.path.join(out_dir, f"e{rank}.pt"))

        class Alice:                       # rank 5 played games that recorded nothing
            records = [0] * (0 if rank == 5 else 40 + rank)
            values = records
        try:
            AlphaSnakeZeroTrainer(8, 8, 8, 1e-3, 0.98)._collect(Alice())
            what = "returned"
        except RuntimeError as e:
            what = str(e)
        open(os.path.join(out_dir, f"c{rank}.txt"), "w").write(what)
        # the fit over 8 ranks: batch-norm sums and gradients all-reduced; 19 rows in batches of 8 -> a trailing batch of 3 rows,
        # fewer than ranks (five ranks join that step's all-reduces with sums over no rows)
        ws = glorot_uniform_weights((5, 5, 3), blocks=1, seed=0)
        rs = np.random.RandomState(3)
        Xf = rs.rand(19, 5, 5, 3).astype(np.float32); Yf = np.tanh(rs.randn(19, 3)).astype(np.float32)
        out = trainer_torch.fit(ws, (5, 5, 3), Xf, Yf, epochs=2, batch_size=8, lr_schedule=([100], [1e-3, 0.0]), device="cpu", seed=0,
                                verbose=False, dtype=torch.float64)
        np.savez(os.path.join(out_dir, f"f{rank}.npz"), *out)
    finally:
        dist.destroy_process_group()


def test_eight_rank_rehearsal_over_gloo(tmp_path):
    """no 8-GPU node has ever run this code, so the N = 8 path is rehearsed on the CPU: 8 ranks over gloo,
    uneven shards (three ranks below their share of 1 280 rows, one holding a single record): every rank computes the same split,
    10 240 distinct rows arrive rank-major on every rank, the counters average over all 262 144 games, the caller's NumPy
    stream is left alone; a rank without records fails on all eight ranks alike; the data-parallel fit (batch-norm sums and
    gradients all-reduced, a trailing batch smaller than the world) ends with the one-process fit's weights on every rank.
    The exchange being rehearsed: alpha_snake_zero_trainer.py:63-75."""
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [p for p in (repo, os.path.join(repo, "alphasnake-zero_amd")) if p not in sys.path]
    world, port = 8, _free_port()
    mp.spawn(_eight_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    res = [torch.load(tmp_path / f"e{r}.pt", weights_only=False) for r in range(world)]
    share = 10240 // 8
    for r in res:
        assert r["counts"] == EIGHT_COUNTS and (r["wanted"], r["batch"]) == (10240, 2048)
        assert r["rows"] == res[0]["rows"] and r["seed"] == res[0]["seed"] and sum(r["rows"]) == 10240
        assert [r["rows"][i] for i in (1, 3, 5)] == [10, 700, 1]                       # short ranks give all they have
        assert all(share <= r["rows"][i] <= EIGHT_COUNTS[i] for i in (0, 2, 4, 6, 7))   # the others make up the difference
        assert r["untouched"], "the exchange drew from the caller's global NumPy stream"
        Xg, Vg = r["Xg"], r["Vg"]
        assert Xg.shape == (10240, 3, 3, 3) and Vg.shape == (10240, 3)
        tags = {(int(a), int(b)) for a, b in zip(Xg[:, 0, 0, 0].tolist(), Xg[:, 0, 0, 1].tolist())}
        assert len(tags) == 10240, "a row was used twice"
        owners = Xg[:, 0, 0, 0].to(torch.int64)
        assert torch.equal(owners, torch.repeat_interleave(torch.arange(8), torch.tensor(r["rows"])))     # rank-major
        assert torch.equal(Vg[:, 0].to(torch.int64), owners)
        assert r["games"] == 262144 and np.allclose(r["avg"], [sum(range(1, 9)) / 262144.0] * 6)
        assert torch.equal(Xg, res[0]["Xg"])
    for r in range(world):
        assert "1 of 8 ranks recorded no state" in open(tmp_path / f"c{r}.txt").read()
    got = [[z[k] for k in z.files] for z in (np.load(tmp_path / f"f{r}.npz") for r in range(world))]
    from utils import trainer_torch
    from snake_engine.net import glorot_uniform_weights
    ws = glorot_uniform_weights((5, 5, 3), blocks=1, seed=0)
    rs = np.random.RandomState(3)
    Xf = rs.rand(19, 5, 5, 3).astype(np.float32); Yf = np.tanh(rs.randn(19, 3)).astype(np.float32)
    one = trainer_torch.fit(ws, (5, 5, 3), Xf, Yf, epochs=2, batch_size=8, lr_schedule=([100], [1e-3, 0.0]), device="cpu", seed=0,
                            verbose=False, dtype=torch.float64)
    for r in range(world):
        for a, b, c in zip(got[r], got[0], one):
            assert np.array_equal(a, b) and np.abs(a - c).max() <= 1e-9 * max(1.0, np.abs(c).max())


def test_shard_range_partitions_everything():
    from snake_engine.dist import shard_range, sample_share
    for total in (1, 7, 8, 4096, 262144):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1
    idx = sample_share(5, 5, np.random.RandomState(0))                # all of a short rank's records, each once
    assert sorted(idx.tolist()) == list(range(5))
    assert len(sample_share(0, 0, np.random.RandomState(0))) == 0
    with pytest.raises(AssertionError):
        sample_share(5, 32, np.random.RandomState(0))                 # more rows than records: share_counts never asks for that


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
