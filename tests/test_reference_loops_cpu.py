"""CPU pins of the oracle against the runs of the reference's own generation loop (tests/golden/trainer.npz, recorded by
tests/golden/make_golden.py::record_trainer from the unmodified alpha_snake_zero_trainer.py) and of the iteration-end
sampling arithmetic (alpha_snake_zero_trainer.py:63-72).  The C restatement of the self-play loop (oracle/mcts_cpu.c)
replays each recorded generation from its start boards, spawn tape and uniform draws: recorded root states byte-identical,
values within 1e-5, the six per-game averages formatted as the reference's log.csv row, byte for byte."""
import hashlib

import numpy as np
import pytest

from conftest import load_golden

KEYS = ("alive", "health", "length", "dir", "nodes", "food", "rewards", "counters")


def rebuild_tape(z, p):
    n, rej = int(z[p + "tape_len"]), z[p + "tape_rejected"]
    raw = np.random.RandomState(int(z[p + "tape_seed"])).random_sample(n + len(rej))
    u = np.delete(raw, rej)
    assert hashlib.blake2b(u.tobytes(), digest_size=16).digest() == z[p + "tape_digest"].tobytes()
    return u


@pytest.mark.parametrize("tag", ["gen0", "gen8", "gen32"])
def test_c_self_play_replays_the_generations_of_the_reference_trainer(oracle, tag):
    from oracle.mcts_cpu import CpuSelfPlay
    from oracle.obs_key import StubNet
    z = load_golden("trainer.npz")
    p = tag + "_"
    tape, tape_pos = rebuild_tape(z, p), z[p + "tape_pos"]
    start = int(z[p + "ctor"][3])
    log_lines = z[p + "log_csv"].tobytes().decode().splitlines()
    rows = [ln for ln in log_lines if ln[0].isdigit()]
    if start == 0:                                  # trainer.py:35-41
        assert log_lines[:2] == ["new model g", "iteration, wall_collision, body_collision, head_collision, "
                                                "starvation, food_eaten, game_length"]
    assert len(rows) == 2 and len(log_lines) == (4 if start == 0 else 2)
    for gi in range(2):
        q = p + f"g{gi}_"
        H, W, S, hd, n = (int(v) for v in z[p + "runner_args"][gi])
        base, training, depth, breadth = (int(v) for v in z[p + "agent_args"][gi])
        assert (hd, base, training) == (9 if start + gi <= 8 else 3 if start + gi <= 32 else 1, 2 + start + gi, 1)
        games = [oracle.Game.from_compact(H, W, S, hd, 0.15, {k: z[q + "init_" + k][g] for k in KEYS}) for g in range(n)]
        net = StubNet(int(z[p + "net_which"][gi]))
        sp = CpuSelfPlay(games, net=lambda X: net.v(X), threads=1, base=base, training=True, max_depth=depth, max_breadth=breadth)
        w = sp.workers[0]
        w.set_tape(tape[int(tape_pos[gi]):])
        spawn = z[q + "spawn"]
        for t in range(len(spawn)):
            w.set_spawn_tape(np.where(spawn[t] < -1, -1, spawn[t]))
            left = sp.root_turn()
        assert left == 0, "every game of the generation ended on the recorded turn"
        end = int(tape_pos[gi + 1]) if gi + 1 < len(tape_pos) else len(tape)
        assert w.L.mc_tape_pos(w.h) == end - int(tape_pos[gi]), "draws consumed"
        rec, val = w.records()
        dig = np.array([np.frombuffer(hashlib.blake2b(r.tobytes(), digest_size=8).digest(), np.uint8) for r in rec])
        assert np.array_equal(dig, z[q + "records_digest"])
        assert np.abs(val - z[q + "values"]).max() <= 1e-5
        tot = w.totals()
        row = str(start + gi) + ", " + ", ".join(str(int(v) / n) for v in tot)     # mp_game_runner.py:71-76, trainer.py:56-58
        assert row == rows[gi]
        # what the loop hands to the fit: the sampled rows, then their mirror images (trainer.py:72-77, 93-100)
        idx = z[q + "sample_idx"]
        X = np.concatenate([rec[idx], rec[idx][:, :, ::-1]])
        assert len(X) == int(z[q + "X_rows"])
        assert hashlib.blake2b(X.tobytes(), digest_size=32).digest() == z[q + "X_digest"].tobytes()
        sp.close()


def test_sample_plan_is_the_reference_arithmetic_for_any_world_size():
    """rows sampled = 2048 * min(5, len(records) // 2048) whatever the number of ranks (trainer.py:63-72 on ALL records)"""
    from snake_engine.dist import sample_plan, sample_share, share_counts
    for n in (2048, 2049, 4095, 4096, 10239, 10240, 51200, 1_000_000):
        ref_batches = min(5, n // 2048)
        for world in (1, 2, 4, 8):
            wanted, batch, share = sample_plan(n, world)
            assert (wanted, batch, share * world) == (2048 * ref_batches, 2048, 2048 * ref_batches)
            # eight ranks with ~n/8 records each still gather the single-process row count
            per_rank = [n // world + (1 if r < n % world else 0) for r in range(world)]
            rows = share_counts(per_rank, wanted, seed=n + world)
            got = sum(len(sample_share(k, rows[r], np.random.RandomState(r))) for r, k in enumerate(per_rank))
            assert got == wanted and all(v == wanted // world for v in rows)
    # fewer than one batch of records: everything is one batch (the reference fails in `flip` here)
    assert sample_plan(1000, 1) == (1000, 1000, 1000)
    assert sample_plan(1001, 2) == (1000, 1000, 500)


def test_pit_script_fixture_is_self_consistent():
    """pit.py:37-55: the scores in the recorded pit.txt follow from the recorded winner indices"""
    z = load_golden("pit_script.npz")
    lines = z["pit_txt"].tobytes().decode().splitlines()
    assert lines[0] == "m3 is set to be the baseline champion."
    for ci in range(int(z["n"])):
        w = z[f"c{ci}_winners"]
        a_cnt = int(z[f"c{ci}_args"][5])
        win = float((w >= a_cnt).sum()) + 0.5 * float((w < 0).sum())
        loss = float(((w >= 0) & (w < a_cnt)).sum()) + 0.5 * float((w < 0).sum())
        score = win / (win + loss)
        tag = f"m{4 + ci}"
        want = (tag + " beats the previouse champion. score = " + str(score) + ". It is the new champion!") if score > 0.51 \
            else (tag + " failed to beat the previouse champion. score = " + str(score) + ".")
        assert lines[1 + ci] == want


def test_ladder_arithmetic_and_line_formats_reproduce_the_reference_pit_txt():
    """alphasnake-zero_amd/pit.py's score arithmetic and its three pit.txt line formats, fed the winner indices the reference's
    own pit.py produced, give the reference's pit.txt byte for byte (the games themselves are replayed on the GPU,
    tests/test_reference_loops_gpu.py)"""
    import importlib
    import sys
    sys.modules.pop("pit", None)
    pit = importlib.import_module("pit")
    z = load_golden("pit_script.npz")
    text = "m3 is set to be the baseline champion.\n"
    for ci in range(int(z["n"])):
        winners = [None if w < 0 else int(w) for w in z[f"c{ci}_winners"]]
        score = pit.challenger_score(winners, int(z[f"c{ci}_args"][5]))
        text += pit.verdict_line(f"m{4 + ci}", score, pit.THRESHOLD)
    assert text.encode() == z["pit_txt"].tobytes()
    assert (pit.PIT_GAMES, pit.HEIGHT, pit.WIDTH, pit.SNAKE_CNT) == tuple(int(v) for v in z["c0_args"][[4, 0, 1, 2]])


def test_log_csv_row_format_of_the_mirror_trainer_is_the_reference_text(tmp_path, monkeypatch):
    """utils.alpha_snake_zero_trainer._log_row / the header written by train(): fed the six averages of a recorded generation
    they write the reference's log.csv bytes (trainer.py:35-41, 56-61)"""
    import utils.alpha_snake_zero_trainer as T
    z = load_golden("trainer.npz")
    lines = z["gen0_log_csv"].tobytes().decode().splitlines()
    monkeypatch.chdir(tmp_path)

    class Runner:
        game_cnt = 48
    r = Runner()
    vals = [float(v) for v in lines[2].split(", ")[1:]]
    for k, v in zip(T.LOG_FIELDS, vals):
        setattr(r, k, v)
    tr = T.AlphaSnakeZeroTrainer(48, 4, 8, 2e-4, 0.9)
    with open("log.csv", "a") as f:                      # what train() writes before the first generation
        f.write("new model g\n")
        f.write("iteration, wall_collision, body_collision, head_collision, starvation, food_eaten, game_length\n")
    tr._log_row(r, 0)
    assert open("log.csv").read().splitlines() == lines[:3]
    assert [T.AlphaSnakeZeroTrainer.health_dec_for(i) for i in (0, 8, 9, 32, 33, 100)] == [9, 9, 3, 3, 1, 1]
