"""GPU tests of the rows SURVEY.md section 8f lists as "next": the pit pair (f-2), AlphaNNet's training half
and checkpoint round trip (f-1, f-3) driven the way the reference's trainer / pit scripts drive them."""
import os

import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu


class TableNet:
    """a net with the reference's .v(X) contract: deterministic stub (test infrastructure)"""

    def v(self, X):
        from oracle.obs_key import stub_q
        return stub_q(np.array(X, np.float32))


KEYS = ("alive", "health", "length", "dir", "nodes", "food", "rewards", "counters")


def _import_golden_boards(runner, z, p, H, W, S, n):
    from snake_engine.engine import state_from_compact
    runner.engine.import_states([state_from_compact(H, W, S, {k: z[p + "init_" + k][g] for k in KEYS}) for g in range(n)])
    for g in runner.games.values():
        g._dirty()


@pytest.mark.parametrize("ci", [0, 1, 2, 3])
def test_pit_winners_match_the_reference_bit_exact(ci):
    """pit_mp_game_runner.MPGameRunner.run(Alice, Bob, Alice_snake_cnt) on the HIP engine: the reference's start boards
    and food-spawn tape in, the reference's winner indices out (1v3, 2v2 by default split, 3v1, 2v2 with health_dec 1);
    the agents are the product's pit agents around the two stub nets the reference run used."""
    from oracle.obs_key import StubNet
    from utils.pit_agent import Agent
    from utils.pit_mp_game_runner import MPGameRunner
    z = load_golden("pit.npz")
    p = f"p{ci}_"
    H, W, S, hd, n, a_cnt = (int(v) for v in z[p + "meta"])
    gr = MPGameRunner(H, W, S, hd, n, seed=1)
    _import_golden_boards(gr, z, p, H, W, S, n)
    winners = gr.run(Agent(StubNet(0)), Agent(StubNet(1)), None if a_cnt < 0 else a_cnt,
                     spawn_tape=lambda turn: z[p + "spawn"][turn - 1])
    assert [-1 if w is None else w for w in winners] == z[p + "winners"].tolist()
    assert len(gr.games) == 0


def test_pit_pair_tracks_the_oracle_on_unrecorded_setups():
    """beyond the four recorded reference runs: 30 random setups (7x7 .. 19x19, 2-6 snakes, any team split, three stub nets);
    the oracle's pit loop plays first, the device replays with the oracle's food spawns: every winner index identical
    (tools/fuzz_pit.py; 400 more setups: profiles/r3_fuzz_pit.log)"""
    import subprocess
    import sys
    from conftest import REPO
    r = subprocess.run([sys.executable, os.path.join(REPO, "tools", "fuzz_pit.py"), "2000", "30"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "fuzz ok: 30 pit runs" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]


def test_pit_agent_device_path_equals_host_path(oracle):
    """pit_agent.Agent.make_moves on a device tensor of planes (the engine-backed runner's form) and on the reference's
    list of arrays give the same greedy moves"""
    import torch
    from utils.pit_agent import Agent
    from oracle.mcts_oracle import argmaxs
    z = load_golden("pit.npz")
    net = TableNet()
    for g in range(4):
        game = oracle.Game.from_compact(11, 11, 4, 3, 0.15, {k: z["p0_init_" + k][g] for k in KEYS})
        sts = game.get_states()
        V = net.v(sts)
        assert Agent(net).make_moves(sts) == argmaxs(V)
        assert Agent(net).make_moves(torch.as_tensor(np.array(sts), device="cuda")) == argmaxs(V)


@pytest.mark.parametrize("ci", [0, 1])
def test_replay_rep_matches_the_reference_text(ci, tmp_path, monkeypatch):
    """MPGameRunner(game_cnt=1).run shows the game: two boards per tick appended to ./replay.rep (game.py:140-141,
    194-195, 281-300); same start board, moves and spawn tape as the reference run -> the same bytes"""
    from utils.mp_game_runner import MPGameRunner
    z = load_golden("replay.npz")
    p = f"r{ci}_"
    H, W, S, hd = (int(v) for v in z[p + "meta"])
    monkeypatch.chdir(tmp_path)
    MPGameRunner.verbose = False
    gr = MPGameRunner(H, W, S, hd, 1, seed=1)
    _import_golden_boards(gr, z, p, H, W, S, 1)

    class Taped:
        t = 0

        def make_moves(self, games, ids):
            out = [int(z[p + "moves"][self.t][gid][sid]) for gid, sid in ids]
            self.t += 1
            return out
    rewards = gr.run(Taped(), spawn_tape=lambda turn: z[p + "spawn"][turn - 1])
    assert np.array_equal(np.array(rewards, np.float32), z[p + "rewards"])
    assert open("replay.rep", "rb").read() == z[p + "text"].tobytes()


def test_run_resumed_in_pieces_equals_one_run():
    """run(max_turns=k) resumed until the games are over: same rewards and per-game averages as one uninterrupted run
    (the counters are totals / game_cnt, never an average of an average)"""
    from utils.mp_game_runner import MPGameRunner
    z = load_golden("runner.npz")
    H, W, S, hd, n = int(z["H"]), int(z["W"]), int(z["S"]), int(z["hd"]), int(z["n_games"])
    MPGameRunner.verbose = False

    class Taped:
        t = 0

        def make_moves(self, games, ids):
            out = [int(z["moves"][self.t][gid][sid]) for gid, sid in ids]
            self.t += 1
            return out
    gr = MPGameRunner(H, W, S, hd, n, seed=1)
    _import_golden_boards(gr, z, "", H, W, S, n)
    agent = Taped()
    steps = 0
    while gr.games:
        rewards = gr.run(agent, spawn_tape=lambda turn: z["spawn"][turn - 1], max_turns=3)
        steps += gr.env_steps
    assert np.array_equal(np.array(rewards, np.float32), z["rewards"])
    got = [gr.wall_collision, gr.body_collision, gr.head_collision, gr.starvation, gr.food_eaten, gr.game_length]
    assert got == z["counters"].tolist()
    assert steps == int(z["game_lengths"].sum()) and gr.turns == len(z["moves"])


def test_trainer_flow_train_copy_save_load(tmp_path, monkeypatch):
    """alpha_snake_zero_trainer.py:52-91 in miniature: self-play -> sample -> mirror -> copy_and_compile(lr) ->
    train -> copy_and_compile() -> save -> AlphaNNet(model_name=...) gives the same Q values"""
    import random
    from utils.agent import Agent
    from utils.alpha_nnet import AlphaNNet
    from utils.mp_game_runner import MPGameRunner
    random.seed(1); np.random.seed(1)
    monkeypatch.chdir(tmp_path)
    os.mkdir("models")
    MPGameRunner.verbose = False
    nnet = AlphaNNet(input_shape=(21, 21, 3)).copy_and_compile()
    nnet.save("t0")
    Alice = Agent(nnet, 2, True, 4, 8, seed=3)
    gr = MPGameRunner(11, 11, 4, 9, 6, seed=4)
    gr.run(Alice)
    n = len(Alice.records)
    assert n == len(Alice.values) and n > 20
    idx = random.sample(range(n), min(n, 64))
    X = [Alice.records[i] for i in idx]
    V = [Alice.values[i] for i in idx]
    Alice.clear()
    X += list(np.flip(X, axis=2))                       # trainer.py:93-100 mirror augmentation
    V += list(np.flip(V, axis=1))
    before = nnet.v(X[:8])
    trained = nnet.copy_and_compile(learning_rate=1e-3)
    assert trained.lr_schedule == ([20, 40, 60, 80, 100], [1e-3, 2.5e-4, 6.25e-5, 1.5625e-5, 3.90625e-6, 0.0])
    trained.train(X, V, epochs=2, batch_size=len(idx))
    after = trained.v(X[:8])
    assert np.isfinite(after).all() and not np.allclose(before, after)
    final = trained.copy_and_compile()
    final.save("t1")
    re = AlphaNNet(model_name="models/t1.h5")
    assert np.array_equal(re.v(X[:8]), final.v(X[:8]))
    assert len(re.v_net.get_weights()) == 54
    with pytest.raises(OSError):
        AlphaNNet(model_name="models/t2.h5")


def test_alpha_nnet_loads_a_keras_shaped_file(tmp_path):
    """alpha_nnet.py:12: `AlphaNNet(model_name)` on a file laid out the way Keras 2.2.4-tf + h5py write `v_net.save` (variable-length
    string attributes, empty float64 weight_names of the weightless layers, optimizer_weights, training_config; written with libhdf5
    by tests/test_checkpoint_cpu.py, not by this build's writer) evaluates like a net made from the same weights; a truncated copy
    raises the OSError pit.py:58 waits on"""
    from test_checkpoint_cpu import _weights, _write_keras_style
    from utils.alpha_nnet import AlphaNNet
    ws = _weights((21, 21, 3), 4, seed=9)
    path = str(tmp_path / "gen7.h5")
    _write_keras_style(path, ws, (21, 21, 3), True)
    X = list(load_golden("states_11x11x4.npz")["raw"][:24])
    a = AlphaNNet(model_name=path)
    b = AlphaNNet(input_shape=(21, 21, 3), _weights=ws)
    assert a.input_shape == (21, 21, 3) and np.array_equal(a.v(X), b.v(X))
    assert all(u.tobytes() == v.tobytes() for u, v in zip(a.v_net.get_weights(), ws))
    blob = open(path, "rb").read()
    open(path, "wb").write(blob[:len(blob) // 3])
    with pytest.raises(OSError):
        AlphaNNet(model_name=path)


def test_alpha_snake_zero_trainer_two_generations(tmp_path, monkeypatch):
    """utils.alpha_snake_zero_trainer.AlphaSnakeZeroTrainer driven exactly as train.py drives it (train.py:32-40), for two
    generations: log.csv gets the reference's header and one row of six per-game averages per generation, the learning rate
    decays, models/<name><n>.h5 appear and load back"""
    import random
    from utils.alpha_nnet import AlphaNNet
    from utils.alpha_snake_zero_trainer import AlphaSnakeZeroTrainer
    from utils.mp_game_runner import MPGameRunner
    random.seed(2); np.random.seed(2)
    monkeypatch.chdir(tmp_path)
    os.mkdir("models")
    MPGameRunner.verbose = False
    ANNet = AlphaNNet(input_shape=(21, 21, 3))
    ANNet.save("gen0")
    trainer = AlphaSnakeZeroTrainer(6, 4, 8, 1e-3, 0.98, 11, 11, 4, None)
    last = trainer.train(ANNet, name="gen", iteration=0, max_iterations=2)
    lines = open("log.csv").read().splitlines()
    assert lines[0] == "new model gen"
    assert lines[1] == "iteration, wall_collision, body_collision, head_collision, starvation, food_eaten, game_length"
    rows = [l.split(", ") for l in lines[2:]]
    assert [r[0] for r in rows] == ["0", "1"] and all(len(r) == 7 for r in rows)
    for r in rows:
        vals = [float(v) for v in r[1:]]
        assert vals[5] > 1 and abs(sum(vals[:4]) - round(sum(vals[:4]) * 6) / 6) < 1e-9       # averages over 6 games
    assert abs(trainer.lr - 1e-3 * 0.98 ** 2) < 1e-12
    assert os.path.exists("models/gen1.h5") and os.path.exists("models/gen2.h5")
    re = AlphaNNet(model_name="models/gen2.h5")
    X = np.zeros((2, 21, 21, 3), np.float32); X[..., 1] = 1.0
    assert np.array_equal(re.v(list(X)), last.v(list(X)))
    w0, w2 = ANNet.v_net.get_weights(), re.v_net.get_weights()
    assert max(float(np.abs(a - b).max()) for a, b in zip(w0, w2)) > 1e-4          # it trained


def test_trainer_generation_on_two_ranks(tmp_path):
    """the generation loop under torch.distributed (two ranks sharing the box's GPU over gloo): the 10 games are cut 5 + 5,
    the sampled rows are all-gathered, the fit runs data-parallel; both ranks end with bit-identical weights, rank 0 alone
    writes log.csv (one row: averages over all 10 games) and the model file"""
    import socket
    import subprocess
    import sys
    from conftest import REPO
    os.mkdir(tmp_path / "models")
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(REPO, "tests", "helpers", "trainer_rank.py"), str(tmp_path)],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs[0][-2000:] + outs[1][-2000:]
    w0, w1 = np.load(tmp_path / "weights_r0.npz"), np.load(tmp_path / "weights_r1.npz")
    assert len(w0.files) == 54
    for k in w0.files:
        assert np.array_equal(w0[k], w1[k]), k
    lines = open(tmp_path / "log.csv").read().splitlines()
    assert lines[0] == "new model dp" and len(lines) == 3 and lines[2].startswith("0, ")
    vals = [float(v) for v in lines[2].split(", ")[1:]]
    assert abs(vals[5] * 10 - round(vals[5] * 10)) < 1e-9            # game_length averaged over all 10 games
    assert os.path.exists(tmp_path / "models" / "dp1.h5") and not os.path.exists(tmp_path / "models" / "dp2.h5")
    # the rows the fit gets = what ONE process would sample from the same total number of records (trainer.py:63-72),
    # times two for the mirror images -- not what the smaller rank's count alone would give
    import json
    from snake_engine.dist import sample_plan
    c0, c1 = (json.load(open(tmp_path / f"collect_r{r}.json")) for r in range(2))
    total = c0["records"] + c1["records"]
    wanted, batch, _ = sample_plan(total, 2)
    assert c0["rows"] == c1["rows"] == 2 * wanted and c0["batch"] == c1["batch"] == batch
    assert wanted == ((total // 2) * 2 if total < 2048 else 2048 * min(5, total // 2048))


def test_pit_scripts_call_sequence_with_real_nets(tmp_path, monkeypatch):
    """the bodies of pit.py (2 snakes, models loaded from .h5, :15-46) and test_pit.py (1v3 both ways, then 2v2, :13-65) with
    two real nets on the device path (observation tensor + obstacle mask straight into the MFMA net)"""
    from utils.pit_agent import Agent
    from utils.alpha_nnet import AlphaNNet
    from utils.pit_mp_game_runner import MPGameRunner
    import random
    random.seed(7); np.random.seed(7)
    monkeypatch.chdir(tmp_path)
    os.mkdir("models")
    AlphaNNet(input_shape=(21, 21, 3)).save("m1")
    AlphaNNet(input_shape=(21, 21, 3)).save("m2")
    nnet1, nnet2 = AlphaNNet(model_name="models/m1.h5"), AlphaNNet(model_name="models/m2.h5")
    Alice, Bob = Agent(nnet1), Agent(nnet2)
    n = 40
    for snake_cnt, a_cnt, first, second in ((4, 1, Alice, Bob), (4, 1, Bob, Alice), (2, 1, Alice, Bob)):
        gr = MPGameRunner(11, 11, snake_cnt, 1, n)
        winner_ids = gr.run(first, second, a_cnt)
        assert len(winner_ids) == n and all(w is None or 0 <= w < snake_cnt for w in winner_ids)
        win = sum(1 for w in winner_ids if w is not None and w < a_cnt)
        draw = sum(1 for w in winner_ids if w is None)
        assert 0 <= win + draw <= n and len(gr.games) == 0
    with pytest.raises(OSError):                                   # pit.py:58 polls for the next generation this way
        AlphaNNet(model_name="models/m3.h5")


def test_entry_scripts_train_then_ladder_then_matches(tmp_path, monkeypatch, capsys):
    """the package's own train.py / pit.py / test_pit.py / test_model.py / test_weights.py (counterparts of the reference's
    entry scripts) on a tiny setting: train.py writes models/<name>0..2.h5 and log.csv, pit.py's ladder plays generations 1
    and 2 against the champion and writes pit.txt in the reference's line formats with the score of the games it played,
    then polls once for generation 3 and gives up; test_pit.py's three matches, test_model.py's replay, test_weights.py"""
    import importlib.util
    import random
    import re
    from conftest import PKG
    from utils.mp_game_runner import MPGameRunner

    def load_script(name):
        spec = importlib.util.spec_from_file_location("entry_" + name, os.path.join(PKG, name + ".py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod
    random.seed(11); np.random.seed(11)
    monkeypatch.chdir(tmp_path)
    MPGameRunner.verbose = False
    train, pit = load_script("train"), load_script("pit")
    train.start("lad", 0, max_iterations=2, self_play_games=6, max_MCTS_depth=4, max_MCTS_breadth=8,
                initial_learning_rate=1e-3)
    assert all(os.path.exists(f"models/lad{g}.h5") for g in (0, 1, 2)) and not os.path.exists("models/lad3.h5")
    assert open("log.csv").read().splitlines()[0] == "new model lad"
    resumed = train.start("lad", 2, max_iterations=0)                     # a later start loads the file (train.py:34-36)
    assert resumed.input_shape == (21, 21, 3)

    played = pit.ladder("lad", 0, pit_games=24, poll_seconds=0, max_polls=1)
    assert [g for g, _, _ in played] == [1, 2]
    lines = open("pit.txt").read().splitlines()
    assert lines[0] == "lad0 is set to be the baseline champion." and len(lines) == 3
    for (g, score, took), line in zip(played, lines[1:]):
        assert 0.0 <= score <= 1.0 and abs(score * 48 - round(score * 48)) < 1e-9     # half points over 24 games
        assert took == (score > 0.51)
        assert line + "\n" == pit.verdict_line(f"lad{g}", score)
        assert re.fullmatch(r"lad\d (beats|failed to beat) the previouse champion\. score = [0-9.e-]+\.( It is the new champion!)?", line)
    assert "A new challenger, lad1" in capsys.readouterr().out

    load_script("test_pit").main(["lad1", "lad2"], games=12)
    out = capsys.readouterr().out
    assert out.count("Running games...") == 3 and out.count("Competing time") == 3
    assert re.search(r"1v3 Win Rate of lad1 [0-9.]+ Draw Rate = [0-9.]+", out) and "2v2 Win Rate of lad2" in out

    rewards, runner = load_script("test_model").play_one("lad2")
    assert len(rewards) == 1 and sorted(set(rewards[0])) in ([-1.0], [-1.0, 1.0])
    boards = open("replay.rep").read().split("\n\n")
    assert len(boards) >= 2 * runner.game_length and boards[0].startswith("[")

    load_script("test_weights").main(["lad2"])
    out = capsys.readouterr().out
    assert out.count("Min weight:") == 54 and "(3, 3, 128, 128)" in out
