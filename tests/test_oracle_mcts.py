"""CPU tests: the oracle's MCTS restatement (oracle/mcts_oracle.py) against the tiny MCTS self-play runs
recorded from the unmodified reference, and softermax/argmaxs against the recorded tables."""
import hashlib

import numpy as np
import pytest

from conftest import load_golden


@pytest.mark.parametrize("tag", ["tiny", "tiny_greedybase", "7x7x2", "19x19x8", "9x9x3"])
def test_mcts_self_play_matches_reference_bitwise(oracle, tag):
    from oracle.mcts_oracle import SelfPlayOracle, Draws
    from oracle.obs_key import StubNet
    z = load_golden(f"mcts_{tag}.npz")
    H, W, S, hd, n = int(z["H"]), int(z["W"]), int(z["S"]), int(z["hd"]), int(z["n_games"])
    games = []
    for g in range(n):
        st = {k: z["init_" + k][g] for k in ("alive", "health", "length", "dir", "nodes", "food", "rewards", "counters")}
        games.append(oracle.Game.from_compact(H, W, S, hd, 0.15, st))
    draws = Draws(tape=z["tape_u"])
    net = StubNet()
    sp = SelfPlayOracle(net, int(z["base"]), True, int(z["depth"]), int(z["breadth"]), draws)
    live = list(range(n))
    for t in range(int(z["n_turns"])):
        evals0 = sp.net_evals
        pos0 = draws.pos
        rows, V, moves = sp.root_turn([games[g] for g in live])
        ids = [(live[gi], s) for gi, s in rows]
        assert ids == [tuple(r) for r in z[f"t{t}_ids"].tolist()], f"turn {t}: ids"
        assert np.array(V, np.float32).tobytes() == z[f"t{t}_V"].tobytes(), f"turn {t}: root Q values"
        assert moves == z[f"t{t}_moves"].tolist(), f"turn {t}: moves"
        assert sp.net_evals - evals0 == z["turn_evals"][t], f"turn {t}: net evaluations (cache de-duplication)"
        assert len(sp.Q) == z["turn_cache"][t], f"turn {t}: cache size after eviction"
        assert draws.pos == z["turn_tape_pos"][t], f"turn {t}: draws consumed ({pos0} -> {draws.pos})"
        dense = {g: np.ones(S, np.uint8) for g in live}
        for (g, s), m in zip(ids, moves):
            dense[g][s] = m
        nxt = []
        for g in live:
            if not games[g].tic(dense[g], spawn_cell=int(z["turn_spawn"][t][g])):
                nxt.append(g)
        live = nxt
    dig = np.array([np.frombuffer(hashlib.blake2b(r.tobytes(), digest_size=16).digest(), np.uint8) for r in sp.records])
    assert np.array_equal(dig, z["records_digest"])
    assert np.array(sp.values, np.float32).tobytes() == z["values_final"].tobytes()
    assert draws.pos == len(z["tape_u"])


def test_softermax_argmax_tables():
    from oracle.mcts_oracle import softermax, argmaxs
    z = load_golden("tables.npz")
    for base in (2, 3, 10, 100):
        got = np.array([softermax(base, v) for v in z["z"]], np.float32)
        assert got.tobytes() == z[f"pmf_b{base}"].tobytes()
    assert argmaxs(list(z["argmax_z"])) == z["argmax"].tolist()
