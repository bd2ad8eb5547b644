"""CPU tests of oracle/mcts_cpu.c -- the C restatement of the MCTS self-play loop that bench.py times as `cpu_baseline`
-- against the runs recorded from the unmodified reference (tests/golden/mcts_tiny*.npz): ids, moves, evaluation counts,
cache sizes, draws consumed and record bytes identical, root Q values within 1e-5 (libm vs NumPy transcendental
functions), with the stub net inside the C code and with the same stub net called back through the batch interface."""
import hashlib

import numpy as np
import pytest

from conftest import load_golden

KEYS = ("alive", "health", "length", "dir", "nodes", "food", "rewards", "counters")


@pytest.mark.parametrize("tag", ["tiny", "tiny_greedybase", "7x7x2", "19x19x8", "9x9x3"])
@pytest.mark.parametrize("callback_net", [False, True])
def test_c_mcts_replays_the_reference_run(oracle, tag, callback_net):
    from oracle.mcts_cpu import CpuSelfPlay
    from oracle.obs_key import StubNet
    z = load_golden(f"mcts_{tag}.npz")
    H, W, S, hd, n = int(z["H"]), int(z["W"]), int(z["S"]), int(z["hd"]), int(z["n_games"])
    games = [oracle.Game.from_compact(H, W, S, hd, 0.15, {k: z["init_" + k][g] for k in KEYS}) for g in range(n)]
    net = StubNet()
    sp = CpuSelfPlay(games, net=(lambda X: net.v(X)) if callback_net else None, threads=1, base=int(z["base"]),
                     training=True, max_depth=int(z["depth"]), max_breadth=int(z["breadth"]))
    w = sp.workers[0]
    w.set_tape(z["tape_u"])
    worst = 0.0
    for t in range(int(z["n_turns"])):
        ev0 = w.stats()["net_evals"]
        w.set_spawn_tape(np.where(z["turn_spawn"][t] < -1, -1, z["turn_spawn"][t]))
        sp.root_turn()
        ids, V, moves = w.last()
        assert ids.tolist() == z[f"t{t}_ids"].tolist(), f"turn {t}: ids"
        assert moves.tolist() == z[f"t{t}_moves"].tolist(), f"turn {t}: moves"
        worst = max(worst, float(np.abs(V - z[f"t{t}_V"]).max()))
        assert w.stats()["net_evals"] - ev0 == z["turn_evals"][t], f"turn {t}: net evaluations"
        assert w.L.mc_cache_size(w.h) == z["turn_cache"][t], f"turn {t}: cache size after eviction"
        assert w.L.mc_tape_pos(w.h) == z["turn_tape_pos"][t], f"turn {t}: draws consumed"
    assert worst <= 1e-5, worst
    rec, val = w.records()
    dig = np.array([np.frombuffer(hashlib.blake2b(r.tobytes(), digest_size=16).digest(), np.uint8) for r in rec])
    assert np.array_equal(dig, z["records_digest"])
    assert w.L.mc_tape_pos(w.h) == len(z["tape_u"])
    if callback_net:
        assert sum(net.calls) == int(z["turn_evals"].sum())
    sp.close()


def test_sharded_run_finishes_and_counts(oracle):
    """3 shards on 3 threads, stub net, run to completion: every game ends, env-steps = sum of game lengths"""
    from oracle.mcts_cpu import CpuSelfPlay, seeded_games
    games = seeded_games(12, health_dec=9, seed=3)
    sp = CpuSelfPlay(games, net=None, threads=3, base=2, max_depth=4, max_breadth=8, seed=5)
    assert sp.threads == 3
    st = sp.run()
    tot = sum(w.totals() for w in sp.workers)
    assert st["env_steps"] == tot[5] and tot[5] >= 12 * 3
    assert all(w.L.mc_n_live(w.h) == 0 for w in sp.workers)
    assert st["n_records"] > 0 and st["net_evals"] > 0 and st["sim_steps"] > st["env_steps"]
    # same seeds -> same work (the baseline is a fixed amount of work)
    sp2 = CpuSelfPlay(seeded_games(12, health_dec=9, seed=3), net=None, threads=3, base=2, max_depth=4, max_breadth=8, seed=5)
    assert sp2.run() == st
    sp.close(); sp2.close()
