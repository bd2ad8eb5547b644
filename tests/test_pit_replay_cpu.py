"""CPU tests of the 'next' rows' fixtures: the oracle's pit loop against the winners the unmodified reference
produced (tests/golden/pit.npz), and the host-side replay.rep drawing (utils.game.Game.draw_tick) against the text the
reference's Game.draw wrote (tests/golden/replay.npz)."""
import numpy as np
import pytest

from conftest import load_golden

KEYS = ("alive", "health", "length", "dir", "nodes", "food", "rewards", "counters")


def _games(oracle, z, p, H, W, S, hd, n):
    return [oracle.Game.from_compact(H, W, S, hd, 0.15, {k: z[p + "init_" + k][g] for k in KEYS}) for g in range(n)]


@pytest.mark.parametrize("ci", [0, 1, 2, 3])
def test_oracle_pit_winners_match_the_reference(oracle, ci):
    from oracle.obs_key import StubNet
    from oracle.pit_oracle import pit_run
    z = load_golden("pit.npz")
    p = f"p{ci}_"
    H, W, S, hd, n, a_cnt = (int(v) for v in z[p + "meta"])
    games = _games(oracle, z, p, H, W, S, hd, n)
    winners, lengths = pit_run(games, StubNet(0), StubNet(1), None if a_cnt < 0 else a_cnt,
                               spawn_tape=lambda turn: z[p + "spawn"][turn - 1])
    assert [-1 if w is None else w for w in winners] == z[p + "winners"].tolist()
    assert lengths == z[p + "lengths"].tolist()


@pytest.mark.parametrize("ci", [0, 1])
def test_replay_rep_text_matches_the_reference(oracle, ci, tmp_path, monkeypatch):
    """both boards of every tick (game.py:140-141, 194-195): the product's host-side drawing, fed with pre/post states
    from the oracle (no GPU needed: Game.draw_tick only looks at host snapshots)"""
    from utils.game import Game
    z = load_golden("replay.npz")
    p = f"r{ci}_"
    H, W, S, hd = (int(v) for v in z[p + "meta"])
    g = _games(oracle, z, p, H, W, S, hd, 1)[0]
    monkeypatch.chdir(tmp_path)
    view = Game.__new__(Game)                      # a Game view without an engine behind it
    view.height, view.width, view.snake_cnt = H, W, S
    for t in range(len(z[p + "moves"])):
        pre = g.compact()
        dense = np.where(z[p + "moves"][t, 0] == 255, 1, z[p + "moves"][t, 0]).astype(np.uint8)
        g.tic(dense, spawn_cell=int(z[p + "spawn"][t, 0]))
        view._cache = g.compact()
        view.draw_tick(pre, dense)
    assert open("replay.rep", "rb").read() == z[p + "text"].tobytes()
    assert [0.0 if r is None else r for r in g.rewards] == z[p + "rewards"][0].tolist()
