"""CPU restatement of the pit loop (TEST INFRASTRUCTURE, see oracle/__init__.py).

Follows /root/reference/code/utils/pit_mp_game_runner.py:14-63 (team split by snake id, greedy agents, early exit when
one team is eliminated, winner = last index with reward +1.0 / first alive snake's id) and pit_agent.py:10-28
(nnet.v + argmaxs with strict '>' comparisons) over oracle/snake_oracle.c games.  Pinned by tests/golden/pit.npz,
which the unmodified reference produced (tests/golden/make_golden.py::record_pit).
"""
import numpy as np

from .mcts_oracle import argmaxs


def pit_run(games, net_a, net_b, alice_snake_cnt=None, spawn_tape=None, draws=None, spawn_log=None):
    """games: list of oracle Game objects (index = game id); spawn_tape(turn) -> spawn cell per game id (-1 none), or
    draws(turn, game id) -> the two uniforms Game.tic would draw (game.py:131-133), the cell chosen then appended to
    spawn_log[turn - 1][game id].  Returns (winners list with None for draws, per-game number of ticks played)."""
    S = games[0].g.S
    if alice_snake_cnt is None:
        alice_snake_cnt = S // 2                                            # pit_mp_game_runner.py:17-18
    winners = [None] * len(games)
    lengths = [0] * len(games)
    live = list(range(len(games)))
    turn = 0
    while live:
        turn += 1
        st_a, id_a, st_b, id_b = [], [], [], []
        for g in live:                                                      # :26-32
            ids = games[g].alive_ids()
            sts = games[g].get_states()
            for s, st in zip(ids, sts):
                if s < alice_snake_cnt:
                    st_a.append(st); id_a.append((g, s))
                else:
                    st_b.append(st); id_b.append((g, s))
        moves = argmaxs(net_a.v(st_a)) + argmaxs(net_b.v(st_b))             # :33, pit_agent.py:10-13
        dense = {g: np.ones(S, np.uint8) for g in live}
        for (g, s), m in zip(id_a + id_b, moves):
            dense[g][s] = m
        nxt = []
        tape = spawn_tape(turn) if spawn_tape is not None else None
        if spawn_log is not None:
            spawn_log.append(np.full(len(games), -2, np.int16))
        for g in live:
            if draws is not None:
                done = games[g].tic(dense[g], draws=draws(turn, g))
                if spawn_log is not None:
                    spawn_log[-1][g] = games[g].last_spawn
            else:
                done = games[g].tic(dense[g], spawn_cell=int(tape[g]) if tape is not None else -1)
            lengths[g] += 1
            if done:                                                        # :43-47
                for i, r in enumerate(games[g].rewards):
                    if r == 1.0:
                        winners[g] = i
            else:                                                           # :48-60
                ids = games[g].alive_ids()
                a = any(s < alice_snake_cnt for s in ids)
                b = any(s >= alice_snake_cnt for s in ids)
                if not a or not b:
                    winners[g] = ids[0]
                else:
                    nxt.append(g)
        live = nxt
    return winners, lengths
