"""The production search at BASELINE.json's full sizes, under the driver's GPU tests: ONE root turn of configs[1] (4 096 games,
max_MCTS_breadth 50), of configs[2] (32 768 games, breadth 200) and of one GPU's share of the configs[4] shape (19x19, 8 snakes,
4 096 games) through the drop-in classes (utils.mp_game_runner.MPGameRunner
-> utils.agent.Agent -> snake_engine.DeviceMCTS: Agent.make_moves agent.py:25-111, the rollout ticks agent.py:161-223,
MPGameRunner.run mp_game_runner.py:23-77) on mid-game boards, with the deterministic device stub net (tests/stubnet_device.py:
no convolution time, so the turn costs seconds) and a transposition table that starts too small (the growth path runs).

Checked against the oracle on a strided sample of >= 2 000 root rows: the recorded state's bytes == oracle Game.make_state;
the chosen move is open per the reference's obstacle test (oracle.obs_key.obstacle_mask, alpha_nnet.py:63-76); V finite, in
[-1, 1], exactly -1 on blocked moves; the root tick itself (moves + the device's food spawn) replays on the C oracle to the same
boards, rewards and counters.  Globally: net evaluations == table entries created, records == alive root snakes, the six
counters of snk_engine_sum_counters_sync == the sums over the exported boards (and over the oracle's boards on the sample), the
runner's totals == the counters of the games that ended, the table grew.
"""
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

FIELDS = ("alive", "health", "length", "dir", "nodes", "food", "rewards", "counters")


def _mid_game_boards(torch, eng, n, ticks, seed):
    """`ticks` root tics of uniformly random OPEN moves (the obstacle mask of k_observe), then every finished game is dealt
    afresh: all n games are live, most of them mid-game with 2-4 snakes, lengths and healths spread out"""
    g = torch.Generator(device="cuda").manual_seed(seed)
    S = eng.S
    sub = torch.arange(n, dtype=torch.int32, device="cuda").repeat_interleave(S)
    pairs = torch.stack([sub, torch.arange(S, dtype=torch.int32, device="cuda").repeat(n)], dim=1).contiguous()
    blocked = torch.empty((S * n, 3), dtype=torch.uint8, device="cuda")
    for _ in range(ticks):
        eng.observe(pairs, S * n, None, blocked, None)
        r = torch.rand((S * n, 3), device="cuda", generator=g) - 2.0 * blocked.float()
        mv = torch.where(blocked.bool().all(dim=1), torch.ones((), dtype=torch.int64, device="cuda"), r.argmax(dim=1))
        eng.step(mv.to(torch.uint8).reshape(n, S).contiguous())
    over = torch.nonzero(eng.alive().sum(dim=1) <= 1).reshape(-1).to(torch.int32).contiguous()
    if over.numel():
        eng.reset(slots=over)
    return int(over.numel())


def _counters_of(states):
    return np.array([st.counters[:] for st in states], np.int64)


@pytest.mark.parametrize("board,S,games,breadth,tt_cap,stride",
                         [(11, 4, 4096, 50, 1 << 20, 5), (11, 4, 32768, 200, 1 << 24, 40), (19, 8, 4096, 16, 1 << 22, 10)],
                         ids=["configs1_4096x50", "configs2_32768x200", "configs4_shape_19x19x8_4096x16"])
def test_one_root_turn_at_full_size(oracle, board, S, games, breadth, tt_cap, stride):
    import torch
    import snake_engine as se
    from snake_engine.engine import compact_from_state
    from oracle.obs_key import obstacle_mask
    from stubnet_device import DeviceStubNNet
    from utils.agent import Agent
    from utils.mp_game_runner import MPGameRunner
    t_start = time.time()
    H = W = board
    old = MPGameRunner.verbose, MPGameRunner.init
    MPGameRunner.verbose, MPGameRunner.init = False, "device"
    try:
        gr = MPGameRunner(H, W, S, 1, games, seed=4242 + games)
    finally:
        MPGameRunner.verbose, MPGameRunner.init = old[0], old[1]
    eng = gr.engine
    redealt = _mid_game_boards(torch, eng, games, 24, seed=games)
    alive0 = eng.alive().cpu().numpy().astype(bool)
    assert (alive0.sum(axis=1) >= 2).all() and redealt < games // 2 and (redealt > 0 or board > 11)
    n_alive0 = alive0.sum(axis=1)
    assert len({int(v) for v in n_alive0}) >= 3, "boards with different numbers of snakes: several rollout depths 8 - 2 (alive - 2), agent.py:45"
    n_ticks = max(1, 8 - 2 * (int(n_alive0.min()) - 2))
    counters0 = np.array(eng.sum_counters(), np.int64)
    sample = np.arange(0, games, stride, dtype=np.int32)
    pre = eng.export(sample)
    row0 = np.concatenate([[0], np.cumsum(alive0.sum(axis=1))])          # first row of game g in ids order (all games are live)
    n_rows = int(row0[-1])

    alice = Agent(DeviceStubNNet(), 2, True, 8, breadth, seed=99, tt_capacity=tt_cap)
    seen = []
    make_moves = alice.make_moves

    def spy(games_, ids):
        out = make_moves(games_, ids)
        seen.append((ids, out))
        return out
    alice.make_moves = spy
    verbose = MPGameRunner.verbose
    MPGameRunner.verbose = False
    try:
        gr.run(alice, max_turns=1)
    finally:
        MPGameRunner.verbose = verbose
    torch.cuda.synchronize()
    t_turn = time.time() - t_start
    mcts = alice._mcts

    # ---- globally ------------------------------------------------------------------------------------------------------------
    assert gr.env_steps == games and len(seen) == 1
    ids, moves = seen[0]
    assert len(ids) == n_rows == len(moves) == len(alice.records) == len(alice.values)
    cap, occ, ovf = mcts.tt.status()
    assert not ovf
    assert mcts.stats["net_evals"] == occ, "every entry the table created was evaluated exactly once (agent.py:177-201)"
    if board == 11:
        assert mcts.tt.generation >= 1 and cap > tt_cap, "the table outgrew its initial capacity between epochs"
    epochs, B = breadth // 8, games * 8
    assert mcts.stats["rollout_ticks"] == epochs * n_ticks               # the deepest rollout is that of the board with the fewest snakes
    assert epochs * B <= mcts.stats["sim_steps"] <= epochs * B * n_ticks
    assert occ >= n_rows // 2 and mcts.stats["net_evals"] >= epochs * games
    post_all = eng.export()
    c_all = _counters_of(post_all)
    assert np.array_equal(np.array(eng.sum_counters(), np.int64), c_all.sum(axis=0)), "snk_engine_sum_counters_sync vs the exported boards"
    assert np.array_equal(c_all[:, 5].sum() - counters0[5], games), "every live game made one tic (game_length)"
    over = np.array([sum(st.alive[:S]) <= 1 for st in post_all])
    assert over.sum() == games - len(gr.games) and (over.sum() > 0 or board > 11)
    from utils.mp_game_runner import LOG_FIELDS
    assert np.array_equal(np.array([gr._totals[k] for k in LOG_FIELDS], np.int64), c_all[over].sum(axis=0)), \
        "the runner's log counters are those of the games that ended (mp_game_runner.py:54-60)"
    V_all = np.asarray(alice.values[:], np.float32)
    assert np.isfinite(V_all).all() and (np.abs(V_all) <= 1.0).all()
    mv_all = np.asarray(moves)
    assert ((mv_all >= 0) & (mv_all <= 2)).all()
    assert min(np.bincount(mv_all, minlength=3)) > n_rows // 20, "training=True samples from softermax(Q): all three moves occur"

    # ---- the strided sample against the oracle -----------------------------------------------------------------------------------
    rec_idx = np.concatenate([np.arange(row0[g], row0[g + 1]) for g in sample])
    assert len(rec_idx) >= 1500
    states = alice.records.fetch(rec_idx)
    post = eng.export(sample)
    k = n_spawn = n_live_ticks = 0
    sample_counters = np.zeros(6, np.int64)
    for j, g in enumerate(sample):
        c0 = compact_from_state(pre[j])
        og = oracle.Game.from_compact(H, W, S, 1, 0.15, c0)
        dense = np.ones(S, np.uint8)
        for s in np.flatnonzero(alive0[g]):
            want = og.make_state(int(s))
            assert states[k].tobytes() == want.tobytes(), f"game {g} snake {s}: recorded state != Game.make_state"
            assert ids[rec_idx[k]] == (int(g), int(s))
            blocked = obstacle_mask(want)[0]
            mv, v = int(mv_all[rec_idx[k]]), V_all[rec_idx[k]]
            dense[s] = mv
            if not blocked.all():
                assert not blocked[mv], f"game {g} snake {s}: chose a blocked move {mv} ({blocked})"
                assert (v[blocked] == -1.0).all() and (v[~blocked] > -1.0).all(), (g, s, v, blocked)
            k += 1
        # the root tick on the oracle: first without a spawn to learn which cell (if any) the device's Philox draw chose
        probe = oracle.Game.from_compact(H, W, S, 1, 0.15, c0)
        probe.tic(dense, spawn_cell=-1)
        c1 = compact_from_state(post[j])
        extra = np.flatnonzero(c1["food"].astype(int) - probe.compact()["food"].astype(int))
        assert len(extra) <= 1 and (len(extra) == 0 or c1["food"][extra[0]] == 1), f"game {g}: food differs by {extra}"
        cell = int(extra[0]) if len(extra) else -1
        og.tic(dense, spawn_cell=cell, want_empty=True)
        if cell >= 0:
            assert og.last_empty[cell] == 1, f"game {g}: food spawned on an occupied cell"
        elif probe.compact()["food"].sum() == 0:
            assert og.last_empty.sum() == 0, f"game {g}: no food left and none spawned (game.py:130)"
        want = og.compact()
        for f in FIELDS:
            assert np.array_equal(c1[f], want[f]), f"game {g}: {f} after the root tick"
        sample_counters += want["counters"]
        n_spawn += cell >= 0
        n_live_ticks += 1
    assert k == len(rec_idx)
    assert np.array_equal(np.array(eng.sum_counters(slots=sample), np.int64), sample_counters)
    assert 0.08 < n_spawn / n_live_ticks < 0.25, n_spawn / n_live_ticks          # food_spawn_chance 0.15 (+ forced spawns)
    print(f"\n{games} games x breadth {breadth}: root turn in {t_turn:.1f} s incl. set-up, {mcts.stats['net_evals']} evaluations, "
          f"{mcts.stats['sim_steps']} rollout tics, table {tt_cap} -> {cap} ({mcts.tt.generation} rebuilds), "
          f"{len(rec_idx)} sampled rows / {len(sample)} games checked in {time.time() - t_start - t_turn:.1f} s")
