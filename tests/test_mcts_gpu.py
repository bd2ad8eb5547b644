"""GPU tests of the device MCTS (csrc/mcts.hip + snake_engine/mcts.py) and of the drop-in classes
(utils.agent.Agent, utils.mp_game_runner.MPGameRunner) against runs recorded from the reference and
against the CPU oracle.  Boards, masks, moves, rewards, counters, cache de-duplication counts: exact.
MCTS statistics (float32 sums whose libm / summation order differs from NumPy's): |dQ| <= 1e-5 in the
sequential parity mode; the production mode (atomics) is compared with the sequential mode on >= 2 000 root states (same
draws: equal to rounding; independent draws: no bias, same move frequencies) and with the C oracle over whole games."""
import ctypes as C
import hashlib

import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch
    assert torch.cuda.is_available()
    import snake_engine
    return torch, snake_engine


def stub_evaluate(planes, mask):
    """deterministic stub net (oracle/obs_key.py: stub_q) -- test infrastructure, evaluated on the host"""
    import torch
    from oracle.obs_key import stub_q
    q = stub_q(planes.cpu().numpy())
    assert np.array_equal(q == -1.0, mask.cpu().numpy().astype(bool)), "device obstacle mask != alpha_nnet.py:63-76"
    return torch.as_tensor(q, device=planes.device)


class StubNNet:
    def v_device(self, planes, mask):
        return stub_evaluate(planes, mask)


def _golden_engine(se, z, chance=0.15):
    from snake_engine.engine import state_from_compact
    H, W, S, hd, n = int(z["H"]), int(z["W"]), int(z["S"]), int(z["hd"]), int(z["n_games"])
    eng = se.Engine(n, H, W, S, hd, chance)
    sts = []
    for g in range(n):
        st = {k: z["init_" + k][g] for k in ("alive", "health", "length", "dir", "nodes", "food", "rewards", "counters")}
        sts.append(state_from_compact(H, W, S, st))
    eng.import_states(sts)
    return eng, (H, W, S, hd, n)


@pytest.mark.parametrize("tag", ["tiny", "tiny_greedybase", "7x7x2", "19x19x8", "9x9x3"])
def test_sequential_mode_replays_the_reference_run(env, tag):
    torch, se = env
    from snake_engine.mcts import DeviceMCTS
    z = load_golden(f"mcts_{tag}.npz")
    eng, (H, W, S, hd, n) = _golden_engine(se, z)
    mcts = DeviceMCTS(stub_evaluate, H, W, S, int(z["base"]), True, int(z["depth"]), int(z["breadth"]),
                      sequential=True, tape_u=z["tape_u"], tt_capacity=1 << 16)
    live = np.arange(n, dtype=np.int32)
    rec_digest = []
    for t in range(int(z["n_turns"])):
        d_slots = torch.as_tensor(live, device="cuda")
        alive = eng.alive(slots=d_slots)
        alive_h = alive.cpu().numpy().astype(bool)
        evals0 = mcts.stats["net_evals"]
        V, moves = mcts.search(eng, d_slots, alive)
        gi, si = np.nonzero(alive_h)
        ids = [[int(live[g]), int(s)] for g, s in zip(gi, si)]
        assert ids == z[f"t{t}_ids"].tolist(), f"turn {t}: ids"
        Vh = V.cpu().numpy()[gi, si]
        assert np.abs(Vh - z[f"t{t}_V"]).max() <= 1e-5, (t, np.abs(Vh - z[f"t{t}_V"]).max())
        assert moves.cpu().numpy()[gi, si].tolist() == z[f"t{t}_moves"].tolist(), f"turn {t}: moves"
        assert mcts.stats["net_evals"] - evals0 == z["turn_evals"][t], f"turn {t}: net evaluations (cache de-duplication)"
        assert mcts.tape_pos == z["turn_tape_pos"][t], f"turn {t}: draws consumed"
        pairs = np.stack([live[gi], si], axis=1).astype(np.int32)
        planes, _, _ = eng.observe_all(pairs, want_mask=False, want_key=False)
        rec_digest += [hashlib.blake2b(p.tobytes(), digest_size=16).digest() for p in planes.cpu().numpy()]
        mcts.end_of_turn()
        done = eng.new((len(live),), torch.uint8, 0)
        tape = torch.as_tensor(z["turn_spawn"][t][live].astype(np.int16), device="cuda")
        eng.step(moves.contiguous(), slots=d_slots, spawn_tape=tape, done=done)
        live = live[~done.cpu().numpy().astype(bool)]
    assert rec_digest == [d.tobytes() for d in z["records_digest"]]
    assert mcts.tape_pos == len(z["tape_u"])


def test_device_stub_net_equals_the_oracle_stub(env):
    """tests/stubnet_device.py (torch ops on the device) == oracle/obs_key.py::stub_q on real observations"""
    torch, se = env
    from oracle.obs_key import stub_q
    from stubnet_device import stub_q_device
    eng = se.Engine(64, 11, 11, 4, 1, 0.15, seed=3)
    eng.reset()
    g = torch.Generator(device="cuda").manual_seed(1)
    for _ in range(6):
        eng.step(torch.randint(0, 3, (64, 4), device="cuda", generator=g, dtype=torch.int32).to(torch.uint8))
    pairs = torch.nonzero(eng.alive()).to(torch.int32).contiguous()
    planes, mask, _ = eng.observe_all(pairs, want_key=False)
    assert np.array_equal(stub_q_device(planes, mask).cpu().numpy(), stub_q(planes.cpu().numpy()))


def _paired_search(se, torch, n_games, turns, breadth, mk_a, mk_b, seed):
    """two DeviceMCTS instances search the SAME root states turn after turn (the games advance with A's moves);
    returns the root Q values and root moves of both for every alive root snake"""
    eng = se.Engine(n_games, 11, 11, 4, 3, 0.15, seed=seed)
    eng.reset()
    a, b = mk_a(), mk_b()
    live = np.arange(n_games, dtype=np.int32)
    VA, VB, MA, MB = [], [], [], []
    for _ in range(turns):
        d_slots = torch.as_tensor(live, device="cuda")
        alive = eng.alive(slots=d_slots)
        va, ma = a.search(eng, d_slots, alive)
        vb, mb = b.search(eng, d_slots, alive)
        if a.tape is not None and b.tape is not None:
            b.tape_pos = a.tape_pos              # same draws next turn even if one rollout diverged
        al = alive.bool()
        VA.append(va[al].cpu().numpy()); VB.append(vb[al].cpu().numpy())
        MA.append(ma[al].cpu().numpy()); MB.append(mb[al].cpu().numpy())
        a.end_of_turn(); b.end_of_turn()
        done = eng.new((len(live),), torch.uint8, 0)
        eng.step(ma.contiguous(), slots=d_slots, done=done)
        live = live[~done.cpu().numpy().astype(bool)]
    return np.concatenate(VA), np.concatenate(VB), np.concatenate(MA), np.concatenate(MB)


def test_production_back_up_equals_the_sequential_one_on_2000_root_states(env):
    """The mode bench.py times (one thread per row, float atomics, est from the statistics as they stand before the
    tick's back-ups) against the reference's order (agent.py:208-220: ids order, live re-reads), same taped uniforms,
    same root states: >= 2 000 root rows over 4 root turns of 512 games.  The two differ only in float32 summation order
    unless a uniform falls within rounding distance of a cdf edge (one rollout of a root then takes another path)."""
    torch, se = env
    from snake_engine.mcts import DeviceMCTS
    from stubnet_device import stub_q_device
    tape = np.random.RandomState(42).random_sample(6_000_000)

    def mk(seq):
        return lambda: DeviceMCTS(stub_q_device, 11, 11, 4, 2, True, 8, 16, sequential=seq, tape_u=tape)
    VA, VB, MA, MB = _paired_search(se, torch, 512, 4, 16, mk(True), mk(False), seed=21)
    d = np.abs(VA - VB)
    n = len(VA)
    tight = float((d.max(axis=1) <= 1e-5).mean())
    print(f"\nproduction vs sequential back-up on {n} root rows: mean |dQ| {d.mean():.3e}, max |dQ| {d.max():.3e}, "
          f"rows within 1e-5: {100 * tight:.2f} %, same root move: {100 * float((MA == MB).mean()):.2f} %")
    assert n >= 2000
    assert tight >= 0.995, tight                  # all but the rare diverged rollouts agree to rounding
    assert d.mean() <= 2e-5, d.mean()
    assert float((MA == MB).mean()) >= 0.995
    # no bias: the paired differences average to zero within 3 standard errors
    diff = (VA - VB)[VA > -1.0]
    assert abs(diff.mean()) <= 3 * diff.std() / np.sqrt(len(diff)) + 1e-7, (diff.mean(), diff.std())


def test_production_draws_give_the_same_root_statistics_as_the_sequential_mode(env):
    """independent random streams (Philox in the production mode, a tape in the sequential one) on the same root states:
    root Q values agree on average within 3 standard errors (a biased back-up would shift them) and the chosen root moves
    have the same frequencies (chi-square, 2 degrees of freedom, p > 0.001)"""
    torch, se = env
    from snake_engine.mcts import DeviceMCTS
    from stubnet_device import stub_q_device
    tape = np.random.RandomState(7).random_sample(6_000_000)
    VA, VB, MA, MB = _paired_search(
        se, torch, 512, 4, 16,
        lambda: DeviceMCTS(stub_q_device, 11, 11, 4, 2, True, 8, 16, sequential=True, tape_u=tape),
        lambda: DeviceMCTS(stub_q_device, 11, 11, 4, 2, True, 8, 16, sequential=False, seed=99), seed=22)
    legal = VA > -1.0
    diff = (VA - VB)[legal]
    se_ = diff.std() / np.sqrt(len(diff))
    fa, fb = np.bincount(MA, minlength=3).astype(float), np.bincount(MB, minlength=3).astype(float)
    tot = fa + fb
    chi2 = float((((fa - tot / 2) ** 2) / (tot / 2) + ((fb - tot / 2) ** 2) / (tot / 2)).sum())
    print(f"\nindependent draws, {len(VA)} root rows: mean dQ {diff.mean():+.2e} (s.e. {se_:.2e}), move counts {fa} vs {fb}, chi2 {chi2:.2f}")
    assert len(VA) >= 2000
    assert abs(diff.mean()) <= 3 * se_, (diff.mean(), se_)
    assert chi2 < 13.82, chi2                     # chi-square(2) at p = 0.001


def test_transposition_table_semantics(env):
    torch, se = env
    from snake_engine._lib import lib, check
    from snake_engine.mcts import TranspositionTable
    L = lib()
    st = torch.cuda.current_stream().cuda_stream
    tt = TranspositionTable(1 << 12)
    rng = np.random.RandomState(0)
    base = rng.randint(1, 1 << 62, size=(700, 2)).astype(np.int64)
    rows = base[rng.randint(0, 700, size=3000)]                 # many duplicates inside one batch
    rows[5] = 0                                                  # dead-snake key
    active = np.ones(3000, np.uint8); active[7] = 0
    key = torch.as_tensor(rows, device="cuda"); act = torch.as_tensor(active, device="cuda")
    entry = torch.empty(3000, dtype=torch.int32, device="cuda"); new = torch.empty(3000, dtype=torch.uint8, device="cuda")

    def lookup(now):
        check(L.snk_tt_lookup_insert(tt.h, key.data_ptr(), act.data_ptr(), 3000, now, 8, entry.data_ptr(), new.data_ptr(), st))
        return entry.cpu().numpy(), new.cpu().numpy()
    e, nw = lookup(1)
    assert e[5] == -1 and e[7] == -1 and nw[5] == 0 and nw[7] == 0
    valid = np.ones(3000, bool); valid[[5, 7]] = False
    distinct = {tuple(r) for r in rows[valid]}
    assert nw.sum() == len(distinct), "exactly one evaluation per distinct new key (agent.py:177-184)"
    by_key = {}
    for i in np.flatnonzero(valid):
        by_key.setdefault(tuple(rows[i]), set()).add(int(e[i]))
    assert all(len(v) == 1 for v in by_key.values()), "duplicates share one entry"
    assert len({next(iter(v)) for v in by_key.values()}) == len(distinct), "distinct keys, distinct entries"
    assert tt.status()[1] == len(distinct)
    e2, nw2 = lookup(2)
    assert nw2.sum() == 0 and np.array_equal(e2, e), "second sight: all hits"
    q = torch.arange(3000 * 3, dtype=torch.float32, device="cuda").reshape(3000, 3)
    check(L.snk_tt_set_priors(tt.h, entry.data_ptr(), None, 3000, q.data_ptr(), None, st))
    # still present at now = touch + max_age + 1, evicted (a miss, re-created in place) one turn later
    e3, nw3 = lookup(2 + 9)
    assert nw3.sum() == 0
    e4, nw4 = lookup(2 + 9 + 10)
    assert nw4.sum() == len(distinct) and np.array_equal(e4, e)
    # physical eviction keeps only entries with now - touch <= max_age
    key2 = torch.as_tensor(rng.randint(1, 1 << 62, size=(100, 2)).astype(np.int64), device="cuda")
    ent2 = torch.empty(100, dtype=torch.int32, device="cuda"); new2 = torch.empty(100, dtype=torch.uint8, device="cuda")
    check(L.snk_tt_lookup_insert(tt.h, key2.data_ptr(), None, 100, 40, 8, ent2.data_ptr(), new2.data_ptr(), st))
    check(L.snk_tt_set_priors(tt.h, ent2.data_ptr(), None, 100, q.data_ptr(), None, st))
    assert tt.status()[1] == len(distinct) + 100
    tt.rebuild(1 << 12, 40, 8)
    assert tt.status()[1] == 100
    check(L.snk_tt_lookup_insert(tt.h, key2.data_ptr(), None, 100, 41, 8, ent2.data_ptr(), new2.data_ptr(), st))
    assert new2.sum().item() == 0
    out = torch.empty((100, 3), dtype=torch.float32, device="cuda")
    check(L.snk_tt_read_q(tt.h, ent2.data_ptr(), 1, 100, out.data_ptr(), st))
    assert torch.equal(out, q[:100]), "statistics survive the rebuild"


def test_softermax_argmax_tables(env):
    torch, se = env
    from utils.agent import Agent
    z = load_golden("tables.npz")
    for base in (2, 3, 10, 100):
        ag = Agent(None, base)
        pmf, _ = ag._soft_arg(z["z"])
        ref = z[f"pmf_b{base}"]
        assert np.array_equal(pmf == 0, ref == 0)
        assert np.abs(pmf - ref).max() <= 2e-6, np.abs(pmf - ref).max()
    ag = Agent(None)
    assert ag.argmaxs(list(z["argmax_z"])) == z["argmax"].tolist()
    assert np.abs(ag.softermax(z["z"][17]) - z["pmf_b100"][17]).max() <= 2e-6


class TapedMoves:
    """an agent with the reference's make_moves(games, ids) contract that replays recorded moves"""

    def __init__(self, moves):
        self.moves, self.t = moves, 0

    def make_moves(self, games, ids):
        out = [int(self.moves[self.t][gid][sid]) for (gid, sid) in ids]
        assert all(m in (0, 1, 2) for m in out)
        self.t += 1
        return out


def test_mp_game_runner_matches_reference_run(env):
    torch, se = env
    from utils.mp_game_runner import MPGameRunner
    from snake_engine.engine import state_from_compact
    z = load_golden("runner.npz")
    H, W, S, hd, n = int(z["H"]), int(z["W"]), int(z["S"]), int(z["hd"]), int(z["n_games"])
    MPGameRunner.verbose = False
    gr = MPGameRunner(H, W, S, hd, n)
    sts = []
    for g in range(n):
        st = {k: z["init_" + k][g] for k in ("alive", "health", "length", "dir", "nodes", "food", "rewards", "counters")}
        sts.append(state_from_compact(H, W, S, st))
    gr.engine.import_states(sts)
    for g in gr.games.values():
        g._dirty()
    rewards = gr.run(TapedMoves(z["moves"]), spawn_tape=lambda turn: z["spawn"][turn - 1])
    assert np.array_equal(np.array(rewards, np.float32), z["rewards"])
    got = [gr.wall_collision, gr.body_collision, gr.head_collision, gr.starvation, gr.food_eaten, gr.game_length]
    assert got == z["counters"].tolist()
    assert len(gr.games) == 0 and gr.turns == len(z["moves"])


def test_self_play_end_to_end_production_mode(env):
    """Agent + MPGameRunner exactly as the trainer drives them (trainer.py:52-54, 63-75), stub net, device RNG"""
    torch, se = env
    from utils.agent import Agent
    from utils.mp_game_runner import MPGameRunner
    import random
    random.seed(3); np.random.seed(3)
    MPGameRunner.verbose = False
    alice = Agent(StubNNet(), 2, True, 8, 16, seed=5)
    gr = MPGameRunner(11, 11, 4, 9, 24, seed=7)
    rewards = gr.run(alice)
    assert len(rewards) == 24 and all(r is not None for r in rewards)
    for r in rewards:
        assert sorted(set(r)) in ([-1.0], [-1.0, 1.0]) and r.count(1.0) <= 1
    assert gr.game_length > 3 and len(alice.records) == len(alice.values) > 24 * 4
    x = alice.records[0]
    assert x.shape == (21, 21, 3) and x.dtype == np.float32 and x[10, 10].tolist() == [-1.0, -1.0, -1.0]
    v = np.array(alice.values[:50])
    assert v.shape == (50, 3) and np.isfinite(v).all() and (np.abs(v) <= 1).all()
    batch = alice.records.fetch(range(0, len(alice.records), 7))
    assert (batch[:, 10, 10, :] == -1).all()
    deaths = (gr.wall_collision + gr.body_collision + gr.head_collision + gr.starvation) * 24
    assert deaths == sum(r.count(-1.0) for r in rewards)
    alice.clear()
    assert len(alice.records) == 0 and len(alice.cached_values) == 0


def test_game_view_api(env):
    """utils.game.Game standalone: the reference's per-object API over a 1-slot engine"""
    torch, se = env
    import random
    from utils.game import Game
    random.seed(11)
    g = Game(0, 11, 11, 4, 1, 0.15)
    assert len(g.snakes) == 4 and g.get_ids() == [(0, 0), (0, 1), (0, 2), (0, 3)]
    sts = g.get_states()
    assert len(sts) == 4 and sts[0].shape == (21, 21, 3)
    s0 = g.snakes[0]
    assert np.array_equal(g.make_state(s0, g.last_moves[0]), sts[0])
    assert np.array_equal(g.make_state(s0, (g.last_moves[0] + 1) % 4), np.rot90(sts[0], 1))
    sub = g.subgame(5)
    assert sub.food_spawn_chance == 0.0 and sub.food == g.food and sub.id == 5
    res = g.tic([1, 1, 1, 1])
    assert res == 0 or isinstance(res, list)
    assert g.game_length == 1 and sub.game_length == 0
    assert len(g.empty_positions) + len(g.heads) + len(g.bodies - set(g.heads)) + len(g.food) == 121


def test_mcts_agent_and_mcts_runner_api(env):
    """the reference's own epoch body (agent.py:39-72) written against MCTSAgent + MCTSMPGameRunner"""
    torch, se = env
    import random
    from utils.agent import Agent, MCTSAgent
    from utils.game import Game
    from utils.mp_game_runner import MPGameRunner, MCTSMPGameRunner, GameDict
    random.seed(2); np.random.seed(2)
    MPGameRunner.verbose = False
    alice = Agent(StubNNet(), 2, True, 4, 8, seed=1)
    gr = MPGameRunner(11, 11, 4, 1, 3, seed=2)
    games = gr.games
    parallel = 8
    sub_eng = se.Engine(len(games) * parallel, 11, 11, 4, 1, 0.0)
    gr.engine.clone_to(sub_eng, src_slots=games.live_slots(), fanout=parallel)
    subgames = GameDict(sub_eng, {i: Game(i, 11, 11, 4, 1, 0.0, _engine=sub_eng, _slot=i) for i in range(len(games) * parallel)})
    depth = {i: 4 for i in subgames}
    n_sub = len(subgames)
    MCTSAlice = MCTSAgent(alice.nnet, alice.softmax_base, subgames, alice.cached_values, alice.total_rewards,
                          alice.visit_cnts, alice.cache_hit)
    rewards = MCTSMPGameRunner(subgames).run(MCTSAlice, depth)
    assert len(rewards) == n_sub and len(subgames) == 0
    for sid in range(n_sub):
        assert len(rewards[sid]) == 4 and all(r in (None, 1.0, -1.0) for r in rewards[sid])
        for snake in range(4):
            k, mv = MCTSAlice.keys[sid][snake], MCTSAlice.moves[sid][snake]
            assert 1 <= len(k) == len(mv) <= 4 and all(len(b) == 16 for b in k) and all(x in (0, 1, 2) for x in mv)
    # all 8 clones of a root game start from the same observation -> same first key
    for g in range(3):
        assert len({MCTSAlice.keys[g * parallel + r][0][0] for r in range(parallel)}) == 1


def test_self_play_19x19_8_snakes(env):
    """config-5 geometry (19x19, 8 snakes: rollout depth 8 - 2*(8-2) = -4 -> single-tick rollouts, agent.py:45)"""
    torch, se = env
    import random
    from utils.agent import Agent
    from utils.mp_game_runner import MPGameRunner
    random.seed(4); np.random.seed(4)
    MPGameRunner.verbose = False
    alice = Agent(StubNNet(), 2, True, 8, 24, seed=6)
    gr = MPGameRunner(19, 19, 8, 1, 5, seed=8)
    gr.run(alice, max_turns=4)
    assert gr.turns == 4 and gr.env_steps == 20
    assert len(alice.records) >= 4 * 5 * 6 and alice.records[0].shape == (37, 37, 3)
    assert alice._mcts.stats["rollout_ticks"] >= 4 * 3      # 3 epochs x 1 tick per root turn while 8 snakes live


def test_transposition_table_grows_between_epochs(env):
    """a deliberately tiny table has to be re-hashed / doubled between epochs and at the end of the turn"""
    torch, se = env
    import random
    from utils.agent import Agent
    from utils.mp_game_runner import MPGameRunner
    random.seed(5); np.random.seed(5)
    MPGameRunner.verbose = False
    alice = Agent(StubNNet(), 2, True, 8, 24, seed=7, tt_capacity=1 << 10)
    gr = MPGameRunner(11, 11, 4, 1, 6, seed=9)
    gr.run(alice, max_turns=3)
    cap, occ, ovf = alice._mcts.tt.status()
    assert not ovf and cap > (1 << 10) and occ * 2 <= cap and occ > 512
    assert len(alice.cached_values) == occ


def test_production_mode_is_distribution_equivalent_to_the_oracle(env, oracle):
    """Self-play to completion with the deterministic stub net: the production device MCTS (float atomics, Philox draws,
    device food spawns) against the sequential C restatement of the reference (oracle/mcts_cpu.c, pinned to the recorded
    reference runs; its own RNG).  Different random streams, so the comparison is statistical: 480 oracle games and 960
    device games, per-game averages of game length, food eaten and the four death causes within 3 combined standard
    errors -- no absolute slack."""
    torch, se = env
    from oracle.mcts_cpu import CpuSelfPlay, seeded_games
    from stubnet_device import DeviceStubNNet
    from utils.agent import Agent
    from utils.mp_game_runner import MPGameRunner
    from snake_engine.engine import compact_from_state
    n_cpu, n_gpu = 480, 960
    sp = CpuSelfPlay(seeded_games(n_cpu, health_dec=3, seed=5), net=None, threads=8, base=2, training=True, max_depth=8,
                     max_breadth=16, seed=3, keep_records=False)
    sp.run()
    cpu = np.array([[w.game(i).g.counters[k] for k in range(6)] for w in sp.workers for i in range(w.n)], float)
    sp.close()
    assert cpu.shape == (n_cpu, 6)
    MPGameRunner.verbose = False
    old_init = MPGameRunner.init
    MPGameRunner.init = "device"
    try:
        alice = Agent(DeviceStubNNet(), 2, True, 8, 16, seed=12)
        gr = MPGameRunner(11, 11, 4, 3, n_gpu, seed=13)
        gr.run(alice)
    finally:
        MPGameRunner.init = old_init
    gpu = np.array([compact_from_state(s)["counters"] for s in gr.engine.export(np.arange(n_gpu, dtype=np.int32))], float)
    rows = []
    for k, name in enumerate(["wall", "body", "head", "starvation", "food_eaten", "game_length"]):
        mc, mg = cpu[:, k].mean(), gpu[:, k].mean()
        se_ = np.sqrt(cpu[:, k].var() / n_cpu + gpu[:, k].var() / n_gpu)
        rows.append((name, mc, mg, se_))
    print("\n" + "\n".join(f"{n:12s} oracle {a:8.3f}  device {b:8.3f}  s.e. {c:.3f}  z {(a - b) / max(c, 1e-12):+.2f}" for n, a, b, c in rows))
    for name, mc, mg, se_ in rows:
        assert abs(mc - mg) <= 3 * se_, f"{name}: oracle {mc:.3f} vs device {mg:.3f} (combined s.e. {se_:.3f})"


def test_agent_accepts_a_net_with_only_the_reference_v_method(env):
    """Agent(nnet) where nnet is any object with the reference's nnet.v(list_of_states) -> (N,3) (alpha_nnet.py:61)"""
    torch, se = env
    from oracle.obs_key import stub_q
    from utils.agent import Agent
    from utils.mp_game_runner import MPGameRunner

    class VOnly:
        calls = 0

        def v(self, X):
            assert isinstance(X, list) and X[0].shape == (21, 21, 3) and X[0].dtype == np.float32
            VOnly.calls += 1
            return stub_q(np.array(X))
    MPGameRunner.verbose = False
    alice = Agent(VOnly(), 10, False, 4, 8, seed=3)
    gr = MPGameRunner(11, 11, 4, 1, 2, seed=4)
    gr.run(alice, max_turns=2)
    assert VOnly.calls > 0 and gr.env_steps == 4 and not hasattr(alice, "records")


def test_cache_views_are_dict_like(env):
    """agent.cached_values / total_rewards / visit_cnts / cache_hit (agent.py:16-19) answer `in`, [key], get() with the
    reference's keys: the observation's bytes (agent.py:175)"""
    torch, se = env
    from stubnet_device import DeviceStubNNet
    from utils.agent import Agent
    from utils.mp_game_runner import MPGameRunner
    MPGameRunner.verbose = False
    alice = Agent(DeviceStubNNet(), 2, True, 8, 16, seed=4)
    assert len(alice.cached_values) == 0 and b"x" * 5292 not in alice.cached_values
    gr = MPGameRunner(11, 11, 4, 1, 3, seed=6)
    gr.run(alice, max_turns=1)
    n = len(alice.records)
    assert n == 12 and len(alice.cached_values) > 12
    for i in range(n):
        key = alice.records[i].tobytes()
        assert key in alice.cached_values and alice.records[i] in alice.visit_cnts
        assert np.array_equal(alice.cached_values[key], alice.values[i])             # root Q of this turn
        tot, vis = alice.total_rewards[key], alice.visit_cnts[key]
        assert vis.dtype == np.float32 and (vis >= 1).all() and np.allclose(tot / vis, alice.values[i], atol=1e-6)
        assert alice.cache_hit[key] == 0                                             # touched this root turn
    unseen = np.zeros((21, 21, 3), np.float32)
    assert unseen.tobytes() not in alice.cached_values and alice.cached_values.get(unseen.tobytes()) is None
    with pytest.raises(KeyError):
        alice.cached_values[unseen.tobytes()]
    with pytest.raises(TypeError, match="digests"):
        iter(alice.cached_values)
    alice.clear()
    assert len(alice.cached_values) == 0 and alice.records is not None


def test_sequential_mode_tracks_the_c_restatement_on_unrecorded_configurations():
    """beyond the five recorded reference runs: 40 random configurations (7x7 .. 13x13, 2-4 snakes, three health decrements,
    breadth 8-24, depth 4-8, four softmax bases, fresh uniform tapes) played on the device in sequential mode and on
    oracle/mcts_cpu.c (itself pinned to the reference's recorded runs) with the same draws: ids, moves, net-evaluation counts,
    draws consumed and final boards identical turn by turn, root Q within 1e-5 (tools/fuzz_mcts.py; 600 more seeds:
    profiles/r3_fuzz_mcts.log)"""
    import os
    import re
    import subprocess
    import sys
    from conftest import REPO
    r = subprocess.run([sys.executable, os.path.join(REPO, "tools", "fuzz_mcts.py"), "1000", "40"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    m = re.search(r"fuzz done: (\d+) of 40 runs identical .* worst root \|dQ\| ([0-9.e+-]+)", r.stdout)
    assert m, r.stdout[-1000:]
    assert int(m.group(1)) >= 39, r.stdout[-1500:]             # a uniform within 1e-6 of a cdf edge may part the two libms: at most one run
    assert float(m.group(2)) <= 1e-5
