"""differential fuzz of the device MCTS in its sequential parity mode against the C restatement of the reference's loop
(oracle/mcts_cpu.c, itself pinned to the recorded reference runs) on start boards, settings and draws nobody recorded
(development aid): fuzz_mcts.py <first seed> <n seeds>
Per seed: random board (7 / 9 / 11 / 13), 2-4 snakes, health decrement, 2-5 games, breadth 8-24, depth 4-8, softmax base; both
sides get the same uniform tape and the same stub net; the device plays (its own food spawns), the C side replays the turn with
the device's spawn cells.  Every turn: ids, moves, net-evaluation counts, draws consumed equal; root Q within 1e-5.
A uniform that falls within 1e-6 of a cdf edge may send the two libms' pmfs different ways: such seeds are reported, not failed."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "alphasnake-zero_amd"), os.path.join(REPO, "tests")]
import numpy as np, torch
import snake_engine as se
from snake_engine.engine import compact_from_state
from snake_engine.mcts import DeviceMCTS
from oracle import snake_oracle as oracle
from oracle.mcts_cpu import CpuSelfPlay
from stubnet_device import stub_q_device
first, count = int(sys.argv[1]), int(sys.argv[2])
ok = diverged = turns_total = 0
worst = 0.0
t0 = time.time()
for seed in range(first, first + count):
    rng = np.random.RandomState(5000 + seed)
    H = int(rng.choice([7, 9, 11, 13])); S = int(rng.randint(2, 5)); hd = int(rng.choice([1, 3, 9]))
    n = int(rng.randint(2, 6)); breadth = int(rng.choice([8, 16, 24])); depth = int(rng.choice([4, 6, 8])); base = int(rng.choice([2, 3, 10, 100]))
    eng = se.Engine(n, H, H, S, hd, 0.15, seed=int(rng.randint(1 << 30)))
    eng.reset()
    start = eng.export()
    games = [oracle.Game.from_compact(H, H, S, hd, 0.15, compact_from_state(start[g])) for g in range(n)]
    tape = np.random.RandomState(seed).random_sample(400000)
    mcts = DeviceMCTS(stub_q_device, H, H, S, base, True, depth, breadth, sequential=True, tape_u=tape, tt_capacity=1 << 17)
    sp = CpuSelfPlay(games, net=None, threads=1, base=base, training=True, max_depth=depth, max_breadth=breadth)
    w = sp.workers[0]
    w.set_tape(tape)
    live = np.arange(n, dtype=np.int32)
    bad = None
    for turn in range(10):
        if len(live) == 0:
            break
        d_slots = torch.as_tensor(live, device="cuda")
        alive = eng.alive(slots=d_slots)
        alive_h = alive.cpu().numpy().astype(bool)
        ev0 = mcts.stats["net_evals"]
        V, moves = mcts.search(eng, d_slots, alive)
        gi, si = np.nonzero(alive_h)
        ids = [[int(live[g]), int(s)] for g, s in zip(gi, si)]
        Vh, mh = V.cpu().numpy()[gi, si], moves.cpu().numpy()[gi, si]
        mcts.end_of_turn()
        done = eng.new((len(live),), torch.uint8, 0)
        spawned = eng.new((len(live),), torch.int16, 0)
        eng.step(moves.contiguous(), slots=d_slots, done=done, spawned=spawned)
        spawn_full = np.full(n, -1, np.int16)
        spawn_full[live] = spawned.cpu().numpy()
        cev0 = w.stats()["net_evals"]
        w.set_spawn_tape(spawn_full)
        sp.root_turn()
        cids, cV, cmv = w.last()
        turns_total += 1
        if cids.tolist() != ids or cmv.tolist() != mh.tolist() or w.stats()["net_evals"] - cev0 != mcts.stats["net_evals"] - ev0 \
                or w.L.mc_tape_pos(w.h) != mcts.tape_pos:
            bad = (turn, "ids" if cids.tolist() != ids else "moves / evaluations / draws")
            break
        worst = max(worst, float(np.abs(cV - Vh).max()))
        live = live[~done.cpu().numpy().astype(bool)]
    if bad is None:
        out = eng.export()
        for g in range(n):
            a, b = compact_from_state(out[g]), w.game(g).compact()
            for k in a:
                assert np.array_equal(a[k], b[k]), (seed, g, k)
        ok += 1
    else:
        diverged += 1
        print(f"seed {seed} ({H}x{H}, {S} snakes, dec {hd}, {n} games, breadth {breadth}, depth {depth}, base {base}): diverged at turn {bad[0]}: {bad[1]}", flush=True)
    sp.close()
    if (seed - first) % 5 == 4:
        print(f"seed {seed}: {ok} runs identical, {diverged} diverged, {turns_total} root turns, worst |dQ| {worst:.1e}, {time.time() - t0:.0f} s", flush=True)
print(f"fuzz done: {ok} of {count} runs identical turn by turn (ids, moves, evaluation counts, draws, final boards), {diverged} diverged; worst root |dQ| {worst:.2e}")
