"""differential fuzz of the pit pair (utils.pit_mp_game_runner.MPGameRunner + utils.pit_agent.Agent on the HIP engine) against
oracle/pit_oracle.py (pinned to the reference's recorded pit runs) on random boards, team splits and stub nets (development
aid): fuzz_pit.py <first seed> <n seeds>.  The oracle plays first (its food spawns from a seeded uniform stream), the device
replays with the oracle's spawn cells; winner indices must be identical."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "alphasnake-zero_amd")]
import numpy as np, torch
from snake_engine.engine import compact_from_state
from utils.pit_agent import Agent
from utils.pit_mp_game_runner import MPGameRunner
from oracle import snake_oracle as oracle
from oracle.obs_key import StubNet
from oracle.pit_oracle import pit_run
first, count = int(sys.argv[1]), int(sys.argv[2])
games_total = 0
t0 = time.time()
for seed in range(first, first + count):
    rng = np.random.RandomState(9000 + seed)
    H = int(rng.choice([7, 9, 11, 13, 19])); S = int(rng.randint(2, 7)); hd = int(rng.choice([1, 3, 9])); n = int(rng.randint(8, 48))
    a_cnt = int(rng.randint(1, S)) if rng.rand() < 0.7 else None
    wa, wb = int(rng.randint(0, 3)), int(rng.randint(0, 3))
    gr = MPGameRunner(H, H, S, hd, n, seed=int(rng.randint(1 << 30)))
    start = gr.engine.export()
    og = [oracle.Game.from_compact(H, H, S, hd, 0.15, compact_from_state(start[g])) for g in range(n)]
    u = np.random.RandomState(seed).random_sample((4000, n, 2))
    log = []
    want, lengths = pit_run(og, StubNet(wa), StubNet(wb), a_cnt, draws=lambda turn, g: tuple(u[turn - 1, g]), spawn_log=log)
    tape = np.array(log)
    got = gr.run(Agent(StubNet(wa)), Agent(StubNet(wb)), a_cnt, spawn_tape=lambda turn: np.where(tape[turn - 1] < -1, -1, tape[turn - 1]))
    assert got == want, (seed, H, S, hd, n, a_cnt, [(i, a, b) for i, (a, b) in enumerate(zip(got, want)) if a != b][:5])
    games_total += n
    if (seed - first) % 10 == 9:
        print(f"seed {seed}: {games_total} games, winner indices identical so far, {time.time() - t0:.0f} s", flush=True)
print(f"fuzz ok: {count} pit runs, {games_total} games: winner indices identical to the oracle's")
