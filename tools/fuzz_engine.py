"""differential fuzz of the tick / observe kernels against the C oracle (development aid; the bounded version is
tests/test_engine_gpu.py::test_random_geometries_track_the_oracle): fuzz_engine.py <first seed> <n seeds>
Random board sizes 5..19, 2..8 snakes, health decrement 1 / 3 / 9, device start boards, random moves, device food spawns;
whole states after every tick and every observation / mask / key at the end must equal the oracle's."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "alphasnake-zero_amd")]
import numpy as np, torch
import snake_engine as se
from snake_engine.engine import compact_from_state
from oracle import snake_oracle as oracle
from oracle.obs_key import obs_key, obstacle_mask
first, count = int(sys.argv[1]), int(sys.argv[2])
t0 = time.time()
ticks = obs = 0
for seed in range(first, first + count):
    rng = np.random.RandomState(1000 + seed)
    for _ in range(4):
        hw, S, hd = int(rng.randint(5, 20)), int(rng.randint(2, 9)), int(rng.choice([1, 3, 9]))
        n, T = 64, int(rng.randint(10, 60))
        p_straight = float(rng.choice([0.3, 0.5, 0.8]))
        eng = se.Engine(n, hw, hw, S, hd, 0.15, seed=int(rng.randint(1 << 30)))
        eng.reset()
        start = eng.export()
        games = [oracle.Game.from_compact(hw, hw, S, hd, 0.15, compact_from_state(start[g])) for g in range(n)]
        spawned = eng.new((n,), torch.int16, 0)
        for t in range(T):
            mv = rng.randint(0, 3, size=(n, S)).astype(np.uint8)
            mv[rng.rand(n, S) < p_straight] = 1
            eng.step(torch.as_tensor(mv, device="cuda"), spawned=spawned)
            sp = spawned.cpu().numpy()
            out = eng.export()
            for g in range(n):
                if sum(games[g].g.alive[:S]) > 1:
                    games[g].tic(mv[g], spawn_cell=int(sp[g]))
                    ticks += 1
                a, b = compact_from_state(out[g]), games[g].compact()
                for k in a:
                    assert np.array_equal(a[k], b[k]), (seed, hw, S, hd, t, g, k)
            if t % 7 == 3 or t == T - 1:
                pairs = np.argwhere(eng.alive().cpu().numpy()).astype(np.int32)
                if len(pairs):
                    planes, mask, key = eng.observe_all(pairs)
                    ph, mh, kh = planes.cpu().numpy(), mask.cpu().numpy(), key.cpu().numpy().view(np.uint64)
                    for i, (g, s_) in enumerate(pairs):
                        ref = games[g].make_state(int(s_))
                        assert ph[i].tobytes() == ref.tobytes(), (seed, hw, S, t, g, s_)
                        assert np.array_equal(mh[i].astype(bool), obstacle_mask(ref)[0]) and np.array_equal(kh[i], obs_key(ref)[0])
                    obs += len(pairs)
    if (seed - first) % 5 == 4:
        print(f"seed {seed}: {ticks} ticks, {obs} observations equal so far, {time.time() - t0:.0f} s", flush=True)
print(f"fuzz ok: seeds {first}..{first + count - 1}, {ticks} game ticks and {obs} observations bit-identical to the oracle")
