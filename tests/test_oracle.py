"""CPU tests: the C oracle (oracle/snake_oracle.c) against golden vectors recorded from the
unmodified reference (tests/golden/make_golden.py).  Bit-exact for states, planes and masks."""
import hashlib

import numpy as np
import pytest

from conftest import load_golden, golden_state

TIC_CFGS = ["11x11x4", "11x11x4_dec9", "7x7x2", "19x19x8", "9x9x3", "15x15x5", "16x16x6", "5x5x2"]
KEYS = ("alive", "health", "length", "dir", "food", "rewards", "counters")


def assert_state_equal(got, exp, ctx=""):
    for k in KEYS:
        assert np.array_equal(got[k], exp[k]), f"{ctx}: field {k}: {got[k]} != {exp[k]}"
    L = exp["nodes"].shape[1]
    assert np.array_equal(got["nodes"][:, :L], exp["nodes"]), f"{ctx}: nodes differ"
    assert (got["nodes"][:, L:] == -1).all(), f"{ctx}: nodes beyond golden width"


@pytest.mark.parametrize("cfg", TIC_CFGS)
def test_tic_trajectories(oracle, cfg):
    z = load_golden(f"tic_{cfg}.npz")
    H, W, S, hd = int(z["H"]), int(z["W"]), int(z["S"]), int(z["health_dec"])
    ptr = z["ptr"]
    tick = 0
    for g in range(len(ptr) - 1):
        game = oracle.Game.from_compact(H, W, S, hd, 0.15, golden_state(z, ptr[g]))
        # derived empty set == the reference's incrementally maintained one
        assert np.array_equal(game.empty_cells(), z["st_empty"][ptr[g]])
        for t in range(ptr[g + 1] - ptr[g] - 1):
            done = game.tic(z["moves"][tick], spawn_cell=int(z["spawn"][tick]), want_empty=True)
            if z["spawn_empty_valid"][tick]:
                ref_empty = np.unpackbits(z["spawn_empty"][tick])[: H * W]
                assert np.array_equal(game.last_empty, ref_empty), f"game {g} tick {t}: empty set at spawn"
            exp = golden_state(z, ptr[g] + t + 1)
            assert_state_equal(game.compact(), exp, f"{cfg} game {g} tick {t}")
            assert done == bool(z["done"][tick])
            assert np.array_equal(game.empty_cells(), z["st_empty"][ptr[g] + t + 1])
            tick += 1
    assert tick == len(z["moves"])


@pytest.mark.parametrize("cfg", TIC_CFGS)
def test_init_tape(oracle, cfg):
    z = load_golden(f"tic_{cfg}.npz")
    H, W, S, hd = int(z["H"]), int(z["W"]), int(z["S"]), int(z["health_dec"])
    for g in range(len(z["ptr"]) - 1):
        game = oracle.Game.new(H, W, S, hd, 0.15, z["init_positions"][g], z["init_dirs"][g], z["init_food"][g])
        assert_state_equal(game.compact(), golden_state(z, z["ptr"][g]), f"{cfg} init {g}")


def test_corner_cases(oracle):
    z = load_golden("corner.npz")
    for i, name in enumerate(z["names"]):
        p = f"c{i}_"
        H, W, S, hd = (int(v) for v in z[p + "meta"])
        chance = float(z[p + "chance"])
        game = oracle.Game.from_compact(H, W, S, hd, chance, golden_state(z, 0, p + "st_"))
        for t in range(len(z[p + "moves"])):
            obs = np.array(game.get_states(), np.float32).reshape(-1, 2 * H - 1, 2 * W - 1, 3)
            assert obs.tobytes() == z[p + f"obs{t}"].tobytes(), f"{name}: obs before tick {t}"
            done = game.tic(z[p + "moves"][t], spawn_cell=int(z[p + "spawn"][t]))
            assert_state_equal(game.compact(), golden_state(z, t + 1, p + "st_"), f"{name} tick {t}")
            assert done == bool(z[p + "done"][t]), name


@pytest.mark.parametrize("cfg", TIC_CFGS)
def test_make_state_bytes(oracle, cfg):
    z = load_golden(f"tic_{cfg}.npz")
    s = load_golden(f"states_{cfg}.npz")
    H, W, S, hd = int(z["H"]), int(z["W"]), int(z["S"]), int(z["health_dec"])
    raw_at = {int(j): k for k, j in enumerate(s["raw_index"])}
    for j in range(len(s["state_index"])):
        game = oracle.Game.from_compact(H, W, S, hd, 0.15, golden_state(z, s["state_index"][j]))
        st = game.make_state(int(s["snake_id"][j]))
        b = st.tobytes()
        assert hashlib.blake2b(b, digest_size=16).digest() == s["digest"][j].tobytes(), f"{cfg} obs {j}"
        if j in raw_at:
            assert b == s["raw"][raw_at[j]].tobytes()
        assert np.array_equal(oracle.obstacle_mask(st), s["mask"][j])
        assert np.array_equal(oracle.obstacle_mask(st, legacy=True), s["mask_legacy"][j])
        assert np.array_equal(oracle.obs_key(st), s["key"][j])


def test_obs_key_is_injective_on_goldens():
    """equal observation bytes <=> equal 128-bit key, over every golden observation"""
    for cfg in TIC_CFGS:
        s = load_golden(f"states_{cfg}.npz")
        dig = [d.tobytes() for d in s["digest"]]
        key = [k.tobytes() for k in s["key"]]
        assert len(set(dig)) == len(set(key)) == len(set(zip(dig, key)))


def test_stub_q_matches_numpy(oracle):
    from oracle.obs_key import stub_q
    s = load_golden("states_11x11x4.npz")
    exp = stub_q(s["raw"])
    got = np.array([oracle.stub_q(x) for x in s["raw"]])
    assert exp.tobytes() == got.tobytes()


def test_subgame(oracle):
    z = load_golden("tic_11x11x4.npz")
    game = oracle.Game.from_compact(11, 11, 4, 1, 0.15, golden_state(z, 57))
    sub = game.subgame()
    a, b = game.compact(), sub.compact()
    for k in ("alive", "health", "length", "dir", "nodes", "food", "rewards"):
        assert np.array_equal(a[k], b[k])
    assert (b["counters"] == 0).all() and sub.g.food_chance == 0.0
