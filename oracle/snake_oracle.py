"""ctypes binding of oracle/snake_oracle.c (TEST INFRASTRUCTURE, see oracle/__init__.py)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libsnake_oracle.so")

MAX_S, MAX_CELLS, MAX_NODES = 8, 361, 384


class OrcGame(C.Structure):
    _fields_ = [
        ("H", C.c_int32), ("W", C.c_int32), ("S", C.c_int32), ("health_dec", C.c_int32),
        ("food_chance", C.c_double),
        ("alive", C.c_uint8 * MAX_S),
        ("health", C.c_int16 * MAX_S),
        ("length", C.c_int16 * MAX_S),
        ("dir", C.c_uint8 * MAX_S),
        ("nodes", (C.c_int16 * MAX_NODES) * MAX_S),
        ("food", C.c_uint8 * MAX_CELLS),
        ("rewards", C.c_int8 * MAX_S),
        ("counters", C.c_int32 * 6),
    ]


def build(force=False):
    src = os.path.join(_HERE, "snake_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        assert L.orc_sizeof_game() == C.sizeof(OrcGame), (L.orc_sizeof_game(), C.sizeof(OrcGame))
        P = C.POINTER
        L.orc_init.argtypes = [P(OrcGame), C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_subgame.argtypes = [P(OrcGame), P(OrcGame)]
        L.orc_empty_cells.argtypes = [P(OrcGame), C.c_void_p]
        L.orc_tic.argtypes = [P(OrcGame), C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_double, C.c_void_p, P(C.c_int)]
        L.orc_tic.restype = C.c_int
        L.orc_make_state.argtypes = [P(OrcGame), C.c_int, C.c_void_p]
        L.orc_obstacle_mask.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.orc_obs_key.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.orc_stub_q.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.orc_tic_batch.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_make_states_batch.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_make_states_batch.restype = C.c_int
        _lib = L
    return _lib


class Game:
    """One oracle game.  Mirrors the slice of utils.game.Game that the hot path uses."""

    def __init__(self, g=None):
        self.g = g if g is not None else OrcGame()

    # -- construction ------------------------------------------------------------------------
    @classmethod
    def new(cls, H, W, S, health_dec, food_chance, positions, dirs, food_choice):
        self = cls()
        p = np.ascontiguousarray(positions, np.uint8)
        d = np.ascontiguousarray(dirs, np.uint8)
        f = np.ascontiguousarray(food_choice, np.uint8)
        lib().orc_init(C.byref(self.g), H, W, S, health_dec, food_chance, p.ctypes.data, d.ctypes.data, f.ctypes.data)
        return self

    @classmethod
    def from_compact(cls, H, W, S, health_dec, food_chance, st):
        """st: dict with alive/health/length/dir/nodes/food/rewards/counters arrays (golden format)."""
        self = cls()
        g = self.g
        g.H, g.W, g.S, g.health_dec, g.food_chance = H, W, S, health_dec, food_chance
        nodes = np.full((MAX_S, MAX_NODES), -1, np.int16)
        nn = np.asarray(st["nodes"])
        nodes[:S, :nn.shape[1]] = nn
        C.memmove(g.nodes, nodes.ctypes.data, nodes.nbytes)
        for s in range(S):
            g.alive[s] = int(st["alive"][s]); g.health[s] = int(st["health"][s])
            g.length[s] = int(st["length"][s]); g.dir[s] = int(st["dir"][s])
            g.rewards[s] = int(st["rewards"][s])
        food = np.zeros(MAX_CELLS, np.uint8)
        food[:H * W] = st["food"]
        C.memmove(g.food, food.ctypes.data, MAX_CELLS)
        for i in range(6):
            g.counters[i] = int(st["counters"][i])
        return self

    def compact(self, L=None):
        g = self.g
        S, n = g.S, g.H * g.W
        nodes = np.ctypeslib.as_array(g.nodes).copy()[:S]
        if L is not None:
            nodes = nodes[:, :L]
        return dict(
            alive=np.array(g.alive[:S], np.uint8), health=np.array(g.health[:S], np.int16),
            length=np.array(g.length[:S], np.int16), dir=np.array(g.dir[:S], np.uint8), nodes=nodes,
            food=np.array(g.food[:n], np.uint8), rewards=np.array(g.rewards[:S], np.int8),
            counters=np.array(g.counters[:], np.int32))

    def subgame(self):
        o = Game()
        lib().orc_subgame(C.byref(self.g), C.byref(o.g))
        return o

    # -- hot path ----------------------------------------------------------------------------
    def tic(self, moves_dense, spawn_cell=-1, draws=None, want_empty=False):
        mv = np.ascontiguousarray(moves_dense, np.uint8)
        empty = np.zeros(MAX_CELLS, np.uint8) if want_empty else None
        spawned = C.c_int(-1)
        if draws is None:
            done = lib().orc_tic(C.byref(self.g), mv.ctypes.data, 0, int(spawn_cell), 0.0, 0.0,
                                 empty.ctypes.data if want_empty else None, C.byref(spawned))
        else:
            done = lib().orc_tic(C.byref(self.g), mv.ctypes.data, 1, -1, float(draws[0]), float(draws[1]),
                                 empty.ctypes.data if want_empty else None, C.byref(spawned))
        self.last_spawn = spawned.value
        self.last_empty = empty[: self.g.H * self.g.W] if want_empty else None
        return bool(done)

    def empty_cells(self):
        e = np.zeros(MAX_CELLS, np.uint8)
        lib().orc_empty_cells(C.byref(self.g), e.ctypes.data)
        return e[: self.g.H * self.g.W]

    def alive_ids(self):
        return [s for s in range(self.g.S) if self.g.alive[s]]

    def make_state(self, snake_id):
        out = np.empty((2 * self.g.H - 1, 2 * self.g.W - 1, 3), np.float32)
        lib().orc_make_state(C.byref(self.g), snake_id, out.ctypes.data)
        return out

    def get_states(self):
        return [self.make_state(s) for s in self.alive_ids()]

    @property
    def rewards(self):
        return [None if r == 0 else float(r) for r in self.g.rewards[: self.g.S]]


def obstacle_mask(state, legacy=False):
    st = np.ascontiguousarray(state, np.float32)
    m = np.zeros(3, np.uint8)
    lib().orc_obstacle_mask(st.ctypes.data, st.shape[0], st.shape[1], int(legacy), m.ctypes.data)
    return m


def obs_key(state):
    st = np.ascontiguousarray(state, np.float32)
    k = np.zeros(2, np.uint64)
    lib().orc_obs_key(st.ctypes.data, st.shape[0] * st.shape[1], k.ctypes.data)
    return k


def stub_q(state):
    st = np.ascontiguousarray(state, np.float32)
    q = np.zeros(3, np.float32)
    lib().orc_stub_q(st.ctypes.data, st.shape[0], st.shape[1], q.ctypes.data)
    return q
