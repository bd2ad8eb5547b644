"""Driver of oracle/mcts_cpu.c (TEST INFRASTRUCTURE, see oracle/__init__.py): the C restatement of the reference's
MCTS self-play loop (agent.py:25-223, mp_game_runner.py:23-115) on host cores, used as bench.py's `cpu_baseline` and
as a second checker.  Root games are cut into shards, one mc_worker (= one Agent + one MPGameRunner, its own cache)
per shard; the shards advance in lock step so that every rollout tick makes ONE net batch for all of them
(agent.py:189-190), evaluated by `net` (PyTorch-CPU, oracle/net_ref.py) -- or, with net=None, by the deterministic
stub net inside the C code.  The per-shard C calls run on a thread pool (ctypes drops the GIL)."""
import ctypes as C
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from . import snake_oracle as so

_bound = False


def _lib():
    global _bound
    L = so.lib()
    if not _bound:
        P = C.POINTER
        assert L.mc_sizeof_game() == C.sizeof(so.OrcGame)
        L.mc_create.restype = C.c_void_p
        L.mc_create.argtypes = [C.c_int] * 5 + [C.c_double] + [C.c_int] * 5 + [C.c_uint64, C.c_int]
        L.mc_destroy.argtypes = [C.c_void_p]
        L.mc_set_game.argtypes = [C.c_void_p, C.c_int, P(so.OrcGame)]
        L.mc_get_game.argtypes = [C.c_void_p, C.c_int, P(so.OrcGame)]
        L.mc_set_tape.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
        L.mc_tape_pos.argtypes = [C.c_void_p]
        L.mc_tape_pos.restype = C.c_int64
        for name in ("mc_max_rows", "mc_n_live", "mc_overflow", "mc_epochs", "mc_end_turn"):
            getattr(L, name).argtypes = [C.c_void_p]
            getattr(L, name).restype = C.c_int
        for name in ("mc_begin_turn", "mc_begin_epoch", "mc_end_epoch"):
            getattr(L, name).argtypes = [C.c_void_p]
            getattr(L, name).restype = None
        L.mc_collect.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.mc_collect.restype = C.c_int
        L.mc_apply.argtypes = [C.c_void_p, C.c_void_p]
        L.mc_apply.restype = None
        L.mc_set_spawn_tape.argtypes = [C.c_void_p, C.c_void_p]
        L.mc_cache_size.argtypes = [C.c_void_p]
        L.mc_cache_size.restype = C.c_int64
        L.mc_run_stub_turns.argtypes = [C.c_void_p, C.c_int]
        L.mc_run_stub_turns.restype = C.c_int
        L.mc_last.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.mc_last.restype = C.c_int
        L.mc_stats.argtypes = [C.c_void_p, C.c_void_p]
        L.mc_totals.argtypes = [C.c_void_p, C.c_void_p]
        L.mc_records.argtypes = [C.c_void_p]
        L.mc_records.restype = C.c_void_p
        L.mc_values.argtypes = [C.c_void_p]
        L.mc_values.restype = C.c_void_p
        _bound = True
    return L


class Worker:
    """one mc_worker: a shard of root games with its own Agent (cache) and runner"""

    def __init__(self, games, base=2, training=True, max_depth=8, max_breadth=50, stub=False, keep_records=True,
                 seed=0, tt_log2=None):
        self.L = _lib()
        g0 = games[0].g
        self.H, self.W, self.S, self.n = g0.H, g0.W, g0.S, len(games)
        par = min(8, max_breadth)
        if tt_log2 is None:      # one root turn inserts at most per_turn keys; mc_end_turn keeps the table at most half full
            per_turn = self.n * par * self.S * max(1, max_depth) * max(1, max_breadth // par)      # (and doubles it when needed)
            tt_log2 = min(28, max(12, int(np.ceil(np.log2(4.0 * per_turn)))))
        self.h = C.c_void_p(self.L.mc_create(self.H, self.W, self.S, g0.health_dec, self.n, float(base), int(training),
                                             int(max_depth), int(max_breadth), int(stub), int(keep_records),
                                             int(seed) & 0xFFFFFFFFFFFFFFFF, int(tt_log2)))
        for i, g in enumerate(games):
            self.L.mc_set_game(self.h, i, C.byref(g.g))
        self.obs_shape = (2 * self.H - 1, 2 * self.W - 1, 3)
        rows = self.L.mc_max_rows(self.h)
        self.stub = stub
        self.planes = None if stub else np.empty((rows,) + self.obs_shape, np.float32)
        self.mask = None if stub else np.empty((rows, 3), np.uint8)
        self._tape = None
        self._spawn = None

    def close(self):
        if self.h:
            self.L.mc_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_tape(self, u):
        self._tape = np.ascontiguousarray(u, np.float64)
        self.L.mc_set_tape(self.h, self._tape.ctypes.data, len(self._tape))

    def set_spawn_tape(self, cells):
        self._spawn = None if cells is None else np.ascontiguousarray(cells, np.int16)
        self.L.mc_set_spawn_tape(self.h, None if cells is None else self._spawn.ctypes.data)

    def collect(self):
        return self.L.mc_collect(self.h, None if self.stub else self.planes.ctypes.data,
                                 None if self.stub else self.mask.ctypes.data)

    def apply(self, q):
        if q is None:
            self.L.mc_apply(self.h, None)
        else:
            q = np.ascontiguousarray(q, np.float32)
            self.L.mc_apply(self.h, q.ctypes.data)

    def last(self):
        n = self.n * self.S
        ids = np.empty((n, 2), np.int32); V = np.empty((n, 3), np.float32); mv = np.empty(n, np.uint8)
        k = self.L.mc_last(self.h, ids.ctypes.data, V.ctypes.data, mv.ctypes.data)
        return ids[:k], V[:k], mv[:k]

    def stats(self):
        out = np.zeros(8, np.int64)
        self.L.mc_stats(self.h, out.ctypes.data)
        return dict(zip(("env_steps", "sim_steps", "net_evals", "lookups", "n_records", "tt_used", "tt_cap", "now"), out.tolist()))

    def totals(self):
        out = np.zeros(6, np.int64)
        self.L.mc_totals(self.h, out.ctypes.data)
        return out

    def records(self):
        n = self.stats()["n_records"]
        if n == 0:
            return np.zeros((0,) + self.obs_shape, np.float32), np.zeros((0, 3), np.float32)
        rec = np.ctypeslib.as_array(C.cast(self.L.mc_records(self.h), C.POINTER(C.c_float)), (n,) + self.obs_shape).copy()
        val = np.ctypeslib.as_array(C.cast(self.L.mc_values(self.h), C.POINTER(C.c_float)), (n, 3)).copy()
        return rec, val

    def game(self, i):
        g = so.Game()
        self.L.mc_get_game(self.h, i, C.byref(g.g))
        return g

    def check(self):
        if self.L.mc_overflow(self.h):
            raise RuntimeError("mcts_cpu: transposition table too small (raise tt_log2)")


class CpuSelfPlay:
    """MPGameRunner.run + Agent over `threads` shards.  net: callable planes[n,h,w,3] float32 -> Q[n,3] float32
    (AlphaNNet.v, obstacle mask applied), or None for the stub net inside the C code."""

    def __init__(self, games, net=None, threads=1, base=2, training=True, max_depth=8, max_breadth=50, seed=0,
                 keep_records=True):
        threads = max(1, min(int(threads), len(games)))
        self.net = net
        cuts = np.linspace(0, len(games), threads + 1).astype(int)
        self.workers = [Worker(games[a:b], base, training, max_depth, max_breadth, stub=net is None,
                               keep_records=keep_records, seed=seed * 1000003 + i)
                        for i, (a, b) in enumerate(zip(cuts[:-1], cuts[1:])) if b > a]
        self.pool = ThreadPoolExecutor(len(self.workers)) if len(self.workers) > 1 else None
        self.threads = len(self.workers)

    def _map(self, fn, items):
        if self.pool is None:
            return [fn(x) for x in items]
        return list(self.pool.map(fn, items))

    def root_turn(self):
        ws = [w for w in self.workers if w.L.mc_n_live(w.h) > 0]
        if not ws:
            return 0
        L = ws[0].L
        if self.net is None:                   # whole turn inside C, one thread per shard
            self._map(lambda w: L.mc_run_stub_turns(w.h, 1), ws)
        else:
            self._map(lambda w: L.mc_begin_turn(w.h), ws)
            for _ in range(L.mc_epochs(ws[0].h)):
                self._map(lambda w: L.mc_begin_epoch(w.h), ws)
                active = list(ws)
                while active:
                    cnt = self._map(lambda w: w.collect(), active)
                    active = [w for w, c in zip(active, cnt) if c >= 0]
                    cnt = [c for c in cnt if c >= 0]
                    if not active:
                        break
                    tot = sum(cnt)
                    if tot:                    # ONE net batch per rollout tick for all shards (agent.py:189-190)
                        q = self.net(np.concatenate([w.planes[:c] for w, c in zip(active, cnt) if c]))
                        offs = np.cumsum([0] + cnt)
                        qs = [q[offs[i]:offs[i + 1]] for i in range(len(active))]
                    else:
                        qs = [np.zeros((0, 3), np.float32)] * len(active)
                    self._map(lambda wq: wq[0].apply(wq[1]), list(zip(active, qs)))
                self._map(lambda w: L.mc_end_epoch(w.h), ws)
            self._map(lambda w: L.mc_end_turn(w.h), ws)
        for w in ws:
            w.check()
        return sum(L.mc_n_live(w.h) for w in self.workers)

    def run(self, max_turns=None):
        t = 0
        while (max_turns is None or t < max_turns) and self.root_turn() > 0:
            t += 1
        return self.stats()

    def stats(self):
        tot = {}
        for w in self.workers:
            for k, v in w.stats().items():
                tot[k] = tot.get(k, 0) + v
        return tot

    def close(self):
        if self.pool is not None:
            self.pool.shutdown()
        for w in self.workers:
            w.close()


def seeded_games(n, H=11, W=11, S=4, health_dec=1, seed=0):
    """n start boards drawn like Game.__init__ (game.py:25-30, 43-47) from python's random.Random(seed)"""
    import random
    rnd = random.Random(seed)
    out = []
    for _ in range(n):
        pos = rnd.sample(range(8), S)
        out.append(so.Game.new(H, W, S, health_dec, 0.15, pos, [rnd.randrange(4) for _ in range(S)],
                               [rnd.randrange(4) for _ in range(S)]))
    return out
