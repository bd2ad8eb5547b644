"""Drop-in mirror of the reference's ``utils.agent`` (agent.py).

``Agent`` keeps the constructor, ``make_moves(games, ids) -> moves``, ``records`` / ``values`` /
``clear()``, ``softermax`` and ``argmaxs`` (agent.py:9-147).  The MCTS itself (sub-game clones,
lock-step rollouts, cache de-duplication, batched net calls, in-rollout and terminal back-ups)
runs on the GPU: see snake_engine/mcts.py and csrc/mcts.hip.  The four cache dicts of the reference
live in one HBM hash table; ``cached_values`` etc. are read-only dict-like views of it (len, in, [key]).
"""
from collections.abc import Sequence

import numpy as np
import torch

from snake_engine import Engine, EngineError
from snake_engine._lib import lib, check
from snake_engine.mcts import DeviceMCTS


_M64 = (1 << 64) - 1


def _sm64(x):
    """splitmix64 finaliser on uint64 arrays (wrap-around arithmetic)"""
    with np.errstate(over="ignore"):
        x = x + np.uint64(0x9E3779B97F4A7C15)
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return x ^ (x >> np.uint64(31))


def state_key(state):
    """The 128-bit digest the device table is keyed on, computed on the host from an observation (an (h, w, 3) float32
    array or the 5 292-byte string the reference uses as its dict key, agent.py:175): sum over the pixels that differ
    from the wall default [0, 1.0, 0] of a splitmix64 mix of (pixel index, the three float bit patterns); the same
    function k_observe evaluates from the compact game state (csrc/engine.hip)."""
    if isinstance(state, (bytes, bytearray, memoryview)):
        bits = np.frombuffer(state, np.uint32)
    else:
        bits = np.ascontiguousarray(state, np.float32).view(np.uint32)
    bits = bits.reshape(-1, 3).astype(np.uint64)
    p = np.arange(bits.shape[0], dtype=np.uint64)
    b0, b1, b2 = bits[:, 0], bits[:, 1], bits[:, 2]
    live = ~((b0 == 0) & (b1 == np.uint64(0x3F800000)) & (b2 == 0))
    x = _sm64((p << np.uint64(32)) | b0)
    x = _sm64(x ^ ((b1 << np.uint64(32)) | b2))
    hi = _sm64(x ^ np.uint64(0xD6E8FEB86659FD93))
    with np.errstate(over="ignore"):
        return np.array([x[live].sum(dtype=np.uint64), hi[live].sum(dtype=np.uint64)], np.uint64)


class _CacheView:
    """One of the reference's four cache dicts (cached_values / total_rewards / visit_cnts / cache_hit, agent.py:16-19)
    as a read-only view of the device transposition table: len(), `key in view`, view[key], view.get(key) with the
    reference's keys (the observation's bytes, agent.py:175) or the observation array itself.  Values are host copies
    ((3,) float32; an int for cache_hit).  The table stores digests, not the keys, so iteration is not possible."""
    _FIELDS = {"cached_values": "q", "total_rewards": "total", "visit_cnts": "visit", "cache_hit": "age"}

    def __init__(self, agent, field="q"):
        self._agent = agent
        self._field = field

    def __len__(self):
        m = self._agent._mcts
        return 0 if m is None or m.tt is None else int(m.tt.status()[1])

    def _probe(self, key):
        m = self._agent._mcts
        if m is None or m.tt is None:
            return None
        k = torch.as_tensor(state_key(key).view(np.int64).reshape(1, 2), device=m.device)
        entry = torch.empty((1,), dtype=torch.int32, device=m.device)
        stat = torch.empty((1, 7), dtype=torch.float32, device=m.device)
        check(lib().snk_tt_find(m.tt.h, k.data_ptr(), 1, m.now, m.max_depth, entry.data_ptr(), stat.data_ptr(),
                                torch.cuda.current_stream().cuda_stream))
        if int(entry.item()) == -1:
            return None
        return stat[0].cpu().numpy()

    def __contains__(self, key):
        return self._probe(key) is not None

    def __getitem__(self, key):
        st = self._probe(key)
        if st is None:
            raise KeyError("state not in the transposition cache")
        if self._field == "q":
            return (st[0:3] / st[3:6]).astype(np.float32)       # cached_values = total / visit (agent.py:72, 199, 220)
        if self._field == "total":
            return st[0:3].copy()
        if self._field == "visit":
            return st[3:6].copy()
        return int(st[6])

    def get(self, key, default=None):
        try:
            return self[key]
        except KeyError:
            return default

    def __iter__(self):
        raise TypeError("the device transposition table stores 128-bit digests of the observations, not the observations: "
                        "its keys cannot be enumerated (look states up with `state.tobytes() in agent.cached_values`)")


class _Records(Sequence):
    """Agent.records (agent.py:96): every root observation, materialised on demand.  The states are
    stored as compact game snapshots in HBM (one per game and turn) and encoded by the observe kernel
    when read, so 10^6 records cost 0.6 GB instead of 5.3 GB."""

    def __init__(self, agent):
        self._a = agent

    def __len__(self):
        return self._a._n_records

    def fetch_device(self, indices):
        """the observations of the given records as one device tensor [k, 2H-1, 2W-1, 3] (no host round trip: what the
        iteration-end all-gather sends)"""
        a = self._a
        idx = np.asarray(indices, np.int64)
        pairs = a._rec_pairs_host()[idx]
        planes, _, _ = a._rec_engine.observe_all(pairs.astype(np.int32), want_mask=False, want_key=False)
        return planes

    def fetch(self, indices):
        return self.fetch_device(indices).cpu().numpy()

    def __getitem__(self, i):
        if isinstance(i, slice):
            return list(self.fetch(range(*i.indices(len(self)))))
        if i < 0:
            i += len(self)
        if not 0 <= i < len(self):
            raise IndexError(i)
        return self.fetch([i])[0]


class _Values(Sequence):
    def __init__(self, agent):
        self._a = agent

    def __len__(self):
        return self._a._n_records

    def __getitem__(self, i):
        v = self._a._values_host()
        if isinstance(i, slice):
            return list(v[i])
        return v[i]


class Agent:
    verbose = None        # None: follow MPGameRunner.verbose; the reference prints one line per MCTS epoch (agent.py:57-58)

    def __init__(self, nnet, softmax_base=100, training=False, max_MCTS_depth=8, max_MCTS_breadth=128,
                 seed=None, sequential=False, tape_u=None, tt_capacity=None):
        self.nnet = nnet
        self.softmax_base = softmax_base
        self.training = training
        self.max_MCTS_depth = max_MCTS_depth
        self.max_MCTS_breadth = max_MCTS_breadth
        self._seed = int(np.random.randint(1 << 62)) if seed is None else int(seed)
        self._sequential, self._tape_u, self._tt_capacity = sequential, tape_u, tt_capacity
        self._mcts = None
        self.cached_values = _CacheView(self, "q")
        self.total_rewards = _CacheView(self, "total")
        self.visit_cnts = _CacheView(self, "visit")
        self.cache_hit = _CacheView(self, "age")
        self._reset_records()
        # record data for training
        if training:
            self.records = _Records(self)
            self.values = _Values(self)

    # ---- records arena -----------------------------------------------------------------------------
    def _reset_records(self):
        self._n_records = 0
        self._rec_engine = None
        self._rec_used = 0
        self._rec_pairs = []        # per turn: int32 [k,2] (arena slot, snake id)
        self._rec_values = []       # per turn: float32 [k,3]
        self._rec_pairs_cat = None
        self._rec_values_cat = None

    def _rec_pairs_host(self):
        if self._rec_pairs_cat is None or len(self._rec_pairs_cat) != self._n_records:
            self._rec_pairs_cat = np.concatenate(self._rec_pairs) if self._rec_pairs else np.zeros((0, 2), np.int32)
        return self._rec_pairs_cat

    def _values_host(self):
        if self._rec_values_cat is None or len(self._rec_values_cat) != self._n_records:
            self._rec_values_cat = np.concatenate(self._rec_values) if self._rec_values else np.zeros((0, 3), np.float32)
        return self._rec_values_cat

    def _record(self, eng, d_slots, alive_h, V):
        G = len(alive_h)
        if self._rec_engine is None or self._rec_used + G > self._rec_engine.n_slots:
            new_cap = max(4 * G, 2 * (self._rec_used + G), 256)
            new = Engine(new_cap, eng.H, eng.W, eng.S, eng.health_dec, 0.0, device=eng.device.index)
            if self._rec_engine is not None and self._rec_used:
                self._rec_engine.clone_to(new, n=self._rec_used, fanout=1)
            self._rec_engine = new
        dst = torch.arange(self._rec_used, self._rec_used + G, dtype=torch.int32, device=eng.device)
        eng.clone_to(self._rec_engine, src_slots=d_slots, dst_slots=dst, fanout=1)
        gi, si = np.nonzero(alive_h)
        self._rec_pairs.append(np.stack([self._rec_used + gi, si], axis=1).astype(np.int32))
        self._rec_values.append(V.cpu().numpy()[gi, si].astype(np.float32))
        self._rec_used += G
        self._n_records += len(gi)

    # ---- the net behind the search ------------------------------------------------------------------
    def _evaluate(self, planes, mask):
        nnet = self.nnet
        m = getattr(self, "_mcts", None)                               # (MCTSAgent has none: its evaluations wait for the guard)
        if m is not None and m.guard is not None:                       # gated ticks: no synchronisation per leaf batch
            return nnet.v_device_unguarded(planes, mask)
        if hasattr(nnet, "v_device"):
            return nnet.v_device(planes, mask)
        V = np.asarray(nnet.v(list(planes.cpu().numpy())), np.float32)      # any object with the reference's .v(X)
        return torch.as_tensor(V, device=planes.device)

    def make_moves(self, games, ids):
        """agent.py:25-111.  ``games``: the runner's dict of engine-backed Games; ``ids``: [(game_id, snake_id)]
        in dict order x alive ids ascending (what MPGameRunner.run passes)."""
        eng = getattr(games, "engine", None)
        if eng is None:
            engs = {id(g._engine) for g in games.values()}
            if len(engs) != 1:
                raise TypeError("Agent.make_moves needs games that live in one snake_engine.Engine (MPGameRunner.games)")
            eng = next(iter(games.values()))._engine
        if self._mcts is None:
            self._mcts = DeviceMCTS(self._evaluate, eng.H, eng.W, eng.S, self.softmax_base, self.training,
                                    self.max_MCTS_depth, self.max_MCTS_breadth, seed=self._seed,
                                    device=eng.device.index, sequential=self._sequential, tape_u=self._tape_u,
                                    tt_capacity=self._tt_capacity)
        self._mcts.guard = getattr(self.nnet, "guard", None)         # the Q-net's range-guard word gates the rollout ticks
        if self.verbose is None:
            from utils.mp_game_runner import MPGameRunner
            self._mcts.verbose = bool(MPGameRunner.verbose)
        else:
            self._mcts.verbose = bool(self.verbose)
        slots = np.fromiter((g._slot for g in games.values()), np.int32, len(games))
        d_slots = torch.as_tensor(slots, device=eng.device)
        alive = eng.alive(slots=d_slots)
        if hasattr(self.nnet, "calibrate") and not getattr(getattr(self.nnet, "_qnet", None), "calibrated", True):
            # new weights: fit the split-f16 kernel's activation scales to real observations (the root states) once
            pairs = torch.nonzero(alive)[:4096].to(torch.int32)
            pairs[:, 0] = d_slots[pairs[:, 0].long()]
            planes, _, _ = eng.observe_all(pairs.contiguous(), want_mask=False, want_key=False)
            self.nnet.calibrate(planes)
        V, moves = self._mcts.search(eng, d_slots, alive)
        alive_h = alive.cpu().numpy().astype(bool)
        if self.training:
            self._record(eng, d_slots, alive_h, V)
        self._mcts.end_of_turn()
        out = moves.cpu().numpy()[alive_h].tolist()
        if len(out) != len(ids):
            raise EngineError(f"ids has {len(ids)} entries but the engine has {len(out)} alive snakes")
        return out

    # a softmax function with customized base (agent.py:114-122), evaluated by the device kernel
    def softermax(self, z):
        return self._soft_arg(np.asarray(z, np.float32).reshape(1, 3))[0][0]

    def argmaxs(self, Z):
        if len(Z) == 0:
            return []
        return self._soft_arg(np.asarray(Z, np.float32).reshape(-1, 3))[1].tolist()

    def _soft_arg(self, z):
        zt = torch.as_tensor(np.ascontiguousarray(z), device="cuda")
        pmf = torch.empty_like(zt)
        am = torch.empty((zt.shape[0],), dtype=torch.uint8, device=zt.device)
        check(lib().snk_softermax_argmax(zt.data_ptr(), zt.shape[0], float(self.softmax_base), pmf.data_ptr(),
                                         am.data_ptr(), torch.cuda.current_stream().cuda_stream))
        return pmf.cpu().numpy(), am.cpu().numpy().astype(int)

    # clear memory (agent.py:140-147)
    def clear(self):
        if self._mcts is not None:
            self._mcts.clear()
        self._reset_records()


class MCTSAgent(Agent):
    """agent.py:149-223: the leaf agent of one rollout epoch -- cache de-duplication, one batched net call,
    randomized move, in-rollout back-up -- with the reference's constructor and its ``keys`` / ``moves``
    path dicts.  ``Agent.make_moves`` does not go through this class (its epochs run fused on the device,
    snake_engine/mcts.py); it exists for callers that drive ``MCTSMPGameRunner`` themselves.  The four cache
    arguments are the parent Agent's cache views (their HBM table is shared); anything else gets a private table."""

    def __init__(self, nnet, softmax_base, games, cached_values, total_rewards, visit_cnts, cache_hit):
        self.nnet = nnet
        self.softmax_base = softmax_base
        self.cached_values = cached_values
        self.total_rewards = total_rewards
        self.visit_cnts = visit_cnts
        self.cache_hit = cache_hit
        self.training = False
        self.keys = {i: {s.id: [] for s in games[i].snakes} for i in games}
        self.moves = {i: {s.id: [] for s in games[i].snakes} for i in games}
        self._entries = {i: {sid: [] for sid in self.keys[i]} for i in games}
        parent = getattr(cached_values, "_agent", None)
        self._parent = parent
        self._own_tt = None
        self._now, self._max_age, self._ctr = 1, 8, 0
        self._seed = int(np.random.randint(1 << 62))
        if parent is not None and parent._mcts is not None:
            self._now, self._max_age = max(1, parent._mcts.now), parent.max_MCTS_depth

    def _table(self):
        from snake_engine.mcts import TranspositionTable
        if self._parent is not None and self._parent._mcts is not None and self._parent._mcts.tt is not None:
            return self._parent._mcts.tt
        if self._own_tt is None:
            self._own_tt = TranspositionTable(1 << 20)
        return self._own_tt

    def make_moves(self, games, ids):
        eng = games.engine
        L, st, dev = lib(), torch.cuda.current_stream().cuda_stream, eng.device
        gid_slot = {gid: g._slot for gid, g in games.items()}
        m = len(ids)
        pairs = torch.as_tensor(np.array([[gid_slot[g], s] for g, s in ids], np.int32).reshape(m, 2), device=dev)
        key = torch.empty((m, 2), dtype=torch.int64, device=dev)
        mask = torch.empty((m, 3), dtype=torch.uint8, device=dev)
        eng.observe(pairs, m, None, mask, key)
        tt = self._table()
        entry = torch.empty((m,), dtype=torch.int32, device=dev)
        new = torch.empty((m,), dtype=torch.uint8, device=dev)
        check(L.snk_tt_lookup_insert(tt.h, key.data_ptr(), None, m, self._now, self._max_age, entry.data_ptr(), new.data_ptr(), st))
        idx = torch.nonzero(new).to(torch.int32).reshape(-1).contiguous()      # agent.py:177-184: one evaluation per new key
        if idx.numel():
            planes = torch.empty((idx.numel(),) + eng.obs_shape, dtype=torch.float32, device=dev)
            eng.observe(pairs.index_select(0, idx).contiguous(), idx.numel(), planes, None, None)
            q = self._evaluate(planes, mask.index_select(0, idx).contiguous()).contiguous()
            check(L.snk_tt_set_priors(tt.h, entry.data_ptr(), idx.data_ptr(), idx.numel(), q.data_ptr(), None, st))
        D = max([len(self._entries[g][s]) for g, s in ids] + [0]) + 1
        pe = np.full((m, D), -1, np.int32); pm = np.zeros((m, D), np.uint8); pl = np.zeros(m, np.int32)
        for i, (g, s) in enumerate(ids):
            n = len(self._entries[g][s])
            pe[i, :n] = self._entries[g][s]; pm[i, :n] = self.moves[g][s]; pl[i] = n
        d_pe, d_pm, d_pl = (torch.as_tensor(a, device=dev) for a in (pe, pm, pl))
        mv = torch.empty((m,), dtype=torch.uint8, device=dev)
        est = torch.empty((m,), dtype=torch.float32, device=dev)
        self._ctr += 1
        check(L.snk_mcts_select(tt.h, entry.data_ptr(), m, float(self.softmax_base), None, None, 0, self._seed, self._ctr, 0x4D41,
                                mv.data_ptr(), est.data_ptr(), None, d_pe.data_ptr(), d_pm.data_ptr(), d_pl.data_ptr(), D, None, st))
        check(L.snk_mcts_backup(tt.h, entry.data_ptr(), m, est.data_ptr(), None, d_pe.data_ptr(), d_pm.data_ptr(),
                                d_pl.data_ptr(), D, 0, None, st))                 # agent.py:208-220
        mv_h, ent_h, key_h = mv.cpu().numpy(), entry.cpu().numpy(), key.cpu().numpy()
        for i, (g, s) in enumerate(ids):                                         # agent.py:221-222
            self.keys[g][s].append(key_h[i].tobytes())
            self.moves[g][s].append(int(mv_h[i]))
            self._entries[g][s].append(int(ent_h[i]))
        return mv_h.tolist()
