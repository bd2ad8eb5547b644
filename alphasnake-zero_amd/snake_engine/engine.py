"""Engine: N Battlesnake games resident in HBM, stepped / cloned / observed by HIP kernels.

Thin, allocation-aware wrapper over the C ABI (include/snake_engine.h).  Device buffers are
torch tensors (plumbing); their raw pointers and the current HIP stream go through ctypes.
"""
import ctypes as C

import numpy as np
import torch

from ._lib import lib, check, SnkGameState, EngineError

NHWC_F32, NCHW_F32, NCHW_BF16 = 0, 1, 2


def _ptr(t):
    return 0 if t is None else t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_raw_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream():
    """the current HIP stream's handle.  torch.cuda.current_stream() builds a Stream object through four Python layers (6.5 us per
    call, five or six calls per rollout tick: round 6's host profile of a small run); the two C getters behind it take 0.3 us"""
    if _raw_stream is not None and _raw_device is not None:
        return _raw_stream(_raw_device())
    return torch.cuda.current_stream().cuda_stream


class Engine:
    def __init__(self, n_slots, height=11, width=11, snake_cnt=4, health_dec=1, food_spawn_chance=0.15,
                 seed=1234, device=None):
        if not torch.cuda.is_available():
            raise EngineError("snake_engine.Engine needs an MI355X (torch.cuda.is_available() is False); "
                              "there is no CPU fallback")
        self.L = lib()
        device = torch.cuda.current_device() if device is None else int(device)      # one process per GPU: the current device
        self.device = torch.device("cuda", device)
        self.n_slots, self.H, self.W, self.S = int(n_slots), int(height), int(width), int(snake_cnt)
        self.health_dec, self.food_spawn_chance = int(health_dec), float(food_spawn_chance)
        h = C.c_void_p()
        check(self.L.snk_engine_create(C.byref(h), self.n_slots, self.H, self.W, self.S, self.health_dec,
                                       self.food_spawn_chance, int(seed) & 0xFFFFFFFFFFFFFFFF, device))
        self.h = h
        sb = C.c_int()
        check(self.L.snk_engine_info(self.h, None, None, None, None, C.byref(sb)))
        self.slot_bytes = sb.value
        self.FW = (self.H * self.W + 63) // 64
        self.obs_shape = (2 * self.H - 1, 2 * self.W - 1, 3)
        self.obs_elems = self.obs_shape[0] * self.obs_shape[1] * 3

    def close(self):
        if getattr(self, "h", None):
            self.L.snk_engine_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- helpers ---------------------------------------------------------------------------
    def _i32(self, x):
        if x is None:
            return None
        if isinstance(x, torch.Tensor):
            assert x.dtype == torch.int32 and x.is_cuda and x.is_contiguous()
            return x
        return torch.as_tensor(np.ascontiguousarray(x, np.int32), device=self.device)

    def new(self, shape, dtype, fill=None):
        if fill is None:
            return torch.empty(shape, dtype=dtype, device=self.device)
        return torch.full(shape, fill, dtype=dtype, device=self.device)

    def set_params(self, health_dec=None, food_spawn_chance=None):
        if health_dec is not None:
            self.health_dec = int(health_dec)
        if food_spawn_chance is not None:
            self.food_spawn_chance = float(food_spawn_chance)
        check(self.L.snk_engine_set_params(self.h, self.health_dec, self.food_spawn_chance))

    # ---- Game.__init__ ---------------------------------------------------------------------
    def reset(self, slots=None, n=None, init_tape=None):
        slots = self._i32(slots)
        n = (len(slots) if slots is not None else self.n_slots) if n is None else n
        tape = None
        if init_tape is not None:
            tape = torch.as_tensor(np.ascontiguousarray(init_tape, np.uint8), device=self.device)
            assert tape.shape == (n, 3, self.S), tape.shape
        check(self.L.snk_engine_reset(self.h, _ptr(slots), n, _ptr(tape), _stream()))

    # ---- Game.subgame ----------------------------------------------------------------------
    def clone_to(self, dst, src_slots=None, n=None, dst_slots=None, fanout=1):
        src_slots, dst_slots = self._i32(src_slots), dst._i32(dst_slots)
        n = (len(src_slots) if src_slots is not None else self.n_slots) if n is None else n
        check(self.L.snk_engine_clone(self.h, _ptr(src_slots), n, dst.h, _ptr(dst_slots), fanout, _stream()))

    # ---- Game.tic --------------------------------------------------------------------------
    def step(self, moves, slots=None, n=None, spawn_tape=None, done=None, spawned=None, empty=None):
        slots = self._i32(slots)
        n = (len(slots) if slots is not None else self.n_slots) if n is None else n
        assert moves.dtype == torch.uint8 and moves.is_cuda and moves.is_contiguous() and moves.numel() >= n * self.S
        if spawn_tape is not None:
            assert spawn_tape.dtype == torch.int16 and spawn_tape.is_cuda and spawn_tape.numel() >= n
        check(self.L.snk_engine_step(self.h, _ptr(slots), n, _ptr(moves), _ptr(spawn_tape), _ptr(done),
                                     _ptr(spawned), _ptr(empty), _stream()))

    def step_active(self, active, moves, n, done=None, skip=0):
        """slots 0..n-1, only those with active[i] != 0 move (the rollout loop's depth-retired sub-games stay frozen);
        skip: address of a device int32 that, when non-zero at launch time, freezes every game (the rollout tick's gate)"""
        assert active.dtype == torch.uint8 and active.is_cuda and active.is_contiguous() and active.numel() >= n
        assert moves.dtype == torch.uint8 and moves.is_cuda and moves.is_contiguous() and moves.numel() >= n * self.S
        check(self.L.snk_engine_step_active(self.h, _ptr(active), n, _ptr(moves), _ptr(done), skip or None, _stream()))

    def alive(self, slots=None, n=None, out=None, n_alive=None):
        slots = self._i32(slots)
        n = (len(slots) if slots is not None else self.n_slots) if n is None else n
        if out is None:
            out = self.new((n, self.S), torch.uint8)
        check(self.L.snk_engine_alive(self.h, _ptr(slots), n, _ptr(out), _ptr(n_alive), _stream()))
        return out

    def ids(self, slots=None, n=None):
        """(slot, snake id) pairs of every alive snake (games in the given order, ids ascending) and their count"""
        slots = self._i32(slots)
        n = (len(slots) if slots is not None else self.n_slots) if n is None else n
        m = max(1, n * self.S)
        pairs = self.new((m, 2), torch.int32)
        cnt = self.new((1,), torch.int32)
        alive = self.new((m,), torch.uint8)
        scratch = self.new((self.L.snk_compact_scratch_elems(m) + m,), torch.int32)
        check(self.L.snk_engine_ids(self.h, _ptr(slots), n, _ptr(pairs), _ptr(cnt), _ptr(alive), _ptr(scratch), _stream()))
        return pairs, cnt

    # ---- Game.get_states + obstacle mask + transposition key ----------------------------------
    def observe(self, pairs, m=None, planes=None, mask=None, key=None, layout=NHWC_F32, legacy_mask=False, index=None,
                sub_active=None, row_active=None):
        """index: int32[m], row i observes pairs[index[i]]; sub_active uint8[n_slots] + row_active uint8[m]: row_active[i] = the
        observing snake is alive and its slot is active (snk_engine_observe_rows: the rollout tick's forms)"""
        pairs = self._i32(pairs)
        m = pairs.shape[0] if m is None else m
        if index is None and row_active is None:
            check(self.L.snk_engine_observe(self.h, _ptr(pairs), m, layout, _ptr(planes), _ptr(mask), _ptr(key),
                                            int(legacy_mask), _stream()))
            return
        assert index is None or (index.dtype == torch.int32 and index.is_cuda and index.is_contiguous() and index.numel() >= m)
        assert (row_active is None) == (sub_active is None)
        if row_active is not None:
            assert sub_active.dtype == row_active.dtype == torch.uint8 and sub_active.is_cuda and row_active.is_cuda and row_active.numel() >= m
        check(self.L.snk_engine_observe_rows(self.h, _ptr(pairs), _ptr(index), m, layout, _ptr(planes), _ptr(mask), _ptr(key),
                                             int(legacy_mask), _ptr(sub_active), _ptr(row_active), _stream()))

    def observe_all(self, pairs, want_planes=True, want_mask=True, want_key=True, layout=NHWC_F32, legacy_mask=False):
        pairs = self._i32(pairs)
        m = pairs.shape[0]
        shape = (m,) + (self.obs_shape if layout == NHWC_F32 else (3,) + self.obs_shape[:2])
        planes = self.new(shape, torch.bfloat16 if layout == NCHW_BF16 else torch.float32) if want_planes else None
        mask = self.new((m, 3), torch.uint8) if want_mask else None
        key = self.new((m, 2), torch.int64) if want_key else None
        self.observe(pairs, m, planes, mask, key, layout, legacy_mask)
        return planes, mask, key

    # ---- host views --------------------------------------------------------------------------
    def export(self, slots=None):
        if slots is None:
            n, hs = self.n_slots, None
        else:
            hs = np.ascontiguousarray(slots, np.int32)
            n = len(hs)
        arr = (SnkGameState * n)()
        check(self.L.snk_engine_export_sync(self.h, hs.ctypes.data if hs is not None else None, n, arr))
        return arr

    def import_states(self, states, slots=None, ring_start=0):
        n = len(states)
        arr = (SnkGameState * n)(*states) if not isinstance(states, C.Array) else states
        hs = None if slots is None else np.ascontiguousarray(slots, np.int32)
        if ring_start:          # tests: the same games with their ring buffers laid out from that index on (snk_engine_import_at_sync)
            check(self.L.snk_engine_import_at_sync(self.h, hs.ctypes.data if hs is not None else None, n, arr, int(ring_start)))
            return
        check(self.L.snk_engine_import_sync(self.h, hs.ctypes.data if hs is not None else None, n, arr))

    def sum_counters(self, slots=None, n=None):
        slots = self._i32(slots)
        n = (len(slots) if slots is not None else self.n_slots) if n is None else n
        out = (C.c_int64 * 6)()
        check(self.L.snk_engine_sum_counters_sync(self.h, _ptr(slots), n, out))
        return [int(v) for v in out]

    def compact(self, flags, n=None):
        """indices (ascending) of non-zero uint8 flags; returns (idx tensor sized n, count tensor[1])"""
        n = flags.numel() if n is None else n
        out = self.new((max(n, 1),), torch.int32)
        cnt = self.new((1,), torch.int32)
        scratch = self.new((self.L.snk_compact_scratch_elems(n),), torch.int32)
        check(self.L.snk_compact_flags(_ptr(flags), n, _ptr(out), _ptr(cnt), _ptr(scratch), _stream()))
        return out, cnt


# ---- conversions between the canonical host struct and the golden-vector dict format ----------
def state_from_compact(H, W, S, st, uid=0):
    g = SnkGameState()
    g.H, g.W, g.S, g.uid = H, W, S, uid
    nodes = np.full((8, 384), -1, np.int16)
    nn = np.asarray(st["nodes"])
    nodes[:S, :nn.shape[1]] = nn
    C.memmove(g.nodes, nodes.ctypes.data, nodes.nbytes)
    for s in range(S):
        g.alive[s] = int(st["alive"][s]); g.health[s] = int(st["health"][s]); g.length[s] = int(st["length"][s])
        g.dir[s] = int(st["dir"][s]); g.rewards[s] = int(st["rewards"][s])
    food = np.zeros(361, np.uint8)
    food[:H * W] = st["food"]
    C.memmove(g.food, food.ctypes.data, 361)
    for i in range(6):
        g.counters[i] = int(st["counters"][i])
    return g


def compact_from_state(g):
    S, n = g.S, g.H * g.W
    return dict(
        alive=np.array(g.alive[:S], np.uint8), health=np.array(g.health[:S], np.int16),
        length=np.array(g.length[:S], np.int16), dir=np.array(g.dir[:S], np.uint8),
        nodes=np.ctypeslib.as_array(g.nodes).copy()[:S],
        food=np.array(g.food[:n], np.uint8), rewards=np.array(g.rewards[:S], np.int8),
        counters=np.array(g.counters[:], np.int32))
