"""DeviceMCTS: the reference's randomized parallel MCTS (Agent.make_moves agent.py:25-111 and the
rollout loop MCTSMPGameRunner.run / MCTSAgent.make_moves, mp_game_runner.py:85-115, agent.py:161-223)
run as kernel launches over HBM-resident games.  This file only sequences C-ABI calls; torch is used
for buffers, index plumbing (gather / cumsum) and the single host read-back per rollout tick (the
number of cache misses, which sizes the net batch).
"""
import ctypes as C
import os
from time import time

import numpy as np
import torch

from ._lib import lib, check, EngineError
from .engine import Engine, _ptr, _stream

PARALLEL = 8     # agent.py:32
# round 6: the tick's "which rows are live" and "the rows to evaluate" ride in the observe launches that hold the records anyway
# (snk_engine_observe_rows) instead of three launches of their own (snk_engine_alive, snk_mcts_row_active, snk_mcts_gather_rows):
# 23 launches per tick instead of 26; SNK_MCTS_FOLD=0 keeps the separate launches (A/B runs; same rows, same results)
_FOLD = os.environ.get("SNK_MCTS_FOLD", "1") != "0"

def _pow2_at_least(v):
    p = 1024
    while p < v:
        p <<= 1
    return p


class TranspositionTable:
    """cached_values / total_rewards / visit_cnts / cache_hit (agent.py:16-19) in HBM."""

    def __init__(self, capacity, device=None):
        self.L = lib()
        device = torch.cuda.current_device() if device is None else int(device)
        h = C.c_void_p()
        check(self.L.snk_tt_create(C.byref(h), int(capacity), device))
        self.h = h
        self.generation = 0          # bumped whenever the table's device buffers are replaced (rebuild)

    def close(self):
        if getattr(self, "h", None):
            self.L.snk_tt_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def clear(self):
        check(self.L.snk_tt_clear(self.h, _stream()))

    def status(self):
        cap, occ, ovf = C.c_int64(), C.c_int64(), C.c_int()
        check(self.L.snk_tt_status_sync(self.h, C.byref(cap), C.byref(occ), C.byref(ovf)))
        return cap.value, occ.value, ovf.value

    def rebuild(self, new_capacity, now, max_age):
        check(self.L.snk_tt_rebuild_sync(self.h, int(new_capacity), int(now), int(max_age)))
        self.generation += 1


class DeviceMCTS:
    def __init__(self, evaluate, height, width, snake_cnt, softmax_base=100, training=False, max_depth=8,
                 max_breadth=128, seed=1234, device=None, sequential=False, tape_u=None, tt_capacity=None,
                 legacy_mask=False):
        """evaluate(planes[n,h,w,3] cuda f32, mask[n,3] cuda u8) -> cuda f32 [n,3]  (= AlphaNNet.v)"""
        if not torch.cuda.is_available():
            raise EngineError("DeviceMCTS needs an MI355X; there is no CPU fallback")
        self.L = lib()
        self.evaluate = evaluate
        self.H, self.W, self.S = height, width, snake_cnt
        self.base, self.training = float(softmax_base), bool(training)
        self.max_depth, self.max_breadth = int(max_depth), int(max_breadth)
        self.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        device = torch.cuda.current_device() if device is None else int(device)
        self.dev_index = device
        self.device = torch.device("cuda", device)
        self.sequential = bool(sequential)
        self.legacy_mask = bool(legacy_mask)
        self.tape = None if tape_u is None else torch.as_tensor(np.asarray(tape_u, np.float64), device=self.device)
        self.tape_pos = 0
        self.now = 0                      # root turns seen (the clock cache_hit counts in)
        self.draw_ctr = 0                 # Philox counter: one value per kernel that draws
        self.tt_capacity = tt_capacity
        self.tt = None
        self.roll = None                  # rollout engine (sub-games; food_spawn_chance 0, game.py:268)
        self._bufs_B = -1
        self.verbose = False              # print the reference's per-epoch line (agent.py:57-58)
        # Range guard of the evaluation (set by the owner: an object with guard_ptr / guard_tripped() / guard_recover(), i.e. a
        # snake_engine.net.QNet, or None).  With a guard, evaluate() must not synchronise: the kernels of a tick that follow the
        # evaluation are gated on the device word guard_ptr (they do nothing when a convolution had to clamp an activation), the
        # host reads the word's mirror at the NEXT tick's existing read-back and, if it is set, lets the guard widen its scales
        # and runs that tick's evaluation and tail again -- a tick is played exactly once, from trusted values, and nothing of
        # an untrusted evaluation ever reaches the table (agent.py:161-223 knows no such event: nnet.v is exact there).
        self.guard = None
        self.guard_redos = 0
        self.stats = dict(net_evals=0, rollout_ticks=0, sim_steps=0, lookups=0)
        self._sim_steps_pending = []      # device counters of the sub-game tics, folded into stats at the turn's end

    # ---- buffers -------------------------------------------------------------------------------
    def _ensure(self, G, health_dec):
        par = min(PARALLEL, self.max_breadth)
        B = G * par
        if self.roll is None or self.roll.n_slots < B:
            self.roll = Engine(max(B, 64), self.H, self.W, self.S, health_dec, 0.0, seed=self.seed, device=self.dev_index)
        self.roll.set_params(health_dec=health_dec, food_spawn_chance=0.0)
        if self.tt is None:
            cap = self.tt_capacity
            if cap is None:     # room for every state of ~max_depth+2 root turns at a 50 % load factor
                per_turn = B * self.S * max(1, self.max_depth) * max(1, self.max_breadth // par)
                cap = min(1 << 30, _pow2_at_least(2 * per_turn * (self.max_depth + 2) // 2))
            self.tt = TranspositionTable(_pow2_at_least(cap), self.dev_index)
        if self._bufs_B != B:
            S, D, dev = self.S, max(1, self.max_depth), self.device
            m = B * S
            sub = torch.arange(B, dtype=torch.int32, device=dev).repeat_interleave(S)
            sn = torch.arange(S, dtype=torch.int32, device=dev).repeat(B)
            self.pairs = torch.stack([sub, sn], dim=1).contiguous()
            self.key = torch.empty((m, 2), dtype=torch.int64, device=dev)
            self.mask = torch.empty((m, 3), dtype=torch.uint8, device=dev)
            self.entry = torch.empty((m,), dtype=torch.int32, device=dev)
            self.is_new = torch.empty((m,), dtype=torch.uint8, device=dev)
            self.moves = torch.empty((m,), dtype=torch.uint8, device=dev)
            self.est = torch.empty((m,), dtype=torch.float32, device=dev)
            self.pmf = torch.empty((m, 3), dtype=torch.float32, device=dev)
            self.path_entry = torch.empty((m, D), dtype=torch.int32, device=dev)
            self.path_move = torch.empty((m, D), dtype=torch.uint8, device=dev)
            self.path_len = torch.zeros((m,), dtype=torch.int32, device=dev)
            self.alive_rows = torch.empty((B, S), dtype=torch.uint8, device=dev)
            self.row_active = torch.empty((m,), dtype=torch.uint8, device=dev)
            self.eval_pairs = torch.empty((m, 2), dtype=torch.int32, device=dev)
            self.eval_mask = torch.empty((m, 3), dtype=torch.uint8, device=dev)
            self.done = torch.empty((B,), dtype=torch.uint8, device=dev)
            self.rewards = torch.empty((m,), dtype=torch.int8, device=dev)
            self.cmp_idx = torch.empty((2, m), dtype=torch.int32, device=dev)       # ping-pong: a tick that is run again needs its own list
            self.cmp_cnt = torch.empty((1,), dtype=torch.int32, device=dev)
            self.cmp_scratch = torch.empty((self.L.snk_compact_scratch_elems(m),), dtype=torch.int32, device=dev)
            self.planes = None
            self._bufs_B = B
        return par, B

    def _planes(self, n):
        if self.planes is None or self.planes.shape[0] < n:
            cap = max(n, 1024)
            self.planes = torch.empty((cap, 2 * self.H - 1, 2 * self.W - 1, 3), dtype=torch.float32, device=self.device)
        return self.planes[:n]

    def _next_ctr(self):
        self.draw_ctr += 1
        return self.draw_ctr & 0xFFFFFFFF, (self.draw_ctr >> 32) & 0xFFFFFFFF

    def clear(self):
        """Agent.clear (agent.py:140-147)"""
        if self.tt is not None:
            self.tt.clear()

    # ---- one root turn: agent.py:25-99 -----------------------------------------------------------
    def search(self, root, live_slots, root_alive):
        """root: Engine holding the root games; live_slots int32 cuda [G]; root_alive uint8 cuda [G,S].
        Returns (V [G,S,3] float32 cuda -- cached_values[first_key] per root snake, moves [G,S] uint8 cuda)."""
        L, S, st = self.L, self.S, _stream()
        G = int(live_slots.shape[0])
        par, B = self._ensure(G, root.health_dec)
        m = B * S
        D = self.path_entry.shape[1]
        self.now += 1                                           # "for key in cache_hit: cache_hit[key] += 1" (agent.py:30-31)
        n_alive = root_alive.sum(dim=1, dtype=torch.int32)
        depth = self.max_depth - 2 * (n_alive - 2)              # agent.py:45
        sub_depth = depth.repeat_interleave(par)
        sub_depth_i32 = sub_depth.to(torch.int32).contiguous()
        n_ticks = max(1, int(depth.max().item()))
        epochs = self.max_breadth // par                        # agent.py:37
        tt = self.tt.h
        seq = int(self.sequential)
        if self.guard is not None:
            # a guard word left set by somebody else (a direct QNet.forward / activation_report that clamped, a search that raised)
            # would make the device skip the tail of every tick that evaluates nothing, unseen: start from a clear word
            self.guard.guard_post()
            torch.cuda.current_stream().synchronize()
            if self.guard.guard_tripped():
                try:
                    self.guard.guard_recover()
                except EngineError as e:
                    # a tower with nothing to widen (f16a) reports the saturation by raising -- but THIS flag was left by an earlier
                    # launch (range_flags has cleared it by now): no evaluation of this search has run yet, nothing to refuse
                    import warnings
                    warnings.warn(f"range flag left set by a launch before this search (cleared): {e}")
        for ep in range(epochs):
            t_epoch = time()
            root.clone_to(self.roll, src_slots=live_slots, n=G, fanout=par)       # game.subgame (agent.py:46-50)
            self.path_len.zero_()
            sub_active = torch.ones((B,), dtype=torch.uint8, device=self.device)
            sim_steps_dev = torch.zeros((), dtype=torch.int64, device=self.device)
            row_active = self.row_active
            guard = self.guard
            gate = guard.guard_ptr if guard is not None else 0
            pend = None                     # the last tick whose evaluation has not been verified yet (guarded runs only)

            def head(pp):
                """the tick up to its read-back: which rows are live, their masks and keys, table lookup, the list of new keys"""
                if _FOLD:
                    self.roll.observe(self.pairs, m, None, self.mask, self.key, legacy_mask=self.legacy_mask, sub_active=sub_active,
                                      row_active=row_active)
                else:
                    self.roll.alive(n=B, out=self.alive_rows)
                    check(L.snk_mcts_row_active(_ptr(self.alive_rows), _ptr(sub_active), B, S, _ptr(row_active), st))
                    self.roll.observe(self.pairs, m, None, self.mask, self.key, legacy_mask=self.legacy_mask)
                check(L.snk_tt_lookup_insert(tt, _ptr(self.key), _ptr(row_active), m, self.now, self.max_depth,
                                             _ptr(self.entry), _ptr(self.is_new), st))
                check(L.snk_compact_flags(_ptr(self.is_new), m, _ptr(self.cmp_idx[pp]), _ptr(self.cmp_cnt),
                                          _ptr(self.cmp_scratch), st))
                return int(self.cmp_cnt.item())                 # the one host read-back of the tick

            def tail(t):
                """evaluation of the new keys + everything that follows it (t: what the tick needs to be run again unchanged)"""
                if t["n_eval"]:
                    idx = self.cmp_idx[t["pp"]][:t["n_eval"]]
                    eval_pairs, eval_mask = self.eval_pairs[:t["n_eval"]], self.eval_mask[:t["n_eval"]]
                    planes = self._planes(t["n_eval"])
                    if _FOLD:       # row i of the batch observes pairs[idx[i]]; its obstacle mask comes out of the same launch
                        self.roll.observe(self.pairs, t["n_eval"], planes, eval_mask, None, legacy_mask=self.legacy_mask, index=idx)
                    else:
                        check(L.snk_mcts_gather_rows(_ptr(idx), t["n_eval"], _ptr(self.pairs), _ptr(self.mask), _ptr(eval_pairs), _ptr(eval_mask), st))
                        self.roll.observe(eval_pairs, t["n_eval"], planes, None, None)
                    q = self.evaluate(planes, eval_mask)            # nnet.v(all_states) (agent.py:190)
                    check(L.snk_tt_set_priors(tt, _ptr(self.entry), _ptr(idx), t["n_eval"], _ptr(q.contiguous()), gate or None, st))
                rank = None
                if self.tape is not None:
                    rank = (torch.cumsum(row_active, 0, dtype=torch.int32) - 1).contiguous()
                check(L.snk_mcts_select(tt, _ptr(self.entry), m, self.base, _ptr(self.tape), _ptr(rank), t["tape_pos"],
                                        self.seed, t["c0"], t["c1"], _ptr(self.moves), _ptr(self.est), _ptr(self.pmf),
                                        _ptr(self.path_entry), _ptr(self.path_move), _ptr(self.path_len), D, gate or None, st))
                check(L.snk_mcts_backup(tt, _ptr(self.entry), m, _ptr(self.est), _ptr(self.pmf), _ptr(self.path_entry),
                                        _ptr(self.path_move), _ptr(self.path_len), D, seq, gate or None, st))
                # tic every live sub-game (mp_game_runner.py:104-106) in one launch over all B slots; sub-games retired by
                # their depth cap must not move: their `active` flag is 0 and the kernel skips them (no host read-back)
                self.roll.step_active(sub_active, self.moves, B, done=self.done, skip=gate)
                # the sub-games that moved are counted, then the finished ones and those at their depth cap retire
                # (mp_game_runner.py:108-113): one launch (this was a chain of eight tensor expressions)
                check(L.snk_mcts_retire(_ptr(sub_active), _ptr(self.done), _ptr(sub_depth_i32), t["tick"], B, _ptr(sim_steps_dev),
                                        gate or None, st))

            def verify():
                """called right after a host synchronisation: did the pending tick's evaluation clamp?  Then the device skipped
                that tick's tail (the gate): widen the scales and run evaluation + tail again, until an evaluation holds."""
                nonlocal pend
                redone = False
                while pend is not None and guard.guard_tripped():
                    if pend["redos"] >= 4:
                        # the tick's new entries have no priors and the word is set: leave neither behind for a caller that catches this
                        guard.guard_recover()
                        self.clear()
                        raise EngineError("the Q-net's range guard tripped four times on the same batch; use SNK_CONV_ALGO=winograd")
                    try:
                        guard.guard_recover()
                    except EngineError:
                        self.clear()       # the tick's new entries have no priors: a caller that catches this must not keep them
                        raise
                    pend["redos"] += 1
                    self.guard_redos += 1
                    tail(pend)
                    torch.cuda.current_stream().synchronize()      # a rare path: wait for the repeated evaluation's verdict
                    redone = True
                pend = None
                return redone

            pp = 0
            for tick in range(1, n_ticks + 1):
                n_eval = head(pp)
                if guard is not None and verify():
                    n_eval = head(pp)      # the previous tick has only now been played: this tick's rows are different ones
                n_rows = int(row_active.sum().item()) if self.tape is not None else 0
                c0, c1 = self._next_ctr()
                t = dict(pp=pp, n_eval=n_eval, tick=tick, c0=c0, c1=c1, tape_pos=self.tape_pos, redos=0)
                tail(t)
                self.tape_pos += n_rows
                self.stats["net_evals"] += n_eval
                self.stats["rollout_ticks"] += 1
                if guard is not None and n_eval:
                    pend = t
                pp ^= 1
            if pend is not None:             # the epoch's last evaluation is verified before its terminal back-up builds on it
                torch.cuda.current_stream().synchronize()
                verify()
            self._sim_steps_pending.append(sim_steps_dev)
            check(L.snk_engine_rewards(self.roll.h, None, B, _ptr(self.rewards), st))
            check(L.snk_mcts_terminal_backup(tt, _ptr(self.rewards), m, _ptr(self.path_entry), _ptr(self.path_move),
                                             _ptr(self.path_len), D, seq, st))   # agent.py:60-72
            if self.verbose and self.training:
                print("MCTS epoch finished. Time spent:", time() - t_epoch)
            if ep + 1 < epochs:
                # paths die with the epoch, so the table may be re-hashed here: keep the load factor below 1/2 even
                # when one root turn inserts more keys than the table was sized for (nothing is evicted: max_age = inf)
                cap, occ, ovf = self.tt.status()
                if ovf:
                    raise EngineError("transposition table overflowed inside an epoch; raise tt_capacity")
                if occ * 2 > cap:
                    self.tt.rebuild(cap * 2, self.now, 1 << 30)
                    tt = self.tt.h
        # V[i] = cached_values[first_key] (agent.py:74-87): the root observation's entry, from clone 0 of each game
        first = self.path_entry.view(G, par, S, D)[:, 0, :, 0].contiguous().reshape(-1)
        root_rows_alive = root_alive.reshape(-1).contiguous()
        first = torch.where(root_rows_alive.bool(), first, torch.full_like(first, -1))
        V = torch.empty((G * S, 3), dtype=torch.float32, device=self.device)
        check(L.snk_tt_read_q(tt, _ptr(first), 1, G * S, _ptr(V), st))
        moves = torch.empty((G * S,), dtype=torch.uint8, device=self.device)
        rank, n_rows = None, 0
        if self.tape is not None and self.training:
            rank = (torch.cumsum(root_rows_alive, 0, dtype=torch.int32) - 1).contiguous()
            n_rows = int(root_rows_alive.sum().item())
        c0, c1 = self._next_ctr()
        check(L.snk_mcts_root_moves(_ptr(V), _ptr(root_rows_alive), G * S, self.base, int(self.training),
                                    _ptr(self.tape) if self.training else None, _ptr(rank), self.tape_pos, self.seed, c0, c1,
                                    _ptr(moves), st))
        self.tape_pos += n_rows
        return V.view(G, S, 3), moves.view(G, S)

    def end_of_turn(self):
        """RAM recycle (agent.py:101-110).  Stale entries already read as misses; here they are dropped
        physically when the table is more than half full (and the table doubles if the survivors still
        fill more than half of it)."""
        cap, occ, ovf = self.tt.status()
        if self._sim_steps_pending:
            self.stats["sim_steps"] += int(torch.stack(self._sim_steps_pending).sum().item())
            self._sim_steps_pending = []
        if ovf:
            raise EngineError("transposition table overflowed; raise tt_capacity")
        if occ * 2 > cap:
            self.tt.rebuild(cap, self.now, self.max_depth)
            cap, occ, _ = self.tt.status()
            if occ * 2 > cap:
                self.tt.rebuild(cap * 2, self.now, self.max_depth)
