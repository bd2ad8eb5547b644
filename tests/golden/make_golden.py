#!/usr/bin/env python3
"""Golden-vector generator: runs the UNMODIFIED reference (imported from
/root/reference/code) in the build container and records inputs + expected outputs
of the self-play hot path as small .npz fixtures next to this file.

Run (build container only; the reference never travels to the GPU box):

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python3 -W ignore /root/repo/tests/golden/make_golden.py

What is recorded:
  tic_<cfg>.npz    seeded game trajectories: compact pre/post states, moves, the
                   food-spawn tape, the reference's empty set, results, counters
                   (Game.tic, game.py:87-205; Game.__init__, game.py:13-61)
  corner.npz       hand-built single-tick corner cases of Appendix A
  states_<cfg>.npz Game.get_states() bytes (raw planes for a subset, 128-bit
                   blake2b for all), obstacle masks (alpha_nnet.py:63-76)
  tables.npz       Agent.softermax / Agent.argmaxs tables (agent.py:114-137)
  mcts_tiny.npz    2-game MCTS self-play with a deterministic stub net and taped
                   uniforms (Agent.make_moves / MCTSAgent.make_moves); mcts_tiny_greedybase,
                   mcts_7x7x2, mcts_9x9x3, mcts_19x19x8: the same at other settings / geometries
  runner.npz       MPGameRunner.run with a taped-move agent: rewards + counters
  pit.npz          pit_mp_game_runner.MPGameRunner.run with two stub nets (1v3, 2v2, 3v1):
                   start boards, spawn tape, moves, winner indices
  replay.npz       the replay.rep text one-game MPGameRunner.run writes (Game.draw, two
                   boards per tick) with the start board, moves and spawn tape behind it
  trainer.npz      AlphaSnakeZeroTrainer.train (the reference's own module) for 3 x 2 generations
  pit_script.npz   pit.py (the reference's own script) over three generations of stub nets
  net_graph.npz    what AlphaNNet.__init__ / copy_and_compile / train / save build and call
                   (layers, arguments, wiring, optimizer, schedule, loss), as JSON

Only the RNG *bindings* inside the imported modules are wrapped (utils.game.random /
choice / sample and utils.agent.choice are plain ``from x import y`` names,
game.py:1, agent.py:2) so draws can be taped; no reference file is edited or copied.
"""
import hashlib
import os
import random as pyrandom
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, "/root/reference/code")
sys.dont_write_bytecode = True

import utils.game as G          # noqa: E402  (reference)
import utils.agent as A         # noqa: E402  (reference)
import utils.mp_game_runner as R  # noqa: E402  (reference)

from oracle.obs_key import obs_key, obstacle_mask, StubNet  # noqa: E402


# ----------------------------------------------------------------------------- RNG taps
class Ctx:
    phase = None          # 'init' | 'tic' | None
    in_subgame = False
    spawn_cell = None     # (y, x) chosen in the current tic
    spawn_empty = None    # sorted list of the reference's empty_positions at spawn time
    init_tape = None      # dict for the current root Game.__init__


_orig_random, _orig_choice, _orig_sample = G.random, G.choice, G.sample


def tap_random():
    return _orig_random()


def tap_choice(seq):
    r = _orig_choice(seq)
    if Ctx.in_subgame:
        return r
    if Ctx.phase == "tic":
        Ctx.spawn_cell = r
        Ctx.spawn_empty = sorted(seq)
    elif Ctx.phase == "init" and Ctx.init_tape is not None:
        if isinstance(r, int):
            Ctx.init_tape["dirs"].append(r)
        else:
            Ctx.init_tape["food_choice"].append(list(seq).index(r))
    return r


def tap_sample(pop, k):
    r = _orig_sample(pop, k)
    if not Ctx.in_subgame and Ctx.phase == "init" and Ctx.init_tape is not None:
        Ctx.init_tape["positions"] = [list(pop).index(p) for p in r]
    return r


G.random, G.choice, G.sample = tap_random, tap_choice, tap_sample

_orig_init = G.Game.__init__
_orig_tic = G.Game.tic
_orig_subgame = G.Game.subgame


def _init(self, *a, **k):
    prev = Ctx.phase
    Ctx.phase = "init"
    if not Ctx.in_subgame:
        Ctx.init_tape = {"dirs": [], "food_choice": [], "positions": None}
    try:
        _orig_init(self, *a, **k)
    finally:
        Ctx.phase = prev
    if not Ctx.in_subgame:
        self._init_tape = Ctx.init_tape
        Ctx.init_tape = None


def _tic(self, moves, show=False):
    prev = Ctx.phase
    Ctx.phase = "tic"
    Ctx.spawn_cell = None
    Ctx.spawn_empty = None
    try:
        r = _orig_tic(self, moves, show)
    finally:
        Ctx.phase = prev
    self._last_spawn = Ctx.spawn_cell
    self._last_spawn_empty = Ctx.spawn_empty
    return r


def _subgame(self, sid):
    prev = Ctx.in_subgame
    Ctx.in_subgame = True
    try:
        return _orig_subgame(self, sid)
    finally:
        Ctx.in_subgame = prev


# class-level wrapping of the imported classes (the reference files are untouched)
G.Game.__init__ = _init
G.Game.tic = _tic
G.Game.subgame = _subgame


# ----------------------------------------------------------------------------- snapshots
def snake_nodes(snake):
    out = []
    n = snake.head
    while n:
        out.append(n.position)
        n = n.next_node
    return out


def snapshot(game, L):
    S, H, W = game.snake_cnt, game.height, game.width
    alive = np.zeros(S, np.uint8)
    health = np.zeros(S, np.int16)
    length = np.zeros(S, np.int16)
    dirs = np.zeros(S, np.uint8)
    nodes = np.full((S, L), -1, np.int16)
    for s in game.snakes:
        alive[s.id] = 1
        health[s.id] = s.health
        length[s.id] = s.length
        dirs[s.id] = game.last_moves[s.id]
        ns = snake_nodes(s)
        assert len(ns) == s.length, (len(ns), s.length)
        assert len(ns) <= L
        for i, (y, x) in enumerate(ns):
            assert 0 <= y < H and 0 <= x < W
            nodes[s.id, i] = y * W + x
    food = np.zeros(H * W, np.uint8)
    for (y, x) in game.food:
        food[y * W + x] = 1
    empty = np.zeros(H * W, np.uint8)
    for (y, x) in game.empty_positions:
        empty[y * W + x] = 1
    rewards = np.array([0 if r is None else int(r) for r in game.rewards], np.int8)
    counters = np.array([game.wall_collision, game.body_collision, game.head_collision,
                         game.starvation, game.food_eaten, game.game_length], np.int32)
    return dict(alive=alive, health=health, length=length, dir=dirs, nodes=nodes,
                food=food, empty=empty, rewards=rewards, counters=counters)


def legal_moves(game, snake):
    """relative moves whose target is on board and not a body/head cell (for long random games)."""
    out = []
    hy, hx = snake.head.position
    last = game.last_moves[snake.id]
    for m in (0, 1, 2):
        d = (m + last - 1) % 4
        y, x = hy + (d == 2) - (d == 0), hx + (d == 1) - (d == 3)
        if 0 <= y < game.height and 0 <= x < game.width and (y, x) not in game.bodies and (y, x) not in game.heads:
            out.append(m)
    return out


class StateSink:
    """collects get_states() outputs for states_<cfg>.npz"""

    def __init__(self, raw_every):
        self.raw_every = raw_every
        self.n = 0
        self.digest, self.key, self.mask, self.mask_legacy = [], [], [], []
        self.state_index, self.snake_id = [], []
        self.raw, self.raw_index = [], []

    def add(self, game, state_idx):
        states = game.get_states()
        for snake, st in zip(game.snakes, states):
            b = st.tobytes()
            self.digest.append(np.frombuffer(hashlib.blake2b(b, digest_size=16).digest(), np.uint8))
            self.key.append(obs_key(st)[0])
            self.mask.append(obstacle_mask(st)[0])
            self.mask_legacy.append(obstacle_mask(st, legacy=True)[0])
            self.state_index.append(state_idx)
            self.snake_id.append(snake.id)
            if self.n % self.raw_every == 0:
                self.raw.append(np.ascontiguousarray(st))
                self.raw_index.append(self.n)
            self.n += 1

    def save(self, path):
        np.savez_compressed(
            path,
            digest=np.array(self.digest, np.uint8), key=np.array(self.key, np.uint64),
            mask=np.array(self.mask, np.uint8), mask_legacy=np.array(self.mask_legacy, np.uint8),
            state_index=np.array(self.state_index, np.int32), snake_id=np.array(self.snake_id, np.uint8),
            raw=np.array(self.raw, np.float32), raw_index=np.array(self.raw_index, np.int32))


# ----------------------------------------------------------------------------- tic traces
def record_trajectories(tag, H, W, S, health_dec, n_games, seed, p_legal, max_ticks, raw_every):
    L = H * W + 2
    pyrandom.seed(seed)
    np.random.seed(seed)
    mover = pyrandom.Random(seed * 7919 + 13)
    st_list, moves_l, spawn_l, spawn_empty_l, spawn_valid_l, done_l = [], [], [], [], [], []
    ptr = [0]
    init_pos, init_dirs, init_food = [], [], []
    sink = StateSink(raw_every)
    for g in range(n_games):
        game = G.Game(g, H, W, S, health_dec)
        tape = game._init_tape
        init_pos.append(tape["positions"])
        init_dirs.append(tape["dirs"])
        init_food.append(tape["food_choice"])
        st_list.append(snapshot(game, L))
        sink.add(game, len(st_list) - 1)
        t = 0
        while True:
            mv_dense = np.full(S, 255, np.uint8)
            mv = []
            for s in game.snakes:
                lm = legal_moves(game, s)
                if lm and mover.random() < p_legal:
                    m = mover.choice(lm)
                else:
                    m = mover.choice((0, 1, 2))
                mv.append(m)
                mv_dense[s.id] = m
            res = game.tic(mv)
            t += 1
            moves_l.append(mv_dense)
            sp = game._last_spawn
            spawn_l.append(-1 if sp is None else sp[0] * W + sp[1])
            em = np.zeros(H * W, np.uint8)
            if game._last_spawn_empty is not None:
                for (y, x) in game._last_spawn_empty:
                    em[y * W + x] = 1
            spawn_empty_l.append(em)
            spawn_valid_l.append(game._last_spawn_empty is not None)
            done_l.append(res != 0)
            st_list.append(snapshot(game, L))
            if res == 0:
                sink.add(game, len(st_list) - 1)
            if res != 0 or t >= max_ticks:
                break
        ptr.append(len(st_list))
    out = {k: np.stack([s[k] for s in st_list]) for k in st_list[0]}
    # trim node columns to the longest snake seen (+1) to keep the fixture small
    used = int((out["nodes"] >= 0).sum(axis=2).max()) + 1
    out["nodes"] = out["nodes"][:, :, :used]
    np.savez_compressed(
        os.path.join(HERE, f"tic_{tag}.npz"),
        H=H, W=W, S=S, health_dec=health_dec, food_chance=0.15, ptr=np.array(ptr, np.int32),
        moves=np.array(moves_l, np.uint8), spawn=np.array(spawn_l, np.int16),
        spawn_empty=np.packbits(np.array(spawn_empty_l, np.uint8), axis=1),
        spawn_empty_valid=np.array(spawn_valid_l, np.uint8), done=np.array(done_l, np.uint8),
        init_positions=np.array(init_pos, np.uint8), init_dirs=np.array(init_dirs, np.uint8),
        init_food=np.array(init_food, np.uint8),
        **{"st_" + k: v for k, v in out.items()})
    sink.save(os.path.join(HERE, f"states_{tag}.npz"))
    print(f"[{tag}] games={n_games} states={len(st_list)} ticks={len(moves_l)} obs={sink.n} raw={len(sink.raw)}")


# ----------------------------------------------------------------------------- corner cases
def build_game(H, W, snakes, food, health_dec=1, chance=0.0, S=None):
    """snakes: list of (id, health, [(y,x) head..tail], last_dir). Mirrors what Game.subgame does
    (game.py:266-276) to install a hand-built position into a reference Game object."""
    S = S or len(snakes)
    Ctx.in_subgame = True
    try:
        game = G.Game(0, H, W, S, health_dec, chance)
    finally:
        Ctx.in_subgame = False
    game.snakes = []
    game.last_moves = {i: 0 for i in range(S)}
    present = set()
    for (sid, health, body, d) in snakes:
        s = G.Snake(sid, health, list(body))
        game.snakes.append(s)
        game.last_moves[sid] = d
        present.add(sid)
    game.rewards = [None if i in present else -1.0 for i in range(S)]
    game.food = set(food)
    game.heads = {}
    for s in game.snakes:
        game.heads.setdefault(s.head.position, set()).add(s)
    game.bodies = {b for s in game.snakes for b in s}
    game.empty_positions = {(y, x) for y in range(H) for x in range(W)} - set(game.heads) - game.bodies - game.food
    return game


def corner_cases():
    H = W = 11
    C = []
    # dirs: 0 up, 1 right, 2 down, 3 left ; relative move 1 = straight
    # 1 equal-length head-on: both die (head_collision 2)
    C.append(("equal_head_on", H, W, [(0, 90, [(5, 3), (5, 2), (5, 1)], 1), (1, 90, [(5, 5), (5, 6), (5, 7)], 3)], [(0, 0)], [1, 1], 1, 0.0))
    # 2 head-on onto food: lower id eats first, grows, survives (game.py:121-127 before 156-161)
    C.append(("head_on_food", H, W, [(0, 50, [(5, 3), (5, 2), (5, 1)], 1), (1, 50, [(5, 5), (5, 6), (5, 7)], 3)], [(5, 4)], [1, 1], 1, 0.0))
    # 3 three heads in one cell, lengths 3,4,4 + a bystander
    C.append(("three_heads", H, W, [(0, 80, [(4, 5), (3, 5), (2, 5)], 2), (1, 80, [(5, 4), (5, 3), (5, 2), (5, 1)], 1),
                                     (2, 80, [(5, 6), (5, 7), (5, 8), (5, 9)], 3), (3, 80, [(9, 9), (9, 8), (9, 7)], 1)], [(0, 0)], [1, 1, 1, 1], 1, 0.0))
    # 4 three heads, lengths 3,4,5: longest survives
    C.append(("three_heads_longest", H, W, [(0, 80, [(4, 5), (3, 5), (2, 5)], 2), (1, 80, [(5, 4), (5, 3), (5, 2), (5, 1)], 1),
                                             (2, 80, [(5, 6), (5, 7), (5, 8), (5, 9), (5, 10)], 3)], [(0, 0)], [1, 1, 1], 1, 0.0))
    # 5 move into a vacated tail (own loop): legal
    C.append(("into_vacated_tail", H, W, [(0, 70, [(5, 5), (5, 6), (6, 6), (6, 5)], 3), (1, 70, [(1, 1), (1, 2), (1, 3)], 3)], [(0, 5)], [0, 1], 1, 0.0))
    # 6 move into another snake's vacated tail: legal
    C.append(("into_other_vacated_tail", H, W, [(0, 70, [(5, 5), (5, 4), (5, 3)], 1), (1, 70, [(7, 6), (6, 6), (5, 6)], 2)], [(0, 5)], [1, 1], 1, 0.0))
    # 7 move into a stacked tail (turn after eating): dies by body collision
    C.append(("into_stacked_tail", H, W, [(0, 70, [(5, 5), (5, 4), (5, 3)], 1), (1, 100, [(7, 6), (6, 6), (5, 6), (5, 6)], 2)], [(0, 5)], [1, 1], 1, 0.0))
    # 8 start-of-game stacked snake (3 nodes on one cell), neighbour walks into it
    C.append(("into_start_stack", H, W, [(0, 100, [(1, 1), (1, 1), (1, 1)], 0), (1, 100, [(1, 2), (1, 3), (1, 4)], 3)], [(5, 5)], [2, 1], 1, 0.0))
    # 9 wall + starvation in the same tick: counted as wall (if/elif chain)
    C.append(("wall_and_starve", H, W, [(0, 1, [(0, 5), (1, 5), (2, 5)], 0), (1, 50, [(9, 9), (9, 8), (9, 7)], 1)], [(5, 5)], [1, 1], 1, 0.0))
    # 10 starvation masked by winning a head-on (health 1 -> 0 survives this tick)
    C.append(("starve_masked_by_head_on", H, W, [(0, 1, [(5, 3), (5, 2), (5, 1), (5, 0)], 1), (1, 50, [(5, 5), (5, 6), (5, 7)], 3),
                                                  (2, 50, [(9, 9), (9, 8), (9, 7)], 1)], [(0, 0)], [1, 1, 0], 1, 0.0))
    # 11 ... and it starves on the next tick (2-tick case)
    C.append(("starve_after_head_on", H, W, [(0, 1, [(5, 3), (5, 2), (5, 1), (5, 0)], 1), (1, 50, [(5, 5), (5, 6), (5, 7)], 3),
                                              (2, 50, [(9, 9), (9, 8), (9, 7)], 1)], [(0, 0)], [[1, 1, 0], [1, 1]], 1, 0.0))
    # 12 last two die together: draw, all -1
    C.append(("draw_all_dead", H, W, [(0, 50, [(0, 3), (1, 3), (2, 3)], 0), (1, 50, [(10, 3), (9, 3), (8, 3)], 2)], [(5, 5)], [1, 1], 1, 0.0))
    # 13 plain starvation with health_dec 9
    C.append(("starve_dec9", H, W, [(0, 9, [(5, 3), (5, 2), (5, 1)], 1), (1, 10, [(7, 5), (7, 6), (7, 7)], 3)], [(0, 0)], [1, 1], 9, 0.0))
    # 14 eat then wall: eating restores health, grows, next tick dies in wall (2 ticks; two bystanders keep the game alive)
    C.append(("eat_then_wall", H, W, [(0, 5, [(1, 5), (2, 5), (3, 5)], 0), (1, 50, [(9, 5), (9, 4), (9, 3)], 1),
                                       (2, 50, [(5, 1), (6, 1), (7, 1)], 0)], [(0, 5)], [[1, 1, 1], [1, 1, 1]], 1, 0.0))
    # 15 body collision with a snake that dies in the same tick (its body still counts)
    C.append(("hit_dying_body", H, W, [(0, 50, [(0, 3), (1, 3), (2, 3)], 0), (1, 50, [(1, 2), (1, 1), (1, 0)], 1)], [(5, 5)], [1, 1], 1, 0.0))
    # 16 no empty cell for food: 5x5 board fully covered by two snakes with stacked tails (nothing is
    #    vacated), no food, chance 1.0 -> choice(()) raises IndexError, swallowed (game.py:132-138)
    def serp(i):
        r = i // 5
        return (r, i % 5 if r % 2 == 0 else 4 - i % 5)
    a = [serp(i) for i in range(13)]
    b = [serp(i) for i in range(24, 12, -1)]
    C.append(("no_empty_cell", 5, 5, [(0, 50, a + [a[-1]], 3), (1, 50, b + [b[-1]], 1)], [], [1, 1], 1, 1.0))
    # 17 forced spawn when no food left (chance small but food empty => no random() draw)
    C.append(("spawn_when_no_food", H, W, [(0, 50, [(5, 3), (5, 2), (5, 1)], 1), (1, 50, [(7, 5), (7, 6), (7, 7)], 3)], [], [1, 1], 1, 0.15))
    # 18 dead snake ids missing from the list (ids 1 and 3 alive out of 4)
    C.append(("sparse_ids", H, W, [(1, 60, [(5, 3), (5, 2), (5, 1)], 1), (3, 60, [(7, 5), (7, 6), (7, 7)], 3)], [(5, 4)], [1, 1], 1, 0.0, 4))
    # 19 head-on: survivor wins and other snakes continue (4 snakes)
    C.append(("head_on_four", H, W, [(0, 60, [(5, 3), (5, 2), (5, 1), (4, 1)], 1), (1, 60, [(5, 5), (5, 6), (5, 7)], 3),
                                      (2, 60, [(9, 9), (9, 8), (9, 7)], 1), (3, 60, [(1, 9), (1, 8), (1, 7)], 1)], [(0, 0)], [1, 1, 0, 2], 1, 0.0))
    # 20 contested food, both equal length 3 -> id0 eats (len 4) and survives, id1 dies; id order reversed in space
    C.append(("contested_food_swap", H, W, [(0, 50, [(5, 5), (5, 6), (5, 7)], 3), (1, 50, [(5, 3), (5, 2), (5, 1)], 1)], [(5, 4)], [1, 1], 1, 0.0))
    return C


def record_corner_cases():
    pyrandom.seed(4242)
    names, recs = [], []
    for case in corner_cases():
        name, H, W, snakes, food, moves, hd, chance = case[:8]
        S = case[8] if len(case) > 8 else len(snakes)
        game = build_game(H, W, snakes, food, hd, chance, S)
        L = H * W + 2
        ticks = moves if isinstance(moves[0], list) else [moves]
        states = [snapshot(game, L)]
        obs = [np.array([np.ascontiguousarray(s) for s in game.get_states()], np.float32)]
        mvs, spawns, results = [], [], []
        for mv in ticks:
            dense = np.full(S, 255, np.uint8)
            for s, m in zip(game.snakes, mv):
                dense[s.id] = m
            res = game.tic(list(mv))
            mvs.append(dense)
            sp = game._last_spawn
            spawns.append(-1 if sp is None else sp[0] * W + sp[1])
            results.append(0 if res == 0 else 1)
            states.append(snapshot(game, L))
            if game.snakes:
                obs.append(np.array([np.ascontiguousarray(s) for s in game.get_states()], np.float32))
            else:
                obs.append(np.zeros((0, 2 * H - 1, 2 * W - 1, 3), np.float32))
        names.append(name)
        recs.append(dict(H=H, W=W, S=S, hd=hd, chance=chance, states=states, moves=mvs, spawns=spawns,
                         results=results, obs=obs))
        print(f"[corner] {name}: rewards={states[-1]['rewards'].tolist()} counters={states[-1]['counters'].tolist()}")
    flat = {"names": np.array(names)}
    for i, r in enumerate(recs):
        p = f"c{i}_"
        flat[p + "meta"] = np.array([r["H"], r["W"], r["S"], r["hd"]], np.int32)
        flat[p + "chance"] = np.float64(r["chance"])
        used = max(int((s["nodes"] >= 0).sum(axis=1).max()) for s in r["states"]) + 1
        for k in r["states"][0]:
            arr = np.stack([s[k] for s in r["states"]])
            if k == "nodes":
                arr = arr[:, :, :used]
            flat[p + "st_" + k] = arr
        flat[p + "moves"] = np.array(r["moves"], np.uint8)
        flat[p + "spawn"] = np.array(r["spawns"], np.int16)
        flat[p + "done"] = np.array(r["results"], np.uint8)
        for t, o in enumerate(r["obs"]):
            flat[p + f"obs{t}"] = o
    np.savez_compressed(os.path.join(HERE, "corner.npz"), **flat)


# ----------------------------------------------------------------------------- softermax / argmaxs
def record_tables():
    rng = np.random.RandomState(7)
    zs = []
    grid = np.array([-1.0, -0.9, -0.5, -0.1, 0.0, 0.1, 0.5, 0.9, 0.999], np.float32)
    for a in grid:
        for b in grid:
            for c in grid:
                zs.append((a, b, c))
    zs += [tuple(v) for v in np.tanh(rng.randn(400, 3)).astype(np.float32)]
    zs = np.array(zs, np.float32)
    out = {"z": zs}
    for base in (2, 3, 10, 100):
        ag = A.Agent(None, base)
        out[f"pmf_b{base}"] = np.array([ag.softermax(z) for z in zs], np.float32)
    ag = A.Agent(None)
    zz = np.concatenate([zs, np.array([[0.5, 0.5, 0.5], [0.5, 0.5, 0.1], [0.1, 0.5, 0.5], [0.5, 0.1, 0.5],
                                       [-1, -1, -1], [0.2, 0.2, 0.3]], np.float32)])
    out["argmax_z"] = zz
    out["argmax"] = np.array(ag.argmaxs(list(zz)), np.int8)
    np.savez_compressed(os.path.join(HERE, "tables.npz"), **out)
    print("[tables] softermax rows", len(zs), "argmax rows", len(zz))


# ----------------------------------------------------------------------------- tiny MCTS
class UniformTape:
    """stands in for numpy.random.choice([0,1,2], p=pmf) (agent.py:91,205): same algorithm
    (cdf = cumsum(p)/cdf[-1]; searchsorted(cdf, u, 'right')) but u comes from a recorded tape."""

    def __init__(self, seed):
        self.rs = np.random.RandomState(seed)
        self.u, self.pmf, self.out = [], [], []

    def __call__(self, a, p=None):
        p64 = np.asarray(p, dtype=np.float64)
        cdf = p64.cumsum()
        cdf /= cdf[-1]
        while True:
            u = self.rs.random_sample()
            # keep every draw >= 1e-4 away from a cdf edge so 1-ulp libm differences cannot flip a move
            if np.all(np.abs(cdf - u) > 1e-4):
                break
        k = int(cdf.searchsorted(u, side="right"))
        self.u.append(u)
        self.pmf.append(np.asarray(p, np.float32))
        self.out.append(k)
        return a[k]


def record_mcts_tiny(tag, n_games, breadth, depth, base, seed, max_turns, H=11, W=11, S=4, hd=1):
    pyrandom.seed(seed)
    np.random.seed(seed)
    tape = UniformTape(seed + 1)
    A.choice = tape
    net = StubNet()
    alice = A.Agent(net, base, True, depth, breadth)
    runner = R.MPGameRunner(H, W, S, hd, n_games)
    L = H * W + 2
    games = runner.games
    init_states = [snapshot(games[g], L) for g in range(n_games)]
    # re-implementation of the loop body of MPGameRunner.run (mp_game_runner.py:31-66) calling the
    # reference objects, so per-turn data can be recorded; a second, independent run through the
    # reference's own run() is recorded in runner.npz.
    turn_ids, turn_moves, turn_V, turn_spawn, turn_evals, turn_cache, turn_tape_pos = [], [], [], [], [], [], []
    rewards = [None] * n_games
    live = dict(games)
    turn = 0
    while live and turn < max_turns:
        turn += 1
        ids = []
        for gid in live:
            ids += live[gid].get_ids()
        n_calls0 = len(net.calls)
        nv0 = len(alice.values)
        moves = alice.make_moves(live, ids)
        V = np.array(alice.values[nv0:], np.float32)   # snapshot now (the reference keeps live aliases)
        turn_ids.append(np.array(ids, np.int32))
        turn_moves.append(np.array(moves, np.uint8))
        turn_V.append(V)
        turn_evals.append(int(sum(net.calls[n_calls0:])))
        turn_cache.append(len(alice.cached_values))
        turn_tape_pos.append(len(tape.u))
        mfg = {gid: [] for gid in live}
        for i, m in enumerate(moves):
            mfg[ids[i][0]].append(m)
        kills = []
        sp_row = np.full(n_games, -2, np.int16)
        for gid in live:
            res = live[gid].tic(mfg[gid])
            sp = live[gid]._last_spawn
            sp_row[gid] = -1 if sp is None else sp[0] * W + sp[1]
            if res != 0:
                rewards[gid] = res
                kills.append(gid)
        turn_spawn.append(sp_row)
        for gid in kills:
            del live[gid]
    final_states = [snapshot(runner.games[g] if g in runner.games else games[g], L) for g in range(n_games)] if False else None
    rec = np.array([np.ascontiguousarray(r) for r in alice.records], np.float32)
    val_final = np.array(alice.values, np.float32)
    flat = dict(H=H, W=W, S=S, hd=hd, n_games=n_games, breadth=breadth, depth=depth, base=base,
                tape_u=np.array(tape.u, np.float64), tape_out=np.array(tape.out, np.uint8),
                tape_pmf=np.array(tape.pmf, np.float32),
                records_digest=np.array([np.frombuffer(hashlib.blake2b(r.tobytes(), digest_size=16).digest(), np.uint8)
                                         for r in rec], np.uint8),
                values_final=val_final,
                turn_evals=np.array(turn_evals, np.int32), turn_cache=np.array(turn_cache, np.int32),
                turn_tape_pos=np.array(turn_tape_pos, np.int64),
                turn_spawn=np.array(turn_spawn, np.int16),
                rewards=np.array([[0 if (r is None or x is None) else int(x) for x in (r or [None] * S)] for r in rewards], np.int8))
    for k in init_states[0]:
        arr = np.stack([s[k] for s in init_states])
        if k == "nodes":
            arr = arr[:, :, :4]
        flat["init_" + k] = arr
    for t in range(len(turn_ids)):
        flat[f"t{t}_ids"] = turn_ids[t]
        flat[f"t{t}_moves"] = turn_moves[t]
        flat[f"t{t}_V"] = turn_V[t]
    flat["n_turns"] = len(turn_ids)
    np.savez_compressed(os.path.join(HERE, f"mcts_{tag}.npz"), **flat)
    print(f"[mcts_{tag}] turns={len(turn_ids)} draws={len(tape.u)} evals={sum(turn_evals)} records={len(rec)}")


# ----------------------------------------------------------------------------- runner with taped moves
class TapedMoveAgent:
    def __init__(self, seed):
        self.rng = pyrandom.Random(seed)
        self.log = []

    def make_moves(self, games, ids):
        moves = []
        for (gid, sid) in ids:
            game = games[gid]
            snake = [s for s in game.snakes if s.id == sid][0]
            lm = legal_moves(game, snake)
            m = self.rng.choice(lm) if lm and self.rng.random() < 0.9 else self.rng.choice((0, 1, 2))
            moves.append(m)
        self.log.append((list(ids), list(moves)))
        return moves


def record_runner(n_games=6, seed=99, H=11, W=11, S=4, hd=3):
    import contextlib
    import io
    pyrandom.seed(seed)
    np.random.seed(seed)
    runner = R.MPGameRunner(H, W, S, hd, n_games)
    L = H * W + 2
    init_states = [snapshot(runner.games[g], L) for g in range(n_games)]
    games_ref = dict(runner.games)
    agent = TapedMoveAgent(seed + 5)
    spawn_log = {g: [] for g in range(n_games)}
    orig = G.Game.tic

    def tic_log(self, moves, show=False):
        r = orig(self, moves, show)
        sp = self._last_spawn
        spawn_log[self.id].append(-1 if sp is None else sp[0] * W + sp[1])
        return r
    G.Game.tic = tic_log
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            rewards = runner.run(agent)          # the reference's own loop (mp_game_runner.py:23-77)
    finally:
        G.Game.tic = orig
    n_turns = len(agent.log)
    moves = np.full((n_turns, n_games, S), 255, np.uint8)
    for t, (ids, mv) in enumerate(agent.log):
        for (gid, sid), m in zip(ids, mv):
            moves[t, gid, sid] = m
    spawn = np.full((n_turns, n_games), -2, np.int16)
    for g in range(n_games):
        spawn[:len(spawn_log[g]), g] = spawn_log[g]
    flat = dict(H=H, W=W, S=S, hd=hd, n_games=n_games, moves=moves, spawn=spawn,
                rewards=np.array(rewards, np.float32),
                counters=np.array([runner.wall_collision, runner.body_collision, runner.head_collision,
                                   runner.starvation, runner.food_eaten, runner.game_length], np.float64),
                game_lengths=np.array([games_ref[g].game_length for g in range(n_games)], np.int32))
    for k in init_states[0]:
        arr = np.stack([s[k] for s in init_states])
        if k == "nodes":
            arr = arr[:, :, :4]
        flat["init_" + k] = arr
    np.savez_compressed(os.path.join(HERE, "runner.npz"), **flat)
    print(f"[runner] turns={n_turns} rewards={np.array(rewards).tolist()} counters={flat['counters'].tolist()}")


# ----------------------------------------------------------------------------- pit runner (winner indices)
def record_pit(seed=21, H=11, W=11, S=4):
    """pit_mp_game_runner.MPGameRunner.run(Alice, Bob, Alice_snake_cnt) through the reference's own loop
    (pit_mp_game_runner.py:14-63) with two deterministic stub nets: start boards, spawn tape, winners."""
    import utils.pit_mp_game_runner as P
    import utils.pit_agent as PA
    flat = {}
    cases = [("1v3", 1, 3, 24), ("2v2", None, 3, 24), ("3v1", 3, 9, 16), ("2v2_dec1", 2, 1, 6)]
    for ci, (name, a_cnt, hd, n_games) in enumerate(cases):
        pyrandom.seed(seed + ci)
        np.random.seed(seed + ci)
        runner = P.MPGameRunner(H, W, S, hd, n_games)
        L = H * W + 2
        init_states = [snapshot(runner.games[g], L) for g in range(n_games)]
        spawn_log = {g: [] for g in range(n_games)}
        moves_log = {g: [] for g in range(n_games)}
        orig = G.Game.tic

        def tic_log(self, moves, show=False):
            dense = np.full(S, 255, np.uint8)
            for s, m in zip(self.snakes, moves):
                dense[s.id] = m
            r = orig(self, moves, show)
            sp = self._last_spawn
            spawn_log[self.id].append(-1 if sp is None else sp[0] * W + sp[1])
            moves_log[self.id].append(dense)
            return r
        G.Game.tic = tic_log
        try:
            alice, bob = PA.Agent(StubNet(0)), PA.Agent(StubNet(1))
            winners = runner.run(alice, bob, a_cnt)
        finally:
            G.Game.tic = orig
        n_turns = max(len(v) for v in spawn_log.values())
        spawn = np.full((n_turns, n_games), -2, np.int16)
        moves = np.full((n_turns, n_games, S), 255, np.uint8)
        for g in range(n_games):
            spawn[:len(spawn_log[g]), g] = spawn_log[g]
            moves[:len(moves_log[g]), g] = moves_log[g]
        p = f"p{ci}_"
        flat[p + "meta"] = np.array([H, W, S, hd, n_games, -1 if a_cnt is None else a_cnt], np.int32)
        flat[p + "winners"] = np.array([-1 if w is None else w for w in winners], np.int8)
        flat[p + "spawn"] = spawn
        flat[p + "moves"] = moves
        flat[p + "lengths"] = np.array([len(spawn_log[g]) for g in range(n_games)], np.int32)
        for k in init_states[0]:
            arr = np.stack([s[k] for s in init_states])
            if k == "nodes":
                arr = arr[:, :, :4]
            flat[p + "init_" + k] = arr
        print(f"[pit {name}] turns={n_turns} winners={flat[p + 'winners'].tolist()}")
    flat["names"] = np.array([c[0] for c in cases])
    np.savez_compressed(os.path.join(HERE, "pit.npz"), **flat)


# ----------------------------------------------------------------------------- replay.rep text (Game.draw)
def record_replay(seed=31, H=11, W=11, S=4, hd=9):
    """MPGameRunner(game_cnt=1).run: show=True, so every tick appends two boards to ./replay.rep
    (game.py:140-141, 194-195, 281-300).  The file's text is the expected output."""
    import contextlib
    import io
    import tempfile
    flat = {}
    for ci, (sd, p_legal) in enumerate([(seed, 0.8), (seed + 1, 0.95)]):
        pyrandom.seed(sd)
        np.random.seed(sd)
        cwd = os.getcwd()
        with tempfile.TemporaryDirectory() as tmp:
            os.chdir(tmp)
            try:
                runner = R.MPGameRunner(H, W, S, hd, 1)
                L = H * W + 2
                init = snapshot(runner.games[0], L)
                mover = pyrandom.Random(sd + 5)
                log = []

                class Ag:
                    def make_moves(self, games, ids):
                        out = []
                        for (gid, sid) in ids:
                            game = games[gid]
                            snake = [s for s in game.snakes if s.id == sid][0]
                            lm = legal_moves(game, snake)
                            out.append(mover.choice(lm) if lm and mover.random() < p_legal else mover.choice((0, 1, 2)))
                        log.append((list(ids), list(out)))
                        return out
                spawns = []
                orig = G.Game.tic

                def tic_log(self, moves, show=False):
                    r = orig(self, moves, show)
                    sp = self._last_spawn
                    spawns.append(-1 if sp is None else sp[0] * W + sp[1])
                    return r
                G.Game.tic = tic_log
                try:
                    with contextlib.redirect_stdout(io.StringIO()):
                        rewards = runner.run(Ag())
                finally:
                    G.Game.tic = orig
                text = open("replay.rep", "rb").read()
            finally:
                os.chdir(cwd)
        n_turns = len(log)
        moves = np.full((n_turns, 1, S), 255, np.uint8)
        for t, (ids, mv) in enumerate(log):
            for (gid, sid), m in zip(ids, mv):
                moves[t, gid, sid] = m
        p = f"r{ci}_"
        flat[p + "meta"] = np.array([H, W, S, hd], np.int32)
        flat[p + "moves"] = moves
        flat[p + "spawn"] = np.array(spawns, np.int16).reshape(n_turns, 1)
        flat[p + "rewards"] = np.array(rewards, np.float32)
        flat[p + "text"] = np.frombuffer(text, np.uint8)
        for k in init:
            arr = init[k][None]
            if k == "nodes":
                arr = arr[:, :, :4]
            flat[p + "init_" + k] = arr
        print(f"[replay {ci}] turns={n_turns} frames={text.count(bytes([10, 10]))} bytes={len(text)}")
    flat["n"] = 2
    np.savez_compressed(os.path.join(HERE, "replay.npz"), **flat)



# ----------------------------------------------------------------------------- generation loop (trainer) + pit.py script
def _tensorflow_stand_in(load_model=None):
    """TensorFlow is not in this image, so ``utils.alpha_nnet`` (alpha_nnet.py:3-6) -- which
    ``utils.alpha_snake_zero_trainer`` and ``pit.py`` import -- cannot be imported as is.  Empty stand-in MODULES are put
    into sys.modules under the names those four import lines ask for; no reference file is edited and nothing of Keras
    is emulated: the net the recorded runs use is the deterministic StubNet, the module only has to import."""
    import types
    names = ["tensorflow", "tensorflow.keras", "tensorflow.keras.layers", "tensorflow.keras.optimizers",
             "tensorflow.keras.regularizers", "tensorflow.keras.models"]
    mods = {n: types.ModuleType(n) for n in names}
    mods["tensorflow"].keras = mods["tensorflow.keras"]
    for n in names[2:]:
        setattr(mods["tensorflow.keras"], n.rsplit(".", 1)[1], mods[n])
    mods["tensorflow.keras.regularizers"].l2 = lambda c: None
    mods["tensorflow.keras.models"].Model = object
    mods["tensorflow.keras.models"].clone_model = None
    mods["tensorflow.keras.models"].load_model = load_model
    sys.modules.update(mods)


class RawUniformTape(UniformTape):
    """UniformTape whose accepted draws can be rebuilt from the seed: the raw stream is
    RandomState(seed).random_sample, `rejected` lists the raw positions that fell within 1e-4 of a cdf edge"""

    def __init__(self, seed):
        super().__init__(seed)
        self.seed = seed
        self.raw = 0
        self.rejected = []
        outer = self

        class Counting:
            def random_sample(self_inner):
                outer.raw += 1
                return outer._rs.random_sample()
        self._rs = self.rs
        self.rs = Counting()

    def __call__(self, a, p=None):
        before, n_u = self.raw, len(self.u)
        r = super().__call__(a, p)
        self.rejected += list(range(before, self.raw - 1))          # every raw draw of this call but the accepted last one
        assert len(self.u) == n_u + 1
        return r


class _SnakesHashById:
    """Game.tic removes the snakes that died in a tick by iterating a SET of Snake objects (game.py:167), whose order follows
    the objects' memory addresses; with several deaths in one tick that order decides the insertion order of
    `empty_positions` and with it which cell a later `choice(tuple(empty_positions))` means.  Any order is the reference's
    behaviour; hashing the snakes by id while a run is recorded makes the recorded run independent of what the process
    allocated before (the older recordings keep the default hash: they are committed as they were made)."""

    def __enter__(self):
        G.Snake.__hash__ = lambda snake: snake.id

    def __exit__(self, *exc):
        del G.Snake.__hash__


class _StopLoop(Exception):
    pass


def record_trainer():
    """AlphaSnakeZeroTrainer.train (alpha_snake_zero_trainer.py:33-91), the reference's own unmodified loop, run for two
    generations from three starting points (generation 0: log header, health_dec 9; 8: 9 -> 3; 32: 3 -> 1) with a stub
    net object: what it constructs (Agent / MPGameRunner arguments), what it writes (log.csv), what it draws
    (random.sample indices), what it hands to nnet.train (X, V, batch_size), the learning rates, the save names."""
    import contextlib
    import io
    import tempfile
    _tensorflow_stand_in()
    import utils.alpha_snake_zero_trainer as T        # the reference's module, as it is
    flat = {}
    runs = [("gen0", 0, 48, 8, 4, 2e-4, 0.9), ("gen8", 8, 44, 8, 4, 1e-4, 0.98), ("gen32", 32, 12, 8, 6, 3e-4, 0.5)]
    for ri, (tag, start, n_games, breadth, depth, lr0, decay) in enumerate(runs):
        seed = 300 + ri
        pyrandom.seed(seed)
        np.random.seed(seed)
        tape = RawUniformTape(seed + 1)
        A.choice = tape
        L = 11 * 11 + 2
        log = dict(agent=[], runner=[], init=[], spawn=[], tape_pos=[], sample=[], copy_lr=[], train=[], save=[], which=[])

        class Net:
            """the slice of AlphaNNet the loop touches; `train` swaps the stub net so that generation n + 1 demonstrably
            plays with what generation n's training returned"""

            def __init__(self, which):
                self.which = which
                self.stub = StubNet(which)

            def v(self, X):
                return self.stub.v(X)

            def copy_and_compile(self, learning_rate=0.0001, TPU=None):
                log["copy_lr"].append(learning_rate)
                return Net(self.which)

            def train(self, X, V, batch_size=2048):
                Xa, Va = np.array(X, np.float32), np.array(V, np.float32)
                log["train"].append((Xa, Va, batch_size))
                self.which = 1 - self.which
                self.stub = StubNet(self.which)

            def save(self, name):
                log["save"].append(name)
                if len(log["save"]) == 2:
                    raise _StopLoop

        class AgentTap(A.Agent):
            def __init__(self, nnet, softmax_base=100, training=False, max_MCTS_depth=8, max_MCTS_breadth=128):
                log["agent"].append((softmax_base, int(training), max_MCTS_depth, max_MCTS_breadth))
                log["which"].append(nnet.which)
                log["tape_pos"].append(len(tape.u))
                super().__init__(nnet, softmax_base, training, max_MCTS_depth, max_MCTS_breadth)
                log["alice"] = self

        class RunnerTap(R.MPGameRunner):
            def __init__(self, height=11, width=11, snake_cnt=4, health_dec=1, game_cnt=1):
                log["runner"].append((height, width, snake_cnt, health_dec, game_cnt))
                super().__init__(height, width, snake_cnt, health_dec, game_cnt)
                log["init"].append([snapshot(self.games[g], L) for g in range(game_cnt)])
                log["spawn"].append({g: [] for g in range(game_cnt)})

        def sample_tap(pop, k):
            r = pyrandom.sample(pop, k)
            log["sample"].append((len(pop), np.array(r, np.int32)))
            alice = log["alice"]                       # the records and values the indices point into (before clear())
            log.setdefault("records", []).append(np.array([np.frombuffer(
                hashlib.blake2b(np.ascontiguousarray(x).tobytes(), digest_size=8).digest(), np.uint8) for x in alice.records]))
            log.setdefault("values", []).append(np.array(alice.values, np.float32))
            return r

        orig_tic = G.Game.tic

        def tic_log(self, moves, show=False):
            r = orig_tic(self, moves, show)
            if self.food_spawn_chance > 0:              # root games only (sub-games: chance 0, game.py:268)
                sp = self._last_spawn
                log["spawn"][-1][self.id].append(-1 if sp is None else sp[0] * self.width + sp[1])
            return r
        G.Game.tic = tic_log
        T.Agent, T.MPGameRunner, T.sample = AgentTap, RunnerTap, sample_tap
        cwd = os.getcwd()
        with tempfile.TemporaryDirectory() as tmp:
            os.chdir(tmp)
            try:
                with contextlib.redirect_stdout(io.StringIO()), _SnakesHashById():
                    try:
                        T.AlphaSnakeZeroTrainer(n_games, depth, breadth, lr0, decay, 11, 11, 4).train(Net(0), "g", start)
                    except _StopLoop:
                        pass
                text = open("log.csv", "rb").read()
            finally:
                os.chdir(cwd)
                G.Game.tic = orig_tic
                T.Agent, T.MPGameRunner, T.sample = A.Agent, R.MPGameRunner, pyrandom.sample
        p = f"{tag}_"
        flat[p + "ctor"] = np.array([n_games, depth, breadth, start], np.int32)
        flat[p + "lr"] = np.array([lr0, decay], np.float64)
        flat[p + "log_csv"] = np.frombuffer(text, np.uint8)
        flat[p + "agent_args"] = np.array(log["agent"], np.int32)
        flat[p + "runner_args"] = np.array(log["runner"], np.int32)
        flat[p + "net_which"] = np.array(log["which"], np.int8)
        flat[p + "copy_lr"] = np.array(log["copy_lr"], np.float64)
        flat[p + "save_names"] = np.array(log["save"])
        flat[p + "tape_seed"] = np.int64(tape.seed)
        flat[p + "tape_len"] = np.int64(len(tape.u))
        flat[p + "tape_rejected"] = np.array(tape.rejected, np.int64)
        flat[p + "tape_digest"] = np.frombuffer(hashlib.blake2b(np.array(tape.u, np.float64).tobytes(), digest_size=16).digest(), np.uint8)
        flat[p + "tape_pos"] = np.array(log["tape_pos"], np.int64)
        for gi in range(2):
            q = p + f"g{gi}_"
            init = log["init"][gi]
            for k in init[0]:
                arr = np.stack([s[k] for s in init])
                flat[q + "init_" + k] = arr[:, :, :4] if k == "nodes" else arr
            sp = log["spawn"][gi]
            n_turns = max(len(v) for v in sp.values())
            spawn = np.full((n_turns, len(sp)), -2, np.int16)
            for g in sp:
                spawn[:len(sp[g]), g] = sp[g]
            flat[q + "spawn"] = spawn
            n_pop, idx = log["sample"][gi]
            Xa, Va, bs = log["train"][gi]
            assert n_pop == len(log["records"][gi]) == len(log["values"][gi])
            flat[q + "n_records"] = np.int64(n_pop)
            flat[q + "sample_idx"] = idx
            flat[q + "records_digest"] = log["records"][gi]
            flat[q + "values"] = log["values"][gi]
            flat[q + "batch_size"] = np.int64(bs)
            flat[q + "X_rows"] = np.int64(len(Xa))
            flat[q + "X_digest"] = np.frombuffer(hashlib.blake2b(Xa.tobytes(), digest_size=32).digest(), np.uint8)
            flat[q + "V_digest"] = np.frombuffer(hashlib.blake2b(Va.tobytes(), digest_size=32).digest(), np.uint8)
            # the values the loop hands to the fit are the recorded ones at the sampled indices, then their mirror images
            assert np.array_equal(Va[:len(idx)], log["values"][gi][idx]) and np.array_equal(Va[len(idx):], Va[:len(idx), ::-1])
            print(f"[trainer {tag} gen {gi}] runner={log['runner'][gi]} agent={log['agent'][gi]} turns={n_turns} "
                  f"records={n_pop} sampled={len(idx)} X={Xa.shape} batch={bs}")
        print(f"[trainer {tag}] copy_lr={log['copy_lr']} saves={log['save']} draws={len(tape.u)} rejected={len(tape.rejected)}")
        print(text.decode())
    flat["names"] = np.array([r[0] for r in runs])
    np.savez_compressed(os.path.join(HERE, "trainer.npz"), **flat)


def record_pit_script():
    """pit.py (pit.py:1-62), the reference's champion-ladder SCRIPT, executed unmodified with runpy: `input()` answers
    "m" / "3", generations 3..5 "exist" (the stand-in load_model hands out stub nets, anything else raises OSError as a
    missing file does), the first `sleep` ends the otherwise endless loop.  Recorded: pit.txt, and per challenger the
    1 000 start boards, the food-spawn tape and the winner indices."""
    import builtins
    import contextlib
    import io
    import runpy
    import tempfile
    import time as time_mod
    from oracle.obs_key import stub_q_from_key

    class StubKeras:
        def __init__(self, which):
            self.which = which

        def predict(self, X):                       # un-masked: AlphaNNet.v applies alpha_nnet.py:63-76 itself
            k = obs_key(X)
            with np.errstate(over="ignore"):
                return stub_q_from_key(k[:, self.which] if self.which < 2 else k[:, 0] + k[:, 1])

    def load_model(path):
        gen = int(path[len("models/m"):-len(".h5")])
        if path != f"models/m{gen}.h5" or not 3 <= gen <= 5:
            raise OSError("no such file: " + path)
        return StubKeras(gen - 3)
    _tensorflow_stand_in(load_model)
    for m in ("utils.alpha_nnet",):
        sys.modules.pop(m, None)                    # (re-)import it against the stand-in that has load_model
    import utils.pit_mp_game_runner as P
    H = W = 11
    S = 2
    L = H * W + 2
    log = dict(init=[], spawn=[], winners=[], args=[])

    class RunnerTap(P.MPGameRunner):
        def __init__(self, height=11, width=11, snake_cnt=4, health_dec=1, game_cnt=1):
            log["args"].append((height, width, snake_cnt, health_dec, game_cnt))
            super().__init__(height, width, snake_cnt, health_dec, game_cnt)
            log["init"].append([snapshot(self.games[g], L) for g in range(game_cnt)])
            log["spawn"].append({g: [] for g in range(game_cnt)})

        def run(self, Alice, Bob, Alice_snake_cnt=None):
            w = super().run(Alice, Bob, Alice_snake_cnt)
            log["winners"].append((Alice_snake_cnt, list(w)))
            return w

    orig_tic = G.Game.tic

    def tic_log(self, moves, show=False):
        r = orig_tic(self, moves, show)
        sp = self._last_spawn
        log["spawn"][-1][self.id].append(-1 if sp is None else sp[0] * W + sp[1])
        return r

    def stop(_seconds):
        raise _StopLoop
    answers = iter(["m", "3"])
    pyrandom.seed(81)               # of seeds 77..83 the one whose run has both outcomes (a title change, then a failed challenge)
    np.random.seed(81)
    orig_runner, orig_sleep, orig_input = P.MPGameRunner, time_mod.sleep, builtins.input
    P.MPGameRunner, time_mod.sleep, builtins.input = RunnerTap, stop, lambda prompt="": next(answers)
    G.Game.tic = tic_log
    cwd = os.getcwd()
    out = io.StringIO()
    with tempfile.TemporaryDirectory() as tmp:
        os.chdir(tmp)
        try:
            with contextlib.redirect_stdout(out), _SnakesHashById():
                try:
                    runpy.run_path("/root/reference/code/pit.py", run_name="__main__")
                except _StopLoop:
                    pass
            text = open("pit.txt", "rb").read()
        finally:
            os.chdir(cwd)
            P.MPGameRunner, time_mod.sleep, builtins.input = orig_runner, orig_sleep, orig_input
            G.Game.tic = orig_tic
    flat = dict(pit_txt=np.frombuffer(text, np.uint8), n=len(log["winners"]),
                stdout=np.array([ln for ln in out.getvalue().splitlines() if not ln.startswith("Competing time")]))
    for ci in range(len(log["winners"])):
        p = f"c{ci}_"
        a_cnt, winners = log["winners"][ci]
        flat[p + "args"] = np.array(log["args"][ci] + (a_cnt,), np.int32)
        flat[p + "winners"] = np.array([-1 if w is None else w for w in winners], np.int8)
        sp = log["spawn"][ci]
        n_turns = max(len(v) for v in sp.values())
        spawn = np.full((n_turns, len(sp)), -2, np.int16)
        for g in sp:
            spawn[:len(sp[g]), g] = sp[g]
        flat[p + "spawn"] = spawn
        init = log["init"][ci]
        for k in init[0]:
            arr = np.stack([s[k] for s in init])
            flat[p + "init_" + k] = arr[:, :, :4] if k == "nodes" else arr
        print(f"[pit_script challenger {ci}] args={flat[p + 'args'].tolist()} turns={n_turns} "
              f"winners: A {sum(w == 0 for w in winners)} B {sum(w == 1 for w in winners)} draw {sum(w is None for w in winners)}")
    print(text.decode())
    np.savez_compressed(os.path.join(HERE, "pit_script.npz"), **flat)


def record_net_graph():
    """AlphaNNet.__init__ / copy_and_compile / train / save (alpha_nnet.py:10-56, 58-59, 78-109), the reference's own code, run
    against RECORDING stand-ins for the Keras names it imports (TensorFlow is absent): which layers it builds with which
    arguments and how it wires them, which optimizer / schedule / loss it compiles with, what it passes to fit and save.
    This pins the STRUCTURE of the net and of the training call to the reference; the arithmetic behind the names stays
    Keras' (unpinned)."""
    import json
    import types
    log = []
    counter = [0]

    class T:                                    # a symbolic tensor: just an id
        def __init__(self):
            counter[0] += 1
            self.id = counter[0]

    def enc(v):
        if isinstance(v, T):
            return {"tensor": v.id}
        if isinstance(v, (list, tuple)):
            return [enc(x) for x in v]
        if isinstance(v, Rec):
            return {"obj": v.kind, "args": enc(v.args), "kwargs": {k: enc(x) for k, x in v.kwargs.items()}}
        return v

    class Rec:
        kind = "?"

        def __init__(self, *args, **kwargs):
            self.args, self.kwargs = args, kwargs

        def __call__(self, x):
            out = T()
            log.append({"layer": self.kind, "args": enc(self.args), "kwargs": {k: enc(v) for k, v in self.kwargs.items()},
                        "in": enc(x), "out": out.id})
            return out

    def layer(name):
        return type(name, (Rec,), {"kind": name})

    def Input(shape):
        t = T()
        log.append({"layer": "Input", "args": [list(shape)], "kwargs": {}, "in": None, "out": t.id})
        return t

    class Model:
        def __init__(self, inputs=None, outputs=None):
            self.inputs, self.outputs = inputs, outputs
            self.layers = [types.SimpleNamespace(input_shape=(None, 21, 21, 3))]
            self.calls = []
            log.append({"layer": "Model", "args": [], "kwargs": {"inputs": enc(inputs), "outputs": enc(outputs)}, "in": None, "out": None})

        def get_weights(self):
            return ["w"]

        def set_weights(self, w):
            self.calls.append(["set_weights", w])

        def build(self, shape):
            self.calls.append(["build", list(shape)])

        def compile(self, **kw):
            self.calls.append(["compile", {k: enc(v) for k, v in kw.items()}])

        def fit(self, X, Y, **kw):
            self.calls.append(["fit", [list(np.shape(X)), str(np.asarray(X).dtype), list(np.shape(Y))], kw])

        def save(self, path):
            self.calls.append(["save", path])

    def clone_model(m):
        c = Model.__new__(Model)
        c.inputs, c.outputs, c.layers, c.calls = m.inputs, m.outputs, m.layers, [["cloned"]]
        return c
    names = ["tensorflow", "tensorflow.keras", "tensorflow.keras.layers", "tensorflow.keras.optimizers",
             "tensorflow.keras.regularizers", "tensorflow.keras.models"]
    mods = {n: types.ModuleType(n) for n in names}
    mods["tensorflow"].keras = mods["tensorflow.keras"]
    for n in names[2:]:
        setattr(mods["tensorflow.keras"], n.rsplit(".", 1)[1], mods[n])
    L = mods["tensorflow.keras.layers"]
    L.Input = Input
    for nm in ("Conv2D", "BatchNormalization", "Activation", "Add", "Flatten", "Dense"):
        setattr(L, nm, layer(nm))
    O = mods["tensorflow.keras.optimizers"]
    O.Adam = layer("Adam")
    O.schedules = types.SimpleNamespace(PiecewiseConstantDecay=layer("PiecewiseConstantDecay"))
    mods["tensorflow.keras.regularizers"].l2 = layer("l2")
    M = mods["tensorflow.keras.models"]
    M.Model, M.clone_model, M.load_model = Model, clone_model, lambda path: (_ for _ in ()).throw(OSError(path))
    sys.modules.update(mods)
    sys.modules.pop("utils.alpha_nnet", None)
    import utils.alpha_nnet as NN                 # the reference's module, as it is
    net = NN.AlphaNNet(input_shape=(21, 21, 3))
    graph = list(log)
    compiled = []
    for lr in (0.0001, 9.8e-05, 0.0003, 1e-4 * 0.98 ** 40):
        c = net.copy_and_compile(lr)
        compiled.append({"learning_rate": lr, "calls": c.v_net.calls})
    default = net.copy_and_compile()
    X = [np.zeros((21, 21, 3), np.float32)] * 5
    Y = [np.zeros(3, np.float32)] * 5
    default.train(X, Y, batch_size=4)
    default.train(X, Y)
    default.save("g7")
    out = {"graph": graph, "compiled": compiled, "default_compile_and_calls": default.v_net.calls,
           "is_obstacle": [[v, bool(net.is_obstacle(np.float32(v)))] for v in (0.0, 0.02, 0.04, 0.06, 1.0)]}
    text = json.dumps(out, sort_keys=True)
    np.savez_compressed(os.path.join(HERE, "net_graph.npz"), json=np.frombuffer(text.encode(), np.uint8))
    convs = [g for g in graph if g["layer"] == "Conv2D"]
    print(f"[net_graph] {len(graph)} graph entries: {len(convs)} Conv2D, {sum(g['layer'] == 'BatchNormalization' for g in graph)} BN, "
          f"{sum(g['layer'] == 'Add' for g in graph)} Add, {sum(g['layer'] == 'Dense' for g in graph)} Dense; {len(text)} bytes of json")
    sys.modules.pop("utils.alpha_nnet", None)

if __name__ == "__main__":
    os.chdir("/tmp")
    which = set(sys.argv[1:]) or {"tic", "tic_more", "corner", "tables", "mcts", "mcts_more", "runner", "pit", "replay", "trainer", "pit_script", "net_graph"}
    if "tic" in which:
        record_trajectories("11x11x4", 11, 11, 4, 1, 40, seed=1, p_legal=0.92, max_ticks=400, raw_every=23)
        record_trajectories("11x11x4_dec9", 11, 11, 4, 9, 10, seed=2, p_legal=0.97, max_ticks=400, raw_every=29)
        record_trajectories("7x7x2", 7, 7, 2, 3, 16, seed=3, p_legal=0.9, max_ticks=300, raw_every=17)
        record_trajectories("19x19x8", 19, 19, 8, 1, 6, seed=4, p_legal=0.95, max_ticks=300, raw_every=61)
    if "tic_more" in which:      # board sizes outside the BASELINE configs (the engine's run-time-geometry kernels)
        record_trajectories("9x9x3", 9, 9, 3, 1, 8, seed=5, p_legal=0.9, max_ticks=200, raw_every=19)
        record_trajectories("15x15x5", 15, 15, 5, 3, 4, seed=6, p_legal=0.93, max_ticks=200, raw_every=37)
        record_trajectories("16x16x6", 16, 16, 6, 1, 3, seed=7, p_legal=0.93, max_ticks=160, raw_every=41)
        record_trajectories("5x5x2", 5, 5, 2, 1, 8, seed=8, p_legal=0.9, max_ticks=100, raw_every=13)
    if "corner" in which:
        record_corner_cases()
    if "tables" in which:
        record_tables()
    if "mcts" in which:
        record_mcts_tiny("tiny", n_games=2, breadth=16, depth=8, base=2, seed=11, max_turns=12)
        record_mcts_tiny("tiny_greedybase", n_games=3, breadth=8, depth=4, base=10, seed=12, max_turns=8)
    if "mcts_more" in which:     # the same recording at other geometries (2 snakes: the deepest rollouts; 8 snakes: depth cap <= 0)
        record_mcts_tiny("7x7x2", n_games=3, breadth=16, depth=8, base=2, seed=13, max_turns=10, H=7, W=7, S=2, hd=3)
        record_mcts_tiny("19x19x8", n_games=1, breadth=8, depth=8, base=3, seed=14, max_turns=5, H=19, W=19, S=8, hd=1)
        record_mcts_tiny("9x9x3", n_games=2, breadth=24, depth=6, base=5, seed=15, max_turns=8, H=9, W=9, S=3, hd=9)
    if "runner" in which:
        record_runner()
    if "pit" in which:
        record_pit()
    if "replay" in which:
        record_replay()
    if "trainer" in which:       # needs the tensorflow stand-in modules: after everything that must not see them
        record_trainer()
    if "pit_script" in which:
        record_pit_script()
    if "net_graph" in which:
        record_net_graph()
