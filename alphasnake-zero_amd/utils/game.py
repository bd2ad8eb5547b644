"""Drop-in mirror of the reference's ``utils.game`` (game.py) on top of the HBM-resident engine.

``Game`` keeps the reference's constructor, methods and attribute names (game.py:11-300) but holds no
board itself: it is a view of one slot of a ``snake_engine.Engine``.  ``tic`` / ``get_states`` /
``subgame`` launch the HIP kernels through the C ABI; ``snakes``, ``food``, ``rewards``, the six
counters and the derived sets are read back from the device on access.  There is no CPU rules
engine here: without the HIP library or a GPU, constructing a Game raises.
"""
from random import sample, choice

import numpy as np
import torch

from snake_engine import Engine
from snake_engine.engine import compact_from_state

WALL = 1.0
MY_HEAD = -1.0
# multipliers (game.py:4-9)
HEALTH_m = 0.01
SNAKE_m = 0.02
HEAD_m = 0.04


def draw_init_tape(snake_cnt):
    """The draws Game.__init__ makes (game.py:25-30, 46), in the reference's order and with the same
    ``random`` calls, so a seeded run starts from the same boards as the reference."""
    positions = sample(range(8), snake_cnt)                       # sample(8 standard cells, snake_cnt)
    dirs = [choice((0, 1, 2, 3)) for _ in range(snake_cnt)]       # last_moves
    food = [choice((0, 1, 2, 3)) for _ in range(snake_cnt)]       # which diagonal neighbour gets food
    return [positions, dirs, food]


class Node:
    def __init__(self, yx):
        self.position = yx
        self.prev_node = None
        self.next_node = None


class Snake:
    """Host snapshot of one snake (game.py:302-379): id, health, length, head/tail nodes, body iteration."""

    def __init__(self, ID, health, head_and_body):
        self.id = ID
        self.health = health
        self.length = len(head_and_body)
        self.head = Node(head_and_body[0])
        self.tail = self.head
        for yx in head_and_body[1:]:
            n = Node(yx)
            n.prev_node = self.tail
            self.tail.next_node = n
            self.tail = n

    def __iter__(self):          # body positions, head excluded (game.py:317-327)
        n = self.head.next_node
        while n:
            yield n.position
            n = n.next_node

    def positions(self):
        out, n = [], self.head
        while n:
            out.append(n.position)
            n = n.next_node
        return out


class Game:
    def __init__(self, ID, height=11, width=11, snake_cnt=4, health_dec=1, food_spawn_chance=0.15,
                 _engine=None, _slot=0):
        self.id = ID
        self.height = height
        self.width = width
        self.snake_cnt = snake_cnt
        self.health_dec = health_dec
        self.food_spawn_chance = food_spawn_chance
        if _engine is None:
            _engine = Engine(1, height, width, snake_cnt, health_dec, food_spawn_chance,
                             seed=np.random.randint(1 << 62))
            _engine.reset(init_tape=np.array([draw_init_tape(snake_cnt)], np.uint8))
            _slot = 0
        self._engine = _engine
        self._slot = int(_slot)
        self._cache = None

    # ---- device -> host snapshot ---------------------------------------------------------------
    def _pull(self):
        if self._cache is None:
            self._cache = compact_from_state(self._engine.export([self._slot])[0])
        return self._cache

    def _dirty(self):
        self._cache = None

    def _yx(self, c):
        return (int(c) // self.width, int(c) % self.width)

    @property
    def snakes(self):
        st = self._pull()
        out = []
        for s in range(self.snake_cnt):
            if st["alive"][s]:
                L = int(st["length"][s])
                out.append(Snake(s, int(st["health"][s]), [self._yx(c) for c in st["nodes"][s, :L]]))
        return out

    @property
    def last_moves(self):
        return {s: int(d) for s, d in enumerate(self._pull()["dir"])}

    @property
    def rewards(self):
        return [None if r == 0 else float(r) for r in self._pull()["rewards"]]

    @property
    def food(self):
        return {self._yx(c) for c in np.flatnonzero(self._pull()["food"])}

    @property
    def heads(self):
        out = {}
        for s in self.snakes:
            out.setdefault(s.head.position, set()).add(s)
        return out

    @property
    def bodies(self):
        return {b for s in self.snakes for b in s}

    @property
    def empty_positions(self):
        occ = set(self.heads) | self.bodies | self.food
        return {(y, x) for y in range(self.height) for x in range(self.width)} - occ

    wall_collision = property(lambda self: int(self._pull()["counters"][0]))
    body_collision = property(lambda self: int(self._pull()["counters"][1]))
    head_collision = property(lambda self: int(self._pull()["counters"][2]))
    starvation = property(lambda self: int(self._pull()["counters"][3]))
    food_eaten = property(lambda self: int(self._pull()["counters"][4]))
    game_length = property(lambda self: int(self._pull()["counters"][5]))

    # ---- reference API ---------------------------------------------------------------------------
    def get_ids(self):
        """game.py:76-77"""
        return [(self.id, s) for s in np.flatnonzero(self._pull()["alive"]).tolist()]

    def get_states(self):
        """game.py:68-69: one (2H-1, 2W-1, 3) float32 array per alive snake, list order = id order"""
        ids = np.flatnonzero(self._pull()["alive"]).astype(np.int32)
        if len(ids) == 0:
            return []
        pairs = np.stack([np.full(len(ids), self._slot, np.int32), ids], axis=1)
        planes, _, _ = self._engine.observe_all(pairs, want_mask=False, want_key=False)
        return list(planes.cpu().numpy())

    def make_state(self, you, last_move):
        """game.py:215-257.  ``you``: a Snake of this game (or its id)."""
        sid = you.id if hasattr(you, "id") else int(you)
        pairs = np.array([[self._slot, sid]], np.int32)
        planes, _, _ = self._engine.observe_all(pairs, want_mask=False, want_key=False)
        st = planes[0].cpu().numpy()
        extra = (int(last_move) - int(self._pull()["dir"][sid])) % 4    # the device rotates by the stored heading
        return np.rot90(st, k=extra) if extra else st

    def tic(self, moves, show=False):
        """game.py:87-205: ``moves`` pairs with the alive snakes in list order. Returns 0 or the rewards list."""
        pre = self._pull()
        alive = np.flatnonzero(pre["alive"])
        dense = np.ones((1, self.snake_cnt), np.uint8)
        dense[0, alive] = np.asarray(moves, np.uint8)[: len(alive)]
        eng = self._engine
        eng.set_params(self.health_dec, self.food_spawn_chance)
        done = eng.new((1,), torch.uint8, 0)
        eng.step(torch.as_tensor(dense, device=eng.device), slots=np.array([self._slot], np.int32), done=done)
        self._dirty()
        if show:
            self.draw_tick(pre, dense[0])
        return self.rewards if int(done.item()) else 0

    def draw_tick(self, pre, dense_moves):
        """the two boards Game.tic(show=True) appends (game.py:140-141 before the dead snakes are removed, :194-195
        after): ``pre`` is the host snapshot taken before the step, ``dense_moves`` the move of every snake id."""
        self._write_board(self.food, self._moved_snakes(pre, dense_moves))
        self.draw()

    def _moved_snakes(self, pre, dense_moves):
        """every snake of the pre-state after the move and eat phases (game.py:90-127), dying ones included: what
        ``self.snakes`` holds at the first draw.  Rebuilt on the host from the snapshot (one game, show mode only);
        the food after spawning is the post-state food (removal never touches it)."""
        food_pre = set(np.flatnonzero(pre["food"]).tolist())
        out = []
        for s in np.flatnonzero(pre["alive"]).tolist():
            L = int(pre["length"][s])
            nodes = [self._yx(c) for c in pre["nodes"][s, :L]]
            d = (int(dense_moves[s]) + int(pre["dir"][s]) - 1) % 4           # game.py:92
            hy, hx = nodes[0]
            head = (hy + (d == 2) - (d == 0), hx + (d == 1) - (d == 3))
            body = nodes[:-1]                                                # old head joins the body, tail popped
            length = L
            if 0 <= head[0] < self.height and 0 <= head[1] < self.width:
                c = head[0] * self.width + head[1]
                if c in food_pre:                                            # first in list order eats (game.py:121-127)
                    food_pre.discard(c)
                    length += 1                                              # grow(): duplicate tail, same cells
            out.append((s, length, head, body))
        return out

    def _write_board(self, food, snakes):
        """game.py:281-300 on (id, length, head, body cells) tuples in list order"""
        board = [[0] * self.width for _ in range(self.height)]
        for (y, x) in food:
            board[y][x] = 9
        for (sid, _, (hy, hx), _) in sorted(snakes, key=lambda t: t[1]):
            if 0 <= hy < self.height and 0 <= hx < self.width:
                board[hy][hx] = -(sid + 1)
        for (sid, _, _, body) in snakes:
            for (y, x) in body:
                board[y][x] = sid + 1
        with open("replay.rep", "a") as f:
            for row in board:
                f.write(str(row) + "\n")
            f.write("\n")

    def subgame(self, subgame_id):
        """game.py:266-276: deep copy that never spawns food, fresh counters, copied rewards"""
        eng = Engine(1, self.height, self.width, self.snake_cnt, self.health_dec, 0.0, device=self._engine.device.index)
        self._engine.clone_to(eng, src_slots=np.array([self._slot], np.int32), fanout=1)
        return Game(subgame_id, self.height, self.width, self.snake_cnt, self.health_dec, 0.0, _engine=eng, _slot=0)

    def draw(self):
        """game.py:281-300: append the board to replay.rep in the text format player.py reads"""
        self._write_board(self.food, [(s.id, s.length, s.head.position, list(s)) for s in self.snakes])
