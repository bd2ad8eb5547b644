"""Drop-in mirror of the reference's ``utils.mp_game_runner`` (mp_game_runner.py).

``MPGameRunner`` keeps the constructor, attributes and ``run(Alice) -> rewards`` contract
(mp_game_runner.py:7-77) but its games are slots of one ``snake_engine.Engine`` in HBM; a root turn
is: ask the agent for all moves, one step-kernel launch over the live slots, read back the done
flags, retire finished games.  ``MCTSMPGameRunner`` is the rollout flavour (mp_game_runner.py:79-115).
"""
from time import time

import numpy as np
import torch

from snake_engine import Engine
from snake_engine._lib import check
from utils.game import Game, draw_init_tape


LOG_FIELDS = ("wall_collision", "body_collision", "head_collision", "starvation", "food_eaten", "game_length")


class GameDict(dict):
    """``{game_id: Game}`` whose games are slots of one engine (what Agent.make_moves works on)."""

    def __init__(self, engine, games):
        super().__init__(games)
        self.engine = engine

    def live_slots(self):
        return np.fromiter((g._slot for g in self.values()), np.int32, len(self))


class MPGameRunner:
    verbose = True        # the reference prints every root turn (mp_game_runner.py:34-37, 68)
    init = "host"         # "host": start boards drawn with python's `random` exactly as game.py:25-30,46
    #                       "device": drawn on the GPU (counter-based Philox), for very large batches

    def __init__(self, height=11, width=11, snake_cnt=4, health_dec=1, game_cnt=1, seed=None):
        self.height = height
        self.width = width
        self.snake_cnt = snake_cnt
        self.health_dec = health_dec
        self.game_cnt = game_cnt
        if seed is None:
            seed = int(np.random.randint(1 << 62))
        self.engine = Engine(game_cnt, height, width, snake_cnt, health_dec, 0.15, seed=seed)
        if self.init == "host":
            self.engine.reset(init_tape=np.array([draw_init_tape(snake_cnt) for _ in range(game_cnt)], np.uint8))
        else:
            self.engine.reset()
        self.games = GameDict(self.engine, {ID: Game(ID, height, width, snake_cnt, health_dec, 0.15,
                                                      _engine=self.engine, _slot=ID) for ID in range(game_cnt)})
        for name in LOG_FIELDS:              # the six log counters the trainer reads (mp_game_runner.py:14-20)
            setattr(self, name, 0)
        # raw sums and results live on the instance so that run(max_turns=...) can be resumed: the public counters are
        # always totals / game_cnt (mp_game_runner.py:71-76), never an average of an average
        self._totals = dict.fromkeys(LOG_FIELDS, 0)
        self._rewards = [None] * game_cnt
        self.turns = 0

    def run(self, Alice, spawn_tape=None, max_turns=None):
        """mp_game_runner.py:23-77.  ``spawn_tape`` (optional, parity runs): callable turn -> int16[game_cnt]
        giving the recorded food-spawn cell (-1 none) of every game for that turn."""
        t0 = time()
        games = self.games
        eng = self.engine
        S = self.snake_cnt
        show = self.game_cnt == 1
        rewards = self._rewards
        turn = self.turns                      # a resumed run continues the turn count (and the spawn tape)
        first_turn = turn
        self.env_steps = 0
        while games:
            if max_turns is not None and turn - first_turn >= max_turns:
                break
            turn += 1
            self.env_steps += len(games)       # one root Game.tic per live game (mp_game_runner.py:52)
            if self.verbose:                 # same lines as the reference prints (mp_game_runner.py:34-37)
                what = "Running the root game." if len(games) == 1 else f"Concurrently running {len(games)} root games."
                print(f"{what} On turn {turn}...")
            gids = list(games.keys())
            slots = games.live_slots()
            d_slots = torch.as_tensor(slots, device=eng.device)
            alive = eng.alive(slots=d_slots)
            alive_h = alive.cpu().numpy().astype(bool)
            # ids in the reference's order: games in dict order, alive snakes ascending (mp_game_runner.py:40-42)
            gi, si = np.nonzero(alive_h)
            ids = [(gids[g], int(s)) for g, s in zip(gi, si)]
            moves = Alice.make_moves(games, ids)
            dense = np.ones((len(gids), S), np.uint8)
            dense[gi, si] = np.asarray(moves, np.uint8)
            done = eng.new((len(gids),), torch.uint8, 0)
            tape = None
            if spawn_tape is not None:
                tape = torch.as_tensor(np.ascontiguousarray(spawn_tape(turn)[slots], np.int16), device=eng.device)
            pre = games[gids[0]]._pull() if show else None
            eng.step(torch.as_tensor(dense, device=eng.device), slots=d_slots, spawn_tape=tape, done=done)
            for g in games.values():
                g._dirty()
            if show:                         # both boards of game.py:140-141, 194-195
                games[gids[0]].draw_tick(pre, dense[0])
            done_h = done.cpu().numpy().astype(bool)
            if done_h.any():
                fin = np.flatnonzero(done_h)
                for name, v in zip(LOG_FIELDS, eng.sum_counters(slots=slots[fin])):      # mp_game_runner.py:54-60
                    self._totals[name] += v
                rw = torch.empty((len(fin), S), dtype=torch.int8, device=eng.device)
                check(eng.L.snk_engine_rewards(eng.h, torch.as_tensor(slots[fin], device=eng.device).data_ptr(), len(fin),
                                               rw.data_ptr(), torch.cuda.current_stream().cuda_stream))
                rw = rw.cpu().numpy()
                for k, j in enumerate(fin):
                    rewards[gids[j]] = [None if r == 0 else float(r) for r in rw[k]]
                    del games[gids[j]]
            if self.verbose:
                print(f"Root game turn {turn} finished. Total time spent: {time() - t0}", end="\n\n")
        self.turns = turn
        for name in LOG_FIELDS:              # per-game averages (mp_game_runner.py:71-76) over the games finished so far
            setattr(self, name, self._totals[name] / self.game_cnt)
        return rewards


class MCTSMPGameRunner(MPGameRunner):
    """mp_game_runner.py:79-115: lock-step rollout of sub-games with a per-game depth cap."""

    def __init__(self, games):
        self.games = games

    # MCTSAlice is the agent
    def run(self, MCTSAlice, MCTS_depth):
        games = self.games
        eng = games.engine
        S = eng.S
        rewards = {game_id: None for game_id in games}
        turn = 0
        while games:
            turn += 1
            gids = list(games.keys())
            slots = games.live_slots()
            d_slots = torch.as_tensor(slots, device=eng.device)
            alive_h = eng.alive(slots=d_slots).cpu().numpy().astype(bool)
            gi, si = np.nonzero(alive_h)
            ids = [(gids[g], int(s)) for g, s in zip(gi, si)]
            moves = MCTSAlice.make_moves(games, ids)
            dense = np.ones((len(gids), S), np.uint8)
            dense[gi, si] = np.asarray(moves, np.uint8)
            done = eng.new((len(gids),), torch.uint8, 0)
            eng.step(torch.as_tensor(dense, device=eng.device), slots=d_slots, done=done)
            done_h = done.cpu().numpy().astype(bool)
            for j, gid in enumerate(gids):
                games[gid]._dirty()
                if done_h[j] or turn >= MCTS_depth[gid]:
                    rewards[gid] = games[gid].rewards
                    del games[gid]
        return rewards
