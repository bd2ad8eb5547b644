"""Drop-in mirror of the reference's ``utils.pit_mp_game_runner`` (pit_mp_game_runner.py:3-63): two agents,
snakes split by id into team A (< Alice_snake_cnt) and team B, early exit when one team is eliminated
(SURVEY.md section 8 row f-2).  Games are slots of one engine; observations of all live snakes come from one
observe-kernel launch per turn and are handed to the agents as device tensors (or as the reference's lists of
arrays when an agent has no device path)."""
import numpy as np
import torch

from snake_engine import Engine
from snake_engine._lib import check
from utils.game import Game, draw_init_tape
from utils.mp_game_runner import GameDict


class MPGameRunner:

    def __init__(self, height=11, width=11, snake_cnt=4, health_dec=1, game_cnt=1, seed=None):
        self.height, self.width, self.snake_cnt = height, width, snake_cnt
        self.health_dec, self.game_cnt = health_dec, game_cnt
        if seed is None:
            seed = int(np.random.randint(1 << 62))
        self.engine = Engine(game_cnt, height, width, snake_cnt, health_dec, 0.15, seed=seed)
        self.engine.reset(init_tape=np.array([draw_init_tape(snake_cnt) for _ in range(game_cnt)], np.uint8))
        self.games = GameDict(self.engine, {ID: Game(ID, height, width, snake_cnt, health_dec, 0.15,
                                                      _engine=self.engine, _slot=ID) for ID in range(game_cnt)})

    # Alice and Bob are agents using different nets
    def run(self, Alice, Bob, Alice_snake_cnt=None, spawn_tape=None):
        games, eng, S = self.games, self.engine, self.snake_cnt
        show = self.game_cnt == 1
        if Alice_snake_cnt is None:
            Alice_snake_cnt = S // 2
        winners = [None] * self.game_cnt
        turn = 0
        while games:
            turn += 1
            gids = list(games.keys())
            slots = games.live_slots()
            d_slots = torch.as_tensor(slots, device=eng.device)
            alive_h = eng.alive(slots=d_slots).cpu().numpy().astype(bool)
            gi, si = np.nonzero(alive_h)
            order = np.concatenate([np.flatnonzero(si < Alice_snake_cnt), np.flatnonzero(si >= Alice_snake_cnt)])
            nA = int((si < Alice_snake_cnt).sum())
            pairs = np.stack([slots[gi], si], axis=1).astype(np.int32)[order]      # team A rows first, then team B
            planes, mask, _ = eng.observe_all(pairs, want_key=False)
            ids = [(gids[g], int(s)) for g, s in zip(gi[order], si[order])]

            def ask(agent, lo, hi):
                if hi == lo:
                    return []
                try:
                    return list(agent.make_moves(planes[lo:hi], ids[lo:hi], mask=mask[lo:hi]))
                except TypeError:           # an agent with the reference's (states, ids) signature
                    return list(agent.make_moves(list(planes[lo:hi].cpu().numpy()), ids[lo:hi]))
            moves = ask(Alice, 0, nA) + ask(Bob, nA, len(ids))
            dense = np.ones((len(gids), S), np.uint8)
            dense[gi[order], si[order]] = np.asarray(moves, np.uint8)
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
            rw = torch.empty((len(gids), S), dtype=torch.int8, device=eng.device)
            check(eng.L.snk_engine_rewards(eng.h, d_slots.data_ptr(), len(gids), rw.data_ptr(),
                                           torch.cuda.current_stream().cuda_stream))
            rw = rw.cpu().numpy()
            alive2 = eng.alive(slots=d_slots).cpu().numpy().astype(bool)
            for j, gid in enumerate(gids):
                if done_h[j]:                                         # pit_mp_game_runner.py:43-47
                    w = np.flatnonzero(rw[j] == 1)
                    if len(w):
                        winners[gid] = int(w[-1])
                    del games[gid]
                else:                                                 # :48-60 the team with snakes left wins
                    ids_alive = np.flatnonzero(alive2[j])
                    A = (ids_alive < Alice_snake_cnt).any()
                    B = (ids_alive >= Alice_snake_cnt).any()
                    if not A or not B:
                        winners[gid] = int(ids_alive[0])
                        del games[gid]
        return winners
