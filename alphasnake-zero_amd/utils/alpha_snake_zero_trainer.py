"""Drop-in mirror of the reference's ``utils.alpha_snake_zero_trainer`` (alpha_snake_zero_trainer.py:10-100) -- the caller of
the hot path, SURVEY.md section 8 row f-1: the endless generation loop that ``train.py`` starts.

One generation = self-play of ``self_play_games`` games with an ``Agent`` whose softmax base is ``2 + iteration``
(trainer.py:52-54), one ``log.csv`` row of the runner's six per-game averages (:56-61), a sample of at most 5 x 2 048
recorded (state, Q) rows drawn without replacement (:63-72), mirror augmentation (flip the W axis of the states, swap
Q[left] and Q[right], :93-100), ``copy_and_compile(lr).train`` (:79-83), ``lr *= decay`` (:85), ``models/<name><n>.h5``
(:89-91).  The health curriculum is 9 for generations <= 8, 3 up to 32, 1 afterwards (:42-47).

Same constructor and ``train(nnet, name, iteration)`` signature; two additions, both inert by default:
  * ``max_iterations`` (keyword of ``train``): stop after that many generations instead of looping for ever (tests);
  * several GPUs: when ``torch.distributed`` is initialised (one process per GPU, ``torchrun train.py``) the games are
    cut into per-rank shards, every rank samples its share of the rows, the rows are all-gathered and the six counters
    all-reduced over RCCL (``snake_engine.dist``), every rank trains data-parallel (``utils.trainer_torch``) and ends with
    the same weights; rank 0 alone writes ``log.csv`` and the model files.
One deliberate difference: with fewer than 2 048 records the reference draws zero samples and fails inside ``flip`` (its
``samples > len(records)`` branch can never be taken); here all records form one batch, which is what that branch says.
"""
from random import sample
from time import time

import numpy as np

from utils.agent import Agent
from utils.mp_game_runner import MPGameRunner, LOG_FIELDS



def _dist():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        return dist
    return None


class AlphaSnakeZeroTrainer:

    def __init__(self, self_play_games, max_MCTS_depth, max_MCTS_breadth, learning_rate, learning_rate_decay,
                 height=11, width=11, snake_cnt=4, TPU=None):
        self.self_play_games = self_play_games
        self.max_MCTS_depth = max_MCTS_depth
        self.max_MCTS_breadth = max_MCTS_breadth
        self.lr = learning_rate
        self.lr_decay = learning_rate_decay
        self.height = height
        self.width = width
        self.snake_cnt = snake_cnt
        self.TPU = TPU               # accepted for signature parity; there is no TPU path

    # ---- pieces of one generation -------------------------------------------------------------------
    @staticmethod
    def health_dec_for(iteration):
        """trainer.py:42-47"""
        if iteration > 32:
            return 1
        if iteration > 8:
            return 3
        return 9

    # the two constructor calls of trainer.py:52-53; tests replace them to feed recorded start boards and draws
    def _make_agent(self, nnet, softmax_base, training, max_MCTS_depth, max_MCTS_breadth):
        return Agent(nnet, softmax_base, training, max_MCTS_depth, max_MCTS_breadth)

    def _make_runner(self, height, width, snake_cnt, health_dec, game_cnt):
        return MPGameRunner(height, width, snake_cnt, health_dec, game_cnt)

    def _self_play(self, nnet, iteration):
        """trainer.py:48-54 (+ the per-rank shard when several GPUs play)"""
        dist = _dist()
        n_games = self.self_play_games
        if dist is not None:
            from snake_engine.dist import shard_range
            lo, hi = shard_range(n_games, dist.get_rank(), dist.get_world_size())
            n_games = hi - lo
        alice = self._make_agent(nnet, 2 + iteration, True, self.max_MCTS_depth, self.max_MCTS_breadth)
        runner = self._make_runner(self.height, self.width, self.snake_cnt, self.health_dec_for(iteration), max(1, n_games))
        runner.run(alice)
        return alice, runner

    def _log_row(self, runner, iteration):
        """trainer.py:56-61: the six per-game averages (over all ranks' games)"""
        values = [getattr(runner, k) for k in LOG_FIELDS]
        dist = _dist()
        if dist is not None:
            from snake_engine.dist import all_reduce_counters
            totals = [v * runner.game_cnt for v in values]
            device = "cuda" if dist.get_backend() == "nccl" else "cpu"
            values, _ = all_reduce_counters(totals, runner.game_cnt, device)
        if dist is None or dist.get_rank() == 0:
            with open("log.csv", "a") as f:
                f.write(str(iteration) + ", " + ", ".join(str(v) for v in values) + "\n")
        return values

    def _collect(self, alice):
        """trainer.py:63-77: (X, V, batch_size) with the mirror images appended"""
        from snake_engine.dist import sample_plan
        n = len(alice.records)
        dist = _dist()
        if dist is None:
            wanted, batch_size, _ = sample_plan(n, 1)
            picked = sample(range(n), wanted)
            if hasattr(alice.records, "fetch"):        # the engine-backed records: one observe launch for all sampled rows
                X = list(alice.records.fetch(picked)) if wanted else []
            else:
                X = [alice.records[i] for i in picked]
            V = [alice.values[i] for i in picked]
        else:                                  # every rank contributes its share of the rows, all ranks get all rows
            import torch
            from snake_engine.dist import gather_counts, share_counts, sample_share, all_gather_samples
            world = dist.get_world_size()
            # every rank's record count: the batch count comes from ALL records (trainer.py:64-68), and a rank that has nothing to
            # sample from is every rank's error -- decided from the gathered counts, so that all ranks raise together instead of
            # one raising while its peers wait inside the all-gather
            counts, seed = gather_counts(n)
            n_all, n_empty = sum(counts), sum(c == 0 for c in counts)
            if n_empty:
                raise RuntimeError(f"{n_empty} of {world} ranks recorded no state to sample from (this rank: {n} records)")
            wanted, batch_size, _ = sample_plan(n_all, world)
            rows = share_counts(counts, wanted, seed)          # equal shares; a rank short of its share is topped up by the others
            idx = sample_share(n, rows[dist.get_rank()], np.random.RandomState(np.random.randint(1 << 31)))
            Xd = alice.records.fetch_device(idx)
            Vd = torch.as_tensor(alice._values_host()[idx], device=Xd.device)
            Xg, Vg = all_gather_samples(Xd, Vd, rows)
            X, V = list(Xg.cpu().numpy()), list(Vg.cpu().numpy())
        alice.clear()
        X += self.mirror_states(X)
        V += self.mirror_values(V)
        return X, V, batch_size

    # ---- the loop train.py starts ----------------------------------------------------------------------
    def train(self, nnet, name="AlphaSnake", iteration=0, max_iterations=None):
        dist = _dist()
        is_root = dist is None or dist.get_rank() == 0
        nnet = nnet.copy_and_compile()
        if iteration == 0 and is_root:         # trainer.py:35-41
            with open("log.csv", "a") as f:
                f.write("new model " + name + "\n")
                f.write("iteration, wall_collision, body_collision, head_collision, "
                        "starvation, food_eaten, game_length\n")
        done = 0
        while max_iterations is None or done < max_iterations:
            print("\nSelf playing games...")
            alice, runner = self._self_play(nnet, iteration)
            self._log_row(runner, iteration)
            X, V, batch_size = self._collect(alice)
            nnet = nnet.copy_and_compile(learning_rate=self.lr, TPU=self.TPU)
            t0 = time()
            nnet.train(X, V, batch_size=batch_size)
            print("Training time", time() - t0)
            nnet = nnet.copy_and_compile()
            self.lr *= self.lr_decay
            X = V = None
            iteration += 1
            done += 1
            if is_root:
                print("\nSaving the model " + name + str(iteration) + "...")
                nnet.save(name + str(iteration))
        return nnet

    def mirror_states(self, states):
        """flip the W axis of every (h, w, 3) state: a list again, so that ``X +=`` appends (trainer.py:93-97)"""
        return list(np.flip(states, axis=2)) if len(states) else []

    def mirror_values(self, values):
        """left <-> right (trainer.py:99-100)"""
        return list(np.flip(values, axis=1)) if len(values) else []
