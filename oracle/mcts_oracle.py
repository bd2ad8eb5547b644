"""CPU restatement of the reference's MCTS self-play loop (TEST INFRASTRUCTURE; see oracle/__init__.py).

Restates, over the C oracle's game primitives (oracle/snake_oracle.c):
  Agent.make_moves            agent.py:25-111      -> SelfPlayOracle.root_turn
  MCTSMPGameRunner.run        mp_game_runner.py:85-115
  MCTSAgent.make_moves        agent.py:161-223     -> SelfPlayOracle._rollout_epoch
  Agent.softermax / argmaxs   agent.py:114-137     -> softermax / argmaxs
  MPGameRunner.run            mp_game_runner.py:23-77 -> SelfPlayOracle.run
The caches are Python dicts keyed by the observation bytes, updated sequentially in ids order with
float32 NumPy arithmetic, exactly like the reference, so on the same taped draws the results are
comparable bit for bit with a run of the reference itself (tests/test_oracle_mcts.py).
Used as (a) the checker for the device MCTS and (b) the `cpu_baseline` of bench.py.
"""
import numpy as np

from oracle import snake_oracle as so


def softermax(base, z):                                   # agent.py:114-122
    normalized = np.power(base, np.arctanh(z))
    sigma = sum(normalized)
    if sigma == 0.0:
        return np.array([1.0 / 3.0] * 3, dtype=np.float32)
    return normalized / sigma


def argmaxs(Z):                                           # agent.py:124-137
    out = []
    for z in Z:
        if z[0] > z[1]:
            out.append(0 if z[0] > z[2] else 2)
        else:
            out.append(1 if z[1] > z[2] else 2)
    return out


class Draws:
    """numpy.random.choice([0,1,2], p=pmf) (agent.py:91,205) with the uniform taken from a tape or an RNG"""

    def __init__(self, tape=None, seed=0):
        self.tape = None if tape is None else np.asarray(tape, np.float64)
        self.pos = 0
        self.rs = np.random.RandomState(seed)

    def __call__(self, pmf):
        cdf = np.asarray(pmf, np.float64).cumsum()
        cdf /= cdf[-1]
        if self.tape is None:
            u = self.rs.random_sample()
        else:
            u = self.tape[self.pos]
        self.pos += 1
        return int(cdf.searchsorted(u, side="right"))


class SelfPlayOracle:
    def __init__(self, net, softmax_base=100, training=False, max_depth=8, max_breadth=128, draws=None):
        self.net, self.base, self.training = net, softmax_base, training
        self.max_depth, self.max_breadth = max_depth, max_breadth
        self.draw = draws or Draws()
        self.Q, self.total, self.visit, self.age = {}, {}, {}, {}
        self.records, self.values = [], []
        self.net_evals = 0
        self.sim_steps = 0

    # one lock-step batch of rollouts for every root game
    def _rollout_epoch(self, roots, parallel):
        Q, total, visit, age = self.Q, self.total, self.visit, self.age
        subs, depth = [], []
        for g in roots:
            d = self.max_depth - 2 * (len(g.alive_ids()) - 2)           # agent.py:45
            for _ in range(parallel):
                subs.append(g.subgame())
                depth.append(d)
        path = {(b, s): [] for b in range(len(subs)) for s in subs[b].alive_ids()}
        live = list(range(len(subs)))
        turn = 0
        while live:                                                      # mp_game_runner.py:91
            turn += 1
            rows = [(b, s) for b in live for s in subs[b].alive_ids()]
            states = [subs[b].make_state(s) for (b, s) in rows]
            keys = [st.tobytes() for st in states]
            V = [None] * len(rows)
            fresh = []
            for i, k in enumerate(keys):                                 # agent.py:172-186
                if k in Q:
                    if Q[k] is not None:
                        V[i] = Q[k]
                else:
                    fresh.append(states[i])
                    Q[k] = None
                age[k] = 0
            if fresh:                                                    # agent.py:189-201
                out = self.net.v(fresh)
                self.net_evals += len(fresh)
                j = 0
                for i, k in enumerate(keys):
                    if V[i] is None:
                        if Q[k] is None:
                            total[k] = out[j]
                            visit[k] = np.array([1.0] * 3, dtype=np.float32)
                            Q[k] = total[k] / visit[k]
                            j += 1
                        V[i] = Q[k]
            pmfs = [softermax(self.base, v) for v in V]                  # agent.py:204-205
            moves = [self.draw(p) for p in pmfs]
            for i, (b, s) in enumerate(rows):                            # agent.py:208-222
                est = pmfs[i] @ V[i]
                for (k, m) in reversed(path[(b, s)]):
                    visit[k][m] += 1.0
                    total[k][m] += est
                    Q[k][m] = total[k][m] / visit[k][m]
                path[(b, s)].append((keys[i], moves[i]))
            dense = {b: np.ones(subs[b].g.S, np.uint8) for b in live}
            for (b, s), m in zip(rows, moves):
                dense[b][s] = m
            nxt = []
            for b in live:                                               # mp_game_runner.py:104-113
                ended = subs[b].tic(dense[b])
                self.sim_steps += 1
                if not (ended or turn >= depth[b]):
                    nxt.append(b)
            live = nxt
        for (b, s), p in path.items():                                   # agent.py:60-72 (dict order = b, s ascending)
            r = subs[b].rewards[s]
            if r is not None:
                for (k, m) in reversed(p):
                    visit[k][m] += 1.0
                    total[k][m] += r
                    Q[k][m] = total[k][m] / visit[k][m]
        return path, parallel

    def root_turn(self, roots):
        """roots: list of live oracle Games.  Returns (rows [(root index, snake)], V list, moves list)."""
        for k in self.age:                                               # agent.py:30-31
            self.age[k] += 1
        parallel = min(8, self.max_breadth)
        path = None
        for _ in range(self.max_breadth // parallel):
            path, _ = self._rollout_epoch(roots, parallel)
        rows = [(g, s) for g in range(len(roots)) for s in roots[g].alive_ids()]
        V = [self.Q[path[(g * parallel, s)][0][0]] for (g, s) in rows]   # agent.py:74-87
        if self.training:                                                # agent.py:90-97
            moves = [self.draw(softermax(self.base, v)) for v in V]
            for g in roots:
                self.records += g.get_states()
            self.values += V
        else:
            moves = argmaxs(V)
        for k in [k for k, a in self.age.items() if a > self.max_depth]:  # agent.py:101-110
            del self.Q[k], self.total[k], self.visit[k], self.age[k]
        return rows, V, moves

    def run(self, games, spawn=None, rng=None, max_turns=None):
        """MPGameRunner.run (mp_game_runner.py:23-77) over oracle Games.
        spawn: callable (turn, game index) -> recorded spawn cell, or None -> draws from `rng`.
        Returns (rewards per game, env_steps)."""
        live = list(range(len(games)))
        rewards = [None] * len(games)
        env_steps = 0
        turn = 0
        rng = rng or np.random.RandomState(0)
        while live and (max_turns is None or turn < max_turns):
            turn += 1
            rows, _, moves = self.root_turn([games[g] for g in live])
            dense = {g: np.ones(games[g].g.S, np.uint8) for g in live}
            for (gi, s), m in zip(rows, moves):
                dense[live[gi]][s] = m
            nxt = []
            for g in live:
                if spawn is not None:
                    ended = games[g].tic(dense[g], spawn_cell=spawn(turn, g))
                else:
                    ended = games[g].tic(dense[g], draws=(rng.random_sample(), rng.random_sample()))
                env_steps += 1
                if ended:
                    rewards[g] = games[g].rewards
                else:
                    nxt.append(g)
            live = nxt
        return rewards, env_steps
