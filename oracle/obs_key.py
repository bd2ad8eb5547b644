"""Observation key + deterministic stub Q-net (NumPy restatement; TEST INFRASTRUCTURE).

The reference keys its transposition cache on the raw bytes of the per-snake
observation (``state.tostring()``, /root/reference/code/utils/agent.py:175) —
5 292 bytes for an 11x11 board.  The MI355X engine keys the same cache on a
128-bit digest of the *same observation*.  This file is the specification of
that digest on the dense observation, so the oracle, the golden generator and
the HIP ``obs_key`` kernel can be checked against each other:

    key = sum over canvas pixels p (row-major index in the rotated
          (2H-1)x(2W-1) frame) whose three float32 values differ from the
          wall default [0, 1.0, 0] of  mix(p, bits(ch0), bits(ch1), bits(ch2))
          (two 64-bit lanes, arithmetic mod 2**64).

A pixel equal to the wall default contributes nothing (an on-board body
segment with tail distance 50 encodes to exactly [0, 1.0, 0], game.py:239, and
is indistinguishable from a wall in the reference's key as well), so the sum
is a function of the observation bytes only: equal bytes <=> equal key up to
2**-128 collisions.

``stub_q`` is a deterministic stand-in for ``AlphaNNet.v`` used by parity
tests (the real net's arithmetic lives in TensorFlow, which is absent):
three float32 values in [-0.9, 0.9) derived from the key, then the
reference's obstacle mask (alpha_nnet.py:63-76).
"""
import numpy as np

M64 = np.uint64(0xFFFFFFFFFFFFFFFF)
_G = np.uint64(0x9E3779B97F4A7C15)
_C1 = np.uint64(0xBF58476D1CE4E5B9)
_C2 = np.uint64(0x94D049BB133111EB)
_HI = np.uint64(0xD6E8FEB86659FD93)


def sm64(x):
    """splitmix64 finaliser on uint64 arrays (wrap-around arithmetic)."""
    with np.errstate(over="ignore"):
        x = (x + _G)
        x = (x ^ (x >> np.uint64(30))) * _C1
        x = (x ^ (x >> np.uint64(27))) * _C2
        return x ^ (x >> np.uint64(31))


WALL_BITS = (np.uint32(0), np.float32(1.0).view(np.uint32), np.uint32(0))


def obs_key(states):
    """states: (N, h, w, 3) float32 (any strides). Returns (N, 2) uint64 [lo, hi]."""
    s = np.ascontiguousarray(states, dtype=np.float32)
    if s.ndim == 3:
        s = s[None]
    n = s.shape[0]
    bits = s.view(np.uint32).reshape(n, -1, 3).astype(np.uint64)
    npx = bits.shape[1]
    p = np.arange(npx, dtype=np.uint64)[None, :]
    b0, b1, b2 = bits[..., 0], bits[..., 1], bits[..., 2]
    live = ~((b0 == np.uint64(WALL_BITS[0])) & (b1 == np.uint64(WALL_BITS[1])) & (b2 == np.uint64(WALL_BITS[2])))
    x = sm64((p << np.uint64(32)) | b0)
    x = sm64(x ^ ((b1 << np.uint64(32)) | b2))
    lo = x
    hi = sm64(x ^ _HI)
    with np.errstate(over="ignore"):
        klo = np.where(live, lo, np.uint64(0)).sum(axis=1, dtype=np.uint64)
        khi = np.where(live, hi, np.uint64(0)).sum(axis=1, dtype=np.uint64)
    return np.stack([klo, khi], axis=1)


def obstacle_mask(states, legacy=False):
    """Restates alpha_nnet.py:63-76. Returns (N, 3) bool: left / straight / right blocked.

    ``legacy=False`` is the behaviour under this container's NumPy 2.x
    (float32(0.04) >= 0.04 is True: compare in float32).  ``legacy=True`` is the
    reference's pinned NumPy 1.18 behaviour (compare in float64).
    """
    s = np.asarray(states)
    if s.ndim == 3:
        s = s[None]
    cy, cx = s.shape[1] // 2, s.shape[2] // 2
    v = np.stack([s[:, cy, cx - 1, 1], s[:, cy - 1, cx, 1], s[:, cy, cx + 1, 1]], axis=1)
    if legacy:
        return v.astype(np.float64) >= 0.04
    return v >= np.float32(0.04)


def stub_q_from_key(klo):
    """float32 (N,3) in [-0.9, 0.9): exact same arithmetic on host, oracle and device."""
    klo = np.asarray(klo, dtype=np.uint64)
    out = np.empty(klo.shape + (3,), dtype=np.float32)
    for m in range(3):
        v = ((klo >> np.uint64(20 * m)) & np.uint64(0xFFFFF)).astype(np.float32)
        t = v * np.float32(2.0 ** -20)          # exact
        t = t * np.float32(1.8)                 # one rounding
        out[..., m] = t - np.float32(0.9)       # one rounding
    return out


def stub_q(states, legacy=False, which=0):
    """Deterministic Q(left, straight, right) with the reference's obstacle mask applied.
    ``which`` = 0 / 1 picks the low / high key word, 2 their sum mod 2**64: different "nets" for the pit runs."""
    k = obs_key(states)
    with np.errstate(over="ignore"):
        word = k[:, which] if which < 2 else k[:, 0] + k[:, 1]
    q = stub_q_from_key(word)
    q[obstacle_mask(states, legacy)] = np.float32(-1.0)
    return q


class StubNet:
    """Object with the ``nnet.v(list_of_states) -> (N,3) float32`` contract (alpha_nnet.py:61-73)."""

    def __init__(self, which=0):
        self.calls = []
        self.which = which

    def v(self, X):
        arr = np.array(X, dtype=np.float32)
        self.calls.append(len(X))
        return stub_q(arr, which=self.which)
