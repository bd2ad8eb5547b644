"""differential fuzz of the sub-rectangle form against the full form of the same kernels: random canvases (square or not),
net depths, numbers of sub-rectangle layers, chunk sizes and observations made of a background canvas with a random box of
random pixels (boxes of one pixel, boxes touching edges and corners, no box, no background) -- Q must be equal bit for bit.
   fuzz_rect.py [seed 0] [trials 200]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "alphasnake-zero_amd")]
import numpy as np
import torch
from snake_engine import net


def random_bn(ws, rng):
    out = [w.copy() for w in ws]
    k = 0
    while k < len(out):
        if out[k].ndim == 4:
            n = out[k].shape[3]
            out[k + 1] = (1.0 + 0.2 * rng.randn(n)).astype(np.float32)
            out[k + 2] = (0.1 * rng.randn(n)).astype(np.float32)
            out[k + 3] = (0.05 * rng.randn(n)).astype(np.float32)
            out[k + 4] = (0.5 + rng.rand(n)).astype(np.float32)
            k += 5
        else:
            k += 1
    return out


def random_planes(rng, n, h, w):
    x = np.tile(np.array(net.BACKGROUND_PIXEL, np.float32), (n, h, w, 1))
    for i in range(n):
        kind = rng.randint(0, 10)
        if kind == 0:
            continue                                             # all background
        if kind == 1:
            x[i] = rng.rand(h, w, 3)                             # no background
            continue
        bh, bw = (1, 1) if kind == 2 else (rng.randint(1, h + 1), rng.randint(1, w + 1))
        y0, x0 = rng.randint(0, h - bh + 1), rng.randint(0, w - bw + 1)
        if kind == 3:
            y0, x0 = (0 if rng.rand() < 0.5 else h - bh), (0 if rng.rand() < 0.5 else w - bw)     # in a corner
        patch = rng.rand(bh, bw, 3).astype(np.float32) * 2 - 0.5
        keep = rng.rand(bh, bw) < 0.3                            # some pixels inside the box are background too
        patch[keep] = np.array(net.BACKGROUND_PIXEL, np.float32)
        x[i, y0:y0 + bh, x0:x0 + bw] = patch
    return torch.as_tensor(x, device="cuda")


def run(seed=0, trials=200, verbose=True):
    rng = np.random.RandomState(seed)
    done = 0
    for t in range(trials):
        h, w = rng.randint(5, 38), rng.randint(5, 38)
        if rng.rand() < 0.5:
            w = h
        blocks = rng.randint(1, 4)
        n = rng.randint(48, 200)
        n_rect = rng.randint(1, 2 * blocks)
        ws = random_bn(net.glorot_uniform_weights((h, w, 3), blocks=blocks, seed=int(rng.randint(1 << 30))), rng)
        planes = random_planes(rng, n, h, w)
        os.environ["SNK_CONV_RECT"] = "0"
        full = net.QNet(ws, (h, w, 3), max_chunk=8192)
        os.environ["SNK_CONV_RECT"] = "1"
        os.environ["SNK_CONV_RECT_LAYERS"] = str(n_rect)
        rect = net.QNet(ws, (h, w, 3), max_chunk=int(rng.choice([8192, 64, 97])))
        rect.rect_min = 1
        del os.environ["SNK_CONV_RECT_LAYERS"]
        assert full.n_rect == 0 and rect.n_rect == n_rect
        q_full = full.forward(planes)
        for buf in rect._workspace(min(n, rect.max_chunk), 0):
            buf.fill_(float("nan"))
        q_rect = rect.forward(planes)
        assert torch.isfinite(q_full).all(), (t, h, w, blocks)
        assert torch.equal(q_full, q_rect), (t, h, w, blocks, n_rect, n, (q_full - q_rect).abs().max().item())
        done += 1
        if verbose and (t + 1) % 25 == 0:
            print(f"{t + 1} trials: all equal (last: {h}x{w}, {blocks} blocks, {n_rect} sub-rectangle layers, {n} observations)", flush=True)
    return done


if __name__ == "__main__":
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    trials = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    print(f"fuzz_rect seed {seed}: {run(seed, trials)} of {trials} nets equal bit for bit in both forms")
