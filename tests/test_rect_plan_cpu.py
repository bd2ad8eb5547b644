"""Host logic of the sub-rectangle form (snake_engine/net.py): how many layers use it, and that every pixel a layer reads was
written by its producer -- checked by replaying the read / write rectangles of the tower on random bounding boxes."""
import numpy as np

from snake_engine.net import rect_layer_count, rect_fill_plan, BACKGROUND_PIXEL


def test_background_pixel_is_the_reference_wall():
    # game.py:218: grid = [[[0.0, WALL, 0.0] ...]] with WALL = 1.0 (game.py:4)
    assert BACKGROUND_PIXEL == (0.0, 1.0, 0.0)


def test_rect_layer_count_per_geometry():
    assert rect_layer_count(21, 21, 8) == 6           # 11x11 board, 4 blocks: layers 0-5
    assert rect_layer_count(13, 13, 8) == 3           # 7x7
    assert rect_layer_count(37, 37, 20) == 12         # 19x19, 10 blocks
    assert rect_layer_count(37, 37, 8) == 7           # never the last layer
    assert rect_layer_count(21, 21, 0) == 0 and rect_layer_count(21, 21, 1) == 0


def _grow(box, g, h, w):
    y0, x0, y1, x1 = box
    return max(y0 - g, 0), max(x0 - g, 0), min(y1 + g, h - 1), min(x1 + g, w - 1)


def _inside(a, b):
    return a[0] >= b[0] and a[1] >= b[1] and a[2] <= b[2] and a[3] <= b[3]


def test_every_pixel_a_layer_reads_is_current():
    """Tower layer i < n_rect computes box + (i + 2) only (the stem box + 1 when both of its readers are sub-rectangle
    layers).  A sub-rectangle reader is told what its producer computed (QNet._rect_args: grow_in = i + 1 for the input,
    grow_res = i for the shortcut) and takes the rest from the producer's background image -- the two numbers must be
    exactly the producer's rectangle; a FULL reader needs the whole canvas written, which rect_fill_plan arranges."""
    rng = np.random.RandomState(0)
    for h, w, n_layers in ((21, 21, 8), (13, 13, 8), (37, 37, 20), (9, 9, 4), (21, 13, 6)):
        canvas = (0, 0, h - 1, w - 1)
        for n_rect in range(0, n_layers):
            fills = rect_fill_plan(n_rect)
            assert len(fills) == n_rect and (n_rect == 0 or fills[-1])
            for _ in range(20):
                y0, y1 = sorted(rng.randint(0, h, 2)); x0, x1 = sorted(rng.randint(0, w, 2))
                box = (y0, x0, y1, x1)
                # what every producer leaves current in its output tensor: -1 = the stem
                written = {-1: _grow(box, 1, h, w) if n_rect >= 2 else canvas}
                computed = {-1: _grow(box, 1, h, w)}
                for i in range(n_layers):
                    rect = _grow(box, i + 2, h, w) if i < n_rect else canvas
                    readers = [(i - 1, i + 1)] + ([(i - 2, i)] if i % 2 == 1 else [])     # (producer, grow the reader is told)
                    for prod, told in readers:
                        if i < n_rect:          # takes box + told from the tensor, the rest from the producer's background
                            assert _grow(box, told, h, w) == computed[prod] and _inside(computed[prod], written[prod]), (h, n_rect, i, prod)
                        else:                   # reads the tensor wherever its taps reach
                            assert written[prod] == canvas, (h, n_rect, i, prod)
                    computed[i] = rect
                    written[i] = canvas if (i >= n_rect or fills[i]) else rect
