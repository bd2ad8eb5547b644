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


def test_every_read_rectangle_was_written():
    """tower layer i computes box + (i + 2) and fills fill_grow[i] more (or the canvas); it reads its input on its rectangle
    grown by one and, for odd i, the shortcut (output of layer i - 2, the stem for i = 1) on its rectangle.  The stem
    writes box + 3 when two or more layers use the form, else the canvas."""
    rng = np.random.RandomState(0)
    for h, w, n_layers in ((21, 21, 8), (13, 13, 8), (37, 37, 20), (9, 9, 4), (21, 13, 6)):
        canvas = (0, 0, h - 1, w - 1)
        for n_rect in range(0, n_layers):
            fill = rect_fill_plan(n_rect)
            assert len(fill) == n_rect
            for _ in range(40):
                y0, y1 = sorted(rng.randint(0, h, 2)); x0, x1 = sorted(rng.randint(0, w, 2))
                box = (y0, x0, y1, x1)
                stem = _grow(box, 3, h, w) if n_rect >= 2 else canvas
                written = []
                for i in range(n_layers):
                    rect = _grow(box, i + 2, h, w) if i < n_rect else canvas
                    reads_in = _grow(rect, 1, h, w)
                    src = stem if i == 0 else written[i - 1]
                    assert _inside(reads_in, src), (h, n_rect, i, box)
                    if i % 2 == 1:
                        sc = stem if i == 1 else written[i - 2]
                        assert _inside(rect, sc), (h, n_rect, i, box)
                    if i < n_rect:
                        written.append(canvas if fill[i] < 0 else _grow(rect, fill[i], h, w))
                    else:
                        written.append(canvas)
