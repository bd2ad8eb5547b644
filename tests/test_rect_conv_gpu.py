"""GPU tests of the sub-rectangle form of the tower's 3x3 layers (csrc/conv_split.hip: k_obs_bbox, k_rect_plan,
k_conv3x3_f16s_rect; snake_engine/net.py: QNet.backgrounds / _rect_plan).

The reference's observation (game.py:215-257) is [0, WALL, 0] outside the board window, so a layer's output outside the
window grown by one pixel per layer is state-independent.  The form computes the grown window only and copies the rest
from a per-layer constant; it has to give the SAME BITS as the full convolution (each output pixel is the same chain of
MFMAs on the same operands), which is what these tests assert -- the 1e-5 parity of the net against the CPU restatement
(tests/test_net_gpu.py) runs through this form too, since it is the default."""
import ctypes as C

import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch
    assert torch.cuda.is_available()
    import snake_engine
    from snake_engine import net
    return torch, snake_engine, net


def _randomised_bn(ws, seed):
    rng = np.random.RandomState(seed)
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


def _mid_game_planes(se, torch, n, board, snakes, ticks, seed):
    eng = se.Engine(n, board, board, snakes, 1, 0.15, seed=seed)
    eng.reset()
    g = torch.Generator(device="cuda").manual_seed(seed)
    for _ in range(ticks):
        pairs = torch.nonzero(eng.alive()).to(torch.int32).contiguous()
        _, mask, _ = eng.observe_all(pairs, want_planes=False, want_key=False)
        pick = torch.multinomial((mask == 0).to(torch.float32) + 1e-3, 1, generator=g).squeeze(1).to(torch.uint8)
        mv = torch.ones((n, snakes), dtype=torch.uint8, device="cuda")
        mv[pairs[:, 0].long(), pairs[:, 1].long()] = pick
        eng.step(mv)
    pairs = torch.nonzero(eng.alive()).to(torch.int32).contiguous()
    planes, mask, _ = eng.observe_all(pairs)
    return planes, mask


def _special_observations(torch, h, w):
    """observations the engine never produces: all background, one foreign pixel in a corner / on an edge / in the
    centre, no background at all, a window touching two canvas edges"""
    bgp = torch.tensor([0.0, 1.0, 0.0])
    obs = []
    x = bgp.repeat(h, w, 1); obs.append(x.clone())
    for (y, xx) in ((0, 0), (h - 1, w - 1), (0, w // 2), (h // 2, 0), (h // 2, w // 2), (h - 1, 0)):
        t = x.clone(); t[y, xx] = torch.tensor([0.3, 0.2, 0.1]); obs.append(t)
    g = torch.Generator().manual_seed(9)
    obs.append(torch.rand(h, w, 3, generator=g))
    t = x.clone(); t[:h // 2 + 1, w // 2:] = torch.rand(h // 2 + 1, w - w // 2, 3, generator=g); obs.append(t)
    t = x.clone(); t[3, 5] = torch.tensor([float("nan"), 1.0, 0.0]); obs.append(t)        # NaN is not the background
    return torch.stack(obs).contiguous()


def _parts(hr, wr, H, W):
    """csrc/conv_split.hip hs_rect_parts: the fewest parts of at most 8 tiles whose strip fits the LDS buffer (352 pixels)
    and the staging items (320 pixels)"""
    T = (hr * wr + 31) // 32
    parts = (T + 7) // 8
    while True:
        tm = (T + parts - 1) // parts
        rows_out = min((tm * 32 + wr - 2) // wr + 1, hr)
        if (rows_out + 2) * (wr + 2) <= 352 and min(rows_out + 2, H) * min(wr + 2, W) <= 320:
            return parts
        assert tm > 1
        parts += 1


def test_plan_matches_a_numpy_model(env):
    """bounding boxes, rectangles and parts of the plan against a NumPy restatement; every image's tiles are covered exactly
    once, blocks come largest first"""
    torch, se, net = env
    from snake_engine._lib import lib, check
    L = lib()
    h = w = 21
    s = load_golden("states_11x11x4.npz")
    planes = torch.cat([torch.as_tensor(s["raw"]), _special_observations(torch, h, w)]).cuda().contiguous()
    n = planes.shape[0]
    grow = [2, 3, 4, 7]
    mb = L.snk_conv_rect_max_blocks(n, h, w)
    assert mb == n * max(_parts(a, b, h, w) for a in range(1, h + 1) for b in range(1, w + 1))
    desc = torch.full((len(grow), mb, 4), -1, dtype=torch.int32, device="cuda")
    counts = torch.zeros((len(grow), 2), dtype=torch.int32, device="cuda")
    bbox = torch.zeros(n, dtype=torch.int32, device="cuda")
    check(L.snk_conv_rect_plan(planes.data_ptr(), 0.0, 1.0, 0.0, n, h, w, len(grow), (C.c_int * 4)(*grow), bbox.data_ptr(), desc.data_ptr(), counts.data_ptr(), torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    ph = planes.cpu().numpy()
    bb = bbox.cpu().numpy().view(np.uint32)
    boxes = []
    for i in range(n):
        nz = np.argwhere(~((ph[i, :, :, 0] == 0.0) & (ph[i, :, :, 1] == 1.0) & (ph[i, :, :, 2] == 0.0)))
        if len(nz) == 0:
            y0 = y1 = h // 2; x0 = x1 = w // 2
        else:
            (y0, x0), (y1, x1) = nz.min(0), nz.max(0)
        boxes.append((y0, x0, y1, x1))
        assert (bb[i] & 255, (bb[i] >> 8) & 255, (bb[i] >> 16) & 255, bb[i] >> 24) == (y0, x0, y1, x1), i
    # the golden observations' boxes are the 11 x 11 board window (the observer's head at the canvas centre)
    assert all(b[2] - b[0] == 10 and b[3] - b[1] == 10 and b[0] <= 10 <= b[2] for b in boxes[:230])
    dh, ch = desc.cpu().numpy().view(np.uint32), counts.cpu().numpy()
    for l, g in enumerate(grow):
        nd = int(ch[l, 0])
        d = dh[l, :nd]
        assert (dh[l, nd:] == 0xFFFFFFFF).all()                          # nothing written past the count
        ntile = (d[:, 2] >> 8) & 255
        assert (ntile >= 1).all() and (ntile <= 8).all()
        seen = {}
        for img, rect, tl, fl in d:
            seen.setdefault(int(img), []).append((int(rect), int(tl), int(fl)))
        assert sorted(seen) == list(range(n))
        tiles_total = 0
        order = []
        for img, rect, tl, fl in d:
            if ((tl >> 16) & 255) == 0:
                order.append(int(img))
        for i in range(n):
            y0, x0, y1, x1 = boxes[i]
            ry0, rx0, ry1, rx1 = max(y0 - g, 0), max(x0 - g, 0), min(y1 + g, h - 1), min(x1 + g, w - 1)
            hr, wr = ry1 - ry0 + 1, rx1 - rx0 + 1
            T = (hr * wr + 31) // 32
            tiles_total += T
            parts = seen[i]
            assert all(p[0] == (ry0 | rx0 << 8 | hr << 16 | wr << 24) for p in parts), (l, i)
            assert len(parts) == _parts(hr, wr, h, w) and all((p[1] >> 24) == len(parts) for p in parts)
            cover = sorted((p[1] & 255, (p[1] >> 8) & 255, (p[1] >> 16) & 255) for p in parts)
            nxt = 0
            for k, (t0, nt, part) in enumerate(cover):
                assert t0 == nxt and part == k
                nxt += nt
            assert nxt == T
            assert all(p[2] == int(bb[i]) for p in parts)               # the box itself rides along (what the producers computed)
        assert int(ch[l, 1]) == tiles_total
        # largest first: the largest part of the images in descriptor order never grows
        biggest = {i: max((p[1] >> 8) & 255 for p in seen[i]) for i in range(n)}
        seq = [biggest[i] for i in order]
        assert seq == sorted(seq, reverse=True)


@pytest.mark.parametrize("board,snakes,blocks,n_games,n_rect", [(11, 4, 4, 160, None), (7, 2, 4, 64, None), (11, 4, 4, 64, 7),
                                                                  (19, 8, 10, 6, None), (5, 2, 2, 32, 3), (11, 4, 4, 1900, None)])
def test_rect_form_gives_the_same_bits_as_the_full_form(env, board, snakes, blocks, n_games, n_rect, monkeypatch):
    """whole net: Q of mid-game observations (and of hand-made ones: blank, single foreign pixels on edges and corners, no
    background at all) through the sub-rectangle layers == through the full layers, bit for bit; and every sub-rectangle
    layer's output on its rectangle (on the whole canvas where it fills) == the full layer's output there"""
    torch, se, net = env
    h = w = 2 * board - 1
    ws = _randomised_bn(net.glorot_uniform_weights((h, w, 3), blocks=blocks, seed=board), 5)
    planes, mask = _mid_game_planes(se, torch, n_games, board, snakes, 12, seed=100 + board)
    planes = torch.cat([planes, _special_observations(torch, h, w).cuda()]).contiguous()
    planes = torch.nan_to_num(planes, nan=3.0)
    m = planes.shape[0]
    monkeypatch.setenv("SNK_CONV_RECT", "0")
    full = net.QNet(ws, (h, w, 3), max_chunk=8192)
    assert full.n_rect == 0
    monkeypatch.setenv("SNK_CONV_RECT", "1")
    if n_rect is not None:
        monkeypatch.setenv("SNK_CONV_RECT_LAYERS", str(n_rect))
    rect = net.QNet(ws, (h, w, 3), max_chunk=8192)
    rect.rect_min = 1                       # (chunks below 48 observations take the full form by default)
    assert rect.n_rect == (n_rect if n_rect is not None else net.rect_layer_count(h, w, 2 * blocks)) and rect.n_rect >= 1
    q_full = full.forward(planes)
    for t in rect._workspace(m, 0):                 # whatever the form leaves unwritten must never be read
        t.fill_(float("nan"))
    q_rect = rect.forward(planes)
    assert torch.isfinite(q_full).all()
    assert torch.equal(q_full, q_rect), (q_full - q_rect).abs().max().item()
    # ragged chunks take the same route
    rect.max_chunk = 37
    assert torch.equal(rect.forward(planes), q_full)
    rect.max_chunk = 8192
    if board == 11 and n_rect is None:      # and the form is within the 1e-5 contract of the CPU restatement (oracle/net_ref.py)
        from oracle import net_ref
        ref = net_ref.forward(ws, planes[:48].cpu().numpy(), apply_mask=False)
        assert np.abs(q_rect[:48].cpu().numpy() - ref).max() <= 1e-5
        rect.rect_min = 48                  # the default: a chunk of 47 observations takes the full form, 48 the sub-rectangles
        assert torch.equal(rect.forward(planes[:47]), q_full[:47]) and torch.equal(rect.forward(planes[:48]), q_full[:48])
        rect.rect_min = 1

    if m > 4000:                            # the bench's chunk size class: the whole-net comparison is the test
        return
    # layer by layer
    st = torch.cuda.current_stream().cuda_stream
    from snake_engine._lib import check

    def tower(qn, use_plan):
        bufs = [torch.full((m, h, w, 128), float("nan"), device="cuda") for _ in range(3)]
        check(qn.L.snk_stem_conv_bn_relu_f32(planes.data_ptr(), qn.stem_w.data_ptr(), qn.stem_sc.data_ptr(), qn.stem_sh.data_ptr(),
                                             bufs[0].data_ptr(), m, h, w, st))
        plan = qn._rect_plan(planes, m, 0, st) if use_plan else None
        outs, (cur, t1, t2) = [], bufs
        for i in range(rect.n_rect):
            if i % 2 == 0:
                qn._conv(i, cur, None, t1, m, st, plan=plan); outs.append(t1.clone())
            else:
                qn._conv(i, t1, cur, t2, m, st, plan=plan); outs.append(t2.clone())
                cur, t2 = t2, cur
        return outs, plan
    o_full, _ = tower(full, False)
    o_rect, plan = tower(rect, True)
    desc = plan[0].cpu().numpy().view(np.uint32)
    counts = plan[1].cpu().numpy()
    saved = 0
    for i in range(rect.n_rect):
        valid = torch.zeros((m, h, w), dtype=torch.bool)
        for img, rc, _, _ in desc[i, :counts[i, 0]]:
            fy, fx, fh, fw = rc & 255, (rc >> 8) & 255, (rc >> 16) & 255, rc >> 24
            valid[int(img), fy:fy + fh, fx:fx + fw] = True
        assert valid.view(m, -1).any(1).all()
        v = valid.cuda()
        if rect.rect_fill[i]:
            v = torch.ones_like(v)
        elif i < 2:                                                     # (first use of the NaN-filled buffer)
            assert torch.isnan(o_rect[i][~v]).all()                     # nothing else is written
        a, b = o_full[i][v], o_rect[i][v]
        assert torch.isfinite(a).all()
        assert torch.equal(a, b), (i, (a - b).abs().max().item())
        saved += m * ((h * w + 31) // 32) - int(counts[i, 1])
    assert saved > 0


@pytest.mark.parametrize("algo", ["f16a", "bf16"])
@pytest.mark.parametrize("board,snakes,blocks,n_games", [(11, 4, 4, 96), (19, 8, 10, 6)])
def test_f16_activation_tower_rect_form_gives_the_same_bits(env, board, snakes, blocks, n_games, algo, monkeypatch):
    """the reduced-precision towers with 16-bit activations (SNK_CONV_ALGO=f16a / bf16, BASELINE configs[4]) through their
    sub-rectangle layers == through their full layers, bit for bit"""
    torch, se, net = env
    monkeypatch.setenv("SNK_CONV_ALGO", algo)
    h = w = 2 * board - 1
    ws = _randomised_bn(net.glorot_uniform_weights((h, w, 3), blocks=blocks, seed=board), 6)
    planes, _ = _mid_game_planes(se, torch, n_games, board, snakes, 10, seed=200 + board)
    planes = torch.nan_to_num(torch.cat([planes, _special_observations(torch, h, w).cuda()]), nan=3.0).contiguous()
    monkeypatch.setenv("SNK_CONV_RECT", "0")
    full = net.QNet(ws, (h, w, 3), max_chunk=8192)
    monkeypatch.setenv("SNK_CONV_RECT", "1")
    rect = net.QNet(ws, (h, w, 3), max_chunk=8192)
    rect.rect_min = 1
    assert full.n_rect == 0 and rect.n_rect >= 2 and rect.backgrounds().dtype == (torch.float16 if algo == "f16a" else torch.bfloat16)
    q_full = full.forward(planes)
    for t in rect._ws.get(("a16", 0), []) if rect._ws else []:
        t.fill_(float("nan"))
    q_rect = rect.forward(planes)
    for t in rect._ws[("a16", 0)][:3]:
        t.fill_(float("nan"))
    assert torch.isfinite(q_full).all() and torch.equal(q_full, q_rect) and torch.equal(rect.forward(planes), q_full)


def test_bounded_fuzz_of_the_rect_form(env):
    """tools/fuzz_rect.py, bounded: random canvases (square or not), depths, layer counts, chunkings and synthetic
    observations (one-pixel boxes, boxes in corners, no box, no background) -- both forms equal bit for bit"""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("fuzz_rect", os.path.join(os.path.dirname(__file__), "..", "tools", "fuzz_rect.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    keep = {k: os.environ.get(k) for k in ("SNK_CONV_RECT", "SNK_CONV_RECT_LAYERS")}
    try:
        assert fz.run(seed=11, trials=30, verbose=False) == 30
    finally:
        for k, v in keep.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def test_self_play_takes_the_same_moves_in_both_forms(env, monkeypatch):
    """the whole path -- MPGameRunner + Agent (production-mode MCTS, Philox draws) + AlphaNNet -- with the same seeds: three
    root turns of 96 games end on the same boards with the same recorded values and evaluation counts whether the tower's
    first layers run on sub-rectangles or on the whole canvas (bit-identical Q values -> identical draws and moves)"""
    torch, se, net = env
    from utils.agent import Agent
    from utils.alpha_nnet import AlphaNNet
    from utils.mp_game_runner import MPGameRunner
    import random
    MPGameRunner.verbose = False
    monkeypatch.setattr(MPGameRunner, "init", "device")      # start boards from the seeded device generator, as bench.py does
    ws = _randomised_bn(net.glorot_uniform_weights((21, 21, 3), blocks=4, seed=3), 9)
    out = {}
    for form in ("0", "1", "1"):
        random.seed(5); np.random.seed(5)
        monkeypatch.setenv("SNK_CONV_RECT", form)
        nn = AlphaNNet(input_shape=(21, 21, 3), _weights=ws)
        assert nn._qnet.n_rect == (6 if form == "1" else 0)
        alice = Agent(nn, 2, True, 8, 16, seed=77)
        gr = MPGameRunner(11, 11, 4, 1, 96, seed=78)
        gr.run(alice, max_turns=3)
        states = gr.engine.export()
        if form in out:                                       # the path is deterministic: the same form twice is the same run
            again = (gr.env_steps, alice._mcts.stats["net_evals"], len(alice.records), bytes(gr.engine.export()))
            assert again == out[form][:3] + (out[form][4],)
            continue
        out[form] = (gr.env_steps, alice._mcts.stats["net_evals"], len(alice.records), np.array(alice.values[:len(alice.records)]),
                     bytes(states))
    a, b = out["0"], out["1"]
    assert a[0] == b[0] and a[1] == b[1] and a[2] == b[2] and a[1] > 5000
    assert np.array_equal(a[3], b[3]) and a[4] == b[4]


def test_backgrounds_follow_the_weights_and_scales(env):
    """the per-layer constants are made again after set_weights and after a change of an activation scale (they are kept
    bit-identical to what the full layers compute)"""
    torch, se, net = env
    ws = net.glorot_uniform_weights((21, 21, 3), blocks=4, seed=2)
    qn = net.QNet(ws, (21, 21, 3))
    bg0 = qn.backgrounds().clone()
    assert bg0.shape == (7, 21, 21, 128) and torch.isfinite(bg0).all()          # the stem's and six tower layers'
    # far from the canvas edge the background of a layer is one pixel value (nothing there sees the zero padding)
    assert torch.equal(bg0[1, 5, 5], bg0[1, 12, 9]) and not torch.equal(bg0[1, 0, 0], bg0[1, 5, 5])
    assert torch.equal(bg0[0, 1, 1], bg0[0, 12, 9]) and not torch.equal(bg0[0, 0, 0], bg0[0, 1, 1])
    qn.set_x_scale(2, qn.conv_x_scale[2] / 4)
    assert qn._bg is None
    qn.set_weights(_randomised_bn(ws, 8))
    assert qn._bg is None and not torch.equal(qn.backgrounds(), bg0)
    # weights of another depth through the same object: the plan's buffers follow the new layer count
    x = _special_observations(torch, 21, 21).cuda().repeat(6, 1, 1, 1).nan_to_num(nan=2.0).contiguous()
    q8 = qn.forward(x)
    ws2 = net.glorot_uniform_weights((21, 21, 3), blocks=2, seed=4)
    qn.set_weights(ws2)
    assert qn.n_rect == 3 and qn.backgrounds().shape[0] == 4
    monkey_full = net.QNet(ws2, (21, 21, 3)); monkey_full.n_rect = 0
    assert torch.equal(qn.forward(x), monkey_full.forward(x)) and torch.isfinite(q8).all()


def test_rect_entry_points_refuse_bad_arguments(env):
    torch, se, net = env
    from snake_engine._lib import lib
    L = lib()
    assert L.snk_conv_rect_max_blocks(4, 2, 21) < 0 and L.snk_conv_rect_max_blocks(4, 21, 81) < 0
    assert L.snk_conv_rect_max_blocks(0, 21, 21) == 0
    x = torch.zeros(8, device="cuda")
    g = (C.c_int * 1)(2)
    assert L.snk_conv_rect_plan(None, 0.0, 1.0, 0.0, 1, 21, 21, 1, g, x.data_ptr(), x.data_ptr(), x.data_ptr(), None) < 0
    assert L.snk_conv_rect_plan(x.data_ptr(), 0.0, 1.0, 0.0, 1, 21, 21, 25, g, x.data_ptr(), x.data_ptr(), x.data_ptr(), None) < 0
    assert b"layers" in L.snk_last_error()
    assert L.snk_conv3x3_bn_f16s_rect(x.data_ptr(), x.data_ptr(), x.data_ptr(), x.data_ptr(), None, x.data_ptr(), x.data_ptr(),
                                      x.data_ptr(), None, 1, None, 0, None, 1, 21, 21, None) < 0           # in place
    y = torch.zeros(8, device="cuda")
    assert L.snk_conv3x3_bn_f16s_rect(x.data_ptr(), x.data_ptr(), x.data_ptr(), x.data_ptr(), None, y.data_ptr(), x.data_ptr(),
                                      x.data_ptr(), None, 200, None, 0, None, 1, 21, 21, None) < 0 and b"grow_in" in L.snk_last_error()
