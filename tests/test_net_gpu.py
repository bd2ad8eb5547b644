"""GPU tests of the Q-net inference kernels (csrc/net.hip) through the C ABI: each layer against a
plain PyTorch reference of the same op, and the whole AlphaNNet.v against the fp32 CPU restatement
(oracle/net_ref.py).  Tolerance: |dQ| <= 1e-5 (BASELINE.json north_star), layers 2e-5 relative to the
activation scale (fp32 accumulation order differs between a k-ordered MFMA chain and the reference)."""
import ctypes as C

import os

import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu
TOL_Q = 1e-5


@pytest.fixture(scope="module")
def env():
    import torch
    assert torch.cuda.is_available()
    import snake_engine
    from snake_engine import net
    return torch, snake_engine, net


def _st():
    import torch
    return torch.cuda.current_stream().cuda_stream


@pytest.mark.parametrize("algo", ["direct", "winograd", "f16s"])
@pytest.mark.parametrize("n,hw,res,relu", [(3, 21, False, True), (5, 21, True, True), (1, 21, True, False), (2, 37, True, True),
                                           (3, 13, True, True), (2, 5, False, True)])
def test_conv3x3_layer(env, n, hw, res, relu, algo):
    torch, se, _ = env
    from snake_engine._lib import lib, check
    g = torch.Generator().manual_seed(n * 100 + hw)
    x = torch.randn(n, hw, hw, 128, generator=g)
    w = torch.randn(3, 3, 128, 128, generator=g) * 0.05
    sc = torch.rand(128, generator=g) + 0.5
    sh = torch.randn(128, generator=g) * 0.1
    r = torch.randn(n, hw, hw, 128, generator=g) if res else None
    ref = torch.nn.functional.conv2d(x.double().permute(0, 3, 1, 2), w.double().permute(3, 2, 0, 1), padding=1).permute(0, 2, 3, 1)
    ref = ref * sc.double() + sh.double()
    if res:
        ref = ref + r.double()
    if relu:
        ref = ref.clamp_min(0)
    xd, wd, scd, shd = x.cuda(), w.cuda().contiguous(), sc.cuda(), sh.cuda()
    rd = r.cuda() if res else None
    wT = torch.empty(16 * 128 * 128, device="cuda")
    out = torch.full((n, hw, hw, 128), float("nan"), device="cuda")
    L = lib()
    prep, conv = {"winograd": (L.snk_conv3x3_prepare_weights_winograd, L.snk_conv3x3_bn_f32_winograd),
                  "f16s": (L.snk_conv3x3_prepare_weights_f16s, L.snk_conv3x3_bn_f16s),
                  "direct": (L.snk_conv3x3_prepare_weights, L.snk_conv3x3_bn_f32)}[algo]
    if algo == "f16s":
        check(prep(wd.data_ptr(), wT.data_ptr(), C.c_float(256.0), _st()))     # |x| up to ~5: 5 * 256 << 65504
    else:
        check(prep(wd.data_ptr(), wT.data_ptr(), _st()))
    check(conv(xd.data_ptr(), wT.data_ptr(), scd.data_ptr(), shd.data_ptr(), rd.data_ptr() if res else None,
               out.data_ptr(), n, hw, hw, int(relu), _st()))
    got = out.cpu().double()
    assert torch.isfinite(got).all()
    err = (got - ref).abs().max().item()
    scale = ref.abs().max().item()
    assert err <= 2e-5 * scale, (err, scale)


@pytest.mark.parametrize("n,H,W", [(2, 13, 21), (3, 21, 13), (5, 3, 3), (1, 4, 80), (2, 9, 37), (9, 21, 21), (1, 1, 5)])
def test_f16s_rectangular_and_edge_shapes(env, n, H, W):
    """the split-f16 layer on non-square images, the widest supported row, single-row images and a batch that is not a
    multiple of the XCD-aware group of 8 images"""
    torch, se, _ = env
    from snake_engine._lib import lib, check
    from snake_engine.net import F16S_WEIGHT_BYTES
    L = lib()
    g = torch.Generator().manual_seed(1000 * H + W)
    x = torch.randn(n, H, W, 128, generator=g)
    r = torch.randn(n, H, W, 128, generator=g)
    w = torch.randn(3, 3, 128, 128, generator=g) * 0.05
    sc, sh = torch.rand(128, generator=g) + 0.5, torch.randn(128, generator=g) * 0.1
    ref = torch.nn.functional.conv2d(x.double().permute(0, 3, 1, 2), w.double().permute(3, 2, 0, 1), padding=1).permute(0, 2, 3, 1)
    ref = (ref * sc.double() + sh.double() + r.double()).clamp_min(0)
    xd, rd, wd, scd, shd = x.cuda(), r.cuda(), w.cuda(), sc.cuda(), sh.cuda()
    wS = torch.empty(F16S_WEIGHT_BYTES, dtype=torch.uint8, device="cuda")
    out = torch.full((n + 1, H, W, 128), float("nan"), device="cuda")          # one image of slack: nothing may be written there
    check(L.snk_conv3x3_prepare_weights_f16s(wd.data_ptr(), wS.data_ptr(), C.c_float(256.0), _st()))
    check(L.snk_conv3x3_bn_f16s(xd.data_ptr(), wS.data_ptr(), scd.data_ptr(), shd.data_ptr(), rd.data_ptr(),
                                out.data_ptr(), n, H, W, 1, _st()))
    got = out.cpu().double()
    assert torch.isnan(got[n]).all()
    err = (got[:n] - ref).abs().max().item()
    assert err <= 5e-6 * max(1.0, ref.abs().max().item()), err


def test_f16s_activation_scale(env):
    """the split-f16 layer keeps float32-level accuracy for small activations when the caller passes the matching power
    of two (hi + lo f16 carry 22 significand bits only while lo is a normal f16 number), and rejects other scales"""
    torch, se, _ = env
    from snake_engine._lib import lib, check, EngineError
    from snake_engine.net import F16S_WEIGHT_BYTES
    L = lib()
    g = torch.Generator().manual_seed(77)
    n, hw = 2, 21
    w = torch.randn(3, 3, 128, 128, generator=g) * 0.05
    sc, sh0 = torch.rand(128, generator=g) + 0.5, torch.randn(128, generator=g)
    wd, scd = w.cuda(), sc.cuda()
    wS = torch.empty(F16S_WEIGHT_BYTES, dtype=torch.uint8, device="cuda")
    for mag, x_scale in ((1e-4, 2.0 ** 21), (1.0, 256.0), (300.0, 1.0)):
        x = torch.randn(n, hw, hw, 128, generator=g) * mag
        sh = sh0 * mag
        ref = torch.nn.functional.conv2d(x.double().permute(0, 3, 1, 2), w.double().permute(3, 2, 0, 1), padding=1).permute(0, 2, 3, 1)
        ref = (ref * sc.double() + sh.double() + x.double()).clamp_min(0)
        xd, shd = x.cuda(), sh.cuda()
        out = torch.full((n, hw, hw, 128), float("nan"), device="cuda")
        check(L.snk_conv3x3_prepare_weights_f16s(wd.data_ptr(), wS.data_ptr(), C.c_float(x_scale), _st()))
        check(L.snk_conv3x3_bn_f16s(xd.data_ptr(), wS.data_ptr(), scd.data_ptr(), shd.data_ptr(), xd.data_ptr(),
                                    out.data_ptr(), n, hw, hw, 1, _st()))
        err = (out.cpu().double() - ref).abs().max().item()
        assert err <= 5e-6 * ref.abs().max().item(), (mag, err, ref.abs().max().item())
    with pytest.raises(EngineError):
        check(L.snk_conv3x3_prepare_weights_f16s(wd.data_ptr(), wS.data_ptr(), C.c_float(3.0), _st()))
    with pytest.raises(EngineError):
        check(L.snk_conv3x3_bn_f16s(xd.data_ptr(), wS.data_ptr(), scd.data_ptr(), shd.data_ptr(), None,
                                    out.data_ptr(), 1, 4, 200, 1, _st()))      # wider than the LDS staging allows: refused, no launch
    with pytest.raises(EngineError):
        check(L.snk_conv3x3_bn_f16s(xd.data_ptr(), wS.data_ptr(), scd.data_ptr(), shd.data_ptr(), None,
                                    out.data_ptr(), 1, 4, 2, 1, _st()))


def test_activation_report_shows_the_f16s_headroom(env):
    """QNet.activation_report: on the gen-0 net and on a net with randomised batch-norm the layer inputs stay far inside
    the range the batch-norm-derived power-of-two scales leave (the kernel clamps beyond 65504 / x_scale)"""
    torch, se, net = env
    s = load_golden("states_11x11x4.npz")
    states = torch.as_tensor(s["raw"][:64], device="cuda")
    for ws in (net.glorot_uniform_weights((21, 21, 3), blocks=4, seed=0),
               _randomised_bn(net.glorot_uniform_weights((21, 21, 3), blocks=4, seed=0), 3)):
        rep = net.QNet(ws, (21, 21, 3)).activation_report(states)
        assert len(rep) == 8
        for amax, xs, headroom in rep:
            assert amax > 0 and headroom >= 64, (amax, xs, headroom)


def test_f16s_fused_head_equals_layer_plus_head(env):
    """snk_conv3x3_bn_f16s_head + snk_head_dense_f32 == snk_conv3x3_bn_f16s(relu) + snk_head_f32 (alpha_nnet.py:46-54)"""
    torch, se, _ = env
    from snake_engine._lib import lib, check
    from snake_engine.net import F16S_WEIGHT_BYTES
    L = lib()
    g = torch.Generator().manual_seed(123)
    n, hw = 37, 21                      # 37 states: three dense-head blocks, the last one ragged
    x = torch.randn(n, hw, hw, 128, generator=g).cuda()
    r = torch.randn(n, hw, hw, 128, generator=g).cuda()
    w = (torch.randn(3, 3, 128, 128, generator=g) * 0.05).cuda()
    sc, sh = (torch.rand(128, generator=g) + 0.5).cuda(), (torch.randn(128, generator=g) * 0.1).cuda()
    w1 = (torch.randn(128, generator=g) * 0.1).cuda()
    s1, b1 = 0.7, 0.05
    f1w, f1b = (torch.randn(hw * hw, 128, generator=g) * 0.05).cuda(), (torch.randn(128, generator=g) * 0.1).cuda()
    f2w, f2b = (torch.randn(128, 3, generator=g) * 0.1).cuda(), (torch.randn(3, generator=g) * 0.1).cuda()
    mask = (torch.rand(n, 3, generator=g) < 0.2).to(torch.uint8).cuda()
    wS = torch.empty(F16S_WEIGHT_BYTES, dtype=torch.uint8, device="cuda")
    check(L.snk_conv3x3_prepare_weights_f16s(w.data_ptr(), wS.data_ptr(), C.c_float(256.0), _st()))
    act = torch.full((n, hw, hw, 128), float("nan"), device="cuda")
    q_ref = torch.full((n, 3), float("nan"), device="cuda")
    check(L.snk_conv3x3_bn_f16s(x.data_ptr(), wS.data_ptr(), sc.data_ptr(), sh.data_ptr(), r.data_ptr(), act.data_ptr(), n, hw, hw, 1, _st()))
    check(L.snk_head_f32(act.data_ptr(), w1.data_ptr(), C.c_float(s1), C.c_float(b1), f1w.data_ptr(), f1b.data_ptr(),
                         f2w.data_ptr(), f2b.data_ptr(), mask.data_ptr(), q_ref.data_ptr(), n, hw, hw, _st()))
    h1_ref = ((act.double().reshape(n, hw * hw, 128) @ w1.double()) * s1 + b1).clamp_min(0)
    for keep_out in (True, False):
        act2 = torch.full((n, hw, hw, 128), float("nan"), device="cuda")
        h1 = torch.full((n, hw * hw), float("nan"), device="cuda")
        q = torch.full((n, 3), float("nan"), device="cuda")
        check(L.snk_conv3x3_bn_f16s_head(x.data_ptr(), wS.data_ptr(), sc.data_ptr(), sh.data_ptr(), r.data_ptr(),
                                         act2.data_ptr() if keep_out else None, w1.data_ptr(), C.c_float(s1), C.c_float(b1),
                                         h1.data_ptr(), n, hw, hw, _st()))
        check(L.snk_head_dense_f32(h1.data_ptr(), f1w.data_ptr(), f1b.data_ptr(), f2w.data_ptr(), f2b.data_ptr(),
                                   mask.data_ptr(), q.data_ptr(), n, hw, hw, _st()))
        if keep_out:
            assert torch.equal(act2, act)
        assert (h1.double() - h1_ref).abs().max().item() <= 2e-6 * max(1.0, h1_ref.abs().max().item())
        assert (q - q_ref).abs().max().item() <= 2e-6
        assert (q[mask.bool()] == -1.0).all()


def test_stem_and_head_layers(env):
    torch, se, _ = env
    from snake_engine._lib import lib, check
    L = lib()
    g = torch.Generator().manual_seed(5)
    n, hw = 4, 21
    x = torch.randn(n, hw, hw, 3, generator=g)
    w = torch.randn(3, 3, 3, 128, generator=g) * 0.2
    sc, sh = torch.rand(128, generator=g) + 0.5, torch.randn(128, generator=g) * 0.1
    ref = torch.nn.functional.conv2d(x.double().permute(0, 3, 1, 2), w.double().permute(3, 2, 0, 1), padding=1).permute(0, 2, 3, 1)
    ref = (ref * sc.double() + sh.double()).clamp_min(0)
    out = torch.full((n, hw, hw, 128), float("nan"), device="cuda")
    xd, wd, scd, shd = x.cuda(), w.cuda(), sc.cuda(), sh.cuda()     # keep the device copies alive across the launch
    check(L.snk_stem_conv_bn_relu_f32(xd.data_ptr(), wd.data_ptr(), scd.data_ptr(), shd.data_ptr(),
                                      out.data_ptr(), n, hw, hw, _st()))
    assert (out.cpu().double() - ref).abs().max().item() <= 2e-6 * ref.abs().max().item()
    # head
    a = torch.randn(n, hw, hw, 128, generator=g)
    w1 = torch.randn(128, generator=g) * 0.1
    s1, b1 = 0.7, 0.05
    f1w, f1b = torch.randn(hw * hw, 128, generator=g) * 0.05, torch.randn(128, generator=g) * 0.1
    f2w, f2b = torch.randn(128, 3, generator=g) * 0.1, torch.randn(3, generator=g) * 0.1
    mask = torch.tensor([[0, 0, 0], [1, 0, 0], [0, 1, 1], [1, 1, 1]], dtype=torch.uint8)
    h1 = ((a.double().reshape(n, hw * hw, 128) @ w1.double()) * s1 + b1).clamp_min(0)
    h2 = (h1 @ f1w.double() + f1b.double()).clamp_min(0)
    q = torch.tanh(h2 @ f2w.double() + f2b.double())
    q[mask.bool()] = -1.0
    qd = torch.full((n, 3), float("nan"), device="cuda")
    dev = [t.cuda() for t in (a, w1, f1w, f1b, f2w, f2b, mask)]
    check(L.snk_head_f32(dev[0].data_ptr(), dev[1].data_ptr(), C.c_float(s1), C.c_float(b1), dev[2].data_ptr(),
                         dev[3].data_ptr(), dev[4].data_ptr(), dev[5].data_ptr(), dev[6].data_ptr(),
                         qd.data_ptr(), n, hw, hw, _st()))
    assert (qd.cpu().double() - q).abs().max().item() <= 2e-6


def _randomised_bn(ws, seed):
    rng = np.random.RandomState(seed)
    out = []
    i = 0
    for w in ws:
        out.append(w.copy())
    # BN arrays are the 1-D float arrays that follow each conv kernel: gamma, beta, mean, var
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
    del i
    return out


@pytest.mark.parametrize("algo", ["direct", "winograd", "f16s"])
@pytest.mark.parametrize("bn_random", [False, True])
def test_full_net_matches_cpu_restatement(env, bn_random, algo, monkeypatch):
    torch, se, net = env
    monkeypatch.setenv("SNK_CONV_ALGO", algo)
    from oracle import net_ref
    s = load_golden("states_11x11x4.npz")
    states = s["raw"][:96]
    ws = net.glorot_uniform_weights((21, 21, 3), blocks=4, seed=0)
    if bn_random:
        ws = _randomised_bn(ws, 3)
    ref = net_ref.forward(ws, states)
    qn = net.QNet(ws, (21, 21, 3), max_chunk=40)     # 3 chunks, the last one ragged
    mask = torch.as_tensor(s["mask"][s["raw_index"][:96]], device="cuda")
    got = qn.forward(torch.as_tensor(states, device="cuda"), mask).cpu().numpy()
    assert np.isfinite(got).all()
    assert np.abs(got - ref).max() <= TOL_Q, np.abs(got - ref).max()
    assert (got[ref == -1.0] == -1.0).all()
    assert len(np.unique(np.round(got, 4))) > 20, "net output is degenerate"


@pytest.mark.parametrize("algo", ["winograd", "f16s"])
def test_full_net_19x19_10_blocks(env, algo, monkeypatch):
    """config 5's shape: (37,37,3) input, 10 residual blocks (a build-side extension of the same pattern)"""
    torch, se, net = env
    monkeypatch.setenv("SNK_CONV_ALGO", algo)
    from oracle import net_ref
    s = load_golden("states_19x19x8.npz")
    states = s["raw"][:6]
    ws = _randomised_bn(net.glorot_uniform_weights((37, 37, 3), blocks=10, seed=1), 4)
    ref = net_ref.forward(ws, states)
    got = net.QNet(ws, (37, 37, 3)).forward(torch.as_tensor(states, device="cuda"),
                                            torch.as_tensor(s["mask"][s["raw_index"][:6]], device="cuda")).cpu().numpy()
    assert np.abs(got - ref).max() <= TOL_Q, np.abs(got - ref).max()


def test_f16_reduced_precision_layer_and_net(env, monkeypatch):
    """configs[4]'s reduced-precision option on the f16s kernel (`SNK_CONV_ALGO=f16`): operands rounded to f16 (after the
    power-of-two scales), one MFMA per product, float32 accumulation.  Against a float64 reference computed from the SAME
    f16-rounded operands the layer is exact to float32 rounding; the whole net stays within 5e-3 of the float32 net."""
    torch, se, net = env
    from snake_engine._lib import lib, check
    from snake_engine.net import F16S_WEIGHT_BYTES
    from oracle import net_ref
    L = lib()
    g = torch.Generator().manual_seed(21)
    n, hw = 3, 21
    x = torch.randn(n, hw, hw, 128, generator=g)
    w = torch.randn(3, 3, 128, 128, generator=g) * 0.05
    sc, sh = torch.rand(128, generator=g) + 0.5, torch.randn(128, generator=g) * 0.1
    r = torch.randn(n, hw, hw, 128, generator=g)
    xs = 256.0
    ws = 2.0 ** (8 - int(torch.floor(torch.log2(w.abs().max())).item()))        # the kernel's weight scale: max|w| -> [256, 512)
    xh = (x * xs).to(torch.float16).double() / xs
    wh = (w * ws).to(torch.float16).double() / ws
    ref = torch.nn.functional.conv2d(xh.permute(0, 3, 1, 2), wh.permute(3, 2, 0, 1), padding=1).permute(0, 2, 3, 1)
    ref = (ref * sc.double() + sh.double() + r.double()).clamp_min(0)
    dev = [t.cuda() for t in (x, w, sc, sh, r)]
    wS = torch.empty(F16S_WEIGHT_BYTES, dtype=torch.uint8, device="cuda")
    out = torch.full((n, hw, hw, 128), float("nan"), device="cuda")
    check(L.snk_conv3x3_prepare_weights_f16s(dev[1].data_ptr(), wS.data_ptr(), C.c_float(xs), _st()))
    check(L.snk_conv3x3_bn_f16(dev[0].data_ptr(), wS.data_ptr(), dev[2].data_ptr(), dev[3].data_ptr(), dev[4].data_ptr(),
                               out.data_ptr(), n, hw, hw, 1, _st()))
    err = (out.cpu().double() - ref).abs().max().item()
    assert err <= 2e-5 * ref.abs().max().item(), err
    monkeypatch.setenv("SNK_CONV_ALGO", "f16")
    s = load_golden("states_11x11x4.npz")
    states = s["raw"][:48]
    wts = _randomised_bn(net.glorot_uniform_weights((21, 21, 3), blocks=4, seed=0), 3)
    got = net.QNet(wts, (21, 21, 3)).forward(torch.as_tensor(states, device="cuda")).cpu().numpy()
    ref32 = net_ref.forward(wts, states, apply_mask=False)
    assert np.isfinite(got).all()
    assert np.abs(got - ref32).max() <= 5e-3, np.abs(got - ref32).max()
    assert np.abs(got - ref32).max() > 1e-6, "the reduced-precision path did not run"


def test_bf16_conv_layer_and_net(env, monkeypatch):
    """configs[4] as BASELINE.json words it, "bf16 MFMA conv" (`SNK_CONV_ALGO=bf16`): the tower's block body (hs_block)
    instantiated for bf16 -- bf16 activations in HBM, bf16 weights, v_mfma_f32_32x32x16_bf16, float32 accumulation and epilogue.
    One layer against a float64 convolution of the SAME bf16 operands, at 21x21 and 37x37 (float32 output: float32 rounding;
    bf16 output: that result rounded once); the whole nets (11x11 / 4 blocks in chunks, 19x19 / 10 blocks; full and
    sub-rectangle forms) against the CPU restatement with the same rounding points (oracle/net_ref.py bf16_act=True)."""
    torch, se, net = env
    from snake_engine._lib import lib, check
    from snake_engine.net import F16S_WEIGHT_BYTES
    from oracle import net_ref
    L = lib()
    g = torch.Generator().manual_seed(11)
    for n, hw in ((3, 21), (2, 37), (1, 13)):
        x = torch.randn(n, hw, hw, 128, generator=g).to(torch.bfloat16)
        r = torch.randn(n, hw, hw, 128, generator=g).to(torch.bfloat16)
        w = torch.randn(3, 3, 128, 128, generator=g) * 0.05
        sc, sh = torch.rand(128, generator=g) + 0.5, torch.randn(128, generator=g) * 0.1
        wb = w.to(torch.bfloat16).double()
        ref = torch.nn.functional.conv2d(x.double().permute(0, 3, 1, 2), wb.permute(3, 2, 0, 1), padding=1).permute(0, 2, 3, 1)
        ref = (ref * sc.double() + sh.double() + r.double()).clamp_min(0)
        xd, rd, wd, scd, shd = x.cuda(), r.cuda(), w.cuda().contiguous(), sc.cuda(), sh.cuda()
        wS = torch.empty(F16S_WEIGHT_BYTES, dtype=torch.uint8, device="cuda")
        check(L.snk_conv3x3_prepare_weights_bf16(wd.data_ptr(), wS.data_ptr(), _st()))
        o32 = torch.full((n, hw, hw, 128), float("nan"), device="cuda")
        check(L.snk_conv3x3_bn_bf16_act16(xd.data_ptr(), wS.data_ptr(), scd.data_ptr(), shd.data_ptr(), rd.data_ptr(),
                                          o32.data_ptr(), 0, n, hw, hw, 1, _st()))
        scale = ref.abs().max().item()
        assert (o32.cpu().double() - ref).abs().max().item() <= 2e-5 * scale
        o16 = torch.full((n, hw, hw, 128), float("nan"), dtype=torch.bfloat16, device="cuda")
        check(L.snk_conv3x3_bn_bf16_act16(xd.data_ptr(), wS.data_ptr(), scd.data_ptr(), shd.data_ptr(), rd.data_ptr(),
                                          o16.data_ptr(), 1, n, hw, hw, 1, _st()))
        got = o16.cpu()
        assert torch.isfinite(got).all()
        want = ref.to(torch.bfloat16)         # the bf16 output is the float32 result rounded once: one bf16 ulp apart at most
        err = (got.double() - want.double()).abs()
        assert (err <= 2.0 ** -7 * want.double().abs() + 2e-5 * scale).all() and (got == want).float().mean().item() > 0.99
        # no residual, no ReLU (the generic epilogue)
        o32.fill_(float("nan"))
        check(L.snk_conv3x3_bn_bf16_act16(xd.data_ptr(), wS.data_ptr(), scd.data_ptr(), shd.data_ptr(), None,
                                          o32.data_ptr(), 0, n, hw, hw, 0, _st()))
        ref0 = torch.nn.functional.conv2d(x.double().permute(0, 3, 1, 2), wb.permute(3, 2, 0, 1), padding=1).permute(0, 2, 3, 1) * sc.double() + sh.double()
        assert (o32.cpu().double() - ref0).abs().max().item() <= 2e-5 * ref0.abs().max().item()
    monkeypatch.setenv("SNK_CONV_ALGO", "bf16")
    s = load_golden("states_11x11x4.npz")
    states = s["raw"][:128]
    wsn = _randomised_bn(net.glorot_uniform_weights((21, 21, 3), blocks=4, seed=0), 5)
    same = net_ref.forward(wsn, states, apply_mask=False, bf16_act=True)
    full = net_ref.forward(wsn, states, apply_mask=False)
    for chunk, rect in ((40, "1"), (4096, "1"), (4096, "0")):          # 40: below the sub-rectangle threshold (full form, fine blocks)
        monkeypatch.setenv("SNK_CONV_RECT", rect)
        qn = net.QNet(wsn, (21, 21, 3), max_chunk=chunk)
        assert qn.n_rect == (6 if rect == "1" else 0) and qn.act16 == torch.bfloat16
        got = qn.forward(torch.as_tensor(states, device="cuda")).cpu().numpy()
        # bf16 rounding of an activation turns a last-bit float32 difference (summation order) into a 2^-8 relative one
        assert np.abs(got - same).max() <= 1.5e-2, np.abs(got - same).max()
        assert np.abs(got - full).max() <= 5e-2, np.abs(got - full).max()
        if rect == "0":
            got_full = got
        elif chunk == 4096:
            got_rect = got
    assert np.array_equal(got_rect, got_full), "the sub-rectangle form of the bf16 tower is bit-identical to the full form"
    monkeypatch.setenv("SNK_CONV_RECT", "1")
    # the 19x19 / 10-block net of configs[4]
    s19 = load_golden("states_19x19x8.npz")
    states = s19["raw"][:6]
    ws = _randomised_bn(net.glorot_uniform_weights((37, 37, 3), blocks=10, seed=1), 4)
    mask = torch.as_tensor(s19["mask"][s19["raw_index"][:6]], device="cuda")
    got = net.QNet(ws, (37, 37, 3)).forward(torch.as_tensor(states, device="cuda"), mask).cpu().numpy()
    ref16 = net_ref.forward(ws, states, bf16_act=True)
    ref32 = net_ref.forward(ws, states)
    assert np.abs(got - ref16).max() <= 3e-2, np.abs(got - ref16).max()
    assert np.abs(got - ref32).max() <= 1e-1, np.abs(got - ref32).max()
    # the same net on all 53 recorded 19x19 observations (above the sub-rectangle threshold: 12 of the 20 layers convolve rectangles):
    # the sub-rectangle form equals the full form bit for bit at this shape too, and the head fused into the last layer's epilogue
    # (no float32 activation in HBM) equals the separate head kernel up to the float32 summation order of its 128-channel dot product
    x53 = torch.as_tensor(s19["raw"], device="cuda")
    q = {}
    for rect, fuse in (("1", "1"), ("0", "1"), ("1", "0")):
        monkeypatch.setenv("SNK_CONV_RECT", rect)
        monkeypatch.setenv("SNK_HEAD_FUSE", fuse)
        qn = net.QNet(ws, (37, 37, 3))
        assert qn.n_rect == (12 if rect == "1" else 0)
        q[rect, fuse] = qn.forward(x53).cpu().numpy()
    assert np.array_equal(q["1", "1"], q["0", "1"]), "19x19 / 10 blocks: sub-rectangle form != full form"
    assert 0 < np.abs(q["1", "1"] - q["1", "0"]).max() <= 2e-6, np.abs(q["1", "1"] - q["1", "0"]).max()
    monkeypatch.setenv("SNK_CONV_RECT", "1")
    monkeypatch.setenv("SNK_HEAD_FUSE", "1")


# ---- range guard of the split-f16 kernel ------------------------------------------------------------------------------
def _adversarial_bn(net_mod):
    """a net whose batch-norm parameters hide a huge activation from the |beta| + 8 |gamma| heuristic: the moving mean
    of the second-to-last tower layer is -3 000 (its output is ~3 000, the heuristic expects <= 8, so the last layer's
    input scale is 64 and 3 000 x 64 > 65 504), the last layer's gamma is 1e-3 so that Q stays away from tanh saturation"""
    ws = net_mod.glorot_uniform_weights((21, 21, 3), blocks=4, seed=0)
    k6, k7 = 5 + 5 * 6, 5 + 5 * 7                      # tower layers 6 and 7: kernel index; BN arrays follow
    ws[k6 + 3] = np.full(128, -3000.0, np.float32)     # moving_mean
    ws[k7 + 1] = np.full(128, 1e-3, np.float32)        # gamma
    return ws


def test_f16s_range_flag_trips_on_adversarial_batch_norm(env, monkeypatch):
    """clamped inputs are reported, not returned: QNet.forward sets the layer's range flag and check_range raises;
    AlphaNNet.v (the host path) widens the layer's scale, re-evaluates and returns float32-accurate values"""
    torch, se, net = env
    monkeypatch.setenv("SNK_CONV_ALGO", "f16s")
    from oracle import net_ref
    from snake_engine import EngineError
    from utils.alpha_nnet import AlphaNNet
    s = load_golden("states_11x11x4.npz")
    states = s["raw"][:48]
    ws = _adversarial_bn(net)
    ref = net_ref.forward(ws, states)
    assert len(np.unique(np.round(ref, 3))) > 10, "adversarial net saturates: the case says nothing"
    qn = net.QNet(ws, (21, 21, 3))
    scale7 = qn.conv_x_scale[7]
    planes = torch.as_tensor(states, device="cuda")
    clamped = qn.forward(planes).cpu().numpy()
    assert np.abs(clamped - net_ref.forward(ws, states, apply_mask=False)).max() > 1e-3      # the clamp really bites ...
    assert qn.range_flags(clear=False) == [0, 0, 0, 0, 0, 0, 0, 1]                           # ... and layer 7 says so
    with pytest.raises(EngineError, match="clamped"):
        qn.check_range()
    assert qn.conv_x_scale[7] == scale7 / 64 and qn.range_flags() == [0] * 8                  # widened, flags cleared
    got = qn.forward(planes).cpu().numpy()
    assert qn.check_range() == []
    assert np.abs(got - net_ref.forward(ws, states, apply_mask=False)).max() <= TOL_Q
    nn_ = AlphaNNet(input_shape=(21, 21, 3), _weights=ws)                                     # fresh scales: v() recovers by itself
    assert np.abs(nn_.v(list(states)) - ref).max() <= TOL_Q
    assert nn_._qnet.range_flags() == [0] * 8
    # calibrate() gets there without tripping anything
    qn2 = net.QNet(ws, (21, 21, 3))
    rep = qn2.calibrate(planes)
    assert rep[7][2] < 1.0 and qn2.conv_x_scale[7] < scale7 and qn2.calibrated
    assert np.abs(qn2.forward(planes).cpu().numpy() - net_ref.forward(ws, states, apply_mask=False)).max() <= TOL_Q
    assert qn2.check_range() == []


def test_f16s_stays_exact_on_weights_that_went_through_fit(env, monkeypatch):
    """~30 optimizer steps of trainer_torch.fit on self-play samples move the batch-norm statistics away from the Keras
    defaults; the split-f16 net with its heuristic scales still matches the float32 CPU restatement to 1e-5 on fresh
    observations and no range flag trips"""
    torch, se, net = env
    monkeypatch.setenv("SNK_CONV_ALGO", "f16s")
    from oracle import net_ref
    from utils import trainer_torch
    s = load_golden("states_11x11x4.npz")
    X = s["raw"][:160]
    rng = np.random.RandomState(5)
    Y = np.tanh(rng.randn(160, 3)).astype(np.float32)
    ws0 = net.glorot_uniform_weights((21, 21, 3), blocks=4, seed=2)
    boundaries, values = [20, 40, 60, 80, 100], [1e-3 * 0.25 ** i for i in range(5)] + [0.0]
    ws = trainer_torch.fit(ws0, (21, 21, 3), X, Y, epochs=10, batch_size=54, lr_schedule=(boundaries, values), seed=1, verbose=False)   # 30 steps
    moved = max(float(np.abs(a - b).max()) for a, b in zip(ws, ws0))
    assert moved > 1e-3
    fresh = s["raw"][160:224]
    ref = net_ref.forward(ws, fresh, apply_mask=False)
    qn = net.QNet(ws, (21, 21, 3))
    got = qn.forward(torch.as_tensor(fresh, device="cuda")).cpu().numpy()
    assert qn.check_range() == []
    assert np.abs(got - ref).max() <= TOL_Q, np.abs(got - ref).max()
    rep = qn.activation_report(torch.as_tensor(fresh, device="cuda"))
    assert min(h for _, _, h in rep) > 8.0, rep          # at least 8x headroom to the f16 limit on every layer


def test_agent_calibrates_new_weights_and_checks_the_range_every_turn(env, monkeypatch):
    """Agent.make_moves: first search after set_weights fits the scales to the root observations; a net that would
    clamp (adversarial batch-norm) therefore self-plays without tripping the guard"""
    torch, se, net = env
    monkeypatch.setenv("SNK_CONV_ALGO", "f16s")
    from utils.agent import Agent
    from utils.alpha_nnet import AlphaNNet
    from utils.mp_game_runner import MPGameRunner
    MPGameRunner.verbose = False
    nn_ = AlphaNNet(input_shape=(21, 21, 3), _weights=_adversarial_bn(net))
    assert not nn_._qnet.calibrated
    alice = Agent(nn_, 2, True, 4, 8, seed=3)
    gr = MPGameRunner(11, 11, 4, 1, 8, seed=5)
    gr.run(alice, max_turns=2)
    assert nn_._qnet.calibrated and nn_._qnet.range_flags() == [0] * 8 and gr.env_steps == 16


@pytest.mark.parametrize("algo", ["f16s", "f16"])
def test_a_clamped_batch_is_evaluated_again_inside_the_search(env, monkeypatch, algo):
    """(f16: the reduced-precision form with scaled float32 inputs clamps in the same staging code and is guarded the same way)
    scales that are far too large for the activations (set after the calibration, as a weight update would leave them): the
    first batch of the next root turn clamps, AlphaNNet.v_device (QNet.forward_guarded) widens the layer's scale and evaluates
    THAT BATCH again before anything reaches the transposition table -- the search runs once, the table keeps the statistics of
    the earlier turns (the reference keeps them until Agent.clear, agent.py:140-147), and records, values and counts equal
    those of an undisturbed twin run"""
    torch, se, net = env
    monkeypatch.setenv("SNK_CONV_ALGO", algo)
    from utils.agent import Agent
    from utils.alpha_nnet import AlphaNNet
    from utils.mp_game_runner import MPGameRunner
    MPGameRunner.verbose = False
    ws = net.glorot_uniform_weights((21, 21, 3), blocks=4, seed=2)

    def play(disturb):
        import random
        random.seed(11)                     # MPGameRunner draws the start boards with python's `random` (game.py:25-30, 46)
        nn_ = AlphaNNet(input_shape=(21, 21, 3), _weights=ws)
        alice = Agent(nn_, 2, True, 4, 8, seed=3)
        gr = MPGameRunner(11, 11, 4, 1, 8, seed=5)
        gr.run(alice, max_turns=1)                                      # calibrates
        good = list(nn_._qnet.conv_x_scale)
        entries1 = len(alice.cached_values)
        if disturb:
            nn_._qnet.set_x_scale(3, good[3] * 2.0 ** 12)               # far beyond the f16 range for this layer's inputs
        searches = []
        orig = alice._mcts.search
        alice._mcts.search = lambda *a, **k: (searches.append(1), orig(*a, **k))[1]
        gr.run(alice, max_turns=2)
        assert len(searches) == 2 and gr.env_steps == 16, "one search per root turn, clamp or not"
        assert nn_._qnet.range_flags() == [0] * 8
        return nn_, alice, good, entries1

    nn_a, a, good, e1a = play(False)
    nn_b, b, _, e1b = play(True)
    assert nn_a._qnet.guard_trips == 0 and nn_b._qnet.guard_trips >= 1
    assert a._mcts.guard_redos == 0 and b._mcts.guard_redos == nn_b._qnet.guard_trips     # every trip: that tick's evaluation + tail run again
    assert good[3] <= nn_b._qnet.conv_x_scale[3] <= good[3] * 64.0 and nn_b._qnet.conv_x_scale[:3] == good[:3]
    assert e1a == e1b > 0 and len(a.cached_values) == len(b.cached_values) > e1a, "nothing was forgotten, both tables grew alike"
    assert a._mcts.stats == b._mcts.stats, (a._mcts.stats, b._mcts.stats)          # evaluations, rollout tics, ticks: no double count
    assert a._mcts.draw_ctr == b._mcts.draw_ctr and a._mcts.now == b._mcts.now == 3
    assert len(a.records) == len(b.records) and all(x.tobytes() == y.tobytes() for x, y in zip(a.records[:], b.records[:]))
    va, vb = np.array(a.values[:]), np.array(b.values[:])
    assert np.abs(va - vb).max() <= TOL_Q, np.abs(va - vb).max()      # (a power-of-two scale does not move an f16 rounding point)
    # a net without a range to watch is "calibrated" at once: no observation pass per root turn
    monkeypatch.setenv("SNK_CONV_ALGO", "winograd")
    nw = AlphaNNet(input_shape=(21, 21, 3), _weights=ws)
    assert not nw._qnet.calibrated
    nw.calibrate(torch.zeros((1, 21, 21, 3), device="cuda"))
    assert nw._qnet.calibrated


def test_gated_ticks_replay_a_clamped_evaluation_wherever_it_happens(env, monkeypatch):
    """the search does not wait for the guard: the kernels of a rollout tick that follow the leaf evaluation are gated on the
    Q-net's device guard word, the host reads the word's mirror at the next tick's read-back (or at the epoch's end) and runs
    that tick's evaluation and tail again.  Activation scales are pushed out of range right before chosen forward calls --
    the first of a root turn, early and middle ones, the very last of a turn's last epoch -- in searches with several ticks
    per epoch (boards with 2-4 snakes, depth 8, two epochs): records, values, evaluation counts, draws and the table equal
    those of an undisturbed twin run"""
    torch, se, net = env
    monkeypatch.setenv("SNK_CONV_ALGO", "f16s")
    import random
    from utils.agent import Agent
    from utils.alpha_nnet import AlphaNNet
    from utils.mp_game_runner import MPGameRunner
    MPGameRunner.verbose = False
    ws = net.glorot_uniform_weights((21, 21, 3), blocks=4, seed=2)

    def play(trip_calls):
        random.seed(12)
        nn_ = AlphaNNet(input_shape=(21, 21, 3), _weights=ws)
        alice = Agent(nn_, 2, True, 8, 16, seed=3)
        gr = MPGameRunner(11, 11, 4, 1, 24, seed=5)
        gr.run(alice, max_turns=8)                                      # calibrates; some snakes die: deeper rollouts
        qn, calls, orig = nn_._qnet, [0], nn_._qnet.forward
        # how far every tower layer's input is from the f16 limit on mid-game observations: a disturbance is sized from it, so that
        # it CLAMPS on any batch whose largest activation is within 8x of these (scaled maximum 8x beyond the limit) and FITS
        # again after the guard's one widening by 2^-6 (8x below the limit): every scheduled site bites, and bites once
        eng_planes = torch.as_tensor(np.stack([alice.records[i] for i in range(0, len(alice.records), 7)][:96]), device="cuda")
        amax = [r[0] for r in qn.activation_report(eng_planes)]
        qn.range_flags()
        fired = []

        def fwd(planes, mask=None, out=None):
            calls[0] += 1
            if calls[0] in trip_calls:
                layer = calls[0] % 8
                headroom = 65504.0 / (amax[layer] * qn.conv_x_scale[layer])
                factor = 2.0 ** (int(np.ceil(np.log2(headroom))) + 3)
                qn.set_x_scale(layer, qn.conv_x_scale[layer] * factor)
                fired.append((calls[0], layer, factor, qn.guard_trips))
            return orig(planes, mask, out)
        qn.forward = fwd
        ticks0 = alice._mcts.stats["rollout_ticks"]
        gr.run(alice, max_turns=3)
        return nn_, alice, gr, calls[0], alice._mcts.stats["rollout_ticks"] - ticks0, fired

    nn_a, a, gr_a, n_calls, n_ticks, _ = play(())
    assert n_ticks > 3 * 2 * 2, "several ticks per epoch"
    # the first call of a root turn, the next one (the repeat of the first: a repeat that clamps again), an early one, two in the
    # middle, the last but one, and one past the undisturbed run's last call (+ 5: the twin's repeats are calls too)
    trips = {1, 2, 5, n_calls // 3, n_calls // 2, n_calls - 2, n_calls + 5}
    nn_b, b, gr_b, n_calls_b, _, fired = play(trips)
    assert len(fired) == len(trips) == 7, f"not every scheduled call happened: {[f[0] for f in fired]} of {sorted(trips)}"
    after = [f[3] for f in fired[1:]] + [nn_b._qnet.guard_trips]
    dead = [(call, layer, factor) for (call, layer, factor, before), aft in zip(fired, after) if aft == before]
    assert not dead, f"scheduled disturbances that did not clamp (call, layer, factor): {dead}"
    assert nn_a._qnet.guard_trips == 0 and b._mcts.guard_redos == nn_b._qnet.guard_trips == len(fired), (
        b._mcts.guard_redos, nn_b._qnet.guard_trips, fired)
    assert n_calls_b == n_calls + b._mcts.guard_redos
    assert nn_b._qnet.range_flags() == [0] * 8 and not nn_b._qnet.guard_tripped()
    assert a._mcts.stats == b._mcts.stats, (a._mcts.stats, b._mcts.stats)
    assert a._mcts.draw_ctr == b._mcts.draw_ctr and len(a.cached_values) == len(b.cached_values)
    assert gr_a.env_steps == gr_b.env_steps and len(a.records) == len(b.records)
    assert all(x.tobytes() == y.tobytes() for x, y in zip(a.records[:], b.records[:]))
    va, vb = np.array(a.values[:]), np.array(b.values[:])
    assert np.abs(va - vb).max() <= TOL_Q, np.abs(va - vb).max()


def test_f16_activation_tower_reports_a_saturated_activation(env, monkeypatch):
    """`SNK_CONV_ALGO=f16a` keeps activations as f16 in HBM and saturates an output beyond 65 504 instead of writing an infinity: that
    is a gross error, not f16 rounding, so the tower layer sets its range flag and the net's guard word (as the split form's clamp
    does) -- callers that can wait get an EngineError from forward_guarded / AlphaNNet.v, the search's gate sees the word -- and the
    bf16 tower, which has float32's range, evaluates the same weights without complaint"""
    torch, se, net = env
    from snake_engine import EngineError
    s = load_golden("states_11x11x4.npz")
    x = torch.as_tensor(s["raw"][:64], device="cuda")
    ws = _adversarial_bn(net)
    ws[5 + 5 * 6 + 3] = np.full(128, -2.0e5, np.float32)          # tower layer 6's moving mean: its output is ~2e5, beyond the f16 range
    monkeypatch.setenv("SNK_CONV_ALGO", "f16a")
    qn = net.QNet(ws, (21, 21, 3))
    assert qn.guard_ptr != 0
    q = qn.forward(x).cpu().numpy()                                # the unguarded forward: saturated, finite ...
    assert np.isfinite(q).all() and qn.range_flags(clear=False)[6] == 1      # ... and flagged on the layer that saturated
    with pytest.raises(EngineError, match="saturated"):
        qn.forward_guarded(x)
    assert qn.range_flags() == [0] * 8 and not qn.guard_tripped()  # the raise leaves no flag behind
    good = net.QNet(net.glorot_uniform_weights((21, 21, 3), blocks=4, seed=0), (21, 21, 3))
    good.forward_guarded(x)
    assert good.guard_trips == 0 and good.range_flags() == [0] * 8
    monkeypatch.setenv("SNK_CONV_ALGO", "bf16")
    qb = net.QNet(ws, (21, 21, 3))
    assert qb.guard_ptr == 0 and np.isfinite(qb.forward_guarded(x).cpu().numpy()).all()


def test_f16_activation_tower_layer_and_net(env, monkeypatch):
    """configs[4]'s fastest reduced-precision option (`SNK_CONV_ALGO=f16a`): f16 operands AND f16 activations in HBM.
    One layer against a float64 convolution of the same f16 inputs (f16 output: within half an f16 ulp + float32 rounding;
    float32 output: float32 rounding); the whole net against the CPU restatement with the same rounding points
    (oracle/net_ref.py f16_act=True) to 2e-3, and within 5e-3 of the float32 net."""
    torch, se, net = env
    from snake_engine._lib import lib, check
    from snake_engine.net import F16S_WEIGHT_BYTES
    from oracle import net_ref
    L = lib()
    g = torch.Generator().manual_seed(33)
    for n, hw in ((3, 21), (2, 37)):
        x = (torch.randn(n, hw, hw, 128, generator=g)).to(torch.float16)
        r = (torch.randn(n, hw, hw, 128, generator=g)).to(torch.float16)
        w = torch.randn(3, 3, 128, 128, generator=g) * 0.05
        sc, sh = torch.rand(128, generator=g) + 0.5, torch.randn(128, generator=g) * 0.1
        ws_ = 2.0 ** (8 - int(torch.floor(torch.log2(w.abs().max())).item()))
        wh = (w * ws_).to(torch.float16).double() / ws_
        ref = torch.nn.functional.conv2d(x.double().permute(0, 3, 1, 2), wh.permute(3, 2, 0, 1), padding=1).permute(0, 2, 3, 1)
        ref = (ref * sc.double() + sh.double() + r.double()).clamp_min(0)
        xd, rd, wd, scd, shd = x.cuda(), r.cuda(), w.cuda().contiguous(), sc.cuda(), sh.cuda()
        wS = torch.empty(F16S_WEIGHT_BYTES, dtype=torch.uint8, device="cuda")
        check(L.snk_conv3x3_prepare_weights_f16_act16(wd.data_ptr(), wS.data_ptr(), _st()))
        o32 = torch.full((n, hw, hw, 128), float("nan"), device="cuda")
        check(L.snk_conv3x3_bn_f16_act16(xd.data_ptr(), wS.data_ptr(), scd.data_ptr(), shd.data_ptr(), rd.data_ptr(),
                                         o32.data_ptr(), 0, n, hw, hw, 1, _st()))
        scale = ref.abs().max().item()
        assert (o32.cpu().double() - ref).abs().max().item() <= 2e-5 * scale
        o16 = torch.full((n, hw, hw, 128), float("nan"), dtype=torch.float16, device="cuda")
        check(L.snk_conv3x3_bn_f16_act16(xd.data_ptr(), wS.data_ptr(), scd.data_ptr(), shd.data_ptr(), rd.data_ptr(),
                                         o16.data_ptr(), 1, n, hw, hw, 1, _st()))
        got = o16.cpu()
        assert torch.isfinite(got).all()
        # the f16 output is the float32 result rounded once: equal to rounding the float64 reference except where the two
        # straddle a rounding boundary (then one f16 ulp apart)
        want = ref.to(torch.float16)
        err = (got.double() - want.double()).abs()
        assert (err <= 2.0 ** -10 * want.double().abs() + 2e-5 * scale).all() and (got == want).float().mean().item() > 0.99
    monkeypatch.setenv("SNK_CONV_ALGO", "f16a")
    s = load_golden("states_11x11x4.npz")
    states = s["raw"][:64]
    wsn = _randomised_bn(net.glorot_uniform_weights((21, 21, 3), blocks=4, seed=0), 5)
    qn = net.QNet(wsn, (21, 21, 3), max_chunk=40)
    got = qn.forward(torch.as_tensor(states, device="cuda")).cpu().numpy()
    same = net_ref.forward(wsn, states, apply_mask=False, f16_act=True)
    full = net_ref.forward(wsn, states, apply_mask=False)
    # f16 rounding of an activation turns a last-bit float32 difference (summation order) into a 2^-11 relative one, so the
    # two agree only to the level of the rounding itself
    assert np.abs(got - same).max() <= 2e-3, np.abs(got - same).max()
    assert np.abs(got - full).max() <= 5e-3, np.abs(got - full).max()
    s19 = load_golden("states_19x19x8.npz")
    ws19 = _randomised_bn(net.glorot_uniform_weights((37, 37, 3), blocks=10, seed=1), 4)
    got19 = net.QNet(ws19, (37, 37, 3)).forward(torch.as_tensor(s19["raw"][:6], device="cuda")).cpu().numpy()
    assert np.abs(got19 - net_ref.forward(ws19, s19["raw"][:6], apply_mask=False, f16_act=True)).max() <= 5e-3


@pytest.mark.parametrize("form", ["mfma", "valu"])
def test_stem_shapes_and_forms(env, form):
    """the stem (3 -> 128, 3x3, BN, ReLU; alpha_nnet.py:21-22) against a float64 convolution on every observation size a
    square board gives (9x9 ... 37x37), rectangular and ragged pixel counts, batches around the persistent grid size --
    for the MFMA form (default) and the packed-FMA form; the f16-output variant against the rounded float32 result.
    The form is chosen when the library first launches a stem, so each form runs in its own process."""
    import subprocess
    import sys
    from conftest import REPO
    code = r'''
import ctypes as C, sys, os
sys.path[:0] = [%r, os.path.join(%r, "alphasnake-zero_amd")]
import torch
from snake_engine._lib import lib, check
L = lib(); st = torch.cuda.current_stream().cuda_stream
g = torch.Generator().manual_seed(5)
for n, H, W in ((3, 9, 9), (2, 13, 13), (5, 21, 21), (1, 29, 29), (2, 37, 37), (4, 7, 30), (1, 3, 3), (600, 21, 21), (1030, 9, 9)):
    x = torch.rand(n, H, W, 3, generator=g) * 6 - 1          # the range observations live in: [-1, 5]
    w = torch.randn(3, 3, 3, 128, generator=g) * 0.2
    sc, sh = torch.rand(128, generator=g) + 0.5, torch.randn(128, generator=g) * 0.1
    ref = torch.nn.functional.conv2d(x.double().permute(0, 3, 1, 2), w.double().permute(3, 2, 0, 1), padding=1).permute(0, 2, 3, 1)
    ref = (ref * sc.double() + sh.double()).clamp_min(0)
    d = [t.cuda() for t in (x, w, sc, sh)]
    out = torch.full((n, H, W, 128), float("nan"), device="cuda")
    check(L.snk_stem_conv_bn_relu_f32(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), out.data_ptr(), n, H, W, st))
    err = (out.cpu().double() - ref).abs().max().item()
    assert err <= 3e-6 * ref.abs().max().item(), (n, H, W, err)
    o16 = torch.full((n, H, W, 128), float("nan"), dtype=torch.float16, device="cuda")
    check(L.snk_stem_conv_bn_relu_f16out(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), o16.data_ptr(), n, H, W, st))
    e16 = (o16.cpu().double() - ref).abs()
    assert (e16 <= 2.0 ** -10 * ref.abs() + 1e-5).all(), (n, H, W, e16.max().item())
print("ok")
''' % (REPO, REPO)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, SNK_STEM=form), timeout=600)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stderr[-3000:]
