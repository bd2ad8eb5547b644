"""differential fuzz of the weight-gradient kernel (window form, csrc/train_wgrad.hip k_wgrad2_f16s; SNK_WGRAD=slabs: the slab form)
against a float64 einsum on random square shapes, batch sizes and magnitudes (development aid): fuzz_wgrad.py [seed 0] [trials 40]
Widths 21 and 37 take the compile-time-width bodies, every other width the run-time-width ones."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "alphasnake-zero_amd")]
import numpy as np, torch
from snake_engine._lib import lib, check
from snake_engine.net import F16S_TAIL_OFFSET, F16S_WEIGHT_BYTES
L, st = lib(), torch.cuda.current_stream().cuda_stream
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
trials = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rs = np.random.RandomState(seed)


def tail_of(x):
    image = torch.empty(F16S_WEIGHT_BYTES, dtype=torch.uint8, device="cuda")
    part = torch.empty(L.snk_bn_train_partials(), device="cuda")
    check(L.snk_conv3x3_f16s_input_scale(x.data_ptr(), x.numel(), image.data_ptr(), part.data_ptr(), st))
    return image[F16S_TAIL_OFFSET:F16S_TAIL_OFFSET + 16].view(torch.float32).clone()


worst, n_def = 0.0, 0
for t in range(trials):
    hw = int(rs.choice([5, 7, 9, 13, 17, 21, 21, 25, 29, 33, 37, 37, 41, 45]))
    if L.snk_conv3x3_wgrad_partials(hw, hw) < 0:
        continue
    n = int(rs.choice([1, 2, 3, 7, 16, 33, 100, 129, 260]))
    n = max(1, min(n, 40000 // (hw * hw)))
    xm, gm = 10.0 ** rs.uniform(-2, 2), 10.0 ** rs.uniform(-5, 0)
    g = torch.Generator(device="cuda").manual_seed(seed * 1000 + t)
    x = torch.relu(torch.randn(n, hw, hw, 128, device="cuda", generator=g)) * xm
    dy = torch.randn(n, hw, hw, 128, device="cuda", generator=g) * gm
    tx, tdy = tail_of(x), tail_of(dy)
    part = torch.full((L.snk_conv3x3_wgrad_partials(hw, hw),), float("nan"), device="cuda")
    dk = torch.full((3, 3, 128, 128), float("nan"), device="cuda")
    check(L.snk_conv3x3_wgrad_f16s(x.data_ptr(), dy.data_ptr(), tx.data_ptr(), tdy.data_ptr(), part.data_ptr(), dk.data_ptr(), n, hw, hw, st))
    x64 = torch.nn.functional.pad(x.double(), (0, 0, 1, 1, 1, 1))
    ref = torch.stack([torch.stack([torch.einsum("nhwc,nhwd->cd", x64[:, a:a + hw, b:b + hw], dy.double()) for b in range(3)]) for a in range(3)])
    err = float((dk.double() - ref).abs().max() / ref.abs().max())
    worst = max(worst, err)
    assert torch.isfinite(dk).all() and err <= 2e-6, (t, n, hw, xm, gm, err)
    if L.snk_train_deferred_bn_supported(hw, hw) == 1:      # the deferred form: X = relu(y * scale + shift) applied on the way into LDS, bit for bit
        ypre = torch.randn(n, hw, hw, 128, device="cuda", generator=g) * xm
        sc = (torch.rand(128, device="cuda", generator=g) + 0.25) * torch.where(torch.rand(128, device="cuda", generator=g) < 0.2, -1.0, 1.0)
        sh = torch.randn(128, device="cuda", generator=g) * xm * 0.3
        xd = torch.relu(ypre * sc + sh).contiguous()
        txd = tail_of(xd)
        dk_a, dk_b = torch.full_like(dk, float("nan")), torch.full_like(dk, float("nan"))
        check(L.snk_conv3x3_wgrad_f16s(xd.data_ptr(), dy.data_ptr(), txd.data_ptr(), tdy.data_ptr(), part.data_ptr(), dk_a.data_ptr(), n, hw, hw, st))
        check(L.snk_conv3x3_wgrad_f16s_deferred(ypre.data_ptr(), sc.data_ptr(), sh.data_ptr(), dy.data_ptr(), txd.data_ptr(), tdy.data_ptr(),
                                                part.data_ptr(), dk_b.data_ptr(), n, hw, hw, st))
        assert torch.equal(dk_a, dk_b), (t, n, hw, "deferred")
        n_def += 1
print(f"fuzz_wgrad seed {seed}: {trials} shapes (widths 5 .. 45, 1 .. 260 images, |x| 1e-2 .. 1e2, |dy| 1e-5 .. 1) within 2e-6 of float64, worst {worst:.1e}; {n_def} of them also in the deferred form, bit-identical to the written activation"
      f"{' (slab form)' if os.environ.get('SNK_WGRAD') == 'slabs' else ''}")
