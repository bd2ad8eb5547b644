"""k_wgrad_f16s alone (csrc/train_wgrad.hip): time per call and error against float64 at the training step's shape
(2 048 images of 21 x 21) and at BASELINE configs[4]'s (37 x 37).  Development tool: wgrad_time.py [n_images]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "alphasnake-zero_amd")]
import torch
from snake_engine._lib import lib, check
from snake_engine.net import F16S_TAIL_OFFSET, F16S_WEIGHT_BYTES
L, st = lib(), torch.cuda.current_stream().cuda_stream


def tail_of(x):
    image = torch.empty(F16S_WEIGHT_BYTES, dtype=torch.uint8, device="cuda")
    part = torch.empty(L.snk_bn_train_partials(), device="cuda")
    check(L.snk_conv3x3_f16s_input_scale(x.data_ptr(), x.numel(), image.data_ptr(), part.data_ptr(), st))
    return image[F16S_TAIL_OFFSET:F16S_TAIL_OFFSET + 16].view(torch.float32).clone()


for hw, n in ((21, int(sys.argv[1]) if len(sys.argv) > 1 else 2048), (37, 640)):
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.relu(torch.randn(n, hw, hw, 128, device="cuda", generator=g))
    dy = torch.randn(n, hw, hw, 128, device="cuda", generator=g) * 1e-3
    tx, tdy = tail_of(x), tail_of(dy)
    part = torch.empty(L.snk_conv3x3_wgrad_partials(hw, hw), device="cuda")
    dk = torch.empty(3, 3, 128, 128, device="cuda")
    run = lambda: check(L.snk_conv3x3_wgrad_f16s(x.data_ptr(), dy.data_ptr(), tx.data_ptr(), tdy.data_ptr(), part.data_ptr(), dk.data_ptr(), n, hw, hw, st))
    for _ in range(20):
        run()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(50):
        run()
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 50
    m = min(n, 64)                                    # float64 check on a slice of the batch
    check(L.snk_conv3x3_wgrad_f16s(x.data_ptr(), dy.data_ptr(), tx.data_ptr(), tdy.data_ptr(), part.data_ptr(), dk.data_ptr(), m, hw, hw, st))
    x64 = torch.nn.functional.pad(x[:m].double(), (0, 0, 1, 1, 1, 1))
    ref = torch.stack([torch.stack([torch.einsum("nhwc,nhwd->cd", x64[:, a_:a_ + hw, b_:b_ + hw], dy[:m].double()) for b_ in range(3)]) for a_ in range(3)])
    err = float((dk.double() - ref).abs().max() / ref.abs().max())
    fl = 2.0 * n * hw * hw * 9 * 128 * 128
    print(f"{n} x {hw} x {hw}: {ms:.3f} ms per call (fold included) = {fl / ms / 1e9:.0f} TFLOP/s algorithmic = {fl / ms / 1e9 / 2500:.3f} of the f16 peak; "
          f"error vs float64 {err:.1e}")
