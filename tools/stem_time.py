"""Times snk_stem_conv_bn_relu_f32 (HIP events) and reports its output bandwidth.  Development tool."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "alphasnake-zero_amd")]
import torch
from snake_engine._lib import lib, check

L = lib()
for n, hw in ((7483, 21), (8192, 21), (512, 21), (1024, 37)):
    x = torch.rand(n, hw, hw, 3, device="cuda")
    w = torch.randn(3, 3, 3, 128, device="cuda") * 0.1
    sc, sh = torch.rand(128, device="cuda") + 0.5, torch.randn(128, device="cuda") * 0.1
    out = torch.empty(n, hw, hw, 128, device="cuda")
    st = torch.cuda.current_stream().cuda_stream

    def run():
        check(L.snk_stem_conv_bn_relu_f32(x.data_ptr(), w.data_ptr(), sc.data_ptr(), sh.data_ptr(), out.data_ptr(), n, hw, hw, st))
    for _ in range(3):
        run()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(20):
        run()
    b.record()
    torch.cuda.synchronize()
    t = a.elapsed_time(b) / 20 * 1e-3
    byts = n * hw * hw * (128 + 3) * 4
    print(f"stem {n} x {hw}x{hw}: {t * 1e6:.1f} us, {byts / t / 1e12:.2f} TB/s (read + write)", flush=True)
