"""times the Winograd conv kernel selected by SNK_WINO_WAVES and checks it against torch conv2d"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "alphasnake-zero_amd")]
import torch
from snake_engine._lib import lib, check
L = lib(); st = torch.cuda.current_stream().cuda_stream
for n in (64, 512, 4096):
    x = torch.randn(n, 21, 21, 128, device="cuda"); o = torch.empty_like(x)
    w = torch.randn(3, 3, 128, 128, device="cuda") * 0.05; U = torch.empty(16 * 128 * 128, device="cuda")
    sc = torch.rand(128, device="cuda") + 0.5; sh = torch.randn(128, device="cuda")
    check(L.snk_conv3x3_prepare_weights_winograd(w.data_ptr(), U.data_ptr(), st))
    f = lambda: check(L.snk_conv3x3_bn_f32_winograd(x.data_ptr(), U.data_ptr(), sc.data_ptr(), sh.data_ptr(), x.data_ptr(), o.data_ptr(), n, 21, 21, 1, st))
    for _ in range(3): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): f()
    b.record(); torch.cuda.synchronize()
    t = a.elapsed_time(b) / 20 * 1e-3
    if n <= 512:
        ref = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double(), w.permute(3, 2, 0, 1).double(), padding=1).permute(0, 2, 3, 1)
        ref = torch.relu(ref * sc.double() + sh.double() + x.double())
        err = (o.double() - ref).abs().max().item()
    else:
        err = float("nan")
    print(f"winograd n={n}: {t*1e3:.3f} ms  {2*n*441*1152*128/t/1e12:.1f} TF-equiv  max|err|={err:.2e}", flush=True)
