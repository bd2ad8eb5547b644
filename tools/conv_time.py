"""times one 3x3 tower-layer kernel and checks it against a float64 torch conv2d:
   conv_time.py [winograd|direct|f16s|f16|f16a|bf16] [H] [batch sizes ...]      (f16a / bf16: 16-bit activations in HBM, timing only)"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "alphasnake-zero_amd")]
import torch
from snake_engine._lib import lib, check
L = lib(); st = torch.cuda.current_stream().cuda_stream
algo = sys.argv[1] if len(sys.argv) > 1 else "winograd"
H = int(sys.argv[2]) if len(sys.argv) > 2 else 21
sizes = [int(v) for v in sys.argv[3:]] or [64, 512, 4096]
if algo in ("f16a", "bf16"):      # f16 / bf16 activations in HBM: time only (tests/test_net_gpu.py checks the values)
    bf = algo == "bf16"
    for n in sizes:
        x = torch.randn(n, H, H, 128, device="cuda").to(torch.bfloat16 if bf else torch.float16); o = torch.empty_like(x)
        w = torch.randn(3, 3, 128, 128, device="cuda") * 0.05; U = torch.empty(9 * 128 * 128 * 4 + 32, dtype=torch.uint8, device="cuda")
        sc = torch.rand(128, device="cuda") + 0.5; sh = torch.randn(128, device="cuda")
        if bf:
            check(L.snk_conv3x3_prepare_weights_bf16(w.data_ptr(), U.data_ptr(), st))
        else:
            check(L.snk_conv3x3_prepare_weights_f16_act16(w.data_ptr(), U.data_ptr(), st))
        f = lambda: check((L.snk_conv3x3_bn_bf16_act16 if bf else L.snk_conv3x3_bn_f16_act16)(x.data_ptr(), U.data_ptr(), sc.data_ptr(), sh.data_ptr(), x.data_ptr(), o.data_ptr(), 1, n, H, H, 1, st))
        for _ in range(3): f()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20): f()
        b.record(); torch.cuda.synchronize()
        t = a.elapsed_time(b) / 20 * 1e-3
        print(f"{algo} {H}x{H} n={n}: {t*1e3:.3f} ms  {2*n*H*H*1152*128/t/1e12:.1f} TF-equiv", flush=True)
    sys.exit(0)
prep, conv, wbytes = {"winograd": (L.snk_conv3x3_prepare_weights_winograd, L.snk_conv3x3_bn_f32_winograd, 16 * 128 * 128 * 4),
                      "direct": (L.snk_conv3x3_prepare_weights, L.snk_conv3x3_bn_f32, 9 * 128 * 128 * 4),
                      "f16s": (L.snk_conv3x3_prepare_weights_f16s, L.snk_conv3x3_bn_f16s, 9 * 128 * 128 * 4 + 32),
                      "f16": (L.snk_conv3x3_prepare_weights_f16s, L.snk_conv3x3_bn_f16, 9 * 128 * 128 * 4 + 32)}[algo]
torch.manual_seed(0)
for n in sizes:
    for mag in (1.0, 1e-3) if n <= 64 else (1.0,):
        x = torch.randn(n, H, H, 128, device="cuda") * mag; o = torch.full_like(x, float("nan"))
        w = torch.randn(3, 3, 128, 128, device="cuda") * 0.05; U = torch.empty(wbytes, dtype=torch.uint8, device="cuda")
        sc = torch.rand(128, device="cuda") + 0.5; sh = torch.randn(128, device="cuda") * mag
        if algo in ("f16s", "f16"):      # the activation scale the net wrapper would pick: inputs up to ~8 |x| -> 2^11
            import math
            check(prep(w.data_ptr(), U.data_ptr(), 2.0 ** (11 - math.ceil(math.log2(8 * mag))), st))
        else:
            check(prep(w.data_ptr(), U.data_ptr(), st))
        f = lambda: check(conv(x.data_ptr(), U.data_ptr(), sc.data_ptr(), sh.data_ptr(), x.data_ptr(), o.data_ptr(), n, H, H, 1, st))
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
            err = (o.double() - ref).abs().max().item() / mag
        else:
            err = float("nan")
        print(f"{algo} {H}x{H} n={n} |x|~{mag:g}: {t*1e3:.3f} ms  {2*n*H*H*1152*128/t/1e12:.1f} TF-equiv  max|err|/|x|={err:.2e}", flush=True)
