"""runs only one conv3x3 kernel variant (for rocprofv3 counter passes): conv_only.py [n] [direct|winograd|f16s]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "alphasnake-zero_amd")]
import torch
from snake_engine._lib import lib, check
L = lib(); st = torch.cuda.current_stream().cuda_stream
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
algo = sys.argv[2] if len(sys.argv) > 2 else "winograd"
x = torch.randn(n, 21, 21, 128, device="cuda"); o = torch.empty_like(x)
w = torch.randn(3, 3, 128, 128, device="cuda") * 0.05; wT = torch.empty(16 * 128 * 128 + 4, device="cuda")
sc = torch.ones(128, device="cuda"); sh = torch.zeros(128, device="cuda")
prep, conv = ((L.snk_conv3x3_prepare_weights_winograd, L.snk_conv3x3_bn_f32_winograd) if algo == "winograd"
              else (L.snk_conv3x3_prepare_weights, L.snk_conv3x3_bn_f32))
if algo == "f16s":
    prep, conv = L.snk_conv3x3_prepare_weights_f16s, L.snk_conv3x3_bn_f16s
    check(prep(w.data_ptr(), wT.data_ptr(), 256.0, st))
else:
    check(prep(w.data_ptr(), wT.data_ptr(), st))
for _ in range(5):
    check(conv(x.data_ptr(), wT.data_ptr(), sc.data_ptr(), sh.data_ptr(), x.data_ptr(), o.data_ptr(), n, 21, 21, 1, st))
torch.cuda.synchronize()
