import os, sys, subprocess
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import os, sys
sys.path[:0] = [%r, %r]
import torch
from snake_engine._lib import lib, check
L = lib(); st = torch.cuda.current_stream().cuda_stream
n = 4096
x = torch.randn(n, 21, 21, 128, device="cuda"); o = torch.empty_like(x)
w = torch.randn(3, 3, 128, 128, device="cuda") * 0.05; U = torch.empty(16 * 128 * 128, device="cuda")
sc = torch.ones(128, device="cuda"); sh = torch.zeros(128, device="cuda")
check(L.snk_conv3x3_prepare_weights_winograd(w.data_ptr(), U.data_ptr(), st))
def run(): check(L.snk_conv3x3_bn_f32_winograd(x.data_ptr(), U.data_ptr(), sc.data_ptr(), sh.data_ptr(), x.data_ptr(), o.data_ptr(), n, 21, 21, 1, st))
for _ in range(3): run()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(10): run()
b.record(); torch.cuda.synchronize()
print("dbg", os.environ.get("SNK_WINO_DBG", "0"), "ms", a.elapsed_time(b) / 10)
''' % (REPO, os.path.join(REPO, "alphasnake-zero_amd"))
for dbg in (0, 1, 2, 4, 6, 8, 16, 7, 15, 31, 23):
    env = dict(os.environ, SNK_WINO_DBG=str(dbg))
    print(subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True).stdout.strip(), flush=True)
