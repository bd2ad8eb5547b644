"""block timing of k_wgrad2_f16s from s_memtime stamps (make -C alphasnake-zero_amd/csrc variant NAME=wgdbg EXTRA=-DWG_STAMPS;
SNK_LIB_PATH=.../libsnake_engine_wgdbg.so):  wgrad_stamps.py [n 2048] [side 21]"""
import ctypes as C, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "alphasnake-zero_amd")]
import numpy as np, torch
from snake_engine._lib import lib, check
from snake_engine.net import F16S_TAIL_OFFSET, F16S_WEIGHT_BYTES
L, st = lib(), torch.cuda.current_stream().cuda_stream
L.snk_dbg_wgrad_stamps.argtypes = [C.c_void_p]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
hw = int(sys.argv[2]) if len(sys.argv) > 2 else 21


def tail_of(x):
    image = torch.empty(F16S_WEIGHT_BYTES, dtype=torch.uint8, device="cuda")
    part = torch.empty(L.snk_bn_train_partials(), device="cuda")
    check(L.snk_conv3x3_f16s_input_scale(x.data_ptr(), x.numel(), image.data_ptr(), part.data_ptr(), st))
    return image[F16S_TAIL_OFFSET:F16S_TAIL_OFFSET + 16].view(torch.float32).clone()


x = torch.relu(torch.randn(n, hw, hw, 128, device="cuda")); dy = torch.randn(n, hw, hw, 128, device="cuda") * 1e-3
tx, tdy = tail_of(x), tail_of(dy)
part = torch.empty(L.snk_conv3x3_wgrad_partials(hw, hw), device="cuda"); dk = torch.empty(3, 3, 128, 128, device="cuda")
run = lambda: check(L.snk_conv3x3_wgrad_f16s(x.data_ptr(), dy.data_ptr(), tx.data_ptr(), tdy.data_ptr(), part.data_ptr(), dk.data_ptr(), n, hw, hw, st))
t_end = time.time() + 1.5
while time.time() < t_end:
    for _ in range(20):
        run()
    torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record(); run(); b.record(); torch.cuda.synchronize()
buf = np.zeros((512, 8), np.uint64)
assert L.snk_dbg_wgrad_stamps(buf.ctypes.data) == 0
buf = buf[buf[:, 2] > 0].astype(np.int64)
tot = buf[:, 2] - buf[:, 0]; pro = buf[:, 1] - buf[:, 0]
clk = tot / np.maximum(1, buf[:, 5] - buf[:, 4]) * 100.0
T = buf[:, 6]
print(f"{n} x {hw} x {hw}: launch + fold {a.elapsed_time(b):.3f} ms; {len(buf)} blocks; block life mean {tot.mean():.0f} cycles (p10 {np.percentile(tot, 10):.0f}, p90 {np.percentile(tot, 90):.0f}), "
      f"prologue {pro.mean():.0f}; windows per block {T.mean():.1f} -> {(tot - pro).mean() / T.mean():.0f} cycles per window; clock {np.median(clk):.0f} MHz; "
      f"first start -> last end {(buf[:, 2].max() - buf[:, 0].min())} cycles")
