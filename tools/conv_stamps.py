"""phase timing of k_conv3x3_f16s from s_memtime stamps (needs the -DHS_STAMPS development build libsnake_conv_dbg.so):
   hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -Iinclude -ffp-contract=off -DHS_STAMPS -shared \
         -o alphasnake-zero_amd/snake_engine/libsnake_conv_dbg.so alphasnake-zero_amd/csrc/conv_split.hip alphasnake-zero_amd/csrc/engine.hip"""
import ctypes as C, os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
L = C.CDLL(os.path.join(REPO, "alphasnake-zero_amd", "snake_engine", "libsnake_conv_dbg.so"))
vp = C.c_void_p
L.snk_conv3x3_prepare_weights_f16s.argtypes = [vp, vp, C.c_float, vp]
L.snk_conv3x3_bn_f16s.argtypes = [vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp]
L.snk_dbg_conv_stamps.argtypes = [vp, C.c_int]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
H = 21
st = torch.cuda.current_stream().cuda_stream
zeros = len(sys.argv) > 2 and sys.argv[2] == "zeros"
x = torch.randn(n, H, H, 128, device="cuda") * (0.0 if zeros else 1.0); o = torch.empty_like(x)
w = torch.randn(3, 3, 128, 128, device="cuda") * (0.0 if zeros else 0.05); U = torch.empty(9 * 128 * 128 * 4 + 32, dtype=torch.uint8, device="cuda")
sc = torch.ones(128, device="cuda"); sh = torch.zeros(128, device="cuda")
assert L.snk_conv3x3_prepare_weights_f16s(w.data_ptr(), U.data_ptr(), 256.0, st) == 0
f = lambda: L.snk_conv3x3_bn_f16s(x.data_ptr(), U.data_ptr(), sc.data_ptr(), sh.data_ptr(), x.data_ptr(), o.data_ptr(), n, H, H, 1, st)
import time
t_end = time.time() + 2.0                     # >= 2 s of back-to-back launches so the clock settles (MI355X_MICROARCH.md, DVFS item 6)
while time.time() < t_end:
    for _ in range(50): assert f() == 0
    torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record(); f(); b.record(); torch.cuda.synchronize()
ms = a.elapsed_time(b)
nb = min(2 * n, 16384)
buf = np.zeros((nb, 8), np.uint64)
assert L.snk_dbg_conv_stamps(buf.ctypes.data, nb) == 0
t = buf[:, :5].astype(np.int64)
span = t[:, 4].max() - t[:, 0].min()
print(f"n={n}: launch {ms:.3f} ms; first start -> last end {span} ticks = {span / ms / 1e3:.1f} ticks/us")
names = ["prologue (zero LDS, stage chunk 0)", "all chunks but the last (with staging)", "last chunk", "epilogue"]
d = np.diff(t, axis=1)
for k, nm in enumerate(names):
    print(f"  {nm:38s} mean {d[:, k].mean():9.0f}  p10 {np.percentile(d[:, k], 10):9.0f}  p90 {np.percentile(d[:, k], 90):9.0f} ticks")
e = buf.astype(np.int64)
clk = (e[:, 3] - e[:, 1]) / np.maximum(1, e[:, 6] - e[:, 5]) * 100.0
print(f"  in-kernel clock over the chunk loop (s_memtime / s_memrealtime x 100 MHz): median {np.median(clk):.0f} MHz  ({'zero' if zeros else 'random'} operands)")
tot = (t[:, 4] - t[:, 0])
print(f"  block total mean {tot.mean():.0f} ticks; sum of block times / (256 CUs x span) = {tot.sum() / 256 / span:.3f}")
