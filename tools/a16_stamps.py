"""phase timing of the conv blocks of the 16-bit tower from s_memtime stamps (stamps build:
make -C alphasnake-zero_amd/csrc variant NAME=dbg EXTRA=-DHS_STAMPS; SNK_LIB_PATH=.../libsnake_engine_dbg.so):
    a16_stamps.py [images 1024] [side 37] [bf16|f16a]       one full-form layer (16-bit in, 16-bit out, shortcut) on random data"""
import ctypes as C, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "alphasnake-zero_amd")]
import numpy as np
import torch
from snake_engine._lib import lib, check

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
H = int(sys.argv[2]) if len(sys.argv) > 2 else 37
algo = sys.argv[3] if len(sys.argv) > 3 else "bf16"
L = lib()
L.snk_dbg_conv_stamps.argtypes = [C.c_void_p, C.c_int]
st = torch.cuda.current_stream().cuda_stream
dt = torch.bfloat16 if algo == "bf16" else torch.float16
x = torch.randn(n, H, H, 128, device="cuda").relu().to(dt)
r = torch.randn(n, H, H, 128, device="cuda").relu().to(dt)
o = torch.empty_like(x)
w = torch.randn(3, 3, 128, 128, device="cuda") * 0.03
U = torch.empty(9 * 128 * 128 * 4 + 32, dtype=torch.uint8, device="cuda")
sc = torch.ones(128, device="cuda"); sh = torch.zeros(128, device="cuda")
if algo == "bf16":
    check(L.snk_conv3x3_prepare_weights_bf16(w.data_ptr(), U.data_ptr(), st))
    f = lambda: L.snk_conv3x3_bn_bf16_act16(x.data_ptr(), U.data_ptr(), sc.data_ptr(), sh.data_ptr(), r.data_ptr(), o.data_ptr(), 1, n, H, H, 1, st)
else:
    check(L.snk_conv3x3_prepare_weights_f16_act16(w.data_ptr(), U.data_ptr(), st))
    f = lambda: L.snk_conv3x3_bn_f16_act16(x.data_ptr(), U.data_ptr(), sc.data_ptr(), sh.data_ptr(), r.data_ptr(), o.data_ptr(), 1, n, H, H, 1, st)
t_end = time.time() + 1.5
while time.time() < t_end:
    for _ in range(20):
        check(f())
    torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record(); check(f()); b.record(); torch.cuda.synchronize()
ms = a.elapsed_time(b)
nb = 16384
buf = np.zeros((nb, 8), np.uint64)
assert L.snk_dbg_conv_stamps(buf.ctypes.data, nb) == 0
buf = buf[buf[:, 4] > 0]
t = buf[:, :5].astype(np.int64)
span = t[:, 4].max() - t[:, 0].min()
flops = 2.0 * n * H * H * 9 * 128 * 128
print(f"{algo} {n} x {H} x {H}: launch {ms:.3f} ms = {flops / ms / 1e9:.0f} TFLOP/s; {len(t)} blocks stamped")
names = ["prologue (zero LDS, stage chunk 0)", "all chunks but the last (with staging)", "last chunk", "epilogue"]
d = np.diff(t, axis=1)
for k, nm in enumerate(names):
    print(f"  {nm:38s} mean {d[:, k].mean():9.0f}  p10 {np.percentile(d[:, k], 10):9.0f}  p90 {np.percentile(d[:, k], 90):9.0f} cycles")
e = buf.astype(np.int64)
clk = (e[:, 3] - e[:, 1]) / np.maximum(1, e[:, 6] - e[:, 5]) * 100.0
print(f"  in-kernel clock over the chunk loop: median {np.median(clk):.0f} MHz; block total mean {(t[:, 4] - t[:, 0]).mean():.0f} cycles")
hw = buf[:, 7]
if hw.any():
    # which blocks shared a CU, and how their lives overlapped: HW_ID bits: wave slot [3:0], SIMD [5:4], CU [11:8], SH [12], SE [15:13], workgroup slot [19:16]
    cu = ((hw >> 32) & 0xF) * 4096 + ((hw >> 13) & 7) * 512 + ((hw >> 12) & 1) * 256 + ((hw >> 8) & 0xF) * 16
    tg = (hw >> 16) & 0xF
    wv = hw & 0xF
    print(f"  distinct CUs {len(set(cu.tolist()))}; workgroup slots seen {sorted(set(tg.tolist()))}; wave slots of wave 0 {sorted(set(wv.tolist()))}")
    # phase of each block's chunk-loop start relative to the block that was running on the same CU at that moment
    order = np.argsort(t[:, 0])
    last = {}
    offs = []
    for i in order:
        c = int(cu[i])
        if c in last:
            j = last[c]
            if t[j, 4] > t[i, 0]:                      # j still alive when i started
                offs.append((t[i, 0] - t[j, 0]) / max(1, t[j, 4] - t[j, 0]))
        last[c] = i
    offs = np.array(offs)
    if len(offs):
        hist = np.histogram(offs, bins=10, range=(0, 1))[0]
        print(f"  start of a block relative to the life of the block it joined on its CU (0 = together, 0.5 = half a life later): deciles {hist.tolist()}")
if hasattr(L, "snk_dbg_conv_tap_stamps") and os.environ.get("A16_TAPS", "1") != "0":
    # wave 0's clock at the start of every tap of every chunk (and at the chunk's end, before its barrier): where a chunk's time goes
    L.snk_dbg_conv_tap_stamps.argtypes = [C.c_void_p]
    tb = np.zeros((2048, 8, 10), np.uint64)
    assert L.snk_dbg_conv_tap_stamps(tb.ctypes.data) == 0
    tb = tb.astype(np.int64)
    nchunk = int((tb[0, :, 0] > 0).sum())
    ok = tb[:, 0, 0] > 0
    d = np.diff(tb[ok][:, :nchunk, :], axis=2)                      # [block][chunk][tap 0..8 duration]
    print(f"  per tap of wave 0 (mean cycles over {int(ok.sum())} blocks; a tap = NI tiles x {2 if nchunk == 4 else 3} MFMAs x 32 cycles of matrix pipe when alone):")
    for c_ in range(nchunk):
        print(f"    chunk {c_}: " + " ".join(f"{v:6.0f}" for v in d[:, c_, :].mean(axis=0)) + f"   sum {d[:, c_, :].sum(axis=1).mean():7.0f}")
