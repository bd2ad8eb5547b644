"""Round-4 experiment: the tower layer as a ONE-DIMENSIONAL Winograd F(2, 3) on the split-f16 operands (conv_wino.hip here; not
part of the library).  Compiles it into gpurun_out/, checks it against a float64 convolution on ten shapes x four epilogues x two
magnitudes next to the library's direct kernel, times both at 21 x 21, and prints the phase stamps of its blocks:
    python tools/experiments/wino1d/run.py            (on the GPU box)"""
import ctypes as C, math, os, subprocess, sys, time
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(os.path.dirname(HERE)))
sys.path[:0] = [REPO, os.path.join(REPO, "alphasnake-zero_amd")]
import torch
from snake_engine._lib import lib, check
so = os.path.join(REPO, "gpurun_out", "libwino1d.so")
os.makedirs(os.path.dirname(so), exist_ok=True)
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-I" + os.path.join(REPO, "include"), "-I" + HERE,
                       "-ffp-contract=off", "-DHW_STAMPS", "-shared", "-o", so, os.path.join(HERE, "conv_wino.hip")])
C.CDLL(os.path.join(REPO, "alphasnake-zero_amd", "snake_engine", "libsnake_engine.so"), mode=C.RTLD_GLOBAL)      # snk_set_error
W = C.CDLL(so)
vp = C.c_void_p
W.snk_conv3x3_prepare_weights_f16sw.argtypes = [vp, vp, C.c_float, vp]
W.snk_conv3x3_bn_f16sw.argtypes = [vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp]
W.snk_dbg_wino_stamps.argtypes = [vp, C.c_int]
L = lib(); st = torch.cuda.current_stream().cuda_stream
torch.manual_seed(1)
worst = [0.0, 0.0]
for (n, H, Wd) in ((3, 21, 21), (2, 13, 13), (5, 9, 9), (2, 37, 37), (3, 7, 30), (2, 20, 20), (1, 5, 3), (9, 21, 21), (1, 3, 4), (2, 33, 17)):
    for relu, use_res in ((1, False), (1, True), (0, True), (0, False)):
        for mag in (1.0, 1e-3):
            x = torch.randn(n, H, Wd, 128, device="cuda") * mag
            x = torch.relu(x) if relu else x
            r = torch.randn(n, H, Wd, 128, device="cuda") * mag
            w = torch.randn(3, 3, 128, 128, device="cuda") * 0.05
            sc = torch.rand(128, device="cuda") + 0.5; sh = torch.randn(128, device="cuda") * mag
            ref = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double(), w.permute(3, 2, 0, 1).double(), padding=1).permute(0, 2, 3, 1)
            ref = ref * sc.double() + sh.double() + (r.double() if use_res else 0)
            ref = torch.relu(ref) if relu else ref
            xsc = 2.0 ** (11 - math.ceil(math.log2(8 * mag)))
            for k, (prep, conv, nb) in enumerate(((W.snk_conv3x3_prepare_weights_f16sw, W.snk_conv3x3_bn_f16sw, 12 * 128 * 128 * 4 + 32),
                                                  (L.snk_conv3x3_prepare_weights_f16s, L.snk_conv3x3_bn_f16s, 9 * 128 * 128 * 4 + 32))):
                U = torch.empty(nb, dtype=torch.uint8, device="cuda")
                assert prep(w.data_ptr(), U.data_ptr(), xsc, st) == 0
                o = torch.full_like(x, float("nan"))
                assert conv(x.data_ptr(), U.data_ptr(), sc.data_ptr(), sh.data_ptr(), r.data_ptr() if use_res else None, o.data_ptr(), n, H, Wd, relu, st) == 0
                torch.cuda.synchronize()
                assert torch.isfinite(o).all()
                worst[k] = max(worst[k], (o.double() - ref).abs().max().item() / mag)
print(f"80 cases: worst max|err| / |x|: winograd {worst[0]:.2e}, direct {worst[1]:.2e}")
assert worst[0] < 2e-5
for n in (256, 2048, 8192):
    x = torch.relu(torch.randn(n, 21, 21, 128, device="cuda")); o = torch.empty_like(x)
    w = torch.randn(3, 3, 128, 128, device="cuda") * 0.05
    sc = torch.rand(128, device="cuda") + 0.5; sh = torch.randn(128, device="cuda")
    for name, prep, conv, nb in (("winograd F(2,3) x split-f16", W.snk_conv3x3_prepare_weights_f16sw, W.snk_conv3x3_bn_f16sw, 12 * 128 * 128 * 4 + 32),
                                 ("direct split-f16 (library)", L.snk_conv3x3_prepare_weights_f16s, L.snk_conv3x3_bn_f16s, 9 * 128 * 128 * 4 + 32)):
        U = torch.empty(nb, dtype=torch.uint8, device="cuda")
        assert prep(w.data_ptr(), U.data_ptr(), 256.0, st) == 0
        f = lambda: conv(x.data_ptr(), U.data_ptr(), sc.data_ptr(), sh.data_ptr(), x.data_ptr(), o.data_ptr(), n, 21, 21, 1, st)
        for _ in range(5): f()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20): f()
        b.record(); torch.cuda.synchronize()
        t = a.elapsed_time(b) / 20 * 1e-3
        print(f"{name:30s} 21x21 n={n}: {t*1e3:.3f} ms  {2*n*441*1152*128/t/1e12:.1f} TFLOP/s algorithmic", flush=True)
nb = 3 * 2048
buf = np.zeros((nb, 8), np.uint64)
n = 2048
x = torch.relu(torch.randn(n, 21, 21, 128, device="cuda")); o = torch.empty_like(x)
U = torch.empty(12 * 128 * 128 * 4 + 32, dtype=torch.uint8, device="cuda")
assert W.snk_conv3x3_prepare_weights_f16sw(w.data_ptr(), U.data_ptr(), 256.0, st) == 0
t_end = time.time() + 1.0
while time.time() < t_end:
    for _ in range(50): W.snk_conv3x3_bn_f16sw(x.data_ptr(), U.data_ptr(), sc.data_ptr(), sh.data_ptr(), None, o.data_ptr(), n, 21, 21, 1, st)
    torch.cuda.synchronize()
assert W.snk_dbg_wino_stamps(buf.ctypes.data, nb) == 0
t = buf[:, :5].astype(np.int64); d = np.diff(t, axis=1)
three = d[(d[:, 3] > 0)]
for k, nm in enumerate(["prologue (zero LDS, stage chunk 0)", "seven chunks with staging", "last chunk (no staging)", "epilogue"]):
    print(f"  {nm:36s} mean {three[:, k].mean():8.0f} cycles")
e = buf.astype(np.int64)
print(f"  clock over the chunk loop {np.median((e[:, 3] - e[:, 1]) / np.maximum(1, e[:, 6] - e[:, 5]) * 100.0):.0f} MHz; MFMA floor of a 3-tile block: 8 x 54 x 32 x 2 waves per SIMD = 27 648 cycles")
