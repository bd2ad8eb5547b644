"""Quick kernel timing on the GPU box (development aid; bench.py is the judged harness)."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "alphasnake-zero_amd")]
import numpy as np, torch
import snake_engine as se
from snake_engine import net
from snake_engine._lib import lib, check


def timeit(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e-3


def main():
    L = lib(); st = torch.cuda.current_stream().cuda_stream
    for n in (512, 4096):
        x = torch.randn(n, 21, 21, 128, device="cuda"); o = torch.empty_like(x)
        w = torch.randn(3, 3, 128, 128, device="cuda") * 0.05; wT = torch.empty(9 * 128 * 128, device="cuda")
        sc = torch.ones(128, device="cuda"); sh = torch.zeros(128, device="cuda")
        check(L.snk_conv3x3_prepare_weights(w.data_ptr(), wT.data_ptr(), st))
        t = timeit(lambda: check(L.snk_conv3x3_bn_f32(x.data_ptr(), wT.data_ptr(), sc.data_ptr(), sh.data_ptr(), x.data_ptr(), o.data_ptr(), n, 21, 21, 1, st)))
        fl = 2 * n * 441 * 1152 * 128
        print(f"conv3x3 f32 n={n}: {t*1e3:.3f} ms  {fl/t/1e12:.1f} TFLOP/s")
    for n in (512, 4096):
        x = torch.randn(n, 21, 21, 128, device="cuda"); o = torch.empty_like(x)
        w = torch.randn(3, 3, 128, 128, device="cuda") * 0.05; U = torch.empty(16 * 128 * 128, device="cuda")
        sc = torch.ones(128, device="cuda"); sh = torch.zeros(128, device="cuda")
        check(L.snk_conv3x3_prepare_weights_winograd(w.data_ptr(), U.data_ptr(), st))
        t = timeit(lambda: check(L.snk_conv3x3_bn_f32_winograd(x.data_ptr(), U.data_ptr(), sc.data_ptr(), sh.data_ptr(), x.data_ptr(), o.data_ptr(), n, 21, 21, 1, st)))
        fl = 2 * n * 441 * 1152 * 128
        print(f"conv3x3 winograd f32 n={n}: {t*1e3:.3f} ms  {fl/t/1e12:.1f} TFLOP/s (direct-equivalent)")
    for n in (8192,):
        xs = torch.randn(n, 21, 21, 3, device="cuda"); o = torch.empty(n, 21, 21, 128, device="cuda")
        w3 = torch.randn(3, 3, 3, 128, device="cuda"); sc = torch.ones(128, device="cuda"); sh = torch.zeros(128, device="cuda")
        t = timeit(lambda: check(L.snk_stem_conv_bn_relu_f32(xs.data_ptr(), w3.data_ptr(), sc.data_ptr(), sh.data_ptr(), o.data_ptr(), n, 21, 21, st)))
        print(f"stem conv n={n}: {t*1e3:.3f} ms  {n*441*128*4/t/1e9:.0f} GB/s written")
    ws = net.glorot_uniform_weights((21, 21, 3))
    qn = net.QNet(ws, (21, 21, 3), max_chunk=4096)
    for n in (16, 128, 4096, 16384):
        p = torch.randn(n, 21, 21, 3, device="cuda")
        t = timeit(lambda: qn.forward(p), iters=3, warm=1)
        print(f"net forward n={n}: {t*1e3:.1f} ms  {n/t:.0f} states/s  {qn.flops_per_state()*n/t/1e12:.1f} TFLOP/s")
    for n in (32768, 262144):
        eng = se.Engine(n, 11, 11, 4, 1, 0.15); eng.reset()
        g = torch.Generator(device="cuda").manual_seed(0)
        for _ in range(20):
            eng.step(torch.randint(0, 3, (n, 4), device="cuda", generator=g, dtype=torch.int32).to(torch.uint8))
        mv = torch.ones((n, 4), dtype=torch.uint8, device="cuda")
        snap = se.Engine(n, 11, 11, 4, 1, 0.15)
        eng.clone_to(snap)
        def stepfn():
            eng.step(mv)
        t = timeit(stepfn, iters=5, warm=1)
        print(f"step n={n}: {t*1e6:.1f} us  {n/t/1e6:.1f} Msteps/s  {n*2*eng.slot_bytes/t/1e9:.0f} GB/s algorithmic")
        t = timeit(lambda: snap.clone_to(eng), iters=10)
        print(f"clone n={n}: {t*1e6:.1f} us  {n*2*eng.slot_bytes/t/1e9:.0f} GB/s")
        alive = eng.alive(); pairs = torch.nonzero(alive).to(torch.int32).contiguous(); m = pairs.shape[0]
        planes = torch.empty((m, 21, 21, 3), device="cuda"); mask = torch.empty((m, 3), dtype=torch.uint8, device="cuda"); key = torch.empty((m, 2), dtype=torch.int64, device="cuda")
        t = timeit(lambda: eng.observe(pairs, m, planes, mask, key), iters=5)
        print(f"observe(planes+mask+key) m={m}: {t*1e6:.1f} us  {(m*5292 + m*eng.slot_bytes)/t/1e9:.0f} GB/s")
        t = timeit(lambda: eng.observe(pairs, m, None, mask, key), iters=5)
        print(f"observe(mask+key) m={m}: {t*1e6:.1f} us  {m/t/1e6:.1f} Mobs/s")
        t = timeit(lambda: eng.observe(pairs, m, planes, None, None), iters=5)
        print(f"observe(planes only) m={m}: {t*1e6:.1f} us  {(m*5292 + m*eng.slot_bytes)/t/1e9:.0f} GB/s")
        del eng, snap, planes


if __name__ == "__main__":
    main()
