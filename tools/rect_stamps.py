"""phase timing of the conv blocks from s_memtime stamps, per M-tile count, full form against sub-rectangle form on the same
mid-game observations (needs the stamps build:  make -C alphasnake-zero_amd/csrc variant NAME=dbg EXTRA=-DHS_STAMPS  and
SNK_LIB_PATH=alphasnake-zero_amd/snake_engine/libsnake_engine_dbg.so):   rect_stamps.py [layer 2] [games 2300]"""
import ctypes as C, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "alphasnake-zero_amd")]
import numpy as np
import torch
import snake_engine as se
from snake_engine import net
from snake_engine._lib import lib, check

layer = int(sys.argv[1]) if len(sys.argv) > 1 else 2
games = int(sys.argv[2]) if len(sys.argv) > 2 else 2300
L = lib()
L.snk_dbg_conv_stamps.argtypes = [C.c_void_p, C.c_int]
eng = se.Engine(games, 11, 11, 4, 1, 0.15, seed=7)
eng.reset()
g = torch.Generator(device="cuda").manual_seed(7)
for _ in range(14):
    pairs = torch.nonzero(eng.alive()).to(torch.int32).contiguous()
    _, mask, _ = eng.observe_all(pairs, want_planes=False, want_key=False)
    pick = torch.multinomial((mask == 0).to(torch.float32) + 1e-3, 1, generator=g).squeeze(1).to(torch.uint8)
    mv = torch.ones((games, 4), dtype=torch.uint8, device="cuda")
    mv[pairs[:, 0].long(), pairs[:, 1].long()] = pick
    eng.step(mv)
pairs = torch.nonzero(eng.alive()).to(torch.int32).contiguous()
planes, _, _ = eng.observe_all(pairs)
m = planes.shape[0]
qn = net.QNet(net.glorot_uniform_weights((21, 21, 3), blocks=4, seed=0), (21, 21, 3), max_chunk=1 << 20)
st = torch.cuda.current_stream().cuda_stream
bufs = [torch.empty((m, 21, 21, 128), device="cuda") for _ in range(3)]


def tower_to(layer, plan):
    """runs the tower up to and including `layer`; returns the event time of that layer's launch"""
    check(qn.L.snk_stem_conv_bn_relu_f32(planes.data_ptr(), qn.stem_w.data_ptr(), qn.stem_sc.data_ptr(), qn.stem_sh.data_ptr(),
                                         bufs[0].data_ptr(), m, 21, 21, st))
    cur, t1, t2 = bufs
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for i in range(layer + 1):
        if i == layer:
            a.record()
        if i % 2 == 0:
            qn._conv(i, cur, None, t1, m, st, plan=plan)
        else:
            qn._conv(i, t1, cur, t2, m, st, plan=plan)
            cur, t2 = t2, cur
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b)


def report(name, ms, ntile):
    nb = min(len(ntile), 16384)
    buf = np.zeros((nb, 8), np.uint64)
    assert L.snk_dbg_conv_stamps(buf.ctypes.data, nb) == 0
    t = buf[:, :5].astype(np.int64)
    d = np.diff(t, axis=1)
    span = t[:, 4].max() - t[:, 0].min()
    print(f"{name}: layer {layer}, {m} observations, {len(ntile)} blocks ({nb} stamped), launch {ms:.3f} ms, span {span} ticks = {span / ms / 1e3:.0f} ticks/us; "
          f"sum of block ticks / (256 CUs x span) = {(t[:, 4] - t[:, 0]).sum() / 256 / span:.3f}")
    print("   tiles  blocks   prologue  chunks 0-6   per tile  last chunk   epilogue(+fill)   total   total per tile")
    for k in sorted(set(ntile[:nb].tolist())):
        s = ntile[:nb] == k
        if s.sum() < 8:
            continue
        dd = d[s].mean(axis=0)
        tot = dd.sum()
        print(f"   {k:5d} {int(s.sum()):7d} {dd[0]:10.0f} {dd[1]:11.0f} {(dd[1] + dd[2]) / k:10.0f} {dd[2]:11.0f} {dd[3]:14.0f} {tot:10.0f} {tot / k:10.0f}")


for _ in range(3):
    tower_to(layer, None)
ms = tower_to(layer, None)
report("full form", ms, np.full(2 * m, 7))
plan = qn._rect_plan(planes, m, 0, st)
for _ in range(3):
    tower_to(layer, plan)
ms = tower_to(layer, plan)
desc, counts = plan[0].cpu().numpy().view(np.uint32), plan[1].cpu().numpy()
nd = int(counts[layer, 0])
report("sub-rectangle form", ms, ((desc[layer, :nd, 2] >> 8) & 255).astype(np.int64))
