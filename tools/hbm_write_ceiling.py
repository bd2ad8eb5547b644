"""what a pure streaming write reaches on this GPU (the stem kernel's ceiling): fill and copy of 1.7 GB"""
import torch
x = torch.empty(7483 * 441 * 128, dtype=torch.float32, device="cuda")
y = torch.empty_like(x)
for name, fn, byts in (("fill (write only)", lambda: x.fill_(1.0), x.numel() * 4), ("copy (read + write)", lambda: y.copy_(x), 2 * x.numel() * 4)):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(20):
        fn()
    b.record(); torch.cuda.synchronize()
    t = a.elapsed_time(b) / 20 * 1e-3
    print(f"{name}: {t * 1e6:.0f} us, {byts / t / 1e12:.2f} TB/s")
