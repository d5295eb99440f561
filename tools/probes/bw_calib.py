"""Developer probe: what HBM bandwidth does this box reach for plain torch streaming kernels?"""
import torch, time
x = torch.empty(1 << 30, dtype=torch.uint8, device="cuda").view(torch.float32)  # 1 GiB
y = torch.empty_like(x)
for name, fn, nbytes in (("copy (r+w)", lambda: y.copy_(x), 2 * x.numel() * 4), ("sum (read)", lambda: x.sum(), x.numel() * 4), ("fill (write)", lambda: y.fill_(1.0), x.numel() * 4)):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"{name:14s} {nbytes / ms / 1e6:8.1f} GB/s")
