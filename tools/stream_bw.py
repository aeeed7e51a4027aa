"""Measured device copy bandwidth (the denominator check for the HBM roofline)."""
import torch
n = 1 << 30                      # 1 GiB source, 1 GiB destination
a = torch.empty(n, dtype=torch.uint8, device="cuda")
b = torch.empty(n, dtype=torch.uint8, device="cuda")
a.fill_(1)
for _ in range(3):
    b.copy_(a)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(20):
    b.copy_(a)
e.record()
torch.cuda.synchronize()
ms = s.elapsed_time(e) / 20
print("device copy 1 GiB -> 1 GiB: %.3f ms, %.0f GB/s read+write (%.0f GB/s one-way)" % (ms, 2 * n / ms / 1e6, n / ms / 1e6))
