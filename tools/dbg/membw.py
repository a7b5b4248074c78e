"""HBM ceilings of this box by torch kernels: fill (write-only), copy (1:1), read-only (sum)."""
import torch
n = 1 << 28  # 1 GiB of float32
x = torch.empty(n, dtype=torch.float32, device='cuda'); y = torch.empty_like(x)
def t(f, reps=20):
    for _ in range(3): f()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e-3
b = n * 4
print(f'fill  {b / t(lambda: x.fill_(1.0)) / 1e12:.2f} TB/s written')
print(f'copy  {2 * b / t(lambda: y.copy_(x)) / 1e12:.2f} TB/s read+written')
print(f'sum   {b / t(lambda: x.sum()) / 1e12:.2f} TB/s read')
x3 = torch.empty(3 * n // 4, dtype=torch.float32, device='cuda')
print(f'cat 1 read : 3 written  {4 * (n // 4) * 4 / t(lambda: torch.cat([x[:n // 4]] * 3, out=x3)) / 1e12:.2f} TB/s read+written')
