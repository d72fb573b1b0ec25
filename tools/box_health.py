"""One line about the box a profile was taken on: device-to-device copy bandwidth and sustained fp32-MFMA-free clock proxy.
(Boxes of the pool differ: bandwidth-bound kernels up to 50 % apart, see profiles/README.md.)"""
import time
import torch

a = torch.empty(1 << 28, dtype=torch.float32, device="cuda")  # 1 GiB
b = torch.empty_like(a)
for _ in range(3):
    b.copy_(a)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 10
for _ in range(n):
    b.copy_(a)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
x = torch.randn(8192, 8192, device="cuda")
for _ in range(2):
    y = x @ x
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    y = x @ x
torch.cuda.synchronize()
dg = (time.perf_counter() - t0) / 5
print(f"box health: 1 GiB device copy {2 * a.numel() * 4 / dt / 1e12:.2f} TB/s (read + write); "
      f"rocBLAS fp32 8192^3 GEMM {2 * 8192**3 / dg / 1e12:.1f} TFLOP/s; device {torch.cuda.get_device_name(0)}")
