"""Known-traffic kernels for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 (MI355X_MICROARCH.md, HBM section:
FETCH_SIZE reports half the bytes of 16-B/lane streaming reads; other widths must be calibrated on a known byte count).

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d OUT -- python3 tools/pmc_calib.py

1 GiB in, 1 GiB out each (well past the 256 MiB Infinity Cache):
  * y = 2*x on aligned views    -> at::native::vectorized_elementwise_kernel      (16 B per lane)
  * y = 2*x on views shifted by 1 and 3 elements -> at::native::elementwise_kernel_manual_unroll (4 B per lane, the access
    width of conv_taps_kernel's activation loads and of lpips_partial_kernel)
  * hipMemcpy D2D (__amd_rocclr_copyBuffer) as a third reference
"""
import torch

N = 1 << 28     # floats = 1 GiB
x = torch.empty(N + 8, dtype=torch.float32, device="cuda").normal_()
y = torch.empty(N + 8, dtype=torch.float32, device="cuda")
torch.cuda.synchronize()
for _ in range(3):
    torch.mul(x[:N], 2.0, out=y[:N])
torch.cuda.synchronize()
for _ in range(3):
    torch.mul(x[3:N + 3], 2.0, out=y[1:N + 1])
torch.cuda.synchronize()
for _ in range(3):
    y[:N].copy_(x[:N])
torch.cuda.synchronize()
print("calib done", float(y[5]))
