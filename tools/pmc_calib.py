"""Known-traffic kernels for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 (MI355X_MICROARCH.md, HBM section:
FETCH_SIZE reports half the bytes of 16-B/lane streaming reads; other widths must be calibrated on a known byte count).

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d OUT -- python3 tools/pmc_calib.py

Two copies of 1 GiB each (well past the 256 MiB Infinity Cache): an aligned one (torch's vectorised 16-B/lane copy) and a
copy between views shifted by one element (4-B/lane accesses).  tools/pmc_traffic.py reads the CSV and prints the factors.
"""
import torch

N = 1 << 28     # floats = 1 GiB
x = torch.empty(N + 8, dtype=torch.float32, device="cuda").normal_()
y = torch.empty(N + 8, dtype=torch.float32, device="cuda")
torch.cuda.synchronize()
for _ in range(3):
    y[:N].copy_(x[:N])                  # aligned: vectorized_elementwise_kernel, 16 B per lane
torch.cuda.synchronize()
for _ in range(3):
    y[1:N + 1].copy_(x[3:N + 3])        # misaligned both sides: unrolled scalar path, 4 B per lane
torch.cuda.synchronize()
print("calib done", float(y[5]))
