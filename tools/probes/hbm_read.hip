// Read-only / write-only / copy stream rates of hand-written kernels (16 bytes per lane, grid-stride), for the roofline of the
// read-heavy passes (pools, tap distances, the data-gradient inputs):
//   hipcc --offload-arch=gfx950 -O3 -o hbm_read tools/probes/hbm_read.hip && ./hbm_read
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ __launch_bounds__(256) void read_kernel(const float4* __restrict__ x, float* __restrict__ out, size_t n4, int unroll) {
    float acc = 0.f;
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * stride < n4; i += 4 * stride) {                  // four independent 16-byte loads in flight per lane
        const float4 a = x[i], b = x[i + stride], c = x[i + 2 * stride], d = x[i + 3 * stride];
        acc += a.x + a.y + a.z + a.w + b.x + b.y + b.z + b.w + c.x + c.y + c.z + c.w + d.x + d.y + d.z + d.w;
    }
    for (; i < n4; i += stride) { const float4 a = x[i]; acc += a.x + a.y + a.z + a.w; }
    if (acc == 123.456f) out[blockIdx.x] = acc;                     // (keeps the loads alive, practically never stores)
}
__global__ __launch_bounds__(256) void read4_kernel(const float* __restrict__ x, float* __restrict__ out, size_t n, int unroll) {
    float acc = 0.f;                                                // 4 bytes per lane, eight independent loads in flight
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 7 * stride < n; i += 8 * stride) {
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = x[i + k * stride];
#pragma unroll
        for (int k = 0; k < 8; ++k) acc += v[k];
    }
    for (; i < n; i += stride) acc += x[i];
    if (acc == 123.456f) out[blockIdx.x] = acc;
}
__global__ __launch_bounds__(256) void write_kernel(float4* __restrict__ y, size_t n4) {
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) y[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}
__global__ __launch_bounds__(256) void copy_kernel(float4* __restrict__ y, const float4* __restrict__ x, size_t n4) {
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + stride < n4; i += 2 * stride) { const float4 a = x[i], b = x[i + stride]; y[i] = a; y[i + stride] = b; }
    for (; i < n4; i += stride) y[i] = x[i];
}

int main() {
    for (double gb : {0.84, 3.36}) {
        const size_t n4 = (size_t)(gb * 1e9 / 16);
        float4 *x, *y; float* out;
        hipMalloc(&x, n4 * 16); hipMalloc(&y, n4 * 16); hipMalloc(&out, 1 << 20);
        hipMemset(x, 0, n4 * 16);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int grid : {2048, 8192, 32768, 131072}) {
            auto timed = [&](auto&& launch) {
                launch(); hipDeviceSynchronize();
                hipEventRecord(e0);
                for (int k = 0; k < 10; ++k) launch();
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); return ms / 10 * 1e-3;
            };
            const double tr = timed([&] { hipLaunchKernelGGL(read_kernel, dim3(grid), dim3(256), 0, 0, x, out, n4, 4); });
            const double tw = timed([&] { hipLaunchKernelGGL(write_kernel, dim3(grid), dim3(256), 0, 0, y, n4); });
            const double tc = timed([&] { hipLaunchKernelGGL(copy_kernel, dim3(grid), dim3(256), 0, 0, y, x, n4); });
            const double t4 = timed([&] { hipLaunchKernelGGL(read4_kernel, dim3(grid), dim3(256), 0, 0, (const float*)x, out, n4 * 4, 8); });
            // the same stream starting one float past a 16-byte boundary: every 256-byte wave access straddles three 128-byte lines
            const double t4u = timed([&] { hipLaunchKernelGGL(read4_kernel, dim3(grid), dim3(256), 0, 0, (const float*)x + 1, out, n4 * 4 - 1, 8); });
            printf("   4-byte loads, misaligned by one float: %.2f TB/s\n", n4 * 16 / t4u / 1e12);
            printf("%.2f GB, %5d workgroups: read %.2f TB/s (4-byte loads %.2f)   write %.2f TB/s   copy %.2f TB/s (read + write)\n", gb, grid,
                   n4 * 16 / tr / 1e12, n4 * 16 / t4 / 1e12, n4 * 16 / tw / 1e12, 2.0 * n4 * 16 / tc / 1e12);
        }
        hipFree(x); hipFree(y); hipFree(out);
    }
    return 0;
}
