// Probe of global_load_lds on gfx950 (lane l of a wave writes its element to LDS base + l * size; 4- and 16-byte forms): hipcc --offload-arch=gfx950 -O2 tools/lds_dma_probe.hip -o /tmp/p && /tmp/p
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(const float* g, float* out) {
    __shared__ float lds[4096];
    const int wave = threadIdx.x >> 6;
    const float4* src = reinterpret_cast<const float4*>(g) + threadIdx.x;
    __builtin_amdgcn_global_load_lds(src, lds + wave * 256, 16, 0, 0);
    __builtin_amdgcn_global_load_lds(g + 2048 + (255 - threadIdx.x), lds + 2048 + wave * 64, 4, 0, 0);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    for (int i = threadIdx.x; i < 1024; i += 256) out[i] = lds[i];
    out[1024 + threadIdx.x] = lds[2048 + threadIdx.x];
}
int main() {
    float *g, *o; hipMalloc(&g, 4096 * 4); hipMalloc(&o, 2048 * 4);
    float h[4096]; for (int i = 0; i < 4096; ++i) h[i] = i;
    hipMemcpy(g, h, sizeof(h), hipMemcpyHostToDevice);
    k<<<1, 256>>>(g, o);
    float r[2048]; hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 1024; ++i) if (r[i] != (float)i) ++bad;
    for (int i = 0; i < 256; ++i) if (r[1024 + i] != (float)(2048 + 255 - i)) ++bad;
    printf("bad %d  r[0..7] %g %g %g %g %g %g %g %g  scalar %g %g\n", bad, r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7], r[1024], r[1025]);
    return 0;
}
