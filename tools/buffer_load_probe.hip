// Probe of raw buffer loads on gfx950: SGPR resource + 32-bit lane offset, out-of-range offsets return 0 (the zero padding of a
// convolution footprint for free): hipcc --offload-arch=gfx950 -O2 tools/buffer_load_probe.hip -o /tmp/p && /tmp/p
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(const float* g, float* out, int n, int soff) {
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)g, 0, n * 4, 0x00020000);
    const unsigned off = threadIdx.x < 200 ? threadIdx.x * 4u : 0xFFFFFFF0u;
    const unsigned bits = __builtin_amdgcn_raw_buffer_load_b32(r, off, soff, 0);
    out[threadIdx.x] = __builtin_bit_cast(float, bits);
}
int main() {
    float *g, *o; hipMalloc(&g, 4096 * 4); hipMalloc(&o, 256 * 4);
    float h[4096]; for (int i = 0; i < 4096; ++i) h[i] = i + 0.5f;
    hipMemcpy(g, h, sizeof(h), hipMemcpyHostToDevice);
    k<<<1, 256>>>(g, o, 128, 40);            // 128 records: lanes whose (offset + soffset) / 4 >= 128 read 0
    float r[256]; hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 256; ++i) {
        const float want = (i < 200 && i + 10 < 128) ? (i + 10) + 0.5f : 0.f;
        if (r[i] != want) { if (bad < 5) printf("lane %d got %g want %g\n", i, r[i], want); ++bad; }
    }
    printf("bad %d  r[0] %g r[117] %g r[118] %g r[199] %g r[255] %g\n", bad, r[0], r[117], r[118], r[199], r[255]);
    return 0;
}
