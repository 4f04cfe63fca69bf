// Probe: does FP32 MFMA (v_mfma_f32_32x32x2_f32) overlap with VALU / LDS work issued (a) by ANOTHER wave on the same SIMD, (b) by the
// same wave between its MFMAs?  Decides whether a producer/consumer (wave-specialised) Winograd kernel can hide the input transform.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_coexec tools/probes/mfma_coexec.hip && ./mfma_coexec
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v2f __attribute__((ext_vector_type(2)));

// mode bits: 1 = waves 0..3 run MFMAs, 2 = waves 4..7 run VALU adds, 4 = waves 4..7 run LDS read/write traffic, 8 = waves 4..7 run packed adds
// same-wave variants: mode 16+k = each MFMA wave issues k VALU adds between two MFMAs; 32+k = k ds_read_b64 between MFMAs
template <int mode, int fill>
__global__ __launch_bounds__(512) void probe(float* out, int iters) {
    __shared__ float lds[16384];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 16384; i += 512) lds[i] = i * 1e-6f;
    __syncthreads();
    float r = 0.f;
    if (wave < 4) {
        if (!(mode & 1)) return;
        f32x16 acc[4];
        for (int j = 0; j < 4; ++j) for (int k = 0; k < 16; ++k) acc[j][k] = 0.f;
        float a = lane * 1e-3f, b = 1.f + lane * 1e-4f;
        float v[8] = {a, b, a + 1, b + 1, a + 2, b + 2, a + 3, b + 3};
        const v2f* lp = reinterpret_cast<const v2f*>(lds) + lane;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
                if ((mode & 16)) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) if (k < fill) v[k] = v[k] + b;
                }
                if ((mode & 32)) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) if (k < fill) { v2f t = lp[((i * 4 + j) * 8 + k) * 64 & 4095]; v[k] += t.x; }
                }
            }
        }
        for (int j = 0; j < 4; ++j) for (int k = 0; k < 16; ++k) r += acc[j][k];
        for (int k = 0; k < 8; ++k) r += v[k];
    } else {
        if (mode & 2) {
            float v[8];
            for (int k = 0; k < 8; ++k) v[k] = lane + k;
            const float c = 1.0001f;
            for (int i = 0; i < iters * fill; ++i) {
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = v[k] + c;
            }
            for (int k = 0; k < 8; ++k) r += v[k];
        } else if (mode & 8) {
            v2f v[8];
            for (int k = 0; k < 8; ++k) v[k] = v2f{(float)lane + k, 1.f};
            const v2f c = {1.0001f, 0.5f};
            for (int i = 0; i < iters * fill; ++i) {
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = v[k] + c;
            }
            for (int k = 0; k < 8; ++k) r += v[k].x + v[k].y;
        } else if (mode & 4) {
            float* mine = lds + (wave - 4) * 4096 + lane;
            float v[8];
            for (int k = 0; k < 8; ++k) v[k] = 0.f;
            for (int i = 0; i < iters * fill; ++i) {
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] += mine[k * 64 + (i & 7) * 256];
#pragma unroll
                for (int k = 0; k < 4; ++k) mine[(k + 4) * 64 + (i & 7) * 256] = v[k];
            }
            for (int k = 0; k < 8; ++k) r += v[k];
        } else return;
    }
    if (r == 123.456f) out[threadIdx.x] = r;
}

template <int mode, int fill>
static float run(int iters) {
    float* out; hipMalloc(&out, 4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    probe<mode, fill><<<256, 512>>>(out, iters / 8); hipDeviceSynchronize();
    hipEventRecord(e0);
    probe<mode, fill><<<256, 512>>>(out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipFree(out);
    return ms;
}

#define R(m, f) run<m, f>(iters)
int main() {
    const int iters = 20000;               // x4 MFMAs per iteration per wave
    const double mfma_cyc = 4.0 * iters * 64;
    printf("pure MFMA expectation at 2.4 GHz: %.3f ms\n", mfma_cyc / 2.4e6);
    printf("MFMA waves only                                  : %.3f ms\n", R(1, 0));
    printf("VALU waves only   (8/16/32 adds per 4 MFMA-slots): %.3f %.3f %.3f ms\n", R(2, 1), R(2, 2), R(2, 4));
    printf("MFMA + VALU waves (8/16/32 adds per 4 MFMAs)     : %.3f %.3f %.3f ms\n", R(3, 1), R(3, 2), R(3, 4));
    printf("pk-add waves only (8/16/32 per 4 MFMA-slots)     : %.3f %.3f %.3f ms\n", R(8, 1), R(8, 2), R(8, 4));
    printf("MFMA + pk-add waves                              : %.3f %.3f %.3f ms\n", R(9, 1), R(9, 2), R(9, 4));
    printf("LDS waves only    (4rd+4wr x 1/2/4 per 4 slots)  : %.3f %.3f %.3f ms\n", R(4, 1), R(4, 2), R(4, 4));
    printf("MFMA + LDS waves                                 : %.3f %.3f %.3f ms\n", R(5, 1), R(5, 2), R(5, 4));
    printf("same wave, 1/2/4/8 VALU adds after each MFMA     : %.3f %.3f %.3f %.3f ms\n", R(17, 1), R(17, 2), R(17, 4), R(17, 8));
    printf("same wave, 1/2/4/8 ds_read_b64 after each MFMA   : %.3f %.3f %.3f %.3f ms\n", R(33, 1), R(33, 2), R(33, 4), R(33, 8));
    return 0;
}
