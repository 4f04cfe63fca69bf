// Probe: issue cost of LDS / buffer instructions interleaved with FP32 MFMAs in ONE wave per SIMD (v_mfma_f32_32x32x2_f32, 64 cycles each).
//   hipcc --offload-arch=gfx950 -O3 -o mfma_lds tools/probes/mfma_lds.hip && ./mfma_lds
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

// KIND: 0 none, 1 ds_read_b32, 2 ds_read_b64, 3 ds_read_b128, 4 ds_write_b32, 5 ds_write_b64, 6 ds_write_b128, 7 v_add_f32, 8 v_pk_add_f32,
//       9 v_mul_f32 (independent), 10 buffer_load_dword (L2-resident 64 KB window)
template <int KIND, int FILL, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void probe(float* out, const float* src, int iters) {
    __shared__ float lds[8192];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 8192; i += 64 * WAVES) lds[i] = i * 1e-6f;
    __syncthreads();
    f32x16 acc[4];
    for (int j = 0; j < 4; ++j) for (int k = 0; k < 16; ++k) acc[j][k] = 0.f;
    float a = lane * 1e-3f, b = 1.f + lane * 1e-4f;
    const unsigned addr32 = (wave * 1024 + lane) * 4, addr64 = (wave * 1024 + lane * 2) * 4, addr128 = (wave * 1024 + lane * 4) * 4;
    float v[8] = {a, b, a + 1, b + 1, a + 2, b + 2, a + 3, b + 3};
    v2f p2[4] = {{a, b}, {b, a}, {a, a}, {b, b}};
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 1 << 20, 0x00020000);
    const unsigned goff = (blockIdx.x * 256 + lane) * 4;
    float sink = 0.f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
#pragma unroll
            for (int k = 0; k < FILL; ++k) {
                if (KIND == 1) { float t; asm volatile("ds_read_b32 %0, %1" : "=v"(t) : "v"(addr32)); asm volatile("" :: "v"(t)); }
                if (KIND == 2) { v2f t; asm volatile("ds_read_b64 %0, %1" : "=v"(t) : "v"(addr64)); asm volatile("" :: "v"(t)); }
                if (KIND == 3) { v4f t; asm volatile("ds_read_b128 %0, %1" : "=v"(t) : "v"(addr128)); asm volatile("" :: "v"(t)); }
                if (KIND == 4) asm volatile("ds_write_b32 %0, %1" :: "v"(addr32), "v"(a));
                if (KIND == 5) asm volatile("ds_write_b64 %0, %1" :: "v"(addr64), "v"(p2[0]));
                if (KIND == 6) { v4f t = {a, b, a, b}; asm volatile("ds_write_b128 %0, %1" :: "v"(addr128), "v"(t)); }
                if (KIND == 7) asm volatile("v_add_f32 %0, %1, %2" : "=v"(v[k & 7]) : "v"(v[k & 7]), "v"(b));
                if (KIND == 8) asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(p2[k & 3]) : "v"(p2[k & 3]), "v"(p2[(k + 1) & 3]));
                if (KIND == 9) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(v[k & 7]) : "v"(v[(k + 1) & 7]), "v"(b));
                if (KIND == 10) { float t = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, goff, k * 256, 0)); asm volatile("" :: "v"(t)); }
            }
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
    for (int j = 0; j < 4; ++j) for (int k = 0; k < 16; ++k) sink += acc[j][k];
    for (int k = 0; k < 8; ++k) sink += v[k];
    for (int k = 0; k < 4; ++k) sink += p2[k].x + p2[k].y;
    if (sink == 123.456f) out[threadIdx.x] = sink;
}

template <int KIND, int FILL, int WAVES>
static float run(int iters) {
    float *out, *src; hipMalloc(&out, 4096); hipMalloc(&src, 1 << 20); hipMemset(src, 0, 1 << 20);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    probe<KIND, FILL, WAVES><<<256, 64 * WAVES>>>(out, src, iters / 8); hipDeviceSynchronize();
    hipEventRecord(e0);
    probe<KIND, FILL, WAVES><<<256, 64 * WAVES>>>(out, src, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipFree(out); hipFree(src);
    return ms;
}

#define ROW(name, K) \
    printf("%-18s 1 wave/SIMD: +1 %.3f  +2 %.3f  +4 %.3f  +8 %.3f | 2 waves/SIMD: +2 %.3f  +4 %.3f  +8 %.3f\n", name, run<K, 1, 4>(iters), run<K, 2, 4>(iters), \
           run<K, 4, 4>(iters), run<K, 8, 4>(iters), run<K, 2, 8>(iters), run<K, 4, 8>(iters), run<K, 8, 8>(iters));

int main() {
    const int iters = 20000;
    printf("4 MFMAs per iteration, %d iterations; ms per launch.  MFMA only: 1 wave/SIMD %.3f   2 waves/SIMD %.3f (twice the MFMAs)\n", iters, run<0, 0, 4>(iters),
           run<0, 0, 8>(iters));
    ROW("ds_read_b32", 1) ROW("ds_read_b64", 2) ROW("ds_read_b128", 3) ROW("ds_write_b32", 4) ROW("ds_write_b64", 5) ROW("ds_write_b128", 6)
    ROW("v_add_f32", 7) ROW("v_pk_add_f32", 8) ROW("v_mul_f32", 9) ROW("buffer_load_b32", 10)
    return 0;
}
