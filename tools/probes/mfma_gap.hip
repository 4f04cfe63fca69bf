// Probe: what does a wave pay per v_mfma_f32_32x32x2_f32 when its MFMAs are issued back to back, and when something else sits between
// them?  Reports shader-clock cycles per MFMA (s_memtime) and the clock those cycles ran at (cycles / wall time).
//   hipcc --offload-arch=gfx950 -O3 -o mfma_gap tools/probes/mfma_gap.hip && ./mfma_gap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

// FILL: 0 none, 1 `s_nop 0`, 2 `s_nop 1`, 3 `s_nop 3`, 4 `s_nop 7`, 5 `v_nop`, 7 two `s_nop 0`, 8 v_add_f32 (independent), 9 s_nop 0 after every 2nd MFMA; 11 v_pk_add_f32, 12 ds_read_b32, 13 buffer_load_dword, 14 / 15 two / four v_add_f32, 16 s_waitcnt, 17 two s_cselect_b32 (scalar ALU)
template <int FILL, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void probe(float* out, long long* cyc, int iters) {
    f32x16 acc[4];
    for (int j = 0; j < 4; ++j) for (int k = 0; k < 16; ++k) acc[j][k] = 0.f;
    const int lane = threadIdx.x & 63;
    float a = lane * 1e-3f, b = 1.f + lane * 1e-4f, v = a;
    unsigned s = 0;
    typedef float v2f __attribute__((ext_vector_type(2)));
    v2f v2 = {a, b}, b2 = {b, a};
    float w = b, x = a + 1, y = b + 1;
    __shared__ float lds[256];
    lds[threadIdx.x & 255] = a;
    const unsigned ldsa = (threadIdx.x & 63) * 4;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)out, 0, 4096, 0x00020000);
    const long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
            if (FILL == 1) asm volatile("s_nop 0");
            if (FILL == 2) asm volatile("s_nop 1");
            if (FILL == 3) asm volatile("s_nop 3");
            if (FILL == 4) asm volatile("s_nop 7");
            if (FILL == 5) asm volatile("v_nop");
            if (FILL == 11) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(v2) : "v"(b2));
            if (FILL == 12) { float t; asm volatile("ds_read_b32 %0, %1" : "=v"(t) : "v"(ldsa)); }
            if (FILL == 13) { float t; asm volatile("buffer_load_dword %0, %1, %2, 0 offen" : "=v"(t) : "v"(0), "s"(rs)); }
            if (FILL == 14) asm volatile("v_add_f32 %0, %0, %1\n\tv_add_f32 %2, %2, %1" : "+v"(v), "+v"(w) : "v"(b));
            if (FILL == 15) asm volatile("v_add_f32 %0, %0, %1\n\tv_add_f32 %2, %2, %1\n\tv_add_f32 %3, %3, %1\n\tv_add_f32 %4, %4, %1" : "+v"(v), "+v"(w), "+v"(x), "+v"(y) : "v"(b));
            if (FILL == 16) asm volatile("s_waitcnt vmcnt(0)");
            if (FILL == 17) asm volatile("s_cselect_b32 s40, s41, s42\n\ts_cselect_b32 s43, s41, s42" ::: "s40", "s43");
            if (FILL == 7) asm volatile("s_nop 0\n\ts_nop 0");
            if (FILL == 8) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v) : "v"(b));
            if (FILL == 9 && (j & 1)) asm volatile("s_nop 0");
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float r = v + (float)s + v2.x + v2.y + w + x + y;
    for (int j = 0; j < 4; ++j) for (int k = 0; k < 16; ++k) r += acc[j][k];
    if (r == 123.456f) out[threadIdx.x] = r;
    if (blockIdx.x == 0 && threadIdx.x == 0) *cyc = t1 - t0;
}

template <int FILL, int WAVES>
static void run(const char* what, int iters) {
    float* out; long long* cyc; (void)hipMalloc(&out, 4096); (void)hipMalloc(&cyc, 8);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    probe<FILL, WAVES><<<256 * 4 / (WAVES >= 4 ? 4 : 1) * (WAVES >= 4 ? 1 : 1), 64 * WAVES>>>(out, cyc, iters / 8); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    probe<FILL, WAVES><<<256 * 4 / (WAVES >= 4 ? 4 : 1), 64 * WAVES>>>(out, cyc, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    long long c; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-34s waves/SIMD %d: %.3f ms, %.2f counter ticks per MFMA, %.1f ns per MFMA\n", what, WAVES >= 4 ? WAVES / 4 : 1, ms, (double)c / (4.0 * iters),
           ms * 1e6 / (4.0 * iters));
    (void)hipFree(out); (void)hipFree(cyc);
}

int main() {
    const int iters = 20000;
    run<0, 4>("back to back", iters);
    run<1, 4>("s_nop 0 after each", iters);
    run<2, 4>("s_nop 1 after each", iters);
    run<3, 4>("s_nop 3 after each", iters);
    run<4, 4>("s_nop 7 after each", iters);
    run<5, 4>("v_nop after each", iters);
    run<7, 4>("2 x s_nop 0 after each", iters);
    run<8, 4>("v_add_f32 after each", iters);
    run<9, 4>("s_nop 0 after every 2nd", iters);
    run<17, 4>("2 x s_cselect_b32 after each", iters);
    run<11, 4>("v_pk_add_f32 after each", iters);
    run<14, 4>("2 x v_add_f32 after each", iters);
    run<15, 4>("4 x v_add_f32 after each", iters);
    run<12, 4>("ds_read_b32 after each", iters);
    run<13, 4>("buffer_load_dword after each", iters);
    run<16, 4>("s_waitcnt vmcnt(0) after each", iters);
    run<0, 8>("back to back", iters);
    run<15, 8>("4 x v_add_f32 after each", iters);
    run<12, 8>("ds_read_b32 after each", iters);
    run<13, 8>("buffer_load_dword after each", iters);
    run<1, 8>("s_nop 0 after each", iters);
    run<8, 8>("v_add_f32 after each", iters);
    return 0;
}
