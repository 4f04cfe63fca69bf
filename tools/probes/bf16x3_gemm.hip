// Probe: is the exact-FP32 plateau worth leaving?  The inner product of the convolution kernels -- D[co][px] +=
// W[co][k] X[k][px], K = input channels x taps -- three ways on the matrix cores of gfx950:
//   f32    : v_mfma_f32_32x32x2_f32, 8 instructions per 16 k (what the product kernels issue; exact f32, 64 cycles each)
//   bf16x3 : every f32 operand split into three bf16 terms a = a1 + a2 + a3 (a1 = bf16(a), a2 = bf16(a - a1), a3 = bf16(a - a1 - a2): 24
//            significant bits), products through v_mfma_f32_32x32x16_bf16 (32 cycles per 16 k) with f32 accumulation --
//            3 products (a1 b1, a1 b2, a2 b1: error ~ 2^-16 per product) or 6 (+ a1 b3, a3 b1, a2 b2: ~ 2^-23)
// The weights are split once on the host (a checkpoint constant); the activations are split IN the loop, by the wave that consumes them
// (the VALU work a real kernel would add), once per 16 k and reused over NT output-channel tiles of 32.
// Reports (a) the rate of each form with operands streamed from L2 -- an upper bound for a kernel built on it: no LDS staging, no epilogue --
// and (b) the worst error against float64 on the shapes of the 512^2 and 1024^2 transposed convolutions and the form-3 Winograd layers.
//   hipcc --offload-arch=gfx950 -O3 -o bf16x3_gemm tools/probes/bf16x3_gemm.hip && ./bf16x3_gemm
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

// ---- host side split (round to nearest even, like the device's cast) ----
static uint16_t bf16_rne(float f) {
    uint32_t u; memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
    return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
static float bf16_to_f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }

// MODE 0: f32 MFMA.  MODE 3 / 6: bf16 split with 3 / 6 products.
// Layouts.  X [K][M] f32 (pixel-contiguous rows, like an NCHW plane).  W f32 [K][N] (the tap-major [cin][cout_pad] image of the product
// kernels).  Wb bf16 [K / 16][term 0..2][N][16 k] -- lane (co = l & 31, h = l >> 5) reads its 8 k of one term as ONE 16-byte load.
// D [N][M] f32.  One wave per (32 px, 32 NT co) tile; grid = (M / 32, N / (32 NT)); 256 threads = 4 independent waves (4 px tiles).
template <int MODE, int NT>
__global__ __launch_bounds__(256) void gemm_probe(float* __restrict__ D, const float* __restrict__ X, const float* __restrict__ W,
                                                  const uint16_t* __restrict__ Wb, int M, int N, int K) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int px0 = (blockIdx.x * 4 + wv) * 32, co0 = blockIdx.y * 32 * NT;
    if (px0 >= M) return;
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    if (MODE == 0) {
        for (int k = 0; k < K; k += 16) {
            float b[8], a[NT][8];
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                b[s] = X[(int64_t)(k + 2 * s + h) * M + px0 + r];
#pragma unroll
                for (int t = 0; t < NT; ++t) a[t][s] = W[(int64_t)(k + 2 * s + h) * N + co0 + 32 * t + r];
            }
#pragma unroll
            for (int s = 0; s < 8; ++s)
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t][s], b[s], acc[t], 0, 0, 0);
        }
    } else {
        for (int k = 0; k < K; k += 16) {
            // this lane's 8 k of the activation operand: k + 8 h + j, pixel px0 + r (8 coalesced dword loads, as many as the f32 form's)
            float x[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = X[(int64_t)(k + 8 * h + j) * M + px0 + r];
            bf16x8 b1, b2, b3;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const __bf16 t1 = (__bf16)x[j];
                const float r1 = x[j] - (float)t1;                  // exact
                const __bf16 t2 = (__bf16)r1;
                b1[j] = t1; b2[j] = t2;
                if (MODE == 6) b3[j] = (__bf16)(r1 - (float)t2);
            }
            const uint16_t* wk = Wb + (int64_t)(k / 16) * 3 * N * 16;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int co = co0 + 32 * t + r;
                const bf16x8 a1 = __builtin_bit_cast(bf16x8, *reinterpret_cast<const s16x8*>(wk + ((int64_t)0 * N + co) * 16 + 8 * h));
                const bf16x8 a2 = __builtin_bit_cast(bf16x8, *reinterpret_cast<const s16x8*>(wk + ((int64_t)1 * N + co) * 16 + 8 * h));
                // smallest terms first (their sum is formed before it meets the large one)
                if (MODE == 6) {
                    const bf16x8 a3 = __builtin_bit_cast(bf16x8, *reinterpret_cast<const s16x8*>(wk + ((int64_t)2 * N + co) * 16 + 8 * h));
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b2, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, b1, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b3, acc[t], 0, 0, 0);
                }
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b1, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b2, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[t], 0, 0, 0);
            }
        }
    }
    // C/D map of every 32x32 form: column (here: pixel) = lane & 31, row (output channel) = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int co = co0 + 32 * t + (i & 3) + 8 * (i >> 2) + 4 * h;
            D[(int64_t)co * M + px0 + r] = acc[t][i];
        }
}

struct Problem { int M, N, K; const char* what; };

template <int MODE, int NT>
static float launch(float* D, const float* X, const float* W, const uint16_t* Wb, int M, int N, int K, int reps, int z = 1) {
    dim3 grid(M / 128, N / (32 * NT), z);      // z > 1: the same tiles again (rate runs: enough workgroups for 256 CUs; identical values are written)
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    gemm_probe<MODE, NT><<<grid, 256>>>(D, X, W, Wb, M, N, K);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) gemm_probe<MODE, NT><<<grid, 256>>>(D, X, W, Wb, M, N, K);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

int main() {
    // (what, M = pixels of the launch's slice, N = output channels, K = input channels x taps that meet in one output)
    const Problem probs[] = {
        {16384, 64, 2304, "transposed conv 256 -> 512 px (cin 128, cout 64): K = 4 taps x 128 .. 1 tap x 128; run at the 9-tap total"},
        {16384, 32, 1152, "transposed conv 512 -> 1024 px (cin 64, cout 32)"},
        {16384, 64, 576, "3x3 at 512 px (cin 64, cout 64), direct-form K"},
        {16384, 32, 288, "3x3 at 1024 px (cin 32, cout 32), direct-form K"},
        {4096, 512, 4608, "3x3 at 64 px (cin 512, cout 512)"},
    };
    printf("%-72s %9s %9s %9s   %s\n", "shape", "f32", "bf16x3/3", "bf16x3/6", "(TFLOP/s of 2 M N K; x = speed-up over f32)");
    for (const Problem& p : probs) {
        const int M = p.M, N = std::max(p.N, 128), K = p.K;      // N padded to 128 so that NT = 4 tiles exist (rate only)
        std::vector<float> hx((size_t)K * M), hw((size_t)K * N);
        srand(1234);
        auto rnd = [] { float u = 0.f; for (int i = 0; i < 6; ++i) u += (float)rand() / (float)RAND_MAX; return (u - 3.f) * 1.4142f; };       // ~ N(0, 1)
        for (auto& v : hx) v = rnd();
        const float ws = 1.f / std::sqrt((float)K);
        for (auto& v : hw) v = rnd() * ws;
        std::vector<uint16_t> hwb((size_t)(K / 16) * 3 * N * 16);
        for (int k = 0; k < K; ++k)
            for (int co = 0; co < N; ++co) {
                const float a = hw[(size_t)k * N + co];
                const uint16_t t1 = bf16_rne(a); const float r1 = a - bf16_to_f(t1);
                const uint16_t t2 = bf16_rne(r1); const float r2 = r1 - bf16_to_f(t2);
                const uint16_t t3 = bf16_rne(r2);
                const uint16_t tt[3] = {t1, t2, t3};
                for (int term = 0; term < 3; ++term) hwb[(((size_t)(k / 16) * 3 + term) * N + co) * 16 + (k % 16)] = tt[term];
            }
        float *dx, *dw, *dd; uint16_t* dwb;
        CHECK(hipMalloc(&dx, hx.size() * 4)); CHECK(hipMalloc(&dw, hw.size() * 4)); CHECK(hipMalloc(&dwb, hwb.size() * 2));
        CHECK(hipMalloc(&dd, (size_t)N * M * 4));
        CHECK(hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(dw, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(dwb, hwb.data(), hwb.size() * 2, hipMemcpyHostToDevice));
        // ---- accuracy on a 64-pixel x 128-channel corner against float64 ----
        const int PM = 64;
        std::vector<double> ref((size_t)N * PM), mag((size_t)N * PM);
        for (int co = 0; co < N; ++co)
            for (int px = 0; px < PM; ++px) {
                double s = 0, m = 0;
                for (int k = 0; k < K; ++k) { const double a = hw[(size_t)k * N + co], b = hx[(size_t)k * M + px]; s += a * b; m += std::fabs(a * b); }
                ref[(size_t)co * PM + px] = s; mag[(size_t)co * PM + px] = m;
            }
        double refmax = 0; for (double v : ref) refmax = std::max(refmax, std::fabs(v));
        std::vector<float> got((size_t)N * M);
        double err[3] = {0, 0, 0}, erm[3] = {0, 0, 0};
        for (int mode = 0; mode < 3; ++mode) {
            CHECK(hipMemset(dd, 0, (size_t)N * M * 4));
            if (mode == 0) launch<0, 4>(dd, dx, dw, dwb, M, N, K, 1);
            else if (mode == 1) launch<3, 4>(dd, dx, dw, dwb, M, N, K, 1);
            else launch<6, 4>(dd, dx, dw, dwb, M, N, K, 1);
            CHECK(hipMemcpy(got.data(), dd, got.size() * 4, hipMemcpyDeviceToHost));
            for (int co = 0; co < N; ++co)
                for (int px = 0; px < PM; ++px) {
                    const double e = std::fabs((double)got[(size_t)co * M + px] - ref[(size_t)co * PM + px]);
                    err[mode] = std::max(err[mode], e / refmax);
                    erm[mode] = std::max(erm[mode], e / mag[(size_t)co * PM + px]);
                }
        }
        // ---- rate ----
        const int Z = 16;
        const double gf = 2.0 * M * N * K * 1e-9 * Z;
        const int reps = 10;
        float t[3][3];
        t[0][0] = launch<0, 1>(dd, dx, dw, dwb, M, N, K, reps, Z); t[0][1] = launch<0, 2>(dd, dx, dw, dwb, M, N, K, reps, Z); t[0][2] = launch<0, 4>(dd, dx, dw, dwb, M, N, K, reps, Z);
        t[1][0] = launch<3, 1>(dd, dx, dw, dwb, M, N, K, reps, Z); t[1][1] = launch<3, 2>(dd, dx, dw, dwb, M, N, K, reps, Z); t[1][2] = launch<3, 4>(dd, dx, dw, dwb, M, N, K, reps, Z);
        t[2][0] = launch<6, 1>(dd, dx, dw, dwb, M, N, K, reps, Z); t[2][1] = launch<6, 2>(dd, dx, dw, dwb, M, N, K, reps, Z); t[2][2] = launch<6, 4>(dd, dx, dw, dwb, M, N, K, reps, Z);
        printf("%s  [M %d, N %d (padded), K %d]\n", p.what, M, N, K);
        for (int nt = 0; nt < 3; ++nt)
            printf("  %3d output channels per wave: %56s %8.1f  %8.1f  %8.1f   x%.2f  x%.2f\n", 32 << nt, "", gf / t[0][nt], gf / t[1][nt], gf / t[2][nt],
                   t[0][nt] / t[1][nt], t[0][nt] / t[2][nt]);
        printf("  worst |err| / max|D|        : f32 %.2e   bf16x3/3 %.2e   bf16x3/6 %.2e\n", err[0], err[1], err[2]);
        printf("  worst |err| / sum|a b|      : f32 %.2e   bf16x3/3 %.2e   bf16x3/6 %.2e\n", erm[0], erm[1], erm[2]);
        CHECK(hipFree(dx)); CHECK(hipFree(dw)); CHECK(hipFree(dwb)); CHECK(hipFree(dd));
    }
    return 0;
}
