// The LBP matching-distance objective of 1024_example_LBP_percept.py on the device, per candidate:
//   misc.to_pil(img)                                (:162-163; misc.py:115-116: rint(x * 127.5 + 127.5) clipped to 0..255, RGB)
//   cv2.cvtColor(im, cv2.COLOR_BGR2GRAY)            (:48: the RGB array handed over as BGR, so RED gets the blue weight)
//   cv2.resize(gray, (224, 224))                    (:49: INTER_LINEAR on uint8, OpenCV's 11-bit fixed-point scheme)
//   skimage.feature.local_binary_pattern(image, 24, 3, 'uniform')    (:50; :35-37)
//   1 - dot(x, y) / (sqrt(dot(x, x)) * sqrt(dot(y, y)))               (:54-55, x / y the flattened code maps of candidate / target, float64)
// Contract: include/mgf.h (mgf_lbp_gray224_u8, mgf_lbp_codes_u8, mgf_lbp_distance_f64).  OpenCV and scikit-image are third-party packages
// absent from the reference tree: their published algorithms are restated (oracle/loss_ref.py holds the CPU restatement and its
// hand-computed answers).  Everything up to the code maps is integer or mirrors skimage's float64 expression order without fused
// multiply-adds, the dot products are integers: the result is bit-reproducible and equal to the oracle's.
#include "mgf_common.h"

namespace {

constexpr int LBP_S = 224;            // side of the resized gray image
constexpr int LBP_P = 24;             // sample points (8 * radius, radius 3)

__device__ __forceinline__ unsigned lbp_quant(float v) {                 // misc.to_pil
    v = rintf(__fadd_rn(__fmul_rn(v, 127.5f), 127.5f));           // two roundings like numpy's `data * scale + bias` (no fused multiply-add)
    return (unsigned)fminf(fmaxf(v, 0.f), 255.f);
}

// grid = (ceil(224 * 224 / 256), n).  tab: int32 [2][224][4] = per output column / row {src index 0, src index 1, coefficient 0, coefficient 1}
// (drivers.cv_resize_tables: half-pixel centres, coefficients rounded to 11 bits like saturate_cast<short>(c * 2048)).
__global__ __launch_bounds__(256) void lbp_gray224_kernel(uint8_t* out, const float* img, const int32_t* tab, int h, int w, int swap_rb) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= LBP_S * LBP_S) return;
    const int oy = i / LBP_S, ox = i - oy * LBP_S;
    const int32_t* tx = tab + ox * 4;
    const int32_t* ty = tab + (LBP_S + oy) * 4;
    const int64_t hw = (int64_t)h * w;
    const float* x = img + (int64_t)blockIdx.y * 3 * hw;
    const float* c0 = swap_rb ? x + 2 * hw : x;                          // the channel in OpenCV's BLUE slot (weight 1868)
    const float* c2 = swap_rb ? x : x + 2 * hw;
    auto gray = [&](int yy, int xx) -> int {
        yy = min(max(yy, 0), h - 1); xx = min(max(xx, 0), w - 1);       // (the tables are the caller's: a wrong index must not leave the image)
        const int64_t o = (int64_t)yy * w + xx;
        return (int)((lbp_quant(c0[o]) * 1868u + lbp_quant(x[hw + o]) * 9617u + lbp_quant(c2[o]) * 4899u + (1u << 13)) >> 14);
    };
    const int s0 = gray(ty[0], tx[0]) * tx[2] + gray(ty[0], tx[1]) * tx[3];                   // HResizeLinear, scale 2048
    const int s1 = gray(ty[1], tx[0]) * tx[2] + gray(ty[1], tx[1]) * tx[3];
    const int v = (((ty[2] * (s0 >> 4)) >> 16) + ((ty[3] * (s1 >> 4)) >> 16) + 2) >> 2;       // VResizeLinear<uchar>
    out[(int64_t)blockIdx.y * LBP_S * LBP_S + i] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// skimage/feature/_texture.pyx (_local_binary_pattern, method 'uniform') on one 224 x 224 image per blockIdx.y, a thread per pixel.
// off: float64 [2][24] = {rp, cp}, the sample offsets rounded to 5 decimals on the host (np.round(-R sin(2 pi p / P), 5), np.round(R cos(..), 5)).
// Writes the code map (codes != NULL) and / or the workgroup's partial sums {sum code * tcode, sum code^2} (part != NULL).
__global__ __launch_bounds__(256) void lbp_code_kernel(uint8_t* codes, unsigned long long* part, const uint8_t* gray, const uint8_t* tcodes,
                                                       const double* off) {
#pragma clang fp contract(off)
    __shared__ double soff[2 * LBP_P];
    __shared__ unsigned long long red[2][256];
    if (threadIdx.x < 2 * LBP_P) soff[threadIdx.x] = off[threadIdx.x];
    __syncthreads();
    const uint8_t* g = gray + (int64_t)blockIdx.y * LBP_S * LBP_S;
    const int i = blockIdx.x * 256 + threadIdx.x;
    unsigned long long dot = 0, sq = 0;
    if (i < LBP_S * LBP_S) {
        const int r = i / LBP_S, c = i - r * LBP_S;
        const double centre = (double)g[i];
        auto px = [&](long rr, long cc) -> double {                        // get_pixel2d, mode 'C', cval 0
            return (rr < 0 || rr >= LBP_S || cc < 0 || cc >= LBP_S) ? 0.0 : (double)g[rr * LBP_S + cc];
        };
        unsigned bits = 0;
#pragma unroll 4
        for (int p = 0; p < LBP_P; ++p) {
            const double rr = (double)r + soff[p], cc = (double)c + soff[LBP_P + p];
            const double fr = floor(rr), fc = floor(cc);
            const long minr = (long)fr, minc = (long)fc, maxr = (long)ceil(rr), maxc = (long)ceil(cc);
            const double dr = rr - (double)minr, dc = cc - (double)minc;
            const double top = (1.0 - dc) * px(minr, minc) + dc * px(minr, maxc);
            const double bottom = (1.0 - dc) * px(maxr, minc) + dc * px(maxr, maxc);
            const double tex = (1.0 - dr) * top + dr * bottom;
            bits |= (tex - centre >= 0.0 ? 1u : 0u) << p;
        }
        const int changes = __popc((bits ^ (bits >> 1)) & ((1u << (LBP_P - 1)) - 1u));          // 0 - 1 changes between neighbours p, p + 1 (not circular)
        const unsigned code = changes <= 2 ? (unsigned)__popc(bits) : (unsigned)(LBP_P + 1);
        if (codes) codes[(int64_t)blockIdx.y * LBP_S * LBP_S + i] = (uint8_t)code;
        if (part) { dot = (unsigned long long)code * tcodes[i]; sq = (unsigned long long)code * code; }
    }
    if (part) {
        red[0][threadIdx.x] = dot; red[1][threadIdx.x] = sq;
        __syncthreads();
        for (int s = 128; s >= 1; s >>= 1) {
            if ((int)threadIdx.x < s) { red[0][threadIdx.x] += red[0][threadIdx.x + s]; red[1][threadIdx.x] += red[1][threadIdx.x + s]; }
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            part[((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 2] = red[0][0];
            part[((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 2 + 1] = red[1][0];
        }
    }
}

// grid = n: 1 - dot(x, y) / (sqrt(dot(x, x)) * sqrt(dot(y, y))) in float64; the dots are exact integers
__global__ __launch_bounds__(64) void lbp_finish_kernel(double* out, const unsigned long long* part, int blocks, const uint8_t* tcodes) {
#pragma clang fp contract(off)
    unsigned long long dot = 0, sq = 0, tsq = 0;
    for (int i = threadIdx.x; i < blocks; i += 64) { dot += part[((int64_t)blockIdx.x * blocks + i) * 2]; sq += part[((int64_t)blockIdx.x * blocks + i) * 2 + 1]; }
    for (int i = threadIdx.x; i < LBP_S * LBP_S; i += 64) tsq += (unsigned long long)tcodes[i] * tcodes[i];          // dot(y, y) of the target (50 176 bytes)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { dot += __shfl_xor(dot, o, 64); sq += __shfl_xor(sq, o, 64); tsq += __shfl_xor(tsq, o, 64); }
    if (threadIdx.x == 0) out[blockIdx.x] = 1.0 - (double)dot / (sqrt((double)sq) * sqrt((double)tsq));
}

}  // namespace

extern "C" int64_t mgf_lbp_scratch_bytes(int32_t n) {
    return n < 1 ? 0 : (int64_t)n * mgf_cdiv(LBP_S * LBP_S, 256) * 2 * (int64_t)sizeof(unsigned long long);
}

extern "C" int mgf_lbp_gray224_u8(uint8_t* gray, const float* img, const int32_t* tables, int32_t n, int32_t h, int32_t w, int32_t true_rgb_order,
                                  mgf_stream_t stream) {
    MGF_REQUIRE(gray && img && tables && n >= 1 && n <= 65535 && h >= 1 && w >= 1, MGF_EINVAL, "lbp_gray224: bad arguments");
    hipLaunchKernelGGL(lbp_gray224_kernel, dim3((unsigned)mgf_cdiv(LBP_S * LBP_S, 256), n), dim3(256), 0, (hipStream_t)stream, gray, img, tables, h, w,
                       true_rgb_order ? 1 : 0);
    MGF_CHECK_LAUNCH("lbp_gray224");
    return MGF_OK;
}

extern "C" int mgf_lbp_codes_u8(uint8_t* codes, const uint8_t* gray, const double* offsets, int32_t n, mgf_stream_t stream) {
    MGF_REQUIRE(codes && gray && offsets && n >= 1 && n <= 65535, MGF_EINVAL, "lbp_codes: bad arguments");
    hipLaunchKernelGGL(lbp_code_kernel, dim3((unsigned)mgf_cdiv(LBP_S * LBP_S, 256), n), dim3(256), 0, (hipStream_t)stream, codes, nullptr, gray, nullptr,
                       offsets);
    MGF_CHECK_LAUNCH("lbp_codes");
    return MGF_OK;
}

extern "C" int mgf_lbp_distance_f64(double* out, const uint8_t* gray, const uint8_t* target_codes, const double* offsets, int32_t n, void* scratch,
                                    mgf_stream_t stream) {
    MGF_REQUIRE(out && gray && target_codes && offsets && scratch && n >= 1 && n <= 65535, MGF_EINVAL, "lbp_distance: bad arguments");
    MGF_REQUIRE((uintptr_t)scratch % 8 == 0, MGF_EINVAL, "lbp_distance: scratch must be 8-byte aligned");
    const unsigned blocks = (unsigned)mgf_cdiv(LBP_S * LBP_S, 256);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(lbp_code_kernel, dim3(blocks, n), dim3(256), 0, st, nullptr, (unsigned long long*)scratch, gray, target_codes, offsets);
    hipLaunchKernelGGL(lbp_finish_kernel, dim3(n), dim3(64), 0, st, out, (const unsigned long long*)scratch, (int)blocks, target_codes);
    MGF_CHECK_LAUNCH("lbp_distance");
    return MGF_OK;
}
