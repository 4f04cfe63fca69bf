// The two ends of the LPIPS(squeeze) stem outside the fused stem kernel (gradient mode keeps the stem's feature map for its backward
// pass, lpips/pretrained_networks.py:7-44 features.0): a 3x3 / stride-2 convolution with THREE input channels, and its data gradient, a
// 3x3 / stride-2 transposed convolution with THREE output channels.  On the MFMA tap-list kernel the narrow side is padded to a
// 32-row tile and to 8-channel K chunks (420 us forward, 741 us + border backward for 8 x 1024^2 images: 27 real multiplies per
// output in a K of 72, 3 live rows of 32); both are streams over the 64-channel map with ~0.25 FLOP per byte, so they run here as
// VALU kernels: the narrow side in registers, the weights broadcast from LDS, every access of the wide map coalesced.
// Contracts: include/mgf.h (mgf_conv3x3s2_few_inputs_f32, mgf_tconv3x3s2_few_outputs_f32).
#include "mgf_common.h"

namespace {

constexpr int NC_MAX_NARROW = 4;      // channels on the narrow side
constexpr int NC_MAX_WIDE = 512;      // channels on the wide side (LDS weight image)

// y[n, co, oy, ox] = act(bias[co] + sum_{ci, kh, kw} wp[kh*3+kw][ci][co] * x[n, ci, 2 oy + kh, 2 ox + kw]),  no padding
__global__ __launch_bounds__(256) void conv3x3s2_few_inputs_kernel(float* __restrict__ y, const float* __restrict__ x, const float* __restrict__ wp,
                                                                   const float* __restrict__ bias, int cin, int in_h, int in_w, int cout,
                                                                   int cout_pad, int out_h, int out_w, int relu) {
    extern __shared__ float ws[];                        // [cout][cin * 9 (+ pad to a multiple of 4)] then [cout] biases
    const int K = cin * 9, KP = (K + 3) & ~3;
    const int tid = threadIdx.y * 64 + threadIdx.x;
    for (int i = tid; i < cout * KP; i += 256) {
        const int co = i / KP, k = i - co * KP;          // k = ci * 9 + tap
        float v = 0.f;
        if (k < K) { const int ci = k / 9, t = k - ci * 9; v = wp[((int64_t)t * cin + ci) * cout_pad + co]; }
        ws[i] = v;
    }
    float* bs = ws + cout * KP;
    for (int i = tid; i < cout; i += 256) bs[i] = bias ? bias[i] : 0.f;
    __syncthreads();
    const int n = blockIdx.z, oy = blockIdx.y * 4 + threadIdx.y, ox = blockIdx.x * 64 + threadIdx.x;
    if (oy >= out_h || ox >= out_w) return;
    float v[NC_MAX_NARROW * 9];
    const float* xb = x + (int64_t)n * cin * in_h * in_w + (int64_t)(2 * oy) * in_w + 2 * ox;
#pragma unroll
    for (int ci = 0; ci < NC_MAX_NARROW; ++ci)
#pragma unroll
        for (int t = 0; t < 9; ++t) v[ci * 9 + t] = ci < cin ? xb[(int64_t)ci * in_h * in_w + (t / 3) * in_w + (t % 3)] : 0.f;
    float* yb = y + (int64_t)n * cout * out_h * out_w + (int64_t)oy * out_w + ox;
    const int64_t plane = (int64_t)out_h * out_w;
    for (int co = 0; co < cout; ++co) {
        const float* wr = ws + co * KP;
        float acc = bs[co];
#pragma unroll
        for (int k = 0; k < NC_MAX_NARROW * 9; ++k)
            if (k < K) acc += wr[k] * v[k];
        yb[co * plane] = relu ? fmaxf(acc, 0.f) : acc;
    }
}

// t[n, co, 2 i + kh, 2 j + kw] += wp[kh*3+kw][ci][co] * x[n, ci, i, j]  (output [2h+1] x [2w+1], row pitch `pitch`): one lane per 2 x 2
// output quad (2i + {0,1}, 2j + {0,1}), i in [0, h], j in [0, w]:
//   (0,0) <- w00 x[i][j] + w02 x[i][j-1] + w20 x[i-1][j] + w22 x[i-1][j-1]     (0,1) <- w01 x[i][j] + w21 x[i-1][j]
//   (1,0) <- w10 x[i][j] + w12 x[i][j-1]                                       (1,1) <- w11 x[i][j]
template <int CO>
__global__ __launch_bounds__(256) void tconv3x3s2_few_outputs_kernel(float* __restrict__ y, const float* __restrict__ x, const float* __restrict__ wp,
                                                                     int cin, int h, int w, int cout_pad, int pitch, int64_t y_plane,
                                                                     int64_t y_batch) {
    extern __shared__ float ws[];                        // [cin][9][CO padded to 4]
    const int tid = threadIdx.y * 64 + threadIdx.x;
    for (int i = tid; i < cin * 36; i += 256) {
        const int c = i / 36, r = i - c * 36, t = r >> 2, o = r & 3;
        ws[i] = o < CO ? wp[((int64_t)t * cin + c) * cout_pad + o] : 0.f;
    }
    __syncthreads();
    const int n = blockIdx.z, i = blockIdx.y * 4 + threadIdx.y, j = blockIdx.x * 64 + threadIdx.x;
    if (i > h || j > w) return;
    const bool r0 = i < h, r1 = i > 0, c0 = j < w, c1 = j > 0;         // x[i][.], x[i-1][.], x[.][j], x[.][j-1] exist
    const float* xb = x + (int64_t)n * cin * h * w + (int64_t)i * w + j;
    const int64_t plane = (int64_t)h * w;
    float a00[CO], a01[CO], a10[CO], a11[CO];
#pragma unroll
    for (int o = 0; o < CO; ++o) a00[o] = a01[o] = a10[o] = a11[o] = 0.f;
#pragma unroll 2
    for (int c = 0; c < cin; ++c) {
        const float* xc = xb + c * plane;
        const float p = (r0 && c0) ? xc[0] : 0.f, q = (r0 && c1) ? xc[-1] : 0.f;              // x[i][j], x[i][j-1]
        const float r = (r1 && c0) ? xc[-w] : 0.f, s = (r1 && c1) ? xc[-w - 1] : 0.f;         // x[i-1][j], x[i-1][j-1]
        const float4* wc = reinterpret_cast<const float4*>(ws + c * 36);
        float wt[9][4];
#pragma unroll
        for (int t = 0; t < 9; ++t) { const float4 f = wc[t]; wt[t][0] = f.x; wt[t][1] = f.y; wt[t][2] = f.z; wt[t][3] = f.w; }
#pragma unroll
        for (int o = 0; o < CO; ++o) {
            a00[o] += wt[0][o] * p + wt[2][o] * q + wt[6][o] * r + wt[8][o] * s;
            a01[o] += wt[1][o] * p + wt[7][o] * r;
            a10[o] += wt[3][o] * p + wt[5][o] * q;
            a11[o] += wt[4][o] * p;
        }
    }
    float* yb = y + (int64_t)n * y_batch + (int64_t)(2 * i) * pitch + 2 * j;
#pragma unroll
    for (int o = 0; o < CO; ++o) {
        float* yo = yb + o * y_plane;
        if (c0) *reinterpret_cast<float2*>(yo) = make_float2(a00[o], a01[o]);
        else yo[0] = a00[o];
        if (r0) {
            if (c0) *reinterpret_cast<float2*>(yo + pitch) = make_float2(a10[o], a11[o]);
            else yo[pitch] = a10[o];
        }
    }
}

}  // namespace

extern "C" int mgf_conv3x3s2_few_inputs_f32(float* y, const float* x, const float* wp, const float* bias, int32_t n, int32_t cin, int32_t in_h,
                                            int32_t in_w, int32_t cout, int32_t cout_pad, int32_t relu, mgf_stream_t stream) {
    MGF_REQUIRE(y && x && wp && n >= 1 && in_h >= 3 && in_w >= 3, MGF_EINVAL, "conv3x3s2_few_inputs: bad arguments");
    MGF_REQUIRE(cin >= 1 && cin <= NC_MAX_NARROW, MGF_EUNSUPPORTED, "conv3x3s2_few_inputs: 1..%d input channels (got %d)", NC_MAX_NARROW, cin);
    MGF_REQUIRE(cout >= 1 && cout <= NC_MAX_WIDE && cout_pad >= cout, MGF_EUNSUPPORTED, "conv3x3s2_few_inputs: 1..%d output channels (got %d, pad %d)",
                NC_MAX_WIDE, cout, cout_pad);
    MGF_REQUIRE(n <= 65535, MGF_ETOOBIG, "conv3x3s2_few_inputs: n must be <= 65535");
    const int out_h = (in_h - 3) / 2 + 1, out_w = (in_w - 3) / 2 + 1;
    MGF_REQUIRE((int64_t)cout * out_h * out_w <= INT32_MAX && (int64_t)cin * in_h * in_w <= INT32_MAX, MGF_ETOOBIG, "conv3x3s2_few_inputs: sample too large");
    const int KP = (cin * 9 + 3) & ~3;
    const size_t lds = (size_t)(cout * KP + cout) * sizeof(float);
    const dim3 grid((unsigned)mgf_cdiv(out_w, 64), (unsigned)mgf_cdiv(out_h, 4), n);
    hipLaunchKernelGGL(conv3x3s2_few_inputs_kernel, grid, dim3(64, 4), lds, (hipStream_t)stream, y, x, wp, bias, cin, in_h, in_w, cout, cout_pad,
                       out_h, out_w, relu);
    MGF_CHECK_LAUNCH("conv3x3s2_few_inputs");
    return MGF_OK;
}

extern "C" int mgf_tconv3x3s2_few_outputs_f32(float* y, const float* x, const float* wp, int32_t n, int32_t cin, int32_t h, int32_t w,
                                              int32_t cout, int32_t cout_pad, int32_t pitch, int64_t y_plane, int64_t y_batch,
                                              mgf_stream_t stream) {
    MGF_REQUIRE(y && x && wp && n >= 1 && h >= 1 && w >= 1, MGF_EINVAL, "tconv3x3s2_few_outputs: bad arguments");
    MGF_REQUIRE(cout >= 1 && cout <= NC_MAX_NARROW && cout_pad >= cout, MGF_EUNSUPPORTED, "tconv3x3s2_few_outputs: 1..%d output channels (got %d)",
                NC_MAX_NARROW, cout);
    MGF_REQUIRE(cin >= 1 && cin <= NC_MAX_WIDE, MGF_EUNSUPPORTED, "tconv3x3s2_few_outputs: 1..%d input channels (got %d)", NC_MAX_WIDE, cin);
    MGF_REQUIRE(pitch >= 2 * w + 1 && pitch % 2 == 0 && y_plane % 2 == 0 && y_batch % 2 == 0 && ((uintptr_t)y % 8) == 0 &&
                y_plane >= (int64_t)(2 * h + 1) * pitch && y_batch >= cout * y_plane, MGF_EINVAL,
                "tconv3x3s2_few_outputs: the output needs an even row pitch >= 2w+1, even plane / sample strides and an 8-byte aligned base");
    MGF_REQUIRE(n <= 65535 && (int64_t)cin * h * w <= INT32_MAX, MGF_ETOOBIG, "tconv3x3s2_few_outputs: tensor too large");
    const size_t lds = (size_t)cin * 36 * sizeof(float);
    const dim3 grid((unsigned)mgf_cdiv(w + 1, 64), (unsigned)mgf_cdiv(h + 1, 4), n);
    hipStream_t st = (hipStream_t)stream;
#define MGF_TCF_LAUNCH(CO) \
    hipLaunchKernelGGL(tconv3x3s2_few_outputs_kernel<CO>, grid, dim3(64, 4), lds, st, y, x, wp, cin, h, w, cout_pad, pitch, y_plane, y_batch)
    if (cout == 1) MGF_TCF_LAUNCH(1); else if (cout == 2) MGF_TCF_LAUNCH(2); else if (cout == 3) MGF_TCF_LAUNCH(3); else MGF_TCF_LAUNCH(4);
#undef MGF_TCF_LAUNCH
    MGF_CHECK_LAUNCH("tconv3x3s2_few_outputs");
    return MGF_OK;
}
