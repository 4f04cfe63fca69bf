// The two ends of the LPIPS(squeeze) stem outside the fused stem kernel (gradient mode keeps the stem's feature map for its backward
// pass, lpips/pretrained_networks.py:7-44 features.0): a 3x3 / stride-2 convolution with THREE input channels, and its data gradient, a
// 3x3 / stride-2 transposed convolution with THREE output channels.  On the MFMA tap-list kernel the narrow side is padded to a
// 32-row tile and to 8-channel K chunks (420 us forward, 741 us + border backward for 8 x 1024^2 images: 27 real multiplies per
// output in a K of 72, 3 live rows of 32); both are streams over the 64-channel map with ~0.25 FLOP per byte, so they run here as
// VALU kernels: the narrow side in registers, the weights through the scalar cache or broadcast from LDS, every access of the wide map
// coalesced.
// Contracts: include/mgf.h (mgf_conv3x3s2_few_inputs_f32, mgf_tconv3x3s2_few_outputs_f32).
#include "mgf_common.h"

namespace {

constexpr int NC_MAX_NARROW = 4;      // channels on the narrow side
constexpr int NC_MAX_WIDE = 1024;     // channels on the wide side of the transposed form (its LDS weight image: 144 bytes per channel)

// y[n, co, oy, ox] = act(bias[co] + sum_{ci, kh, kw} wp[kh*3+kw][ci][co] * x[n, ci, 2 oy + kh, 2 ox + kw]),  no padding
// (weights: the torch layout [cout][cin][3][3], read through the scalar cache -- every lane of a wave uses the same 27 values per
// output channel; broadcast reads of an LDS image cost 8 cycles per 16 bytes and bound the kernel at 1.8 TB/s)
typedef const float __attribute__((address_space(4)))* nc_cfp;

template <int CIN>
__global__ __launch_bounds__(256) void conv3x3s2_few_inputs_kernel(float* __restrict__ y, const float* __restrict__ x, const float* w,
                                                                   const float* bias, int in_h, int in_w, int cout, int out_h, int out_w,
                                                                   int relu) {
    constexpr int K = CIN * 9;
    // 256 x 1 outputs per workgroup: 1 KB of every output plane per workgroup (64 x 4 tiles wrote 256-byte pieces: 255 -> 208 us)
    const int n = blockIdx.z, oy = blockIdx.y, ox = blockIdx.x * 256 + threadIdx.x;
    if (ox >= out_w) return;
    float v[K];
    const float* xb = x + (int64_t)n * CIN * in_h * in_w + (int64_t)(2 * oy) * in_w + 2 * ox;
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] = xb[(int64_t)(k / 9) * in_h * in_w + ((k % 9) / 3) * in_w + (k % 3)];
    float* yb = y + (int64_t)n * cout * out_h * out_w + (int64_t)oy * out_w + ox;
    const int64_t plane = (int64_t)out_h * out_w;
    const nc_cfp ws = (nc_cfp)w, bs = (nc_cfp)bias;
#ifndef MGF_NC_UNROLL
#define MGF_NC_UNROLL 2
#endif
#pragma unroll MGF_NC_UNROLL
    for (int co = 0; co < cout; ++co) {
        const nc_cfp wr = ws + co * K;
        float acc = bias ? bs[co] : 0.f;
#pragma unroll
        for (int k = 0; k < K; ++k) acc += wr[k] * v[k];
        yb[co * plane] = relu ? fmaxf(acc, 0.f) : acc;
    }
}

// t[n, co, 2 i + kh, 2 j + kw] += wp[kh*3+kw][ci][co] * x[n, ci, i, j]  (output [2h+1] x [2w+1], row pitch `pitch`): one lane per 2 x 2
// output quad (2i + {0,1}, 2j + {0,1}), i in [0, h], j in [0, w]:
//   (0,0) <- w00 x[i][j] + w02 x[i][j-1] + w20 x[i-1][j] + w22 x[i-1][j-1]     (0,1) <- w01 x[i][j] + w21 x[i-1][j]
//   (1,0) <- w10 x[i][j] + w12 x[i][j-1]                                       (1,1) <- w11 x[i][j]
// (weights as an LDS image [cin][9][4], read as broadcasts: for THIS kernel -- nine 16-byte reads per channel -- that measured faster
// than nine scalar loads per channel, 294 vs 406 us)
template <int CO>
__global__ __launch_bounds__(256) void tconv3x3s2_few_outputs_kernel(float* __restrict__ y, const float* __restrict__ x, const float* __restrict__ wp,
                                                                     int cin, int h, int w, int cout_pad, int pitch, int64_t y_plane,
                                                                     int64_t y_batch) {
    extern __shared__ float ws[];                        // [cin][9][CO padded to 4]
    const int tid = threadIdx.x;
    for (int i = tid; i < cin * 36; i += 256) {
        const int c = i / 36, r = i - c * 36, t = r >> 2, o = r & 3;
        ws[i] = o < CO ? wp[((int64_t)t * cin + c) * cout_pad + o] : 0.f;
    }
    __syncthreads();
    // 256 x 2 quads per workgroup (1 KB pieces of every input plane), a lane takes quads (i, j) and (i + 1, j): three input rows serve
    // both and the weight image is read once per channel for the two (its broadcast reads are what bounds the kernel)
    const int n = blockIdx.z, i = 2 * blockIdx.y, j = blockIdx.x * 256 + threadIdx.x;
    if (j > w) return;
    const bool c0 = j < w, c1 = j > 0;                                 // x[.][j], x[.][j-1] exist
    const bool ra = i > 0, rb = i < h, rc = i + 1 < h;                 // rows i-1, i, i+1 of x exist
    const bool q1 = i + 1 <= h;                                        // the second quad row exists
    const float* xb = x + (int64_t)n * cin * h * w + (int64_t)i * w + j;
    const int64_t plane = (int64_t)h * w;
    float a00[2][CO], a01[2][CO], a10[2][CO], a11[2][CO];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int o = 0; o < CO; ++o) a00[u][o] = a01[u][o] = a10[u][o] = a11[u][o] = 0.f;
#pragma unroll 2
    for (int c = 0; c < cin; ++c) {
        const float* xc = xb + c * plane;
        float xr[3][2];                                                // rows i-1, i, i+1; columns j, j-1
        xr[0][0] = (ra && c0) ? xc[-w] : 0.f;    xr[0][1] = (ra && c1) ? xc[-w - 1] : 0.f;
        xr[1][0] = (rb && c0) ? xc[0] : 0.f;     xr[1][1] = (rb && c1) ? xc[-1] : 0.f;
        xr[2][0] = (rc && c0) ? xc[w] : 0.f;     xr[2][1] = (rc && c1) ? xc[w - 1] : 0.f;
        const float4* wc = reinterpret_cast<const float4*>(ws + c * 36);
        float wt[9][4];
#pragma unroll
        for (int t = 0; t < 9; ++t) { const float4 f = wc[t]; wt[t][0] = f.x; wt[t][1] = f.y; wt[t][2] = f.z; wt[t][3] = f.w; }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const float p = xr[u + 1][0], q = xr[u + 1][1], r = xr[u][0], s = xr[u][1];      // x[i'][j], x[i'][j-1], x[i'-1][j], x[i'-1][j-1]
#pragma unroll
            for (int o = 0; o < CO; ++o) {
                a00[u][o] += wt[0][o] * p + wt[2][o] * q + wt[6][o] * r + wt[8][o] * s;
                a01[u][o] += wt[1][o] * p + wt[7][o] * r;
                a10[u][o] += wt[3][o] * p + wt[5][o] * q;
                a11[u][o] += wt[4][o] * p;
            }
        }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        if (u == 1 && !q1) break;
        const bool r0 = i + u < h;                                     // the quad's second output row exists
        float* yb = y + (int64_t)n * y_batch + (int64_t)(2 * (i + u)) * pitch + 2 * j;
#pragma unroll
        for (int o = 0; o < CO; ++o) {
            float* yo = yb + o * y_plane;
            if (c0) *reinterpret_cast<float2*>(yo) = make_float2(a00[u][o], a01[u][o]);
            else yo[0] = a00[u][o];
            if (r0) {
                if (c0) *reinterpret_cast<float2*>(yo + pitch) = make_float2(a10[u][o], a11[u][o]);
                else yo[pitch] = a10[u][o];
            }
        }
    }
}

}  // namespace

extern "C" int mgf_conv3x3s2_few_inputs_f32(float* y, const float* x, const float* w, const float* bias, int32_t n, int32_t cin, int32_t in_h,
                                            int32_t in_w, int32_t cout, int32_t relu, mgf_stream_t stream) {
    MGF_REQUIRE(y && x && w && n >= 1 && in_h >= 3 && in_w >= 3, MGF_EINVAL, "conv3x3s2_few_inputs: bad arguments");
    MGF_REQUIRE(cin >= 1 && cin <= NC_MAX_NARROW, MGF_EUNSUPPORTED, "conv3x3s2_few_inputs: 1..%d input channels (got %d)", NC_MAX_NARROW, cin);
    MGF_REQUIRE(cout >= 1, MGF_EINVAL, "conv3x3s2_few_inputs: bad channel count");
    MGF_REQUIRE(n <= 65535, MGF_ETOOBIG, "conv3x3s2_few_inputs: n must be <= 65535");
    const int out_h = (in_h - 3) / 2 + 1, out_w = (in_w - 3) / 2 + 1;
    MGF_REQUIRE((int64_t)cout * out_h * out_w <= INT32_MAX && (int64_t)cin * in_h * in_w <= INT32_MAX, MGF_ETOOBIG, "conv3x3s2_few_inputs: sample too large");
    MGF_REQUIRE(out_h <= 65535, MGF_ETOOBIG, "conv3x3s2_few_inputs: at most 65535 output rows");
    const dim3 grid((unsigned)mgf_cdiv(out_w, 256), (unsigned)out_h, n);
    hipStream_t st = (hipStream_t)stream;
#define MGF_CFI_LAUNCH(CI) \
    hipLaunchKernelGGL(conv3x3s2_few_inputs_kernel<CI>, grid, dim3(256), 0, st, y, x, w, bias, in_h, in_w, cout, out_h, out_w, relu)
    if (cin == 1) MGF_CFI_LAUNCH(1); else if (cin == 2) MGF_CFI_LAUNCH(2); else if (cin == 3) MGF_CFI_LAUNCH(3); else MGF_CFI_LAUNCH(4);
#undef MGF_CFI_LAUNCH
    MGF_CHECK_LAUNCH("conv3x3s2_few_inputs");
    return MGF_OK;
}

extern "C" int mgf_tconv3x3s2_few_outputs_f32(float* y, const float* x, const float* wp, int32_t n, int32_t cin, int32_t h, int32_t w,
                                              int32_t cout, int32_t cout_pad, int32_t pitch, int64_t y_plane, int64_t y_batch,
                                              mgf_stream_t stream) {
    MGF_REQUIRE(y && x && wp && n >= 1 && h >= 1 && w >= 1, MGF_EINVAL, "tconv3x3s2_few_outputs: bad arguments");
    MGF_REQUIRE(cout >= 1 && cout <= NC_MAX_NARROW && cout_pad >= 4, MGF_EUNSUPPORTED, "tconv3x3s2_few_outputs: 1..%d output channels (got %d)",
                NC_MAX_NARROW, cout);
    MGF_REQUIRE(cin >= 1 && cin <= NC_MAX_WIDE, MGF_EUNSUPPORTED, "tconv3x3s2_few_outputs: 1..%d input channels (got %d)", NC_MAX_WIDE, cin);
    MGF_REQUIRE(pitch >= 2 * w + 1 && pitch % 2 == 0 && y_plane % 2 == 0 && y_batch % 2 == 0 && ((uintptr_t)y % 8) == 0 &&
                y_plane >= (int64_t)(2 * h + 1) * pitch && y_batch >= cout * y_plane, MGF_EINVAL,
                "tconv3x3s2_few_outputs: the output needs an even row pitch >= 2w+1, even plane / sample strides and an 8-byte aligned base");
    MGF_REQUIRE(n <= 65535 && (int64_t)cin * h * w <= INT32_MAX, MGF_ETOOBIG, "tconv3x3s2_few_outputs: tensor too large");
    MGF_REQUIRE(h < 65535, MGF_ETOOBIG, "tconv3x3s2_few_outputs: at most 65534 input rows");
    const dim3 grid((unsigned)mgf_cdiv(w + 1, 256), (unsigned)mgf_cdiv(h + 1, 2), n);
    hipStream_t st = (hipStream_t)stream;
#define MGF_TCF_LAUNCH(CO) \
    hipLaunchKernelGGL(tconv3x3s2_few_outputs_kernel<CO>, grid, dim3(256), (size_t)cin * 36 * sizeof(float), st, y, x, wp, cin, h, w, cout_pad, pitch, y_plane, y_batch)
    if (cout == 1) MGF_TCF_LAUNCH(1); else if (cout == 2) MGF_TCF_LAUNCH(2); else if (cout == 3) MGF_TCF_LAUNCH(3); else MGF_TCF_LAUNCH(4);
#undef MGF_TCF_LAUNCH
    MGF_CHECK_LAUNCH("tconv3x3s2_few_outputs");
    return MGF_OK;
}
