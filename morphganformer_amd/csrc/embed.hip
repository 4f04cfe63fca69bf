// Small ops of the biometric branch (IResNet embedder, backbones/iresnet.py): contract in include/mgf.h.
// The 3x3 / 1x1 convolutions of the network run on conv_taps (BatchNorm folded into its out_scale/bias ports, the identity
// shortcut into its residual port); what is left are three bandwidth-bound element-wise / GEMV steps.
#include "mgf_common.h"

namespace {

// y = prelu_c(x * scale_c + shift_c); any of scale / shift / slope may be NULL (1 / 0 / no activation).  x, y: [n, c, hw]
__global__ __launch_bounds__(256) void channel_affine_prelu_kernel(float* y, const float* x, const float* scale, const float* shift,
                                                                   const float* slope, int c, int64_t hw, int64_t total) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int ch = (int)((i / hw) % c);
        float v = x[i];
        if (scale) v = v * scale[ch];
        if (shift) v = v + shift[ch];
        if (slope) v = v > 0.f ? v : v * slope[ch];
        y[i] = v;
    }
}

// y[s, o] = b[o] + sum_i w[o, i] * x[s, i]  -- one workgroup per output feature, all samples at once (n <= 16): the weight row
// is streamed once with float4 loads, the sample vectors are L2 resident.
constexpr int LIN_MAX_N = 16;
__global__ __launch_bounds__(256) void linear_kernel(float* y, const float* x, const float* w, const float* b, int n, int in_f, int out_f) {
    __shared__ float sm[4][LIN_MAX_N];
    const int o = blockIdx.x;
    const float* wr = w + (int64_t)o * in_f;
    float acc[LIN_MAX_N];
#pragma unroll
    for (int s = 0; s < LIN_MAX_N; ++s) acc[s] = 0.f;
    const int nvec = in_f / 4;
    for (int i = threadIdx.x; i < nvec; i += 256) {
        const float4 wv = reinterpret_cast<const float4*>(wr)[i];
#pragma unroll
        for (int s = 0; s < LIN_MAX_N; ++s) {
            if (s < n) {
                const float4 xv = reinterpret_cast<const float4*>(x + (int64_t)s * in_f)[i];
                acc[s] += wv.x * xv.x + wv.y * xv.y + wv.z * xv.z + wv.w * xv.w;
            }
        }
    }
    for (int i = nvec * 4 + threadIdx.x; i < in_f; i += 256) {
#pragma unroll
        for (int s = 0; s < LIN_MAX_N; ++s)
            if (s < n) acc[s] += wr[i] * x[(int64_t)s * in_f + i];
    }
#pragma unroll
    for (int s = 0; s < LIN_MAX_N; ++s) {
        const float v = wave_sum(acc[s]);
        if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6][s] = v;
    }
    __syncthreads();
    if (threadIdx.x < n) y[(int64_t)threadIdx.x * out_f + o] = sm[0][threadIdx.x] + sm[1][threadIdx.x] + sm[2][threadIdx.x] + sm[3][threadIdx.x] + (b ? b[o] : 0.f);
}

// torch.nn.functional.interpolate(x, size=(oh, ow), mode="bilinear", align_corners=False), no antialias
__global__ __launch_bounds__(256) void resize_bilinear_kernel(float* y, const float* x, int nc, int ih, int iw, int oh, int ow, float sy, float sx) {
    const int64_t total = (int64_t)nc * oh * ow;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int ox = (int)(i % ow);
        const int64_t r = i / ow;
        const int oy = (int)(r % oh);
        const int64_t pl = r / oh;
        float fy = ((float)oy + 0.5f) * sy - 0.5f, fx = ((float)ox + 0.5f) * sx - 0.5f;
        fy = fy < 0.f ? 0.f : fy; fx = fx < 0.f ? 0.f : fx;
        const int y0 = (int)fy, x0 = (int)fx;
        const int y1 = y0 + (y0 < ih - 1 ? 1 : 0), x1 = x0 + (x0 < iw - 1 ? 1 : 0);
        const float ly = fy - (float)y0, lx = fx - (float)x0;
        const float hy = 1.f - ly, hx = 1.f - lx;
        const float* xp = x + pl * ih * iw;
        y[i] = hy * (hx * xp[(int64_t)y0 * iw + x0] + lx * xp[(int64_t)y0 * iw + x1]) +
               ly * (hx * xp[(int64_t)y1 * iw + x0] + lx * xp[(int64_t)y1 * iw + x1]);
    }
}

// y[p] = mean over the hw elements of plane p (nn.AdaptiveAvgPool2d(1) of the InceptionResnetV1 head): one wave per plane, fixed order
__global__ __launch_bounds__(256) void spatial_mean_kernel(float* y, const float* x, int nc, int64_t hw) {
    const int pl = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (pl >= nc) return;
    const float* xp = x + (int64_t)pl * hw;
    float a = 0.f;
    for (int64_t i = lane; i < hw; i += 64) a += xp[i];
    a = wave_sum(a);
    if (lane == 0) y[pl] = a / (float)hw;
}

// y[r] = x[r] / max(||x[r]||_2, eps)  (F.normalize(x, p=2, dim=1)): one workgroup per row
__global__ __launch_bounds__(256) void l2_normalize_kernel(float* y, const float* x, int d, float eps) {
    __shared__ float sm[4];
    const float* xr = x + (int64_t)blockIdx.x * d;
    float a = 0.f;
    for (int i = threadIdx.x; i < d; i += 256) a += xr[i] * xr[i];
    a = wave_sum(a);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = a;
    __syncthreads();
    const float nrm = sqrtf(sm[0] + sm[1] + sm[2] + sm[3]);
    const float inv = 1.f / (nrm > eps ? nrm : eps);
    for (int i = threadIdx.x; i < d; i += 256) y[(int64_t)blockIdx.x * d + i] = xr[i] * inv;
}

// ---- gradient mode of the FaceNet head and its concat buffers ----
// y = x / max(||x||, eps):  dx = (dy - y <y, dy>) / max(||x||, eps)   (the clamped branch ||x|| <= eps: dx = dy / eps)
__global__ __launch_bounds__(256) void l2_normalize_bwd_kernel(float* dx, const float* dy, const float* x, int d, float eps) {
    __shared__ float sm[8];
    const float* xr = x + (int64_t)blockIdx.x * d;
    const float* gr = dy + (int64_t)blockIdx.x * d;
    float a = 0.f, b = 0.f;
    for (int i = threadIdx.x; i < d; i += 256) { a += xr[i] * xr[i]; b += xr[i] * gr[i]; }
    a = wave_sum(a); b = wave_sum(b);
    if ((threadIdx.x & 63) == 0) { sm[threadIdx.x >> 6] = a; sm[4 + (threadIdx.x >> 6)] = b; }
    __syncthreads();
    const float nrm = sqrtf(sm[0] + sm[1] + sm[2] + sm[3]), xg = sm[4] + sm[5] + sm[6] + sm[7];
    if (nrm > eps) {
        const float inv = 1.f / nrm, k = xg * inv * inv * inv;             // <y, dy> / ||x|| * (1 / ||x||) folded: dx = dy / r - x <x, dy> / r^3
        for (int i = threadIdx.x; i < d; i += 256) dx[(int64_t)blockIdx.x * d + i] = gr[i] * inv - xr[i] * k;
    } else {
        for (int i = threadIdx.x; i < d; i += 256) dx[(int64_t)blockIdx.x * d + i] = gr[i] / eps;
    }
}

// dx[p, i] = dy[p] / hw  (adjoint of spatial_mean_kernel)
__global__ __launch_bounds__(256) void spatial_mean_bwd_kernel(float* dx, const float* dy, int64_t hw, int64_t total) {
    const float inv = 1.f / (float)hw;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) dx[i] = dy[i / hw] * inv;
}

// dx[n, c, p] = y[n, y_choff + c, p] > 0 ? dy[n, dy_choff + c, p] : 0 -- the ReLU backward of a BRANCH of a concat buffer: dy and y are
// channel slices [choff, choff + c) of tensors with dy_ctot / y_ctot channels, dx is dense [n, c, hw]
__global__ __launch_bounds__(256) void relu_bwd_slice_kernel(float* dx, const float* dy, int dy_ctot, int dy_choff, const float* y, int y_ctot,
                                                             int y_choff, int c, int64_t hw, int64_t total) {
    const int64_t per = (int64_t)c * hw;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t n = i / per, r = i - n * per;
        const float yv = y[(n * y_ctot + y_choff) * hw + r];
        dx[i] = yv > 0.f ? dy[(n * dy_ctot + dy_choff) * hw + r] : 0.f;
    }
}

// ---- gradient mode ----
// dx = dy * (y > 0 ? 1 : slope[c]) from the PReLU OUTPUT y (valid for positive slopes: sign(y) = sign(pre-activation))
__global__ __launch_bounds__(256) void prelu_bwd_kernel(float* dx, const float* dy, const float* y, const float* slope, int c, int64_t hw,
                                                        int64_t total) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int ch = (int)((i / hw) % c);
        dx[i] = dy[i] * (y[i] > 0.f ? 1.f : slope[ch]);
    }
}

// dx[s, i] = sum_o dy[s, o] * w[o, i]: one lane per input feature (coalesced rows of w), all samples at once
__global__ __launch_bounds__(256) void linear_bwd_kernel(float* dx, const float* dy, const float* w, int n, int in_f, int out_f) {
    extern __shared__ float dys[];                 // [n][out_f]
    for (int i = threadIdx.x; i < n * out_f; i += 256) dys[i] = dy[i];
    __syncthreads();
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= in_f) return;
    float acc[LIN_MAX_N];
#pragma unroll
    for (int s = 0; s < LIN_MAX_N; ++s) acc[s] = 0.f;
#pragma unroll 4
    for (int o = 0; o < out_f; ++o) {
        const float wv = w[(int64_t)o * in_f + i];
#pragma unroll
        for (int s = 0; s < LIN_MAX_N; ++s)
            if (s < n) acc[s] += wv * dys[s * out_f + o];
    }
#pragma unroll
    for (int s = 0; s < LIN_MAX_N; ++s)
        if (s < n) dx[(int64_t)s * in_f + i] = acc[s];
}

// adjoint of resize_bilinear_kernel: every output pixel's gradient is scattered to its four source pixels (dx pre-zeroed by the
// caller).  When the map is shrunk by >= 2x no two output pixels share a source pixel, so the atomics never race on an address.
__global__ __launch_bounds__(256) void resize_bilinear_bwd_kernel(float* dx, const float* dy, int nc, int ih, int iw, int oh, int ow, float sy, float sx) {
    const int64_t total = (int64_t)nc * oh * ow;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int ox = (int)(i % ow);
        const int64_t r = i / ow;
        const int oy = (int)(r % oh);
        const int64_t pl = r / oh;
        float fy = ((float)oy + 0.5f) * sy - 0.5f, fx = ((float)ox + 0.5f) * sx - 0.5f;
        fy = fy < 0.f ? 0.f : fy; fx = fx < 0.f ? 0.f : fx;
        const int y0 = (int)fy, x0 = (int)fx;
        const int y1 = y0 + (y0 < ih - 1 ? 1 : 0), x1 = x0 + (x0 < iw - 1 ? 1 : 0);
        const float ly = fy - (float)y0, lx = fx - (float)x0;
        const float hy = 1.f - ly, hx = 1.f - lx;
        float* xp = dx + pl * ih * iw;
        const float g = dy[i];
        atomicAdd(xp + (int64_t)y0 * iw + x0, hy * hx * g);
        atomicAdd(xp + (int64_t)y0 * iw + x1, hy * lx * g);
        atomicAdd(xp + (int64_t)y1 * iw + x0, ly * hx * g);
        atomicAdd(xp + (int64_t)y1 * iw + x1, ly * lx * g);
    }
}

}  // namespace

extern "C" int mgf_spatial_mean_f32(float* y, const float* x, int32_t nc, int64_t hw, mgf_stream_t stream) {
    MGF_REQUIRE(y && x && nc >= 1 && hw >= 1, MGF_EINVAL, "spatial_mean: bad arguments");
    hipLaunchKernelGGL(spatial_mean_kernel, dim3((unsigned)mgf_cdiv(nc, 4)), dim3(256), 0, (hipStream_t)stream, y, x, nc, hw);
    MGF_CHECK_LAUNCH("spatial_mean");
    return MGF_OK;
}

extern "C" int mgf_l2_normalize_f32(float* y, const float* x, int32_t n, int32_t d, float eps, mgf_stream_t stream) {
    MGF_REQUIRE(y && x && n >= 1 && d >= 1 && eps > 0.f, MGF_EINVAL, "l2_normalize: bad arguments");
    hipLaunchKernelGGL(l2_normalize_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, y, x, d, eps);
    MGF_CHECK_LAUNCH("l2_normalize");
    return MGF_OK;
}

extern "C" int mgf_l2_normalize_bwd_f32(float* dx, const float* dy, const float* x, int32_t n, int32_t d, float eps, mgf_stream_t stream) {
    MGF_REQUIRE(dx && dy && x && n >= 1 && d >= 1 && eps > 0.f, MGF_EINVAL, "l2_normalize_bwd: bad arguments");
    hipLaunchKernelGGL(l2_normalize_bwd_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, dx, dy, x, d, eps);
    MGF_CHECK_LAUNCH("l2_normalize_bwd");
    return MGF_OK;
}

extern "C" int mgf_spatial_mean_bwd_f32(float* dx, const float* dy, int32_t nc, int64_t hw, mgf_stream_t stream) {
    MGF_REQUIRE(dx && dy && nc >= 1 && hw >= 1, MGF_EINVAL, "spatial_mean_bwd: bad arguments");
    const int64_t total = (int64_t)nc * hw;
    hipLaunchKernelGGL(spatial_mean_bwd_kernel, dim3(mgf_stream_grid(total, 256, 4)), dim3(256), 0, (hipStream_t)stream, dx, dy, hw, total);
    MGF_CHECK_LAUNCH("spatial_mean_bwd");
    return MGF_OK;
}

extern "C" int mgf_relu_bwd_slice_f32(float* dx, const float* dy, int32_t dy_channels, int32_t dy_choff, const float* y, int32_t y_channels,
                                      int32_t y_choff, int32_t n, int32_t c, int64_t hw, mgf_stream_t stream) {
    MGF_REQUIRE(dx && dy && y && n >= 1 && c >= 1 && hw >= 1, MGF_EINVAL, "relu_bwd_slice: bad arguments");
    MGF_REQUIRE(dy_choff >= 0 && dy_choff + c <= dy_channels && y_choff >= 0 && y_choff + c <= y_channels, MGF_EINVAL,
                "relu_bwd_slice: channel slice [%d, %d) / [%d, %d) outside its tensor (%d / %d channels)", dy_choff, dy_choff + c, y_choff,
                y_choff + c, dy_channels, y_channels);
    const int64_t total = (int64_t)n * c * hw;
    hipLaunchKernelGGL(relu_bwd_slice_kernel, dim3(mgf_stream_grid(total, 256, 4)), dim3(256), 0, (hipStream_t)stream, dx, dy, dy_channels,
                       dy_choff, y, y_channels, y_choff, c, hw, total);
    MGF_CHECK_LAUNCH("relu_bwd_slice");
    return MGF_OK;
}

extern "C" int mgf_prelu_bwd_f32(float* dx, const float* dy, const float* y, const float* slope, int32_t n, int32_t c, int64_t hw,
                                 mgf_stream_t stream) {
    MGF_REQUIRE(dx && dy && y && slope && n >= 1 && c >= 1 && hw >= 1, MGF_EINVAL, "prelu_bwd: bad arguments");
    const int64_t total = (int64_t)n * c * hw;
    hipLaunchKernelGGL(prelu_bwd_kernel, dim3(mgf_stream_grid(total, 256, 4)), dim3(256), 0, (hipStream_t)stream, dx, dy, y, slope, c, hw, total);
    MGF_CHECK_LAUNCH("prelu_bwd");
    return MGF_OK;
}

extern "C" int mgf_linear_bwd_f32(float* dx, const float* dy, const float* w, int32_t n, int32_t in_features, int32_t out_features,
                                  mgf_stream_t stream) {
    MGF_REQUIRE(dx && dy && w && n >= 1 && in_features >= 1 && out_features >= 1, MGF_EINVAL, "linear_bwd: bad arguments");
    MGF_REQUIRE(n <= LIN_MAX_N, MGF_EUNSUPPORTED, "linear_bwd: at most %d rows per call (got %d)", LIN_MAX_N, n);
    const size_t lds = (size_t)n * out_features * sizeof(float);
    MGF_REQUIRE(lds <= 64 * 1024, MGF_EUNSUPPORTED, "linear_bwd: n * out_features = %d floats exceed 64 KiB of LDS", n * out_features);
    hipLaunchKernelGGL(linear_bwd_kernel, dim3((unsigned)mgf_cdiv(in_features, 256)), dim3(256), lds, (hipStream_t)stream, dx, dy, w, n,
                       in_features, out_features);
    MGF_CHECK_LAUNCH("linear_bwd");
    return MGF_OK;
}

extern "C" int mgf_resize_bilinear_bwd_f32(float* dx, const float* dy, int32_t nc, int32_t in_h, int32_t in_w, int32_t out_h, int32_t out_w,
                                           mgf_stream_t stream) {
    MGF_REQUIRE(dx && dy && nc >= 1 && in_h >= 1 && in_w >= 1 && out_h >= 1 && out_w >= 1, MGF_EINVAL, "resize_bilinear_bwd: bad arguments");
    const int64_t total = (int64_t)nc * out_h * out_w;
    hipLaunchKernelGGL(resize_bilinear_bwd_kernel, dim3(mgf_stream_grid(total, 256, 1)), dim3(256), 0, (hipStream_t)stream, dx, dy, nc, in_h,
                       in_w, out_h, out_w, (float)in_h / (float)out_h, (float)in_w / (float)out_w);
    MGF_CHECK_LAUNCH("resize_bilinear_bwd");
    return MGF_OK;
}

extern "C" int mgf_channel_affine_prelu_f32(float* y, const float* x, const float* scale, const float* shift, const float* slope,
                                            int32_t n, int32_t c, int64_t hw, mgf_stream_t stream) {
    MGF_REQUIRE(y && x && n >= 1 && c >= 1 && hw >= 1, MGF_EINVAL, "channel_affine_prelu: bad arguments");
    const int64_t total = (int64_t)n * c * hw;
    hipLaunchKernelGGL(channel_affine_prelu_kernel, dim3(mgf_stream_grid(total, 256, 4)), dim3(256), 0, (hipStream_t)stream, y, x, scale,
                       shift, slope, c, hw, total);
    MGF_CHECK_LAUNCH("channel_affine_prelu");
    return MGF_OK;
}

extern "C" int mgf_linear_f32(float* y, const float* x, const float* w, const float* b, int32_t n, int32_t in_features,
                              int32_t out_features, mgf_stream_t stream) {
    MGF_REQUIRE(y && x && w && n >= 1 && in_features >= 1 && out_features >= 1, MGF_EINVAL, "linear: bad arguments");
    MGF_REQUIRE(n <= LIN_MAX_N, MGF_EUNSUPPORTED, "linear: at most %d rows per call (got %d)", LIN_MAX_N, n);
    MGF_REQUIRE(in_features % 4 == 0, MGF_EUNSUPPORTED, "linear: in_features must be a multiple of 4 (got %d)", in_features);
    MGF_REQUIRE((((uintptr_t)x | (uintptr_t)w) % 16) == 0, MGF_EINVAL, "linear: x and w must be 16-byte aligned");
    hipLaunchKernelGGL(linear_kernel, dim3(out_features), dim3(256), 0, (hipStream_t)stream, y, x, w, b, n, in_features, out_features);
    MGF_CHECK_LAUNCH("linear");
    return MGF_OK;
}

extern "C" int mgf_resize_bilinear_f32(float* y, const float* x, int32_t nc, int32_t in_h, int32_t in_w, int32_t out_h, int32_t out_w,
                                       mgf_stream_t stream) {
    MGF_REQUIRE(y && x && nc >= 1 && in_h >= 1 && in_w >= 1 && out_h >= 1 && out_w >= 1, MGF_EINVAL, "resize_bilinear: bad arguments");
    const int64_t total = (int64_t)nc * out_h * out_w;
    hipLaunchKernelGGL(resize_bilinear_kernel, dim3(mgf_stream_grid(total, 256, 2)), dim3(256), 0, (hipStream_t)stream, y, x, nc, in_h,
                       in_w, out_h, out_w, (float)in_h / (float)out_h, (float)in_w / (float)out_w);
    MGF_CHECK_LAUNCH("resize_bilinear");
    return MGF_OK;
}
