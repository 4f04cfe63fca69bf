// upfirdn2d for gfx950: pad -> zero-insert upsample -> 2-D FIR -> decimate, with an optional fused
// noise/bias/activation/residual epilogue.  Contract: include/mgf.h (mgf_upfirdn2d); reference semantics:
// torch_utils/ops/upfirdn2d.py:161-200 and upfirdn2d.cu:21-333.
//
// Two kernels:
//  * upfirdn_tiled_f32<UP>: the hot-path shapes (dense NCHW fp32, <=4x4 filter, up in {1,2}, down 1).  A workgroup
//    stages the input footprint of a 64x16 output tile in LDS (coalesced row reads), every lane produces a 1x4
//    output column strip from LDS with the polyphase taps unrolled, rows are written as 256-byte segments.
//  * upfirdn_generic<T>: everything else (any strides incl. channels_last, f16/f64, any up/down/filter): one output
//    per lane, polyphase tap skipping, filter in LDS.
#include "mgf_common.h"
#include <hip/hip_fp16.h>

namespace {

struct UFParams {
    void* y;
    const void* x;
    const float* f;
    int n, c, in_h, in_w;
    int64_t sn, sc, sh, sw;
    int out_h, out_w;
    int64_t yn, yc, yh, yw;
    int fh, fw, upx, upy, downx, downy, padx0, pady0, flip;
    float gain;
    mgf_epilogue ep;
    int has_ep;
    int sep_ok;        // the (device) filter is known to be an outer product fy (x) fx
    int vec_in;        // fir_down2_tiled: rows of x can be read with 16-byte loads (in_w, strides multiples of 4, base aligned)
};

__device__ __forceinline__ float apply_epilogue(const mgf_epilogue& ep, float v, int n, int c, int oy, int ox, int out_h,
                                                int out_w, int64_t yoff) {
    if (ep.noise) {
        float ns = ep.noise_strength ? *ep.noise_strength : 1.0f;
        int nn = ep.noise_n > 1 ? n : 0;
        v += ep.noise[((int64_t)nn * out_h + oy) * out_w + ox] * ns;
    }
    if (ep.bias) v += ep.bias[c];
    if (ep.act == MGF_ACT_LRELU) v = v > 0.f ? v : v * ep.alpha;
    else if (ep.act == MGF_ACT_RELU) v = v > 0.f ? v : 0.f;
    v *= ep.gain;
    if (ep.residual) v += ep.residual[yoff];
    return v;
}

template <typename T> __device__ __forceinline__ double ld(const T* p) { return (double)*p; }
template <> __device__ __forceinline__ double ld<__half>(const __half* p) { return (double)__half2float(*p); }
template <typename T> __device__ __forceinline__ void st(T* p, double v) { *p = (T)v; }
template <> __device__ __forceinline__ void st<__half>(__half* p, double v) { *p = __float2half((float)v); }

// ACC = float for f16/f32, double for f64.
template <typename T, typename ACC>
__global__ __launch_bounds__(256) void upfirdn_generic(UFParams p) {
    extern __shared__ float sfilt[];
    const int ntaps = p.fh * p.fw;
    for (int i = threadIdx.x; i < ntaps; i += blockDim.x) {
        int fy = i / p.fw, fx = i - fy * p.fw;
        // store the filter so that window offset j multiplies sfilt[j]: true convolution flips the taps
        int ky = p.flip ? fy : p.fh - 1 - fy, kx = p.flip ? fx : p.fw - 1 - fx;
        sfilt[i] = p.f[ky * p.fw + kx] * p.gain;
    }
    __syncthreads();
    const int64_t total = (int64_t)p.n * p.c * p.out_h * p.out_w;
    const T* X = (const T*)p.x;
    T* Y = (T*)p.y;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        int ox = (int)(idx % p.out_w);
        int64_t r = idx / p.out_w;
        int oy = (int)(r % p.out_h); r /= p.out_h;
        int c = (int)(r % p.c);
        int n = (int)(r / p.c);
        // window start in upsampled coordinates
        int uy0 = oy * p.downy - p.pady0, ux0 = ox * p.downx - p.padx0;
        // first window offset that lands on a real sample (u % up == 0)
        int jy0 = ((-uy0) % p.upy + p.upy) % p.upy;
        int jx0 = ((-ux0) % p.upx + p.upx) % p.upx;
        const T* xb = X + (int64_t)n * p.sn + (int64_t)c * p.sc;
        ACC acc = 0;
        for (int jy = jy0; jy < p.fh; jy += p.upy) {
            int iy = (uy0 + jy) / p.upy;
            if (iy < 0 || iy >= p.in_h) continue;
            for (int jx = jx0; jx < p.fw; jx += p.upx) {
                int ix = (ux0 + jx) / p.upx;
                if (ix < 0 || ix >= p.in_w) continue;
                acc += (ACC)ld<T>(xb + (int64_t)iy * p.sh + (int64_t)ix * p.sw) * (ACC)sfilt[jy * p.fw + jx];
            }
        }
        int64_t yoff = (int64_t)n * p.yn + (int64_t)c * p.yc + (int64_t)oy * p.yh + (int64_t)ox * p.yw;
        if (p.has_ep) acc = (ACC)apply_epilogue(p.ep, (float)acc, n, c, oy, ox, p.out_h, p.out_w, yoff);
        st<T>(Y + yoff, (double)acc);
    }
}


// Small maps (the 4^2 .. 16^2 layers of one image: gradient mode at one target), 4x4 filter, up and down in {1, 2}: the same sums in the
// same order as upfirdn_generic, with the tap loops unrolled, 32-bit indices and one output per lane -- a few thousand outputs are a
// launch-latency problem, and the generic kernel's run-time loops and 64-bit divisions made it 17 - 20 us where this takes ~5.
template <int UP, int DOWN>
__global__ __launch_bounds__(256) void upfirdn_small4(UFParams p) {
    __shared__ float sfilt[16];
    if (threadIdx.x < 16) {
        const int fy = threadIdx.x >> 2, fx = threadIdx.x & 3;
        const int ky = p.flip ? fy : 3 - fy, kx = p.flip ? fx : 3 - fx;
        sfilt[threadIdx.x] = p.f[ky * 4 + kx] * p.gain;
    }
    __syncthreads();
    const unsigned total = (unsigned)p.n * p.c * p.out_h * p.out_w;
    const unsigned idx = blockIdx.x * 256u + threadIdx.x;
    if (idx >= total) return;
    const int ox = (int)(idx % (unsigned)p.out_w);
    unsigned r = idx / (unsigned)p.out_w;
    const int oy = (int)(r % (unsigned)p.out_h); r /= (unsigned)p.out_h;
    const int c = (int)(r % (unsigned)p.c), n = (int)(r / (unsigned)p.c);
    const int uy0 = oy * DOWN - p.pady0, ux0 = ox * DOWN - p.padx0;
    const int jy0 = UP == 1 ? 0 : (uy0 & 1), jx0 = UP == 1 ? 0 : (ux0 & 1);       // first window offset on a real sample
    const float* xb = (const float*)p.x + (int64_t)n * p.sn + (int64_t)c * p.sc;
    float acc = 0.f;
#pragma unroll
    for (int a = 0; a < 4 / UP; ++a) {
        const int jy = jy0 + a * UP;
        const int iy = UP == 1 ? uy0 + jy : (uy0 + jy) >> 1;                      // (uy0 + jy is even: exact also when negative)
        if (iy < 0 || iy >= p.in_h) continue;
#pragma unroll
        for (int b = 0; b < 4 / UP; ++b) {
            const int jx = jx0 + b * UP;
            const int ix = UP == 1 ? ux0 + jx : (ux0 + jx) >> 1;
            if (ix < 0 || ix >= p.in_w) continue;
            acc += xb[(int64_t)iy * p.sh + (int64_t)ix * p.sw] * sfilt[jy * 4 + jx];
        }
    }
    const int64_t yoff = (int64_t)n * p.yn + (int64_t)c * p.yc + (int64_t)oy * p.yh + (int64_t)ox * p.yw;
    if (p.has_ep) acc = apply_epilogue(p.ep, acc, n, c, oy, ox, p.out_h, p.out_w, yoff);
    ((float*)p.y)[yoff] = acc;
}

// down = 2, up = 1, filter <= 4x4, dense rows (the gradient of the 2x FIR upsampling of the resnet skip branch, gradient mode).
// Output tile 64 (x) x 16 (y) per workgroup; its (2*64+2) x (2*16+2) input footprint goes through LDS once, lane (lx, ly) then
// produces outputs (oy0 + 4*ly + {0..3}, ox0 + lx) from 16 LDS reads each.
__global__ __launch_bounds__(256) void fir_down2_tiled(UFParams p) {
    constexpr int TW = 64, TH = 16, FMAX = 4;
    constexpr int IW = 2 * (TW - 1) + FMAX, IH = 2 * (TH - 1) + FMAX;
    __shared__ float sx[IH][IW + 1];
    __shared__ float sf[FMAX][FMAX];
    const int tid = threadIdx.x;
    if (tid < FMAX * FMAX) {
        const int jy = tid / FMAX, jx = tid % FMAX;
        float v = 0.f;
        if (jy < p.fh && jx < p.fw) {
            const int ky = p.flip ? jy : p.fh - 1 - jy, kx = p.flip ? jx : p.fw - 1 - jx;
            v = p.f[ky * p.fw + kx] * p.gain;
        }
        sf[jy][jx] = v;
    }
    const int tiles_x = (p.out_w + TW - 1) / TW, tiles_y = (p.out_h + TH - 1) / TH;
    int b = blockIdx.x;
    const int tx = b % tiles_x; b /= tiles_x;
    const int ty = b % tiles_y; b /= tiles_y;
    const int c = b % p.c, n = b / p.c;
    const int ox0 = tx * TW, oy0 = ty * TH;
    const int ix0 = 2 * ox0 - p.padx0, iy0 = 2 * oy0 - p.pady0;
    const float* xb = (const float*)p.x + (int64_t)n * p.sn + (int64_t)c * p.sc;
    // the patch by rows: 16-byte loads from the 4-column boundary at or before the patch's first column when the rows allow it (the
    // 4-byte form streamed at 3.1 TB/s, the up-sampling kernels with 16-byte loads at 4.2), else 128 columns x 2 rows per sweep
    if (p.vec_in) {
        const int off = ((ix0 % 4) + 4) % 4, xa = ix0 - off;         // xa % 4 == 0 (also for negative ix0)
        constexpr int NV = (IW + 3 + 3) / 4;                         // float4 per row: covers off + IW columns
        for (int i = tid; i < IH * NV; i += 256) {
            const int r = i / NV, v4 = i - r * NV;
            const int iy = iy0 + r, ix = xa + 4 * v4;
            float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
            if (iy >= 0 && iy < p.in_h && ix >= 0 && ix < p.in_w) q = *reinterpret_cast<const float4*>(xb + (int64_t)iy * p.sh + ix);   // in_w % 4 == 0
            const int cc = 4 * v4 - off;
            const float e[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (cc + k >= 0 && cc + k < IW) sx[r][cc + k] = e[k];
        }
    } else {
        static_assert(IW >= 128 && (IW - 128) * IH <= 256, "patch sweep assumes 128 < IW <= 128 + 256 / IH");
        const int cc = tid & 127, ix = ix0 + cc;
        const bool xin = ix >= 0 && ix < p.in_w;
#pragma unroll
        for (int r = tid >> 7; r < IH; r += 2) {
            const int iy = iy0 + r;
            sx[r][cc] = (xin && iy >= 0 && iy < p.in_h) ? xb[(int64_t)iy * p.sh + ix] : 0.f;
        }
        if (tid < (IW - 128) * IH) {
            const int r = tid / (IW - 128), c2 = 128 + tid - r * (IW - 128);
            const int iy = iy0 + r, ix2 = ix0 + c2;
            sx[r][c2] = (iy >= 0 && iy < p.in_h && ix2 >= 0 && ix2 < p.in_w) ? xb[(int64_t)iy * p.sh + ix2] : 0.f;
        }
    }
    __syncthreads();
    const int lx = tid & 63, ly = tid >> 6;
    float* yb = (float*)p.y + (int64_t)n * p.yn + (int64_t)c * p.yc;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int wy = 4 * ly + q;
        const int oy = oy0 + wy, ox = ox0 + lx;
        float acc = 0.f;
#pragma unroll
        for (int jy = 0; jy < FMAX; ++jy)
#pragma unroll
            for (int jx = 0; jx < FMAX; ++jx) acc += sx[2 * wy + jy][2 * lx + jx] * sf[jy][jx];
        if (oy < p.out_h && ox < p.out_w) yb[(int64_t)oy * p.yh + ox] = acc;
    }
}

// Hot-path kernel.  Output tile 64 (x) x 16 (y) per workgroup of 256 lanes; lane (lx = tid & 63, ly = tid >> 6)
// produces outputs (oy0 + 4*ly + {0..3}, ox0 + lx).
template <int UP>
__global__ __launch_bounds__(256) void upfirdn_tiled_f32(UFParams p) {
    constexpr int TW = 64, TH = 16, FMAX = 4;
    // input footprint of the tile
    constexpr int IW = (UP == 1) ? TW + FMAX - 1 : TW / 2 + FMAX / 2 + 1;
    constexpr int IH = (UP == 1) ? TH + FMAX - 1 : TH / 2 + FMAX / 2 + 1;
    constexpr int IWP = IW + 1;
    __shared__ float sx[IH][IWP];
    __shared__ float sf[FMAX][FMAX];
    const int tid = threadIdx.x;
    if (tid < FMAX * FMAX) {
        int jy = tid / FMAX, jx = tid % FMAX;
        float v = 0.f;
        if (jy < p.fh && jx < p.fw) {
            int ky = p.flip ? jy : p.fh - 1 - jy, kx = p.flip ? jx : p.fw - 1 - jx;
            v = p.f[ky * p.fw + kx] * p.gain;
        }
        sf[jy][jx] = v;
    }
    const int tiles_x = (p.out_w + TW - 1) / TW;
    const int tiles_y = (p.out_h + TH - 1) / TH;
    const int tile = blockIdx.x % (tiles_x * tiles_y);
    const int plane = blockIdx.x / (tiles_x * tiles_y);          // n * c + c
    const int ox0 = (tile % tiles_x) * TW, oy0 = (tile / tiles_x) * TH;
    const int n = plane / p.c, c = plane - n * p.c;
    const float* xb = (const float*)p.x + (int64_t)n * p.sn + (int64_t)c * p.sc;
    // first input sample touched by the tile (floor division, may be negative)
    const int uy0 = oy0 - p.pady0, ux0 = ox0 - p.padx0;
    // UP==2: ceil(u0/2) for either sign (C++ division truncates, i.e. rounds negatives up)
    const int iy0 = (UP == 1) ? uy0 : (uy0 >= 0 ? uy0 + UP - 1 : uy0) / UP;
    const int ix0 = (UP == 1) ? ux0 : (ux0 >= 0 ? ux0 + UP - 1 : ux0) / UP;
    for (int i = tid; i < IH * IW; i += 256) {
        int r = i / IW, q = i - r * IW;
        int iy = iy0 + r, ix = ix0 + q;
        float v = 0.f;
        if (iy >= 0 && iy < p.in_h && ix >= 0 && ix < p.in_w) v = xb[(int64_t)iy * p.sh + ix];
        sx[r][q] = v;
    }
    __syncthreads();
    const int lx = tid & 63, ly = tid >> 6;
    const int ox = ox0 + lx;
    float* Y = (float*)p.y;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int oy = oy0 + ly * 4 + k;
        float acc = 0.f;
        const int uy = oy - p.pady0, ux = ox - p.padx0;
        if (UP == 1) {
            const int ry = uy - iy0, rx = ux - ix0;
#pragma unroll
            for (int jy = 0; jy < FMAX; ++jy)
#pragma unroll
                for (int jx = 0; jx < FMAX; ++jx) acc += sx[ry + jy][rx + jx] * sf[jy][jx];
        } else {
            // taps jy with (uy + jy) even; input row (uy + jy) / 2
            const int py = uy & 1, px = ux & 1;   // parity (two's complement & works for negatives)
#pragma unroll
            for (int a = 0; a < FMAX / 2; ++a) {
                const int jy = py + 2 * a;
                const int ry = ((uy + jy) >> 1) - iy0;
#pragma unroll
                for (int b = 0; b < FMAX / 2; ++b) {
                    const int jx = px + 2 * b;
                    const int rx = ((ux + jx) >> 1) - ix0;
                    acc += sx[ry][rx] * sf[jy][jx];
                }
            }
        }
        if (oy < p.out_h && ox < p.out_w) {
            int64_t yoff = (int64_t)n * p.yn + (int64_t)c * p.yc + (int64_t)oy * p.yh + ox;
            if (p.has_ep) acc = apply_epilogue(p.ep, acc, n, c, oy, ox, p.out_h, p.out_w, yoff);
            Y[yoff] = acc;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Wide kernels for the two hot shapes of the synthesis network (4x4 filter, down 1, dense NCHW fp32, 16-byte aligned rows):
//   fir_up1_wide : the blur after the stride-2 transposed conv (up 1, pad x0 = y0 = 1), optional fused epilogue
//   fir_up2_wide : the skip-path upsample (up 2, pad x0 = y0 = 2)
// Output tile 64 x 16 per workgroup; a lane produces a 1x4 strip and stores it as one float4; the input footprint is
// staged with ALIGNED float4 loads (window starts 4 columns left of the tile) -- 3.4x fewer, 4x wider memory instructions
// than the scalar tiled kernel, which is what this bandwidth-bound op is limited by.
constexpr int FIR_SUB = 4;       // vertically adjacent tiles per workgroup of the wide FIR kernels

template <bool EP>
__global__ __launch_bounds__(256) void fir_up1_wide(UFParams p) {
    constexpr int TW = 64, TH = 16, WV = 18, IH = TH + 3;          // window: 18 float4 = 72 columns, 19 rows
    __shared__ float4 sx[IH][WV + 1];
    __shared__ float sf[4][4];
    const int tid = threadIdx.x;
    if (tid < 16) {
        const int jy = tid >> 2, jx = tid & 3;
        float v = 0.f;
        if (jy < p.fh && jx < p.fw) {
            const int ky = p.flip ? jy : p.fh - 1 - jy, kx = p.flip ? jx : p.fw - 1 - jx;
            v = p.f[ky * p.fw + kx] * p.gain;
        }
        sf[jy][jx] = v;
    }
    // a workgroup walks FIR_SUB vertically adjacent 64x16 tiles: 4x fewer, longer-lived workgroups (filter set-up and index
    // arithmetic amortised; +1% on the whole iteration)
    const int tiles_x = p.out_w / TW, tiles_y = (p.out_h + TH * FIR_SUB - 1) / (TH * FIR_SUB);
    const int tile = blockIdx.x % (tiles_x * tiles_y), plane = blockIdx.x / (tiles_x * tiles_y);
    const int ox0 = (tile % tiles_x) * TW;
    const int n = plane / p.c, c = plane - n * p.c;
    const float* xb = (const float*)p.x + (int64_t)n * p.sn + (int64_t)c * p.sc;
    for (int sub_t = 0; sub_t < FIR_SUB; ++sub_t) {
    const int oy0 = ((tile / tiles_x) * FIR_SUB + sub_t) * TH;
    if (oy0 >= p.out_h) break;
    if (sub_t) __syncthreads();                                    // the previous sub-tile's window reads are done
    const int xa = ox0 - 4, iy0 = oy0 - p.pady0;                  // window origin (input coordinates)
    for (int i = tid; i < IH * WV; i += 256) {
        const int r = i / WV, v4 = i - r * WV;
        const int iy = iy0 + r, ix = xa + 4 * v4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (iy >= 0 && iy < p.in_h && ix >= 0 && ix < p.in_w) {
            v = *reinterpret_cast<const float4*>(xb + (int64_t)iy * p.sh + ix);      // row pitch is padded to a multiple of 4
            if (ix + 1 >= p.in_w) v.y = 0.f;
            if (ix + 2 >= p.in_w) v.z = 0.f;
            if (ix + 3 >= p.in_w) v.w = 0.f;
        }
        sx[r][v4] = v;
    }
    __syncthreads();
    const int lx = tid & 15, ly = tid >> 4;
    const int oy = oy0 + ly;
    // window column of output (ox0 + 4*lx + e), tap jx:  4 + 4*lx + e - padx0 + jx ; padx0 == 1 -> 4*lx + 3 + e + jx
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int jy = 0; jy < 4; ++jy) {
        const float4 a = sx[ly + jy][lx], b = sx[ly + jy][lx + 1], cc = sx[ly + jy][lx + 2];
        const float w[12] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, cc.x, cc.y, cc.z, cc.w};
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int jx = 0; jx < 4; ++jx) acc[e] += w[3 + e + jx] * sf[jy][jx];
    }
    if (oy < p.out_h) {
        const int ox = ox0 + 4 * lx;
        const int64_t yoff = (int64_t)n * p.yn + (int64_t)c * p.yc + (int64_t)oy * p.yh + ox;
        if (EP) {
            float nz[4] = {0.f, 0.f, 0.f, 0.f};
            if (p.ep.noise) {
                const float ns = p.ep.noise_strength ? *p.ep.noise_strength : 1.0f;
                const float4 nv = *reinterpret_cast<const float4*>(p.ep.noise + ((int64_t)(p.ep.noise_n > 1 ? n : 0) * p.out_h + oy) * p.out_w + ox);
                nz[0] = nv.x * ns; nz[1] = nv.y * ns; nz[2] = nv.z * ns; nz[3] = nv.w * ns;
            }
            const float bb = p.ep.bias ? p.ep.bias[c] : 0.f;
            float rr[4] = {0.f, 0.f, 0.f, 0.f};
            if (p.ep.residual) {
                const float4 rv = *reinterpret_cast<const float4*>(p.ep.residual + yoff);
                rr[0] = rv.x; rr[1] = rv.y; rr[2] = rv.z; rr[3] = rv.w;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float v = acc[e] + nz[e];
                v += bb;
                if (p.ep.act == MGF_ACT_LRELU) v = v > 0.f ? v : v * p.ep.alpha;
                else if (p.ep.act == MGF_ACT_RELU) v = v > 0.f ? v : 0.f;
                acc[e] = v * p.ep.gain + rr[e];
            }
        }
        *reinterpret_cast<float4*>((float*)p.y + yoff) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    }
    }
}

// Separable form of the same blur (f = fy (x) fx, true for every filter upfirdn2d.setup_filter builds from a 1-D tap list): every
// thread produces a 4x4 output patch from a 7-row x 12-column register window -- horizontal pass on the 7 rows (4 taps), vertical
// pass on the 4x4 patch (4 taps): 11 FMAs and 1.3 LDS reads per output instead of 16 and 3.  Workgroup = 64x64 outputs, window of
// 67 rows x 18 float4 staged once (read amplification 1.18 instead of 1.33).
template <bool EP, int PADX = 1, bool RAGGED = false>
__global__ __launch_bounds__(256) void fir_up1_sep(UFParams p) {
    constexpr int T = 64, WV = 18, IH = T + 3;
    __shared__ float4 sx[IH][WV + 1];
    __shared__ float sfx[4], sfy[4];
    const int tid = threadIdx.x;
    if (tid < 4) {                                                   // f[jy][jx] = fy[jy] * fx[jx]; gain goes with fy
        const int k = p.flip ? tid : 3 - tid;
        const float f00 = p.f[0];
        sfx[tid] = p.f[k] / f00;                                     // row 0 normalised
        sfy[tid] = p.f[k * p.fw] * p.gain;                           // column 0 (carries f00)
    }
    // PADX = 2 / RAGGED: the blur's own gradient (pad 2 on every side, output one wider than the input and of any width: rows are not
    // 16-byte aligned, so the last stage stores element-wise) -- the backward pass of gradient mode ran on the generic tiled kernel
    const int tiles_x = RAGGED ? (p.out_w + T - 1) / T : p.out_w / T, tiles_y = (p.out_h + T - 1) / T;
    const int tile = blockIdx.x % (tiles_x * tiles_y), plane = blockIdx.x / (tiles_x * tiles_y);
    const int ox0 = (tile % tiles_x) * T, oy0 = (tile / tiles_x) * T;
    const int n = plane / p.c, c = plane - n * p.c;
    const float* xb = (const float*)p.x + (int64_t)n * p.sn + (int64_t)c * p.sc;
    const int xa = ox0 - 4, iy0 = oy0 - p.pady0;
    for (int i = tid; i < IH * WV; i += 256) {
        const int r = i / WV, v4 = i - r * WV;
        const int iy = iy0 + r, ix = xa + 4 * v4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (iy >= 0 && iy < p.in_h && ix >= 0 && ix < p.in_w) {
            v = *reinterpret_cast<const float4*>(xb + (int64_t)iy * p.sh + ix);
            if (ix + 1 >= p.in_w) v.y = 0.f;
            if (ix + 2 >= p.in_w) v.z = 0.f;
            if (ix + 3 >= p.in_w) v.w = 0.f;
        }
        sx[r][v4] = v;
    }
    __syncthreads();
    const int lx = tid & 15, ly = tid >> 4;
    const float fx0 = sfx[0], fx1 = sfx[1], fx2 = sfx[2], fx3 = sfx[3];
    float hz[7][4];                                                  // horizontal pass: rows 4*ly .. 4*ly+6, outputs e = 0..3
#pragma unroll
    for (int r = 0; r < 7; ++r) {
        const float4 a = sx[4 * ly + r][lx], b = sx[4 * ly + r][lx + 1], cc = sx[4 * ly + r][lx + 2];
        const float w[12] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, cc.x, cc.y, cc.z, cc.w};
#pragma unroll
        for (int e = 0; e < 4; ++e)                                  // input column of tap k: ox + e - PADX + k = window slot 4 - PADX + e + k
            hz[r][e] = w[4 - PADX + e] * fx0 + w[5 - PADX + e] * fx1 + w[6 - PADX + e] * fx2 + w[7 - PADX + e] * fx3;
    }
    const float fy0 = sfy[0], fy1 = sfy[1], fy2 = sfy[2], fy3 = sfy[3];
    const float ns = (EP && p.ep.noise && p.ep.noise_strength) ? *p.ep.noise_strength : 1.0f;
    const float bb = (EP && p.ep.bias) ? p.ep.bias[c] : 0.f;
    const int ox = ox0 + 4 * lx;
    if (RAGGED) {
        // rows of any width start at any alignment: four scalar stores per lane at a 16-byte lane stride reach 3.1 TB/s; the tile goes
        // back through LDS (over the input window, no longer needed) and leaves as full 256-byte row segments instead
        constexpr int RS = T + 4;
        float* so = reinterpret_cast<float*>(&sx[0][0]);
        static_assert(T * RS <= IH * (WV + 1) * 4, "output tile must fit in the input window's LDS");
        __syncthreads();                                             // every lane has read its window rows
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            float4 o;
            o.x = hz[a][0] * fy0 + hz[a + 1][0] * fy1 + hz[a + 2][0] * fy2 + hz[a + 3][0] * fy3;
            o.y = hz[a][1] * fy0 + hz[a + 1][1] * fy1 + hz[a + 2][1] * fy2 + hz[a + 3][1] * fy3;
            o.z = hz[a][2] * fy0 + hz[a + 1][2] * fy1 + hz[a + 2][2] * fy2 + hz[a + 3][2] * fy3;
            o.w = hz[a][3] * fy0 + hz[a + 1][3] * fy1 + hz[a + 2][3] * fy2 + hz[a + 3][3] * fy3;
            *reinterpret_cast<float4*>(so + (4 * ly + a) * RS + 4 * lx) = o;
        }
        __syncthreads();
        const int col = tid & 63, r0 = tid >> 6;
        if (ox0 + col < p.out_w) {
            float* yb = (float*)p.y + (int64_t)n * p.yn + (int64_t)c * p.yc + ox0 + col;
#pragma unroll
            for (int k = 0; k < T / 4; ++k) {
                const int row = r0 + 4 * k;
                if (oy0 + row < p.out_h) yb[(int64_t)(oy0 + row) * p.yh] = so[row * RS + col];
            }
        }
        return;
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int oy = oy0 + 4 * ly + a;
        if (oy >= p.out_h) break;
        const int64_t yoff = (int64_t)n * p.yn + (int64_t)c * p.yc + (int64_t)oy * p.yh + ox;
        float acc[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] = hz[a][e] * fy0 + hz[a + 1][e] * fy1 + hz[a + 2][e] * fy2 + hz[a + 3][e] * fy3;
        if (EP) {
            float4 nv = make_float4(0.f, 0.f, 0.f, 0.f), rv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (p.ep.noise) nv = *reinterpret_cast<const float4*>(p.ep.noise + ((int64_t)(p.ep.noise_n > 1 ? n : 0) * p.out_h + oy) * p.out_w + ox);
            if (p.ep.residual) rv = *reinterpret_cast<const float4*>(p.ep.residual + yoff);
            const float nz[4] = {nv.x * ns, nv.y * ns, nv.z * ns, nv.w * ns};
            const float rr[4] = {rv.x, rv.y, rv.z, rv.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float v = acc[e] + nz[e];
                v += bb;
                if (p.ep.act == MGF_ACT_LRELU) v = v > 0.f ? v : v * p.ep.alpha;
                else if (p.ep.act == MGF_ACT_RELU) v = v > 0.f ? v : 0.f;
                acc[e] = v * p.ep.gain + rr[e];
            }
        }
        *reinterpret_cast<float4*>((float*)p.y + yoff) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    }
}

// Streaming form of the same separable blur (+ fused noise / bias / activation) for wide maps: NO LDS and no barrier.  A wave owns a strip
// of 256 output columns (lane = 4 consecutive outputs = one 16-byte load, one 16-byte store per row) and walks FS_ROWS output rows top
// to bottom: the horizontal 4-tap pass needs x[ox - 1 .. ox + 5], i.e. the lane's own float4, the last element of its left neighbour and
// the first two of its right neighbour -- three DPP wave shifts, the strip's two halo columns coming from one 8-byte load that only lanes 0
// and 63 execute; the vertical pass runs on a ring of four horizontally filtered rows in registers.  The tiled kernel above (load window
// -> barrier -> 176 FMAs per lane -> store) holds ~ 50 KB per CU in flight and streams at 4.4 - 4.8 TB/s; here every wave keeps two
// groups of four rows (8 KB + its noise rows) in flight with nothing to wait for but its own loads.  Rows are re-read 3 / FS_ROWS times.
#ifndef MGF_FS_ROWS
#define MGF_FS_ROWS 64
#endif
constexpr int FS_ROWS = MGF_FS_ROWS;

template <bool EP>
__global__ __launch_bounds__(256) void fir_up1_stream(UFParams p) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int strips = p.out_w >> 8, segs = (p.out_h + FS_ROWS - 1) / FS_ROWS;
    int wv = __builtin_amdgcn_readfirstlane((int)blockIdx.x * 4 + (tid >> 6));
    const int strip = wv % strips; wv /= strips;
    const int seg = wv % segs;
    const int plane = wv / segs;
    if (plane >= p.n * p.c) return;
    const int n = plane / p.c, c = plane - n * p.c;
    const float* xb = (const float*)p.x + (int64_t)n * p.sn + (int64_t)c * p.sc;
    const float f00 = p.f[0];
    float fx[4], fy[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int k = p.flip ? t : 3 - t;
        fx[t] = p.f[k] / f00;
        fy[t] = p.f[k * p.fw] * p.gain;
    }
    const int ox = strip * 256 + 4 * lane;
    const int oy_begin = seg * FS_ROWS, ib = oy_begin - p.pady0;
    // halo lanes: lane 0 reads x[ox - 2 .. ox - 1] (uses .y), lane 63 reads x[ox + 4 .. ox + 5]
    const int hx = lane == 0 ? ox - 2 : ox + 4;
    const bool h_on = (lane == 0 && ox > 0) || (lane == 63 && hx < p.in_w);
    const bool m1 = ox + 1 < p.in_w, m2 = ox + 2 < p.in_w, m3 = ox + 3 < p.in_w, mh1 = hx + 1 < p.in_w;
    const bool v_on = ox < p.in_w;
    const float ns = (EP && p.ep.noise && p.ep.noise_strength) ? *p.ep.noise_strength : 1.0f;
    const float bb = (EP && p.ep.bias) ? p.ep.bias[c] : 0.f;
    const float* nzb = (EP && p.ep.noise) ? p.ep.noise + (int64_t)(p.ep.noise_n > 1 ? n : 0) * p.out_h * p.out_w + ox : nullptr;
    float* yb = (float*)p.y + (int64_t)n * p.yn + (int64_t)c * p.yc + ox;
    const float* rb = (EP && p.ep.residual) ? p.ep.residual + (int64_t)n * p.yn + (int64_t)c * p.yc + ox : nullptr;

    auto load_row = [&](int iy, float4& v, float2& h) {
        v = make_float4(0.f, 0.f, 0.f, 0.f);
        h = make_float2(0.f, 0.f);
        if (iy >= 0 && iy < p.in_h) {                                // (wave-uniform)
            const float* row = xb + (int64_t)iy * p.sh;
            if (v_on) v = *reinterpret_cast<const float4*>(row + ox);
            if (h_on) h = *reinterpret_cast<const float2*>(row + hx);
        }
    };
    auto hpass = [&](float (&hz)[4], float4 v, float2 h) {
        if (!m1) v.y = 0.f;
        if (!m2) v.z = 0.f;
        if (!m3) v.w = 0.f;
        if (!mh1) h.y = 0.f;
        // lane i <- lane i - 1 (wave_shr:1, lane 0 keeps `old` = its halo), lane i <- lane i + 1 (wave_shl:1, lane 63 keeps its halo)
        const float l = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, h.y), __builtin_bit_cast(int, v.w), 0x138, 0xf, 0xf, false));
        const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, h.x), __builtin_bit_cast(int, v.x), 0x130, 0xf, 0xf, false));
        const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, h.y), __builtin_bit_cast(int, v.y), 0x130, 0xf, 0xf, false));
        const float w[7] = {l, v.x, v.y, v.z, v.w, r0, r1};
#pragma unroll
        for (int e = 0; e < 4; ++e) hz[e] = w[e] * fx[0] + w[e + 1] * fx[1] + w[e + 2] * fx[2] + w[e + 3] * fx[3];
    };
    float hz[4][4];
    float4 va[4], vb[4], na[4], nb[4];
    float2 ha[4], hb[4];
    auto load_group = [&](int g, float4 (&v)[4], float2 (&h)[4], float4 (&nz)[4]) {      // input rows ib + 3 + 4 g + j, noise of output rows oy_begin + 4 g + j
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            load_row(ib + 3 + 4 * g + j, v[j], h[j]);
            nz[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            const int oy = oy_begin + 4 * g + j;
            if (EP && nzb && oy < p.out_h) nz[j] = *reinterpret_cast<const float4*>(nzb + (int64_t)oy * p.out_w);
        }
    };
    auto do_group = [&](int g, float4 (&v)[4], float2 (&h)[4], float4 (&nz)[4]) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            hpass(hz[(3 + j) & 3], v[j], h[j]);
            const int oy = oy_begin + 4 * g + j;
            if (oy >= p.out_h) continue;                             // (wave-uniform)
            float acc[4];
#pragma unroll
            for (int e = 0; e < 4; ++e)
                acc[e] = hz[j & 3][e] * fy[0] + hz[(j + 1) & 3][e] * fy[1] + hz[(j + 2) & 3][e] * fy[2] + hz[(j + 3) & 3][e] * fy[3];
            if (EP) {
                float4 rv = make_float4(0.f, 0.f, 0.f, 0.f);
                if (rb) rv = *reinterpret_cast<const float4*>(rb + (int64_t)oy * p.yh);
                const float nzv[4] = {nz[j].x * ns, nz[j].y * ns, nz[j].z * ns, nz[j].w * ns};
                const float rr[4] = {rv.x, rv.y, rv.z, rv.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float t = acc[e] + nzv[e];
                    t += bb;
                    if (p.ep.act == MGF_ACT_LRELU) t = t > 0.f ? t : t * p.ep.alpha;
                    else if (p.ep.act == MGF_ACT_RELU) t = t > 0.f ? t : 0.f;
                    acc[e] = t * p.ep.gain + rr[e];
                }
            }
            *reinterpret_cast<float4*>(yb + (int64_t)oy * p.yh) = make_float4(acc[0], acc[1], acc[2], acc[3]);
        }
    };
    {   // rows ib .. ib + 2 prime the ring (slots 0 .. 2)
        float4 v0[3];
        float2 h0[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) load_row(ib + j, v0[j], h0[j]);
        load_group(0, va, ha, na);
#pragma unroll
        for (int j = 0; j < 3; ++j) hpass(hz[j], v0[j], h0[j]);
    }
    constexpr int NG = FS_ROWS / 4;
    static_assert(NG % 2 == 0, "the row groups alternate between two register sets");
    for (int g = 0; g < NG; g += 2) {
        load_group(g + 1, vb, hb, nb);
        do_group(g, va, ha, na);
        if (g + 2 < NG) load_group(g + 2, va, ha, na);
        do_group(g + 1, vb, hb, nb);
    }
}

__global__ __launch_bounds__(256) void fir_up2_wide(UFParams p) {
    constexpr int TW = 64, TH = 16, WV = 10, IH = TH / 2 + 2;       // window: 10 float4 = 40 input columns, 10 input rows
    __shared__ float sx[IH][WV * 4 + 4];
    __shared__ float sf[4][4];
    const int tid = threadIdx.x;
    if (tid < 16) {
        const int jy = tid >> 2, jx = tid & 3;
        float v = 0.f;
        if (jy < p.fh && jx < p.fw) {
            const int ky = p.flip ? jy : p.fh - 1 - jy, kx = p.flip ? jx : p.fw - 1 - jx;
            v = p.f[ky * p.fw + kx] * p.gain;
        }
        sf[jy][jx] = v;
    }
    const int tiles_x = p.out_w / TW, tiles_y = (p.out_h + TH * FIR_SUB - 1) / (TH * FIR_SUB);
    const int tile = blockIdx.x % (tiles_x * tiles_y), plane = blockIdx.x / (tiles_x * tiles_y);
    const int ox0 = (tile % tiles_x) * TW;
    const int n = plane / p.c, c = plane - n * p.c;
    const float* xb = (const float*)p.x + (int64_t)n * p.sn + (int64_t)c * p.sc;
    for (int sub_t = 0; sub_t < FIR_SUB; ++sub_t) {
    const int oy0 = ((tile / tiles_x) * FIR_SUB + sub_t) * TH;
    if (oy0 >= p.out_h) break;
    if (sub_t) __syncthreads();
    // pad (2, 2): output 2m + b reads input columns m - 1 + b (tap jx = b) and m + b (tap jx = b + 2); same for rows
    const int xa = ox0 / 2 - 4, ya = oy0 / 2 - 1;
    for (int i = tid; i < IH * WV; i += 256) {
        const int r = i / WV, v4 = i - r * WV;
        const int iy = ya + r, ix = xa + 4 * v4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (iy >= 0 && iy < p.in_h && ix >= 0 && ix < p.in_w) v = *reinterpret_cast<const float4*>(xb + (int64_t)iy * p.sh + ix);
        sx[r][4 * v4 + 0] = v.x; sx[r][4 * v4 + 1] = v.y; sx[r][4 * v4 + 2] = v.z; sx[r][4 * v4 + 3] = v.w;
    }
    __syncthreads();
    const int lx = tid & 15, ly = tid >> 4;
    const int oy = oy0 + ly;
    const int a2 = ly & 1;                                           // output row parity (oy0 is even)
    const int r0 = (ly >> 1) + a2;                                   // window row of input row (oy/2 - 1 + a2)
    // outputs 4*lx + e, e = 0..3 -> m = 2*lx + (e >> 1), b = e & 1 ; window column of input column j is j - xa = j - ox0/2 + 4
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ry = 0; ry < 2; ++ry) {
        const int jy = a2 + 2 * ry;
        float w[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) w[q] = sx[r0 + ry][2 * lx + 3 + q];       // input columns 2*lx - 1 .. 2*lx + 2
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int m = e >> 1, b = e & 1;
            acc[e] += w[m + b] * sf[jy][b] + w[m + b + 1] * sf[jy][b + 2];
        }
    }
    if (oy < p.out_h) {
        const int64_t yoff = (int64_t)n * p.yn + (int64_t)c * p.yc + (int64_t)oy * p.yh + ox0 + 4 * lx;
        *reinterpret_cast<float4*>((float*)p.y + yoff) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    }
    }
}

}  // namespace

extern "C" int mgf_upfirdn2d(void* y, const void* x, const float* f, int dtype, int32_t n, int32_t c, int32_t in_h,
                             int32_t in_w, int64_t sn, int64_t sc, int64_t sh, int64_t sw, int32_t out_h, int32_t out_w,
                             int64_t yn, int64_t yc, int64_t yh, int64_t yw, int32_t fh, int32_t fw, int32_t upx,
                             int32_t upy, int32_t downx, int32_t downy, int32_t padx0, int32_t padx1, int32_t pady0,
                             int32_t pady1, int32_t flip, float gain, const mgf_epilogue* ep, mgf_stream_t stream) {
    MGF_REQUIRE(dtype == MGF_F32 || dtype == MGF_F64 || dtype == MGF_F16, MGF_EUNSUPPORTED, "upfirdn2d: unsupported dtype %d", dtype);
    MGF_REQUIRE(n >= 0 && c >= 0 && in_h >= 1 && in_w >= 1, MGF_EINVAL, "upfirdn2d: bad input shape");
    MGF_REQUIRE(fh >= 1 && fw >= 1 && fh * fw <= 8192, MGF_EINVAL, "upfirdn2d: filter must have 1..8192 taps (got %dx%d)", fh, fw);
    MGF_REQUIRE(upx >= 1 && upy >= 1 && downx >= 1 && downy >= 1, MGF_EINVAL, "upfirdn2d: up/down factors must be >= 1");
    const int64_t eh = ((int64_t)in_h * upy + pady0 + pady1 - fh + downy) / downy;
    const int64_t ew = ((int64_t)in_w * upx + padx0 + padx1 - fw + downx) / downx;
    MGF_REQUIRE(eh >= 1 && ew >= 1, MGF_EINVAL, "upfirdn2d: output would be empty (%lld x %lld)", (long long)eh, (long long)ew);
    MGF_REQUIRE(eh == out_h && ew == out_w, MGF_EINVAL, "upfirdn2d: output shape mismatch: expected %lldx%lld, got %dx%d",
                (long long)eh, (long long)ew, out_h, out_w);
    // the reference's contract (upfirdn2d.cpp:14-15,28): numel <= INT_MAX.  MGF_FILTER_LARGE (the engine's own calls) lifts it for the
    // kernels that address a plane through a 64-bit base: every tiled / streaming kernel below; the element-indexed ones refuse
    const bool big = (int64_t)n * c * in_h * in_w > INT32_MAX || (int64_t)n * c * out_h * out_w > INT32_MAX;
    MGF_REQUIRE(!big || ((flip & MGF_FILTER_LARGE) && (int64_t)in_h * in_w <= INT32_MAX && (int64_t)out_h * out_w <= INT32_MAX), MGF_ETOOBIG,
                "upfirdn2d: tensor too large");
    if (n == 0 || c == 0) return MGF_OK;
    MGF_REQUIRE(x && y && f, MGF_EINVAL, "upfirdn2d: null pointer");
    if (ep) {
        MGF_REQUIRE(dtype == MGF_F32, MGF_EUNSUPPORTED, "upfirdn2d: fused epilogue needs float32");
        MGF_REQUIRE(ep->act == MGF_ACT_LINEAR || ep->act == MGF_ACT_LRELU || ep->act == MGF_ACT_RELU || ep->act == 0,
                    MGF_EUNSUPPORTED, "upfirdn2d: epilogue activation %d unsupported", ep->act);
    }
    UFParams p;
    p.y = y; p.x = x; p.f = f; p.n = n; p.c = c; p.in_h = in_h; p.in_w = in_w;
    p.sn = sn; p.sc = sc; p.sh = sh; p.sw = sw; p.out_h = out_h; p.out_w = out_w;
    p.yn = yn; p.yc = yc; p.yh = yh; p.yw = yw; p.fh = fh; p.fw = fw;
    p.upx = upx; p.upy = upy; p.downx = downx; p.downy = downy; p.padx0 = padx0; p.pady0 = pady0;
    p.flip = flip & 1; p.gain = gain; p.has_ep = ep != nullptr; p.sep_ok = (flip & MGF_FILTER_SEPARABLE) != 0; p.vec_in = 0;
    if (ep) { p.ep = *ep; if (p.ep.act == 0) p.ep.act = MGF_ACT_LINEAR; } else { p.ep = mgf_epilogue{}; p.ep.gain = 1.f; }
    hipStream_t stq = (hipStream_t)stream;
    const bool tiled = dtype == MGF_F32 && sw == 1 && yw == 1 && fh <= 4 && fw <= 4 && upx == upy && (upx == 1 || upx == 2) &&
                       downx == 1 && downy == 1 && out_w >= 32 && (int64_t)n * c * mgf_cdiv(out_h, 16) * mgf_cdiv(out_w, 64) <= INT32_MAX;
    // wide float4 kernels: 4x4 filter, out_w a multiple of 64, 16-byte aligned rows on both sides
    const bool wide_ok = tiled && fh == 4 && fw == 4 && out_w % 64 == 0 && sh % 4 == 0 && sc % 4 == 0 && sn % 4 == 0 &&
                         yh % 4 == 0 && yc % 4 == 0 && yn % 4 == 0 && ((uintptr_t)x % 16 == 0) && ((uintptr_t)y % 16 == 0) &&
                         (!ep || ((!ep->noise || (uintptr_t)ep->noise % 16 == 0) && (!ep->residual || (uintptr_t)ep->residual % 16 == 0)));
    if (wide_ok && upx == 1 && padx0 == 1 && sh >= (int64_t)((in_w + 3) / 4) * 4 && pady0 >= 0 && pady0 <= 3) {
        static const char* sep_env = mgf_knob("MGF_FIR_SEP");          // tuning hook (experiments only): 0 = never take the separable kernel
        static const char* stream_env = mgf_knob("MGF_FIR_STREAM");    // tuning hook: 0 = the LDS-tiled separable kernel on wide maps too
        const int64_t swaves = (int64_t)n * c * (out_w / 256) * mgf_cdiv(out_h, FS_ROWS);
        if (p.sep_ok && out_w % 256 == 0 && out_h >= FS_ROWS && swaves >= 2048 && swaves <= INT32_MAX - 4 && !(stream_env && stream_env[0] == '0')) {
            const int blocks = (int)mgf_cdiv(swaves, 4);
            if (ep) hipLaunchKernelGGL((fir_up1_stream<true>), dim3(blocks), dim3(256), 0, stq, p);
            else hipLaunchKernelGGL((fir_up1_stream<false>), dim3(blocks), dim3(256), 0, stq, p);
        } else if (p.sep_ok && !(sep_env && sep_env[0] == '0')) {
            const int blocks = n * c * (int)mgf_cdiv(out_h, 64) * (out_w / 64);
            if (ep) hipLaunchKernelGGL((fir_up1_sep<true>), dim3(blocks), dim3(256), 0, stq, p);
            else hipLaunchKernelGGL((fir_up1_sep<false>), dim3(blocks), dim3(256), 0, stq, p);
        } else {
        const int blocks = n * c * (int)mgf_cdiv(out_h, 16 * FIR_SUB) * (out_w / 64);
        if (ep) hipLaunchKernelGGL((fir_up1_wide<true>), dim3(blocks), dim3(256), 0, stq, p);
        else hipLaunchKernelGGL((fir_up1_wide<false>), dim3(blocks), dim3(256), 0, stq, p);
        }
    } else if (tiled && fh == 4 && fw == 4 && upx == 1 && padx0 == 2 && pady0 >= 0 && pady0 <= 3 && !ep && p.sep_ok && in_w % 4 == 0 &&
               sh % 4 == 0 && sc % 4 == 0 && sn % 4 == 0 && ((uintptr_t)x % 16 == 0) && sh >= in_w &&
               (int64_t)n * c * mgf_cdiv(out_h, 64) * mgf_cdiv(out_w, 64) <= INT32_MAX) {
        // the gradient of the post-transposed-conv blur: separable kernel, pad 2, element-wise stores into rows of any width
        const int blocks = n * c * (int)mgf_cdiv(out_h, 64) * (int)mgf_cdiv(out_w, 64);
        hipLaunchKernelGGL((fir_up1_sep<false, 2, true>), dim3(blocks), dim3(256), 0, stq, p);
    } else if (wide_ok && upx == 2 && padx0 == 2 && pady0 == 2 && !ep && in_w % 4 == 0 && out_h % 2 == 0) {
        const int blocks = n * c * (int)mgf_cdiv(out_h, 16 * FIR_SUB) * (out_w / 64);
        hipLaunchKernelGGL(fir_up2_wide, dim3(blocks), dim3(256), 0, stq, p);
    } else if (dtype == MGF_F32 && sw == 1 && yw == 1 && fh <= 4 && fw <= 4 && upx == 1 && upy == 1 && downx == 2 && downy == 2 && !ep &&
               out_w >= 32 && (int64_t)n * c * mgf_cdiv(out_h, 16) * mgf_cdiv(out_w, 64) <= INT32_MAX) {
        const int blocks = n * c * (int)mgf_cdiv(out_h, 16) * (int)mgf_cdiv(out_w, 64);
        p.vec_in = in_w % 4 == 0 && sh % 4 == 0 && sc % 4 == 0 && sn % 4 == 0 && ((uintptr_t)x % 16) == 0;
        hipLaunchKernelGGL(fir_down2_tiled, dim3(blocks), dim3(256), 0, stq, p);
    } else if (tiled) {
        const int blocks = n * c * (int)mgf_cdiv(out_h, 16) * (int)mgf_cdiv(out_w, 64);
        if (upx == 1) hipLaunchKernelGGL((upfirdn_tiled_f32<1>), dim3(blocks), dim3(256), 0, stq, p);
        else hipLaunchKernelGGL((upfirdn_tiled_f32<2>), dim3(blocks), dim3(256), 0, stq, p);
    } else if (dtype == MGF_F32 && fh == 4 && fw == 4 && upx == upy && downx == downy && (upx == 1 || upx == 2) && (downx == 1 || downx == 2) &&
               !(upx == 2 && downx == 2) && (int64_t)n * c * out_h * out_w <= (int64_t)1 << 22) {
        const unsigned blocks = (unsigned)mgf_cdiv((int64_t)n * c * out_h * out_w, 256);
        if (upx == 2) hipLaunchKernelGGL((upfirdn_small4<2, 1>), dim3(blocks), dim3(256), 0, stq, p);
        else if (downx == 2) hipLaunchKernelGGL((upfirdn_small4<1, 2>), dim3(blocks), dim3(256), 0, stq, p);
        else hipLaunchKernelGGL((upfirdn_small4<1, 1>), dim3(blocks), dim3(256), 0, stq, p);
    } else {
        MGF_REQUIRE(!big, MGF_ETOOBIG, "upfirdn2d: tensor too large for the general kernel (32-bit element indices)");
        const int64_t total = (int64_t)n * c * out_h * out_w;
        const int grid = mgf_stream_grid(total, 256, 4);
        const size_t lds = (size_t)fh * fw * sizeof(float);
        if (dtype == MGF_F32) hipLaunchKernelGGL((upfirdn_generic<float, float>), dim3(grid), dim3(256), lds, stq, p);
        else if (dtype == MGF_F64) hipLaunchKernelGGL((upfirdn_generic<double, double>), dim3(grid), dim3(256), lds, stq, p);
        else hipLaunchKernelGGL((upfirdn_generic<__half, float>), dim3(grid), dim3(256), lds, stq, p);
    }
    MGF_CHECK_LAUNCH("upfirdn2d");
    return MGF_OK;
}
