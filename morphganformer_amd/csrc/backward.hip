// Gradient mode: dLoss/dlatent through the synthesis network (SURVEY.md section 8a row P0 "gradient mode"; the ops the reference
// differentiates with autograd: bias_act.py:137-198, upfirdn2d.py:237-256, modulated_conv2d networks.py:253-328,
// TransformerLayer.forward :748-822).  Weights are constants, so per layer only three things are needed:
//   * d(input activation): a "dgrad" convolution -- the forward tap-list kernel with channel-transposed taps (conv_taps.hip);
//   * d(style):   y_o = d_o * sum_i s_i (W_oi * x_i)   =>   dL/ds_i = <x_i, g_i> - s_i * sum_o <dc_o, c_o> d_o^2 Wsq_oi
//                 with g_i = sum_o W_oi^T * (d_o dc_o) the un-modulated dgrad result -- two per-channel dot products instead of a
//                 per-sample weight-gradient GEMM;
//   * d(attention value table) and d(x) through the duplex attention.
// Every reduction is two-stage and deterministic (per-chunk partials, summed in a fixed order by the consumer).
// Contract: include/mgf.h ("Gradient mode").
#include "mgf_common.h"

namespace {

constexpr int BWD_CHUNK = 4096;          // elements of one (sample, channel) plane handled by one workgroup

__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// ------------------------------------------------------------------------------------------ activation backward
struct ActBwdParams {
    float* dz;
    float* part;
    const float* dy;
    const float* y;
    const float* res;
    const float* bias;
    const float* noise;
    const float* nstr;
    int noise_n, c, nchunk;
    int64_t hw;
    float alpha, gain;
    const float* res_low;   // LOW: the residual at half resolution [n][c][h/2][w/2] (res is null), up-sampled here
    int w;                  // LOW: row length of the maps (w % 4 == 0, h = hw / w even)
};

// The residual given at HALF resolution (the resnet skip branch before its 2x FIR up-sampling: [1,3,3,1] x [1,3,3,1] / 64, gain 4, pad
// (2,1,2,1)): the four values of row y from column x0 (x0 % 4 == 0), in the arithmetic of the form-3 Winograd epilogue that adds the
// same up-sampled residual in the forward (csrc/wino3.hip: rows first -- (m-1, m) x (1/4, 3/4) for an even row 2m, (m, m+1) x (3/4, 1/4)
// for an odd one -- then columns alike; samples outside the map count as zero).
__device__ __forceinline__ float4 up2_residual4(const float* lowp, int lh, int lw, int y, int x0) {
    const int orow = y & 1, ra = (y >> 1) - 1 + orow, rb = ra + 1;
    const float wa = orow ? 0.75f : 0.25f, wb = 1.f - wa;
    const bool ina = ra >= 0 && ra < lh, inb = rb >= 0 && rb < lh;
    const int mc = x0 >> 1;
    float cc[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int col = mc - 1 + k;
        const bool in = col >= 0 && col < lw;
        const float a = (in && ina) ? lowp[ra * lw + col] : 0.f, b = (in && inb) ? lowp[rb * lw + col] : 0.f;
        cc[k] = wa * a + wb * b;
    }
    return make_float4(0.25f * cc[0] + 0.75f * cc[1], 0.75f * cc[1] + 0.25f * cc[2], 0.25f * cc[1] + 0.75f * cc[2], 0.75f * cc[2] + 0.25f * cc[3]);
}

// grid (nchunk, c, n).  y = lrelu(z) * gain + res  with  z = cval + noise * strength + bias:
//   dz = dy * gain * (y - res > 0 ? 1 : alpha);    part[n,c,chunk] = sum dz * cval   (cval recovered by inverting the activation)
template <bool VEC, bool LOW = false>
__global__ __launch_bounds__(256) void act_bwd_kernel(ActBwdParams p) {
    static_assert(VEC || !LOW, "the half-resolution residual rides on the 16-byte path");
    __shared__ float red[4];
    const int chunk = blockIdx.x, ch = blockIdx.y, n = blockIdx.z;
    const int64_t base = ((int64_t)n * p.c + ch) * p.hw;
    const int64_t i0 = (int64_t)chunk * BWD_CHUNK;
    const int64_t i1 = min(p.hw, i0 + (int64_t)BWD_CHUNK);
    const float b = p.bias ? p.bias[ch] : 0.f;
    const float ns = p.noise ? (p.nstr ? *p.nstr : 1.f) : 0.f;
    const float* nz = p.noise ? p.noise + (int64_t)(p.noise_n > 1 ? n : 0) * p.hw : nullptr;
    const float inv_gain = 1.f / p.gain, inv_alpha = 1.f / p.alpha;
    float acc = 0.f;
    auto one = [&](float yv, float rv, float dyv, float nv, float& dzv) {
        const float v = yv - rv;
        const bool pos = v > 0.f;
        dzv = dyv * p.gain * (pos ? 1.f : p.alpha);
        if (p.part) {
            const float zv = (pos ? v : v * inv_alpha) * inv_gain;
            acc += dzv * (zv - b - nv * ns);
        }
    };
    if (VEC) {          // hw % 4 == 0 and 16-byte aligned operands: four elements per lane and access
        for (int64_t i = i0 + 4 * threadIdx.x; i < i1; i += 1024) {
            const float4 yv = *reinterpret_cast<const float4*>(p.y + base + i);
            const float4 dv = *reinterpret_cast<const float4*>(p.dy + base + i);
            float4 rv;
            if (LOW) {
                const int yy = (int)(i / p.w);
                rv = up2_residual4(p.res_low + ((int64_t)n * p.c + ch) * (p.hw >> 2), (int)(p.hw / p.w) >> 1, p.w >> 1, yy, (int)(i - (int64_t)yy * p.w));
            } else {
                rv = p.res ? *reinterpret_cast<const float4*>(p.res + base + i) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
            const float4 nv = nz ? *reinterpret_cast<const float4*>(nz + i) : make_float4(0.f, 0.f, 0.f, 0.f);
            float4 o;
            one(yv.x, rv.x, dv.x, nv.x, o.x); one(yv.y, rv.y, dv.y, nv.y, o.y);
            one(yv.z, rv.z, dv.z, nv.z, o.z); one(yv.w, rv.w, dv.w, nv.w, o.w);
            *reinterpret_cast<float4*>(p.dz + base + i) = o;
        }
    } else {
        for (int64_t i = i0 + threadIdx.x; i < i1; i += 256) {
            float o;
            one(p.y[base + i], p.res ? p.res[base + i] : 0.f, p.dy[base + i], nz ? nz[i] : 0.f, o);
            p.dz[base + i] = o;
        }
    }
    if (p.part) {
        const float tot = block_sum(acc, red);
        if (threadIdx.x == 0) p.part[((int64_t)n * p.c + ch) * p.nchunk + chunk] = tot;
    }
}

// part[n,c,chunk] = sum a * b over the chunk
__device__ __forceinline__ void channel_dot_body(float* part, const float* a, const float* b, int c, int64_t hw, int nchunk, int chunk, int ch, int n,
                                                 float* red) {
    const int64_t base = ((int64_t)n * c + ch) * hw;
    const int64_t i0 = (int64_t)chunk * BWD_CHUNK;
    const int64_t i1 = min(hw, i0 + (int64_t)BWD_CHUNK);
    float acc = 0.f;
    for (int64_t i = i0 + threadIdx.x; i < i1; i += 256) acc += a[base + i] * b[base + i];
    const float tot = block_sum(acc, red);
    if (threadIdx.x == 0) part[((int64_t)n * c + ch) * nchunk + chunk] = tot;
}

__global__ __launch_bounds__(256) void channel_dot_kernel(float* part, const float* a, const float* b, int c, int64_t hw, int nchunk) {
    __shared__ float red[4];
    channel_dot_body(part, a, b, c, hw, nchunk, blockIdx.x, blockIdx.y, blockIdx.z, red);
}

// part[n,c,chunk] = sum x * g ;  dx (+)= s[n,c] * g
template <bool VEC>
__global__ __launch_bounds__(256) void style_grad_kernel(float* part, float* dx, const float* x, const float* g, const float* s, int c,
                                                         int64_t hw, int nchunk, int accumulate) {
    __shared__ float red[4];
    const int chunk = blockIdx.x, ch = blockIdx.y, n = blockIdx.z;
    const int64_t base = ((int64_t)n * c + ch) * hw;
    const int64_t i0 = (int64_t)chunk * BWD_CHUNK;
    const int64_t i1 = min(hw, i0 + (int64_t)BWD_CHUNK);
    const float sv = s ? s[(int64_t)n * c + ch] : 1.f;
    float acc = 0.f;
    if (VEC) {
        for (int64_t i = i0 + 4 * threadIdx.x; i < i1; i += 1024) {
            const float4 gv = *reinterpret_cast<const float4*>(g + base + i);
            const float4 xv = *reinterpret_cast<const float4*>(x + base + i);
            acc += xv.x * gv.x + xv.y * gv.y + xv.z * gv.z + xv.w * gv.w;
            float4 o = make_float4(sv * gv.x, sv * gv.y, sv * gv.z, sv * gv.w);
            if (accumulate) {
                const float4 d = *reinterpret_cast<const float4*>(dx + base + i);
                o.x += d.x; o.y += d.y; o.z += d.z; o.w += d.w;
            }
            *reinterpret_cast<float4*>(dx + base + i) = o;
        }
    } else {
        for (int64_t i = i0 + threadIdx.x; i < i1; i += 256) {
            const float gv = g[base + i];
            acc += x[base + i] * gv;
            const float o = sv * gv;
            dx[base + i] = accumulate ? dx[base + i] + o : o;
        }
    }
    const float tot = block_sum(acc, red);
    if (threadIdx.x == 0) part[((int64_t)n * c + ch) * nchunk + chunk] = tot;
}

// ------------------------------------------------------------------------------------------ duplex attention backward
// Forward (attention.hip):  S = x^T wqc + spos;  P = softmax_t(S);  r = rsqrt(mean_c x^2 + 1e-8);  g[c] = sum_t P[t] vwb[c,t];
//                           a[c] = x[c] * r * g[c]
// Backward per pixel, given da:
//   dg[c] = da[c] x[c] r                  (-> dvwb[c,t] = sum_f dg[c,f] P[f,t], second kernel)
//   dP[t] = sum_c dg[c] vwb[c,t];   PdP = sum_t P[t] dP[t]  ( = r * sum_c da[c] x[c] g[c], i.e. the norm term for free )
//   dS[t] = P[t] (dP[t] - PdP)
//   dx[c] = da[c] r g[c]  -  (PdP r^2 / C) x[c]  +  sum_t dS[t] wqc[c,t]
constexpr int TMAX = 16;
struct AttnBwdParams {
    float* dx;
    float* dg;
    float* probs;
    const float* da;
    const float* x;
    const float* wqc;
    const float* spos;
    const float* vwb;
    int n, c, f, t, c_pad;
};

template <int PXB>
__global__ __launch_bounds__(256) void duplex_attention_bwd_kernel(AttnBwdParams p) {
    constexpr int G = 256 / PXB;
    extern __shared__ float lds[];
    float* tq = lds;                                  // [c_pad][16]
    float* tv = tq + (size_t)p.c_pad * TMAX;          // [c_pad][16]
    float* part = tv + (size_t)p.c_pad * TMAX;        // [G][17][PXB]
    const int tid = threadIdx.x;
    const int px = tid % PXB, grp = tid / PXB;
    const int n = blockIdx.y;
    const int f = blockIdx.x * PXB + px;
    const bool valid = f < p.f;
    const int fc = valid ? f : p.f - 1;
    const int T = p.t;
    const int64_t plane = (int64_t)n * p.c * p.f;
    const float* xn = p.x + plane;
    const float* dan = p.da + plane;
    const float* vw = p.vwb + (int64_t)n * p.c * T;
    for (int i = tid; i < p.c_pad * TMAX; i += 256) {
        const int c = i / TMAX, t = i % TMAX;
        const bool ok = c < p.c && t < T;
        tq[i] = ok ? p.wqc[(int64_t)c * T + t] : 0.f;
        tv[i] = ok ? vw[(int64_t)c * T + t] : 0.f;
    }
    __syncthreads();

    float s[TMAX];
#pragma unroll
    for (int t = 0; t < TMAX; ++t) s[t] = 0.f;
    float sq = 0.f;
    // the channel sweeps issue ABU independent row loads before the first use: these layers are 4x4 .. 128x128 maps, i.e. latency --
    // one load per iteration made every sweep a chain of C / G dependent memory round trips (55 us per launch whatever the grid)
    constexpr int ABU = 8;
    for (int c0 = grp; c0 < p.c; c0 += G * ABU) {
        float xv[ABU];
#pragma unroll
        for (int u = 0; u < ABU; ++u) {
            const int c = c0 + u * G;
            xv[u] = c < p.c ? xn[(int64_t)c * p.f + fc] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < ABU; ++u) {
            const int c = c0 + u * G;
            if (c < p.c) {
                const float4* w4 = reinterpret_cast<const float4*>(tq + c * TMAX);
                sq += xv[u] * xv[u];
#pragma unroll
                for (int q = 0; q < TMAX / 4; ++q) {
                    const float4 w = w4[q];
                    s[4 * q + 0] += xv[u] * w.x; s[4 * q + 1] += xv[u] * w.y; s[4 * q + 2] += xv[u] * w.z; s[4 * q + 3] += xv[u] * w.w;
                }
            }
        }
    }
#pragma unroll
    for (int t = 0; t < TMAX; ++t) part[(grp * (TMAX + 1) + t) * PXB + px] = s[t];
    part[(grp * (TMAX + 1) + TMAX) * PXB + px] = sq;
    __syncthreads();
    float P[TMAX];
    float m = -3.0e38f;
#pragma unroll
    for (int t = 0; t < TMAX; ++t) {
        float v = 0.f;
        for (int g = 0; g < G; ++g) v += part[(g * (TMAX + 1) + t) * PXB + px];
        if (t < T) { v += p.spos[(int64_t)fc * T + t]; m = fmaxf(m, v); }
        P[t] = v;
    }
    sq = 0.f;
    for (int g = 0; g < G; ++g) sq += part[(g * (TMAX + 1) + TMAX) * PXB + px];
    float den = 0.f;
#pragma unroll
    for (int t = 0; t < TMAX; ++t) {
        P[t] = t < T ? __expf(P[t] - m) : 0.f;
        den += P[t];
    }
    const float inv = 1.f / den;
#pragma unroll
    for (int t = 0; t < TMAX; ++t) P[t] *= inv;
    const float r = rsqrtf(sq / (float)p.c + 1e-8f);
    if (grp == 0 && valid && p.probs)
        for (int t = 0; t < T; ++t) p.probs[((int64_t)n * p.f + f) * T + t] = P[t];
    __syncthreads();

    // ---- pass 2: dP ----
#pragma unroll
    for (int t = 0; t < TMAX; ++t) s[t] = 0.f;
    for (int c0 = grp; c0 < p.c; c0 += G * ABU) {
        float dv[ABU], xv[ABU];
#pragma unroll
        for (int u = 0; u < ABU; ++u) {
            const int c = c0 + u * G;
            const int64_t o = (int64_t)(c < p.c ? c : 0) * p.f + fc;
            dv[u] = dan[o]; xv[u] = xn[o];
        }
#pragma unroll
        for (int u = 0; u < ABU; ++u) {
            const int c = c0 + u * G;
            if (c < p.c) {
                const float dgv = dv[u] * xv[u] * r;
                if (valid && p.dg) p.dg[plane + (int64_t)c * p.f + f] = dgv;
                const float4* w4 = reinterpret_cast<const float4*>(tv + c * TMAX);
#pragma unroll
                for (int q = 0; q < TMAX / 4; ++q) {
                    const float4 w = w4[q];
                    s[4 * q + 0] += dgv * w.x; s[4 * q + 1] += dgv * w.y; s[4 * q + 2] += dgv * w.z; s[4 * q + 3] += dgv * w.w;
                }
            }
        }
    }
#pragma unroll
    for (int t = 0; t < TMAX; ++t) part[(grp * (TMAX + 1) + t) * PXB + px] = s[t];
    __syncthreads();
    float pdp = 0.f;
#pragma unroll
    for (int t = 0; t < TMAX; ++t) {
        float v = 0.f;
        for (int g = 0; g < G; ++g) v += part[(g * (TMAX + 1) + t) * PXB + px];
        s[t] = v;
        pdp += P[t] * v;
    }
#pragma unroll
    for (int t = 0; t < TMAX; ++t) s[t] = P[t] * (s[t] - pdp);          // dS
    const float coef = pdp * r * r / (float)p.c;

    // ---- pass 3: dx ----
    float* dxn = p.dx + plane;
    for (int c0 = grp; c0 < p.c; c0 += G * ABU) {
        float dv[ABU], xv[ABU];
#pragma unroll
        for (int u = 0; u < ABU; ++u) {
            const int c = c0 + u * G;
            const int64_t o = (int64_t)(c < p.c ? c : 0) * p.f + fc;
            dv[u] = dan[o]; xv[u] = xn[o];
        }
#pragma unroll
        for (int u = 0; u < ABU; ++u) {
            const int c = c0 + u * G;
            if (c < p.c) {
                const float4* v4 = reinterpret_cast<const float4*>(tv + c * TMAX);
                const float4* q4 = reinterpret_cast<const float4*>(tq + c * TMAX);
                float g = 0.f, qs = 0.f;
#pragma unroll
                for (int q = 0; q < TMAX / 4; ++q) {
                    const float4 a = v4[q], b = q4[q];
                    g += P[4 * q + 0] * a.x + P[4 * q + 1] * a.y + P[4 * q + 2] * a.z + P[4 * q + 3] * a.w;
                    qs += s[4 * q + 0] * b.x + s[4 * q + 1] * b.y + s[4 * q + 2] * b.z + s[4 * q + 3] * b.w;
                }
                const float v = dv[u] * r * g - coef * xv[u] + qs;
                if (valid) dxn[(int64_t)c * p.f + f] = v;
            }
        }
    }
}

// MFMA form of the kernel above for the generator's layers (C = 256 / 512, 16 latents, whole 32-pixel tiles) -- the backward twin of
// duplex_attention_mfma_kernel (csrc/attention.hip), same tiling: a workgroup = 32 pixels, wave w = channels [64 w, 64 w + 64), x and da
// read ONCE into the 32 + 32 registers that are the lane's B-operand slots of the C -> 16 GEMMs and its accumulator slots of the 16 -> C
// GEMMs (k-steps walk the channels in accumulator order).  Four small GEMMs on v_mfma_f32_32x32x2_f32:
//   scores S = wqc^T x (+ spos) -> softmax P, r = rsqrt(mean x^2 + eps)        [partials + sum x^2 meet in LDS]
//   dP = vwb^T dg,  dg = da x r                                                [partials meet in LDS]
//   dS = P (dP - <P, dP>),   coef = <P, dP> r^2 / C   (the layer-norm term: sum_c dg g = <P, dP>)
//   per 32-channel block:  G = vwb P,  Q = wqc dS,  dx = da r G + Q - coef x
typedef float adb_f32x16 __attribute__((ext_vector_type(16)));
template <int NW>
__global__ __launch_bounds__(NW * 64, NW == 4 ? 3 : 1) void duplex_attention_bwd_mfma_kernel(AttnBwdParams p) {
    __shared__ float red[2][NW][9][64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const int n = blockIdx.y, f0 = blockIdx.x * 32;
    const int cw0 = wv * 64;
    const int64_t nb = (int64_t)n * p.c * p.f;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x + nb), 0, p.c * p.f * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rda = __builtin_amdgcn_make_buffer_rsrc((void*)(p.da + nb), 0, p.c * p.f * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc((void*)p.wqc, 0, p.c * TMAX * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void*)(p.vwb + (int64_t)n * p.c * TMAX), 0, p.c * TMAX * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsp = __builtin_amdgcn_make_buffer_rsrc((void*)p.spos, 0, p.f * TMAX * 4, 0x00020000);
    auto rowof = [](int r) { return (r & 3) + 8 * (r >> 2); };         // accumulator register -> row of the 32x32 block (+ 4 half)

    const unsigned xo = (unsigned)((cw0 + 4 * half) * p.f + f0 + l31) * 4u;
    const unsigned to = l31 < TMAX ? (unsigned)((cw0 + 4 * half) * TMAX + l31) * 4u : 0xFFFFFFF0u;     // [c][t] table, transposed operand (t = lane)
    float xr[2][16], dar[2][16];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ch = 32 * cb + rowof(r);
            xr[cb][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, xo, ch * p.f * 4, 0));
            dar[cb][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rda, xo, ch * p.f * 4, 0));
        }
    float sp[8];
#pragma unroll
    for (int r = 0; r < 8; ++r)
        sp[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsp, (unsigned)((f0 + l31) * TMAX + 4 * half) * 4u, rowof(r) * 4, 0));

    // ---- scores and sum x^2 ----
    adb_f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float sq = 0.f;
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float a = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rq, to, (32 * cb + rowof(r)) * TMAX * 4, 0));
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, xr[cb][r], acc, 0, 0, 0);
            sq += xr[cb][r] * xr[cb][r];
        }
#pragma unroll
    for (int r = 0; r < 8; ++r) red[0][wv][r][lane] = acc[r];
    red[0][wv][8][lane] = sq;
    __syncthreads();
    float P[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        float v = sp[r];
#pragma unroll
        for (int w = 0; w < NW; ++w) v += red[0][w][r][lane];
        P[r] = v;
    }
    sq = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) sq += red[0][w][8][lane];
    sq += __shfl_xor(sq, 32, 64);
    float m = P[0];
#pragma unroll
    for (int r = 1; r < 8; ++r) m = fmaxf(m, P[r]);
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float den = 0.f;
#pragma unroll
    for (int r = 0; r < 8; ++r) { P[r] = __expf(P[r] - m); den += P[r]; }
    den += __shfl_xor(den, 32, 64);
    const float inv = 1.f / den;
#pragma unroll
    for (int r = 0; r < 8; ++r) P[r] *= inv;
    const float rn = rsqrtf(sq / (float)p.c + 1e-8f);
    if (wv == 0 && p.probs) {
#pragma unroll
        for (int r = 0; r < 8; ++r) p.probs[((int64_t)n * p.f + f0 + l31) * TMAX + rowof(r) + 4 * half] = P[r];
    }

    // ---- dg = da x r (written out for the value gradient), dP = vwb^T dg ----
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float* dgn = p.dg ? p.dg + nb : nullptr;
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ch = 32 * cb + rowof(r);
            const float a = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rv, to, ch * TMAX * 4, 0));
            const float dgv = dar[cb][r] * xr[cb][r] * rn;
            if (dgn) dgn[(int64_t)(cw0 + ch + 4 * half) * p.f + f0 + l31] = dgv;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, dgv, acc, 0, 0, 0);
        }
#pragma unroll
    for (int r = 0; r < 8; ++r) red[1][wv][r][lane] = acc[r];
    __syncthreads();
    float dS[8], pdp = 0.f;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) v += red[1][w][r][lane];
        dS[r] = v;                                   // dP
        pdp += P[r] * v;
    }
    pdp += __shfl_xor(pdp, 32, 64);
#pragma unroll
    for (int r = 0; r < 8; ++r) dS[r] = P[r] * (dS[r] - pdp);
    const float coef = pdp * rn * rn / (float)p.c;

    // ---- per 32-channel block: G = vwb P, Q = wqc dS, dx = da r G + Q - coef x ----
    float* dxn = p.dx + nb;
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
        const unsigned vo = (unsigned)((cw0 + 32 * cb + l31) * TMAX + 4 * half) * 4u;      // row of channel l31 of the block, my half's latents
        const float4 va = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rv, vo, 0, 0));
        const float4 vb = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rv, vo + 32u, 0, 0));
        const float4 qa = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rq, vo, 0, 0));
        const float4 qb = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rq, vo + 32u, 0, 0));
        const float av[8] = {va.x, va.y, va.z, va.w, vb.x, vb.y, vb.z, vb.w};
        const float aq[8] = {qa.x, qa.y, qa.z, qa.w, qb.x, qb.y, qb.z, qb.w};
        adb_f32x16 g, q;
#pragma unroll
        for (int r = 0; r < 16; ++r) { g[r] = 0.f; q[r] = 0.f; }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            g = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], P[j], g, 0, 0, 0);
            q = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[j], dS[j], q, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float v = dar[cb][r] * rn * g[r] - coef * xr[cb][r] + q[r];
            dxn[(int64_t)(cw0 + 32 * cb + rowof(r) + 4 * half) * p.f + f0 + l31] = v;
        }
    }
}

// dvwb[n,c,t] = sum_f dg[n,c,f] * P[n,f,t]; grid (cdiv(c,4), n): a workgroup owns 4 channels and streams all pixels once
__global__ __launch_bounds__(256) void attn_values_grad_kernel(float* dvwb, const float* dg, const float* probs, int c, int f, int T) {
    __shared__ float red[4][4 * TMAX];
    const int c0 = blockIdx.x * 4, n = blockIdx.y, tid = threadIdx.x;
    float acc[4][TMAX];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int t = 0; t < TMAX; ++t) acc[u][t] = 0.f;
    const float* pn = probs + (int64_t)n * f * T;
    const float* dgn = dg + (int64_t)n * c * f;
    // two pixel slices per trip: their loads are independent, so twice the bytes are in flight per round trip of this latency-bound sweep
    for (int i0 = tid; i0 < f; i0 += 512) {
        float pv[2][TMAX], dv[2][4];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int i = i0 + 256 * h;
            const bool ok = i < f;
            const int ic = ok ? i : 0;
#pragma unroll
            for (int t = 0; t < TMAX; ++t) pv[h][t] = (ok && t < T) ? pn[(int64_t)ic * T + t] : 0.f;
#pragma unroll
            for (int u = 0; u < 4; ++u) dv[h][u] = (ok && c0 + u < c) ? dgn[(int64_t)(c0 + u) * f + ic] : 0.f;
        }
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int t = 0; t < TMAX; ++t) acc[u][t] += dv[h][u] * pv[h][t];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int t = 0; t < TMAX; ++t) {
            const float v = wave_sum(acc[u][t]);
            if ((tid & 63) == 0) red[tid >> 6][u * TMAX + t] = v;
        }
    __syncthreads();
    if (tid < 4 * TMAX) {
        const int u = tid / TMAX, t = tid % TMAX;
        if (c0 + u < c && t < T) dvwb[((int64_t)n * c + c0 + u) * T + t] = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
    }
}

// The same sum as a register-operand MFMA GEMM (cf. csrc/pointwise.hip): out[t][c] = sum_f P[f][t] dg[c][f] with M = the 16 latents (half
// of a 32-row tile), N = 32 channels per wave, K = pixels.  Lane (l31, half) holds A = P[f][t = l31] and B = dg[c = l31][f] for the pixel
// f = f0 + 4 half + ks of k-step ks: four k-steps per 16-byte load of its own dg row (each lane walks ONE row: 32 rows per load
// instruction, but 16 consecutive k-steps stay inside a row's 128-byte line) and four 64-byte-coalesced dword loads of P.  The pixel axis
// is cut into `slices` (grid.x) so that one sample still fills the chip: partial sums go to a workspace [n][slices][c][16], and
// attn_values_reduce_kernel adds the slices in index order (bit-reproducible).  The VALU kernel above gave every workgroup 4 channels and
// ALL pixels: 64 workgroups for a 128^2 x 256-channel layer at one sample, each re-reading the 1 MB probability map: 150 us.
typedef float avg_f32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ void attn_values_grad_mfma_body(float* part, const float* dg, const float* probs, int c, int f, int slices, int sl,
                                                           int cbq, int n) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const int cb = cbq * 4 + wv;
    if (cb * 32 >= c) return;
    const int fs = f / slices, f0 = sl * fs;                      // pixels of this slice (host: a multiple of 8)
    const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void*)(dg + (int64_t)n * c * f), 0, c * f * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc((void*)(probs + (int64_t)n * f * TMAX), 0, f * TMAX * 4, 0x00020000);
    const int crow = cb * 32 + l31;
    const unsigned dvo = crow < c ? (unsigned)(crow * f + f0 + 4 * half) * 4u : 0xFFFFFFF0u;
    const unsigned pvo = l31 < TMAX ? (unsigned)((f0 + 4 * half) * TMAX + l31) * 4u : 0xFFFFFFF0u;     // rows t >= 16 of the A operand are zero
    avg_f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int groups = fs / 8;                                     // 8 pixels (4 k-steps x 2 halves) per group
    auto load = [&](float4& dv, float (&pa)[4], int g) {
        const __amdgpu_buffer_rsrc_t d_ = g < groups ? rd : __builtin_amdgcn_make_buffer_rsrc((void*)dg, 0, 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t p_ = g < groups ? rp : __builtin_amdgcn_make_buffer_rsrc((void*)dg, 0, 0, 0x00020000);
        dv = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(d_, dvo, g * 32, 0));
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)          // (the k-step rides in the SCALAR offset: added to the vector offset it would wrap the out-of-range mark)
            pa[ks] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(p_, pvo, (g * 8 + ks) * TMAX * 4, 0));
    };
    auto mm = [&](const float4& dv, const float (&pa)[4]) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[0], dv.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[1], dv.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[2], dv.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[3], dv.w, acc, 0, 0, 0);
    };
    float4 d0, d1, d2, d3;
    float p0[4], p1[4], p2[4], p3[4];
    load(d0, p0, 0); load(d1, p1, 1); load(d2, p2, 2);             // three groups in flight (the sweep is latency-bound, not MFMA-bound)
    for (int g = 0; g < groups; g += 4) {
        load(d3, p3, g + 3); __builtin_amdgcn_sched_barrier(0); mm(d0, p0); __builtin_amdgcn_sched_barrier(0);
        load(d0, p0, g + 4); __builtin_amdgcn_sched_barrier(0); mm(d1, p1); __builtin_amdgcn_sched_barrier(0);
        load(d1, p1, g + 5); __builtin_amdgcn_sched_barrier(0); mm(d2, p2); __builtin_amdgcn_sched_barrier(0);
        load(d2, p2, g + 6); __builtin_amdgcn_sched_barrier(0); mm(d3, p3); __builtin_amdgcn_sched_barrier(0);
    }
    // accumulator register r = latent (r & 3) + 8 (r >> 2) + 4 half (rows >= 16 are padding), column = channel crow
    if (crow < c) {
        float* o = part + (((int64_t)n * slices + sl) * c + crow) * TMAX;
        *reinterpret_cast<float4*>(o + 4 * half) = make_float4(acc[0], acc[1], acc[2], acc[3]);
        *reinterpret_cast<float4*>(o + 8 + 4 * half) = make_float4(acc[4], acc[5], acc[6], acc[7]);
    }
}

__global__ __launch_bounds__(256) void attn_values_grad_mfma_kernel(float* part, const float* dg, const float* probs, int c, int f, int slices) {
    attn_values_grad_mfma_body(part, dg, probs, c, f, slices, blockIdx.x, blockIdx.y, blockIdx.z);
}

// One record per attention layer of the deferred passes below (mgf_attn_grad_multi): the layer's value-gradient GEMM over its pixel slices,
// its demodulation dot products, then the slice sums.  blk_* = first workgroup of the layer in the flat grids.
struct AttnGradJob {
    const float* dg; const float* probs; const float* dc; const float* cpre;
    float* part; float* dvwb; float* dc_part;
    int32_t c, f, slices, nchunk, blk_grad, blk_dot, blk_red, pad_;
};
static_assert(sizeof(AttnGradJob) == 88, "AttnGradJob layout (morphganformer_amd/_lib.py mirrors it)");

// launch 1 of 2: the value-gradient GEMMs of ALL attention layers and (no dependency between them) their <dc, c> demodulation partials
__global__ __launch_bounds__(256) void attn_grad_multi_kernel(const AttnGradJob* jobs, int njobs, int n_samples, int grad_blocks) {
    __shared__ float red[4];
    const int b = blockIdx.x;
    const bool dot = b >= grad_blocks;
    const int local = dot ? b - grad_blocks : b;
    int j = 0;
    for (int q = 1; q < njobs; ++q) j = ((dot ? jobs[q].blk_dot : jobs[q].blk_grad) <= local) ? q : j;       // (tables are sorted: the last job that starts at or before)
    const AttnGradJob J = jobs[j];
    if (!dot) {
        int r = local - J.blk_grad;
        const int sl = r % J.slices; r /= J.slices;
        const int cbq4 = (int)((J.c + 127) / 128);
        const int cbq = r % cbq4, n = r / cbq4;
        if (n < n_samples) attn_values_grad_mfma_body(J.part, J.dg, J.probs, J.c, J.f, J.slices, sl, cbq, n);
    } else if (J.dc_part) {
        int r = local - J.blk_dot;
        const int chunk = r % J.nchunk; r /= J.nchunk;
        const int ch = r % J.c, n = r / J.c;
        if (n < n_samples) channel_dot_body(J.dc_part, J.dc, J.cpre, J.c, (int64_t)J.f, J.nchunk, chunk, ch, n, red);
    }
}

// launch 2 of 2: dvwb = the slices of `part` added in index order, all layers
__global__ __launch_bounds__(256) void attn_reduce_multi_kernel(const AttnGradJob* jobs, int njobs, int n_samples) {
    const int b = blockIdx.x;
    int j = 0;
    for (int q = 1; q < njobs; ++q) j = (jobs[q].blk_red <= b) ? q : j;
    const AttnGradJob J = jobs[j];
    const int64_t per_sample = (int64_t)J.c * TMAX;
    const int64_t bps = (per_sample + 255) / 256;                       // workgroups per sample
    const int64_t r = b - J.blk_red;
    const int n = (int)(r / bps);
    const int64_t i = (r % bps) * 256 + threadIdx.x;
    if (n >= n_samples || i >= per_sample) return;
    const float* pp = J.part + (int64_t)n * J.slices * per_sample + i;
    float a = 0.f;
    int s = 0;
    for (; s + 8 <= J.slices; s += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = pp[(int64_t)(s + u) * per_sample];
#pragma unroll
        for (int u = 0; u < 8; ++u) a += v[u];
    }
    for (; s < J.slices; ++s) a += pp[(int64_t)s * per_sample];
    J.dvwb[(int64_t)n * per_sample + i] = a;
}

__global__ __launch_bounds__(256) void attn_values_reduce_kernel(float* dvwb, const float* part, int64_t per_sample, int slices) {
    const int n = blockIdx.y;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < per_sample; i += (int64_t)gridDim.x * 256) {
        const float* pp = part + (int64_t)n * slices * per_sample + i;
        float a = 0.f;
        int s = 0;
        for (; s + 8 <= slices; s += 8) {            // eight independent loads per trip, summed in index order
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = pp[(int64_t)(s + u) * per_sample];
#pragma unroll
            for (int u = 0; u < 8; ++u) a += v[u];
        }
        for (; s < slices; ++s) a += pp[(int64_t)s * per_sample];
        dvwb[(int64_t)n * per_sample + i] = a;
    }
}

// ------------------------------------------------------------------------------------------ latent-side backward
// grid (njobs, n): d(style) -> d(global latent component) for one modulated layer (see the header comment)
// Sum the per-chunk partials of `rows` rows ([rows][chunks], contiguous) and hand each total to emit(row, sum).  Many chunks (the
// large maps): one wave per row, lanes across the chunks; few: one thread per row.  Fixed order either way.
template <class F>
__device__ __forceinline__ void sum_chunk_partials(const float* part, int rows, int chunks, F&& emit) {
    const int tid = threadIdx.x;
    if (chunks >= 16) {
        const int lane = tid & 63, wv = tid >> 6;
        for (int r = wv; r < rows; r += 4) {
            float a = 0.f;
            for (int q = lane; q < chunks; q += 64) a += part[(int64_t)r * chunks + q];
#pragma unroll
            for (int o = 32; o; o >>= 1) a += __shfl_xor(a, o);
            if (lane == 0) emit(r, a);
        }
    } else {
        for (int r = tid; r < rows; r += 256) {
            float a = 0.f;
            for (int q = 0; q < chunks; ++q) a += part[(int64_t)r * chunks + q];
            emit(r, a);
        }
    }
}

__device__ __forceinline__ void style_demod_bwd_body(float* dwg, const mgf_style_bwd_job& j, int job, int njobs, int wdim, float* tl, float* dst,
                                                     float* red) {
    const int n = blockIdx.y, tid = threadIdx.x;
    const bool demod = j.wsq && j.d && j.dc_part;
    if (demod)
        sum_chunk_partials(j.dc_part + (int64_t)n * j.cout * j.d_chunks, j.cout, j.d_chunks, [&](int o, float a) {
            const float dv = j.d[(int64_t)n * j.cout + o];
            tl[o] = a * dv * dv;
        });
    sum_chunk_partials(j.ds_part + (int64_t)n * j.cin * j.s_chunks, j.cin, j.s_chunks, [&](int i, float a) { dst[i] = a; });
    __syncthreads();
    if (demod) {
        // dem[i] = sum_o tl[o] * wsq[o][i]: the [cout x cin] table is read once, four input channels per lane, the output channels
        // split over as many slices as 256 lanes allow; the slices meet in LDS
        const int q4 = j.cin >> 2;
        if ((j.cin & 3) == 0 && q4 <= 256 && 256 % q4 == 0) {
            const int i4 = tid % q4, sl = tid / q4, nsl = 256 / q4;
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
            const float4* wq = (const float4*)j.wsq + i4;
#pragma unroll 8
            for (int o = sl; o < j.cout; o += nsl) {
                const float4 w = wq[(int64_t)o * q4];
                const float t = tl[o];
                a.x += t * w.x; a.y += t * w.y; a.z += t * w.z; a.w += t * w.w;
            }
            ((float4*)red)[tid] = a;                 // [nsl][q4] float4
            __syncthreads();
            for (int i = tid; i < j.cin; i += 256) {
                float dem = 0.f;
                for (int q = 0; q < nsl; ++q) dem += red[(q * q4 + (i >> 2)) * 4 + (i & 3)];
                dst[i] -= j.s[(int64_t)n * j.cin + i] * dem;
            }
        } else {
            for (int i = tid; i < j.cin; i += 256) {
                float dem = 0.f;
#pragma unroll 8
                for (int o = 0; o < j.cout; ++o) dem += tl[o] * j.wsq[(int64_t)o * j.cin + i];
                dst[i] -= j.s[(int64_t)n * j.cin + i] * dem;
            }
        }
    }
    __syncthreads();
    const int k = tid % wdim, sl = tid / wdim, nsl = 256 / wdim;
    float acc = 0.f;
    for (int i = sl; i < j.cin; i += nsl) acc += dst[i] * j.aff_w[(int64_t)i * wdim + k];
    red[tid] = acc;
    __syncthreads();
    if (tid < wdim) {
        float v = 0.f;
        for (int q = 0; q < nsl; ++q) v += red[q * wdim + tid];
        dwg[((int64_t)n * njobs + job) * wdim + tid] = v * j.aff_gain * j.style_gain;
    }
}

__global__ __launch_bounds__(256) void style_demod_bwd_kernel(float* dwg, const mgf_style_bwd_job* jobs, int njobs, int wdim) {
    __shared__ __attribute__((aligned(16))) float tl[2048];
    __shared__ __attribute__((aligned(16))) float dst[2048];
    __shared__ __attribute__((aligned(16))) float red[1024];
    style_demod_bwd_body(dwg, jobs[blockIdx.x], blockIdx.x, njobs, wdim, tl, dst, red);
}

// dyc[n, job, t, k] = sum_c dvwb[n,c,t] * wmv[c,k].  16 latents x 32 latent dimensions (every checkpoint of the drivers): the channels are
// dealt round-robin to the four waves, a lane keeps one latent row and eight columns (one dvwb load + two 16-byte weight loads per eight
// FMAs, c / 4 dependent steps instead of c: the chain of 512 dependent loads per lane was the whole 50 us), the waves meet in LDS in index
// order.  Any other shape: one output per lane over all channels.
__device__ __forceinline__ void attn_values_bwd_body(float* dyc, const mgf_attn_bwd_job& j, int job, int njobs, int T, int wdim, float* red) {
    const int n = blockIdx.y, tid = threadIdx.x;
    const float* dv = j.dvwb + (int64_t)n * j.c * T;
    float* out = dyc + ((int64_t)n * njobs + job) * T * wdim;
    if (T == 16 && wdim == 32 && ((uintptr_t)j.wmv % 16) == 0) {
        const int g = tid >> 6, l = tid & 63, t = l >> 2, k0 = (l & 3) * 8;
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
        for (int c = g; c < j.c; c += 4) {
            const float d = dv[c * 16 + t];
            const float4 w0 = *reinterpret_cast<const float4*>(j.wmv + (int64_t)c * 32 + k0), w1 = *reinterpret_cast<const float4*>(j.wmv + (int64_t)c * 32 + k0 + 4);
            acc[0] += d * w0.x; acc[1] += d * w0.y; acc[2] += d * w0.z; acc[3] += d * w0.w;
            acc[4] += d * w1.x; acc[5] += d * w1.y; acc[6] += d * w1.z; acc[7] += d * w1.w;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) red[g * 512 + t * 32 + k0 + u] = acc[u];
        __syncthreads();
        for (int i = tid; i < 512; i += 256) out[i] = ((red[i] + red[512 + i]) + red[1024 + i]) + red[1536 + i];
        return;
    }
    for (int idx = tid; idx < T * wdim; idx += 256) {
        const int t = idx / wdim, k = idx % wdim;
        float acc = 0.f;
#pragma unroll 8
        for (int c = 0; c < j.c; ++c) acc += dv[(int64_t)c * T + t] * j.wmv[(int64_t)c * wdim + k];
        out[idx] = acc;
    }
}

// grid (njobs, n)
__global__ __launch_bounds__(256) void attn_values_bwd_kernel(float* dyc, const mgf_attn_bwd_job* jobs, int njobs, int T, int wdim) {
    __shared__ float red[2048];
    attn_values_bwd_body(dyc, jobs[blockIdx.x], blockIdx.x, njobs, T, wdim, red);
}

// Both latent-side reductions of the backward pass in ONE launch, grid (style jobs + attention jobs, n): the two kernels above are
// single-workgroup latency chains of ~50 us each at one sample and do not depend on each other.
__global__ __launch_bounds__(256) void latent_bwd_multi_kernel(float* dwg, const mgf_style_bwd_job* sjobs, int n_sjobs, float* dyc,
                                                               const mgf_attn_bwd_job* ajobs, int n_ajobs, int T, int wdim) {
    __shared__ __attribute__((aligned(16))) float sm[2048 + 2048 + 1024];
    if ((int)blockIdx.x < n_sjobs) style_demod_bwd_body(dwg, sjobs[blockIdx.x], blockIdx.x, n_sjobs, wdim, sm, sm + 2048, sm + 4096);
    else attn_values_bwd_body(dyc, ajobs[blockIdx.x - n_sjobs], blockIdx.x - n_sjobs, n_ajobs, T, wdim, sm);
}

// dw[n, t < T, :] = sum_jobs dyc ; dw[n, T, :] = sum_jobs dwg       (k = T + 1 rows), times `scale`
__global__ __launch_bounds__(256) void latent_grad_gather_kernel(float* dw, const float* dwg, int njs, const float* dyc, int nja, int k,
                                                                 int wdim, float scale) {
    const int n = blockIdx.x, T = k - 1;
    for (int idx = threadIdx.x; idx < k * wdim; idx += 256) {
        const int row = idx / wdim, col = idx % wdim;
        float acc = 0.f;
        if (row < T) {
            for (int q = 0; q < nja; ++q) acc += dyc[(((int64_t)n * nja + q) * T + row) * wdim + col];
        } else {
            for (int q = 0; q < njs; ++q) acc += dwg[((int64_t)n * njs + q) * wdim + col];
        }
        dw[(int64_t)n * k * wdim + idx] = acc * scale;
    }
}

// ------------------------------------------------------------------------------------------ loss-side backward
// LPIPS tap (networks_basic.py:70-87): l = (scale / hw) * sum_p sum_c lin[c] (f0[c] q - u1[c])^2 with q = 1 / (|f0| + 1e-10).
// One lane per pixel, two sweeps over the channels:  A = sum f0^2, B = sum lin f0^2, Cc = sum lin u1 f0  give
//   <du0, f0> = k (q B - Cc),   df0[c] = q k lin[c] (f0[c] q - u1[c]) - <du0, f0> q^2 / |f0| * f0[c],   k = 2 scale / hw.
// An all-zero pixel (|f0| = 0) gets a zero gradient (torch autograd yields NaN there: sqrt'(0) * 0).
// RELU: the tap is a ReLU output (all of them are) and the ReLU's backward rides along: dz = f0 > 0 ? din + df0 : 0 (din may be
// null), channels [0, c_split) to oa [n, c_split, hw], the rest to ob [n, c - c_split, hw] (mgf_relu_bwd_split_f32's layout); without
// RELU oa is df0 [n, c, hw] and `accumulate` adds to it.
// CPT > 0: c <= CPT * (256 / PXB), and the lane keeps its CPT channels of f0 and f1 in registers between the two sweeps: 629 -> 588 us
// at 8 x 64 x 511^2 with 16 channels per lane; with 32 (128 channels) the registers cost more residency than the second sweep's
// mostly cache-resident reads (321 -> 435 us), so only CPT = 16 is built.  Wider pixel blocks (128, 256) measured no faster than 64.
template <int PXB, bool RELU, int CPT>
__global__ __launch_bounds__(256) void lpips_layer_bwd_kernel(float* oa, float* ob, const float* din, const float* f0, const float* f1u,
                                                              const float* lin, int c, int c_split, int64_t hw, int64_t f1_bs, float k,
                                                              int accumulate, const float* stats) {
    // a workgroup owns PXB consecutive pixels; its G = 256 / PXB lane groups split the channels and meet in LDS
    constexpr int G = 256 / PXB;
    constexpr int NV = CPT > 0 ? CPT : 1;
    __shared__ float part[3][G][PXB];
    const int n = blockIdx.y, px = threadIdx.x % PXB, grp = threadIdx.x / PXB;
    const int64_t p = (int64_t)blockIdx.x * PXB + px;
    const bool valid = p < hw;
    const int64_t pc = valid ? p : hw - 1;
    // Buffer addressing (round 3, like the forward kernel): a thread's channels are G planes apart, so the channel part of every address is
    // a scalar offset (j G hw) on one per-lane offset (grp hw + pixel); the range check covers both, so a channel past c -- or past
    // c_split in the first output, before it in the second -- reads 0 / is not stored, without a per-lane test.  Host: c * hw * 4 < 2^32.
    const unsigned plane_b = 4u * (unsigned)hw;
    const unsigned vo = (unsigned)grp * plane_b + 4u * (unsigned)pc;
    const unsigned cbytes = (unsigned)c * plane_b;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)(f0 + (int64_t)n * c * hw), 0, (int)cbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)(f1u + (int64_t)n * f1_bs), 0, (int)cbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rl = __builtin_amdgcn_make_buffer_rsrc((void*)lin, 0, 4 * c, 0x00020000);
    auto ld = [&](const __amdgpu_buffer_rsrc_t& r, int j) {
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, vo, (int)((unsigned)(j * G) * plane_b), 0));
    };
    auto ldlin = [&](int j) { return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rl, 4u * (unsigned)grp, 4 * j * G, 0)); };
    const int nj = (c + G - 1) / G;                   // channel steps (a lane whose channel of the last step is past c reads zeros)
    float va[NV], vb[NV];
    float A = 0.f, B = 0.f, Cc = 0.f;
    // stats: the per-pixel sums A, B, C of the forward (mgf_lpips_layer_stats_f32, [n][3][hw]) -- no first sweep, no meeting in LDS
    if (CPT == 0 && stats) {
        const float* sp = stats + (int64_t)n * 3 * hw + pc;
        A = sp[0]; B = sp[hw]; Cc = sp[2 * hw];
    } else {
    if (CPT > 0) {
#pragma unroll
        for (int j = 0; j < NV; ++j) { va[j] = ld(ra, j); vb[j] = ld(rb, j); }
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const float l = ldlin(j);
            A += va[j] * va[j];
            B += l * va[j] * va[j];
            Cc += l * vb[j] * va[j];
        }
    } else {
#pragma unroll 4
        for (int j = 0; j < nj; ++j) {
            const float v = ld(ra, j), l = ldlin(j);
            A += v * v;
            B += l * v * v;
            Cc += l * ld(rb, j) * v;
        }
    }
    part[0][grp][px] = A; part[1][grp][px] = B; part[2][grp][px] = Cc;
    __syncthreads();
    A = B = Cc = 0.f;
#pragma unroll
    for (int g = 0; g < G; ++g) { A += part[0][g][px]; B += part[1][g][px]; Cc += part[2][g][px]; }
    }
    const float nrm = sqrtf(A);
    const float q = 1.f / (nrm + 1e-10f);
    const float dot = k * (q * B - Cc);
    const float coef = nrm > 0.f ? dot * q * q / nrm : 0.f;
    // outputs: RELU -> channels [0, c_split) to oa, the rest to ob; else oa holds all c channels
    const unsigned vst = valid ? vo : 0xFFFFFFF0u;
    const int ca = RELU ? c_split : c;
    const __amdgpu_buffer_rsrc_t rda = __builtin_amdgcn_make_buffer_rsrc((void*)(oa + (int64_t)n * ca * hw), 0, (int)((unsigned)ca * plane_b), 0x00020000);
    const bool has_b = RELU && ob != nullptr && c > c_split;
    const __amdgpu_buffer_rsrc_t rdb = __builtin_amdgcn_make_buffer_rsrc((void*)(has_b ? ob + (int64_t)n * (c - c_split) * hw : oa), 0,
                                                                         has_b ? (int)((unsigned)(c - c_split) * plane_b) : 0, 0x00020000);
    const bool has_di = RELU && din != nullptr;
    const __amdgpu_buffer_rsrc_t rdi = __builtin_amdgcn_make_buffer_rsrc((void*)(has_di ? din + (int64_t)n * c * hw : f0), 0, has_di ? (int)cbytes : 0, 0x00020000);
    auto one = [&](int j, float v, float u) {
        const int so = (int)((unsigned)(j * G) * plane_b);
        float g = q * k * ldlin(j) * (v * q - u) - coef * v;
        if (RELU) {
            g = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rdi, vo, so, 0)) + g;       // (no din: zero-size descriptor, reads 0)
            g = v > 0.f ? g : 0.f;
            const int ch = grp + j * G, cb = j * G - c_split;
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, g), rda, vst, so, 0);       // ch >= c_split: past rda's end
            // second output: channel ch - c_split.  With c_split a multiple of G the step's channel base j G - c_split is wave-uniform (a
            // negative one means this step lies in the first output); otherwise the whole offset is per lane.  (The range check adds
            // vector and scalar offset WITHOUT 32-bit wrap: an offset that "wraps back" into range is still out of range.)
            const bool even = c_split % G == 0;
            const unsigned vsb = !valid ? 0xFFFFFFF0u
                               : even ? (cb >= 0 ? vo : 0xFFFFFFF0u)
                                      : (ch >= c_split ? (unsigned)(ch - c_split) * plane_b + 4u * (unsigned)pc : 0xFFFFFFF0u);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, g), rdb, vsb, (even && cb >= 0) ? (int)((unsigned)cb * plane_b) : 0, 0);
        } else {
            if (accumulate) g += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rda, vo, so, 0));
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, g), rda, vst, so, 0);
        }
    };
    if (CPT > 0) {
#pragma unroll
        for (int j = 0; j < NV; ++j) one(j, va[j], vb[j]);
    } else {
#pragma unroll 4
        for (int j = 0; j < nj; ++j) one(j, ld(ra, j), ld(rb, j));
    }
}

// dz = dy where y > 0 else 0, channels [0, cs) to dz_a [n, cs, hw] and [cs, c) to dz_b [n, c - cs, hw] (Fire concat halves)
__global__ __launch_bounds__(256) void relu_bwd_split_kernel(float* dz_a, float* dz_b, const float* dy, const float* y, int c, int cs,
                                                             int64_t hw, int64_t total) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t p = i % hw, r = i / hw;
        const int ch = (int)(r % c);
        const int64_t nn = r / c;
        const float v = y[i] > 0.f ? dy[i] : 0.f;
        if (ch < cs) dz_a[(nn * cs + ch) * hw + p] = v;
        else dz_b[(nn * (c - cs) + (ch - cs)) * hw + p] = v;
    }
}

// MaxPool2d(3, 2, ceil_mode=True) backward: the gradient of a window goes to its FIRST maximum in row-major order (torch's argmax rule).
// A workgroup owns a 32 x 32 tile of INPUT elements of one plane.  The 17 x 17 windows that touch it (one row / column of them
// belongs to the neighbouring tile and is recomputed) read a 35 x 35 input patch from LDS; every window's argmax (as a patch offset)
// and gradient are computed once, then each input element collects from the <= 4 windows that cover it.
// Input tile owned by a workgroup: 64 columns x 32 rows (both even: windows start on even rows / columns), the (32/2 + 1) x (64/2 + 1)
// windows that touch it, and their (32 + 3) x (64 + 3) input patch.  The kernel is bound by its instruction count, not by memory
// (2.1 TB/s in its first form): no coordinates are divided, the patch is padded with -inf so the window scan needs no bounds tests,
// and a lane scatters the windows' gradients to one 2 x 2 block of inputs (whose four window memberships are fixed by parity).
constexpr int MPTX = 64, MPTY = 32, MPWX = MPTX / 2 + 1, MPWY = MPTY / 2 + 1, MPIX = MPTX + 3, MPIY = MPTY + 3, MPS = MPIX + 1;
__global__ __launch_bounds__(256) void maxpool3x3s2_bwd_kernel(float* dx, const float* dy, const float* x, int ih, int iw, int oh, int ow,
                                                               int tiles_x, int tiles_y) {
    __shared__ float patch[MPIY * MPS];
    __shared__ int16_t arg[MPWY][MPWX + 1];
    __shared__ float gw[MPWY][MPWX + 1];
    const int tid = threadIdx.x;
    const int tx = blockIdx.x % tiles_x, ty = (blockIdx.x / tiles_x) % tiles_y;
    const int64_t pl = blockIdx.x / ((int64_t)tiles_x * tiles_y);
    const int iy0 = ty * MPTY, ix0 = tx * MPTX;          // first owned input row / column
    const int py0 = iy0 - 2, px0 = ix0 - 2;              // patch origin: patch[2 wy + ky][2 wx + kx] is tap (ky, kx) of local window (wy, wx)
    const int wy0 = iy0 / 2 - 1, wx0 = ix0 / 2 - 1;      // first window row / column (may be -1)
    const float* xp = x + pl * ih * iw;
    const float* dp = dy + pl * oh * ow;
    const float NEG = -__builtin_inff();
    {
        const int cc = tid & 63, r0 = tid >> 6;
        const int xx = px0 + cc;
        const bool xin = xx >= 0 && xx < iw;
#pragma unroll
        for (int r = r0; r < MPIY; r += 4) {
            const int yy = py0 + r;
            patch[r * MPS + cc] = (xin && yy >= 0 && yy < ih) ? xp[(int64_t)yy * iw + xx] : NEG;
        }
        if (tid < 3 * MPIY) {                            // the three columns past the 64th
            const int r = tid / 3, c3 = 64 + tid - 3 * r;
            const int yy = py0 + r, x3 = px0 + c3;
            patch[r * MPS + c3] = (x3 < iw && yy >= 0 && yy < ih) ? xp[(int64_t)yy * iw + x3] : NEG;
        }
    }
    __syncthreads();
    for (int i = tid; i < MPWY * MPWX; i += 256) {
        const int wy = i / MPWX, wx = i - wy * MPWX;
        const int oy = wy0 + wy, ox = wx0 + wx;
        int best = -1;
        float g = 0.f;
        if (oy >= 0 && ox >= 0 && oy < oh && ox < ow) {  // tap (0, 0) of a window that exists is inside the input
            const int base = 2 * wy * MPS + 2 * wx;
            float bv = patch[base];
            best = base;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    if (ky == 0 && kx == 0) continue;
                    const float v = patch[base + ky * MPS + kx];
                    if (v > bv) { bv = v; best = base + ky * MPS + kx; }       // first maximum in row-major order (torch)
                }
            g = dp[(int64_t)oy * ow + ox];
        }
        arg[wy][wx] = (int16_t)best;
        gw[wy][wx] = g;
    }
    __syncthreads();
    // input block (2 br, 2 bc) + {0,1}^2 of the tile: windows (br, bc) .. (br + 1, bc + 1) in local indices; element (0,0) is in
    // all four, (0,1) in the right two, (1,0) in the lower two, (1,1) in the last -- summed in ascending window order like torch
#pragma unroll
    for (int i = tid; i < (MPTY / 2) * (MPTX / 2); i += 256) {
        const int br = i / (MPTX / 2), bc = i - br * (MPTX / 2);
        const int yy = iy0 + 2 * br, xx = ix0 + 2 * bc;
        if (yy >= ih || xx >= iw) continue;
        const int me = (2 * br + 2) * MPS + 2 * bc + 2;
        const int a00 = arg[br][bc], a01 = arg[br][bc + 1], a10 = arg[br + 1][bc], a11 = arg[br + 1][bc + 1];
        const float g00 = gw[br][bc], g01 = gw[br][bc + 1], g10 = gw[br + 1][bc], g11 = gw[br + 1][bc + 1];
        float e00 = 0.f, e01 = 0.f, e10 = 0.f, e11 = 0.f;
        if (a00 == me) e00 += g00;
        if (a01 == me) e00 += g01;
        if (a10 == me) e00 += g10;
        if (a11 == me) e00 += g11;
        if (a01 == me + 1) e01 += g01;
        if (a11 == me + 1) e01 += g11;
        if (a10 == me + MPS) e10 += g10;
        if (a11 == me + MPS) e10 += g11;
        if (a11 == me + MPS + 1) e11 += g11;
        float* o = dx + pl * ih * iw + (int64_t)yy * iw + xx;
        const bool x1 = xx + 1 < iw;
        o[0] = e00;
        if (x1) o[1] = e01;
        if (yy + 1 < ih) {
            o[iw] = e10;
            if (x1) o[iw + 1] = e11;
        }
    }
}

// MaxPool2d(2, 2) backward (VGG): windows do not overlap, an input element belongs to at most one
__global__ __launch_bounds__(256) void maxpool2x2s2_bwd_kernel(float* dx, const float* dy, const float* x, int ih, int iw, int oh, int ow,
                                                               int64_t total) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int ix = (int)(i % iw);
        const int64_t r = i / iw;
        const int iy = (int)(r % ih);
        const int64_t pl = r / ih;
        const int oy = iy / 2, ox = ix / 2;
        float g = 0.f;
        if (oy < oh && ox < ow) {
            const float* xp = x + pl * ih * iw + (int64_t)(2 * oy) * iw + 2 * ox;
            const float v00 = xp[0], v01 = xp[1], v10 = xp[iw], v11 = xp[iw + 1];
            int best = 0;
            float bv = v00;
            if (v01 > bv) { bv = v01; best = 1; }
            if (v10 > bv) { bv = v10; best = 2; }
            if (v11 > bv) { bv = v11; best = 3; }
            if (best == (iy & 1) * 2 + (ix & 1)) g = dy[pl * oh * ow + (int64_t)oy * ow + ox];
        }
        dx[i] = g;
    }
}

// dimg (+)= scale * 2 (a - b) / numel    (gradient of scale * mean((a - b)^2) per sample)
__global__ __launch_bounds__(256) void mse_grad_kernel(float* d, const float* a, const float* b, int64_t numel, int64_t b_bs, float k,
                                                       int accumulate, int64_t total) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t nn = i / numel, p = i % numel;
        const float g = k * (a[i] - b[nn * b_bs + p]);
        d[i] = accumulate ? d[i] + g : g;
    }
}

// torch.optim.Adam (single tensor, amsgrad off, weight_decay optional as L2 like torch) on a small parameter, one workgroup;
// the optimizer's own step count lives on the device (it only advances when a step is taken, like optimizer.step()).
__global__ __launch_bounds__(256) void adam_step_kernel(float* param, float* m, float* v, int32_t* t_ctr, const float* grad, const float* lr_table,
                                                        const int32_t* step, const int32_t* valid, int64_t numel, int steps_total, float beta1,
                                                        float beta2, float eps, float weight_decay) {
    const int s = *step;
    if (s >= steps_total || (valid && valid[s] == 0)) return;               // past the end / "no face": `continue` before optimizer.step()
    const int t = *t_ctr + 1;
    __syncthreads();
    const double bc1 = 1.0 - pow((double)beta1, (double)t), bc2 = 1.0 - pow((double)beta2, (double)t);
    const float step_size = (float)((double)lr_table[s] / bc1);
    const float bc2_sqrt = (float)sqrt(bc2);
    for (int64_t i = threadIdx.x; i < numel; i += 256) {
        float g = grad[i];
        if (weight_decay != 0.f) g += weight_decay * param[i];
        const float mi = m[i] + (g - m[i]) * (1.f - beta1);                 // exp_avg.lerp_(grad, 1 - beta1)
        const float vi = v[i] * beta2 + (1.f - beta2) * g * g;
        m[i] = mi;
        v[i] = vi;
        param[i] -= step_size * (mi / (sqrtf(vi) / bc2_sqrt + eps));
    }
    if (threadIdx.x == 0) *t_ctr = t;
}

}  // namespace

extern "C" int32_t mgf_bwd_chunks(int64_t hw) { return (int32_t)mgf_cdiv(hw, BWD_CHUNK); }

extern "C" int mgf_layer_act_bwd_f32(float* dz, float* dot_part, const float* dy, const float* y, const float* residual, const float* bias,
                                     const float* noise, const float* noise_strength, int32_t noise_n, int32_t n, int32_t c, int64_t hw,
                                     float alpha, float gain, mgf_stream_t stream) {
    return mgf_layer_act_bwd_low_f32(dz, dot_part, dy, y, residual, nullptr, 0, bias, noise, noise_strength, noise_n, n, c, hw, alpha, gain, stream);
}

extern "C" int mgf_layer_act_bwd_low_f32(float* dz, float* dot_part, const float* dy, const float* y, const float* residual,
                                         const float* residual_low, int32_t w, const float* bias, const float* noise,
                                         const float* noise_strength, int32_t noise_n, int32_t n, int32_t c, int64_t hw, float alpha, float gain,
                                         mgf_stream_t stream) {
    MGF_REQUIRE(dz && dy && y && n >= 1 && c >= 1 && hw >= 1, MGF_EINVAL, "layer_act_bwd: bad arguments");
    MGF_REQUIRE(alpha != 0.f && gain != 0.f, MGF_EINVAL, "layer_act_bwd: alpha and gain must be non-zero (the activation is inverted)");
    MGF_REQUIRE(n <= 65535 && c <= 65535, MGF_ETOOBIG, "layer_act_bwd: n and c must be <= 65535");
    MGF_REQUIRE(!(residual && residual_low), MGF_EINVAL, "layer_act_bwd: the residual comes at full OR at half resolution");
    ActBwdParams p{dz, dot_part, dy, y, residual, bias, noise, noise_strength, noise_n, c, (int)mgf_cdiv(hw, BWD_CHUNK), hw, alpha, gain,
                   residual_low, w};
    auto al16 = [](const void* q) { return q == nullptr || ((uintptr_t)q % 16) == 0; };
    const bool vec = hw % 4 == 0 && al16(dz) && al16(dy) && al16(y) && al16(residual) && al16(noise);
    if (residual_low) {
        MGF_REQUIRE(vec && w >= 4 && w % 4 == 0 && hw % w == 0 && (hw / w) % 2 == 0 && hw / w <= INT32_MAX, MGF_EUNSUPPORTED,
                    "layer_act_bwd: the half-resolution residual needs 16-byte aligned maps with w %% 4 == 0 and an even height (w %d, hw %lld)", w, (long long)hw);
        hipLaunchKernelGGL((act_bwd_kernel<true, true>), dim3(p.nchunk, c, n), dim3(256), 0, (hipStream_t)stream, p);
    } else if (vec) hipLaunchKernelGGL((act_bwd_kernel<true>), dim3(p.nchunk, c, n), dim3(256), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((act_bwd_kernel<false>), dim3(p.nchunk, c, n), dim3(256), 0, (hipStream_t)stream, p);
    MGF_CHECK_LAUNCH("layer_act_bwd");
    return MGF_OK;
}

extern "C" int mgf_channel_dot_f32(float* dot_part, const float* a, const float* b, int32_t n, int32_t c, int64_t hw, mgf_stream_t stream) {
    MGF_REQUIRE(dot_part && a && b && n >= 1 && c >= 1 && hw >= 1, MGF_EINVAL, "channel_dot: bad arguments");
    MGF_REQUIRE(n <= 65535 && c <= 65535, MGF_ETOOBIG, "channel_dot: n and c must be <= 65535");
    const int nchunk = (int)mgf_cdiv(hw, BWD_CHUNK);
    hipLaunchKernelGGL(channel_dot_kernel, dim3(nchunk, c, n), dim3(256), 0, (hipStream_t)stream, dot_part, a, b, c, hw, nchunk);
    MGF_CHECK_LAUNCH("channel_dot");
    return MGF_OK;
}

extern "C" int mgf_style_grad_f32(float* dot_part, float* dx, const float* x, const float* g, const float* s, int32_t n, int32_t c,
                                  int64_t hw, int32_t accumulate, mgf_stream_t stream) {
    MGF_REQUIRE(dot_part && dx && x && g && n >= 1 && c >= 1 && hw >= 1, MGF_EINVAL, "style_grad: bad arguments");
    MGF_REQUIRE(n <= 65535 && c <= 65535, MGF_ETOOBIG, "style_grad: n and c must be <= 65535");
    const int nchunk = (int)mgf_cdiv(hw, BWD_CHUNK);
    const bool vec = hw % 4 == 0 && ((uintptr_t)dx % 16) == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)g % 16) == 0;
    if (vec) hipLaunchKernelGGL(style_grad_kernel<true>, dim3(nchunk, c, n), dim3(256), 0, (hipStream_t)stream, dot_part, dx, x, g, s, c, hw,
                                nchunk, accumulate);
    else hipLaunchKernelGGL(style_grad_kernel<false>, dim3(nchunk, c, n), dim3(256), 0, (hipStream_t)stream, dot_part, dx, x, g, s, c, hw,
                            nchunk, accumulate);
    MGF_CHECK_LAUNCH("style_grad");
    return MGF_OK;
}

// style_grad of layer L+1 fused with the activation backward of layer L, whose OUTPUT is layer L+1's input x (the conv1 -> conv0 pair of a
// synthesis block): one pass over (x, g) gives  part_s = <x, g>,  d = s g (never stored),  dz = d gain (x > 0 ? 1 : alpha)  and, when
// part_dc is given,  part_dc = <dz, c>  with c recovered by inverting the activation -- exactly the arithmetic of the two kernels run one
// after the other (d is rounded to float32 before it is used), at 3 tensor passes instead of 6.  Layer L must have no residual.
struct StyleActParams {
    float* part_s; float* part_dc; float* dz; float* dx;
    const float* x; const float* g; const float* s; const float* res; const float* bias; const float* noise; const float* nstr;
    int noise_n, c, nchunk;
    int64_t hw;
    float alpha, gain;
    const float* res_low;   // LOW: the residual at half resolution (res is null), see ActBwdParams
    int w;
};
namespace {
// RES: the earlier layer's output carries a residual (x = lrelu(..) gain + res: the activation is inverted on x - res) and s g is
// also stored (dx: the block's skip branch reads it).
template <bool VEC, bool RES, bool LOW = false>
__global__ __launch_bounds__(256) void style_grad_act_bwd_kernel(StyleActParams p) {
    static_assert(!LOW || (VEC && RES), "the half-resolution residual rides on the 16-byte residual path");
    __shared__ float red[4];
    const int chunk = blockIdx.x, ch = blockIdx.y, n = blockIdx.z;
    const int64_t base = ((int64_t)n * p.c + ch) * p.hw;
    const int64_t i0 = (int64_t)chunk * BWD_CHUNK;
    const int64_t i1 = min(p.hw, i0 + (int64_t)BWD_CHUNK);
    const float sv = p.s ? p.s[(int64_t)n * p.c + ch] : 1.f;
    const float b = p.bias ? p.bias[ch] : 0.f;
    const float ns = p.noise ? (p.nstr ? *p.nstr : 1.f) : 0.f;
    const float* nz = p.noise ? p.noise + (int64_t)(p.noise_n > 1 ? n : 0) * p.hw : nullptr;
    const float inv_gain = 1.f / p.gain, inv_alpha = 1.f / p.alpha;
    float acc_s = 0.f, acc_c = 0.f;
    auto one = [&](float xv, float rv, float gv, float nv, float& dv, float& dzv) {
        dv = sv * gv;                                // (the value style_grad would have stored)
        const float v = RES ? xv - rv : xv;
        const bool pos = v > 0.f;
        dzv = dv * p.gain * (pos ? 1.f : p.alpha);
        if (p.part_dc) {
            const float zv = (pos ? v : v * inv_alpha) * inv_gain;
            acc_c += dzv * (zv - b - nv * ns);
        }
    };
    if (VEC) {
        for (int64_t i = i0 + 4 * threadIdx.x; i < i1; i += 1024) {
            const float4 gv = *reinterpret_cast<const float4*>(p.g + base + i);
            const float4 xv = *reinterpret_cast<const float4*>(p.x + base + i);
            const float4 nv = nz ? *reinterpret_cast<const float4*>(nz + i) : make_float4(0.f, 0.f, 0.f, 0.f);
            float4 rv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (LOW) {
                const int yy = (int)(i / p.w);
                rv = up2_residual4(p.res_low + ((int64_t)n * p.c + ch) * (p.hw >> 2), (int)(p.hw / p.w) >> 1, p.w >> 1, yy, (int)(i - (int64_t)yy * p.w));
            } else if (RES) rv = *reinterpret_cast<const float4*>(p.res + base + i);
            acc_s += xv.x * gv.x + xv.y * gv.y + xv.z * gv.z + xv.w * gv.w;       // (the association of style_grad_kernel: bit-identical partials)
            float4 o, d;
            one(xv.x, rv.x, gv.x, nv.x, d.x, o.x); one(xv.y, rv.y, gv.y, nv.y, d.y, o.y);
            one(xv.z, rv.z, gv.z, nv.z, d.z, o.z); one(xv.w, rv.w, gv.w, nv.w, d.w, o.w);
            *reinterpret_cast<float4*>(p.dz + base + i) = o;
            if (RES) *reinterpret_cast<float4*>(p.dx + base + i) = d;
        }
    } else {
        for (int64_t i = i0 + threadIdx.x; i < i1; i += 256) {
            float o, d;
            const float xs = p.x[base + i], gs = p.g[base + i];
            acc_s += xs * gs;
            one(xs, RES ? p.res[base + i] : 0.f, gs, nz ? nz[i] : 0.f, d, o);
            p.dz[base + i] = o;
            if (RES) p.dx[base + i] = d;
        }
    }
    const float ts = block_sum(acc_s, red);
    if (threadIdx.x == 0) p.part_s[((int64_t)n * p.c + ch) * p.nchunk + chunk] = ts;
    if (p.part_dc) {
        __syncthreads();
        const float tc = block_sum(acc_c, red);
        if (threadIdx.x == 0) p.part_dc[((int64_t)n * p.c + ch) * p.nchunk + chunk] = tc;
    }
}
}  // namespace

extern "C" int mgf_style_grad_act_bwd_f32(float* style_part, float* dot_part, float* dz, float* dx, const float* x, const float* g,
                                          const float* s, const float* residual, const float* residual_low, int32_t w, const float* bias,
                                          const float* noise, const float* noise_strength, int32_t noise_n, int32_t n, int32_t c, int64_t hw,
                                          float alpha, float gain, mgf_stream_t stream) {
    MGF_REQUIRE(style_part && dz && x && g && n >= 1 && c >= 1 && hw >= 1, MGF_EINVAL, "style_grad_act_bwd: bad arguments");
    MGF_REQUIRE(!(residual && residual_low), MGF_EINVAL, "style_grad_act_bwd: the residual comes at full OR at half resolution");
    MGF_REQUIRE((dx != nullptr) == (residual != nullptr || residual_low != nullptr), MGF_EINVAL,
                "style_grad_act_bwd: dx and a residual go together (both or neither)");
    MGF_REQUIRE(alpha != 0.f && gain != 0.f, MGF_EINVAL, "style_grad_act_bwd: alpha and gain must be non-zero (the activation is inverted)");
    MGF_REQUIRE(n <= 65535 && c <= 65535, MGF_ETOOBIG, "style_grad_act_bwd: n and c must be <= 65535");
    StyleActParams p{style_part, dot_part, dz, dx, x, g, s, residual, bias, noise, noise_strength, noise_n, c, (int)mgf_cdiv(hw, BWD_CHUNK),
                     hw, alpha, gain, residual_low, w};
    auto al16 = [](const void* q) { return q == nullptr || ((uintptr_t)q % 16) == 0; };
    const bool vec = hw % 4 == 0 && al16(dz) && al16(x) && al16(g) && al16(noise) && al16(dx) && al16(residual);
    const dim3 grid(p.nchunk, c, n);
    if (residual_low) {
        MGF_REQUIRE(vec && w >= 4 && w % 4 == 0 && hw % w == 0 && (hw / w) % 2 == 0 && hw / w <= INT32_MAX, MGF_EUNSUPPORTED,
                    "style_grad_act_bwd: the half-resolution residual needs 16-byte aligned maps with w %% 4 == 0 and an even height (w %d, hw %lld)", w, (long long)hw);
        hipLaunchKernelGGL((style_grad_act_bwd_kernel<true, true, true>), grid, dim3(256), 0, (hipStream_t)stream, p);
    } else if (residual) {
        if (vec) hipLaunchKernelGGL((style_grad_act_bwd_kernel<true, true>), grid, dim3(256), 0, (hipStream_t)stream, p);
        else hipLaunchKernelGGL((style_grad_act_bwd_kernel<false, true>), grid, dim3(256), 0, (hipStream_t)stream, p);
    } else {
        if (vec) hipLaunchKernelGGL((style_grad_act_bwd_kernel<true, false>), grid, dim3(256), 0, (hipStream_t)stream, p);
        else hipLaunchKernelGGL((style_grad_act_bwd_kernel<false, false>), grid, dim3(256), 0, (hipStream_t)stream, p);
    }
    MGF_CHECK_LAUNCH("style_grad_act_bwd");
    return MGF_OK;
}

// The pair above followed by the blur's gradient, in one pass: layer L is an up-sampling layer without attention (256^2 and larger), so its
// dz goes through nothing but the adjoint of the 4x4 resample filter (upfirdn2d with pad 2: the [h+1, w+1] map the stride-2 dgrad convolution
// reads) -- dz itself is never needed in memory.  A workgroup produces a 64 x 64 tile of that map from a 67-row x 72-column window of (x, g):
// dz is computed element-wise while the window is staged into LDS (every element of the window; the two dot products only over the elements
// the tile OWNS, rows / columns [tile origin, + 64): every element of the plane is owned by exactly one tile), then the separable blur of
// csrc/upfirdn2d.hip's fir_up1_sep runs on it.  Traffic per element: 2 x 4 B x 1.18 in, 4 B out, against 12 B + 8.7 B for the two launches.
struct StyleActFirParams {
    float* part_s; float* part_dc; float* dt;
    const float* x; const float* g; const float* s; const float* bias; const float* noise; const float* nstr; const float* f;
    int noise_n, n, c, h, w, flip, ntiles, tiles_x;
    float alpha, gain, fir_gain;
};
namespace {
__global__ __launch_bounds__(256) void style_act_fir_bwd_kernel(StyleActFirParams p) {
    constexpr int T = 64, WV = 18, IH = T + 3, RS = T + 4;
    __shared__ float4 sx[IH][WV + 1];
    __shared__ float sfx[4], sfy[4];
    __shared__ float red[4];
    const int tid = threadIdx.x;
    if (tid < 4) {                                                   // f[jy][jx] = fy[jy] * fx[jx]; the gain goes with fy (fir_up1_sep)
        const int k = p.flip ? tid : 3 - tid;
        const float f00 = p.f[0];
        sfx[tid] = p.f[k] / f00;
        sfy[tid] = p.f[k * 4] * p.fir_gain;
    }
    const int tile = blockIdx.x, ch = blockIdx.y, n = blockIdx.z;
    const int ox0 = (tile % p.tiles_x) * T, oy0 = (tile / p.tiles_x) * T;
    const int out_h = p.h + 1, out_w = p.w + 1;
    const int64_t plane = (int64_t)p.h * p.w;
    const int64_t base = ((int64_t)n * p.c + ch) * plane;
    const float sv = p.s ? p.s[(int64_t)n * p.c + ch] : 1.f;
    const float b = p.bias ? p.bias[ch] : 0.f;
    const float ns = p.noise ? (p.nstr ? *p.nstr : 1.f) : 0.f;
    const float* nz = p.noise ? p.noise + (int64_t)(p.noise_n > 1 ? n : 0) * plane : nullptr;
    const float inv_gain = 1.f / p.gain, inv_alpha = 1.f / p.alpha;
    float acc_s = 0.f, acc_c = 0.f;
    const int xa = ox0 - 4, iy0 = oy0 - 2;                          // pad 2: output (oy, ox) reads input rows oy - 2 .. oy + 1
    for (int i = tid; i < IH * WV; i += 256) {
        const int r = i / WV, v4 = i - r * WV;
        const int iy = iy0 + r, ix = xa + 4 * v4;
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
        if (iy >= 0 && iy < p.h && ix >= 0 && ix < p.w) {            // (w % 4 == 0: a float4 is inside the row or outside it)
            const int64_t off = (int64_t)iy * p.w + ix;
            const float4 gv = *reinterpret_cast<const float4*>(p.g + base + off);
            const float4 xv = *reinterpret_cast<const float4*>(p.x + base + off);
            const float4 nv = nz ? *reinterpret_cast<const float4*>(nz + off) : make_float4(0.f, 0.f, 0.f, 0.f);
            const bool own = iy >= oy0 && iy < oy0 + T && ix >= ox0 && ix < ox0 + T;
            const float xs[4] = {xv.x, xv.y, xv.z, xv.w}, gs[4] = {gv.x, gv.y, gv.z, gv.w}, nn[4] = {nv.x, nv.y, nv.z, nv.w};
            float dz[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float dv = sv * gs[e];                         // (what style_grad would have stored: rounded before it is used)
                const bool pos = xs[e] > 0.f;
                dz[e] = dv * p.gain * (pos ? 1.f : p.alpha);
                if (own) {
                    acc_s += xs[e] * gs[e];
                    if (p.part_dc) {
                        const float zv = (pos ? xs[e] : xs[e] * inv_alpha) * inv_gain;
                        acc_c += dz[e] * (zv - b - nn[e] * ns);
                    }
                }
            }
            o = make_float4(dz[0], dz[1], dz[2], dz[3]);
        }
        sx[r][v4] = o;
    }
    __syncthreads();
    const int lx = tid & 15, ly = tid >> 4;
    const float fx0 = sfx[0], fx1 = sfx[1], fx2 = sfx[2], fx3 = sfx[3];
    float hz[7][4];                                                  // horizontal pass: rows 4 ly .. 4 ly + 6, outputs e = 0 .. 3
#pragma unroll
    for (int r = 0; r < 7; ++r) {
        const float4 a = sx[4 * ly + r][lx], bq = sx[4 * ly + r][lx + 1], cc = sx[4 * ly + r][lx + 2];
        const float w[12] = {a.x, a.y, a.z, a.w, bq.x, bq.y, bq.z, bq.w, cc.x, cc.y, cc.z, cc.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) hz[r][e] = w[2 + e] * fx0 + w[3 + e] * fx1 + w[4 + e] * fx2 + w[5 + e] * fx3;
    }
    const float fy0 = sfy[0], fy1 = sfy[1], fy2 = sfy[2], fy3 = sfy[3];
    // rows of w + 1 floats start at any alignment: the tile goes back through LDS and leaves as whole 256-byte row segments
    float* so = reinterpret_cast<float*>(&sx[0][0]);
    static_assert(T * RS <= IH * (WV + 1) * 4, "output tile must fit in the input window's LDS");
    __syncthreads();                                                 // every lane has read its window rows
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
    if (ox0 + col < out_w) {
        float* yb = p.dt + ((int64_t)n * p.c + ch) * ((int64_t)out_h * out_w) + ox0 + col;
#pragma unroll
        for (int k = 0; k < T / 4; ++k) {
            const int row = r0 + 4 * k;
            if (oy0 + row < out_h) yb[(int64_t)(oy0 + row) * out_w] = so[row * RS + col];
        }
    }
    const float ts = block_sum(acc_s, red);
    if (tid == 0) p.part_s[((int64_t)n * p.c + ch) * p.ntiles + tile] = ts;
    if (p.part_dc) {
        __syncthreads();
        const float tc = block_sum(acc_c, red);
        if (tid == 0) p.part_dc[((int64_t)n * p.c + ch) * p.ntiles + tile] = tc;
    }
}
}  // namespace

extern "C" int32_t mgf_style_act_fir_tiles(int32_t h, int32_t w) { return (int32_t)(mgf_cdiv(h + 1, 64) * mgf_cdiv(w + 1, 64)); }

extern "C" int mgf_style_act_fir_bwd_f32(float* style_part, float* dot_part, float* dt, const float* x, const float* g, const float* s,
                                         const float* bias, const float* noise, const float* noise_strength, int32_t noise_n, const float* f,
                                         int32_t flip, float fir_gain, int32_t n, int32_t c, int32_t h, int32_t w, float alpha, float gain,
                                         mgf_stream_t stream) {
    MGF_REQUIRE(style_part && dt && x && g && f && n >= 1 && c >= 1 && h >= 1 && w >= 4, MGF_EINVAL, "style_act_fir_bwd: bad arguments");
    MGF_REQUIRE(alpha != 0.f && gain != 0.f, MGF_EINVAL, "style_act_fir_bwd: alpha and gain must be non-zero (the activation is inverted)");
    MGF_REQUIRE(w % 4 == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)g % 16) == 0 && (!noise || ((uintptr_t)noise % 16) == 0), MGF_EUNSUPPORTED,
                "style_act_fir_bwd: needs 16-byte aligned maps with w %% 4 == 0 (w = %d)", w);
    MGF_REQUIRE(n <= 65535 && c <= 65535, MGF_ETOOBIG, "style_act_fir_bwd: n and c must be <= 65535");
    StyleActFirParams p{style_part, dot_part, dt, x, g, s, bias, noise, noise_strength, f, noise_n, n, c, h, w, flip & 1,
                        mgf_style_act_fir_tiles(h, w), (int)mgf_cdiv(w + 1, 64), alpha, gain, fir_gain};
    hipLaunchKernelGGL(style_act_fir_bwd_kernel, dim3(p.ntiles, c, n), dim3(256), 0, (hipStream_t)stream, p);
    MGF_CHECK_LAUNCH("style_act_fir_bwd");
    return MGF_OK;
}

extern "C" int mgf_duplex_attention_bwd(float* dx, float* dg, float* probs, const float* da, const float* x, const float* wqc,
                                        const float* spos, const float* vwb, int32_t n, int32_t c, int32_t f, int32_t t,
                                        mgf_stream_t stream) {
    MGF_REQUIRE(dx && da && x && wqc && spos && vwb, MGF_EINVAL, "duplex_attention_bwd: null pointer");
    MGF_REQUIRE(n >= 1 && c >= 1 && f >= 1, MGF_EINVAL, "duplex_attention_bwd: bad shape");
    MGF_REQUIRE(t >= 1 && t <= TMAX, MGF_EUNSUPPORTED, "duplex_attention_bwd: supports 1..%d latent components (got %d)", TMAX, t);
    MGF_REQUIRE(n <= 65535 && (int64_t)n * c * f <= INT32_MAX, MGF_ETOOBIG, "duplex_attention_bwd: tensor too large");
    constexpr int PXB = 16, G = 256 / PXB;
    AttnBwdParams p{dx, dg, probs, da, x, wqc, spos, vwb, n, c, f, t, (int)(mgf_cdiv(c, 4) * 4)};
    // the generator's layers: the MFMA form (MGF_ATTN_BWD_MFMA=0 keeps the register kernel: tuning hook, tests)
    static const char* mf_env = mgf_knob("MGF_ATTN_BWD_MFMA");
    if (t == TMAX && (c == 256 || c == 512) && f % 32 == 0 && !(mf_env && mf_env[0] == '0')) {
        if (c == 256) hipLaunchKernelGGL((duplex_attention_bwd_mfma_kernel<4>), dim3((unsigned)(f / 32), n), dim3(256), 0, (hipStream_t)stream, p);
        else hipLaunchKernelGGL((duplex_attention_bwd_mfma_kernel<8>), dim3((unsigned)(f / 32), n), dim3(512), 0, (hipStream_t)stream, p);
        MGF_CHECK_LAUNCH("duplex_attention_bwd");
        return MGF_OK;
    }
    const size_t lds = ((size_t)2 * p.c_pad * TMAX + (size_t)G * (TMAX + 1) * PXB) * sizeof(float);
    MGF_REQUIRE(lds <= 150 * 1024, MGF_EUNSUPPORTED, "duplex_attention_bwd: %d channels need %zu bytes of LDS", c, lds);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)duplex_attention_bwd_kernel<PXB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) { mgf_set_error("duplex_attention_bwd: cannot raise dynamic LDS to %zu: %s", lds, hipGetErrorString(e)); return MGF_ELAUNCH; }
    }
    hipLaunchKernelGGL(duplex_attention_bwd_kernel<PXB>, dim3((unsigned)mgf_cdiv(f, PXB), n), dim3(256), lds, (hipStream_t)stream, p);
    MGF_CHECK_LAUNCH("duplex_attention_bwd");
    return MGF_OK;
}

extern "C" int mgf_attn_values_grad(float* dvwb, const float* dg, const float* probs, int32_t n, int32_t c, int32_t f, int32_t t,
                                    mgf_stream_t stream) {
    MGF_REQUIRE(dvwb && dg && probs && n >= 1 && c >= 1 && f >= 1, MGF_EINVAL, "attn_values_grad: bad arguments");
    MGF_REQUIRE(t >= 1 && t <= TMAX, MGF_EUNSUPPORTED, "attn_values_grad: supports 1..%d latent components (got %d)", TMAX, t);
    MGF_REQUIRE(n <= 65535, MGF_ETOOBIG, "attn_values_grad: n must be <= 65535");
    hipLaunchKernelGGL(attn_values_grad_kernel, dim3((unsigned)mgf_cdiv(c, 4), n), dim3(256), 0, (hipStream_t)stream, dvwb, dg, probs, c, f, t);
    MGF_CHECK_LAUNCH("attn_values_grad");
    return MGF_OK;
}

extern "C" int64_t mgf_attn_values_grad_workspace_floats(int32_t n, int32_t c) {
    return (int64_t)128 * (n > 0 ? n : 1) * (c > 0 ? c : 1) * TMAX;         // at most 128 pixel slices
}

// pixel slices of the MFMA value-gradient form for one layer (0: the layer's shape does not take that form)
static int attn_grad_slices(int n, int c, int f, int t) {
    int slices = 1;
    const int cblocks = (int)mgf_cdiv(c, 32);
    while (slices < 128 && (int64_t)n * slices * cblocks < 1024 && f % (slices * 2 * 8) == 0 && f / (slices * 2) >= 32) slices *= 2;
    const bool ok = t == TMAX && f % (8 * slices) == 0 && (int64_t)c * f * 4 < (1LL << 31);
    return ok ? slices : 0;
}

extern "C" int32_t mgf_attn_values_grad_slices(int32_t n, int32_t c, int32_t f, int32_t t) { return attn_grad_slices(n, c, f, t); }

extern "C" int64_t mgf_attn_grad_job_bytes(void) { return (int64_t)sizeof(AttnGradJob); }

extern "C" int mgf_attn_grad_multi(const void* jobs_dev, int32_t njobs, int32_t n, int32_t grad_blocks, int32_t dot_blocks, int32_t reduce_blocks,
                                   mgf_stream_t stream) {
    MGF_REQUIRE(jobs_dev && njobs >= 1 && njobs <= 64 && n >= 1 && grad_blocks >= 1 && dot_blocks >= 0 && reduce_blocks >= 1, MGF_EINVAL,
                "attn_grad_multi: bad arguments");
    MGF_REQUIRE(((uintptr_t)jobs_dev % 8) == 0, MGF_EINVAL, "attn_grad_multi: the job table must be 8-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(attn_grad_multi_kernel, dim3(grad_blocks + dot_blocks), dim3(256), 0, st, reinterpret_cast<const AttnGradJob*>(jobs_dev), njobs, n,
                       grad_blocks);
    hipLaunchKernelGGL(attn_reduce_multi_kernel, dim3(reduce_blocks), dim3(256), 0, st, reinterpret_cast<const AttnGradJob*>(jobs_dev), njobs, n);
    MGF_CHECK_LAUNCH("attn_grad_multi");
    return MGF_OK;
}

extern "C" int mgf_attn_values_grad_ws(float* dvwb, const float* dg, const float* probs, int32_t n, int32_t c, int32_t f, int32_t t,
                                       float* workspace, int64_t workspace_floats, mgf_stream_t stream) {
    MGF_REQUIRE(dvwb && dg && probs && n >= 1 && c >= 1 && f >= 1, MGF_EINVAL, "attn_values_grad: bad arguments");
    MGF_REQUIRE(t >= 1 && t <= TMAX, MGF_EUNSUPPORTED, "attn_values_grad: supports 1..%d latent components (got %d)", TMAX, t);
    MGF_REQUIRE(n <= 65535, MGF_ETOOBIG, "attn_values_grad: n must be <= 65535");
    // the MFMA form: 16 latents, pixel count a multiple of 8, 32-bit offsets inside a sample; pixel slices so that ~1024 waves run
    const int slices = attn_grad_slices(n, c, f, t);
    const int cblocks = (int)mgf_cdiv(c, 32);
    const bool ok = slices > 0 && workspace && workspace_floats >= (int64_t)n * slices * c * TMAX && ((uintptr_t)workspace % 16) == 0;
    static const char* mf_env = mgf_knob("MGF_ATTN_GRAD_MFMA");     // tuning hook (experiments, tests): 0 = the VALU kernel
    if (!ok || (mf_env && mf_env[0] == '0')) return mgf_attn_values_grad(dvwb, dg, probs, n, c, f, t, stream);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(attn_values_grad_mfma_kernel, dim3(slices, (unsigned)mgf_cdiv(cblocks, 4), n), dim3(256), 0, st, workspace, dg, probs, c, f, slices);
    const int64_t per_sample = (int64_t)c * TMAX;
    hipLaunchKernelGGL(attn_values_reduce_kernel, dim3((unsigned)mgf_cdiv(per_sample, 256), n), dim3(256), 0, st, dvwb, workspace, per_sample, slices);
    MGF_CHECK_LAUNCH("attn_values_grad");
    return MGF_OK;
}

extern "C" int mgf_style_demod_bwd_multi(float* dwg, const mgf_style_bwd_job* jobs_dev, int32_t njobs, int32_t n, int32_t wdim,
                                         int32_t max_channels, mgf_stream_t stream) {
    MGF_REQUIRE(dwg && jobs_dev && njobs >= 1 && n >= 1, MGF_EINVAL, "style_demod_bwd_multi: bad arguments");
    MGF_REQUIRE(wdim >= 1 && wdim <= 256 && 256 % wdim == 0, MGF_EUNSUPPORTED, "style_demod_bwd_multi: wdim must divide 256 (got %d)", wdim);
    MGF_REQUIRE(max_channels >= 1 && max_channels <= 2048, MGF_EUNSUPPORTED, "style_demod_bwd_multi: at most 2048 channels per layer (got %d)", max_channels);
    MGF_REQUIRE(njobs <= 65535 && n <= 65535, MGF_ETOOBIG, "style_demod_bwd_multi: too many jobs/samples");
    hipLaunchKernelGGL(style_demod_bwd_kernel, dim3(njobs, n), dim3(256), 0, (hipStream_t)stream, dwg, jobs_dev, njobs, wdim);
    MGF_CHECK_LAUNCH("style_demod_bwd_multi");
    return MGF_OK;
}

extern "C" int mgf_attn_values_bwd_multi(float* dyc, const mgf_attn_bwd_job* jobs_dev, int32_t njobs, int32_t n, int32_t t, int32_t wdim,
                                         mgf_stream_t stream) {
    MGF_REQUIRE(dyc && jobs_dev && njobs >= 1 && n >= 1 && t >= 1 && wdim >= 1, MGF_EINVAL, "attn_values_bwd_multi: bad arguments");
    MGF_REQUIRE(njobs <= 65535 && n <= 65535, MGF_ETOOBIG, "attn_values_bwd_multi: too many jobs/samples");
    hipLaunchKernelGGL(attn_values_bwd_kernel, dim3(njobs, n), dim3(256), 0, (hipStream_t)stream, dyc, jobs_dev, njobs, t, wdim);
    MGF_CHECK_LAUNCH("attn_values_bwd_multi");
    return MGF_OK;
}

extern "C" int mgf_latent_bwd_multi(float* dwg, const mgf_style_bwd_job* style_jobs_dev, int32_t n_style_jobs, float* dyc,
                                    const mgf_attn_bwd_job* attn_jobs_dev, int32_t n_attn_jobs, int32_t n, int32_t t, int32_t wdim,
                                    int32_t max_channels, mgf_stream_t stream) {
    MGF_REQUIRE(dwg && style_jobs_dev && n_style_jobs >= 1 && n >= 1, MGF_EINVAL, "latent_bwd_multi: bad arguments");
    MGF_REQUIRE(n_attn_jobs == 0 || (dyc && attn_jobs_dev && t >= 1), MGF_EINVAL, "latent_bwd_multi: attention jobs need dyc, the job table and t");
    MGF_REQUIRE(wdim >= 1 && wdim <= 256 && 256 % wdim == 0, MGF_EUNSUPPORTED, "latent_bwd_multi: wdim must divide 256 (got %d)", wdim);
    MGF_REQUIRE(max_channels >= 1 && max_channels <= 2048, MGF_EUNSUPPORTED, "latent_bwd_multi: at most 2048 channels per layer (got %d)", max_channels);
    MGF_REQUIRE(n_style_jobs + n_attn_jobs <= 65535 && n <= 65535, MGF_ETOOBIG, "latent_bwd_multi: too many jobs/samples");
    hipLaunchKernelGGL(latent_bwd_multi_kernel, dim3(n_style_jobs + n_attn_jobs, n), dim3(256), 0, (hipStream_t)stream, dwg, style_jobs_dev,
                       n_style_jobs, dyc, attn_jobs_dev, n_attn_jobs, t, wdim);
    MGF_CHECK_LAUNCH("latent_bwd_multi");
    return MGF_OK;
}

extern "C" int mgf_latent_grad_gather(float* dw, const float* dwg, int32_t n_style_jobs, const float* dyc, int32_t n_attn_jobs, int32_t n,
                                      int32_t k, int32_t wdim, float scale, mgf_stream_t stream) {
    MGF_REQUIRE(dw && n >= 1 && k >= 2 && wdim >= 1, MGF_EINVAL, "latent_grad_gather: bad arguments");
    MGF_REQUIRE((dwg || n_style_jobs == 0) && (dyc || n_attn_jobs == 0), MGF_EINVAL, "latent_grad_gather: null partials");
    hipLaunchKernelGGL(latent_grad_gather_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, dw, dwg, n_style_jobs, dyc, n_attn_jobs, k, wdim,
                       scale);
    MGF_CHECK_LAUNCH("latent_grad_gather");
    return MGF_OK;
}

// Pixels per workgroup: 64 (256-byte row segments per channel) while that leaves >= 512 workgroups, else 32 while >= 256, else 16 -- the
// 16-pixel form moves 64-byte segments and streams at half the rate (1.7 vs 3.4 TB/s at 8 x 256 x 127^2).  MGF_LPIPS_BWD_PXB pins one,
// MGF_LPIPS_BWD_CACHE=0 turns the register-resident second sweep off (tuning).
template <bool RELU>
static void lpips_bwd_launch(float* oa, float* ob, const float* din, const float* f0, const float* f1u, const float* lin, int n, int c,
                             int c_split, int64_t hw, int64_t f1_bs, float k, int accumulate, hipStream_t st, const float* stats = nullptr) {
    static const int env_pxb = [] { const char* e = mgf_knob("MGF_LPIPS_BWD_PXB"); return e ? atoi(e) : 0; }();
    int pxb = mgf_cdiv(hw, 64) * n >= 512 ? 64 : mgf_cdiv(hw, 32) * n >= 256 ? 32 : 16;
    if (env_pxb == 16 || env_pxb == 32 || env_pxb == 64) pxb = env_pxb;
    static const bool no_cache = [] { const char* e = mgf_knob("MGF_LPIPS_BWD_CACHE"); return e && e[0] == '0'; }();
    const dim3 grid((unsigned)mgf_cdiv(hw, pxb), n);
#define MGF_LPB_LAUNCH(PX, CP) hipLaunchKernelGGL((lpips_layer_bwd_kernel<PX, RELU, CP>), grid, dim3(256), 0, st, oa, ob, din, f0, f1u, lin, c, c_split, hw, f1_bs, k, accumulate, stats)
    if (pxb == 64) {
        if (c <= 64 && !no_cache && !stats) MGF_LPB_LAUNCH(64, 16);
        else MGF_LPB_LAUNCH(64, 0);
    } else if (pxb == 32) {
        if (c <= 128 && !no_cache && !stats) MGF_LPB_LAUNCH(32, 16);
        else MGF_LPB_LAUNCH(32, 0);
    } else MGF_LPB_LAUNCH(16, 0);
#undef MGF_LPB_LAUNCH
}

extern "C" int mgf_lpips_layer_bwd_f32(float* df0, const float* f0, const float* f1_unit, const float* lin, int32_t n, int32_t c, int64_t hw,
                                       int64_t f1_batch_stride, float scale, int32_t accumulate, mgf_stream_t stream) {
    MGF_REQUIRE(df0 && f0 && f1_unit && lin && n >= 1 && c >= 1 && hw >= 1, MGF_EINVAL, "lpips_layer_bwd: bad arguments");
    MGF_REQUIRE((int64_t)c * hw < (1LL << 30), MGF_ETOOBIG, "lpips_layer_bwd: one sample's tap must stay below 4 GiB (32-bit buffer offsets)");
    MGF_REQUIRE(n <= 65535, MGF_ETOOBIG, "lpips_layer_bwd: n must be <= 65535");
    const float kk = 2.f * scale / (float)hw;
    lpips_bwd_launch<false>(df0, nullptr, nullptr, f0, f1_unit, lin, n, c, c, hw, f1_batch_stride, kk, accumulate, (hipStream_t)stream);
    MGF_CHECK_LAUNCH("lpips_layer_bwd");
    return MGF_OK;
}

extern "C" int mgf_lpips_layer_bwd_relu_f32(float* dz_a, float* dz_b, const float* dy, const float* f0, const float* f1_unit,
                                            const float* lin, int32_t n, int32_t c, int32_t c_split, int64_t hw, int64_t f1_batch_stride,
                                            float scale, mgf_stream_t stream) {
    return mgf_lpips_layer_bwd_relu_stats_f32(dz_a, dz_b, dy, f0, f1_unit, lin, nullptr, n, c, c_split, hw, f1_batch_stride, scale, stream);
}

extern "C" int mgf_lpips_layer_bwd_relu_stats_f32(float* dz_a, float* dz_b, const float* dy, const float* f0, const float* f1_unit,
                                                  const float* lin, const float* stats, int32_t n, int32_t c, int32_t c_split, int64_t hw,
                                                  int64_t f1_batch_stride, float scale, mgf_stream_t stream) {
    MGF_REQUIRE(dz_a && f0 && f1_unit && lin && n >= 1 && c >= 1 && hw >= 1, MGF_EINVAL, "lpips_layer_bwd_relu: bad arguments");
    MGF_REQUIRE(c_split >= 1 && c_split <= c && (dz_b || c_split == c), MGF_EINVAL, "lpips_layer_bwd_relu: bad split %d of %d channels", c_split, c);
    MGF_REQUIRE(n <= 65535, MGF_ETOOBIG, "lpips_layer_bwd_relu: n must be <= 65535");
    MGF_REQUIRE((int64_t)c * hw < (1LL << 30), MGF_ETOOBIG, "lpips_layer_bwd_relu: one sample's tap must stay below 4 GiB (32-bit buffer offsets)");
    const float kk = 2.f * scale / (float)hw;
    lpips_bwd_launch<true>(dz_a, dz_b, dy, f0, f1_unit, lin, n, c, c_split, hw, f1_batch_stride, kk, 0, (hipStream_t)stream, stats);
    MGF_CHECK_LAUNCH("lpips_layer_bwd_relu");
    return MGF_OK;
}

extern "C" int mgf_relu_bwd_split_f32(float* dz_a, float* dz_b, const float* dy, const float* y, int32_t n, int32_t c, int32_t c_split,
                                      int64_t hw, mgf_stream_t stream) {
    MGF_REQUIRE(dz_a && dy && y && n >= 1 && c >= 1 && hw >= 1, MGF_EINVAL, "relu_bwd_split: bad arguments");
    MGF_REQUIRE(c_split >= 1 && c_split <= c && (dz_b || c_split == c), MGF_EINVAL, "relu_bwd_split: bad split %d of %d channels", c_split, c);
    const int64_t total = (int64_t)n * c * hw;
    hipLaunchKernelGGL(relu_bwd_split_kernel, dim3(mgf_stream_grid(total, 256, 4)), dim3(256), 0, (hipStream_t)stream, dz_a, dz_b, dy, y, c,
                       c_split, hw, total);
    MGF_CHECK_LAUNCH("relu_bwd_split");
    return MGF_OK;
}

namespace {
// dx of the 3x3 / stride-2 ceil-mode pool from the forward's stored tap indices: one lane per 2 x 2 input block (2 bi, 2 bj) + {0,1}^2,
// whose elements sit at fixed taps of the <= 4 windows around it -- (0,0): tap 8 of window (bi-1, bj-1), 6 of (bi-1, bj), 2 of
// (bi, bj-1), 0 of (bi, bj); (0,1): 7 of (bi-1, bj), 1 of (bi, bj); (1,0): 5 of (bi, bj-1), 3 of (bi, bj); (1,1): 4 of (bi, bj) --
// summed in ascending window order like torch.  No LDS, no input map: a stream of stores.
__global__ __launch_bounds__(256) void maxpool3x3s2_bwd_idx_kernel(float* __restrict__ dx, const float* __restrict__ dy,
                                                                   const uint8_t* __restrict__ idx, int ih, int iw, int oh, int ow) {
    const int bj = blockIdx.x * 256 + threadIdx.x, bi = blockIdx.y;
    const int64_t pl = blockIdx.z;
    const int yy = 2 * bi, xx = 2 * bj;
    if (xx >= iw) return;
    const float* dp = dy + pl * oh * ow;
    const uint8_t* ip = idx + pl * oh * ow;
    float g[2][2];
    int t[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int oy = bi - 1 + a, ox = bj - 1 + b;
            const bool in = oy >= 0 && ox >= 0 && oy < oh && ox < ow;
            g[a][b] = in ? dp[(int64_t)oy * ow + ox] : 0.f;
            t[a][b] = in ? (int)ip[(int64_t)oy * ow + ox] : -1;
        }
    float e00 = 0.f, e01 = 0.f, e10 = 0.f, e11 = 0.f;
    if (t[0][0] == 8) e00 += g[0][0];
    if (t[0][1] == 6) e00 += g[0][1];
    if (t[1][0] == 2) e00 += g[1][0];
    if (t[1][1] == 0) e00 += g[1][1];
    if (t[0][1] == 7) e01 += g[0][1];
    if (t[1][1] == 1) e01 += g[1][1];
    if (t[1][0] == 5) e10 += g[1][0];
    if (t[1][1] == 3) e10 += g[1][1];
    if (t[1][1] == 4) e11 += g[1][1];
    float* o = dx + pl * ih * iw + (int64_t)yy * iw + xx;
    const bool x1 = xx + 1 < iw;
    o[0] = e00;
    if (x1) o[1] = e01;
    if (yy + 1 < ih) {
        o[iw] = e10;
        if (x1) o[iw + 1] = e11;
    }
}
}  // namespace

extern "C" int mgf_maxpool3x3s2_ceil_bwd_idx_f32(float* dx, const float* dy, const uint8_t* idx, int32_t nc, int32_t in_h, int32_t in_w,
                                                 int32_t out_h, int32_t out_w, mgf_stream_t stream) {
    MGF_REQUIRE(dx && dy && idx && nc >= 1 && in_h >= 1 && in_w >= 1 && out_h >= 1 && out_w >= 1, MGF_EINVAL, "maxpool3x3s2_ceil_bwd_idx: bad arguments");
    MGF_REQUIRE(2 * (out_h - 1) < in_h && 2 * (out_w - 1) < in_w, MGF_EINVAL, "maxpool3x3s2_ceil_bwd_idx: output extent does not match the input");
    MGF_REQUIRE(nc <= 65535 && (in_h + 1) / 2 <= 65535, MGF_ETOOBIG, "maxpool3x3s2_ceil_bwd_idx: at most 65535 planes and 131070 rows");
    const dim3 grid((unsigned)mgf_cdiv((in_w + 1) / 2, 256), (unsigned)((in_h + 1) / 2), nc);
    hipLaunchKernelGGL(maxpool3x3s2_bwd_idx_kernel, grid, dim3(256), 0, (hipStream_t)stream, dx, dy, idx, in_h, in_w, out_h, out_w);
    MGF_CHECK_LAUNCH("maxpool3x3s2_ceil_bwd_idx");
    return MGF_OK;
}

extern "C" int mgf_maxpool3x3s2_ceil_bwd_f32(float* dx, const float* dy, const float* x, int32_t nc, int32_t in_h, int32_t in_w, int32_t out_h,
                                             int32_t out_w, mgf_stream_t stream) {
    MGF_REQUIRE(dx && dy && x && nc >= 1 && in_h >= 1 && in_w >= 1 && out_h >= 1 && out_w >= 1, MGF_EINVAL, "maxpool3x3s2_ceil_bwd: bad arguments");
    MGF_REQUIRE(2 * (out_h - 1) < in_h && 2 * (out_w - 1) < in_w, MGF_EINVAL, "maxpool3x3s2_ceil_bwd: output extent does not match the input");
    const int tiles_x = (int)mgf_cdiv(in_w, MPTX), tiles_y = (int)mgf_cdiv(in_h, MPTY);
    MGF_REQUIRE((int64_t)nc * tiles_x * tiles_y <= INT32_MAX, MGF_ETOOBIG, "maxpool3x3s2_ceil_bwd: too many tiles");
    hipLaunchKernelGGL(maxpool3x3s2_bwd_kernel, dim3((unsigned)((int64_t)nc * tiles_x * tiles_y)), dim3(256), 0, (hipStream_t)stream, dx, dy, x,
                       in_h, in_w, out_h, out_w, tiles_x, tiles_y);
    MGF_CHECK_LAUNCH("maxpool3x3s2_ceil_bwd");
    return MGF_OK;
}

extern "C" int mgf_mse_grad_f32(float* d, const float* a, const float* b, int32_t n, int64_t numel, int64_t b_batch_stride, float scale,
                                int32_t accumulate, mgf_stream_t stream) {
    MGF_REQUIRE(d && a && b && n >= 1 && numel >= 1, MGF_EINVAL, "mse_grad: bad arguments");
    const int64_t total = (int64_t)n * numel;
    hipLaunchKernelGGL(mse_grad_kernel, dim3(mgf_stream_grid(total, 256, 4)), dim3(256), 0, (hipStream_t)stream, d, a, b, numel, b_batch_stride,
                       2.f * scale / (float)numel, accumulate, total);
    MGF_CHECK_LAUNCH("mse_grad");
    return MGF_OK;
}

extern "C" int mgf_adam_step_f32(float* param, float* exp_avg, float* exp_avg_sq, int32_t* adam_t, const float* grad, const float* lr_table,
                                 const int32_t* step, const int32_t* valid, int64_t numel, int32_t steps_total, float beta1, float beta2,
                                 float eps, float weight_decay, mgf_stream_t stream) {
    MGF_REQUIRE(param && exp_avg && exp_avg_sq && adam_t && grad && lr_table && step, MGF_EINVAL, "adam_step: null pointer");
    MGF_REQUIRE(numel >= 1 && steps_total >= 1, MGF_EINVAL, "adam_step: bad sizes");
    MGF_REQUIRE(numel <= (1 << 20), MGF_EUNSUPPORTED, "adam_step: single-workgroup kernel for latent-sized parameters (numel <= 2^20)");
    hipLaunchKernelGGL(adam_step_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, param, exp_avg, exp_avg_sq, adam_t, grad, lr_table, step,
                       valid, numel, steps_total, beta1, beta2, eps, weight_decay);
    MGF_CHECK_LAUNCH("adam_step");
    return MGF_OK;
}

extern "C" int mgf_maxpool_s2_floor_bwd_f32(float* dx, const float* dy, const float* x, int32_t nc, int32_t in_h, int32_t in_w, int32_t ksize,
                                            mgf_stream_t stream) {
    MGF_REQUIRE(dx && dy && x && nc >= 1 && (ksize == 2 || ksize == 3) && in_h >= ksize && in_w >= ksize, MGF_EINVAL, "maxpool_s2_floor_bwd: bad arguments");
    const int out_h = (in_h - ksize) / 2 + 1, out_w = (in_w - ksize) / 2 + 1;
    if (ksize == 3) return mgf_maxpool3x3s2_ceil_bwd_f32(dx, dy, x, nc, in_h, in_w, out_h, out_w, stream);      // same rule, fewer windows
    const int64_t total = (int64_t)nc * in_h * in_w;
    hipLaunchKernelGGL(maxpool2x2s2_bwd_kernel, dim3(mgf_stream_grid(total, 256, 2)), dim3(256), 0, (hipStream_t)stream, dx, dy, x, in_h, in_w,
                       out_h, out_w, total);
    MGF_CHECK_LAUNCH("maxpool_s2_floor_bwd");
    return MGF_OK;
}
