// Duplex (image <- latents) attention of a SynthesisLayer for gfx950, re-associated so that no F x C x C GEMM remains.
// Contract: include/mgf.h (mgf_duplex_attention).  Reference: TransformerLayer.forward / integrate / att_norm
// (training/networks.py:748-822, 657-672, 341-358) + the noise / bias_act tail of SynthesisLayer.forward (:1036-1040).
//
//   S[f,t] = sum_c x[c,f] * wqc[c,t] + spos[f,t]       (query projection, positional term, att_weight, centroids and 1/sqrt(C)
//                                                        are folded into wqc [C,T] and spos [F,T] once per checkpoint)
//   P      = softmax_t(S)
//   y[c,f] = epilogue( x[c,f] * rsqrt(mean_c x^2 + 1e-8) * sum_t P[f,t] * vwb[c,t] )   (vwb = V Wm^T + bm + 1 per sample)
//
// HBM/L2-bound streaming kernel.  A workgroup (4 waves) owns PXB consecutive pixels and splits the channels over
// G = 256 / PXB lane groups: PXB = 64 for feature maps >= 32x32 (one 256-byte row segment per channel and wave), PXB = 16
// below that so that the 4x4 .. 16x16 layers still spread over several workgroups.  The [C,16] tables (wqc, then vwb) are
// staged in LDS once per workgroup and read back as wave-uniform 16-byte broadcasts; the channel loop is unrolled 4-deep so
// four independent row loads are in flight per lane.  Score partials meet in LDS, every lane finishes the softmax of its
// own pixel in registers, then the groups sweep their channels again (L2-resident re-read) and apply the epilogue.
#include "mgf_common.h"

namespace {

typedef float v2f __attribute__((ext_vector_type(2)));
// The two 16-wide inner products of the register kernels in PACKED fp32 (v_pk_fma_f32: two lanes of math per issue slot; the scalar form
// issued 16 v_fmac per channel and phase): scores  s[0..15] += x * row,  gain  g = <s, row>.  `row` = 16 floats of an LDS table.
__device__ __forceinline__ void att_fma_row(v2f (&s2)[8], float xv, const float4* w4) {
    const v2f x2 = {xv, xv};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4 w = w4[q];
        const v2f lo = {w.x, w.y}, hi = {w.z, w.w};
        s2[2 * q] += x2 * lo;
        s2[2 * q + 1] += x2 * hi;
    }
}
__device__ __forceinline__ float att_dot_row(const v2f (&s2)[8], const float4* w4) {
    v2f g2 = {0.f, 0.f};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4 w = w4[q];
        const v2f lo = {w.x, w.y}, hi = {w.z, w.w};
        g2 += s2[2 * q] * lo;
        g2 += s2[2 * q + 1] * hi;
    }
    return g2.x + g2.y;
}

constexpr int TMAX = 16;
constexpr int UNR = 4;

struct AttnParams {
    float* y;
    const float* x;
    const float* wqc;     // [c, t]
    const float* spos;    // [f, t]
    const float* vwb;     // [n, c, t]
    int n, c, f, t;
    int c_pad;            // c rounded up to a multiple of UNR * G (table rows beyond c are zero)
    mgf_epilogue ep;
    int has_ep;
    float* probs;         // [n, f, t] or null
    int32_t* argmax;      // [n, f] or null
};

// NCH = channels per lane group (C / G) when it is one of 8/16/32/64: the lane then keeps its x values in registers for both
// sweeps (one batch of independent loads, no re-read); NCH = 0 is the generic two-sweep form for any C.
template <int PXB, int NCH>
__global__ __launch_bounds__(256) void duplex_attention_kernel(AttnParams p) {
    constexpr int G = 256 / PXB;                 // channel groups per workgroup
    extern __shared__ float lds[];
    float* tab = lds;                             // [c_pad][TMAX]
    float* part = lds + (size_t)p.c_pad * TMAX;   // [G][TMAX + 1][PXB]
    const int tid = threadIdx.x;
    const int px = tid % PXB, grp = tid / PXB;
    const int n = blockIdx.y;
    const int f0 = blockIdx.x * PXB;
    const int f = f0 + px;
    const bool valid = f < p.f;
    const int fc = valid ? f : p.f - 1;
    const float* xn = p.x + (int64_t)n * p.c * p.f;
    const int T = p.t;

    // channels of this lane: c = (k / UNR) * G * UNR + grp * UNR + (k % UNR), k = 0 .. NCH-1 (same walk as the generic loop)
    float xreg[NCH > 0 ? NCH : 1];
    if (NCH > 0) {
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const int c = (k / UNR) * G * UNR + grp * UNR + (k % UNR);
            xreg[k] = xn[(int64_t)c * p.f + fc];
        }
    }

    // ---- stage a [c][T] table zero-padded to [c_pad][16]; T == 16 is a straight 16-byte copy, four loads in flight ----
    auto stage_table = [&](const float* src) {
        if (T == TMAX) {
            const int n4 = p.c * (TMAX / 4), n4_pad = p.c_pad * (TMAX / 4);
            const float4* s4 = reinterpret_cast<const float4*>(src);
            float4* d4 = reinterpret_cast<float4*>(tab);
            for (int i0 = tid; i0 < n4_pad; i0 += 256 * 4) {
                float4 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int i = i0 + 256 * u;
                    v[u] = i < n4 ? s4[i] : make_float4(0.f, 0.f, 0.f, 0.f);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int i = i0 + 256 * u;
                    if (i < n4_pad) d4[i] = v[u];
                }
            }
        } else {
            for (int i = tid; i < p.c_pad * TMAX; i += 256) {
                const int c = i / TMAX, t = i % TMAX;
                tab[i] = (c < p.c && t < T) ? src[(int64_t)c * T + t] : 0.f;
            }
        }
    };
    stage_table(p.wqc);
    __syncthreads();

    float s[TMAX];
#pragma unroll
    for (int t = 0; t < TMAX; ++t) s[t] = 0.f;
    float sq = 0.f;
    auto accumulate = [&](float xv, int c) {
        const float4* w4 = reinterpret_cast<const float4*>(tab + c * TMAX);
        sq += xv * xv;
#pragma unroll
        for (int q = 0; q < TMAX / 4; ++q) {
            const float4 w = w4[q];
            s[4 * q + 0] += xv * w.x; s[4 * q + 1] += xv * w.y;
            s[4 * q + 2] += xv * w.z; s[4 * q + 3] += xv * w.w;
        }
    };
    if (NCH > 0) {
#pragma unroll
        for (int k = 0; k < NCH; ++k) accumulate(xreg[k], (k / UNR) * G * UNR + grp * UNR + (k % UNR));
    } else {
        for (int c0 = grp * UNR; c0 < p.c; c0 += G * UNR) {
            float xv[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) xv[u] = (c0 + u < p.c) ? xn[(int64_t)(c0 + u) * p.f + fc] : 0.f;
#pragma unroll
            for (int u = 0; u < UNR; ++u) accumulate(xv[u], c0 + u);
        }
    }
#pragma unroll
    for (int t = 0; t < TMAX; ++t) part[(grp * (TMAX + 1) + t) * PXB + px] = s[t];
    part[(grp * (TMAX + 1) + TMAX) * PXB + px] = sq;
    __syncthreads();

    // ---- every lane finishes the softmax of its pixel; meanwhile the table is re-staged with this sample's vwb ----
    float m = -3.0e38f;
#pragma unroll
    for (int t = 0; t < TMAX; ++t) {
        float v = 0.f;
        for (int g = 0; g < G; ++g) v += part[(g * (TMAX + 1) + t) * PXB + px];
        if (t < T) {
            v += p.spos[(int64_t)fc * T + t];
            m = fmaxf(m, v);
        }
        s[t] = v;
    }
    sq = 0.f;
    for (int g = 0; g < G; ++g) sq += part[(g * (TMAX + 1) + TMAX) * PXB + px];
    stage_table(p.vwb + (int64_t)n * p.c * T);
    float den = 0.f;
    int best = 0;
    float bestv = -3.0e38f;
#pragma unroll
    for (int t = 0; t < TMAX; ++t) {
        if (t < T) {
            if (s[t] > bestv) { bestv = s[t]; best = t; }
            s[t] = __expf(s[t] - m);
            den += s[t];
        } else {
            s[t] = 0.f;
        }
    }
    const float inv = 1.f / den;
    const float rs = rsqrtf(sq / (float)p.c + 1e-8f);
#pragma unroll
    for (int t = 0; t < TMAX; ++t) s[t] *= inv;
    if (grp == 0 && valid) {
        if (p.probs)
            for (int t = 0; t < T; ++t) p.probs[((int64_t)n * p.f + f) * T + t] = s[t];
        if (p.argmax) p.argmax[(int64_t)n * p.f + f] = best;
    }
#pragma unroll
    for (int t = 0; t < TMAX; ++t) s[t] *= rs;            // fold the layer norm into the probabilities once
    __syncthreads();

    float nz = 0.f;
    if (p.has_ep && p.ep.noise) {
        const float ns = p.ep.noise_strength ? *p.ep.noise_strength : 1.f;
        nz = p.ep.noise[(int64_t)(p.ep.noise_n > 1 ? n : 0) * p.f + fc] * ns;
    }
    float* yn = p.y + (int64_t)n * p.c * p.f;
    const float* rn = (p.has_ep && p.ep.residual) ? p.ep.residual + (int64_t)n * p.c * p.f : nullptr;
    auto finish = [&](float xv, float rv, int c) {
        const float4* w4 = reinterpret_cast<const float4*>(tab + c * TMAX);
        float g = 0.f;
#pragma unroll
        for (int q = 0; q < TMAX / 4; ++q) {
            const float4 w = w4[q];
            g += s[4 * q + 0] * w.x + s[4 * q + 1] * w.y + s[4 * q + 2] * w.z + s[4 * q + 3] * w.w;
        }
        float v = xv * g;
        if (p.has_ep) {
            v += nz;
            if (p.ep.bias && c < p.c) v += p.ep.bias[c];
            if (p.ep.act == MGF_ACT_LRELU) v = v > 0.f ? v : v * p.ep.alpha;
            else if (p.ep.act == MGF_ACT_RELU) v = v > 0.f ? v : 0.f;
            v *= p.ep.gain;
            v += rv;
        }
        if (valid && c < p.c) yn[(int64_t)c * p.f + f] = v;
    };
    if (NCH > 0) {
#pragma unroll
        for (int k0 = 0; k0 < NCH; k0 += 8) {
            float rv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int k = k0 + u;
                const int c = (k / UNR) * G * UNR + grp * UNR + (k % UNR);
                rv[u] = rn ? rn[(int64_t)c * p.f + fc] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int k = k0 + u;
                finish(xreg[k], rv[u], (k / UNR) * G * UNR + grp * UNR + (k % UNR));
            }
        }
    } else {
        for (int c0 = grp * UNR; c0 < p.c; c0 += G * UNR) {
            float xv[UNR], rv[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const bool ok = c0 + u < p.c;
                xv[u] = ok ? xn[(int64_t)(c0 + u) * p.f + fc] : 0.f;
                rv[u] = (ok && rn) ? rn[(int64_t)(c0 + u) * p.f + fc] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u) finish(xv[u], rv[u], c0 + u);
        }
    }
}

// Register-resident form for C = NCH * G, T == 16: every global operand of the workgroup (x, residual, both tables, spos, noise)
// is requested up front in one burst of independent loads, so the kernel pays ONE memory round trip instead of ~10 dependent
// ones -- these layers are 4x4 .. 128x128 maps, i.e. pure latency.
template <int PXB, int NCH>
__global__ __launch_bounds__(256) void duplex_attention_reg_kernel(AttnParams p) {
    constexpr int G = 256 / PXB;
    constexpr int TV = NCH * G / 64;              // float4 per lane that cover one [C][16] table (C*4 float4 / 256 lanes)
    extern __shared__ float lds[];
    float* tab = lds;                             // [C][16]
    float* part = lds + (size_t)p.c * TMAX;       // [G][17][PXB]
    const int tid = threadIdx.x;
    const int px = tid % PXB, grp = tid / PXB;
    const int n = blockIdx.y;
    const int f = blockIdx.x * PXB + px;
    const bool valid = f < p.f;
    const int fc = valid ? f : p.f - 1;
    const float* xn = p.x + (int64_t)n * p.c * p.f;
    const float* rn = (p.has_ep && p.ep.residual) ? p.ep.residual + (int64_t)n * p.c * p.f : nullptr;
    const float4* wq4 = reinterpret_cast<const float4*>(p.wqc);
    const float4* vw4 = reinterpret_cast<const float4*>(p.vwb + (int64_t)n * p.c * TMAX);

    // ---- one burst of loads ----
    float xreg[NCH], rreg[NCH];
    float4 tq[TV], tv[TV];
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        const int c = (k / UNR) * G * UNR + grp * UNR + (k % UNR);
        xreg[k] = xn[(int64_t)c * p.f + fc];
        rreg[k] = rn ? rn[(int64_t)c * p.f + fc] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < TV; ++u) { tq[u] = wq4[tid + 256 * u]; tv[u] = vw4[tid + 256 * u]; }
    float sp[TMAX];
#pragma unroll
    for (int t = 0; t < TMAX; ++t) sp[t] = p.spos[(int64_t)fc * TMAX + t];
    float nz = 0.f;
    if (p.has_ep && p.ep.noise) {
        const float ns = p.ep.noise_strength ? *p.ep.noise_strength : 1.f;
        nz = p.ep.noise[(int64_t)(p.ep.noise_n > 1 ? n : 0) * p.f + fc] * ns;
    }

    float4* t4 = reinterpret_cast<float4*>(tab);
#pragma unroll
    for (int u = 0; u < TV; ++u) t4[tid + 256 * u] = tq[u];
    __syncthreads();

    float s[TMAX];
#pragma unroll
    for (int t = 0; t < TMAX; ++t) s[t] = 0.f;
    float sq = 0.f;
    v2f s2[TMAX / 2];
#pragma unroll
    for (int q = 0; q < TMAX / 2; ++q) s2[q] = v2f{0.f, 0.f};
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        const int c = (k / UNR) * G * UNR + grp * UNR + (k % UNR);
        const float xv = xreg[k];
        sq += xv * xv;
        att_fma_row(s2, xv, reinterpret_cast<const float4*>(tab + c * TMAX));
    }
#pragma unroll
    for (int t = 0; t < TMAX; ++t) part[(grp * (TMAX + 1) + t) * PXB + px] = (t & 1) ? s2[t >> 1].y : s2[t >> 1].x;
    part[(grp * (TMAX + 1) + TMAX) * PXB + px] = sq;
    __syncthreads();                               // all reads of the wqc table are done: overwrite it with vwb
#pragma unroll
    for (int u = 0; u < TV; ++u) t4[tid + 256 * u] = tv[u];

    // the G partial sums of the 17 values of a pixel: group g totals value t = g (+ G, ...) and publishes it -- every group summing
    // all 17 itself was 272 LDS reads + adds per thread where 33 + 16 do
    float* fin = part + G * (TMAX + 1) * PXB;      // [17][PXB]
    for (int t = grp; t < TMAX + 1; t += G) {
        float v = 0.f;
#pragma unroll
        for (int g = 0; g < G; ++g) v += part[(g * (TMAX + 1) + t) * PXB + px];
        fin[t * PXB + px] = v;
    }
    __syncthreads();
    float m = -3.0e38f;
#pragma unroll
    for (int t = 0; t < TMAX; ++t) {
        const float v = fin[t * PXB + px] + sp[t];
        m = fmaxf(m, v);
        s[t] = v;
    }
    sq = fin[TMAX * PXB + px];
    float den = 0.f;
    int best = 0;
    float bestv = -3.0e38f;
#pragma unroll
    for (int t = 0; t < TMAX; ++t) {
        if (s[t] > bestv) { bestv = s[t]; best = t; }
        s[t] = __expf(s[t] - m);
        den += s[t];
    }
    const float inv = 1.f / den;
    const float rs = rsqrtf(sq / (float)p.c + 1e-8f);
#pragma unroll
    for (int t = 0; t < TMAX; ++t) s[t] *= inv;
    if (grp == 0 && valid) {
        if (p.probs)
            for (int t = 0; t < TMAX; ++t) p.probs[((int64_t)n * p.f + f) * TMAX + t] = s[t];
        if (p.argmax) p.argmax[(int64_t)n * p.f + f] = best;
    }
#pragma unroll
    for (int q = 0; q < TMAX / 2; ++q) s2[q] = v2f{s[2 * q] * rs, s[2 * q + 1] * rs};
    // (the barrier behind the totals also ordered the vwb table: no further barrier)

    float* yn = p.y + (int64_t)n * p.c * p.f;
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        const int c = (k / UNR) * G * UNR + grp * UNR + (k % UNR);
        const float g = att_dot_row(s2, reinterpret_cast<const float4*>(tab + c * TMAX));
        float v = xreg[k] * g;
        if (p.has_ep) {
            v += nz;
            if (p.ep.bias) v += p.ep.bias[c];
            if (p.ep.act == MGF_ACT_LRELU) v = v > 0.f ? v : v * p.ep.alpha;
            else if (p.ep.act == MGF_ACT_RELU) v = v > 0.f ? v : 0.f;
            v *= p.ep.gain;
            v += rreg[k];
        }
        if (valid) yn[(int64_t)c * p.f + f] = v;
    }
}

// Multi-block variant of the kernel above (the 128x128 layers: 16384 pixels x 256 channels per sample): a workgroup walks NBLK consecutive pixel blocks with BOTH
// [C][16] tables resident in LDS.  With one block per workgroup the two tables (32 KB from L2, 8 LDS stores of 16 bytes per lane)
// were twice the bytes of the 16 KB of activations they served; the loads of block b+1 are issued before block b is finished.
template <int PXB, int NCH, int NBLK>
__global__ __launch_bounds__(256) void duplex_attention_blocks_kernel(AttnParams p) {
    constexpr int G = 256 / PXB;
    constexpr int TV = NCH * G / 64;              // float4 per lane that cover one [C][16] table (C*4 float4 / 256 lanes)
    constexpr bool TWO = NBLK > 1;                // both tables resident
    extern __shared__ float lds[];
    float* tab = lds;                             // [C][16] wqc (then vwb when !TWO)
    float* tabv = TWO ? lds + (size_t)p.c * TMAX : lds;                       // [C][16] vwb
    float* part = lds + (size_t)p.c * TMAX * (TWO ? 2 : 1);                   // [G][17][PXB]
    const int tid = threadIdx.x;
    const int px = tid % PXB, grp = tid / PXB;
    const int n = blockIdx.y;
    const float* xn = p.x + (int64_t)n * p.c * p.f;
    const float* rn = (p.has_ep && p.ep.residual) ? p.ep.residual + (int64_t)n * p.c * p.f : nullptr;
    const float4* wq4 = reinterpret_cast<const float4*>(p.wqc);
    const float4* vw4 = reinterpret_cast<const float4*>(p.vwb + (int64_t)n * p.c * TMAX);
    const float ns = (p.has_ep && p.ep.noise) ? (p.ep.noise_strength ? *p.ep.noise_strength : 1.f) : 0.f;

    float xreg[NCH], rreg[NCH], sp[TMAX];
    float nz = 0.f;
    auto load_block = [&](int blk) {               // one burst of independent loads
        const int f = (blockIdx.x * NBLK + blk) * PXB + px;
        const int fc = f < p.f ? f : p.f - 1;
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const int c = (k / UNR) * G * UNR + grp * UNR + (k % UNR);
            xreg[k] = xn[(int64_t)c * p.f + fc];
            rreg[k] = rn ? rn[(int64_t)c * p.f + fc] : 0.f;
        }
#pragma unroll
        for (int t = 0; t < TMAX; ++t) sp[t] = p.spos[(int64_t)fc * TMAX + t];
        if (p.has_ep && p.ep.noise) nz = p.ep.noise[(int64_t)(p.ep.noise_n > 1 ? n : 0) * p.f + fc] * ns;
    };
    float4 tq[TV], tv[TV];
    load_block(0);
#pragma unroll
    for (int u = 0; u < TV; ++u) { tq[u] = wq4[tid + 256 * u]; tv[u] = vw4[tid + 256 * u]; }
    float4* t4 = reinterpret_cast<float4*>(tab);
    float4* t4v = reinterpret_cast<float4*>(tabv);
#pragma unroll
    for (int u = 0; u < TV; ++u) t4[tid + 256 * u] = tq[u];
    if (TWO) {
#pragma unroll
        for (int u = 0; u < TV; ++u) t4v[tid + 256 * u] = tv[u];
    }
    __syncthreads();

#pragma unroll 1
    for (int blk = 0; blk < NBLK; ++blk) {
        const int f = (blockIdx.x * NBLK + blk) * PXB + px;
        const bool valid = f < p.f;
        float s[TMAX];
#pragma unroll
        for (int t = 0; t < TMAX; ++t) s[t] = 0.f;
        float sq = 0.f;
        float xk[NCH], rk[NCH];                    // this block's values; the registers of load_block go to the next block
#pragma unroll
        for (int k = 0; k < NCH; ++k) { xk[k] = xreg[k]; rk[k] = rreg[k]; }
        float spk[TMAX];
#pragma unroll
        for (int t = 0; t < TMAX; ++t) spk[t] = sp[t];
        const float nzk = nz;
        if (TWO && blk + 1 < NBLK) load_block(blk + 1);
        v2f s2[TMAX / 2];
#pragma unroll
        for (int q = 0; q < TMAX / 2; ++q) s2[q] = v2f{0.f, 0.f};
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const int c = (k / UNR) * G * UNR + grp * UNR + (k % UNR);
            const float xv = xk[k];
            sq += xv * xv;
            att_fma_row(s2, xv, reinterpret_cast<const float4*>(tab + c * TMAX));
        }
#pragma unroll
        for (int t = 0; t < TMAX; ++t) part[(grp * (TMAX + 1) + t) * PXB + px] = (t & 1) ? s2[t >> 1].y : s2[t >> 1].x;
        part[(grp * (TMAX + 1) + TMAX) * PXB + px] = sq;
        __syncthreads();                           // (one table: all reads of the wqc table are done, overwrite it with vwb)
        if (!TWO) {
#pragma unroll
            for (int u = 0; u < TV; ++u) t4[tid + 256 * u] = tv[u];
        }
        float* fin = part + G * (TMAX + 1) * PXB;  // [17][PXB]: group g totals value t = g (+ G, ...) of its pixel (see the kernel above)
        for (int t = grp; t < TMAX + 1; t += G) {
            float v = 0.f;
#pragma unroll
            for (int g = 0; g < G; ++g) v += part[(g * (TMAX + 1) + t) * PXB + px];
            fin[t * PXB + px] = v;
        }
        __syncthreads();
        float m = -3.0e38f;
#pragma unroll
        for (int t = 0; t < TMAX; ++t) {
            const float v = fin[t * PXB + px] + spk[t];
            m = fmaxf(m, v);
            s[t] = v;
        }
        sq = fin[TMAX * PXB + px];
        float den = 0.f;
        int best = 0;
        float bestv = -3.0e38f;
#pragma unroll
        for (int t = 0; t < TMAX; ++t) {
            if (s[t] > bestv) { bestv = s[t]; best = t; }
            s[t] = __expf(s[t] - m);
            den += s[t];
        }
        const float inv = 1.f / den;
        const float rs = rsqrtf(sq / (float)p.c + 1e-8f);
#pragma unroll
        for (int t = 0; t < TMAX; ++t) s[t] *= inv;
        if (grp == 0 && valid) {
            if (p.probs)
                for (int t = 0; t < TMAX; ++t) p.probs[((int64_t)n * p.f + f) * TMAX + t] = s[t];
            if (p.argmax) p.argmax[(int64_t)n * p.f + f] = best;
        }
#pragma unroll
        for (int q = 0; q < TMAX / 2; ++q) s2[q] = v2f{s[2 * q] * rs, s[2 * q + 1] * rs};
        // (no barrier here: `part` and `fin` of the next block are written behind barriers every wave passes after its reads)

        float* yn = p.y + (int64_t)n * p.c * p.f;
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const int c = (k / UNR) * G * UNR + grp * UNR + (k % UNR);
            const float g = att_dot_row(s2, reinterpret_cast<const float4*>(tabv + c * TMAX));
            float v = xk[k] * g;
            if (p.has_ep) {
                v += nzk;
                if (p.ep.bias) v += p.ep.bias[c];
                if (p.ep.act == MGF_ACT_LRELU) v = v > 0.f ? v : v * p.ep.alpha;
                else if (p.ep.act == MGF_ACT_RELU) v = v > 0.f ? v : 0.f;
                v *= p.ep.gain;
                v += rk[k];
            }
            if (valid) yn[(int64_t)c * p.f + f] = v;
        }
    }
}

// MFMA form of the register kernels for the generator's attention layers (C = 256 or 512, 16 latents; F a multiple of 32): both small
// products of the layer are 1x1 convolutions -- scores[t][px] = sum_c wqc[c][t] x[c][px] (C -> 16 channels) and gain[c][px] = sum_t
// vwb[c][t] p[t][px] (16 -> C) -- and run on v_mfma_f32_32x32x2_f32 with register operands like csrc/pointwise.hip.  A workgroup takes
// 32 pixels (every global access of a wave is two 128-byte row segments), wave w the channels [64 w, 64 w + 64):
//   * a lane loads x for the 32 channel/pixel pairs that are BOTH its B-operand slots of the score GEMM and its accumulator slots of the
//     gain GEMM: the k-steps walk the channels in accumulator order -- k-step r of block cb = channels 32 cb + (r & 3) + 8 (r >> 2) +
//     {0, 4} for the lane halves -- so x is read once, stays in 32 registers, and meets the gain without any data movement;
//   * the partial scores of the waves (rows t of a 32x32 accumulator: 8 registers per lane) and sum x^2 meet in LDS (one barrier);
//   * the softmax over 16 latents is 8 values in the lane + 8 in lane ^ 32 (two cross-lane exchanges);
//   * the probabilities (times the layer-norm factor) are exactly the B-operand slots of the gain GEMM when its k-steps pair the latents
//     as the accumulator rows pair them: k-step j = latents (j & 3) + 8 (j >> 2) + {0, 4}.
// No LDS tables, no per-channel LDS reads, 17 + 16 FMAs per element replaced by 1/32 + 1/4 MFMA.  HBM-bound: x and residual in, y out.
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NW>
__global__ __launch_bounds__(NW * 64, NW == 4 ? 4 : 2) void duplex_attention_mfma_kernel(AttnParams p) {
    __shared__ float red[NW][9][64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const int n = blockIdx.y, f0 = blockIdx.x * 32;
    const int cw0 = wv * 64;
    const int64_t nb = (int64_t)n * p.c * p.f;
    const bool ep = p.has_ep != 0;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x + nb), 0, p.c * p.f * 4, 0x00020000);
    const float* resp = (ep && p.ep.residual) ? p.ep.residual + nb : nullptr;          // (absent operands: zero-size resources read zeros)
    const __amdgpu_buffer_rsrc_t rres = __builtin_amdgcn_make_buffer_rsrc((void*)(resp ? resp : p.x), 0, resp ? p.c * p.f * 4 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc((void*)p.wqc, 0, p.c * TMAX * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void*)(p.vwb + (int64_t)n * p.c * TMAX), 0, p.c * TMAX * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsp = __builtin_amdgcn_make_buffer_rsrc((void*)p.spos, 0, p.f * TMAX * 4, 0x00020000);
    const float* biasp = (ep && p.ep.bias) ? p.ep.bias : nullptr;
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)(biasp ? biasp : p.x), 0, biasp ? p.c * 4 : 0, 0x00020000);
    auto rowof = [](int r) { return (r & 3) + 8 * (r >> 2); };         // accumulator register -> row of the 32x32 block (+ 4 half)

    // ---- one burst of loads: x (kept), the score weights, the positional scores, the noise ----
    const unsigned xo = (unsigned)((cw0 + 4 * half) * p.f + f0 + l31) * 4u;
    const unsigned qo = l31 < TMAX ? (unsigned)((cw0 + 4 * half) * TMAX + l31) * 4u : 0xFFFFFFF0u;    // rows t >= 16 of the A operand are zero
    float xr[2][16], a1[2][16];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ch = 32 * cb + rowof(r);
            xr[cb][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, xo, ch * p.f * 4, 0));
            a1[cb][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rq, qo, ch * TMAX * 4, 0));
        }
    float sp[8];
#pragma unroll
    for (int r = 0; r < 8; ++r)
        sp[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsp, (unsigned)((f0 + l31) * TMAX + 4 * half) * 4u, rowof(r) * 4, 0));
    float nz = 0.f;
    if (ep && p.ep.noise) nz = p.ep.noise[(int64_t)(p.ep.noise_n > 1 ? n : 0) * p.f + f0 + l31] * (p.ep.noise_strength ? *p.ep.noise_strength : 1.f);

    // ---- scores of this wave's 64 channels, sum x^2 ----
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float sq = 0.f;
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[cb][r], xr[cb][r], acc, 0, 0, 0);
            sq += xr[cb][r] * xr[cb][r];
        }
    // the residual rows are requested now: their latency runs behind the reduction and the softmax
    float rr[2][16];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            rr[cb][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rres, xo, (32 * cb + rowof(r)) * p.f * 4, 0));
#pragma unroll
    for (int r = 0; r < 8; ++r) red[wv][r][lane] = acc[r];
    red[wv][8][lane] = sq;
    __syncthreads();
    float s[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        float v = sp[r];
#pragma unroll
        for (int w = 0; w < NW; ++w) v += red[w][r][lane];
        s[r] = v;
    }
    sq = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) sq += red[w][8][lane];
    sq += __shfl_xor(sq, 32, 64);                  // the lane halves hold different channels
    // ---- softmax over the 16 latents: 8 here (t = rowof(r) + 4 half), 8 in lane ^ 32 ----
    float m = s[0];
    int best = 4 * half;
#pragma unroll
    for (int r = 1; r < 8; ++r) {
        if (s[r] > m) { m = s[r]; best = rowof(r) + 4 * half; }
    }
    {
        const float om = __shfl_xor(m, 32, 64);
        const int ob = __shfl_xor(best, 32, 64);
        if (om > m || (om == m && ob < best)) { m = om; best = ob; }      // first maximum, like the sequential scan
    }
    float den = 0.f;
#pragma unroll
    for (int r = 0; r < 8; ++r) { s[r] = __expf(s[r] - m); den += s[r]; }
    den += __shfl_xor(den, 32, 64);
    const float inv = 1.f / den;
    const float rs = rsqrtf(sq / (float)p.c + 1e-8f);
    if (wv == 0) {
        if (p.probs) {
#pragma unroll
            for (int r = 0; r < 8; ++r) p.probs[((int64_t)n * p.f + f0 + l31) * TMAX + rowof(r) + 4 * half] = s[r] * inv;
        }
        if (p.argmax && half == 0) p.argmax[(int64_t)n * p.f + f0 + l31] = best;
    }
    const float ps = inv * rs;                      // the layer norm folded into the probabilities
#pragma unroll
    for (int r = 0; r < 8; ++r) s[r] *= ps;

    // ---- gain = vwb . p per 32-channel block, y = x * gain, epilogue ----
    float* yn = p.y + nb;
    // (branch-free epilogue: no epilogue = noise / bias / residual read as zeros above, slope and gain 1)
    const float slope = !ep ? 1.f : (p.ep.act == MGF_ACT_LRELU ? p.ep.alpha : (p.ep.act == MGF_ACT_RELU ? 0.f : 1.f));
    const float gain = ep ? p.ep.gain : 1.f;
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
        // A operand of k-step j: vwb[channel cw0 + 32 cb + l31][latent rowof(j) + 4 half] = two 16-byte pieces of the channel's row
        const unsigned vo = (unsigned)((cw0 + 32 * cb + l31) * TMAX + 4 * half) * 4u;
        const float4 va = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rv, vo, 0, 0));
        const float4 vb = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rv, vo + 32u, 0, 0));
        float bv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r)
            bv[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rb, (unsigned)(cw0 + 32 * cb + 4 * half) * 4u, rowof(r) * 4, 0));
        const float a2[8] = {va.x, va.y, va.z, va.w, vb.x, vb.y, vb.z, vb.w};
        f32x16 g;
#pragma unroll
        for (int r = 0; r < 16; ++r) g[r] = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) g = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[j], s[j], g, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float v = xr[cb][r] * g[r];
            v += nz;
            v += bv[r];
            v = v > 0.f ? v : v * slope;
            v = v * gain + rr[cb][r];
            yn[(int64_t)(cw0 + 32 * cb + rowof(r) + 4 * half) * p.f + f0 + l31] = v;
        }
    }
}

// list2tensor (networks.py:1222-1242): one layer's attention map [n, s*s, t] replicated (nearest neighbour = upsample2d with the
// all-ones kernel) to the image resolution, written as slice `layer` of the stacked tensor [n, t, layers, 1, R, R]
__global__ __launch_bounds__(256) void att_map_upsample_kernel(float* out, const float* probs, int t, int s, int R, int layer, int layers,
                                                               int64_t total) {
    const int f = R / s;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int x = (int)(i % R);
        int64_t r = i / R;
        const int y = (int)(r % R);
        r /= R;
        const int tt = (int)(r % t);
        const int64_t n = r / t;
        out[(((n * t + tt) * layers + layer) * R + y) * (int64_t)R + x] = probs[(n * s * s + (int64_t)(y / f) * s + x / f) * t + tt];
    }
}

template <int PXB>
void launch_attention(const AttnParams& p, size_t lds, hipStream_t st) {
    constexpr int G = 256 / PXB;
    const int nch = (p.c % (G * UNR) == 0) ? p.c / G : 0;
    if (p.t == TMAX && (nch == 16 || nch == 32 || nch == 64) && (nch * G) % 64 == 0) {
        const size_t lds_r = ((size_t)p.c * TMAX + (size_t)(G + 1) * (TMAX + 1) * PXB) * sizeof(float);
        // large maps with few channels (the 128x128 x 256 layers): 4 pixel blocks per workgroup, both tables resident
        static const char* nb_env = mgf_knob("MGF_ATTN_NBLK");      // tuning hook (experiments only): 1 = one block per workgroup
        constexpr int NBLK = 4;
        if (PXB == 16 && nch == 16 && p.f >= 4096 && p.f % (PXB * NBLK) == 0 && !(nb_env && nb_env[0] == '1')) {
            const size_t lds_2 = ((size_t)2 * p.c * TMAX + (size_t)(G + 1) * (TMAX + 1) * PXB) * sizeof(float);
            hipLaunchKernelGGL((duplex_attention_blocks_kernel<PXB, 16, NBLK>), dim3((unsigned)(p.f / (PXB * NBLK)), p.n), dim3(256), lds_2, st, p);
            return;
        }
        const dim3 grid((unsigned)mgf_cdiv(p.f, PXB), p.n);
        if (nch == 16) hipLaunchKernelGGL((duplex_attention_reg_kernel<PXB, 16>), grid, dim3(256), lds_r, st, p);
        else if (nch == 32) hipLaunchKernelGGL((duplex_attention_reg_kernel<PXB, 32>), grid, dim3(256), lds_r, st, p);
        else hipLaunchKernelGGL((duplex_attention_reg_kernel<PXB, 64>), grid, dim3(256), lds_r, st, p);
        return;
    }
    const dim3 grid((unsigned)mgf_cdiv(p.f, PXB), p.n);
    hipLaunchKernelGGL((duplex_attention_kernel<PXB, 0>), grid, dim3(256), lds, st, p);
}

}  // namespace

extern "C" int mgf_duplex_attention(float* y, const float* x, const float* wqc, const float* spos, const float* vwb, int32_t n,
                                    int32_t c, int32_t f, int32_t t, const mgf_epilogue* ep, int32_t ep_w, float* probs,
                                    int32_t* argmax, mgf_stream_t stream) {
    (void)ep_w;
    MGF_REQUIRE(y && x && wqc && spos && vwb, MGF_EINVAL, "duplex_attention: null pointer");
    MGF_REQUIRE(n >= 1 && c >= 1 && f >= 1, MGF_EINVAL, "duplex_attention: bad shape");
    MGF_REQUIRE(t >= 1 && t <= TMAX, MGF_EUNSUPPORTED, "duplex_attention: supports 1..%d latent components (got %d)", TMAX, t);
    MGF_REQUIRE(n <= 65535 && (int64_t)n * c * f <= INT32_MAX, MGF_ETOOBIG, "duplex_attention: tensor too large");
    if (ep) MGF_REQUIRE(ep->act == 0 || ep->act == MGF_ACT_LINEAR || ep->act == MGF_ACT_LRELU || ep->act == MGF_ACT_RELU,
                        MGF_EUNSUPPORTED, "duplex_attention: epilogue activation %d unsupported", ep->act);
    AttnParams p;
    p.y = y; p.x = x; p.wqc = wqc; p.spos = spos; p.vwb = vwb; p.n = n; p.c = c; p.f = f; p.t = t;
    p.has_ep = ep != nullptr; p.probs = probs; p.argmax = argmax;
    if (ep) { p.ep = *ep; if (p.ep.act == 0) p.ep.act = MGF_ACT_LINEAR; } else { p.ep = mgf_epilogue{}; p.ep.gain = 1.f; }
    MGF_REQUIRE(((uintptr_t)wqc % 16 == 0) && ((uintptr_t)vwb % 16 == 0), MGF_EINVAL, "duplex_attention: tables must be 16-byte aligned");
    // the generator's layers (256 / 512 channels, 16 latents, whole 32-pixel tiles): the MFMA form.  MGF_ATTN_MFMA=0 keeps the register
    // kernels (tuning hook, tests).
    static const char* mf_env = mgf_knob("MGF_ATTN_MFMA");
    if (t == TMAX && (c == 256 || c == 512) && f % 32 == 0 && !(mf_env && mf_env[0] == '0')) {
        hipStream_t st0 = (hipStream_t)stream;
        if (c == 256) hipLaunchKernelGGL((duplex_attention_mfma_kernel<4>), dim3((unsigned)(f / 32), n), dim3(256), 0, st0, p);
        else hipLaunchKernelGGL((duplex_attention_mfma_kernel<8>), dim3((unsigned)(f / 32), n), dim3(512), 0, st0, p);
        MGF_CHECK_LAUNCH("duplex_attention");
        return MGF_OK;
    }
    static const char* pxb_env = mgf_knob("MGF_ATTN_PXB");      // tuning hook (experiments only)
    const int pxb = pxb_env ? atoi(pxb_env) : (f > 16384 ? 64 : 16);
    const int g = 256 / pxb;
    p.c_pad = (int)(mgf_cdiv(c, UNR * g) * UNR * g);
    const size_t lds = ((size_t)p.c_pad * TMAX + (size_t)g * (TMAX + 1) * pxb) * sizeof(float);
    MGF_REQUIRE(lds <= 64 * 1024, MGF_EUNSUPPORTED, "duplex_attention: %d channels need %zu bytes of LDS (> 64 KiB)", c, lds);
    hipStream_t st = (hipStream_t)stream;
    if (pxb == 64) launch_attention<64>(p, lds, st);
    else if (pxb == 32) launch_attention<32>(p, lds, st);
    else launch_attention<16>(p, lds, st);
    MGF_CHECK_LAUNCH("duplex_attention");
    return MGF_OK;
}

extern "C" int mgf_att_map_upsample_f32(float* out, const float* probs, int32_t n, int32_t side, int32_t t, int32_t out_res, int32_t layer,
                                        int32_t n_layers, mgf_stream_t stream) {
    MGF_REQUIRE(out && probs && n >= 1 && side >= 1 && t >= 1 && n_layers >= 1, MGF_EINVAL, "att_map_upsample: bad arguments");
    MGF_REQUIRE(layer >= 0 && layer < n_layers, MGF_EINVAL, "att_map_upsample: layer %d outside 0..%d", layer, n_layers - 1);
    MGF_REQUIRE(out_res >= side && out_res % side == 0, MGF_EINVAL, "att_map_upsample: the image resolution %d must be a multiple of the map side %d", out_res, side);
    const int64_t total = (int64_t)n * t * out_res * out_res;
    hipLaunchKernelGGL(att_map_upsample_kernel, dim3(mgf_stream_grid(total, 256, 4)), dim3(256), 0, (hipStream_t)stream, out, probs, t, side,
                       out_res, layer, n_layers, total);
    MGF_CHECK_LAUNCH("att_map_upsample");
    return MGF_OK;
}
