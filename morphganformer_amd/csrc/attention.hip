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
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        const int c = (k / UNR) * G * UNR + grp * UNR + (k % UNR);
        const float4* w4 = reinterpret_cast<const float4*>(tab + c * TMAX);
        const float xv = xreg[k];
        sq += xv * xv;
#pragma unroll
        for (int q = 0; q < TMAX / 4; ++q) {
            const float4 w = w4[q];
            s[4 * q + 0] += xv * w.x; s[4 * q + 1] += xv * w.y;
            s[4 * q + 2] += xv * w.z; s[4 * q + 3] += xv * w.w;
        }
    }
#pragma unroll
    for (int t = 0; t < TMAX; ++t) part[(grp * (TMAX + 1) + t) * PXB + px] = s[t];
    part[(grp * (TMAX + 1) + TMAX) * PXB + px] = sq;
    __syncthreads();                               // all reads of the wqc table are done: overwrite it with vwb
#pragma unroll
    for (int u = 0; u < TV; ++u) t4[tid + 256 * u] = tv[u];

    float m = -3.0e38f;
#pragma unroll
    for (int t = 0; t < TMAX; ++t) {
        float v = 0.f;
        for (int g = 0; g < G; ++g) v += part[(g * (TMAX + 1) + t) * PXB + px];
        v += sp[t];
        m = fmaxf(m, v);
        s[t] = v;
    }
    sq = 0.f;
    for (int g = 0; g < G; ++g) sq += part[(g * (TMAX + 1) + TMAX) * PXB + px];
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
    for (int t = 0; t < TMAX; ++t) s[t] *= rs;
    __syncthreads();

    float* yn = p.y + (int64_t)n * p.c * p.f;
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        const int c = (k / UNR) * G * UNR + grp * UNR + (k % UNR);
        const float4* w4 = reinterpret_cast<const float4*>(tab + c * TMAX);
        float g = 0.f;
#pragma unroll
        for (int q = 0; q < TMAX / 4; ++q) {
            const float4 w = w4[q];
            g += s[4 * q + 0] * w.x + s[4 * q + 1] * w.y + s[4 * q + 2] * w.z + s[4 * q + 3] * w.w;
        }
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
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const int c = (k / UNR) * G * UNR + grp * UNR + (k % UNR);
            const float4* w4 = reinterpret_cast<const float4*>(tab + c * TMAX);
            const float xv = xk[k];
            sq += xv * xv;
#pragma unroll
            for (int q = 0; q < TMAX / 4; ++q) {
                const float4 w = w4[q];
                s[4 * q + 0] += xv * w.x; s[4 * q + 1] += xv * w.y;
                s[4 * q + 2] += xv * w.z; s[4 * q + 3] += xv * w.w;
            }
        }
#pragma unroll
        for (int t = 0; t < TMAX; ++t) part[(grp * (TMAX + 1) + t) * PXB + px] = s[t];
        part[(grp * (TMAX + 1) + TMAX) * PXB + px] = sq;
        __syncthreads();                           // (one table: all reads of the wqc table are done, overwrite it with vwb)
        if (!TWO) {
#pragma unroll
            for (int u = 0; u < TV; ++u) t4[tid + 256 * u] = tv[u];
        }
        float m = -3.0e38f;
#pragma unroll
        for (int t = 0; t < TMAX; ++t) {
            float v = 0.f;
            for (int g = 0; g < G; ++g) v += part[(g * (TMAX + 1) + t) * PXB + px];
            v += spk[t];
            m = fmaxf(m, v);
            s[t] = v;
        }
        sq = 0.f;
        for (int g = 0; g < G; ++g) sq += part[(g * (TMAX + 1) + TMAX) * PXB + px];
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
        for (int t = 0; t < TMAX; ++t) s[t] *= rs;
        __syncthreads();                           // `part` is free for the next block (one table: vwb is in place)

        float* yn = p.y + (int64_t)n * p.c * p.f;
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const int c = (k / UNR) * G * UNR + grp * UNR + (k % UNR);
            const float4* w4 = reinterpret_cast<const float4*>(tabv + c * TMAX);
            float g = 0.f;
#pragma unroll
            for (int q = 0; q < TMAX / 4; ++q) {
                const float4 w = w4[q];
                g += s[4 * q + 0] * w.x + s[4 * q + 1] * w.y + s[4 * q + 2] * w.z + s[4 * q + 3] * w.w;
            }
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
        const size_t lds_r = ((size_t)p.c * TMAX + (size_t)G * (TMAX + 1) * PXB) * sizeof(float);
        // large maps with few channels (the 128x128 x 256 layers): 4 pixel blocks per workgroup, both tables resident
        static const char* nb_env = getenv("MGF_ATTN_NBLK");      // tuning hook (experiments only): 1 = one block per workgroup
        constexpr int NBLK = 4;
        if (PXB == 16 && nch == 16 && p.f >= 4096 && p.f % (PXB * NBLK) == 0 && !(nb_env && nb_env[0] == '1')) {
            const size_t lds_2 = ((size_t)2 * p.c * TMAX + (size_t)G * (TMAX + 1) * PXB) * sizeof(float);
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
    static const char* pxb_env = getenv("MGF_ATTN_PXB");      // tuning hook (experiments only)
    const int pxb = pxb_env ? atoi(pxb_env) : (f > 16384 ? 64 : 16);
    const int g = 256 / pxb;
    p.c_pad = (int)(mgf_cdiv(c, UNR * g) * UNR * g);
    const size_t lds = ((size_t)p.c_pad * TMAX + (size_t)g * (TMAX + 1) * pxb) * sizeof(float);
    MGF_REQUIRE(lds <= 64 * 1024, MGF_EUNSUPPORTED, "duplex_attention: %d channels need %zu bytes of LDS (> 64 KiB)", c, lds);
    hipStream_t st = (hipStream_t)stream;
    if (pxb == 64) launch_attention<64>(p, lds, st);
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
