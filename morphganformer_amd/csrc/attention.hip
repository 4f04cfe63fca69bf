// Duplex (image <- latents) attention of a SynthesisLayer for gfx950, re-associated so that no F x C x C GEMM remains.
// Contract: include/mgf.h (mgf_duplex_attention).  Reference: TransformerLayer.forward / integrate / att_norm
// (training/networks.py:748-822, 657-672, 341-358) + the noise / bias_act tail of SynthesisLayer.forward (:1036-1040).
//
//   S[f,t] = sum_c x[c,f] * wqc[c,t] + spos[f,t]       (query projection, positional term, att_weight, centroids and 1/sqrt(C)
//                                                        are folded into wqc [C,T] and spos [F,T] once per checkpoint)
//   P      = softmax_t(S)
//   y[c,f] = epilogue( x[c,f] * rsqrt(mean_c x^2 + 1e-8) * sum_t P[f,t] * vwb[c,t] )   (vwb = V Wm^T + bm + 1 per sample)
//
// HBM/L2-bound streaming kernel.  A workgroup owns 64 consecutive pixels (one 256-byte row segment per channel); wave w
// accumulates the 16 scores and the second moment over channels w, w+4, ...: one coalesced 256-B load + 17 FMAs per
// channel with the 16 wqc scalars of that channel fetched by scalar loads (wave-uniform).  Partials meet in LDS, every lane
// finishes the softmax of its own pixel in registers, then the waves sweep their channels again (L2-resident re-read).
#include "mgf_common.h"

namespace {

constexpr int TMAX = 16;

struct AttnParams {
    float* y;
    const float* x;
    const float* wqc;     // [c, t]
    const float* spos;    // [f, t]
    const float* vwb;     // [n, c, t]
    int n, c, f, t;
    mgf_epilogue ep;
    int has_ep;
    int ep_w;             // image width (noise is indexed [n, f] flat, so only f matters)
    float* probs;         // [n, f, t] or null
    int32_t* argmax;      // [n, f] or null
};

__global__ __launch_bounds__(256) void duplex_attention_kernel(AttnParams p) {
    __shared__ float part[4][TMAX + 1][64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = blockIdx.y;
    const int f0 = blockIdx.x * 64;
    const int f = f0 + lane;
    const bool valid = f < p.f;
    const int fc = valid ? f : p.f - 1;
    const float* xn = p.x + (int64_t)n * p.c * p.f;
    const int T = p.t;

    float s[TMAX];
#pragma unroll
    for (int t = 0; t < TMAX; ++t) s[t] = 0.f;
    float sq = 0.f;
    for (int c = wave; c < p.c; c += 4) {
        const float xv = xn[(int64_t)c * p.f + fc];
        const float* wr = p.wqc + (int64_t)c * T;       // wave-uniform address -> scalar loads
        sq += xv * xv;
#pragma unroll
        for (int t = 0; t < TMAX; ++t)
            if (t < T) s[t] += xv * wr[t];
    }
#pragma unroll
    for (int t = 0; t < TMAX; ++t) part[wave][t][lane] = s[t];
    part[wave][TMAX][lane] = sq;
    __syncthreads();
    float m = -3.0e38f;
#pragma unroll
    for (int t = 0; t < TMAX; ++t) {
        if (t < T) {
            s[t] = part[0][t][lane] + part[1][t][lane] + part[2][t][lane] + part[3][t][lane] + p.spos[(int64_t)fc * T + t];
            m = fmaxf(m, s[t]);
        }
    }
    sq = part[0][TMAX][lane] + part[1][TMAX][lane] + part[2][TMAX][lane] + part[3][TMAX][lane];
    float den = 0.f;
    int best = 0;
    float bestv = -3.0e38f;
#pragma unroll
    for (int t = 0; t < TMAX; ++t) {
        if (t < T) {
            if (s[t] > bestv) { bestv = s[t]; best = t; }
            s[t] = __expf(s[t] - m);
            den += s[t];
        }
    }
    const float inv = 1.f / den;
    const float rs = rsqrtf(sq / (float)p.c + 1e-8f);
#pragma unroll
    for (int t = 0; t < TMAX; ++t) s[t] = t < T ? s[t] * inv : 0.f;
    if (wave == 0 && valid) {
        if (p.probs)
            for (int t = 0; t < T; ++t) p.probs[((int64_t)n * p.f + f) * T + t] = s[t];
        if (p.argmax) p.argmax[(int64_t)n * p.f + f] = best;
    }
    // fold the norm into the probabilities once
#pragma unroll
    for (int t = 0; t < TMAX; ++t) s[t] *= rs;

    float nz = 0.f;
    if (p.has_ep && p.ep.noise) {
        const float ns = p.ep.noise_strength ? *p.ep.noise_strength : 1.f;
        nz = p.ep.noise[(int64_t)(p.ep.noise_n > 1 ? n : 0) * p.f + fc] * ns;
    }
    float* yn = p.y + (int64_t)n * p.c * p.f;
    const float* vn = p.vwb + (int64_t)n * p.c * T;
    const float* rn = (p.has_ep && p.ep.residual) ? p.ep.residual + (int64_t)n * p.c * p.f : nullptr;
    for (int c = wave; c < p.c; c += 4) {
        const float xv = xn[(int64_t)c * p.f + fc];
        const float* vr = vn + (int64_t)c * T;
        float g = 0.f;
#pragma unroll
        for (int t = 0; t < TMAX; ++t)
            if (t < T) g += s[t] * vr[t];
        float v = xv * g;
        if (p.has_ep) {
            v += nz;
            if (p.ep.bias) v += p.ep.bias[c];
            if (p.ep.act == MGF_ACT_LRELU) v = v > 0.f ? v : v * p.ep.alpha;
            else if (p.ep.act == MGF_ACT_RELU) v = v > 0.f ? v : 0.f;
            v *= p.ep.gain;
            if (rn) v += rn[(int64_t)c * p.f + fc];
        }
        if (valid) yn[(int64_t)c * p.f + f] = v;
    }
}

}  // namespace

extern "C" int mgf_duplex_attention(float* y, const float* x, const float* wqc, const float* spos, const float* vwb, int32_t n,
                                    int32_t c, int32_t f, int32_t t, const mgf_epilogue* ep, int32_t ep_w, float* probs,
                                    int32_t* argmax, mgf_stream_t stream) {
    MGF_REQUIRE(y && x && wqc && spos && vwb, MGF_EINVAL, "duplex_attention: null pointer");
    MGF_REQUIRE(n >= 1 && c >= 1 && f >= 1, MGF_EINVAL, "duplex_attention: bad shape");
    MGF_REQUIRE(t >= 1 && t <= TMAX, MGF_EUNSUPPORTED, "duplex_attention: supports 1..%d latent components (got %d)", TMAX, t);
    MGF_REQUIRE(n <= 65535 && (int64_t)n * c * f <= INT32_MAX, MGF_ETOOBIG, "duplex_attention: tensor too large");
    if (ep) MGF_REQUIRE(ep->act == 0 || ep->act == MGF_ACT_LINEAR || ep->act == MGF_ACT_LRELU || ep->act == MGF_ACT_RELU,
                        MGF_EUNSUPPORTED, "duplex_attention: epilogue activation %d unsupported", ep->act);
    AttnParams p;
    p.y = y; p.x = x; p.wqc = wqc; p.spos = spos; p.vwb = vwb; p.n = n; p.c = c; p.f = f; p.t = t;
    p.has_ep = ep != nullptr; p.ep_w = ep_w; p.probs = probs; p.argmax = argmax;
    if (ep) { p.ep = *ep; if (p.ep.act == 0) p.ep.act = MGF_ACT_LINEAR; } else { p.ep = mgf_epilogue{}; p.ep.gain = 1.f; }
    hipLaunchKernelGGL(duplex_attention_kernel, dim3((unsigned)mgf_cdiv(f, 64), n), dim3(256), 0, (hipStream_t)stream, p);
    MGF_CHECK_LAUNCH("duplex_attention");
    return MGF_OK;
}
