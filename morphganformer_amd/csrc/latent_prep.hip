// Per-iteration latent-side preparation kernels (tiny, launch-latency bound -> batched over layers):
//   * styles + demodulation coefficients of every modulated conv (training/networks.py:131-150,288-291,1022,1056-1059)
//   * folded attention value tables vwb = V Wm^T + bm + 1 (networks.py:759,812-814,662-668 re-associated)
//   * the mapping network z -> w (networks.py:894-942, MLP :179-221, ResnetLayer :154-172, latent self-attention :748-822)
// Contract: include/mgf.h.
#include "mgf_common.h"

namespace {

// ------------------------------------------------------------------------------------------ styles + demod
// grid = (co_blocks, njobs, n); 256 lanes.  Every block recomputes the (cheap) style vector into LDS, block x == 0
// publishes it, then the 4 waves walk this block's share of wsq rows: lanes stride over ci (coalesced), wave-reduce.
__device__ void style_demod_body(const mgf_style_job& j, const float* ws, int64_t ws_stride_n, int wdim, int n, float* s_lds,
                                 bool styles_only = false) {
    const int tid = threadIdx.x;
    const float* wg = ws + (int64_t)n * ws_stride_n + j.w_offset;
    for (int ci = tid; ci < j.cin; ci += blockDim.x) {
        const float* row = j.aff_w + (int64_t)ci * wdim;
        float acc = 0.f;
        for (int k = 0; k < wdim; ++k) acc += wg[k] * row[k];
        float s = (acc * j.aff_gain + j.aff_b[ci]) * j.style_gain;
        s_lds[ci] = s;
        if (blockIdx.x == 0) j.s[(int64_t)n * j.cin + ci] = s;
    }
    if (!j.wsq || !j.d || styles_only) return;
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6, nwaves = blockDim.x >> 6;
    const int per_block = (j.cout + gridDim.x - 1) / gridDim.x;
    const int co_begin = blockIdx.x * per_block;
    const int co_end = min(j.cout, co_begin + per_block);
    for (int co = co_begin + wave; co < co_end; co += nwaves) {
        const float* row = j.wsq + (int64_t)co * j.cin;
        float acc = 0.f;
        for (int ci = lane; ci < j.cin; ci += 64) { float s = s_lds[ci]; acc += row[ci] * s * s; }
        acc = wave_sum(acc);
        if (lane == 0) j.d[(int64_t)n * j.cout + co] = rsqrtf(acc + 1e-8f);
    }
}

__global__ __launch_bounds__(256) void style_demod_multi_kernel(const mgf_style_job* jobs, const float* ws, int64_t ws_stride_n, int wdim,
                                                                int styles_only) {
    __shared__ float s_lds[2048];
    const mgf_style_job j = jobs[blockIdx.y];
    style_demod_body(j, ws, ws_stride_n, wdim, blockIdx.z, s_lds, styles_only != 0);
}

// Batched demodulation: grid = (co_blocks, njobs); ONE workgroup serves all samples.  The styles of every sample (computed by a
// styles-only launch just before) go to LDS ([n][cin]), then
// each wave streams its rows of wsq ONCE and accumulates the n demodulation sums side by side (the per-sample form re-reads the
// 1 MB table of a 512-channel layer for every sample).  NB <= 32 samples, n * cin floats of dynamic LDS.
constexpr int SD_NB = 32;
__global__ __launch_bounds__(256) void style_demod_batched_kernel(const mgf_style_job* jobs, const float* ws, int64_t ws_stride_n, int wdim, int n) {
    extern __shared__ float s_all[];                 // [n][cin], squared styles after the publish step
    const mgf_style_job j = jobs[blockIdx.y];
    const int tid = threadIdx.x;
    if (!j.wsq || !j.d) return;
    for (int i = tid; i < n * j.cin; i += 256) {                  // styles were published by the preceding styles-only launch
        const float sv = j.s[i];
        s_all[i] = sv * sv;
    }
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6;
    const int per_block = (j.cout + gridDim.x - 1) / gridDim.x;
    const int co_begin = blockIdx.x * per_block;
    const int co_end = min(j.cout, co_begin + per_block);
    for (int co = co_begin + wave; co < co_end; co += 4) {
        const float* row = j.wsq + (int64_t)co * j.cin;
        float acc[SD_NB];
#pragma unroll
        for (int q = 0; q < SD_NB; ++q) acc[q] = 0.f;
        for (int ci = lane; ci < j.cin; ci += 64) {
            const float wv = row[ci];
#pragma unroll
            for (int q = 0; q < SD_NB; ++q)
                if (q < n) acc[q] += wv * s_all[q * j.cin + ci];
        }
#pragma unroll
        for (int q = 0; q < SD_NB; ++q) {
            if (q < n) {
                const float v = wave_sum(acc[q]);
                if (lane == 0) j.d[(int64_t)q * j.cout + co] = rsqrtf(v + 1e-8f);
            }
        }
    }
}

__global__ __launch_bounds__(256) void style_demod_single_kernel(mgf_style_job j, const float* ws, int64_t ws_stride_n, int wdim) {
    __shared__ float s_lds[2048];
    style_demod_body(j, ws, ws_stride_n, wdim, blockIdx.z, s_lds);
}

// ------------------------------------------------------------------------------------------ demodulation alone (operator API)
// torch_utils.ops.conv2d_resample.modulated_conv2d receives the styles as a tensor (networks.py:253-328 computes them outside): only the
// demodulation coefficients are left to do, d[n,co] = rsqrt(sum_ci wsq[co,ci] s[n,ci]^2 + 1e-8), and their adjoint
// ds[n,ci] = -s[n,ci] sum_co dd[n,co] d[n,co]^3 wsq[co,ci].  grid = (blocks, n).
__global__ __launch_bounds__(256) void demod_kernel(float* __restrict__ d, const float* __restrict__ s, const float* __restrict__ wsq, int cin, int cout) {
    const int n = blockIdx.y, lane = threadIdx.x & 63;
    const float* sn = s + (int64_t)n * cin;
    for (int co = blockIdx.x * 4 + (threadIdx.x >> 6); co < cout; co += gridDim.x * 4) {
        const float* row = wsq + (int64_t)co * cin;
        float acc = 0.f;
        for (int ci = lane; ci < cin; ci += 64) { const float sv = sn[ci]; acc += row[ci] * sv * sv; }
        acc = wave_sum(acc);
        if (lane == 0) d[(int64_t)n * cout + co] = rsqrtf(acc + 1e-8f);
    }
}

__global__ __launch_bounds__(256) void demod_bwd_kernel(float* __restrict__ ds, const float* __restrict__ dd, const float* __restrict__ d,
                                                        const float* __restrict__ s, const float* __restrict__ wsq, int cin, int cout) {
    const int n = blockIdx.y;
    const int ci = blockIdx.x * 256 + threadIdx.x;
    if (ci >= cin) return;
    const float* ddn = dd + (int64_t)n * cout;
    const float* dn = d + (int64_t)n * cout;
    float acc = 0.f;
    for (int co = 0; co < cout; ++co) { const float dv = dn[co]; acc += ddn[co] * dv * dv * dv * wsq[(int64_t)co * cin + ci]; }
    ds[(int64_t)n * cin + ci] = -s[(int64_t)n * cin + ci] * acc;
}

// ------------------------------------------------------------------------------------------ attention value tables
// vwb[n][c][t] (t fastest: 16 contiguous scalars per channel for the attention kernel's scalar loads)
//   = sum_j ycomp[n,t,j] * wmv[c,j] + bmv[c]
__device__ void attn_values_body(const mgf_attn_job& j, const float* ws, int64_t ws_stride_n, int64_t ws_stride_t, int wdim, int t_len, int n) {
    // the sample's latent components go to LDS once; a thread then owns one channel: its weight row is read once (float4 when
    // aligned) and reused for all t_len components, the t_len results are contiguous in vwb
    __shared__ float ys[16 * 64];
    const bool fast = t_len <= 16 && wdim <= 64 && wdim % 4 == 0 && ((uintptr_t)j.wmv % 16) == 0;
    if (!fast) {
        const int total = j.c * t_len;
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
            const int t = i % t_len, c = i / t_len;
            const float* y = ws + (int64_t)n * ws_stride_n + (int64_t)t * ws_stride_t + j.w_offset;
            const float* row = j.wmv + (int64_t)c * wdim;
            float acc = 0.f;
            for (int k = 0; k < wdim; ++k) acc += y[k] * row[k];
            j.vwb[((int64_t)n * j.c + c) * t_len + t] = acc + j.bmv[c];
        }
        return;
    }
    for (int i = threadIdx.x; i < t_len * wdim; i += blockDim.x)
        ys[i] = ws[(int64_t)n * ws_stride_n + (int64_t)(i / wdim) * ws_stride_t + j.w_offset + i % wdim];
    __syncthreads();
    for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < j.c; c += gridDim.x * blockDim.x) {
        const float4* row = reinterpret_cast<const float4*>(j.wmv + (int64_t)c * wdim);
        float acc[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) acc[t] = 0.f;
        for (int k4 = 0; k4 < wdim / 4; ++k4) {
            const float4 wv = row[k4];
#pragma unroll
            for (int t = 0; t < 16; ++t)
                if (t < t_len) {
                    const float* y = ys + t * wdim + 4 * k4;
                    acc[t] += y[0] * wv.x + y[1] * wv.y + y[2] * wv.z + y[3] * wv.w;
                }
        }
        const float bb = j.bmv[c];
        float* o = j.vwb + ((int64_t)n * j.c + c) * t_len;
#pragma unroll
        for (int t = 0; t < 16; ++t)
            if (t < t_len) o[t] = acc[t] + bb;
    }
}

__global__ __launch_bounds__(256) void attn_values_multi_kernel(const mgf_attn_job* jobs, const float* ws, int64_t sn, int64_t stt, int wdim, int t_len) {
    const mgf_attn_job j = jobs[blockIdx.y];
    attn_values_body(j, ws, sn, stt, wdim, t_len, blockIdx.z);
}
__global__ __launch_bounds__(256) void attn_values_single_kernel(mgf_attn_job j, const float* ws, int64_t sn, int64_t stt, int wdim, int t_len) {
    attn_values_body(j, ws, sn, stt, wdim, t_len, blockIdx.z);
}

// ------------------------------------------------------------------------------------------ mapping network
// One workgroup (256 lanes) per sample.  All activations live in LDS; every FC is a [rows x 32] . [32 x 32]^T product with the
// (gain-folded) weight matrix read from the packed parameter blob (L2-resident, 100 KB).
constexpr int MD = 32;          // latent width handled by this kernel
constexpr int MT_MAX = 32;      // max local components

struct MapLayout {              // float offsets inside the blob (see engine.py: pack_mapping_params)
    // global mlp: per res layer {W0,b0,W1,b1}, then {Wout,bout}
    // local  mlp: per res layer {Wq,bq_pos[T*D],Wk,bk_pos[T*D],Wv,bv,Wm,bm,W0,b0,W1,b1}, then {Wout,bout}
};

__device__ __forceinline__ float lrelu02(float v) { return v > 0.f ? v : 0.2f * v; }

// One D x D layer the way the thread that uses it holds it: row o of W for the forward product out[r][o] = <in[r], W[o]> + b, or
// column i for the backward product out[r][i] = sum_o in[r][o] W[o][i], with o / i = tid % D -- the same for each of the thread's
// outputs (rows tid/D, tid/D + 8, ...) -- and those outputs' biases.  A layer's matrices are all loaded at its top, ahead of use:
// the network is a chain of ~50 tiny dependent products run by one workgroup, and this leaves one exposed memory latency per
// layer instead of one per product.  Sums run i = 0..D-1 from zero and the bias is added last, in every form of these kernels.
constexpr int MAP_BLOCK = 256;
constexpr int MAP_PASSES = MT_MAX * MD / MAP_BLOCK;       // outputs per thread of a [T x D] product
constexpr int MAP_MAX_RES = 7;                            // the global path keeps its 2 n_res + 1 weights in 8 groups x 2 register sets
struct FcW { float w[MD]; float b[MAP_PASSES]; };

__device__ __forceinline__ void load_fc_row(FcW& f, const float* W, const float* b, bool row_bias, int rows) {
    const int o = threadIdx.x & (MD - 1), r0 = threadIdx.x / MD;
    const float4* p = (const float4*)(W + o * MD);
#pragma unroll
    for (int j = 0; j < MD / 4; ++j) {
        const float4 q = p[j];
        f.w[4 * j] = q.x; f.w[4 * j + 1] = q.y; f.w[4 * j + 2] = q.z; f.w[4 * j + 3] = q.w;
    }
#pragma unroll
    for (int j = 0; j < MAP_PASSES; ++j) {
        const int r = r0 + (MAP_BLOCK / MD) * j;
        f.b[j] = row_bias ? (r < rows ? b[r * MD + o] : 0.f) : b[o];
    }
}
__device__ __forceinline__ void load_fc_col(FcW& f, const float* W) {
    const int i = threadIdx.x & (MD - 1);
#pragma unroll
    for (int o = 0; o < MD; ++o) f.w[o] = W[o * MD + i];
}

// emit(idx, j, <in[idx / D], f.w> (+ bias)) for the thread's outputs idx = tid + 256 j < rows * D
template <bool BIAS, class F>
__device__ __forceinline__ void fc_apply(const float* in, const FcW& f, int rows, F&& emit) {
#pragma unroll
    for (int j = 0; j < MAP_PASSES; ++j) {
        const int idx = threadIdx.x + MAP_BLOCK * j;
        if (idx < rows * MD) {
            const float* xr = in + (idx / MD) * MD;
            float acc = 0.f;
#pragma unroll
            for (int i = 0; i < MD; ++i) acc += xr[i] * f.w[i];
            emit(idx, j, BIAS ? acc + f.b[j] : acc);
        }
    }
}
// one row (the global component): the 32 lanes of one group
__device__ __forceinline__ float fc_row1(const float* in, const FcW& f) {
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < MD; ++i) acc += in[i] * f.w[i];
    return acc;
}

struct MapShared {
    float X[MT_MAX * MD], Xin[MT_MAX * MD], Q[MT_MAX * MD], K[MT_MAX * MD], V[MT_MAX * MD], H[MT_MAX * MD];
    float Pr[MT_MAX * MT_MAX];
    float G[MD], Gin[MD], GH[MD];
    float red[4];
};

// Per-sample scratch of the backward pass (floats): what the saving forward leaves behind.
//   global path, per res layer: {H0 (post-lrelu fc0), Xo (layer output)} = 2 D
//   local  path, per res layer: {Q, K, V, H0, Xo} (T D each), P (T T)
struct MapSave {
    int T, n_res;
    __host__ __device__ int64_t glayer(int l) const { return (int64_t)l * 2 * MD; }
    __host__ __device__ int64_t lbase() const { return (int64_t)n_res * 2 * MD; }
    __host__ __device__ int64_t lstride() const { return (int64_t)5 * T * MD + (int64_t)T * T; }
    __host__ __device__ int64_t llayer(int l) const { return lbase() + l * lstride(); }
    __host__ __device__ int64_t total() const { return lbase() + n_res * lstride(); }
};

constexpr float MAP_SQ2 = 1.41421356237309515f;
constexpr int MAP_WSZ = MD * MD;
constexpr int MAP_GSTRIDE = 2 * MAP_WSZ + 2 * MD;                          // global res layer {W0,b0,W1,b1}
__device__ __forceinline__ int64_t map_lstride(int T) { return 6 * (int64_t)MAP_WSZ + 2 * (int64_t)T * MD + 4 * MD; }
__device__ __forceinline__ const float* map_local_params(const float* P, int n_res) { return P + n_res * MAP_GSTRIDE + MAP_WSZ + MD; }

// The global component and the T local components never meet inside the mapping network: each sample runs as two workgroups
// (blockIdx.y: 0 = local MLP with latent self-attention, 1 = global MLP), forward and backward.
//
// Global path, forward.  Its 2 n_res + 1 products have one row each: group g (32 lanes) of the workgroup keeps the weights of
// products g and g + 8 in registers (loaded at the start, all in flight together) and runs them; the row travels through LDS.
// SAVE additionally stores the per-layer activations the backward pass needs into `sv` and the normalisation factor.
template <bool SAVE>
__device__ __forceinline__ void mapping_forward_global(MapShared& sh, float* w_n, float* w_copy, const float* zn, const float* P, int k, int n_res,
                                       int normalize_global, float* sv, float* norm_out) {
    float *G = sh.G, *GH = sh.GH;
    const int tid = threadIdx.x, g = tid / MD, o = tid & (MD - 1);
    const int T = k - 1, nfc = 2 * n_res + 1;
    const MapSave ms{T, n_res};
    auto fcw = [&](int j) { return j < 2 * n_res ? P + (j >> 1) * MAP_GSTRIDE + (j & 1) * (MAP_WSZ + MD) : P + n_res * MAP_GSTRIDE; };
    FcW s0, s1;
    if (g < nfc) load_fc_row(s0, fcw(g), fcw(g) + MAP_WSZ, false, 1);
    if (g + 8 < nfc) load_fc_row(s1, fcw(g + 8), fcw(g + 8) + MAP_WSZ, false, 1);
    if (tid < 64) {
        float v = tid < MD ? zn[T * MD + tid] : 0.f;
        float ss = wave_sum(v * v);
        float fg = normalize_global ? rsqrtf(ss / (float)MD + 1e-8f) : 1.f;
        if (tid < MD) G[tid] = v * fg;
        if (SAVE && tid == 0) norm_out[1] = fg;
    }
    __syncthreads();
    for (int l = 0; l < n_res; ++l) {
        int j = 2 * l;
        if (g == (j & 7)) {
            const float v = (j >> 3) ? fc_row1(G, s1) + s1.b[0] : fc_row1(G, s0) + s0.b[0];
            const float h = lrelu02(v) * MAP_SQ2;
            GH[o] = h;
            if (SAVE) sv[ms.glayer(l) + o] = h;
        }
        __syncthreads();
        j = 2 * l + 1;
        if (g == (j & 7)) {
            const float v = (j >> 3) ? fc_row1(GH, s1) + s1.b[0] : fc_row1(GH, s0) + s0.b[0];
            const float x = lrelu02(v + G[o]);       // the product reads GH; G[o] is this lane's own element
            G[o] = x;
            if (SAVE) sv[ms.glayer(l) + MD + o] = x;
        }
        __syncthreads();
    }
    {
        const int j = 2 * n_res;
        if (g == (j & 7)) {
            const float v = (j >> 3) ? fc_row1(G, s1) + s1.b[0] : fc_row1(G, s0) + s0.b[0];
            const float w = lrelu02(v) * MAP_SQ2;
            w_n[T * MD + o] = w;
            if (w_copy) w_copy[T * MD + o] = w;
        }
    }
}

// Local path, forward.
template <bool SAVE>
__device__ __forceinline__ void mapping_forward_local(MapShared& sh, float* w_n, float* w_copy, const float* zn, const float* P, int k, int n_res,
                                      float* sv, float* norm_out) {
    float *X = sh.X, *Xin = sh.Xin, *Q = sh.Q, *K = sh.K, *V = sh.V, *H = sh.H, *Pr = sh.Pr;
    float* red = sh.red;
    const int tid = threadIdx.x;
    const int T = k - 1, TD = T * MD;
    const MapSave ms{T, n_res};
    const float* p = map_local_params(P, n_res);
    // ---- normalize (networks.py:30-37): joint second moment over the T x D local block ----
    float part = 0.f;
    for (int i = tid; i < TD; i += MAP_BLOCK) { float v = zn[i]; part += v * v; }
    part = wave_sum(part);
    if ((tid & 63) == 0) red[tid >> 6] = part;
    __syncthreads();
    const float fl = rsqrtf((red[0] + red[1] + red[2] + red[3]) / (float)TD + 1e-8f);
    for (int i = tid; i < TD; i += MAP_BLOCK) X[i] = zn[i] * fl;
    if (SAVE && tid == 0) norm_out[0] = fl;
    __syncthreads();
    // ---- res layers: latent self-attention, then two dense layers ----
    for (int l = 0; l < n_res; ++l) {
        const float* Wq = p;            const float* bq = Wq + MAP_WSZ;
        const float* Wk = bq + TD;      const float* bk = Wk + MAP_WSZ;
        const float* Wv = bk + TD;      const float* bv = Wv + MAP_WSZ;
        const float* Wm = bv + MD;      const float* bm = Wm + MAP_WSZ;
        const float* W0 = bm + MD;      const float* b0 = W0 + MAP_WSZ;
        const float* W1 = b0 + MD;      const float* b1 = W1 + MAP_WSZ;
        p = b1 + MD;
        FcW fq, fk, fv, fm, f0, f1;
        load_fc_row(fq, Wq, bq, true, T);            // 1/sqrt(D) and the positional term are folded into Wq / bq
        load_fc_row(fk, Wk, bk, true, T);
        load_fc_row(fv, Wv, bv, false, T);
        load_fc_row(fm, Wm, bm, false, T);
        load_fc_row(f0, W0, b0, false, T);
        load_fc_row(f1, W1, b1, false, T);
        float* svl = SAVE ? sv + ms.llayer(l) : nullptr;
        for (int i = tid; i < TD; i += MAP_BLOCK) Xin[i] = X[i];
        fc_apply<true>(X, fq, T, [&](int i, int, float v) { Q[i] = v; if (SAVE) svl[i] = v; });
        fc_apply<true>(X, fk, T, [&](int i, int, float v) { K[i] = v; if (SAVE) svl[TD + i] = v; });
        fc_apply<true>(X, fv, T, [&](int i, int, float v) { V[i] = v; if (SAVE) svl[2 * TD + i] = v; });
        __syncthreads();
        for (int idx = tid; idx < T * T; idx += MAP_BLOCK) {
            const int a = idx / T, b = idx % T;
            float acc = 0.f;
#pragma unroll
            for (int i = 0; i < MD; ++i) acc += Q[a * MD + i] * K[b * MD + i];
            Pr[a * MT_MAX + b] = acc;
        }
        __syncthreads();
        if (tid < T) {
            float m = -3.0e38f;
            for (int b = 0; b < T; ++b) m = fmaxf(m, Pr[tid * MT_MAX + b]);
            float s = 0.f;
            for (int b = 0; b < T; ++b) { float e = expf(Pr[tid * MT_MAX + b] - m); Pr[tid * MT_MAX + b] = e; s += e; }
            const float inv = 1.f / s;
            for (int b = 0; b < T; ++b) {
                Pr[tid * MT_MAX + b] *= inv;
                if (SAVE) svl[5 * TD + tid * T + b] = Pr[tid * MT_MAX + b];
            }
        }
        __syncthreads();
        for (int idx = tid; idx < TD; idx += MAP_BLOCK) {
            const int a = idx / MD, o = idx % MD;
            float acc = 0.f;
            for (int b = 0; b < T; ++b) acc += Pr[a * MT_MAX + b] * V[b * MD + o];
            H[idx] = acc;
        }
        __syncthreads();
        fc_apply<true>(H, fm, T, [&](int i, int, float v) { X[i] += v; });                   // integration 'add'
        __syncthreads();
        fc_apply<true>(X, f0, T, [&](int i, int, float v) {
            const float h = lrelu02(v) * MAP_SQ2;
            H[i] = h;
            if (SAVE) svl[3 * TD + i] = h;
        });
        __syncthreads();
        fc_apply<true>(H, f1, T, [&](int i, int, float v) {
            const float x = lrelu02(v + Xin[i]);
            X[i] = x;
            if (SAVE) svl[4 * TD + i] = x;
        });
        __syncthreads();
    }
    {
        const float *Wo = p, *bo = p + MAP_WSZ;
        FcW fo;
        load_fc_row(fo, Wo, bo, false, T);
        fc_apply<true>(X, fo, T, [&](int i, int, float v) {
            const float w = lrelu02(v) * MAP_SQ2;
            w_n[i] = w;
            if (w_copy) w_copy[i] = w;
        });
    }
}

__global__ __launch_bounds__(MAP_BLOCK) void mapping_kernel(float* w, const float* z, const float* P, int k, int n_res, int normalize_global) {
    __shared__ MapShared sh;
    const int n = blockIdx.x;
    float* wn = w + (int64_t)n * k * MD;
    const float* zn = z + (int64_t)n * k * MD;
    if (blockIdx.y) mapping_forward_global<false>(sh, wn, nullptr, zn, P, k, n_res, normalize_global, nullptr, nullptr);
    else mapping_forward_local<false>(sh, wn, nullptr, zn, P, k, n_res, nullptr, nullptr);
}

// The forward with its per-layer activations kept in the backward kernel's scratch slab, so that the backward need not recompute
// it.  Scratch slab: [MapSave::total() activations][k*D copy of w][2 normalisation factors].
__global__ __launch_bounds__(MAP_BLOCK) void mapping_save_kernel(float* w, const float* z, const float* P, float* scratch, int64_t slab,
                                                                 int k, int n_res, int normalize_global) {
    __shared__ MapShared sh;
    const int n = blockIdx.x;
    const MapSave ms{k - 1, n_res};
    float* sv = scratch + (int64_t)n * slab;
    float* wrec = sv + ms.total();
    float* nrm = wrec + (int64_t)k * MD;
    float* wn = w + (int64_t)n * k * MD;
    const float* zn = z + (int64_t)n * k * MD;
    if (blockIdx.y) mapping_forward_global<true>(sh, wn, wrec, zn, P, k, n_res, normalize_global, sv, nrm);
    else mapping_forward_local<true>(sh, wn, wrec, zn, P, k, n_res, sv, nrm);
}

__device__ __forceinline__ float dlrelu02(float y) { return y > 0.f ? 1.f : 0.2f; }

struct MapBwdShared {
    float dX[MT_MAX * MD], A[MT_MAX * MD], sQ[MT_MAX * MD], sK[MT_MAX * MD], sV[MT_MAX * MD], sP[MT_MAX * MT_MAX];
};

// Global path, backward: the 2 n_res + 1 transposed products in reverse, held by the groups as in the forward (product jj = 0 is
// the out layer, then W1 and W0 of the res layers from the last to the first), each with the one saved activation its stage needs.
__device__ __forceinline__ void mapping_backward_global(MapShared& sh, float* dzn, const float* dwn, const float* zn, const float* P, const float* sv,
                                        const float* wrec, const float* nrm, int k, int n_res, int normalize_global) {
    float *dG = sh.G, *tA = sh.Gin, *tB = sh.GH;
    const int tid = threadIdx.x, g = tid / MD, o = tid & (MD - 1);
    const int T = k - 1, TD = T * MD, nfc = 2 * n_res + 1;
    const MapSave ms{T, n_res};
    // product jj: its weight, and the saved activation whose lrelu slope multiplies what the stage hands on
    auto fcw = [&](int jj) {
        if (jj == 0) return P + n_res * MAP_GSTRIDE;
        const int l = n_res - 1 - ((jj - 1) >> 1);
        return P + l * MAP_GSTRIDE + ((jj - 1) & 1 ? 0 : MAP_WSZ + MD);                  // odd jj: W1, even: W0
    };
    auto aux = [&](int jj) {
        if (jj == 0) return n_res > 0 ? sv[ms.glayer(n_res - 1) + MD + o] : 1.f;         // Xo of the last layer
        const int l = n_res - 1 - ((jj - 1) >> 1);
        if ((jj - 1) & 1) return l > 0 ? sv[ms.glayer(l - 1) + MD + o] : 1.f;            // after W0_l: Xo of layer l-1
        return sv[ms.glayer(l) + o];                                                     // after W1_l: H0 of layer l
    };
    FcW s0, s1;
    float a0 = 1.f, a1 = 1.f;
    if (g < nfc) { load_fc_col(s0, fcw(g)); a0 = aux(g); }
    if (g + 8 < nfc) { load_fc_col(s1, fcw(g + 8)); a1 = aux(g + 8); }
    if (tid < MD) tA[tid] = dwn[TD + tid] * MAP_SQ2 * dlrelu02(wrec[TD + tid]);
    __syncthreads();
    // stage 0: dG = tA . Wout; what the next stage reads (dG times the slope at the last layer's output) goes to tB
    if (g == 0) {
        const float v = fc_row1(tA, s0);
        dG[o] = v;
        tB[o] = v * dlrelu02(a0);
    }
    __syncthreads();
    for (int l = n_res - 1, jj = 1; l >= 0; --l, jj += 2) {
        if (g == (jj & 7)) {                         // tB through W1, times the slope at H0 -> tA
            const float v = (jj >> 3) ? fc_row1(tB, s1) : fc_row1(tB, s0);
            tA[o] = v * (MAP_SQ2 * dlrelu02((jj >> 3) ? a1 : a0));
        }
        __syncthreads();
        const int j2 = jj + 1;
        if (g == (j2 & 7)) {                         // tA through W0, plus the skip path (tB[o]: this lane's own element) -> dG
            const float v = ((j2 >> 3) ? fc_row1(tA, s1) : fc_row1(tA, s0)) + tB[o];
            dG[o] = v;
            tB[o] = v * dlrelu02((j2 >> 3) ? a1 : a0);                                   // the previous layer's slope (1 at l = 0: unused)
        }
        __syncthreads();
    }
    if (tid < 64) {
        const float zv = tid < MD ? zn[TD + tid] : 0.f;
        const float gv = tid < MD ? dG[tid] : 0.f;
        const float dot = wave_sum(zv * gv);
        if (tid < MD) {
            if (normalize_global) {
                const float fg = nrm[1];
                dzn[TD + tid] = fg * gv - zv * fg * fg * fg * dot / (float)MD;
            } else {
                dzn[TD + tid] = gv;
            }
        }
    }
}

// Local path, backward.
__device__ __forceinline__ void mapping_backward_local(MapShared& sh, MapBwdShared& bs, float* dzn, const float* dwn, const float* zn, const float* P,
                                       const float* sv, const float* wrec, const float* nrm, int k, int n_res) {
    const int tid = threadIdx.x;
    const int T = k - 1, TD = T * MD;
    const MapSave ms{T, n_res};
    const int64_t lstride = map_lstride(T);
    const float* Pl = map_local_params(P, n_res);
    const float* Pl_out = Pl + n_res * lstride;
    float *B = sh.H, *Cx = sh.X, *Dh = sh.Xin, *dQ = sh.Q, *dK = sh.K, *dV = sh.V, *Pr = sh.Pr;
    float *dX = bs.dX, *A = bs.A, *sQ = bs.sQ, *sK = bs.sK, *sV = bs.sV, *sP = bs.sP;
    {
        FcW fo;
        load_fc_col(fo, Pl_out);
        for (int i = tid; i < TD; i += MAP_BLOCK) A[i] = dwn[i] * MAP_SQ2 * dlrelu02(wrec[i]);
        __syncthreads();
        fc_apply<false>(A, fo, T, [&](int i, int, float v) { dX[i] = v; });                   // through the out layer
        __syncthreads();
    }
    for (int l = n_res - 1; l >= 0; --l) {
        const float* p = Pl + l * lstride;
        const float* Wq = p;            const float* bq = Wq + MAP_WSZ;
        const float* Wk = bq + TD;      const float* bk = Wk + MAP_WSZ;
        const float* Wv = bk + TD;      const float* bv = Wv + MAP_WSZ;
        const float* Wm = bv + MD;      const float* bm = Wm + MAP_WSZ;
        const float* W0 = bm + MD;      const float* b0 = W0 + MAP_WSZ;
        const float* W1 = b0 + MD;
        const float* svl = sv + ms.llayer(l);
        // everything this layer reads from memory, issued together: six weight columns, and the saved activations (the two used
        // element-wise stay in registers, the four used as matrices go to LDS)
        FcW f1, f0, fm, fq, fk, fv;
        load_fc_col(f1, W1); load_fc_col(f0, W0); load_fc_col(fm, Wm);
        load_fc_col(fq, Wq); load_fc_col(fk, Wk); load_fc_col(fv, Wv);
        float xo[MAP_PASSES], h0[MAP_PASSES];
#pragma unroll
        for (int j = 0; j < MAP_PASSES; ++j) {
            const int i = tid + MAP_BLOCK * j;
            const bool in = i < TD;
            xo[j] = in ? svl[4 * TD + i] : 0.f;
            h0[j] = in ? svl[3 * TD + i] : 0.f;
            if (in) { sQ[i] = svl[i]; sK[i] = svl[TD + i]; sV[i] = svl[2 * TD + i]; }
            if (i < T * T) sP[i] = svl[5 * TD + i];
        }
#pragma unroll
        for (int j = 0; j < MAP_PASSES; ++j) {
            const int i = tid + MAP_BLOCK * j;
            if (i < TD) A[i] = dX[i] * dlrelu02(xo[j]);                                  // d(F1 + Xin)
        }
        __syncthreads();
        fc_apply<false>(A, f1, T, [&](int i, int j, float v) { B[i] = v * (MAP_SQ2 * dlrelu02(h0[j])); });
        __syncthreads();
        fc_apply<false>(B, f0, T, [&](int i, int, float v) { Cx[i] = v; });                   // dXs (= dM, and the direct path into X)
        __syncthreads();
        fc_apply<false>(Cx, fm, T, [&](int i, int, float v) { Dh[i] = v; });                  // d(P V)
        __syncthreads();
        for (int idx = tid; idx < T * T; idx += MAP_BLOCK) {
            const int a = idx / T, b = idx % T;
            float acc = 0.f;
#pragma unroll
            for (int o = 0; o < MD; ++o) acc += Dh[a * MD + o] * sV[b * MD + o];
            Pr[a * MT_MAX + b] = acc;                // dP
        }
        for (int idx = tid; idx < TD; idx += MAP_BLOCK) {
            const int b = idx / MD, o = idx % MD;
            float acc = 0.f;
            for (int a = 0; a < T; ++a) acc += sP[a * T + b] * Dh[a * MD + o];
            dV[idx] = acc;
        }
        __syncthreads();
        if (tid < T) {
            float pdp = 0.f;
            for (int b = 0; b < T; ++b) pdp += sP[tid * T + b] * Pr[tid * MT_MAX + b];
            for (int b = 0; b < T; ++b) Pr[tid * MT_MAX + b] = sP[tid * T + b] * (Pr[tid * MT_MAX + b] - pdp);     // dScores
        }
        __syncthreads();
        for (int idx = tid; idx < TD; idx += MAP_BLOCK) {
            const int r = idx / MD, i = idx % MD;
            float aq = 0.f, ak = 0.f;
            for (int b = 0; b < T; ++b) {
                aq += Pr[r * MT_MAX + b] * sK[b * MD + i];
                ak += Pr[b * MT_MAX + r] * sQ[b * MD + i];
            }
            dQ[idx] = aq;
            dK[idx] = ak;
        }
        __syncthreads();
        // dX = ((Cx + A) + dQ . Wq) + dK . Wk) + dV . Wv, each output by its own thread
        float acc[MAP_PASSES];
        fc_apply<false>(dQ, fq, T, [&](int i, int j, float v) { acc[j] = (Cx[i] + A[i]) + v; });
        fc_apply<false>(dK, fk, T, [&](int, int j, float v) { acc[j] += v; });
        fc_apply<false>(dV, fv, T, [&](int i, int j, float v) { dX[i] = acc[j] + v; });
        __syncthreads();
    }
    // normalize backward: X0 = z * fl,  fl = rsqrt(mean z^2 + eps)  =>  dz = fl dX0 - z fl^3 <dX0, z> / (T D)
    {
        float part = 0.f;
        for (int i = tid; i < TD; i += MAP_BLOCK) part += dX[i] * zn[i];
        part = wave_sum(part);
        if ((tid & 63) == 0) sh.red[tid >> 6] = part;
        __syncthreads();
        const float dot = sh.red[0] + sh.red[1] + sh.red[2] + sh.red[3];
        const float fl = nrm[0];
        const float coef = fl * fl * fl * dot / (float)TD;
        for (int i = tid; i < TD; i += MAP_BLOCK) dzn[i] = fl * dX[i] - zn[i] * coef;
    }
}

// dz from dw.  RECOMPUTE: the forward is run again with SAVE into this sample's scratch slab first; otherwise the slab is the one
// mapping_save_kernel filled for the same z.  Two workgroups per sample, as in the forward.
template <bool RECOMPUTE>
__global__ __launch_bounds__(MAP_BLOCK) void mapping_backward_kernel(float* dz, const float* dw, const float* z, const float* P,
                                                                     float* scratch, int64_t slab, int k, int n_res,
                                                                     int normalize_global) {
    __shared__ MapShared sh;
    __shared__ MapBwdShared bs;
    const int n = blockIdx.x;
    const MapSave ms{k - 1, n_res};
    float* sv = scratch + (int64_t)n * slab;
    float* wrec = sv + ms.total();
    float* nrm = wrec + (int64_t)k * MD;
    const float* zn = z + (int64_t)n * k * MD;
    const float* dwn = dw + (int64_t)n * k * MD;
    float* dzn = dz + (int64_t)n * k * MD;
    if (blockIdx.y) {
        if (RECOMPUTE) {
            mapping_forward_global<true>(sh, wrec, nullptr, zn, P, k, n_res, normalize_global, sv, nrm);
            __syncthreads();                         // the slab was written by this workgroup: visible after the barrier
        }
        mapping_backward_global(sh, dzn, dwn, zn, P, sv, wrec, nrm, k, n_res, normalize_global);
    } else {
        if (RECOMPUTE) {
            mapping_forward_local<true>(sh, wrec, nullptr, zn, P, k, n_res, sv, nrm);
            __syncthreads();
        }
        mapping_backward_local(sh, bs, dzn, dwn, zn, P, sv, wrec, nrm, k, n_res);
    }
}

}  // namespace

extern "C" int mgf_demod_f32(float* d, const float* s, const float* wsq, int32_t n, int32_t cin, int32_t cout, mgf_stream_t stream) {
    MGF_REQUIRE(d && s && wsq && n >= 1 && cin >= 1 && cout >= 1, MGF_EINVAL, "demod: bad arguments");
    MGF_REQUIRE(n <= 65535, MGF_ETOOBIG, "demod: too many samples (%d)", n);
    hipLaunchKernelGGL(demod_kernel, dim3((unsigned)mgf_cdiv(cout, 4) < 1024u ? (unsigned)mgf_cdiv(cout, 4) : 1024u, n), dim3(256), 0, (hipStream_t)stream,
                       d, s, wsq, cin, cout);
    MGF_CHECK_LAUNCH("demod");
    return MGF_OK;
}

extern "C" int mgf_demod_bwd_f32(float* ds, const float* dd, const float* d, const float* s, const float* wsq, int32_t n, int32_t cin,
                                 int32_t cout, mgf_stream_t stream) {
    MGF_REQUIRE(ds && dd && d && s && wsq && n >= 1 && cin >= 1 && cout >= 1, MGF_EINVAL, "demod_bwd: bad arguments");
    MGF_REQUIRE(n <= 65535, MGF_ETOOBIG, "demod_bwd: too many samples (%d)", n);
    hipLaunchKernelGGL(demod_bwd_kernel, dim3((unsigned)mgf_cdiv(cin, 256), n), dim3(256), 0, (hipStream_t)stream, ds, dd, d, s, wsq, cin, cout);
    MGF_CHECK_LAUNCH("demod_bwd");
    return MGF_OK;
}

extern "C" int mgf_style_demod_multi(const mgf_style_job* jobs_dev, int32_t njobs, const float* ws, int64_t ws_stride_n,
                                     int32_t n, int32_t wdim, int32_t max_cin, mgf_stream_t stream) {
    MGF_REQUIRE(jobs_dev && ws && njobs >= 1 && n >= 1 && wdim >= 1, MGF_EINVAL, "style_demod_multi: bad arguments");
    MGF_REQUIRE(njobs <= 65535 && n <= 65535, MGF_ETOOBIG, "style_demod_multi: too many jobs/samples");
    // all samples in one workgroup when their styles fit in LDS: the wsq tables are read once instead of once per sample
    MGF_REQUIRE(max_cin >= 0 && max_cin <= 2048, MGF_EINVAL, "style_demod_multi: max_cin must be 0 (unknown) or the largest job cin (<= 2048)");
    if (n > 1 && n <= SD_NB && max_cin > 0 && (size_t)n * max_cin * sizeof(float) <= 128 * 1024) {
        const size_t lds = (size_t)n * max_cin * sizeof(float);
        if (lds > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute((const void*)style_demod_batched_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) { mgf_set_error("style_demod_multi: cannot raise dynamic LDS to %zu: %s", lds, hipGetErrorString(e)); return MGF_ELAUNCH; }
        }
        hipLaunchKernelGGL(style_demod_multi_kernel, dim3(1, njobs, n), dim3(256), 0, (hipStream_t)stream, jobs_dev, ws, ws_stride_n, wdim, 1);
        // 64 workgroups per job (8 rows of a 512-channel table each): 157 -> 54 us for the 19 layers at 25 samples against 16 -- each
        // wave walks its rows one after the other, so the launch wants many short walks.  MGF_SD_BLOCKS overrides (tuning)
        static const int sdb_env = [] { const char* e = mgf_knob("MGF_SD_BLOCKS"); return e ? atoi(e) : 0; }();
        hipLaunchKernelGGL(style_demod_batched_kernel, dim3(sdb_env > 0 ? sdb_env : 64, njobs), dim3(256), lds, (hipStream_t)stream, jobs_dev, ws, ws_stride_n, wdim, n);
    } else {
        static const int sdm_env = [] { const char* e = mgf_knob("MGF_SD_BLOCKS"); return e ? atoi(e) : 0; }();
        hipLaunchKernelGGL(style_demod_multi_kernel, dim3(sdm_env > 0 ? sdm_env : 16, njobs, n), dim3(256), 0, (hipStream_t)stream, jobs_dev, ws, ws_stride_n, wdim, 0);
    }
    MGF_CHECK_LAUNCH("style_demod_multi");
    return MGF_OK;
}

extern "C" int mgf_style_demod(const mgf_style_job* job, const float* ws, int64_t ws_stride_n, int32_t n, int32_t wdim,
                               mgf_stream_t stream) {
    MGF_REQUIRE(job && ws && n >= 1 && wdim >= 1, MGF_EINVAL, "style_demod: bad arguments");
    MGF_REQUIRE(job->cin >= 1 && job->cin <= 2048, MGF_EUNSUPPORTED, "style_demod: cin must be in 1..2048 (got %d)", job->cin);
    MGF_REQUIRE(job->aff_w && job->aff_b && job->s, MGF_EINVAL, "style_demod: null pointer in job");
    hipLaunchKernelGGL(style_demod_single_kernel, dim3(16, 1, n), dim3(256), 0, (hipStream_t)stream, *job, ws, ws_stride_n, wdim);
    MGF_CHECK_LAUNCH("style_demod");
    return MGF_OK;
}

extern "C" int mgf_attn_values_multi(const mgf_attn_job* jobs_dev, int32_t njobs, const float* ws, int64_t ws_stride_n,
                                     int64_t ws_stride_t, int32_t n, int32_t t, int32_t wdim, mgf_stream_t stream) {
    MGF_REQUIRE(jobs_dev && ws && njobs >= 1 && n >= 1 && t >= 1 && wdim >= 1, MGF_EINVAL, "attn_values_multi: bad arguments");
    hipLaunchKernelGGL(attn_values_multi_kernel, dim3(8, njobs, n), dim3(256), 0, (hipStream_t)stream, jobs_dev, ws, ws_stride_n,
                       ws_stride_t, wdim, t);
    MGF_CHECK_LAUNCH("attn_values_multi");
    return MGF_OK;
}

extern "C" int mgf_attn_values(const mgf_attn_job* job, const float* ws, int64_t ws_stride_n, int64_t ws_stride_t, int32_t n,
                               int32_t t, int32_t wdim, mgf_stream_t stream) {
    MGF_REQUIRE(job && ws && n >= 1 && t >= 1 && wdim >= 1, MGF_EINVAL, "attn_values: bad arguments");
    MGF_REQUIRE(job->wmv && job->bmv && job->vwb && job->c >= 1, MGF_EINVAL, "attn_values: bad job");
    hipLaunchKernelGGL(attn_values_single_kernel, dim3(8, 1, n), dim3(256), 0, (hipStream_t)stream, *job, ws, ws_stride_n, ws_stride_t,
                       wdim, t);
    MGF_CHECK_LAUNCH("attn_values");
    return MGF_OK;
}

extern "C" int64_t mgf_mapping_param_floats(int32_t k, int32_t dim, int32_t n_res_layers) {
    const int64_t W = (int64_t)dim * dim, T = k - 1;
    const int64_t glob = n_res_layers * (2 * W + 2 * dim) + W + dim;
    const int64_t loc = n_res_layers * (6 * W + 2 * T * dim + 4 * dim) + W + dim;
    return glob + loc;
}

extern "C" int64_t mgf_mapping_bwd_scratch_floats(int32_t k, int32_t dim, int32_t n_res_layers) {
    if (dim != MD || k < 2) return -1;
    const MapSave ms{k - 1, n_res_layers};
    return ms.total() + (int64_t)k * MD + 2;
}

extern "C" int mgf_mapping_backward(float* dz, const float* dw, const float* z, const float* params, float* scratch, int32_t n, int32_t k,
                                    int32_t dim, int32_t n_res_layers, int32_t normalize_global, mgf_stream_t stream) {
    MGF_REQUIRE(dz && dw && z && params && scratch, MGF_EINVAL, "mapping_backward: null pointer");
    MGF_REQUIRE(dim == MD, MGF_EUNSUPPORTED, "mapping_backward: latent width must be %d (got %d)", MD, dim);
    MGF_REQUIRE(k >= 2 && k - 1 <= MT_MAX, MGF_EUNSUPPORTED, "mapping_backward: k must be in 2..%d (got %d)", MT_MAX + 1, k);
    MGF_REQUIRE(n >= 1 && n_res_layers >= 0, MGF_EINVAL, "mapping_backward: bad sizes");
    MGF_REQUIRE(n_res_layers <= MAP_MAX_RES, MGF_EUNSUPPORTED, "mapping_backward: at most %d residual layers (got %d)", MAP_MAX_RES, n_res_layers);
    hipLaunchKernelGGL(mapping_backward_kernel<true>, dim3(n, 2), dim3(MAP_BLOCK), 0, (hipStream_t)stream, dz, dw, z, params, scratch,
                       mgf_mapping_bwd_scratch_floats(k, dim, n_res_layers), k, n_res_layers, normalize_global);
    MGF_CHECK_LAUNCH("mapping_backward");
    return MGF_OK;
}

namespace {
// N(0, 1) draws for the per-layer noise maps of noise_mode="random" (networks.py:1016-1017 draws them with torch.randn per layer and call):
// Philox4x32-10 keyed by `seed`, counter = *counter + lane index, four normals per counter value through two Box-Muller pairs.  The
// stream position lives in DEVICE memory and the launch advances it itself (the last workgroup to finish adds the number of counter
// values used), so a replayed hipGraph draws fresh numbers every time without any host involvement.
__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}

__global__ __launch_bounds__(256) void randn_kernel(float* out, int64_t n, uint64_t seed, unsigned long long* counter, unsigned* ticket) {
    const unsigned long long base = *counter;                       // (every workgroup reads it before the last one to finish advances it)
    const int64_t quads = (n + 3) / 4;
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < quads; q += (int64_t)gridDim.x * 256) {
        const unsigned long long ctr = base + (unsigned long long)q;
        uint32_t c[4] = {(uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u};
        uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
        for (int r = 0; r < 10; ++r) { philox_round(c, k0, k1); k0 += 0x9E3779B9u; k1 += 0xBB67AE85u; }
        float z[4];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const float u1 = ((float)c[2 * h] + 0.5f) * 2.3283064365386963e-10f;          // (0, 1]: + 0.5 keeps the logarithm finite
            const float u2 = ((float)c[2 * h + 1] + 0.5f) * 2.3283064365386963e-10f;
            const float r = sqrtf(-2.f * logf(u1 < 1.f ? u1 : 0.99999994f));
            float sn, cs;
            sincospif(2.f * u2, &sn, &cs);
            z[2 * h] = r * cs; z[2 * h + 1] = r * sn;
        }
        if (4 * q + 4 <= n && (((uintptr_t)out) & 15) == 0) *reinterpret_cast<float4*>(out + 4 * q) = make_float4(z[0], z[1], z[2], z[3]);
        else for (int e = 0; e < 4 && 4 * q + e < n; ++e) out[4 * q + e] = z[e];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned prev = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (prev == gridDim.x - 1) {                                  // the last workgroup: every other one has read `base` by now
            __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(counter, base + (unsigned long long)quads, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// rgbw[n, c, co] = w[c, co] * s[n, co]: the per-sample ToRGB weights the fused conv_last epilogue projects with (networks.py:1056-1063)
__global__ __launch_bounds__(256) void rgb_weights_kernel(float* out, const float* w, const float* s, int c, int co, int total) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int o = i % co, n = i / (c * co);
    out[i] = w[i % (c * co)] * s[n * co + o];
}
}  // namespace

extern "C" int mgf_randn_f32(float* out, int64_t n, uint64_t seed, void* state, mgf_stream_t stream) {
    MGF_REQUIRE(out && state && n >= 1, MGF_EINVAL, "randn: bad arguments");
    MGF_REQUIRE(((uintptr_t)state % 8) == 0, MGF_EINVAL, "randn: the 16-byte state {uint64 counter, uint32 ticket, pad} must be 8-byte aligned");
    const int grid = mgf_stream_grid((n + 3) / 4, 256, 4);
    hipLaunchKernelGGL(randn_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, out, n, seed, reinterpret_cast<unsigned long long*>(state),
                       reinterpret_cast<unsigned*>(reinterpret_cast<char*>(state) + 8));
    MGF_CHECK_LAUNCH("randn");
    return MGF_OK;
}

extern "C" int mgf_rgb_weights_f32(float* out, const float* w, const float* s, int32_t n, int32_t c, int32_t cout, mgf_stream_t stream) {
    MGF_REQUIRE(out && w && s && n >= 1 && c >= 1 && cout >= 1 && (int64_t)n * c * cout <= INT32_MAX, MGF_EINVAL, "rgb_weights: bad arguments");
    const int total = n * c * cout;
    hipLaunchKernelGGL(rgb_weights_kernel, dim3((unsigned)mgf_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, out, w, s, c, cout, total);
    MGF_CHECK_LAUNCH("rgb_weights");
    return MGF_OK;
}

extern "C" int mgf_mapping_forward_save(float* w, const float* z, const float* params, float* scratch, int32_t n, int32_t k, int32_t dim,
                                        int32_t n_res_layers, int32_t normalize_global, mgf_stream_t stream) {
    MGF_REQUIRE(w && z && params && scratch, MGF_EINVAL, "mapping_forward_save: null pointer");
    MGF_REQUIRE(dim == MD, MGF_EUNSUPPORTED, "mapping_forward_save: latent width must be %d (got %d)", MD, dim);
    MGF_REQUIRE(k >= 2 && k - 1 <= MT_MAX, MGF_EUNSUPPORTED, "mapping_forward_save: k must be in 2..%d (got %d)", MT_MAX + 1, k);
    MGF_REQUIRE(n >= 1 && n_res_layers >= 0, MGF_EINVAL, "mapping_forward_save: bad sizes");
    MGF_REQUIRE(n_res_layers <= MAP_MAX_RES, MGF_EUNSUPPORTED, "mapping_forward_save: at most %d residual layers (got %d)", MAP_MAX_RES, n_res_layers);
    hipLaunchKernelGGL(mapping_save_kernel, dim3(n, 2), dim3(MAP_BLOCK), 0, (hipStream_t)stream, w, z, params, scratch,
                       mgf_mapping_bwd_scratch_floats(k, dim, n_res_layers), k, n_res_layers, normalize_global);
    MGF_CHECK_LAUNCH("mapping_forward_save");
    return MGF_OK;
}

extern "C" int mgf_mapping_backward_saved(float* dz, const float* dw, const float* z, const float* params, float* scratch, int32_t n,
                                          int32_t k, int32_t dim, int32_t n_res_layers, int32_t normalize_global, mgf_stream_t stream) {
    MGF_REQUIRE(dz && dw && z && params && scratch, MGF_EINVAL, "mapping_backward_saved: null pointer");
    MGF_REQUIRE(dim == MD, MGF_EUNSUPPORTED, "mapping_backward_saved: latent width must be %d (got %d)", MD, dim);
    MGF_REQUIRE(k >= 2 && k - 1 <= MT_MAX, MGF_EUNSUPPORTED, "mapping_backward_saved: k must be in 2..%d (got %d)", MT_MAX + 1, k);
    MGF_REQUIRE(n >= 1 && n_res_layers >= 0, MGF_EINVAL, "mapping_backward_saved: bad sizes");
    MGF_REQUIRE(n_res_layers <= MAP_MAX_RES, MGF_EUNSUPPORTED, "mapping_backward_saved: at most %d residual layers (got %d)", MAP_MAX_RES, n_res_layers);
    hipLaunchKernelGGL(mapping_backward_kernel<false>, dim3(n, 2), dim3(MAP_BLOCK), 0, (hipStream_t)stream, dz, dw, z, params, scratch,
                       mgf_mapping_bwd_scratch_floats(k, dim, n_res_layers), k, n_res_layers, normalize_global);
    MGF_CHECK_LAUNCH("mapping_backward_saved");
    return MGF_OK;
}

extern "C" int mgf_mapping_forward(float* w, const float* z, const float* params, int32_t n, int32_t k, int32_t dim,
                                   int32_t n_res_layers, int32_t normalize_global, mgf_stream_t stream) {
    MGF_REQUIRE(w && z && params, MGF_EINVAL, "mapping_forward: null pointer");
    MGF_REQUIRE(dim == MD, MGF_EUNSUPPORTED, "mapping_forward: latent width must be %d (got %d)", MD, dim);
    MGF_REQUIRE(k >= 2 && k - 1 <= MT_MAX, MGF_EUNSUPPORTED, "mapping_forward: k must be in 2..%d (got %d)", MT_MAX + 1, k);
    MGF_REQUIRE(n >= 1 && n_res_layers >= 0, MGF_EINVAL, "mapping_forward: bad sizes");
    MGF_REQUIRE(n_res_layers <= MAP_MAX_RES, MGF_EUNSUPPORTED, "mapping_forward: at most %d residual layers (got %d)", MAP_MAX_RES, n_res_layers);
    hipLaunchKernelGGL(mapping_kernel, dim3(n, 2), dim3(MAP_BLOCK), 0, (hipStream_t)stream, w, z, params, k, n_res_layers, normalize_global);
    MGF_CHECK_LAUNCH("mapping_forward");
    return MGF_OK;
}
