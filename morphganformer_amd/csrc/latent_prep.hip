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

// out[r][o] = sum_i in[r][i] * W[o][i] + (row_bias ? brow[r*D + o] : b[o]);  rows x 32 outputs spread over the block
__device__ void fc_rows(float* out, const float* in, const float* W, const float* b, bool row_bias, int rows) {
    for (int idx = threadIdx.x; idx < rows * MD; idx += blockDim.x) {
        const int r = idx / MD, o = idx % MD;
        const float* wr = W + o * MD;
        const float* xr = in + r * MD;
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < MD; ++i) acc += xr[i] * wr[i];
        out[idx] = acc + (row_bias ? b[r * MD + o] : b[o]);
    }
}

// out[r][i] = sum_o in[r][o] * W[o][i]   (the transposed product of the backward pass)
__device__ void fc_rows_t(float* out, const float* in, const float* W, int rows, bool accumulate) {
    for (int idx = threadIdx.x; idx < rows * MD; idx += blockDim.x) {
        const int r = idx / MD, i = idx % MD;
        const float* xr = in + r * MD;
        float acc = 0.f;
#pragma unroll
        for (int o = 0; o < MD; ++o) acc += xr[o] * W[o * MD + i];
        out[idx] = accumulate ? out[idx] + acc : acc;
    }
}

struct MapShared {
    float X[MT_MAX * MD], Xin[MT_MAX * MD], Q[MT_MAX * MD], K[MT_MAX * MD], V[MT_MAX * MD], H[MT_MAX * MD];
    float Pr[MT_MAX * MT_MAX];
    float G[MD], Gin[MD], GH[MD];
    float red[4];
};

// Per-sample scratch of the backward pass (floats): what the recomputed forward leaves behind.
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

// Forward of one sample; SAVE additionally stores the per-layer activations the backward pass needs into `sv`.
template <bool SAVE>
__device__ void mapping_forward_body(MapShared& sh, float* w_n, const float* zn, const float* P, int k, int n_res, int normalize_global,
                                     float* sv, float* norm_out) {
    float *X = sh.X, *Xin = sh.Xin, *Q = sh.Q, *K = sh.K, *V = sh.V, *H = sh.H, *Pr = sh.Pr, *G = sh.G, *Gin = sh.Gin, *GH = sh.GH;
    float* red = sh.red;
    const int tid = threadIdx.x;
    const int T = k - 1;
    const MapSave ms{T, n_res};
    // ---- normalize (networks.py:30-37): joint second moment over the T x D local block; global row separately ----
    float part = 0.f;
    for (int i = tid; i < T * MD; i += blockDim.x) { float v = zn[i]; part += v * v; }
    part = wave_sum(part);
    if ((tid & 63) == 0) red[tid >> 6] = part;
    __syncthreads();
    const float fl = rsqrtf((red[0] + red[1] + red[2] + red[3]) / (float)(T * MD) + 1e-8f);
    for (int i = tid; i < T * MD; i += blockDim.x) X[i] = zn[i] * fl;
    __syncthreads();
    if (tid < 64) {
        float v = tid < MD ? zn[T * MD + tid] : 0.f;
        float ss = wave_sum(v * v);
        float fg = normalize_global ? rsqrtf(ss / (float)MD + 1e-8f) : 1.f;
        if (tid < MD) G[tid] = v * fg;
        if (SAVE && tid == 0) { norm_out[0] = fl; norm_out[1] = fg; }
    }
    __syncthreads();

    const int WSZ = MD * MD;
    const float* p = P;
    // ---- global MLP ----
    for (int l = 0; l < n_res; ++l) {
        const float *W0 = p, *b0 = p + WSZ, *W1 = b0 + MD, *b1 = W1 + WSZ;
        p = b1 + MD;
        if (tid < MD) Gin[tid] = G[tid];
        __syncthreads();
        fc_rows(GH, G, W0, b0, false, 1);
        __syncthreads();
        if (tid < MD) {
            GH[tid] = lrelu02(GH[tid]) * 1.41421356237309515f;
            if (SAVE) sv[ms.glayer(l) + tid] = GH[tid];
        }
        __syncthreads();
        fc_rows(G, GH, W1, b1, false, 1);
        __syncthreads();
        if (tid < MD) {
            G[tid] = lrelu02(G[tid] + Gin[tid]);
            if (SAVE) sv[ms.glayer(l) + MD + tid] = G[tid];
        }
        __syncthreads();
    }
    {
        const float *Wo = p, *bo = p + WSZ;
        p = bo + MD;
        fc_rows(GH, G, Wo, bo, false, 1);
        __syncthreads();
        if (tid < MD) w_n[T * MD + tid] = lrelu02(GH[tid]) * 1.41421356237309515f;
    }
    // ---- local MLP with latent self-attention ----
    for (int l = 0; l < n_res; ++l) {
        const float* Wq = p;            const float* bq = Wq + WSZ;
        const float* Wk = bq + T * MD;  const float* bk = Wk + WSZ;
        const float* Wv = bk + T * MD;  const float* bv = Wv + WSZ;
        const float* Wm = bv + MD;      const float* bm = Wm + WSZ;
        const float* W0 = bm + MD;      const float* b0 = W0 + WSZ;
        const float* W1 = b0 + MD;      const float* b1 = W1 + WSZ;
        p = b1 + MD;
        float* svl = SAVE ? sv + ms.llayer(l) : nullptr;
        const int TD = T * MD;
        for (int i = tid; i < T * MD; i += blockDim.x) Xin[i] = X[i];
        fc_rows(Q, X, Wq, bq, true, T);          // 1/sqrt(D) and the positional term are folded into Wq / bq
        fc_rows(K, X, Wk, bk, true, T);
        fc_rows(V, X, Wv, bv, false, T);
        __syncthreads();
        if (SAVE)
            for (int i = tid; i < TD; i += blockDim.x) { svl[i] = Q[i]; svl[TD + i] = K[i]; svl[2 * TD + i] = V[i]; }
        for (int idx = tid; idx < T * T; idx += blockDim.x) {
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
        for (int idx = tid; idx < T * MD; idx += blockDim.x) {
            const int a = idx / MD, o = idx % MD;
            float acc = 0.f;
            for (int b = 0; b < T; ++b) acc += Pr[a * MT_MAX + b] * V[b * MD + o];
            H[idx] = acc;
        }
        __syncthreads();
        fc_rows(Q, H, Wm, bm, false, T);
        __syncthreads();
        for (int i = tid; i < T * MD; i += blockDim.x) X[i] += Q[i];        // integration 'add'
        __syncthreads();
        fc_rows(H, X, W0, b0, false, T);
        __syncthreads();
        for (int i = tid; i < T * MD; i += blockDim.x) {
            H[i] = lrelu02(H[i]) * 1.41421356237309515f;
            if (SAVE) svl[3 * TD + i] = H[i];
        }
        __syncthreads();
        fc_rows(Q, H, W1, b1, false, T);
        __syncthreads();
        for (int i = tid; i < T * MD; i += blockDim.x) {
            X[i] = lrelu02(Q[i] + Xin[i]);
            if (SAVE) svl[4 * TD + i] = X[i];
        }
        __syncthreads();
    }
    {
        const float *Wo = p, *bo = p + WSZ;
        fc_rows(H, X, Wo, bo, false, T);
        __syncthreads();
        for (int i = tid; i < T * MD; i += blockDim.x) w_n[i] = lrelu02(H[i]) * 1.41421356237309515f;
    }
}

__global__ __launch_bounds__(256) void mapping_kernel(float* w, const float* z, const float* P, int k, int n_res, int normalize_global) {
    __shared__ MapShared sh;
    const int n = blockIdx.x;
    mapping_forward_body<false>(sh, w + (int64_t)n * k * MD, z + (int64_t)n * k * MD, P, k, n_res, normalize_global, nullptr, nullptr);
}

__device__ __forceinline__ float dlrelu02(float y) { return y > 0.f ? 1.f : 0.2f; }

// dz from dw: the forward is recomputed with SAVE into this sample's scratch slab, then the layers are walked in reverse.
// Scratch slab: [MapSave::total() activations][k*D recomputed w][2 normalisation factors].
__global__ __launch_bounds__(256) void mapping_backward_kernel(float* dz, const float* dw, const float* z, const float* P, float* scratch,
                                                               int64_t slab, int k, int n_res, int normalize_global) {
    __shared__ MapShared sh;
    __shared__ float dX[MT_MAX * MD], A[MT_MAX * MD];
    const int n = blockIdx.x, tid = threadIdx.x;
    const int T = k - 1, TD = T * MD, WSZ = MD * MD;
    const MapSave ms{T, n_res};
    float* sv = scratch + (int64_t)n * slab;
    float* wrec = sv + ms.total();
    float* nrm = wrec + (int64_t)k * MD;
    const float* zn = z + (int64_t)n * k * MD;
    const float* dwn = dw + (int64_t)n * k * MD;
    float* dzn = dz + (int64_t)n * k * MD;
    mapping_forward_body<true>(sh, wrec, zn, P, k, n_res, normalize_global, sv, nrm);
    __syncthreads();                                 // the slab was written by this workgroup: visible after the barrier
    const float SQ2 = 1.41421356237309515f;
    const int64_t gstride = 2 * WSZ + 2 * MD;
    const float* Pg_out = P + n_res * gstride;
    const float* Pl = Pg_out + WSZ + MD;
    const int64_t lstride = 6 * (int64_t)WSZ + 2 * (int64_t)TD + 4 * MD;
    const float* Pl_out = Pl + n_res * lstride;
    float *B = sh.H, *Cx = sh.X, *Dh = sh.Xin, *dQ = sh.Q, *dK = sh.K, *dV = sh.V, *Pr = sh.Pr;

    // ---- local path ----
    for (int i = tid; i < TD; i += blockDim.x) A[i] = dwn[i] * SQ2 * dlrelu02(wrec[i]);
    __syncthreads();
    fc_rows_t(dX, A, Pl_out, T, false);              // through the out layer
    __syncthreads();
    for (int l = n_res - 1; l >= 0; --l) {
        const float* p = Pl + l * lstride;
        const float* Wq = p;            const float* bq = Wq + WSZ;
        const float* Wk = bq + TD;      const float* bk = Wk + WSZ;
        const float* Wv = bk + TD;      const float* bv = Wv + WSZ;
        const float* Wm = bv + MD;      const float* bm = Wm + WSZ;
        const float* W0 = bm + MD;      const float* b0 = W0 + WSZ;
        const float* W1 = b0 + MD;
        const float* svl = sv + ms.llayer(l);
        const float *sQ = svl, *sK = svl + TD, *sV = svl + 2 * TD, *sH0 = svl + 3 * TD, *sXo = svl + 4 * TD, *sP = svl + 5 * TD;
        for (int i = tid; i < TD; i += blockDim.x) A[i] = dX[i] * dlrelu02(sXo[i]);           // d(F1 + Xin)
        __syncthreads();
        fc_rows_t(B, A, W1, T, false);
        __syncthreads();
        for (int i = tid; i < TD; i += blockDim.x) B[i] *= SQ2 * dlrelu02(sH0[i]);
        __syncthreads();
        fc_rows_t(Cx, B, W0, T, false);              // dXs (= dM, and the direct path into X)
        __syncthreads();
        fc_rows_t(Dh, Cx, Wm, T, false);             // d(P V)
        __syncthreads();
        for (int idx = tid; idx < T * T; idx += blockDim.x) {
            const int a = idx / T, b = idx % T;
            float acc = 0.f;
#pragma unroll
            for (int o = 0; o < MD; ++o) acc += Dh[a * MD + o] * sV[b * MD + o];
            Pr[a * MT_MAX + b] = acc;                // dP
        }
        for (int idx = tid; idx < TD; idx += blockDim.x) {
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
        for (int idx = tid; idx < TD; idx += blockDim.x) {
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
        for (int i = tid; i < TD; i += blockDim.x) dX[i] = Cx[i] + A[i];
        __syncthreads();
        fc_rows_t(dX, dQ, Wq, T, true);
        __syncthreads();
        fc_rows_t(dX, dK, Wk, T, true);
        __syncthreads();
        fc_rows_t(dX, dV, Wv, T, true);
        __syncthreads();
    }
    // normalize backward: X0 = z * fl,  fl = rsqrt(mean z^2 + eps)  =>  dz = fl dX0 - z fl^3 <dX0, z> / (T D)
    {
        float part = 0.f;
        for (int i = tid; i < TD; i += blockDim.x) part += dX[i] * zn[i];
        part = wave_sum(part);
        __syncthreads();
        if ((tid & 63) == 0) sh.red[tid >> 6] = part;
        __syncthreads();
        const float dot = sh.red[0] + sh.red[1] + sh.red[2] + sh.red[3];
        const float fl = nrm[0];
        const float coef = fl * fl * fl * dot / (float)TD;
        for (int i = tid; i < TD; i += blockDim.x) dzn[i] = fl * dX[i] - zn[i] * coef;
    }
    __syncthreads();
    // ---- global path (one row) ----
    float* dG = sh.G;
    float* tA = sh.Gin;
    float* tB = sh.GH;
    if (tid < MD) tA[tid] = dwn[TD + tid] * SQ2 * dlrelu02(wrec[TD + tid]);
    __syncthreads();
    fc_rows_t(dG, tA, Pg_out, 1, false);
    __syncthreads();
    for (int l = n_res - 1; l >= 0; --l) {
        const float* p = P + l * gstride;
        const float *W0 = p, *b0 = p + WSZ, *W1 = b0 + MD;
        const float* svl = sv + ms.glayer(l);
        if (tid < MD) tA[tid] = dG[tid] * dlrelu02(svl[MD + tid]);
        __syncthreads();
        fc_rows_t(tB, tA, W1, 1, false);
        __syncthreads();
        if (tid < MD) tB[tid] *= SQ2 * dlrelu02(svl[tid]);
        __syncthreads();
        fc_rows_t(dG, tB, W0, 1, false);
        __syncthreads();
        if (tid < MD) dG[tid] += tA[tid];
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

}  // namespace

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
        hipLaunchKernelGGL(style_demod_batched_kernel, dim3(16, njobs), dim3(256), lds, (hipStream_t)stream, jobs_dev, ws, ws_stride_n, wdim, n);
    } else {
        hipLaunchKernelGGL(style_demod_multi_kernel, dim3(16, njobs, n), dim3(256), 0, (hipStream_t)stream, jobs_dev, ws, ws_stride_n, wdim, 0);
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
    hipLaunchKernelGGL(mapping_backward_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, dz, dw, z, params, scratch,
                       mgf_mapping_bwd_scratch_floats(k, dim, n_res_layers), k, n_res_layers, normalize_global);
    MGF_CHECK_LAUNCH("mapping_backward");
    return MGF_OK;
}

extern "C" int mgf_mapping_forward(float* w, const float* z, const float* params, int32_t n, int32_t k, int32_t dim,
                                   int32_t n_res_layers, int32_t normalize_global, mgf_stream_t stream) {
    MGF_REQUIRE(w && z && params, MGF_EINVAL, "mapping_forward: null pointer");
    MGF_REQUIRE(dim == MD, MGF_EUNSUPPORTED, "mapping_forward: latent width must be %d (got %d)", MD, dim);
    MGF_REQUIRE(k >= 2 && k - 1 <= MT_MAX, MGF_EUNSUPPORTED, "mapping_forward: k must be in 2..%d (got %d)", MT_MAX + 1, k);
    MGF_REQUIRE(n >= 1 && n_res_layers >= 0, MGF_EINVAL, "mapping_forward: bad sizes");
    hipLaunchKernelGGL(mapping_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, w, z, params, k, n_res_layers, normalize_global);
    MGF_CHECK_LAUNCH("mapping_forward");
    return MGF_OK;
}
