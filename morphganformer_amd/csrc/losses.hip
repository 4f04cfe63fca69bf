// Losses and bookkeeping of the projection loop, all device-side so an iteration never synchronises with the host.
// Contract: include/mgf.h.  Every op is BATCHED over n independent candidates (the literal loop's steps do not depend on each
// other, so several of them are evaluated per generator forward); reductions are deterministic: fixed grid, per-block
// partials in `scratch`, one finishing block per sample sums them in index order (no float atomics -> bit-reproducible).
#include "mgf_common.h"

namespace {

constexpr int RED_BLOCKS = 16384;     // scratch floats per sample

__device__ __forceinline__ float block_sum_256(float v, float* sm) {
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = sm[0] + sm[1] + sm[2] + sm[3];
    __syncthreads();
    return r;
}

// out[s] (+)= scale * sum(scratch[s][0 .. nparts))   -- one workgroup per sample
__global__ __launch_bounds__(256) void finish_kernel(float* out, const float* scratch, int nparts, float scale, int accumulate) {
    __shared__ float sm[4];
    const float* sc = scratch + (int64_t)blockIdx.x * RED_BLOCKS;
    float v = 0.f;
    for (int i = threadIdx.x; i < nparts; i += 256) v += sc[i];
    v = block_sum_256(v, sm);
    if (threadIdx.x == 0) out[blockIdx.x] = (accumulate ? out[blockIdx.x] : 0.f) + v * scale;
}

// the same finish for up to 8 partial sets at once (the taps of one LPIPS evaluation), added to out[s] in set order with the arithmetic of
// that many finish_kernel launches: out = (accumulate ? out : 0) + v_0 * scale_0; out += v_1 * scale_1; ...
struct FinishSets { int nparts[8]; float scale[8]; };
__global__ __launch_bounds__(256) void finish_multi_kernel(float* out, const float* scratch, int64_t set_stride, int nsets, FinishSets fs, int accumulate) {
    __shared__ float sm[4];
    float acc = accumulate ? out[blockIdx.x] : 0.f;
    for (int t = 0; t < nsets; ++t) {
        const float* sc = scratch + (int64_t)t * set_stride + (int64_t)blockIdx.x * RED_BLOCKS;
        float v = 0.f;
        for (int i = threadIdx.x; i < fs.nparts[t]; i += 256) v += sc[i];
        v = block_sum_256(v, sm);
        acc = acc + v * fs.scale[t];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[blockIdx.x] = acc;
}

// grid = (blocks, n)
__global__ __launch_bounds__(256) void mse_partial_kernel(float* scratch, const float* a, const float* b, int64_t numel, int64_t b_stride) {
    __shared__ float sm[4];
    const float* as = a + (int64_t)blockIdx.y * numel;
    const float* bs = b + (int64_t)blockIdx.y * b_stride;
    float acc = 0.f;
    const int64_t nvec = numel / 4;
    const float4* a4 = reinterpret_cast<const float4*>(as);
    const float4* b4 = reinterpret_cast<const float4*>(bs);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
        float4 u = a4[i], v = b4[i];
        float d0 = u.x - v.x, d1 = u.y - v.y, d2 = u.z - v.z, d3 = u.w - v.w;
        acc += d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
    }
    for (int64_t i = nvec * 4 + (int64_t)blockIdx.x * 256 + threadIdx.x; i < numel; i += (int64_t)gridDim.x * 256) {
        float d0 = as[i] - bs[i];
        acc += d0 * d0;
    }
    acc = block_sum_256(acc, sm);
    if (threadIdx.x == 0) scratch[(int64_t)blockIdx.y * RED_BLOCKS + blockIdx.x] = acc;
}

// grid = (pixel blocks, n).  A workgroup owns PXB consecutive pixels of one sample (lanes = consecutive pixels, so a wave reads
// whole 64/256-byte segments of a channel plane); its G = 256 / PXB lane groups take the channels grp, grp + G, grp + 2G ...
// ONE sweep over HBM: each thread keeps its <= CPT channel values in registers, the squared norms meet in LDS in a fixed order
// (bit-reproducible), then the registers are either written back unit-normalised (UNIT_OUT: the reference image's taps, once
// per target) or compared with the stored unit-normalised reference taps and reduced to one partial per workgroup.
template <int PXB, int CPT, bool UNIT_OUT, bool STATS = false>
__global__ __launch_bounds__(256) void lpips_layer_kernel(float* scratch, float* unit_out, const float* f0, const float* f1u,
                                                           const float* lin, int c, int64_t hw, int64_t f1_stride, int nsamp, int nblk,
                                                           int xcd_per, float* stats) {
    constexpr int G = 256 / PXB;
    __shared__ float red[G][PXB];
    __shared__ float redb[STATS ? G : 1][PXB], redc[STATS ? G : 1][PXB];      // STATS: the two other per-pixel sums the gradient needs
    __shared__ float sm[4];
    const int px = threadIdx.x % PXB, grp = threadIdx.x / PXB;
    // work order: XCD b % 8 walks a contiguous item range with the SAMPLE as the fastest index -- the n candidates of a pixel block read
    // the same block of the target's stored taps and now share it through one L2 (it came from the Infinity Cache once per candidate:
    // as many bytes again as the candidates' own taps)
    int item = (blockIdx.x & 7) * xcd_per + (blockIdx.x >> 3);
    if (item >= nsamp * nblk) return;
    const int nn = item % nsamp;
    const int blk = item / nsamp;
    const int64_t i = (int64_t)blk * PXB + px;
    const bool valid = i < hw;
    const int64_t pp = valid ? i : hw - 1;
    // Buffer addressing: a thread's channels are G planes apart, so the channel part of every address is a wave-uniform scalar offset
    // (j * G * hw) on ONE per-lane offset (grp * hw + pixel); a channel past c gets an offset beyond num_records and reads 0.  With flat
    // addresses each of the 2 * CPT loads carried a 64-bit multiply-add (quarter-rate integer multiplies) and its own exec-mask branch.
    const unsigned plane_bytes = 4u * (unsigned)hw;                                   // host: c * hw * 4 < 2^32
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)(f0 + (int64_t)nn * c * hw), 0, (int)((unsigned)c * plane_bytes), 0x00020000);
    const unsigned vo = (unsigned)grp * plane_bytes + 4u * (unsigned)pp;
    float u[CPT], v[UNIT_OUT ? 1 : CPT];
    float na = 0.f;
#pragma unroll
    for (int j = 0; j < CPT; ++j)
        u[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ra, grp + j * G < c ? vo : 0xFFFFFFF0u, (int)((unsigned)(j * G) * plane_bytes), 0));
    if (!UNIT_OUT) {                    // the reference taps are requested in the same burst: one memory round trip per workgroup
        const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)(f1u + (int64_t)nn * f1_stride), 0, (int)((unsigned)c * plane_bytes), 0x00020000);
#pragma unroll
        for (int j = 0; j < CPT; ++j)
            v[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rb, grp + j * G < c ? vo : 0xFFFFFFF0u, (int)((unsigned)(j * G) * plane_bytes), 0));
    }
#pragma unroll
    for (int j = 0; j < CPT; ++j) na += u[j] * u[j];
    red[grp][px] = na;
    __syncthreads();
    na = 0.f;
#pragma unroll
    for (int g = 0; g < G; ++g) na += red[g][px];
    const float ia = 1.f / (sqrtf(na) + 1e-10f);
    if (UNIT_OUT) {
        const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void*)(unit_out + (int64_t)nn * c * hw), 0, (int)((unsigned)c * plane_bytes), 0x00020000);
#pragma unroll
        for (int j = 0; j < CPT; ++j)
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, u[j] * ia), ro, (valid && grp + j * G < c) ? vo : 0xFFFFFFF0u,
                                                  (int)((unsigned)(j * G) * plane_bytes), 0);
        return;
    }
    float d = 0.f, nb = 0.f, nc = 0.f;
    const __amdgpu_buffer_rsrc_t rl = __builtin_amdgcn_make_buffer_rsrc((void*)lin, 0, 4 * c, 0x00020000);
#pragma unroll
    for (int j = 0; j < CPT; ++j) {
        // (no `k < c` branch: a channel past c arrived as u = v = 0 and its `lin` read is out of range, i.e. 0 as well)
        // separate statements: the product is rounded before the subtraction (build uses -ffp-contract=on) exactly like the
        // stored reference taps (u * ia above), so identical images give exactly zero
        const float ua = u[j] * ia;
        const float e = ua - v[UNIT_OUT ? 0 : j];
        const float l = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rl, 4u * (unsigned)grp, 4 * j * G, 0));
        d += l * e * e;
        if (STATS) { nb += l * u[j] * u[j]; nc += l * v[UNIT_OUT ? 0 : j] * u[j]; }
    }
    // STATS (gradient mode): A = sum f0^2, B = sum lin f0^2, C = sum lin u1 f0 per pixel -> [n][3][hw]; the backward
    // (mgf_lpips_layer_bwd_relu_stats_f32) then skips its own sweep over both maps for them.  Formed beside the distance (same operands, two
    // more FMAs; as a loop of their own they cost the forward 75 % more time than they saved the backward)
    if (STATS) {
        redb[grp][px] = nb;
        redc[grp][px] = nc;
        __syncthreads();
        if (grp == 0 && valid) {
            float sb = 0.f, sc = 0.f;
#pragma unroll
            for (int g = 0; g < G; ++g) { sb += redb[g][px]; sc += redc[g][px]; }
            float* so = stats + (int64_t)nn * 3 * hw + i;
            so[0] = na; so[hw] = sb; so[2 * hw] = sc;
        }
    }
    const float acc = block_sum_256(valid ? d : 0.f, sm);
    if (threadIdx.x == 0) scratch[(int64_t)nn * RED_BLOCKS + blk] = acc;
}

template <bool UNIT_OUT>
int launch_lpips_layer(float* scratch, float* unit_out, const float* f0, const float* f1u, const float* lin, int n, int c, int64_t hw,
                       int64_t f1_stride, hipStream_t st, int* grid_out, float* stats = nullptr) {
    // 64 pixels per workgroup (256-byte segments) whenever that still yields >= 4 workgroups per CU, 16 for the small deep taps
    static const int pxb_env = [] { const char* e = mgf_knob("MGF_LPIPS_PXB"); return e ? atoi(e) : 0; }();      // tuning hook: 16 | 32 | 64
    // (re-tuned on buffer addressing, 32 x {128 @ 255^2, 256 @ 127^2, 384 @ 63^2, 512 @ 63^2}, us for 64 / 32 / 16-pixel blocks:
    // 264 / 306 / 386, 145 / 155 / 195, 100 / 60 / 58, 126 / 70 / 74 -- tools/lpips_layer_micro.py; with flat addressing 64 values per thread
    // had been the slow case and the 256-channel tap ran on 32-pixel blocks)
    const int pxb = pxb_env ? pxb_env
                  : (c <= 256 && (hw >= 65536 || (int64_t)n * mgf_cdiv(hw, 64) >= 1024)) ? 64
                  : (c > 384 && (int64_t)n * mgf_cdiv(hw, 32) >= 1024) ? 32 : 16;
    const int64_t grid64 = mgf_cdiv(hw, pxb);
    MGF_REQUIRE(grid64 <= RED_BLOCKS, MGF_ETOOBIG, "lpips_layer: %lld pixels per sample need %lld scratch floats (have %d per sample)",
                (long long)hw, (long long)grid64, RED_BLOCKS);
    MGF_REQUIRE(c <= 512, MGF_EUNSUPPORTED, "lpips_layer: at most 512 channels per tap (got %d)", c);
    MGF_REQUIRE((int64_t)c * hw < (1LL << 30), MGF_ETOOBIG, "lpips_layer: one sample's tap must stay below 4 GiB");
    MGF_REQUIRE(grid64 * n <= INT32_MAX - 8, MGF_ETOOBIG, "lpips_layer: too many workgroups");
    const int xcd_per = (int)mgf_cdiv(grid64 * n, 8);
    const dim3 grid((unsigned)(xcd_per * 8));
    *grid_out = (int)grid64;
#define MGF_LPIPS_LAUNCH(PXB, CPT)                                                                                                            \
    do {                                                                                                                                      \
        if (!UNIT_OUT && stats)                                                                                                               \
            hipLaunchKernelGGL((lpips_layer_kernel<PXB, CPT, false, true>), grid, dim3(256), 0, st, scratch, unit_out, f0, f1u, lin, c, hw,    \
                               f1_stride, n, (int)grid64, xcd_per, stats);                                                                    \
        else                                                                                                                                  \
            hipLaunchKernelGGL((lpips_layer_kernel<PXB, CPT, UNIT_OUT>), grid, dim3(256), 0, st, scratch, unit_out, f0, f1u, lin, c, hw,      \
                               f1_stride, n, (int)grid64, xcd_per, nullptr);                                                                  \
    } while (0)
    if (pxb == 64) {
        if (c <= 128) MGF_LPIPS_LAUNCH(64, 32); else if (c <= 256) MGF_LPIPS_LAUNCH(64, 64); else MGF_LPIPS_LAUNCH(64, 128);
    } else if (pxb == 32) {
        if (c <= 128) MGF_LPIPS_LAUNCH(32, 16); else if (c <= 256) MGF_LPIPS_LAUNCH(32, 32); else MGF_LPIPS_LAUNCH(32, 64);
    } else {
        if (c <= 128) MGF_LPIPS_LAUNCH(16, 8); else if (c <= 256) MGF_LPIPS_LAUNCH(16, 16); else MGF_LPIPS_LAUNCH(16, 32);
    }
#undef MGF_LPIPS_LAUNCH
    return MGF_OK;
}

// one workgroup per candidate: out[j] = WingLoss(pred row (*pred_step + j), target) in float64
__global__ __launch_bounds__(256) void wing_kernel(double* out, const double* pred, const double* target, int64_t numel, double omega,
                                                   double epsilon, const int32_t* pred_step, int max_row) {
    __shared__ double sm[4];
    int row = (pred_step ? *pred_step : 0) + (int)blockIdx.x;
    if (max_row >= 0 && row > max_row) row = max_row;
    pred += (int64_t)row * numel;
    const double cc = omega - omega * log(1.0 + omega / epsilon);
    double acc = 0.0;
    for (int64_t i = threadIdx.x; i < numel; i += 256) {
        double dlt = fabs(target[i] - pred[i]);
        acc += dlt < omega ? omega * log(1.0 + dlt / epsilon) : dlt - cc;
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = (sm[0] + sm[1] + sm[2] + sm[3]) / (double)numel;
}

// AdaptiveWingLoss (adaptive_wing_loss.py:20-39): the exponent alpha - y depends on the TARGET value y; mean over all elements
__global__ __launch_bounds__(256) void awing_kernel(double* out, const double* pred, const double* target, int64_t numel, double omega,
                                                    double theta, double epsilon, double alpha, const int32_t* pred_step, int max_row) {
    __shared__ double sm[4];
    int row = (pred_step ? *pred_step : 0) + (int)blockIdx.x;
    if (max_row >= 0 && row > max_row) row = max_row;
    pred += (int64_t)row * numel;
    double acc = 0.0;
    for (int64_t i = threadIdx.x; i < numel; i += 256) {
        const double y = target[i];
        const double dlt = fabs(y - pred[i]);
        const double pw = alpha - y;
        if (dlt < theta) {
            acc += omega * log(1.0 + pow(dlt / omega, pw));
        } else {
            const double tp = pow(theta / epsilon, pw);
            const double A = omega * (1.0 / (1.0 + tp)) * pw * pow(theta / epsilon, pw - 1.0) * (1.0 / epsilon);
            const double C = theta * A - omega * log(1.0 + tp);
            acc += A * dlt - C;
        }
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = (sm[0] + sm[1] + sm[2] + sm[3]) / (double)numel;
}

// K x K window, stride 2, windows clipped at the bottom/right edge (ceil_mode partial windows).  A workgroup covers 4 output rows of one
// plane; a thread produces TWO adjacent outputs from a K x (K + 2) input window, so a wave reads 3 contiguous row segments (every byte
// of them used) and no index needs a division per element (one per workgroup row).  HBM-bound: in + out bytes once.
// IDX (K = 3): also the window's first maximum in row-major order as a tap index 0..8 (torch's rule for the gradient), one byte per
// output -- gradient mode's backward then needs neither the input map nor the 9-tap scan (mgf_maxpool3x3s2_ceil_bwd_idx_f32).
template <int K, bool IDX = false>
__global__ __launch_bounds__(256) void maxpool_kernel(float* y, const float* x, int nc, int in_h, int in_w, int out_h, int out_w,
                                                      uint8_t* idx = nullptr) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);              // (plane, output row) flattened
    if (row >= nc * out_h) return;
    const int pl = row / out_h, oy = row - pl * out_h;
    const float* xp = x + (int64_t)pl * in_h * in_w;
    float* yp = y + ((int64_t)pl * out_h + oy) * out_w;
    const int lane = threadIdx.x & 63;
    for (int ox = 2 * lane; ox < out_w; ox += 128) {
        float m0 = -3.0e38f, m1 = -3.0e38f;
        int t0 = 0, t1 = 0;
#pragma unroll
        for (int dy = 0; dy < K; ++dy) {
            const int iy = oy * 2 + dy;
            if (iy >= in_h) continue;
            const float* rp = xp + (int64_t)iy * in_w + 2 * ox;
            float v[K + 2];
#pragma unroll
            for (int dx = 0; dx < K + 2; ++dx) v[dx] = (2 * ox + dx < in_w) ? rp[dx] : -3.0e38f;
#pragma unroll
            for (int dx = 0; dx < K; ++dx) {
                if (IDX) {                                               // strict >: the first maximum wins; tap (0, 0) always exists
                    if (v[dx] > m0) { m0 = v[dx]; t0 = dy * K + dx; }
                    if (v[dx + 2] > m1) { m1 = v[dx + 2]; t1 = dy * K + dx; }
                } else {
                    m0 = fmaxf(m0, v[dx]); m1 = fmaxf(m1, v[dx + 2]);
                }
            }
        }
        yp[ox] = m0;
        if (IDX) idx[((int64_t)pl * out_h + oy) * out_w + ox] = (uint8_t)t0;
        if (ox + 1 < out_w) {
            yp[ox + 1] = m1;
            if (IDX) idx[((int64_t)pl * out_h + oy) * out_w + ox + 1] = (uint8_t)t1;
        }
    }
}

// latent_n[j] = latent_in + eps[s_j] * sigma[s_j], s_j = min(*step + j, steps_total - 1), j < batch
__global__ __launch_bounds__(256) void perturb_kernel(float* latent_n, const float* latent_in, const float* eps, const float* sigma,
                                                      const int32_t* step, int batch, int steps_total, int64_t numel) {
    const int64_t total = (int64_t)batch * numel;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int j = (int)(idx / numel);
        const int64_t i = idx - (int64_t)j * numel;
        int s = *step + j;
        if (s > steps_total - 1) s = steps_total - 1;
        float prod = eps[(int64_t)s * numel + i] * sigma[s];
        asm volatile("" : "+v"(prod));                 // opaque to the fma combiner: two roundings like torch -> bit-exact latents
        latent_n[idx] = latent_in[i] + prod;
    }
}

// projection_example_v2_percept.py:131-166: the optimised latent is [1, copies = 18, k D], every copy gets its own noise, and the generator
// sees `torch.mean(latent_n, 1)`.  The mean must come out bit for bit (the kept latent IS that mean): torch's CPU sum over an outer dimension
// (SumKernel.cpp, cascade_sum / multi_row_sum) adds the rows in blocks of 16 -- each block sequentially from zero, the block sums sequentially
// into a second accumulator -- adds the remaining rows sequentially from zero, then tail + blocks; the mean divides by the count.  Same
// order here, every step of it a separate float32 rounding (no fma: the products are two roundings like torch's `randn_like(..) * strength`).
__global__ __launch_bounds__(256) void perturb_mean_kernel(float* latent_n, const float* latent_in, const float* eps, const float* sigma,
                                                           const int32_t* step, int batch, int steps_total, int64_t numel, int copies) {
    const int64_t total = (int64_t)batch * numel;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int j = (int)(idx / numel);
        const int64_t i = idx - (int64_t)j * numel;
        int s = *step + j;
        if (s > steps_total - 1) s = steps_total - 1;
        const float sg = sigma[s], base = latent_in[i];
        const float* e = eps + ((int64_t)s * copies) * numel + i;
        float blocks = 0.f, run = 0.f;
        for (int c = 0; c < copies; ++c) {
            float prod = e[(int64_t)c * numel] * sg;
            asm volatile("" : "+v"(prod));                 // opaque to the fma combiner
            float v = base + prod;
            asm volatile("" : "+v"(v));
            run = run + v;
            asm volatile("" : "+v"(run));
            if ((c & 15) == 15) { blocks = blocks + run; asm volatile("" : "+v"(blocks)); run = 0.f; }
        }
        latent_n[idx] = (run + blocks) / (float)copies;
    }
}

// candidates j = 0 .. batch-1 are examined in step order, exactly as the sequential loop would
__global__ __launch_bounds__(256) void select_kernel(double* min_loss, float* best_latent, int32_t* best_step, double* losses_out,
                                                     const float* latent_n, int64_t numel, const float* p_loss, const double* w_loss,
                                                     const float* mse_loss, double lamda, float beta, int32_t* step, const int32_t* valid_tab,
                                                     int batch, int steps_total, int32_t* take_slot, int32_t* trail_count, int trail_capacity,
                                                     int32_t* trail_steps, double* trail_losses) {
    __shared__ int take;
    const int s0 = *step;
    int last_full_j = -1;            // (thread 0) candidate of this batch that already sits in the last trail slot
    for (int j = 0; j < batch; ++j) {
        const int s = s0 + j;
        if (s >= steps_total) {
            if (take_slot && threadIdx.x == 0) take_slot[j] = -1;
            continue;
        }
        if (threadIdx.x == 0) {
            const int valid = valid_tab ? valid_tab[s] : 1;
            // same evaluation order and promotions as `p_loss + lamda * w_loss + beta * mse_loss` with a float64 wing term
            double total = 0.0;
            if (p_loss) total += (double)p_loss[j];
            if (w_loss) total += lamda * w_loss[j];
            if (mse_loss) total += (double)(beta * mse_loss[j]);
            if (losses_out) losses_out[s] = valid ? total : __longlong_as_double(0x7ff8000000000000LL);
            take = valid && total < min_loss[0];
            if (take) { min_loss[0] = total; best_step[0] = s; }
            if (take_slot) {
                int slot = -1;
                if (take) {
                    const int cnt = *trail_count;
                    slot = cnt < trail_capacity ? cnt : trail_capacity - 1;
                    if (slot == trail_capacity - 1) {            // two candidates of one batch must not both be copied into the last slot
                        if (last_full_j >= 0) take_slot[last_full_j] = -1;
                        last_full_j = j;
                    }
                    trail_steps[slot] = s;
                    trail_losses[slot] = total;
                    *trail_count = cnt + 1;
                }
                take_slot[j] = slot;
            }
        }
        __syncthreads();
        if (take)
            for (int64_t i = threadIdx.x; i < numel; i += 256) best_latent[i] = latent_n[(int64_t)j * numel + i];
        __syncthreads();
    }
    if (threadIdx.x == 0) *step = s0 + batch < steps_total ? s0 + batch : steps_total;
}

// trail_imgs[take_slot[j]] = imgs[j] for the improving candidates of a batch; grid (blocks, batch), float4 body + scalar tail
__global__ __launch_bounds__(256) void keep_improvements_kernel(float* __restrict__ trail, const float* __restrict__ imgs, int64_t numel,
                                                                const int32_t* __restrict__ take_slot) {
    const int j = blockIdx.y;
    const int slot = take_slot[j];
    if (slot < 0) return;
    const float* src = imgs + (int64_t)j * numel;
    float* dst = trail + (int64_t)slot * numel;
    const int64_t stride = (int64_t)gridDim.x * 256;
    const int64_t tid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if ((numel & 3) == 0) {
        const int64_t n4 = numel >> 2;
        for (int64_t i = tid; i < n4; i += stride) reinterpret_cast<float4*>(dst)[i] = reinterpret_cast<const float4*>(src)[i];
    } else {
        for (int64_t i = tid; i < numel; i += stride) dst[i] = src[i];
    }
}

__global__ __launch_bounds__(256) void to_uint8_kernel(uint8_t* out, const float* img, int c, int h, int w) {
    const int64_t hw = (int64_t)h * w, total = hw * c;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int ch = (int)(i % c);
        const int64_t px = i / c;
        float v = __fadd_rn(__fmul_rn(img[(int64_t)ch * hw + px], 127.5f), 127.5f);      // two roundings like numpy's `data * scale + bias` (misc.py:103-104)
        v = rintf(v);
        v = fminf(fmaxf(v, 0.f), 255.f);
        out[i] = (uint8_t)v;
    }
}

// ---- the gray uint8 image the drivers hand to dlib (...sqz_MSE.py:159-163): cv2.normalize(img, None, 0, 255, NORM_MINMAX, CV_8U) over the
// WHOLE float image, then cv2.cvtColor(COLOR_BGR2GRAY) on RGB-ordered data.  Two passes per candidate: partial minima / maxima, then the
// conversion (every block of the second pass first folds the candidate's partials: min / max do not depend on the order).
constexpr int GRAY_PARTS = 256;

__global__ __launch_bounds__(256) void minmax_partial_kernel(float* part, const float* img, int64_t per) {
    __shared__ float slo[4], shi[4];
    const float* x = img + (int64_t)blockIdx.y * per;
    float lo = INFINITY, hi = -INFINITY;
    const int64_t n4 = per / 4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const float4 v = reinterpret_cast<const float4*>(x)[i];
        lo = fminf(fminf(lo, fminf(v.x, v.y)), fminf(v.z, v.w));
        hi = fmaxf(fmaxf(hi, fmaxf(v.x, v.y)), fmaxf(v.z, v.w));
    }
    for (int64_t i = n4 * 4 + (int64_t)blockIdx.x * 256 + threadIdx.x; i < per; i += (int64_t)gridDim.x * 256) { lo = fminf(lo, x[i]); hi = fmaxf(hi, x[i]); }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { lo = fminf(lo, __shfl_xor(lo, o, 64)); hi = fmaxf(hi, __shfl_xor(hi, o, 64)); }
    if ((threadIdx.x & 63) == 0) { slo[threadIdx.x >> 6] = lo; shi[threadIdx.x >> 6] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float* p = part + ((int64_t)blockIdx.y * GRAY_PARTS + blockIdx.x) * 2;
        p[0] = fminf(fminf(slo[0], slo[1]), fminf(slo[2], slo[3]));
        p[1] = fmaxf(fmaxf(shi[0], shi[1]), fmaxf(shi[2], shi[3]));
    }
}

// u8 = clip(rint((double(x) - lo) * (255.0 / (hi - lo)))), the scale 0 for a flat image, exactly the arithmetic of drivers.reference_gray_u8;
// gray = (c0 * 1868 + c1 * 9617 + c2 * 4899 + (1 << 13)) >> 14: OpenCV's 8-bit BGR2GRAY coefficients (B 1868, G 9617, R 4899) with channel 0 -- RED
// in the generator's RGB order -- in the blue slot, as the drivers call it.  img [n,3,h,w] planar -> gray [n,h,w]
__global__ __launch_bounds__(256) void gray_u8_kernel(uint8_t* gray, const float* img, const float* part, int nparts, int64_t hw) {
    __shared__ float slo[4], shi[4];
    const int n = blockIdx.y;
    float lo = INFINITY, hi = -INFINITY;
    for (int i = threadIdx.x; i < nparts; i += 256) { lo = fminf(lo, part[((int64_t)n * GRAY_PARTS + i) * 2]); hi = fmaxf(hi, part[((int64_t)n * GRAY_PARTS + i) * 2 + 1]); }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { lo = fminf(lo, __shfl_xor(lo, o, 64)); hi = fmaxf(hi, __shfl_xor(hi, o, 64)); }
    if ((threadIdx.x & 63) == 0) { slo[threadIdx.x >> 6] = lo; shi[threadIdx.x >> 6] = hi; }
    __syncthreads();
    const double dlo = (double)fminf(fminf(slo[0], slo[1]), fminf(slo[2], slo[3])), dhi = (double)fmaxf(fmaxf(shi[0], shi[1]), fmaxf(shi[2], shi[3]));
    const double scale = dhi > dlo ? 255.0 / (dhi - dlo) : 0.0;
    const float* x = img + (int64_t)n * 3 * hw;
    uint8_t* g = gray + (int64_t)n * hw;
    auto q = [&](float v) -> unsigned {
        double t = rint(__dmul_rn(__dsub_rn((double)v, dlo), scale));
        t = t < 0.0 ? 0.0 : (t > 255.0 ? 255.0 : t);
        return (unsigned)t;
    };
    for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < hw; i += (int64_t)gridDim.x * 1024) {
        if (i + 4 <= hw && (hw & 3) == 0) {
            const float4 a = *reinterpret_cast<const float4*>(x + i), b = *reinterpret_cast<const float4*>(x + hw + i),
                         c = *reinterpret_cast<const float4*>(x + 2 * hw + i);
            const unsigned g0 = (q(a.x) * 1868u + q(b.x) * 9617u + q(c.x) * 4899u + (1u << 13)) >> 14;
            const unsigned g1 = (q(a.y) * 1868u + q(b.y) * 9617u + q(c.y) * 4899u + (1u << 13)) >> 14;
            const unsigned g2 = (q(a.z) * 1868u + q(b.z) * 9617u + q(c.z) * 4899u + (1u << 13)) >> 14;
            const unsigned g3 = (q(a.w) * 1868u + q(b.w) * 9617u + q(c.w) * 4899u + (1u << 13)) >> 14;
            *reinterpret_cast<uint32_t*>(g + i) = g0 | (g1 << 8) | (g2 << 16) | (g3 << 24);
        } else {
            for (int64_t j = i; j < hw && j < i + 4; ++j)
                g[j] = (uint8_t)((q(x[j]) * 1868u + q(x[hw + j]) * 9617u + q(x[2 * hw + j]) * 4899u + (1u << 13)) >> 14);
        }
    }
}

}  // namespace

extern "C" int64_t mgf_reference_gray_scratch_floats(void) { return 2 * GRAY_PARTS; }

extern "C" int mgf_reference_gray_u8(uint8_t* gray, const float* img, int32_t n, int32_t h, int32_t w, float* scratch, mgf_stream_t stream) {
    MGF_REQUIRE(gray && img && scratch && n >= 1 && n <= 65535 && h >= 1 && w >= 1, MGF_EINVAL, "reference_gray_u8: bad arguments");
    MGF_REQUIRE(((uintptr_t)img % 16) == 0 && ((uintptr_t)gray % 4) == 0, MGF_EINVAL, "reference_gray_u8: img must be 16-byte, gray 4-byte aligned");
    const int64_t hw = (int64_t)h * w, per = 3 * hw;
    MGF_REQUIRE(per % 4 == 0 || n == 1, MGF_EINVAL, "reference_gray_u8: a batch needs 16-byte aligned samples (3 h w a multiple of 4)");
    const int parts = (int)(mgf_cdiv(per, 256 * 16) < GRAY_PARTS ? mgf_cdiv(per, 256 * 16) : GRAY_PARTS);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(minmax_partial_kernel, dim3(parts, n), dim3(256), 0, st, scratch, img, per);
    const int blocks = (int)(mgf_cdiv(hw, 1024 * 4) < 1024 ? mgf_cdiv(hw, 1024 * 4) : 1024);
    hipLaunchKernelGGL(gray_u8_kernel, dim3(blocks < 1 ? 1 : blocks, n), dim3(256), 0, st, gray, img, scratch, parts, hw);
    MGF_CHECK_LAUNCH("reference_gray_u8");
    return MGF_OK;
}

extern "C" int64_t mgf_reduce_scratch_floats(void) { return RED_BLOCKS; }

extern "C" int mgf_mse_f32(float* out, const float* a, const float* b, int32_t n, int64_t numel, int64_t b_batch_stride, float scale,
                           int32_t accumulate, float* scratch, mgf_stream_t stream) {
    MGF_REQUIRE(out && a && b && scratch && numel >= 1 && n >= 1 && n <= 65535, MGF_EINVAL, "mse: bad arguments");
    MGF_REQUIRE(((uintptr_t)a % 16 == 0) && ((uintptr_t)b % 16 == 0) && (numel % 4 == 0 || n == 1) && b_batch_stride % 4 == 0, MGF_EINVAL,
                "mse: inputs must be 16-byte aligned per sample");
    const int grid = (int)(mgf_cdiv(numel, 256 * 8) < 1024 ? mgf_cdiv(numel, 256 * 8) : 1024);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(mse_partial_kernel, dim3(grid, n), dim3(256), 0, st, scratch, a, b, numel, b_batch_stride);
    hipLaunchKernelGGL(finish_kernel, dim3(n), dim3(256), 0, st, out, scratch, grid, scale / (float)numel, accumulate);
    MGF_CHECK_LAUNCH("mse");
    return MGF_OK;
}

extern "C" int mgf_lpips_unit_f32(float* out, const float* f, int32_t n, int32_t c, int64_t hw, mgf_stream_t stream) {
    MGF_REQUIRE(out && f && n >= 1 && n <= 65535 && c >= 1 && hw >= 1, MGF_EINVAL, "lpips_unit: bad arguments");
    int grid = 0;
    const int rc = launch_lpips_layer<true>(nullptr, out, f, nullptr, nullptr, n, c, hw, 0, (hipStream_t)stream, &grid);
    if (rc != MGF_OK) return rc;
    MGF_CHECK_LAUNCH("lpips_unit");
    return MGF_OK;
}

extern "C" int mgf_lpips_layer_f32(float* out, const float* f0, const float* f1_unit, const float* lin, int32_t n, int32_t c, int64_t hw,
                                   int64_t f1_batch_stride, int32_t accumulate, float* scratch, mgf_stream_t stream) {
    return mgf_lpips_layer_stats_f32(out, nullptr, f0, f1_unit, lin, n, c, hw, f1_batch_stride, accumulate, scratch, stream);
}

extern "C" int mgf_lpips_layer_stats_f32(float* out, float* stats, const float* f0, const float* f1_unit, const float* lin, int32_t n, int32_t c,
                                         int64_t hw, int64_t f1_batch_stride, int32_t accumulate, float* scratch, mgf_stream_t stream) {
    MGF_REQUIRE(out && f0 && f1_unit && lin && scratch && n >= 1 && n <= 65535 && c >= 1 && hw >= 1, MGF_EINVAL, "lpips_layer: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    int grid = 0;
    const int rc = launch_lpips_layer<false>(scratch, nullptr, f0, f1_unit, lin, n, c, hw, f1_batch_stride, st, &grid, stats);
    if (rc != MGF_OK) return rc;
    // spatial mean per sample (networks_basic.py:85-87)
    hipLaunchKernelGGL(finish_kernel, dim3(n), dim3(256), 0, st, out, scratch, grid, 1.0f / (float)hw, accumulate);
    MGF_CHECK_LAUNCH("lpips_layer");
    return MGF_OK;
}

/* mgf_lpips_layer_stats_f32 without its finish launch: the tap's partial sums stay in `scratch` (n * mgf_reduce_scratch_floats() floats: one
 * set per tap) and *nparts_out says how many there are per sample; mgf_lpips_finish_taps_f32 then adds all taps to out in tap order -- the
 * same sums in the same order as one finish per tap, in ONE launch (the loss VALUE is a by-product in gradient mode: nothing waits for it). */
extern "C" int mgf_lpips_layer_defer_f32(float* scratch, float* stats, const float* f0, const float* f1_unit, const float* lin, int32_t n, int32_t c,
                                         int64_t hw, int64_t f1_batch_stride, int32_t* nparts_out, mgf_stream_t stream) {
    MGF_REQUIRE(scratch && f0 && f1_unit && lin && nparts_out && n >= 1 && n <= 65535 && c >= 1 && hw >= 1, MGF_EINVAL, "lpips_layer_defer: bad arguments");
    int grid = 0;
    const int rc = launch_lpips_layer<false>(scratch, nullptr, f0, f1_unit, lin, n, c, hw, f1_batch_stride, (hipStream_t)stream, &grid, stats);
    if (rc != MGF_OK) return rc;
    *nparts_out = grid;
    MGF_CHECK_LAUNCH("lpips_layer_defer");
    return MGF_OK;
}

extern "C" int mgf_lpips_finish_taps_f32(float* out, const float* scratch, int64_t set_stride_floats, int32_t ntaps, const int32_t* nparts,
                                         const float* scales, int32_t n, int32_t accumulate, mgf_stream_t stream) {
    MGF_REQUIRE(out && scratch && nparts && scales && ntaps >= 1 && ntaps <= 8 && n >= 1 && n <= 65535, MGF_EINVAL, "lpips_finish_taps: bad arguments (1..8 taps)");
    FinishSets fs;
    for (int t = 0; t < 8; ++t) { fs.nparts[t] = t < ntaps ? nparts[t] : 0; fs.scale[t] = t < ntaps ? scales[t] : 0.f; }
    for (int t = 0; t < ntaps; ++t) MGF_REQUIRE(nparts[t] >= 1 && nparts[t] <= RED_BLOCKS, MGF_EINVAL, "lpips_finish_taps: bad partial count");
    hipLaunchKernelGGL(finish_multi_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, out, scratch, set_stride_floats, ntaps, fs, accumulate);
    MGF_CHECK_LAUNCH("lpips_finish_taps");
    return MGF_OK;
}

extern "C" int mgf_wing_loss_f64(double* out, const double* pred, const double* target, int32_t n, int64_t numel, double omega,
                                 double epsilon, const int32_t* pred_step, int32_t max_row, mgf_stream_t stream) {
    MGF_REQUIRE(out && pred && target && numel >= 1 && n >= 1, MGF_EINVAL, "wing_loss: bad arguments");
    hipLaunchKernelGGL(wing_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, out, pred, target, numel, omega, epsilon, pred_step, max_row);
    MGF_CHECK_LAUNCH("wing_loss");
    return MGF_OK;
}

extern "C" int mgf_adaptive_wing_loss_f64(double* out, const double* pred, const double* target, int32_t n, int64_t numel, double omega,
                                          double theta, double epsilon, double alpha, const int32_t* pred_step, int32_t max_row,
                                          mgf_stream_t stream) {
    MGF_REQUIRE(out && pred && target && numel >= 1 && n >= 1, MGF_EINVAL, "adaptive_wing_loss: bad arguments");
    hipLaunchKernelGGL(awing_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, out, pred, target, numel, omega, theta, epsilon, alpha,
                       pred_step, max_row);
    MGF_CHECK_LAUNCH("adaptive_wing_loss");
    return MGF_OK;
}

namespace {
// 3x3 / stride-2 ceil-mode pool through LDS: 64 x 8 outputs per workgroup from a 129 x 17 input patch loaded by rows, two outputs per
// lane -- for maps wider than the 512 columns maxpool3x3s2_rows_kernel keeps in registers.  Same 2.9 TB/s as the row-per-wave form on
// the large maps (3.1 against 2.45 on 127^2): what the pool wants is neither -- see maxpool3x3s2_rows_kernel.
__global__ __launch_bounds__(256) void maxpool3x3s2_tiled_kernel(float* __restrict__ y, const float* __restrict__ x, int in_h, int in_w, int out_h,
                                                                 int out_w, int tiles_x, int tiles_y) {
    constexpr int TW = 64, TH = 8, PW = 2 * TW + 1, PH = 2 * TH + 1;
    __shared__ float patch[PH][PW + 1];
    const int tid = threadIdx.x;
    const int tx = blockIdx.x % tiles_x, ty = (blockIdx.x / tiles_x) % tiles_y;
    const int64_t pl = blockIdx.x / ((int64_t)tiles_x * tiles_y);
    const int ox0 = tx * TW, oy0 = ty * TH, ix0 = 2 * ox0, iy0 = 2 * oy0;
    const float* xp = x + pl * in_h * in_w;
    const float NEG = -3.0e38f;
    {
        const int cc = tid & 127, ix = ix0 + cc;
        const bool xin = ix < in_w;
#pragma unroll
        for (int r = tid >> 7; r < PH; r += 2) {
            const int iy = iy0 + r;
            patch[r][cc] = (xin && iy < in_h) ? xp[(int64_t)iy * in_w + ix] : NEG;
        }
        if (tid < PH) {                                              // the 129th column
            const int iy = iy0 + tid, ix2 = ix0 + 128;
            patch[tid][128] = (ix2 < in_w && iy < in_h) ? xp[(int64_t)iy * in_w + ix2] : NEG;
        }
    }
    __syncthreads();
    const int lx = tid & 63;
    if (ox0 + lx >= out_w) return;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int ly = (tid >> 6) + 4 * q;
        const int oy = oy0 + ly;
        if (oy >= out_h) break;
        float m = NEG;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) m = fmaxf(m, patch[2 * ly + dy][2 * lx + dx]);
        y[(pl * out_h + oy) * out_w + ox0 + lx] = m;
    }
}
}  // namespace

namespace {
// What the strided window costs (tools/maxpool_fwd_micro.py, 25 x 64 x 511^2): five overlapping 4-byte loads per row and output pair at
// a lane stride of 8 bytes ran at 2.9 TB/s, two non-overlapping ones (wrong results) at 4.4, without the stores still 2.9; a read-only
// stream reaches 6 TB/s (tools/probes/hbm_read.hip).  So:
// 3x3 / stride-2 ceil-mode pool, one wave per output row with every input element loaded ONCE per wave and fully coalesced: lane l holds
// the column-wise maximum of the row's three input rows at columns l, l + 64, ... (S slots), the stride-2 three-wide windows are
// then formed across lanes (two shuffles per slot; the last lanes of a slot take the next slot's first ones) and the even lanes store.
// IDX: also each window's first maximum in row-major order as a tap index (torch's gradient rule): the column maximum remembers its
// first row, and among the window's columns with the maximal value the smallest row, then the smallest column, wins.
template <int S, bool IDX = false>
__global__ __launch_bounds__(256) void maxpool3x3s2_rows_kernel(float* __restrict__ y, const float* __restrict__ x, int nc, int in_h, int in_w,
                                                                int out_h, int out_w, uint8_t* __restrict__ idx = nullptr) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);              // (plane, output row) flattened
    if (row >= nc * out_h) return;
    const int pl = row / out_h, oy = row - pl * out_h;
    const int lane = threadIdx.x & 63;
    const float* xp = x + (int64_t)pl * in_h * in_w + (int64_t)(2 * oy) * in_w;
    const float NEG = -3.0e38f;
    const bool r1 = 2 * oy + 1 < in_h, r2 = 2 * oy + 2 < in_h;
    float v[S + 1];
    int rw[IDX ? S + 1 : 1];                                         // IDX: first row of the column maximum
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const int e = 64 * s + lane;
        const bool in = e < in_w;
        const float a = in ? xp[e] : NEG, b = (in && r1) ? xp[in_w + e] : NEG, c = (in && r2) ? xp[2 * in_w + e] : NEG;
        if (IDX) {
            float m = a; int r = 0;
            if (b > m) { m = b; r = 1; }
            if (c > m) { m = c; r = 2; }
            v[s] = m; rw[s] = r;
        } else {
            v[s] = fmaxf(fmaxf(a, b), c);
        }
    }
    v[S] = NEG;
    if (IDX) rw[S] = 0;
    float* yp = y + ((int64_t)pl * out_h + oy) * out_w;
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const float t1 = __shfl(v[s], (lane + 1) & 63), n1 = __shfl(v[s + 1], (lane + 1) & 63);
        const float t2 = __shfl(v[s], (lane + 2) & 63), n2 = __shfl(v[s + 1], (lane + 2) & 63);
        const float c1 = lane < 63 ? t1 : n1, c2 = lane < 62 ? t2 : n2;
        const int ox = 32 * s + (lane >> 1);
        const bool st = !(lane & 1) && ox < out_w;
        if (IDX) {
            const int q1 = __shfl(rw[s], (lane + 1) & 63), p1 = __shfl(rw[s + 1], (lane + 1) & 63);
            const int q2 = __shfl(rw[s], (lane + 2) & 63), p2 = __shfl(rw[s + 1], (lane + 2) & 63);
            const int w1 = lane < 63 ? q1 : p1, w2 = lane < 62 ? q2 : p2;
            float m = v[s]; int r = rw[s], dx = 0;
            if (c1 > m || (c1 == m && w1 < r)) { m = c1; r = w1; dx = 1; }
            if (c2 > m || (c2 == m && w2 < r)) { m = c2; r = w2; dx = 2; }
            if (st) { yp[ox] = m; idx[((int64_t)pl * out_h + oy) * out_w + ox] = (uint8_t)(3 * r + dx); }
        } else {
            if (st) yp[ox] = fmaxf(fmaxf(v[s], c1), c2);
        }
    }
}
}  // namespace

extern "C" int mgf_maxpool3x3s2_ceil_f32(float* y, const float* x, int32_t nc, int32_t in_h, int32_t in_w, int32_t out_h, int32_t out_w,
                                         mgf_stream_t stream) {
    MGF_REQUIRE(y && x && nc >= 1 && in_h >= 1 && in_w >= 1, MGF_EINVAL, "maxpool: bad arguments");
    // ceil_mode output size; the last window must start inside the input (torch.nn.MaxPool2d rule)
    auto osz = [](int in) { int o = (in - 3 + 1) / 2 + 1; if ((o - 1) * 2 >= in) --o; return o; };
    MGF_REQUIRE(out_h == osz(in_h) && out_w == osz(in_w), MGF_EINVAL, "maxpool: output must be %dx%d (got %dx%d)", osz(in_h), osz(in_w),
                out_h, out_w);
    MGF_REQUIRE((int64_t)nc * out_h <= INT32_MAX - 4, MGF_ETOOBIG, "maxpool: too many rows");
    static const int rows_env = [] { const char* e = mgf_knob("MGF_POOL_ROWS"); return e ? atoi(e) : -1; }();
    if (rows_env != 0 && in_w <= 512 && in_w >= 64) {
        const dim3 grid((unsigned)mgf_cdiv((int64_t)nc * out_h, 4));
        const int slots = (int)mgf_cdiv(in_w, 64);
#define MGF_POOL_ROWS(SL) hipLaunchKernelGGL(maxpool3x3s2_rows_kernel<SL>, grid, dim3(256), 0, (hipStream_t)stream, y, x, nc, in_h, in_w, out_h, out_w)
        switch (slots) {
            case 1: MGF_POOL_ROWS(1); break; case 2: MGF_POOL_ROWS(2); break; case 3: MGF_POOL_ROWS(3); break; case 4: MGF_POOL_ROWS(4); break;
            case 5: MGF_POOL_ROWS(5); break; case 6: MGF_POOL_ROWS(6); break; case 7: MGF_POOL_ROWS(7); break; default: MGF_POOL_ROWS(8); break;
        }
#undef MGF_POOL_ROWS
        MGF_CHECK_LAUNCH("maxpool");
        return MGF_OK;
    }
    static const int tiled_env = [] { const char* e = mgf_knob("MGF_POOL_TILED"); return e ? atoi(e) : -1; }();
    const int64_t tiles = (int64_t)nc * mgf_cdiv(out_w, 64) * mgf_cdiv(out_h, 8);
    if (tiled_env != 0 && out_w >= 32 && tiles <= INT32_MAX) {
        hipLaunchKernelGGL(maxpool3x3s2_tiled_kernel, dim3((unsigned)tiles), dim3(256), 0, (hipStream_t)stream, y, x, in_h, in_w, out_h, out_w,
                           (int)mgf_cdiv(out_w, 64), (int)mgf_cdiv(out_h, 8));
        MGF_CHECK_LAUNCH("maxpool");
        return MGF_OK;
    }
    hipLaunchKernelGGL(maxpool_kernel<3>, dim3((unsigned)mgf_cdiv((int64_t)nc * out_h, 4)), dim3(256), 0, (hipStream_t)stream, y, x, nc, in_h, in_w,
                       out_h, out_w);
    MGF_CHECK_LAUNCH("maxpool");
    return MGF_OK;
}

extern "C" int mgf_maxpool3x3s2_ceil_idx_f32(float* y, uint8_t* idx, const float* x, int32_t nc, int32_t in_h, int32_t in_w, int32_t out_h,
                                             int32_t out_w, mgf_stream_t stream) {
    MGF_REQUIRE(y && idx && x && nc >= 1 && in_h >= 1 && in_w >= 1, MGF_EINVAL, "maxpool_idx: bad arguments");
    auto osz = [](int in) { int o = (in - 3 + 1) / 2 + 1; if ((o - 1) * 2 >= in) --o; return o; };
    MGF_REQUIRE(out_h == osz(in_h) && out_w == osz(in_w), MGF_EINVAL, "maxpool_idx: output must be %dx%d (got %dx%d)", osz(in_h), osz(in_w),
                out_h, out_w);
    MGF_REQUIRE((int64_t)nc * out_h <= INT32_MAX - 4, MGF_ETOOBIG, "maxpool_idx: too many rows");
    static const int rows_env = [] { const char* e = mgf_knob("MGF_POOL_ROWS"); return e ? atoi(e) : -1; }();
    if (rows_env != 0 && in_w <= 512 && in_w >= 64) {
        const dim3 grid((unsigned)mgf_cdiv((int64_t)nc * out_h, 4));
#define MGF_POOL_ROWS(SL) hipLaunchKernelGGL((maxpool3x3s2_rows_kernel<SL, true>), grid, dim3(256), 0, (hipStream_t)stream, y, x, nc, in_h, in_w, out_h, out_w, idx)
        switch ((int)mgf_cdiv(in_w, 64)) {
            case 1: MGF_POOL_ROWS(1); break; case 2: MGF_POOL_ROWS(2); break; case 3: MGF_POOL_ROWS(3); break; case 4: MGF_POOL_ROWS(4); break;
            case 5: MGF_POOL_ROWS(5); break; case 6: MGF_POOL_ROWS(6); break; case 7: MGF_POOL_ROWS(7); break; default: MGF_POOL_ROWS(8); break;
        }
#undef MGF_POOL_ROWS
        MGF_CHECK_LAUNCH("maxpool_idx");
        return MGF_OK;
    }
    hipLaunchKernelGGL((maxpool_kernel<3, true>), dim3((unsigned)mgf_cdiv((int64_t)nc * out_h, 4)), dim3(256), 0, (hipStream_t)stream, y, x, nc,
                       in_h, in_w, out_h, out_w, idx);
    MGF_CHECK_LAUNCH("maxpool_idx");
    return MGF_OK;
}

extern "C" int mgf_maxpool_s2_floor_f32(float* y, const float* x, int32_t nc, int32_t in_h, int32_t in_w, int32_t ksize, mgf_stream_t stream) {
    MGF_REQUIRE(y && x && nc >= 1 && (ksize == 2 || ksize == 3) && in_h >= ksize && in_w >= ksize, MGF_EINVAL, "maxpool_s2_floor: bad arguments");
    const int out_h = (in_h - ksize) / 2 + 1, out_w = (in_w - ksize) / 2 + 1;
    MGF_REQUIRE((int64_t)nc * out_h <= INT32_MAX - 4, MGF_ETOOBIG, "maxpool_s2_floor: too many rows");
    const dim3 grid((unsigned)mgf_cdiv((int64_t)nc * out_h, 4));
    if (ksize == 2) hipLaunchKernelGGL(maxpool_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, y, x, nc, in_h, in_w, out_h, out_w);
    else hipLaunchKernelGGL(maxpool_kernel<3>, grid, dim3(256), 0, (hipStream_t)stream, y, x, nc, in_h, in_w, out_h, out_w);
    MGF_CHECK_LAUNCH("maxpool_s2_floor");
    return MGF_OK;
}

extern "C" int mgf_latent_perturb(float* latent_n, const float* latent_in, const float* eps, const float* sigma, const int32_t* step,
                                  int32_t batch, int32_t steps_total, int64_t numel, mgf_stream_t stream) {
    MGF_REQUIRE(latent_n && latent_in && eps && sigma && step && numel >= 1 && batch >= 1 && steps_total >= 1, MGF_EINVAL,
                "latent_perturb: bad arguments");
    hipLaunchKernelGGL(perturb_kernel, dim3((unsigned)mgf_cdiv(numel * batch, 256)), dim3(256), 0, (hipStream_t)stream, latent_n, latent_in,
                       eps, sigma, step, batch, steps_total, numel);
    MGF_CHECK_LAUNCH("latent_perturb");
    return MGF_OK;
}

extern "C" int mgf_latent_perturb_mean(float* latent_n, const float* latent_in, const float* eps, const float* sigma, const int32_t* step,
                                       int32_t batch, int32_t steps_total, int64_t numel, int32_t copies, mgf_stream_t stream) {
    MGF_REQUIRE(latent_n && latent_in && eps && sigma && step && numel >= 1 && batch >= 1 && steps_total >= 1, MGF_EINVAL,
                "latent_perturb_mean: bad arguments");
    MGF_REQUIRE(copies >= 1 && copies <= 255, MGF_EUNSUPPORTED, "latent_perturb_mean: 1 .. 255 copies (torch's two-level summation order; got %d)", copies);
    hipLaunchKernelGGL(perturb_mean_kernel, dim3((unsigned)mgf_cdiv(numel * batch, 256)), dim3(256), 0, (hipStream_t)stream, latent_n, latent_in,
                       eps, sigma, step, batch, steps_total, numel, copies);
    MGF_CHECK_LAUNCH("latent_perturb_mean");
    return MGF_OK;
}

extern "C" int mgf_select_best(double* min_loss, float* best_latent, int32_t* best_step, double* losses_out, const float* latent_n,
                               int64_t numel, const float* p_loss, const double* w_loss, const float* mse_loss, double lamda, float beta,
                               int32_t* step, const int32_t* valid, int32_t batch, int32_t steps_total, int32_t* take_slot,
                               int32_t* trail_count, int32_t trail_capacity, int32_t* trail_steps, double* trail_losses, mgf_stream_t stream) {
    MGF_REQUIRE(min_loss && best_latent && best_step && latent_n && step && numel >= 1 && batch >= 1 && steps_total >= 1, MGF_EINVAL,
                "select_best: bad arguments");
    MGF_REQUIRE(!take_slot || (trail_count && trail_steps && trail_losses && trail_capacity >= 1), MGF_EINVAL,
                "select_best: take_slot needs trail_count, trail_steps, trail_losses and a capacity >= 1");
    hipLaunchKernelGGL(select_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, min_loss, best_latent, best_step, losses_out, latent_n,
                       numel, p_loss, w_loss, mse_loss, lamda, beta, step, valid, batch, steps_total, take_slot, trail_count, trail_capacity,
                       trail_steps, trail_losses);
    MGF_CHECK_LAUNCH("select_best");
    return MGF_OK;
}

extern "C" int mgf_keep_improvements(float* trail_imgs, const float* imgs, int64_t numel, const int32_t* take_slot, int32_t batch,
                                     mgf_stream_t stream) {
    MGF_REQUIRE(trail_imgs && imgs && take_slot && numel >= 1 && batch >= 1 && batch <= 65535, MGF_EINVAL, "keep_improvements: bad arguments");
    MGF_REQUIRE((numel & 3) != 0 || ((((uintptr_t)trail_imgs | (uintptr_t)imgs) & 15) == 0), MGF_EINVAL, "keep_improvements: buffers must be 16-byte aligned");
    const int64_t want = (numel / 4 + 255) / 256 + 1;
    const int blocks = (int)(want < 512 ? want : 512);
    hipLaunchKernelGGL(keep_improvements_kernel, dim3(blocks, batch), dim3(256), 0, (hipStream_t)stream, trail_imgs, imgs, numel, take_slot);
    MGF_CHECK_LAUNCH("keep_improvements");
    return MGF_OK;
}

extern "C" int mgf_to_uint8_hwc(uint8_t* out, const float* img, int32_t c, int32_t h, int32_t w, mgf_stream_t stream) {
    MGF_REQUIRE(out && img && c >= 1 && h >= 1 && w >= 1, MGF_EINVAL, "to_uint8: bad arguments");
    const int64_t total = (int64_t)c * h * w;
    hipLaunchKernelGGL(to_uint8_kernel, dim3(mgf_stream_grid(total, 256, 4)), dim3(256), 0, (hipStream_t)stream, out, img, c, h, w);
    MGF_CHECK_LAUNCH("to_uint8");
    return MGF_OK;
}

// ---- DSSIM of the uint8 images (`dssim`, 1024_example_SSIM.py:115-117 = lpips/__init__.py:54-55: (1 - compare_ssim(p0, p1, data_range=255,
// multichannel=True)) / 2, skimage's defaults: 7x7 uniform window, sample covariance, K1 = 0.01, K2 = 0.03, the window positions that lie
// whole inside the image, mean over positions then over channels).  Both images are quantised like the image the drivers save
// (misc.to_pil:115-116: rint(x * 127.5 + 127.5) clipped to 0..255), so the five window sums (x, y, xx, yy, xy over 49 pixels) are exact
// integers (< 2^22); the SSIM quotient and every sum above it are float64 in a fixed order.
// grid = (tiles, c, n); a workgroup owns 32 x 32 window positions of one channel plane and reads the 38 x 38 pixels under them.
constexpr int DS_T = 32, DS_W = 7, DS_R = DS_T + DS_W - 1;
__device__ __forceinline__ int ds_quant(float v) {
    v = rintf(__fadd_rn(__fmul_rn(v, 127.5f), 127.5f));
    return (int)fminf(fmaxf(v, 0.f), 255.f);
}
__global__ __launch_bounds__(256) void dssim_partial_kernel(double* part, const float* img, const float* tgt, int h, int w, int64_t t_stride,
                                                            int tiles_x, double c1, double c2) {
    __shared__ int xs[DS_R][DS_R + 1], ys[DS_R][DS_R + 1];
    __shared__ int hs[5][DS_R][DS_T + 1];
    __shared__ double red[256];
    const int tile = blockIdx.x, ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const int c = gridDim.y;
    const int64_t plane = (int64_t)h * w;
    const float* a = img + ((int64_t)blockIdx.z * c + blockIdx.y) * plane;
    const float* b = tgt + (int64_t)blockIdx.z * t_stride + (int64_t)blockIdx.y * plane;
    const int r0 = ty * DS_T, q0 = tx * DS_T;
    for (int i = threadIdx.x; i < DS_R * DS_R; i += 256) {
        const int r = i / DS_R, q = i - r * DS_R;
        const int rr = r0 + r, qq = q0 + q;
        int xv = 0, yv = 0;
        if (rr < h && qq < w) { xv = ds_quant(a[(int64_t)rr * w + qq]); yv = ds_quant(b[(int64_t)rr * w + qq]); }
        xs[r][q] = xv; ys[r][q] = yv;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < DS_R * DS_T; i += 256) {
        const int r = i / DS_T, q = i - r * DS_T;
        int sx = 0, sy = 0, sxx = 0, syy = 0, sxy = 0;
#pragma unroll
        for (int t = 0; t < DS_W; ++t) {
            const int xv = xs[r][q + t], yv = ys[r][q + t];
            sx += xv; sy += yv; sxx += xv * xv; syy += yv * yv; sxy += xv * yv;
        }
        hs[0][r][q] = sx; hs[1][r][q] = sy; hs[2][r][q] = sxx; hs[3][r][q] = syy; hs[4][r][q] = sxy;
    }
    __syncthreads();
    double acc = 0.0;
    const int vh = h - (DS_W - 1), vw = w - (DS_W - 1);         // window positions per plane
    for (int i = threadIdx.x; i < DS_T * DS_T; i += 256) {
        const int r = i / DS_T, q = i - r * DS_T;
        if (r0 + r >= vh || q0 + q >= vw) continue;
        int s[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            int v = 0;
#pragma unroll
            for (int t = 0; t < DS_W; ++t) v += hs[k][r + t][q];
            s[k] = v;
        }
        constexpr double inv = 1.0 / (DS_W * DS_W), cov = (double)(DS_W * DS_W) / (DS_W * DS_W - 1);
        const double ux = s[0] * inv, uy = s[1] * inv, uxx = s[2] * inv, uyy = s[3] * inv, uxy = s[4] * inv;
        const double vx = cov * (uxx - ux * ux), vy = cov * (uyy - uy * uy), vxy = cov * (uxy - ux * uy);
        const double a1 = 2.0 * ux * uy + c1, a2 = 2.0 * vxy + c2, b1 = ux * ux + uy * uy + c1, b2 = vx + vy + c2;
        acc += (a1 * a2) / (b1 * b2);
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s2 = 128; s2 >= 1; s2 >>= 1) {
        if ((int)threadIdx.x < s2) red[threadIdx.x] += red[threadIdx.x + s2];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[((int64_t)blockIdx.z * c + blockIdx.y) * gridDim.x + tile] = red[0];
}

// grid = n: out = (accumulate ? out : 0) + scale * (1 - mean_c mean_positions S) / 2, the value as float32 like the script's FloatTensor (:159)
__global__ __launch_bounds__(256) void dssim_finish_kernel(float* out, const double* part, int c, int tiles, double positions, float scale,
                                                           int accumulate) {
    __shared__ double red[256];
    double m = 0.0;
    for (int ch = 0; ch < c; ++ch) {
        const double* p = part + ((int64_t)blockIdx.x * c + ch) * tiles;
        double v = 0.0;
        for (int i = threadIdx.x; i < tiles; i += 256) v += p[i];
        red[threadIdx.x] = v;
        __syncthreads();
        for (int s2 = 128; s2 >= 1; s2 >>= 1) {
            if ((int)threadIdx.x < s2) red[threadIdx.x] += red[threadIdx.x + s2];
            __syncthreads();
        }
        m += red[0] / positions;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float v = (float)((1.0 - m / c) / 2.0) * scale;
        out[blockIdx.x] = accumulate ? out[blockIdx.x] + v : v;
    }
}

extern "C" int64_t mgf_dssim_scratch_bytes(int32_t n, int32_t c, int32_t h, int32_t w) {
    if (n < 1 || c < 1 || h < DS_W || w < DS_W) return 0;
    const int64_t tiles = (int64_t)mgf_cdiv(h - (DS_W - 1), DS_T) * mgf_cdiv(w - (DS_W - 1), DS_T);
    return (int64_t)n * c * tiles * (int64_t)sizeof(double);
}

extern "C" int mgf_dssim_u8_f32(float* out, const float* img, const float* target, int32_t n, int32_t c, int32_t h, int32_t w,
                                int64_t t_batch_stride, float data_range, float scale, int32_t accumulate, void* scratch,
                                mgf_stream_t stream) {
    MGF_REQUIRE(out && img && target && scratch && n >= 1 && n <= 65535 && c >= 1 && c <= 65535, MGF_EINVAL, "dssim: bad arguments");
    MGF_REQUIRE(h >= DS_W && w >= DS_W, MGF_EINVAL, "dssim: the image is smaller than the 7x7 window (skimage raises here)");
    MGF_REQUIRE(data_range > 0.f && (uintptr_t)scratch % 8 == 0, MGF_EINVAL, "dssim: data_range must be positive, scratch 8-byte aligned");
    const int tiles_y = (int)mgf_cdiv(h - (DS_W - 1), DS_T), tiles_x = (int)mgf_cdiv(w - (DS_W - 1), DS_T);
    const double c1 = (0.01 * (double)data_range) * (0.01 * (double)data_range), c2 = (0.03 * (double)data_range) * (0.03 * (double)data_range);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(dssim_partial_kernel, dim3(tiles_x * tiles_y, c, n), dim3(256), 0, st, (double*)scratch, img, target, h, w,
                       t_batch_stride, tiles_x, c1, c2);
    hipLaunchKernelGGL(dssim_finish_kernel, dim3(n), dim3(256), 0, st, out, (const double*)scratch, c, tiles_x * tiles_y,
                       (double)(h - (DS_W - 1)) * (double)(w - (DS_W - 1)), scale, accumulate);
    MGF_CHECK_LAUNCH("dssim");
    return MGF_OK;
}
