// LPIPS(SqueezeNet1.1) stem in one pass over the image -- contract: include/mgf.h (mgf_lpips_stem_f32).
//
//   features.0  conv 3->64, 3x3, stride 2, no padding (+ScalingLayer folded into w/b)   lpips/pretrained_networks.py:6-56
//   features.1  ReLU                    -> LPIPS tap 0: unit-normalise over channels, squared difference to the reference
//                                          tap, 1x1 'lin' weights, spatial mean          lpips/networks_basic.py:64-92
//   features.2  MaxPool 3x3 stride 2, ceil_mode                                          -> written out for the fire modules
//
// The 64-channel tap-0 map (67 MB per 1024^2 image) is the largest tensor of the LPIPS branch; unfused it is written by the
// conv, read by the pool and read again (twice) by the distance kernel.  Here it only ever exists in registers: HBM sees the
// image (12.6 MB), the reference tap (read, distance mode) and the pooled map (16.6 MB).
//
// One wave = one work item = a strip of 32 conv columns x up to 35 conv rows, walked top to bottom:
//   * conv: FP32 MFMA 32x32x2, M = 64 output channels (2 tiles), N = 32 pixels of one conv row, K = 27 (+1: the bias rides on
//     the padded k = 27 with a constant-one B operand).  The A operand (weights) lives in 28 registers for the whole kernel;
//     the B operand comes from a 4-row ring of input rows in LDS, stored de-interleaved (even | odd columns) so the stride-2
//     taps read consecutive addresses.
//   * pool: horizontal 3-max with two DPP wave shifts, vertical 3-max carried in registers from row to row -- no LDS, no
//     cross-wave traffic.  Strips overlap by one conv row/two columns (recomputed, 3%/7%), each conv pixel is OWNED by exactly
//     one strip for the distance sum.
//   * distance: deterministic -- per-item partials in `scratch`, one finishing workgroup per sample adds them in index order.
#include "mgf_common.h"
#include <type_traits>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int SLOT = 256;            // floats per input-row slot: 3 channels x (36 even + 36 odd columns)
constexpr int CH_STRIDE = 72, ODD_OFF = 36;
constexpr int ONE_A = 4 * SLOT, ONE_B = ONE_A ^ (2 * SLOT);     // cells holding 1.0f (reached with and without the odd-row flip)
constexpr int LIN_OFF = ONE_B + 4;                               // the 64 'lin' weights (distance mode)
constexpr int LDS_FLOATS = LIN_OFF + 64;
constexpr int TILE_COLS = 30;        // conv columns owned per item (32 computed)
constexpr int PR = 17;               // pooled rows per item (2*PR + 1 conv rows computed)
constexpr int STEM_RED = 16384;      // scratch floats per sample (same slab size as the other loss reductions)

struct StemParams {
    float* pooled; const float* x; const float* w; const float* b;
    float* feat_out; const float* feat_ref; const float* lin; float* scratch;
    int n, h, w_in, ch, cw, ph, pw, tiles_x, strips, xcd_per;
};


// v_max_f32 / v_max3_f32 issued as such.  `fmaxf` lowers to llvm.maxnum, whose IEEE semantics make the compiler put a canonicalising
// `v_max x, x` in front of every operand it cannot prove quiet (MFMA results, DPP moves, loop-carried values): 128 of them per conv row
// next to 96 real maxima.  NaNs do not occur here (finite weights, finite images), and for finite inputs the results are the same.
__device__ __forceinline__ float vmax(float a, float b) { float d; asm("v_max_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
// Horizontal 3-max of four registers in place: x <- max(x[lane], x[lane + 1], x[lane + 2]) as two v_max_f32 whose first operand carries the
// DPP shift (wave_shl:1; bound_ctrl: lane 63 reads 0, the identity for post-ReLU values).  One statement for four values so that every DPP
// operand was written at least three instructions earlier: the two wait states a DPP read of a fresh VALU result needs are then free, and
// the one s_nop in front covers whatever the compiler put before the statement (it does not look inside asm for that hazard).
#define STEM_DPP " wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
__device__ __forceinline__ void hmax3_x4(float& x0, float& x1, float& x2, float& x3) {
    float m0, m1, m2, m3;
    asm volatile("s_nop 1\n\t"
                 "v_max_f32_dpp %4, %0, %0" STEM_DPP "v_max_f32_dpp %5, %1, %1" STEM_DPP
                 "v_max_f32_dpp %6, %2, %2" STEM_DPP "v_max_f32_dpp %7, %3, %3" STEM_DPP
                 "v_max_f32_dpp %0, %4, %4" STEM_DPP "v_max_f32_dpp %1, %5, %5" STEM_DPP
                 "v_max_f32_dpp %2, %6, %6" STEM_DPP "v_max_f32_dpp %3, %7, %7" STEM_DPP
                 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "=&v"(m0), "=&v"(m1), "=&v"(m2), "=&v"(m3));
}

#ifndef STEM_WAVES
#define STEM_WAVES 2
#endif
template <bool FEAT>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(STEM_WAVES, STEM_WAVES))) void lpips_stem_kernel(StemParams p) {
    __shared__ float lds[LDS_FLOATS];
    const int lane = threadIdx.x, l31 = lane & 31, half = lane >> 5;
    // Work order: workgroups are dealt round-robin over the 8 XCDs (each with its own L2); XCD b % 8 walks the contiguous item range
    // [(b % 8) * xcd_per, +xcd_per) with the SAMPLE as the fastest index, so the n candidates of one strip -- which all read the same
    // strip of the target's stored tap (67 MB per target at 1024^2: 25 candidates used to pull it from the Infinity Cache 25 times, more
    // bytes than the kernel's own image + pooled map) -- run on one XCD back to back and share it through that L2.
    int item = (blockIdx.x & 7) * p.xcd_per + (blockIdx.x >> 3);
    if (item >= p.n * p.tiles_x * p.strips) return;
    const int n = item % p.n; item /= p.n;
    const int tx = item % p.tiles_x;
    const int st = item / p.tiles_x;
    const int x0 = tx * TILE_COLS, r0 = st * 2 * PR;
    const int r_end = r0 + 2 * PR < p.ch - 1 ? r0 + 2 * PR : p.ch - 1;        // last conv row of this strip (inclusive)
    const bool last_tx = tx == p.tiles_x - 1, last_st = st == p.strips - 1;

    // ---- constant operands: weights (A), per-lane LDS offsets of the 14 K steps (B), lin weights ----
    float a[2][14];
    int koff[14];
#pragma unroll
    for (int kk = 0; kk < 14; ++kk) {
        const int k = 2 * kk + half;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const int co = m * 32 + l31;
            a[m][kk] = k < 27 ? p.w[co * 27 + k] : (k == 27 ? p.b[co] : 0.f);
        }
        const int ci = k / 9, dy = (k % 9) / 3, dx = k % 3;
        koff[kk] = k < 27 ? dy * SLOT + ci * CH_STRIDE + (dx == 1 ? ODD_OFF : 0) + l31 + (dx == 2 ? 1 : 0) : ONE_A;
    }
    if (lane == 0) { lds[ONE_A] = 1.0f; lds[ONE_B] = 1.0f; }
    // Channel of accumulator register q of M tile m: m*32 + (q&3) + 8*(q>>2) + 4*half.  The first three terms are wave-uniform,
    // so every per-channel address below is a scalar base (SGPR) plus ONE per-lane 32-bit offset that carries 4*half planes.
#define STEM_CH(m, q) ((m) * 32 + ((q) & 3) + 8 * ((q) >> 2))

    // ---- input rows: global -> 4 registers per lane -> LDS ring slot (row & 3), even | odd columns apart ----
    // (buffer addressing: the channel / row part of an address is wave-uniform and rides in the scalar offset; a row outside the image or
    // a column past its end gets an offset beyond num_records and reads 0)
    const int plane = p.h * p.w_in;                                          // host: 3 * h * w_in * 4 B < 2^32
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x + (int64_t)n * 3 * plane), 0, (int)(12u * (unsigned)plane), 0x00020000);
    const unsigned xoff_a = 2 * x0 + lane < p.w_in ? 4u * (2 * x0 + lane) : 0xFFFFFFF0u;
    const unsigned xoff_b = (lane < 3 && 2 * x0 + 64 < p.w_in) ? 4u * (lane * plane + 2 * x0 + 64) : 0xFFFFFFF0u;
    auto load_row = [&](int ir, float (&g)[4]) {
        const bool rv = ir < p.h;
        const int ro = ir * p.w_in;
#pragma unroll
        for (int ci = 0; ci < 3; ++ci)
            g[ci] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, rv ? xoff_a : 0xFFFFFFF0u, (int)(4u * (unsigned)(ci * plane + ro)), 0));
        g[3] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, rv ? xoff_b : 0xFFFFFFF0u, (int)(4u * (unsigned)ro), 0));
    };
    auto store_row = [&](int ir, const float (&g)[4]) {
        const int base = (ir & 3) * SLOT + ((lane & 1) ? ODD_OFF : 0) + (lane >> 1);
#pragma unroll
        for (int ci = 0; ci < 3; ++ci) lds[base + ci * CH_STRIDE] = g[ci];
        if (lane < 3) lds[(ir & 3) * SLOT + lane * CH_STRIDE + 32] = g[3];
    };
    {
        float g0[4], g1[4], g2[4];
        load_row(2 * r0, g0); load_row(2 * r0 + 1, g1); load_row(2 * r0 + 2, g2);
        store_row(2 * r0, g0); store_row(2 * r0 + 1, g1); store_row(2 * r0 + 2, g2);
    }
    __syncthreads();

    const int xc = x0 + l31;
    const bool colvalid = xc < p.cw;
    const bool edge_tile = x0 + 31 >= p.cw;
    const bool colown = colvalid && (l31 < TILE_COLS || last_tx);
    float P[2][16];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int q = 0; q < 16; ++q) P[m][q] = 0.f;
    float dsum = 0.f;
    const int cplane = p.ch * p.cw;                                         // host guarantees 64 * ch * cw < 2^30
    const int pplane = p.ph * p.pw;
    const int lane_feat = 4 * half * cplane + (colvalid ? xc : p.cw - 1);    // per-lane part of a tap-0 address
    const int lane_pool = 4 * half * pplane + (x0 >> 1) + (l31 >> 1);        // per-lane part of a pooled address
    // Buffer addressing: the per-channel plane offsets are wave-uniform, so they ride in the scalar offset operand and a row costs
    // no vector address arithmetic (flat addressing spent one 64-bit vector add per element on it).  64 planes * 4 B < 2^32 (host).
    const __amdgpu_buffer_rsrc_t rfeat = __builtin_amdgcn_make_buffer_rsrc(
        FEAT ? (void*)(p.feat_out + (int64_t)n * 64 * cplane) : (void*)p.feat_ref, 0, (int)(256u * (unsigned)cplane), 0x00020000);
    const __amdgpu_buffer_rsrc_t rpool = __builtin_amdgcn_make_buffer_rsrc((void*)(p.pooled + (int64_t)n * 64 * pplane), 0, (int)(256u * (unsigned)pplane), 0x00020000);
    // Four per-lane offsets (channel & 3 folded in) x eight scalar offsets per M tile, instead of one per-lane offset x 32 scalar ones: the
    // compiler keeps every loop-invariant scalar offset in an SGPR, and 64 of them (tap + pooled) spill into VGPR lanes -- each use then
    // costs a v_readlane and the VALU-writes-SGPR -> VMEM wait states.
    unsigned vfeat[4], vpool[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { vfeat[j] = 4u * (unsigned)(lane_feat + j * cplane); vpool[j] = 4u * (unsigned)(lane_pool + j * pplane); }

    // The 'lin' weights of this lane's 32 channels stay in registers: read from LDS at their use they cost one exposed LDS round trip per four
    // channels (the distance is a serial chain, the compiler places each read right in front of its first use).
    float linr[2][16];
    if (!FEAT) {
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int q = 0; q < 16; ++q) linr[m][q] = p.lin[STEM_CH(m, q) + 4 * half];
    }

    auto emit = [&](const float (&E)[2][16], int py) {   // pooled row py <- E (lanes on even columns; windows x .. x+2 stay inside the 31 good lanes)
        const int px = (x0 >> 1) + (l31 >> 1);
        if (!(l31 & 1) && l31 <= 28 && px < p.pw && py < p.ph) {
            const int o = py * p.pw;                                                       // wave-uniform
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int q = 0; q < 16; ++q)
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, E[m][q]), rpool, vpool[q & 3], (int)(4u * (unsigned)(o + STEM_CH(m, q & ~3) * pplane)), 0);
        }
    };

    // One conv row.  EVEN (rows r0, r0 + 2, ...; r0 is even) is a compile-time tag and the loop below walks row PAIRS: with the parity known,
    // "the even row opens the next window" (P = v) is a renaming instead of 32 register moves on three control-flow paths, and the LDS ring's
    // odd-row flip is a constant.
    auto row = [&](const int r, auto even_tag) {
        constexpr bool EVEN = decltype(even_tag)::value;
        const int rr = r - r0;
        float gA[4], gB[4];
        if (r < r_end) { load_row(2 * r + 3, gA); load_row(2 * r + 4, gB); }
        float t[2][16];
        if (!FEAT) {
            const int tb = r * p.cw;                                                        // wave-uniform
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int q = 0; q < 16; ++q)
                    t[m][q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rfeat, vfeat[q & 3], (int)(4u * (unsigned)(tb + STEM_CH(m, q & ~3) * cplane)), 0));
        }
        // conv row r: rows 2r + dy sit in slots (dy + 2 (r & 1)) & 3
        constexpr int flip = EVEN ? 0 : 2 * SLOT;
        f32x16 acc[2];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[m][q] = 0.f;
        float bv[14];
#pragma unroll
        for (int kk = 0; kk < 14; ++kk) bv[kk] = lds[koff[kk] ^ flip];
#pragma unroll
        for (int kk = 0; kk < 14; ++kk) {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0][kk], bv[kk], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1][kk], bv[kk], acc[1], 0, 0, 0);
        }
        __syncthreads();                         // (one wave per workgroup: orders the LDS reads above before the overwrites below)
        if (r < r_end) { store_row(2 * r + 3, gA); store_row(2 * r + 4, gB); }
        __syncthreads();

        // ReLU (bias is already in), zero outside the map so it is neutral for the pool
        float v[2][16];
        float s = 0.f;
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const float u = vmax(acc[m][q], 0.f);        // (v_max_f32: written as `a > b ? a : b` the compiler must keep a compare + select for NaN's sake -- 2 of this kernel's ~ 14 VALU instructions per element went there)
                v[m][q] = u;
            }
        if (edge_tile) {                         // wave-uniform: only a map's last column tile has lanes outside it
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int q = 0; q < 16; ++q) v[m][q] = colvalid ? v[m][q] : 0.f;
        }
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int q = 0; q < 16; ++q) s += v[m][q] * v[m][q];
        s += __shfl_xor(s, 32, 64);              // the other 32 channels of this pixel live in the other half of the wave
        const float inv = 1.f / (sqrtf(s) + 1e-10f);
        const bool own = colown && (rr < 2 * PR || last_st);
        if (FEAT) {
            if (own) {
                const int fo = r * p.cw;                                                   // wave-uniform
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int q = 0; q < 16; ++q)
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[m][q] * inv), rfeat, vfeat[q & 3], (int)(4u * (unsigned)(fo + STEM_CH(m, q & ~3) * cplane)), 0);
            }
        } else {
            float d = 0.f;
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const float ua = v[m][q] * inv;
                    const float e = ua - t[m][q];
                    d += linr[m][q] * e * e;
                }
            asm volatile("" : "+v"(d));
            dsum += own ? d : 0.f;
        }
        // pool: horizontal 3-max, then the vertical window carried in P
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int q = 0; q < 16; q += 4) hmax3_x4(v[m][q], v[m][q + 1], v[m][q + 2], v[m][q + 3]);
        if (EVEN) {                              // closes window rr/2 - 1, opens window rr/2
            if (rr > 0) {
                float E[2][16];
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int q = 0; q < 16; ++q) E[m][q] = vmax(P[m][q], v[m][q]);
                emit(E, (r0 >> 1) + (rr >> 1) - 1);
            }
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int q = 0; q < 16; ++q) P[m][q] = v[m][q];
        } else {
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int q = 0; q < 16; ++q) P[m][q] = vmax(P[m][q], v[m][q]);
        }
    };
    int r = r0;
    for (; r < r_end; r += 2) {                  // (no branch inside the pair: a conditional odd row brings the moves back at its join)
        row(r, std::true_type{});
        row(r + 1, std::false_type{});
    }
    if (r == r_end) row(r, std::true_type{});
    // ceil_mode: a map with an even number of rows ends on a two-row window
    if ((r_end - r0) & 1) emit(P, (r0 >> 1) + ((r_end - r0) >> 1));

    if (!FEAT) {
        dsum = wave_sum(dsum);
        if (lane == 0) p.scratch[(int64_t)n * STEM_RED + st * p.tiles_x + tx] = dsum;
    }
}

__global__ __launch_bounds__(256) void stem_finish_kernel(float* out, const float* scratch, int nparts, float scale, int accumulate) {
    __shared__ float sm[4];
    const float* sc = scratch + (int64_t)blockIdx.x * STEM_RED;
    float v = 0.f;
    for (int i = threadIdx.x; i < nparts; i += 256) v += sc[i];
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = (accumulate ? out[blockIdx.x] : 0.f) + (sm[0] + sm[1] + sm[2] + sm[3]) * scale;
}

int pool_out(int in) { int o = (in - 3 + 1) / 2 + 1; if ((o - 1) * 2 >= in) --o; return o; }

}  // namespace

extern "C" int mgf_lpips_stem_f32(float* pooled, const float* x, const float* w, const float* b, float* feat_out,
                                  const float* feat_ref, const float* lin, float* out, int32_t n, int32_t h, int32_t w_in,
                                  int32_t accumulate, float* scratch, mgf_stream_t stream) {
    MGF_REQUIRE(pooled && x && w && b && n >= 1 && h >= 7 && w_in >= 7, MGF_EINVAL, "lpips_stem: bad arguments");
    MGF_REQUIRE((feat_out != nullptr) != (feat_ref != nullptr), MGF_EINVAL,
                "lpips_stem: give feat_out (reference mode) or feat_ref (distance mode), not both or neither");
    MGF_REQUIRE(feat_out || (lin && out && scratch), MGF_EINVAL, "lpips_stem: distance mode needs lin, out and scratch");
    StemParams p;
    p.pooled = pooled; p.x = x; p.w = w; p.b = b; p.feat_out = feat_out; p.feat_ref = feat_ref; p.lin = lin; p.scratch = scratch;
    p.n = n; p.h = h; p.w_in = w_in;
    p.ch = (h - 3) / 2 + 1; p.cw = (w_in - 3) / 2 + 1;
    p.ph = pool_out(p.ch); p.pw = pool_out(p.cw);
    p.tiles_x = (int)mgf_cdiv(p.pw, TILE_COLS / 2);
    p.strips = (int)mgf_cdiv(p.ph, PR);
    const int64_t per_sample = (int64_t)p.tiles_x * p.strips;
    MGF_REQUIRE(per_sample <= STEM_RED && per_sample * n <= INT32_MAX, MGF_ETOOBIG, "lpips_stem: image too large (%d x %d)", h, w_in);
    MGF_REQUIRE((int64_t)64 * p.ch * p.cw < (1LL << 30), MGF_ETOOBIG, "lpips_stem: image too large (%d x %d)", h, w_in);
    hipStream_t stq = (hipStream_t)stream;
    p.xcd_per = (int)mgf_cdiv(per_sample * n, 8);
    const dim3 grid((unsigned)(p.xcd_per * 8));
    if (feat_out) {
        hipLaunchKernelGGL((lpips_stem_kernel<true>), grid, dim3(64), 0, stq, p);
    } else {
        hipLaunchKernelGGL((lpips_stem_kernel<false>), grid, dim3(64), 0, stq, p);
        hipLaunchKernelGGL(stem_finish_kernel, dim3(n), dim3(256), 0, stq, out, scratch, (int)per_sample,
                           1.0f / ((float)p.ch * (float)p.cw), accumulate);
    }
    MGF_CHECK_LAUNCH("lpips_stem");
    return MGF_OK;
}
