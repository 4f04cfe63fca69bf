// 1x1 convolution (a plain GEMM per sample) with both MFMA operands loaded straight into registers: no LDS, no barrier.
// Contract: include/mgf.h (mgf_conv1x1_f32).  Replaces the tap-list kernel for the un-modulated 1x1 layers: the resnet skip
// projections of the synthesis blocks (training/networks.py:1102-1105, conv2d_resample with a 1x1 kernel) and the SqueezeNet Fire
// squeeze / expand1x1 layers of LPIPS (lpips/pretrained_networks.py:7-44).
//
// Why not the tap-list kernel: it stages every input element in LDS and reuses it for `taps x cout` multiplies; with ONE tap the
// staging (an LDS store costs ~8 matrix-pipe cycles per dword, tools/probes/mfma_lds.hip) is as expensive as the arithmetic, and the
// Fire layers (16..64 channels on one side) are HBM streams that its two workgroups per CU cannot keep in flight.  Here
//   y[n, co, p] = epilogue( sum_ci w[ci][co] * x[n, ci, p] ),        p = flattened pixel
// is tiled [32 CB output channels] x [128 pixels] per WAVE: v_mfma_f32_32x32x2_f32 takes A[32 co][2 ci] as one dword per lane
// (lane -> co = lane % 32, ci = lane / 32: two 128-byte rows of the [cin][cout_pad] weight image, L2-resident) and B[2 ci][32 px] as
// one dword per lane -- so a lane loads exactly its own operand slots, 4 pixel blocks per k-step, and 4 CB MFMAs follow from 4 + CB
// loaded dwords (one 16-byte load + CB dwords when the map is 4-pixel aligned: lane -> 4 CONSECUTIVE pixels, the four pixel blocks are
// an interleaving of the tile, and the accumulators of a row leave as one 16-byte store).  Loads run through a ring of register sets (8 groups of one k-step: a
// group is requested seven groups of MFMAs before its use); reads past the last channel fall outside the buffer resources and return 0, so the
// loop has no tail code.
#include "mgf_common.h"
#include <type_traits>
#include <cstdlib>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct PwParams {
    float* y;
    const float* x;           // [n][cin][hw]
    const float* w;           // [cin][cout_pad] (mgf_pack_conv_weights of a 1x1 kernel)
    const float* in_scale;    // [n][cin] or null: per-sample style on the input channels (ToRGB: modulated 1x1, no demodulation)
    int n, cin, hw, cout, cout_pad;
    int px_tiles, co_tiles;   // workgroup tiles per sample / over the output channels
    int xcd_per;              // > 0: XCD-aware work order (all channel tiles of a pixel tile on one XCD)
    int64_t y_batch;          // elements between samples of y (y may be a channel slice of a concat buffer; the residual likewise)
    int y_choff;
    mgf_epilogue ep;
    int has_ep;
};

// Ring geometry (tools/pw_ring_micro.py, 14 layer shapes at 16 samples, same box, us in all: 4 k-steps x 2 groups -- the double buffer of
// rounds 3-4 -- 2114; 2 x 4: 2033; 2 x 3: 2048; 1 x 8: 2000; 4 x 3 and 2 x 6, whose extra registers cost a resident workgroup: 2623 / 2761).
// The same 48 staging registers cut into 8 groups of one k-step put a load 7 x 256 matrix-pipe cycles ahead of its use instead of 1024:
// FaceNet's 1792 -> 192 layer 160 -> 127 us, fire9's squeeze 71 -> 60.
#ifndef PW_KU
#define PW_KU 1
#endif
#ifndef PW_NR
#define PW_NR 8
#endif
constexpr int PWKU = PW_KU;   // k-steps (of 2 channels) per load group
constexpr int PWNR = PW_NR;   // load groups in the register ring: a group is requested PWNR - 1 groups of MFMAs before its use

// WK = 4: the four waves of a workgroup share ONE tile and split the input channels (layers with too few tiles to fill the chip, where a
// wave's serial walk over K at one load group in flight is the whole run time); their accumulators meet in LDS and wave 0 stores.
// NJ = 32-pixel blocks per wave: 4 (a 128-pixel tile) or -- small maps under split-K, where the launch has too few waves to fill the chip and a wave's
// serial walk over its K share is the whole run time (512 -> 512 at 32^2, one sample: 128 workgroups, 64 k-steps x 4 MFMAs per wave) -- 1: a quarter
// of the matrix work per wave, four times the waves.
template <int CB, int WCO, int WK, bool VEC, int NJ = 4>
__global__ __launch_bounds__(256, CB == 2 ? 2 : (WK > 1 ? 3 : 4)) void pw_conv_kernel(PwParams p) {
    static_assert(WK == 1 || (WK == 4 && WCO == 1), "split-K workgroups put all four waves on one tile");
    static_assert(NJ == 4 || (NJ == 1 && !VEC && WK == 4 && CB == 1), "the one-block tile is the split-K form's, with element loads");
    constexpr int WPX = 4 / (WCO * WK);
    __shared__ float red[WK > 1 ? 3 * 64 * CB * 64 : 1];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    int b_ = blockIdx.x;
    if (p.xcd_per > 0) {
        b_ = (b_ & 7) * p.xcd_per + (b_ >> 3);
        if (b_ >= p.n * p.px_tiles * p.co_tiles) return;
    }
    const int cot = b_ % p.co_tiles; b_ /= p.co_tiles;
    const int pt = b_ % p.px_tiles;
    const int n = b_ / p.px_tiles;
    const int co0 = (cot * WCO + wv % WCO) * 32 * CB;
    const int p0 = (pt * WPX + (WK > 1 ? 0 : wv / WCO)) * (32 * NJ);
    if (p0 >= p.hw || co0 >= p.cout) return;                       // wave-uniform (workgroup-uniform with WK > 1: its barrier is safe)

    // the style multiplies the WEIGHT operand (one dword per lane and k-step, next to the weight's own): w[ci][co] s[n][ci]
    const bool has_s = p.in_scale != nullptr;
    // The channel part of every address rides in the SCALAR offset (no vector add per load).  The range check of a raw buffer access
    // covers vector + scalar offset (tools/buffer_load_probe.hip: offset + soffset >= num_records reads 0, without 32-bit wrap -- a sentinel
    // vector offset stays out of range), so a k-step past cin -- or the second channel of an odd cin's last k-step, whose lanes start one
    // plane / weight row / style value further in -- reads zeros through the ordinary descriptors.
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x + (int64_t)n * p.cin * p.hw), 0, p.cin * p.hw * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.cin * p.cout_pad * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(has_s ? p.in_scale + (int64_t)n * p.cin : p.x), 0, has_s ? p.cin * 4 : 0, 0x00020000);
    const unsigned xo = (unsigned)(half * p.hw + p0 + (VEC ? 4 * l31 : l31)) * 4u;
    const unsigned wo = (unsigned)(half * p.cout_pad + co0 + l31) * 4u;
    const unsigned so = (unsigned)half * 4u;
    const unsigned xstep = (unsigned)p.hw * 8u, wstep = (unsigned)p.cout_pad * 8u;      // one k-step = 2 channels
    unsigned xoj[NJ], woc[CB];
#pragma unroll
    for (int j = 0; j < NJ; ++j) xoj[j] = xo + 128u * j;
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) woc[cb] = wo + 128u * cb;

    f32x16 acc[CB][NJ];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[cb][j][r] = 0.f;

    // S: the style of a k-step's two channels, multiplied into the weight operand when it is CONSUMED: multiplied where it is loaded, every
    // group's loads ended in an s_waitcnt vmcnt(0) and the next group could not be in flight behind the MFMAs
    float Br[PWNR][PWKU][NJ], Ar[PWNR][PWKU][CB], Sr[PWNR][PWKU];
    auto load = [&](float (&B)[PWKU][NJ], float (&A)[PWKU][CB], float (&S)[PWKU], int it) {
#pragma unroll
        for (int ks = 0; ks < PWKU; ++ks) {
            const int kk = it * PWKU + ks;                                   // wave-uniform
            const int xs = (int)((unsigned)kk * xstep), ws = (int)((unsigned)kk * wstep);
            if constexpr (VEC) {
                const float4 v = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rx, xo, xs, 0));
                B[ks][0] = v.x; B[ks][1] = v.y; B[ks][2] = v.z; B[ks][3] = v.w;
            } else {
#pragma unroll
                for (int j = 0; j < NJ; ++j) B[ks][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, xoj[j], xs, 0));
            }
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) A[ks][cb] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rw, woc[cb], ws, 0));
            if (has_s) S[ks] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, so, kk * 8, 0));
        }
    };
    auto mm = [&](const float (&B)[PWKU][NJ], const float (&A)[PWKU][CB], const float (&S)[PWKU]) {
#pragma unroll
        for (int ks = 0; ks < PWKU; ++ks)
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) {
                const float a = has_s ? A[ks][cb] * S[ks] : A[ks][cb];
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[cb][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, B[ks][j], acc[cb][j], 0, 0, 0);
            }
    };
    int nit = (p.cin + 2 * PWKU - 1) / (2 * PWKU), it0 = 0;
    // (a multiple of PWNR groups per wave: the loop below consumes them PWNR at a time, and a remainder would be the next wave's first;
    // groups past cin read zeros)
    if (WK > 1) { const int per = PWNR * ((nit + PWNR * WK - 1) / (PWNR * WK)); it0 = wv * per; nit = it0 + per; }
    // (scheduling fences: left alone, the compiler sinks each load group to just above its first use and waits for it there)
#pragma unroll
    for (int r = 0; r < PWNR - 1; ++r) load(Br[r], Ar[r], Sr[r], it0 + r);
    for (int it = it0; it < nit; it += PWNR) {
#pragma unroll
        for (int r = 0; r < PWNR; ++r) {
            load(Br[(r + PWNR - 1) % PWNR], Ar[(r + PWNR - 1) % PWNR], Sr[(r + PWNR - 1) % PWNR], it + r + PWNR - 1);
            __builtin_amdgcn_sched_barrier(0);
            mm(Br[r], Ar[r], Sr[r]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    if (WK > 1) {
        if (wv > 0) {
#pragma unroll
            for (int cb = 0; cb < CB; ++cb)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) red[(((wv - 1) * CB + cb) * 64 + j * 16 + r) * 64 + lane] = acc[cb][j][r];
        }
        __syncthreads();
        if (wv > 0) return;
#pragma unroll 1
        for (int k = 0; k < 3; ++k)                                 // (not unrolled: 192 LDS reads in flight would spill)
#pragma unroll
            for (int cb = 0; cb < CB; ++cb)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[cb][j][r] += red[((k * CB + cb) * 64 + j * 16 + r) * 64 + lane];
    }
    // accumulator register r of a 32x32 block = output channel (r & 3) + 8 (r >> 2) + 4 half of the block, pixel column l31.
    // Branch-free (with `if (co < cout)`, `if (px < hw)`, `if (residual)` around every element this was > 1000 basic blocks): the operands
    // and the result go through buffer resources -- an absent bias / residual is a zero-size resource (reads 0), a channel row past cout
    // selects the zero-size resource for its loads and stores (wave-uniform, scalar select; needs cout % 8 == 0 so that the two lane
    // halves of a row are valid together, otherwise the per-lane half test rides in the vector offset), a pixel past hw an out-of-range
    // vector offset.
    const bool do_ep = p.has_ep != 0;
    const float slope = !do_ep ? 1.f : (p.ep.act == MGF_ACT_LRELU ? p.ep.alpha : (p.ep.act == MGF_ACT_RELU ? 0.f : 1.f));
    const float gain = do_ep ? p.ep.gain : 1.f;
    const int64_t ybase = (int64_t)n * p.y_batch + (int64_t)p.y_choff * p.hw;
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)(p.y + ybase), 0, p.cout * p.hw * 4, 0x00020000);
    const float* resp = (do_ep && p.ep.residual) ? p.ep.residual + ybase : nullptr;
    const __amdgpu_buffer_rsrc_t rres = __builtin_amdgcn_make_buffer_rsrc((void*)(resp ? resp : p.x), 0, resp ? p.cout * p.hw * 4 : 0, 0x00020000);
    const float* biasp = (do_ep && p.ep.bias) ? p.ep.bias : nullptr;
    const __amdgpu_buffer_rsrc_t rbias = __builtin_amdgcn_make_buffer_rsrc((void*)(biasp ? biasp : p.x), 0, biasp ? p.cout * 4 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rnone = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, 0, 0x00020000);
    const bool rows8 = (p.cout & 7) == 0;             // rows of both lane halves are inside / outside together
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    // per-lane pixel offsets (bytes) inside a channel row; the half's 4 rows ride along
    unsigned pvo[NJ > 1 ? NJ : 4];
    if constexpr (VEC) {
        const int px = p0 + 4 * l31;
        pvo[0] = px < p.hw ? (unsigned)(4 * half * p.hw + px) * 4u : 0xFFFFFFF0u;
        pvo[1] = pvo[2] = pvo[3] = 0xFFFFFFF0u;
    } else {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int px = p0 + l31 + 32 * j;
            pvo[j] = px < p.hw ? (unsigned)(4 * half * p.hw + px) * 4u : 0xFFFFFFF0u;
        }
    }
    // (two instantiations behind one wave-uniform branch: without a residual the loop used to issue its 64 residual loads all the same --
    // against the zero-size resource -- and every store waited for one of them)
    // MGF_ACT_RELU_POST: the ReLU comes AFTER the residual add, y = relu((acc + bias) * gain + residual) -- the residual blocks of
    // InceptionResnetV1 (`out = relu(conv(cat) * scale + x)`); its own instantiation, so the other launches keep their instruction count
    const bool post_relu = do_ep && p.ep.act == MGF_ACT_RELU_POST;
    // 16-byte accesses on rows that are NOT a multiple of four pixels long (FaceNet's 125^2 maps, SqueezeNet's 255^2 / 127^2 / 63^2; round 6):
    // the rows then start at 4-byte alignment only -- a raw buffer access takes that at full width -- and the vector over a row's END would
    // spill into the next channel's row.  Harmless for the loads of x (those accumulator columns are never stored); for the stores and the
    // residual loads the row's LAST pixel tile (wave-uniform) goes through a resource that ends at the row's end, one lane half at a time
    // (the halves write different rows): the range check of a 16-byte access is per dword on this part (tools/probes/buffer_b128_unaligned.hip:
    // the in-range dwords are read / written, the others return 0 / are dropped), so the ragged vector needs no element path.
    const bool rag = VEC && (p.hw & 3) != 0 && p0 + 128 > p.hw;
    auto store_rows = [&](auto res_tag, auto post_tag) {
        constexpr bool RES = decltype(res_tag)::value;
        constexpr bool POST = decltype(post_tag)::value;
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int cor = co0 + 32 * cb + (r & 3) + 8 * (r >> 2);          // row of lane half 0 (wave-uniform); half 1: + 4
                const bool in = rows8 ? cor < p.cout : cor + 4 * half < p.cout;    // (uniform when rows8)
                const bool any = cor < p.cout;                                      // some lane of the row is inside (uniform)
                const __amdgpu_buffer_rsrc_t ryr = any ? ry : rnone, rrr = any ? rres : rnone, rbr = any ? rbias : rnone;
                const int soff = cor * p.hw * 4;
                const unsigned lane_ok = (rows8 || in) ? 0u : 0xFFFFFFF0u;         // (per-lane only for ragged channel counts)
                const float bv = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rbr, (unsigned)(4 * half) * 4u | lane_ok, cor * 4, 0));
                float v[NJ];
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const float t = acc[cb][j][r] + bv;
                    float ts = t * slope, m;                                       // slope <= 1: leaky / plain ReLU / identity = max(t, slope t)
                    asm("v_max_f32 %0, %1, %2" : "=v"(m) : "v"(t), "v"(ts));      // (as an instruction: `fmaxf` adds canonicalising v_max x, x)
                    v[j] = m * gain;
                }
                if constexpr (VEC) { if (rag) {
                    const int px = p0 + 4 * l31;
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int row = cor + 4 * h;                               // (wave-uniform)
                        const bool row_in = row < p.cout;
                        const __amdgpu_buffer_rsrc_t ryh = __builtin_amdgcn_make_buffer_rsrc((void*)(p.y + ybase + (int64_t)(row_in ? row : 0) * p.hw), 0,
                                                                                             row_in ? p.hw * 4 : 0, 0x00020000);
                        const unsigned off = (half == h && px < p.hw) ? (unsigned)px * 4u : 0xFFFFFFF0u;
                        float4 o = make_float4(v[0], v[1], v[2], v[3]);
                        if (RES) {
                            const __amdgpu_buffer_rsrc_t rrh = __builtin_amdgcn_make_buffer_rsrc((void*)(resp + (int64_t)(row_in ? row : 0) * p.hw), 0,
                                                                                                 row_in ? p.hw * 4 : 0, 0x00020000);
                            const float4 q = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rrh, off, 0, 0));
                            o = make_float4(v[0] + q.x, v[1] + q.y, v[2] + q.z, v[3] + q.w);
                            if (POST) o = make_float4(fmaxf(o.x, 0.f), fmaxf(o.y, 0.f), fmaxf(o.z, 0.f), fmaxf(o.w, 0.f));
                        }
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, o), ryh, off, 0, 0);
                    }
                } else {
                    float4 o = make_float4(v[0], v[1], v[2], v[3]);
                    if (RES) {
                        const float4 q = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rrr, pvo[0] | lane_ok, soff, 0));
                        o = make_float4(v[0] + q.x, v[1] + q.y, v[2] + q.z, v[3] + q.w);
                        if (POST) o = make_float4(fmaxf(o.x, 0.f), fmaxf(o.y, 0.f), fmaxf(o.z, 0.f), fmaxf(o.w, 0.f));
                    }
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, o), ryr, pvo[0] | lane_ok, soff, 0);
                } } else {
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        float o = v[j];
                        if (RES) o += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rrr, pvo[j] | lane_ok, soff, 0));
                        if (RES && POST) o = fmaxf(o, 0.f);
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, o), ryr, pvo[j] | lane_ok, soff, 0);
                    }
                }
            }
        }
    };
    if (resp && post_relu) store_rows(std::true_type{}, std::true_type{});
    else if (resp) store_rows(std::true_type{}, std::false_type{});
    else store_rows(std::false_type{}, std::false_type{});
}

// cout <= 4 (ToRGB outside the fused conv_last launch: gradient mode, odd sizes): the layer is a weighted sum of the channel rows,
// 0.1 FLOP per byte -- a streaming VALU kernel, PX pixels per lane, eight channel rows in flight per lane, weights and styles through
// the scalar cache.  (On the GEMM kernel a 32-row tile with 3 live rows and a K loop of 32 ran at 2.8 TB/s: 380 us for 8 x 32 x 1024^2.)
template <bool VEC>
__global__ __launch_bounds__(256) void pw_narrow_kernel(PwParams p) {
    constexpr int PX = VEC ? 4 : 1;
    const int n = blockIdx.y;
    const int64_t px0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * PX;
    if (px0 >= p.hw) return;
    const float* __restrict__ xb = p.x + (int64_t)n * p.cin * p.hw + px0;
    const float* __restrict__ w = p.w;
    const float* __restrict__ sc = p.in_scale ? p.in_scale + (int64_t)n * p.cin : nullptr;
    float acc[4][PX];
#pragma unroll
    for (int o = 0; o < 4; ++o)
#pragma unroll
        for (int e = 0; e < PX; ++e) acc[o][e] = 0.f;
    for (int k0 = 0; k0 < p.cin; k0 += 8) {
        float v[8][PX];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int k = min(k0 + u, p.cin - 1);                    // (rows past the last one re-read it with a zero weight)
            if (VEC) {
                const float4 q = *reinterpret_cast<const float4*>(xb + (int64_t)k * p.hw);
                v[u][0] = q.x; v[u][PX > 1 ? 1 : 0] = q.y; v[u][PX > 2 ? 2 : 0] = q.z; v[u][PX > 3 ? 3 : 0] = q.w;
            } else {
                v[u][0] = xb[(int64_t)k * p.hw];
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int k = k0 + u;
            const bool in = k < p.cin;
            const int kc = in ? k : p.cin - 1;
            const float s = !in ? 0.f : (sc ? sc[kc] : 1.f);
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                const float wv = o < p.cout ? w[(int64_t)kc * p.cout_pad + o] * s : 0.f;
#pragma unroll
                for (int e = 0; e < PX; ++e) acc[o][e] += wv * v[u][e];
            }
        }
    }
    const bool do_ep = p.has_ep != 0;
    const float slope = !do_ep ? 1.f : (p.ep.act == MGF_ACT_LRELU ? p.ep.alpha : (p.ep.act == MGF_ACT_RELU ? 0.f : 1.f));
    const float gain = do_ep ? p.ep.gain : 1.f;
    const int64_t ybase = (int64_t)n * p.y_batch + (int64_t)p.y_choff * p.hw + px0;
    for (int o = 0; o < p.cout; ++o) {
        const float bv = (do_ep && p.ep.bias) ? p.ep.bias[o] : 0.f;
        float r[PX];
#pragma unroll
        for (int e = 0; e < PX; ++e) {
            float t = (o == 0 ? acc[0][e] : o == 1 ? acc[1][e] : o == 2 ? acc[2][e] : acc[3][e]) + bv;
            t = t > 0.f ? t : t * slope;
            r[e] = t * gain;
        }
        float* yo = p.y + ybase + (int64_t)o * p.hw;
        if (VEC) {
            float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
            if (do_ep && p.ep.residual) q = *reinterpret_cast<const float4*>(p.ep.residual + ybase + (int64_t)o * p.hw);
            *reinterpret_cast<float4*>(yo) = make_float4(r[0] + q.x, r[PX > 1 ? 1 : 0] + q.y, r[PX > 2 ? 2 : 0] + q.z, r[PX > 3 ? 3 : 0] + q.w);
        } else {
            yo[0] = r[0] + ((do_ep && p.ep.residual) ? p.ep.residual[ybase + (int64_t)o * p.hw] : 0.f);
        }
    }
}

// cin <= 4 (ToRGB's data gradient: 3 -> 32 channels): every output row is a combination of <= 4 input rows held in registers -- a
// stream of stores, PX pixels per lane.
template <bool VEC>
__global__ __launch_bounds__(256) void pw_few_inputs_kernel(PwParams p) {
    constexpr int PX = VEC ? 4 : 1;
    const int n = blockIdx.y;
    const int64_t px0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * PX;
    if (px0 >= p.hw) return;
    const float* __restrict__ xb = p.x + (int64_t)n * p.cin * p.hw + px0;
    typedef const float __attribute__((address_space(4)))* cfp;
    const cfp w = (cfp)p.w;
    float v[4][PX];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int kc = min(k, p.cin - 1);
        const float s = k < p.cin ? (p.in_scale ? p.in_scale[(int64_t)n * p.cin + kc] : 1.f) : 0.f;
        if (VEC) {
            const float4 q = *reinterpret_cast<const float4*>(xb + (int64_t)kc * p.hw);
            v[k][0] = q.x * s; v[k][PX > 1 ? 1 : 0] = q.y * s; v[k][PX > 2 ? 2 : 0] = q.z * s; v[k][PX > 3 ? 3 : 0] = q.w * s;
        } else {
            v[k][0] = xb[(int64_t)kc * p.hw] * s;
        }
    }
    const bool do_ep = p.has_ep != 0;
    const float slope = !do_ep ? 1.f : (p.ep.act == MGF_ACT_LRELU ? p.ep.alpha : (p.ep.act == MGF_ACT_RELU ? 0.f : 1.f));
    const float gain = do_ep ? p.ep.gain : 1.f;
    const int64_t ybase = (int64_t)n * p.y_batch + (int64_t)p.y_choff * p.hw + px0;
    const int k1 = p.cin > 1 ? p.cout_pad : 0, k2 = p.cin > 2 ? 2 * p.cout_pad : 0, k3 = p.cin > 3 ? 3 * p.cout_pad : 0;
#pragma unroll 4
    for (int o = 0; o < p.cout; ++o) {
        const float w0 = w[o], w1 = w[k1 + o], w2 = w[k2 + o], w3 = w[k3 + o];          // (rows past cin carry zeros in v)
        const float bv = (do_ep && p.ep.bias) ? p.ep.bias[o] : 0.f;
        float r[PX];
#pragma unroll
        for (int e = 0; e < PX; ++e) {
            float t = ((w0 * v[0][e] + w1 * v[1][e]) + w2 * v[2][e]) + w3 * v[3][e] + bv;
            t = t > 0.f ? t : t * slope;
            r[e] = t * gain;
        }
        float* yo = p.y + ybase + (int64_t)o * p.hw;
        if (VEC) {
            float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
            if (do_ep && p.ep.residual) q = *reinterpret_cast<const float4*>(p.ep.residual + ybase + (int64_t)o * p.hw);
            *reinterpret_cast<float4*>(yo) = make_float4(r[0] + q.x, r[PX > 1 ? 1 : 0] + q.y, r[PX > 2 ? 2 : 0] + q.z, r[PX > 3 ? 3 : 0] + q.w);
        } else {
            yo[0] = r[0] + ((do_ep && p.ep.residual) ? p.ep.residual[ybase + (int64_t)o * p.hw] : 0.f);
        }
    }
}

template <int CB, int WCO, int WK = 1>
void pw_launch(const PwParams& p, bool vec, dim3 grid, hipStream_t st) {
    if (vec) hipLaunchKernelGGL((pw_conv_kernel<CB, WCO, WK, true>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((pw_conv_kernel<CB, WCO, WK, false>), grid, dim3(256), 0, st, p);
}
void pw_launch_narrow(const PwParams& p, dim3 grid, hipStream_t st) { hipLaunchKernelGGL((pw_conv_kernel<1, 1, 4, false, 1>), grid, dim3(256), 0, st, p); }

}  // namespace

static int g_pw_forced_cb = 0;

extern "C" int mgf_conv1x1_force_shape(int32_t channel_blocks) {
    MGF_REQUIRE(channel_blocks >= 0 && channel_blocks <= 2, MGF_EINVAL, "conv1x1_force_shape: 0 (auto), 1 or 2 channel blocks per wave (got %d)", channel_blocks);
    g_pw_forced_cb = channel_blocks;
    return MGF_OK;
}

extern "C" int mgf_conv1x1_f32(float* y, const float* x, const float* w, const float* in_scale, int32_t n, int32_t cin, int32_t hw, int32_t cout,
                               int32_t cout_pad, int64_t y_batch, int32_t y_choff, const mgf_epilogue* ep, mgf_stream_t stream) {
    MGF_REQUIRE(y && x && w && n >= 1 && cin >= 1 && hw >= 1 && cout >= 1, MGF_EINVAL, "conv1x1: bad arguments");
    MGF_REQUIRE(cout_pad >= cout && cout_pad % 32 == 0, MGF_EINVAL, "conv1x1: the weight image must be [cin][cout_pad], cout_pad a multiple of 32 (got %d for %d)",
                cout_pad, cout);
    MGF_REQUIRE(y_choff >= 0 && (y_batch == 0 || y_batch >= (int64_t)(y_choff + cout) * hw), MGF_EINVAL, "conv1x1: bad output slice");
    // 32-bit byte offsets inside one sample / the weight image, with room for the reads past the last channel (they must not wrap: a wrapped
    // offset would fetch in-range data where the buffer rule returns zeros).  The furthest read: with split-K (4 waves) every wave's share
    // is rounded up to a multiple of PWNR load groups and the ring runs PWNR - 1 groups ahead -- group index < 4 PWNR ceil(nit / 4 PWNR) +
    // PWNR - 1 <= nit + 5 PWNR, at 2 PWKU channels per group
    constexpr int64_t PW_OVERREAD = 2LL * PWKU * (5 * PWNR + 1);
    MGF_REQUIRE(((int64_t)cin + PW_OVERREAD) * hw * 4 + 4096 <= (int64_t)UINT32_MAX && ((int64_t)cin + PW_OVERREAD) * cout_pad * 4 + 4096 <= (int64_t)UINT32_MAX,
                MGF_ETOOBIG, "conv1x1: one sample / the weight image must stay below 4 GiB (32-bit buffer offsets)");
    if (ep) {
        MGF_REQUIRE(ep->act == 0 || ep->act == MGF_ACT_LINEAR || ep->act == MGF_ACT_LRELU || ep->act == MGF_ACT_RELU || ep->act == MGF_ACT_RELU_POST,
                    MGF_EUNSUPPORTED, "conv1x1: epilogue activation %d unsupported", ep->act);
        MGF_REQUIRE(ep->act != MGF_ACT_RELU_POST || (ep->residual && cout > 4 && cin > 4), MGF_EUNSUPPORTED,
                    "conv1x1: MGF_ACT_RELU_POST (ReLU after the residual add) needs a residual and the GEMM form (more than 4 channels on both sides)");
        MGF_REQUIRE(!ep->noise, MGF_EUNSUPPORTED, "conv1x1: no noise input (the tap-list kernel has it)");
        MGF_REQUIRE(ep->act != MGF_ACT_LRELU || (ep->alpha >= 0.f && ep->alpha <= 1.f), MGF_EUNSUPPORTED,
                    "conv1x1: leaky-ReLU slope %g outside [0, 1] (the epilogue forms max(t, slope t))", (double)ep->alpha);
    }
    PwParams p;
    p.y = y; p.x = x; p.w = w; p.in_scale = in_scale; p.n = n; p.cin = cin; p.hw = hw; p.cout = cout; p.cout_pad = cout_pad;
    p.y_batch = y_batch ? y_batch : (int64_t)cout * hw; p.y_choff = y_choff;
    p.has_ep = ep != nullptr;
    if (ep) { p.ep = *ep; if (p.ep.act == 0) p.ep.act = MGF_ACT_LINEAR; } else { p.ep = mgf_epilogue{}; p.ep.gain = 1.f; p.ep.act = MGF_ACT_LINEAR; }
    // 16-byte accesses.  The streaming kernels for <= 4 channels on one side (plain pointers) want every row of x, y and the residual
    // 16-byte aligned; the GEMM kernel's raw buffer accesses take any 4-byte alignment and any row length (see `rag` in the kernel), so it
    // always runs its vector form -- MGF_PW_VEC=0 (tuning hook) brings the element form back for A/B runs
    const bool vec = hw % 4 == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)y % 16) == 0 && (p.y_batch % 4) == 0 &&
                     (!ep || !ep->residual || ((uintptr_t)ep->residual % 16) == 0);
    static const bool gemm_vec_off = [] { const char* e = mgf_knob("MGF_PW_VEC"); return e && e[0] == '0'; }();
    const bool gemm_vec = vec || !gemm_vec_off;
    static const bool narrow_off = [] { const char* e = mgf_knob("MGF_PW_NARROW"); return e && e[0] == '0'; }();
    if (cout <= 4 && !narrow_off && n <= 65535) {
        hipStream_t st = (hipStream_t)stream;
        mgf_prof_external_begin(st, "pw_narrow_kernel", 2.0 * cin * (double)cout * hw * n,
                                4.0 * ((double)n * cin * hw + (double)cin * cout + (double)n * cout * hw * ((ep && ep->residual) ? 2 : 1)));
        const dim3 grid((unsigned)mgf_cdiv(hw, vec ? 1024 : 256), n);
        if (vec) hipLaunchKernelGGL(pw_narrow_kernel<true>, grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL(pw_narrow_kernel<false>, grid, dim3(256), 0, st, p);
        mgf_prof_external_end(st);
        MGF_CHECK_LAUNCH("conv1x1");
        return MGF_OK;
    }
    if (cin <= 4 && cout > 4 && !narrow_off && n <= 65535) {
        hipStream_t st = (hipStream_t)stream;
        mgf_prof_external_begin(st, "pw_few_inputs_kernel", 2.0 * cin * (double)cout * hw * n,
                                4.0 * ((double)n * cin * hw + (double)cin * cout + (double)n * cout * hw * ((ep && ep->residual) ? 2 : 1)));
        const dim3 grid((unsigned)mgf_cdiv(hw, vec ? 1024 : 256), n);
        if (vec) hipLaunchKernelGGL(pw_few_inputs_kernel<true>, grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL(pw_few_inputs_kernel<false>, grid, dim3(256), 0, st, p);
        mgf_prof_external_end(st);
        MGF_CHECK_LAUNCH("conv1x1");
        return MGF_OK;
    }
    // Shape: ONE 32-channel block per wave (64 accumulators, 4 workgroups per CU).  Two blocks per wave (128 accumulators, 2 workgroups
    // per CU) halve the B-operand loads but were never faster (tools/pw_micro.py, 25 samples, us for 1 / 2 blocks: skip 512->512 at
    // 32^2 142 / 145, 512->256 at 64^2 242 / 266, 256->128 at 128^2 250 / 272, Fire expand 16->64 at 255^2 152 / 192): residency hides
    // the load latency, the operand loads are L1 hits.  MGF_PW_CB = 2 or mgf_conv1x1_force_shape(2) select the wide shape (tuning, tests).
    static const int env_cb = [] { const char* e = mgf_knob("MGF_PW_CB"); return e ? atoi(e) : 0; }();
    // (round 3, after the epilogue lost its dummy residual loads: the wide shape wins where K is at most 32 -- the Fire expand layers 16 -> 64
    // at 255^2, 163 -> 157 us, and 32 -> 128 at 127^2, 102 -> 88 us at 32 samples -- and still loses from 48 input channels on)
    int cb = (cin <= 32 && cout_pad % 64 == 0 && cout >= 64) ? 2 : 1;
    const int forced = g_pw_forced_cb ? g_pw_forced_cb : env_cb;
    if ((forced == 1) || (forced == 2 && cout_pad % 64 == 0)) cb = forced;
    const int cw = cout_pad / (32 * cb);                           // wave columns needed
    // Few tiles (the 4^2..16^2 skips, most layers at one sample): all four waves on one tile, K split four ways.  Measured
    // (tools/pw_micro.py, 1 / 8 / 25 samples): a gain below ~800 waves (512 -> 512 at 4^2, 25 samples: 39 -> 26 us; 512 -> 64 at 63^2,
    // one sample: 49 -> 24 us), a loss above (512 -> 512 at 16^2, 25 samples = 800 waves: 42 -> 50 us).
    // MGF_PW_SPLITK_WAVES = the wave count below which it is used (tuning; 0 = never).
    static const int64_t splitk_below = [] { const char* e = mgf_knob("MGF_PW_SPLITK_WAVES"); return e ? atoll(e) : (int64_t)768; }();
    const bool splitk = cb == 1 && cin >= 32 && (int64_t)n * mgf_cdiv(hw, 128) * cw < splitk_below;
    // ... and with few of THOSE workgroups (<= 32^2 maps at one sample: the gradient mode's skips and transformer projections) one 32-pixel block per
    // wave: four times the workgroups, a quarter of the serial k walk each.  MGF_PW_NARROW_WGS = the 128-pixel workgroup count below which (0 = never).
    static const int64_t narrow_below = [] { const char* e = mgf_knob("MGF_PW_NARROW_WGS"); return e ? atoll(e) : (int64_t)256; }();
    const bool narrow = splitk && (int64_t)n * mgf_cdiv(hw, 128) * cw < narrow_below;
    const int wco = splitk ? 1 : (cw >= 3 ? 4 : cw);               // 1, 2 or 4 waves of a workgroup side by side over the channels
    p.co_tiles = (cw + wco - 1) / wco;
    p.px_tiles = (int)mgf_cdiv(hw, narrow ? 32 : splitk ? 128 : 128 * (4 / wco));
    int64_t blocks = (int64_t)n * p.px_tiles * p.co_tiles;
    MGF_REQUIRE(blocks <= INT32_MAX - 8, MGF_ETOOBIG, "conv1x1: too many workgroups");
    p.xcd_per = 0;
    if (p.co_tiles > 1 && blocks >= 16) { p.xcd_per = (int)((blocks + 7) / 8); blocks = (int64_t)p.xcd_per * 8; }
    static const char* names[2][3] = {{"pw_conv_kernel<1, 1>", "pw_conv_kernel<1, 2>", "pw_conv_kernel<1, 4>"},
                                      {"pw_conv_kernel<2, 1>", "pw_conv_kernel<2, 2>", "pw_conv_kernel<2, 4>"}};
    hipStream_t st = (hipStream_t)stream;
    mgf_prof_external_begin(st, narrow ? "pw_conv_kernel<1, 1, 4, 1>" : splitk ? "pw_conv_kernel<1, 1, 4>" : names[cb - 1][wco == 4 ? 2 : wco - 1], 2.0 * cin * (double)cout * hw * n,
                            4.0 * ((double)n * cin * hw + (double)cin * cout + (double)n * cout * hw * ((ep && ep->residual) ? 2 : 1)));
    const dim3 grid((unsigned)blocks);
    if (narrow) pw_launch_narrow(p, grid, st);
    else if (splitk) pw_launch<1, 1, 4>(p, gemm_vec, grid, st);
    else if (cb == 2) {
        if (wco == 4) pw_launch<2, 4>(p, gemm_vec, grid, st); else if (wco == 2) pw_launch<2, 2>(p, gemm_vec, grid, st); else pw_launch<2, 1>(p, gemm_vec, grid, st);
    } else {
        if (wco == 4) pw_launch<1, 4>(p, gemm_vec, grid, st); else if (wco == 2) pw_launch<1, 2>(p, gemm_vec, grid, st); else pw_launch<1, 1>(p, gemm_vec, grid, st);
    }
    mgf_prof_external_end(st);
    MGF_CHECK_LAUNCH("conv1x1");
    return MGF_OK;
}
