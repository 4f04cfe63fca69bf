// Tap-list convolution on the FP32 matrix cores of gfx950 (v_mfma_f32_32x32x2_f32, exact f32 fma chains).
// Contract: include/mgf.h (mgf_conv_taps_f32, mgf_pack_conv_weights).  Replaces the cuDNN conv2d /
// conv_transpose2d calls behind conv2d_resample (torch_utils/ops/conv2d_resample.py:21-46,99-139) and the
// per-sample weight modulation of modulated_conv2d (training/networks.py:288-303).
//
// GEMM view (per sample):  D[co][pixel] = sum_{tap, ci} W[tap][ci][co] * X[ci][pixel + offset(tap)]
//   A operand = weights  (M = 32 output channels per MFMA tile, one f32 per lane: A[i = l&31][k = l>>5])
//   B operand = pixels   (N = 32 consecutive pixels of one tile row:            B[k = l>>5][j = l&31])
//   D layout  = column (pixel) on the lane, rows (channels) across 16 registers -> every store instruction writes
//               two 128-byte row segments of the NCHW output.
// Workgroup = 4 waves.  A workgroup owns CO_T = 32*WM output channels x PX = 128*WN pixels ((PX/TW) rows x TW
// columns, TW = 32 for feature maps >= 32 wide).  The K dimension (taps x input channels) is walked in chunks of CK = 8
// input channels; per chunk the workgroup holds in LDS
//   Xs[CK][FH][FW]  the input footprint (halo included, zero padded, style-modulated on the way in) and
//   Ws[T][CK][CO_T] the weight slab,
// and every wave runs T*CK/2 k-steps of WM*WN MFMAs fed by WM + WN conflict-free ds_read_b32.
//
// Pipeline: two LDS buffers; the global loads of chunk i+1 are issued into registers before the MFMA phase of chunk i
// and written to the other buffer after it (one barrier per chunk, loads in flight behind the matrix pipe).  All
// staging addresses are chunk-invariant and precomputed per lane (no integer division in the loop).
//
// Split-K: when a layer has too few output tiles to fill 256 CUs (4x4 .. 64x64 maps with 512 channels) the input
// channels are split over blockIdx.z; partial tiles go to a caller-provided workspace and a second kernel sums them
// in a fixed order (deterministic), applies demodulation + epilogue and writes the output.
//
// MODE 1 is the stride-2 transposed convolution: the 9 taps feed 4 output-parity accumulator sets (compile-time
// tap -> parity map), so it runs at the transposed conv's own FLOP count and writes the parity pairs interleaved.
#include "mgf_common.h"
#include <vector>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int CK = 8;          // input channels per K chunk
constexpr int CKB = 16;        // ... of the bf16x3 arithmetic (AR = 1): one v_mfma_f32_32x32x16_bf16 covers 16 channels
// per-lane register staging slots of the pipelined kernels: XS floats of the input footprint, WS float4 of the weight slab
template <int WM, int WN> struct Slots {
    // 16x32 tile: 8*18*34 = 4896 floats; 12x32: 3808; 8x32: 2720; 4x32: 1632 -- and 8*9*65 = 4680 for the 4x32 tile of a STRIDE-2 3x3
    // conv (the data gradient of the up-sampling layers), which makes the 128-lane tile's slot count 19 instead of 7
    static constexpr int XS = WN == 4 ? 20 : (WN == 3 ? 15 : (WN == 1 ? 19 : 11));
    static constexpr int WS = WM == 2 ? 5 : 3;        // 9*8*64/4 = 1152 float4; 9*8*32/4 = 576
};

// table of ones used as the modulation vector when a launch has no in_scale (keeps the staging code branch-free)
struct Ones { float v[2048]; constexpr Ones() : v() { for (int i = 0; i < 2048; ++i) v[i] = 1.0f; } };
__device__ const Ones g_ones = Ones();

struct ConvParams {
    float* y;
    const float* x;
    const float* wp;
    const float* in_scale;
    const float* out_scale;
    mgf_conv_desc d;
    mgf_epilogue ep;
    int has_ep;
    int tw, rows;         // pixel tile = rows x tw lanes (rows * tw <= PX; lanes beyond it idle); tw = 32 for most layers
    int tw_magic;         // ceil(65536 / tw): pix / tw == (pix * tw_magic) >> 16 for pix < 512, tw <= 64
    int tiles_x, tiles_y;
    int fh, fw;           // LDS footprint of a pixel tile
    int dy_min, dx_min;
    int co_tiles;
    int ksplit;           // number of K slices (1 = direct)
    int chunks_per_split;
    int off32_ok;         // one sample of y (and of the split-K slab) spans < 2^30 elements: 32-bit store offsets are safe
    float* partial;       // [ksplit][n][cout][out_h][y_pitch] when ksplit > 1
    int fixed_geo;        // transposed conv on its usual tile (8 x 32 quads: footprint 9 x 33, canonical taps): instantiation NTP = 10
    const void* wb;       // AR = 1: the weights split into two bf16 terms, [cin / 16][term 2][tap 9][half 2][cout_pad][8 channels]
    int xcd_per;          // > 0: XCD-aware item order -- workgroup b (dispatched round-robin to XCD b % 8) walks the contiguous
                          // item range [(b % 8) * xcd_per, +xcd_per): neighbouring tiles and the co-tiles of one pixel tile share
                          // one XCD's L2 instead of being re-fetched by eight of them
};

__device__ __forceinline__ float epi(const mgf_epilogue& ep, float v, int n, int co, int oy, int ox, int out_h, int out_w,
                                     int64_t yoff) {
    if (ep.noise) {
        float ns = ep.noise_strength ? *ep.noise_strength : 1.0f;
        int nn = ep.noise_n > 1 ? n : 0;
        v += ep.noise[((int64_t)nn * out_h + oy) * out_w + ox] * ns;
    }
    if (ep.bias) v += ep.bias[co];
    if (ep.act == MGF_ACT_LRELU) v = v > 0.f ? v : v * ep.alpha;
    else if (ep.act == MGF_ACT_RELU) v = v > 0.f ? v : 0.f;
    v *= ep.gain;
    if (ep.residual) v += ep.residual[yoff];
    return v;
}

// parity group of tap t (t = kh*3 + kw) for the stride-2 transposed conv: kh (kw) == 1 feeds odd rows (cols)
__host__ __device__ constexpr int tconv_group(int t) { return ((t / 3) == 1 ? 2 : 0) + ((t % 3) == 1 ? 1 : 0); }

struct TileCtx { int n, ks, co0, ty0, tx0, c_begin, c_end; };

// NTP = compile-time tap count (9 = 3x3, 1 = 1x1; 0 = any count read from the descriptor at run time; 10 = the 9 taps of the transposed
// conv on its usual 8 x 32-quad tile -- footprint 9 x 33, canonical tap offsets -- whose operand reads take compile-time LDS offsets;
// 11 = likewise the 3x3 stride-2 conv on its 4 x 32 tile -- footprint 9 x 65, row-major taps: the data gradient of an up-sampling layer).
// Work items = (sample, K slice, pixel tile, channel tile), channel tile fastest.  PIPE kernels are PERSISTENT: a 1-D grid of
// at most (workgroups per CU) x 256 workgroups walks the item list with stride gridDim.x, and the register/LDS pipeline runs
// ACROSS item boundaries -- the first chunk of the next tile is prefetched behind the last MFMA phase of the current one and
// the epilogue's stores drain while the next tile computes -- so HBM traffic and matrix work overlap chip-wide instead of
// alternating in lock-step bursts.  Non-PIPE kernels take one item per workgroup and stage synchronously.
//
// AR = 1: the "bf16x3" arithmetic (opt-in engine mode, NOT the reference's: tools/probes/bf16x3_gemm.hip has its error and rate).  Every f32
// operand is split into two bf16 terms a = a1 + a2 (a1 = bf16(a), a2 = bf16(a - a1): 16 significant bits) and a product becomes three
// v_mfma_f32_32x32x16_bf16 (a2 b1 + a1 b2 + a1 b1, f32 accumulation): 3 x 32 matrix-pipe cycles per 16 channels instead of 8 x 64.  The
// weights arrive split (a checkpoint constant, p.wb); the activations are split where they are staged -- once per footprint element and
// workgroup, with the style folded into them (w (s x) instead of (w s) x) -- and both sit in LDS as 16-byte vectors of 8 channels, the
// fragment of one lane half: Xs[term][half][footprint pixel], Ws[term][tap][half][channel].  Tile walk, persistence, split-K and epilogue
// are the f32 kernel's.
// (A form with wave roles -- 8 waves, one workgroup per CU, waves 4 .. 7 only loading two chunks ahead, waves 0 .. 3 only running matrix
// phases and epilogues -- was built and measured: 816 against 834 iterations/s for this form.  The launch is bound by HBM traffic whose
// load-heavy and store-heavy phases alternate per CU, not by exposed load latency; two independent workgroups per CU overlap them better.)
template <int WM, int WN, int MODE, bool PIPE, int NTP, int AR = 0>
__global__ __launch_bounds__(256, 2) void conv_taps_kernel(ConvParams p) {
    static_assert(AR == 0 || (PIPE && (NTP == 9 || NTP == 10)), "bf16x3: the pipelined 3x3 / transposed-conv instantiations only");
    constexpr bool FX = NTP == 10 || NTP == 11;      // fixed geometry: transposed conv (10), stride-2 conv (11)
    constexpr int FXW = NTP == 10 ? 33 : 65;         // footprint width (height 9 in both)
    constexpr int NT = FX ? 9 : NTP;
    constexpr int CO_T = 32 * WM, PX = 128 * WN, NG = MODE == 1 ? 4 : 1;
    constexpr int CKK = AR ? CKB : CK;               // input channels per K chunk
    extern __shared__ float lds[];
    const mgf_conv_desc& d = p.d;
    const int T = NT > 0 ? NT : d.ntaps;
    int toffs[MGF_MAX_TAPS];                         // LDS offset of each tap inside the footprint (wave-uniform)
#pragma unroll
    for (int t = 0; t < MGF_MAX_TAPS; ++t) toffs[t] = t < T ? (d.dy[t] - p.dy_min) * p.fw + (d.dx[t] - p.dx_min) : 0;
    const int chs = p.fh * p.fw;                 // LDS channel stride of Xs
    const int xs_floats = CKK * chs;             // (AR: 2 terms x 2 halves x chs vectors of 16 bytes = 16 chs floats as well)
    // pipelined mode pads both LDS regions to whole staging slots (branch-free register -> LDS copies)
    const int xs_region = AR ? xs_floats : (PIPE ? Slots<WM, WN>::XS * 256 : xs_floats);
    const int buf_floats = AR ? xs_region + 144 * CO_T : (PIPE ? xs_region + Slots<WM, WN>::WS * 1024 : xs_floats + T * CK * CO_T);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int l31_ = l31, half_ = half;
    const int TW = p.tw;
    const int rows = p.rows;
    const int ptiles = p.tiles_x * p.tiles_y;
    const int total_items = ptiles * p.co_tiles * d.n * p.ksplit;
    const int plane = d.in_h * d.in_w;
    const int wrow4 = CO_T / 4;                       // float4 per weight row
    const int wsz4 = T * CK * wrow4;

    auto decode = [&](int w, TileCtx& c) {
        const int cot = w % p.co_tiles;
        int r = w / p.co_tiles;
        const int pt = r % ptiles;
        r /= ptiles;
        c.ks = r % p.ksplit;
        c.n = r / p.ksplit;
        c.co0 = cot * CO_T;
        c.ty0 = (pt / p.tiles_x) * rows;
        c.tx0 = (pt % p.tiles_x) * TW;
        c.c_begin = c.ks * p.chunks_per_split * CKK;
        c.c_end = c.c_begin + p.chunks_per_split * CKK;
        if (c.c_end > d.cin) c.c_end = d.cin;
        // The item index is wave-uniform, but its divisions by run-time values are expanded on the vector ALU and everything derived from
        // them would then live in VGPRs -- loop bounds compared per lane, and a buffer descriptor built from them costs a
        // readfirstlane "waterfall" loop around EVERY load.  Pin the decoded fields to scalar registers here.
        c.n = __builtin_amdgcn_readfirstlane(c.n); c.ks = __builtin_amdgcn_readfirstlane(c.ks); c.co0 = __builtin_amdgcn_readfirstlane(c.co0);
        c.ty0 = __builtin_amdgcn_readfirstlane(c.ty0); c.tx0 = __builtin_amdgcn_readfirstlane(c.tx0);
        c.c_begin = __builtin_amdgcn_readfirstlane(c.c_begin); c.c_end = __builtin_amdgcn_readfirstlane(c.c_end);
    };

    // per-lane LDS base offset of each of this wave's WN pixel groups (tile independent)
    int pbase[WN];
#pragma unroll
    for (int g = 0; g < WN; ++g) {
        const int pix = (wave * WN + g) * 32 + l31;
        const int pr = (pix * p.tw_magic) >> 16, pc = pix - pr * TW;
        pbase[g] = pr < rows ? pr * d.istride * p.fw + pc * d.istride : 0;      // idle lanes read a valid LDS address
    }

    f32x16 acc[NG][WM][WN];
    auto zero_acc = [&]() {
#pragma unroll
        for (int q = 0; q < NG; ++q)
#pragma unroll
            for (int m = 0; m < WM; ++m)
#pragma unroll
                for (int g = 0; g < WN; ++g)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[q][m][g][r] = 0.f;
    };
    zero_acc();

    // ---- staging state of the tile being LOADED (pipelined path): chunk-invariant per-lane slots ----
    // X slot j covers LDS element i = tid + 256*j -> (ch, r, q); xoff = offset inside the chunk's channel block, -1 = zero
    // padding (also the surplus slots past the footprint, which land in the padded LDS tail).  W slot j covers float4
    // i = tid + 256*j -> (t, ch, c4).
    constexpr int XS = Slots<WM, WN>::XS, WS = Slots<WM, WN>::WS;
    // Buffer addressing (round 3): the chunk's channel / weight-row offset is wave-uniform and rides in the scalar offset operand, the
    // per-lane part is ONE 32-bit byte offset fixed for the tile, and a padding slot carries an offset past num_records -- the load returns
    // 0 by itself.  Flat addressing spent ~ 50 vector instructions per chunk here (64-bit address adds, clamps, the zero-select at store
    // time), and the modulation read through a may-be-either pointer became a FLAT load, which ties every LDS wait to global memory.
    unsigned xoff[XS];                               // byte offset inside the chunk's channel block, 0xFFFFFFF0 = zero padding
    int woff[WS], wch[WS];
    __amdgpu_buffer_rsrc_t rx_l = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw_l = __builtin_amdgcn_make_buffer_rsrc((void*)p.wp, 0, -1, 0x00020000);
    // no modulation: multiply by a table of ones (branch-free staging)
    __amdgpu_buffer_rsrc_t rs_l = __builtin_amdgcn_make_buffer_rsrc((void*)g_ones.v, 0, -1, 0x00020000);
    const int sc_step = p.in_scale ? 4 : 0;          // bytes per channel in the modulation vector (the ones table is read at [0, CK))
    auto setup_slots = [&](const TileCtx& c) {
        rx_l = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x + (int64_t)c.n * d.cin * plane), 0, (int)(4u * (unsigned)(d.cin * plane)), 0x00020000);
        if (p.in_scale) rs_l = __builtin_amdgcn_make_buffer_rsrc((void*)(p.in_scale + (int64_t)c.n * d.cin), 0, -1, 0x00020000);
#pragma unroll
        for (int j = 0; j < WS; ++j) {
            int i = tid + 256 * j;
            if (i >= wsz4) i = wsz4 - 1;              // surplus slots re-read the last row into the padded LDS tail
            const int c4 = i % wrow4;
            const int rest = i / wrow4;
            wch[j] = rest % CK;
            woff[j] = ((rest / CK) * d.cin + wch[j]) * d.cout_pad + c.co0 + c4 * 4;
        }
        // (ch, r, q) of slot 0 by division once, then advanced by 256 elements per slot with carries (no division per slot)
        const int iy0 = c.ty0 * d.istride + p.dy_min, ix0 = c.tx0 * d.istride + p.dx_min;
        int ch = tid / chs;
        int r = (tid - ch * chs) / p.fw;
        int q = tid - ch * chs - r * p.fw;
        const int dr = 256 / p.fw, dq = 256 - dr * p.fw;
#pragma unroll
        for (int j = 0; j < XS; ++j) {
            const int i = tid + 256 * j;
            const int iy = iy0 + r, ix = ix0 + q;
            xoff[j] = (i < xs_floats && iy >= 0 && iy < d.in_h && ix >= 0 && ix < d.in_w) ? 4u * (unsigned)(ch * plane + iy * d.in_w + ix) : 0xFFFFFFF0u;
            q += dq; r += dr;
            if (q >= p.fw) { q -= p.fw; ++r; }
            if (r >= p.fh) { r -= p.fh; ++ch; }
            if (r >= p.fh) { r -= p.fh; ++ch; }        // 256 / fw + 1 < 2 * fh for every tile geometry the host picks
        }
    };
    float xr[XS];
    float4 wr[WS];
    float wsc[WS];
    // load_chunk issues ONLY loads (padding slots carry an out-of-range offset and arrive as zeros), so no
    // instruction that consumes a loaded value -- and hence no s_waitcnt vmcnt -- sits between the loads and the MFMA phase.
    // The style modulation s[ci] is applied to the weight rows (the reference's w * s, networks.py:289) at store time.
    auto load_chunk = [&](int c0) {
        const int xso = (int)(4u * (unsigned)(c0 * plane)), wso = (int)(4u * (unsigned)(c0 * d.cout_pad)), sso = c0 * sc_step;
#pragma unroll
        for (int j = 0; j < XS; ++j) xr[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx_l, xoff[j], xso, 0));
#pragma unroll
        for (int j = 0; j < WS; ++j) {
            wr[j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rw_l, 4u * (unsigned)woff[j], wso, 0));
            wsc[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_l, 4u * (unsigned)wch[j], sso, 0));
        }
    };
    auto store_chunk = [&](float* buf) {              // registers -> LDS, branch-free (padded regions arrive as zeros)
#pragma unroll
        for (int j = 0; j < XS; ++j) buf[tid + 256 * j] = xr[j];
        float* Wd = buf + xs_region;
#pragma unroll
        for (int j = 0; j < WS; ++j) {
            float4 v = wr[j];
            v.x *= wsc[j]; v.y *= wsc[j]; v.z *= wsc[j]; v.w *= wsc[j];
            *reinterpret_cast<float4*>(Wd + (tid + 256 * j) * 4) = v;
        }
    };
    // ---- AR = 1 staging.  Waves 0, 1 stage lane half 0 (channels 0 .. 7 of the chunk), waves 2, 3 half 1: an ITEM is one footprint pixel
    // x 8 channels = 8 dword loads (lanes = consecutive pixels: coalesced), split into the two bf16 terms and written as two 16-byte vectors.
    constexpr int XI = AR ? 3 : 1;                   // items per lane: 128 lanes x 3 >= the 297 / 340 pixels of a footprint (host checks)
    constexpr int WSB = AR ? (36 * CO_T + 255) / 256 : 1;     // 16-byte weight vectors per lane
    const int stid = tid;
    const int h_st = __builtin_amdgcn_readfirstlane(stid >> 7), lp = stid & 127;
    unsigned xoffb[XI];
    unsigned woffb[WSB];
    struct StageSet { float x[XI][8]; u32x4 w[WSB]; float4 s[2]; };
    // tile-invariant parts of the slot addresses, once per kernel (no integer division on the per-tile path): footprint row / column of
    // each item, and the weight vector's offset inside a channel tile's slab
    int itr[XI], itq[XI];
    unsigned wbase0[WSB];
    if (AR) {
#pragma unroll
        for (int j = 0; j < XI; ++j) {
            const int px = lp + 128 * j;
            itr[j] = px / p.fw;
            itq[j] = px - itr[j] * p.fw;
            if (px >= chs) itr[j] = 1 << 20;           // (past the footprint: out of every map)
        }
#pragma unroll
        for (int j = 0; j < WSB; ++j) {
            int i = stid + 256 * j;
            if (i >= 36 * CO_T) i = 36 * CO_T - 1;    // (surplus slots: loaded, never stored)
            const int cc = i % CO_T, rest = i / CO_T; // rest = (term * 9 + tap) * 2 + half
            wbase0[j] = 16u * (unsigned)(rest * d.cout_pad + cc);
        }
    }
    const __amdgpu_buffer_rsrc_t rwb_l = __builtin_amdgcn_make_buffer_rsrc((void*)p.wb, 0, -1, 0x00020000);
    auto setup_slots_b = [&](const TileCtx& c) {
        rx_l = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x + (int64_t)c.n * d.cin * plane), 0, (int)(4u * (unsigned)(d.cin * plane)), 0x00020000);
        if (p.in_scale) rs_l = __builtin_amdgcn_make_buffer_rsrc((void*)(p.in_scale + (int64_t)c.n * d.cin), 0, -1, 0x00020000);
        const int iy0 = c.ty0 * d.istride + p.dy_min, ix0 = c.tx0 * d.istride + p.dx_min;
#pragma unroll
        for (int j = 0; j < XI; ++j) {
            const int iy = iy0 + itr[j], ix = ix0 + itq[j];
            xoffb[j] = (iy >= 0 && iy < d.in_h && ix >= 0 && ix < d.in_w) ? 4u * (unsigned)(8 * h_st * plane + iy * d.in_w + ix) : 0xFFFFFFF0u;
        }
#pragma unroll
        for (int j = 0; j < WSB; ++j) woffb[j] = wbase0[j] + 16u * (unsigned)c.co0;
    };
    auto load_chunk_b = [&](StageSet& S, int c0) {
        const int wso = (int)(16u * (unsigned)((c0 / CKB) * 36 * d.cout_pad)), sso = c0 * sc_step;
#pragma unroll
        for (int j = 0; j < XI; ++j)
#pragma unroll
            for (int k = 0; k < 8; ++k)
                S.x[j][k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx_l, xoffb[j], (int)(4u * (unsigned)((c0 + k) * plane)), 0));
#pragma unroll
        for (int j = 0; j < WSB; ++j) S.w[j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rwb_l, woffb[j], wso, 0));
        // the style of this lane half's 8 channels (the ones table without modulation)
        S.s[0] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs_l, 32u * (unsigned)h_st, sso, 0));
        S.s[1] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs_l, 32u * (unsigned)h_st + 16u, sso, 0));
    };
    auto store_chunk_b = [&](const StageSet& S, float* buf) {
        bf16x8* X1 = reinterpret_cast<bf16x8*>(buf) + h_st * chs;
        bf16x8* X2 = X1 + 2 * chs;
        const float sv[8] = {S.s[0].x, S.s[0].y, S.s[0].z, S.s[0].w, S.s[1].x, S.s[1].y, S.s[1].z, S.s[1].w};
#pragma unroll
        for (int j = 0; j < XI; ++j) {
            const int px = lp + 128 * j;
            bf16x8 t1, t2;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float v = S.x[j][k] * sv[k];
                const __bf16 a1 = (__bf16)v;
                t1[k] = a1;
                t2[k] = (__bf16)(v - (float)a1);                   // (exact difference: a1 keeps v's leading 8 bits)
            }
            if (px < chs) { X1[px] = t1; X2[px] = t2; }
        }
        u32x4* Wd = reinterpret_cast<u32x4*>(buf + xs_region);
#pragma unroll
        for (int j = 0; j < WSB; ++j)
            if (stid + 256 * j < 36 * CO_T) Wd[stid + 256 * j] = S.w[j];
    };
    auto mfma_chunk_b = [&](const float* buf) {
        auto toff = [&](int t) { return NTP == 10 ? ((t / 3) == 2 ? 0 : 33) + ((t % 3) == 2 ? 0 : 1) : toffs[t]; };
        const bf16x8* X1 = reinterpret_cast<const bf16x8*>(buf) + half * chs;
        const bf16x8* X2 = X1 + 2 * chs;
        const bf16x8* W1 = reinterpret_cast<const bf16x8*>(buf + xs_region) + half * CO_T + l31;      // [term][tap][half][CO_T]
        const bf16x8* W2 = W1 + 18 * CO_T;
        bf16x8 fa1[2][WM], fa2[2][WM], fb1[2][WN], fb2[2][WN];
        auto fetch = [&](int t, int set) {
#pragma unroll
            for (int m = 0; m < WM; ++m) { fa1[set][m] = W1[t * 2 * CO_T + m * 32]; fa2[set][m] = W2[t * 2 * CO_T + m * 32]; }
#pragma unroll
            for (int g = 0; g < WN; ++g) { fb1[set][g] = X1[pbase[g] + toff(t)]; fb2[set][g] = X2[pbase[g] + toff(t)]; }
        };
        fetch(0, 0);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            if (t + 1 < 9) fetch(t + 1, (t + 1) & 1);
            const int q = (MODE == 1 && NG == 4) ? tconv_group(t) : 0;
#pragma unroll
            for (int m = 0; m < WM; ++m)
#pragma unroll
                for (int g = 0; g < WN; ++g) {
                    // the two small products first: their sum is formed before it meets the large one
                    acc[q][m][g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa2[t & 1][m], fb1[t & 1][g], acc[q][m][g], 0, 0, 0);
                    acc[q][m][g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa1[t & 1][m], fb2[t & 1][g], acc[q][m][g], 0, 0, 0);
                    acc[q][m][g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa1[t & 1][m], fb1[t & 1][g], acc[q][m][g], 0, 0, 0);
                }
        }
    };
    auto stage_direct = [&](const TileCtx& c, int c0, float* buf) {   // synchronous staging (non-pipelined kernels)
        const float* xn = p.x + (int64_t)c.n * d.cin * plane;
        const float* sc = p.in_scale ? p.in_scale + (int64_t)c.n * d.cin : nullptr;
        const int iy0 = c.ty0 * d.istride + p.dy_min, ix0 = c.tx0 * d.istride + p.dx_min;
        for (int i = tid; i < xs_floats; i += 256) {
            const int ch = i / chs;
            const int rem = i - ch * chs;
            const int r = rem / p.fw, q = rem - r * p.fw;
            const int ci = c0 + ch, iy = iy0 + r, ix = ix0 + q;
            float v = 0.f;
            if (ci < c.c_end && iy >= 0 && iy < d.in_h && ix >= 0 && ix < d.in_w) {
                v = xn[((int64_t)ci * d.in_h + iy) * d.in_w + ix];
                if (sc) v *= sc[ci];
            }
            buf[i] = v;
        }
        float* Wd = buf + xs_region;
        for (int i = tid; i < wsz4; i += 256) {
            const int c4 = i % wrow4;
            const int rest = i / wrow4;
            const int ch = rest % CK, t = rest / CK;
            const int ci = c0 + ch;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ci < c.c_end) v = *reinterpret_cast<const float4*>(p.wp + ((int64_t)t * d.cin + ci) * d.cout_pad + c.co0 + c4 * 4);
            *reinterpret_cast<float4*>(Wd + i * 4) = v;
        }
    };
    // Operand fragments of one tap (CK/2 k-steps) are fetched from LDS into a register set one tap AHEAD of the MFMAs that
    // consume them (two sets, statically indexed after unrolling), so the matrix pipe never waits on an LDS round trip.
    auto mfma_chunk = [&](const float* buf) {
        // FX: footprint geometry and tap offsets are compile-time constants, so every B-operand read is `ds_read_b32 v, base offset:imm`;
        // with run-time geometry each read carries its own v_add_u32 (72 per chunk of 72 MFMAs, ~4 matrix-pipe cycles each)
        const int chs = FX ? 9 * FXW : p.fh * p.fw;
        auto toff = [&](int t) {
            return NTP == 10 ? ((t / 3) == 2 ? 0 : 33) + ((t % 3) == 2 ? 0 : 1) : (NTP == 11 ? (t / 3) * 65 + (t % 3) : toffs[t]);
        };
        const float* Xs = buf + half * chs;                       // lane halves read channels 2kk and 2kk+1
        const float* Ws = buf + xs_region + half * CO_T + l31;
        if (NT > 0) {
            float fa[2][CK / 2][WM], fb[2][CK / 2][WN];
            auto fetch = [&](int t, int set) {
#pragma unroll
                for (int kk = 0; kk < CK / 2; ++kk) {
#pragma unroll
                    for (int m = 0; m < WM; ++m) fa[set][kk][m] = Ws[(t * CK + 2 * kk) * CO_T + m * 32];
#pragma unroll
                    for (int g = 0; g < WN; ++g) fb[set][kk][g] = Xs[2 * kk * chs + pbase[g] + toff(t)];
                }
            };
            fetch(0, 0);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                if (t + 1 < NT) fetch(t + 1, (t + 1) & 1);
                // interleave: one LDS read of the next tap's operands after each MFMA of this tap (instead of all reads first), so a
                // wave's MFMA stream has no read-only gaps for the co-resident wave to fill (+1 % on the large layers)
                if (t + 1 < NT) {
#pragma unroll
                    for (int i = 0; i < (CK / 2) * (WM + WN); ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     // 1 MFMA
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);     // 1 DS read
                    }
                }
                const int q = (MODE == 1 && NG == 4) ? tconv_group(t) : 0;
#pragma unroll
                for (int kk = 0; kk < CK / 2; ++kk)
#pragma unroll
                    for (int m = 0; m < WM; ++m)
#pragma unroll
                        for (int g = 0; g < WN; ++g)
                            acc[q][m][g] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[t & 1][kk][m], fb[t & 1][kk][g], acc[q][m][g], 0, 0, 0);
            }
        } else {
            for (int t = 0; t < T; ++t) {                         // generic tap count (e.g. 2x2, 1x3): not pipelined
                const int toff = (d.dy[t] - p.dy_min) * p.fw + (d.dx[t] - p.dx_min);
#pragma unroll
                for (int kk = 0; kk < CK / 2; ++kk) {
                    float a[WM], b[WN];
#pragma unroll
                    for (int m = 0; m < WM; ++m) a[m] = Ws[(t * CK + 2 * kk) * CO_T + m * 32];
#pragma unroll
                    for (int g = 0; g < WN; ++g) b[g] = Xs[2 * kk * chs + pbase[g] + toff];
#pragma unroll
                    for (int m = 0; m < WM; ++m)
#pragma unroll
                        for (int g = 0; g < WN; ++g)
                            acc[0][m][g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m], b[g], acc[0][m][g], 0, 0, 0);
                }
            }
        }
    };

    // ---- epilogue of one finished tile: demodulate, noise/bias/activation/gain/residual, store (raw partial store when split-K) ----
    const bool partial = p.ksplit > 1;
    const bool do_ep = p.has_ep && !partial;
    const float ep_ns = (do_ep && p.ep.noise) ? (p.ep.noise_strength ? *p.ep.noise_strength : 1.0f) : 0.f;
    auto epilogue = [&](const TileCtx& c) {
        const int n = c.n, co0 = c.co0;
        // opaque copies of the lane coordinates: keeps the compiler from hoisting the epilogue's per-lane address arithmetic
        // out of the persistent tile loop, where it would sit in VGPRs across the MFMA phases
        int half = half_, l31 = l31_;
        asm volatile("" : "+v"(half), "+v"(l31));
        float* yn = partial ? p.partial + ((int64_t)c.ks * d.n + n) * ((int64_t)d.cout * d.out_h * d.y_pitch) : p.y + (int64_t)n * d.y_batch;
        const int64_t y_plane = partial ? (int64_t)d.out_h * d.y_pitch : d.y_plane;
        const int choff = partial ? 0 : d.y_choff;
        const float* osc = (p.out_scale && !partial) ? p.out_scale + (int64_t)n * d.out_scale_stride : nullptr;
        if (MODE == 0 && WM == 1 && d.rgb_out) {
            // Fused ToRGB: this workgroup holds all (<= 32) output channels of its pixels, 16 per lane half.  Each lane forms the
            // partial 1x1 projection of its channels, the two halves are combined with one cross-lane exchange, and only the
            // rgb image is written -- the conv result itself never goes to memory.
            const int rc = d.rgb_channels;
            const float* wr = d.rgb_w + (int64_t)n * rc * d.cout;
            float wv[4][16], osv[16], bv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = (r & 3) + 8 * (r >> 2) + 4 * half;
                const bool ok = co < d.cout;
                osv[r] = (osc && ok) ? osc[co] : 1.0f;
                bv[r] = (do_ep && p.ep.bias && ok) ? p.ep.bias[co] : 0.f;
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) wv[cc][r] = (cc < rc && ok) ? wr[cc * d.cout + co] : 0.f;
            }
#pragma unroll
            for (int g = 0; g < WN; ++g) {
                const int pix = (wave * WN + g) * 32 + l31;
                const int pr = (pix * p.tw_magic) >> 16;
                const int ty = c.ty0 + pr, tx = c.tx0 + (pix - pr * TW);
                const int oy = ty * d.ostride + d.oy[0], ox = tx * d.ostride + d.ox[0];
                const bool ovalid = pr < rows && ty < d.tile_h && tx < d.tile_w && oy < d.out_h && ox < d.out_w;
                float nz = 0.f;
                if (do_ep && p.ep.noise && ovalid) nz = p.ep.noise[((int64_t)(p.ep.noise_n > 1 ? n : 0) * d.out_h + oy) * d.out_w + ox] * ep_ns;
                float sum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = acc[0][0][g][r] * osv[r];
                    if (do_ep) {
                        v += nz;
                        v += bv[r];
                        if (p.ep.act == MGF_ACT_LRELU) v = v > 0.f ? v : v * p.ep.alpha;
                        else if (p.ep.act == MGF_ACT_RELU) v = v > 0.f ? v : 0.f;
                        v *= p.ep.gain;
                    }
#pragma unroll
                    for (int cc = 0; cc < 4; ++cc) sum[cc] += v * wv[cc][r];
                }
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) sum[cc] += __shfl_xor(sum[cc], 32, 64);
                if (half == 0 && ovalid) {
                    for (int cc = 0; cc < rc; ++cc)
                        d.rgb_out[(((int64_t)n * rc + cc) * d.out_h + oy) * d.out_w + ox] = sum[cc] + (d.rgb_bias ? d.rgb_bias[cc] : 0.f);
                }
            }
            return;
        }
        // Fast path (every interior tile of a layer whose channel count fills the tile): no per-element predication and
        // 32-bit per-lane offsets from wave-uniform base pointers, i.e. one VALU add per global access instead of 64-bit
        // multiply/add chains -- the epilogue is instruction-issue bound otherwise.
        const bool fast = p.off32_ok && rows * TW == PX && co0 + CO_T <= d.cout && c.ty0 + rows <= d.tile_h && c.tx0 + TW <= d.tile_w &&
                          (c.ty0 + rows - 1) * d.ostride + (MODE == 1 ? 1 : d.oy[0]) < d.out_h &&
                          (c.tx0 + TW - 1) * d.ostride + (MODE == 1 ? 1 : d.ox[0]) < d.out_w;
        if (fast) {
            const uint32_t plane32 = (uint32_t)y_plane, pitch32 = (uint32_t)d.y_pitch;
            float* ybase = yn + (int64_t)(choff + co0) * y_plane;                                   // wave-uniform
            const float* rbase = (do_ep && p.ep.residual) ? p.ep.residual + (int64_t)n * d.y_batch + (int64_t)(choff + co0) * y_plane : nullptr;
            const float* obase = osc ? osc + co0 : nullptr;
            const float* bbase = (do_ep && p.ep.bias) ? p.ep.bias + co0 : nullptr;
            const float* nbase = (do_ep && p.ep.noise) ? p.ep.noise + (int64_t)(p.ep.noise_n > 1 ? n : 0) * d.out_h * d.out_w : nullptr;
            const uint32_t hoff = (uint32_t)(4 * half) * plane32;
#pragma unroll
            for (int m = 0; m < WM; ++m) {
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    float osv[8], bv[8];
#pragma unroll
                    for (int r8 = 0; r8 < 8; ++r8) {
                        const int cl = m * 32 + ((hh * 8 + r8) & 3) + 8 * ((hh * 8 + r8) >> 2) + 4 * half;   // channel inside the tile
                        osv[r8] = obase ? obase[cl] : 1.0f;
                        bv[r8] = bbase ? bbase[cl] : 0.f;
                    }
#pragma unroll
                    for (int g = 0; g < WN; ++g) {
                        const int pix = (wave * WN + g) * 32 + l31;
                        const uint32_t pr = ((uint32_t)pix * (uint32_t)p.tw_magic) >> 16;
                        const uint32_t ty = c.ty0 + pr, tx = c.tx0 + ((uint32_t)pix - pr * (uint32_t)TW);
                        if (MODE == 0) {
                            const uint32_t oy = ty * d.ostride + d.oy[0], ox = tx * d.ostride + d.ox[0];
                            const uint32_t og = oy * pitch32 + ox + hoff;
                            const float nz = nbase ? nbase[oy * (uint32_t)d.out_w + ox] * ep_ns : 0.f;
                            float rv[8];
#pragma unroll
                            for (int r8 = 0; r8 < 8; ++r8) {
                                const uint32_t cu = m * 32 + ((hh * 8 + r8) & 3) + 8 * ((hh * 8 + r8) >> 2);
                                rv[r8] = rbase ? rbase[og + cu * plane32] : 0.f;
                            }
#pragma unroll
                            for (int r8 = 0; r8 < 8; ++r8) {
                                const uint32_t cu = m * 32 + ((hh * 8 + r8) & 3) + 8 * ((hh * 8 + r8) >> 2);
                                float v = acc[0][m][g][hh * 8 + r8] * osv[r8];
                                if (do_ep) {
                                    v += nz;
                                    v += bv[r8];
                                    if (p.ep.act == MGF_ACT_LRELU) v = v > 0.f ? v : v * p.ep.alpha;
                                    else if (p.ep.act == MGF_ACT_RELU) v = v > 0.f ? v : 0.f;
                                    v = v * p.ep.gain + rv[r8];
                                }
                                ybase[og + cu * plane32] = v;
                            }
                        } else {
                            // parity sets: q = 2*a + b -> row 2*ty + a, cols 2*tx + {0,1} written as an aligned pair
                            const uint32_t og = (2 * ty) * pitch32 + 2 * tx + hoff;
#pragma unroll
                            for (int r8 = 0; r8 < 8; ++r8) {
                                const uint32_t cu = m * 32 + ((hh * 8 + r8) & 3) + 8 * ((hh * 8 + r8) >> 2);
#pragma unroll
                                for (int a2 = 0; a2 < 2; ++a2) {
                                    const float v0 = acc[NG == 4 ? 2 * a2 : 0][m][g][hh * 8 + r8] * osv[r8];
                                    const float v1 = acc[NG == 4 ? 2 * a2 + 1 : 0][m][g][hh * 8 + r8] * osv[r8];
                                    *reinterpret_cast<float2*>(ybase + (og + cu * plane32 + a2 * pitch32)) = make_float2(v0, v1);
                                }
                            }
                        }
                    }
                }
            }
            return;
        }
        if (MODE == 0) {
            // One output pixel per lane, 16 channels per 32-channel tile across the accumulator registers.  Work in half
            // tiles of 8 registers to bound live temporaries: per-channel operands (demodulation, bias) are fetched once per
            // half tile, the residual values of the 8 outputs are gathered together (independent loads in flight), then the
            // 8 results are computed and stored -- no load ever waits behind a store.
            float nzv[WN];
            bool ovalid[WN];
            int64_t offv[WN];
#pragma unroll
            for (int g = 0; g < WN; ++g) {
                const int ty_first = (((wave * WN + g) * 32) * p.tw_magic) >> 16;
                const int pix = (wave * WN + g) * 32 + l31;
                const int pr = (pix * p.tw_magic) >> 16;
                const int ty = c.ty0 + pr, tx = c.tx0 + (pix - pr * TW);
                const int oy = ty * d.ostride + d.oy[0], ox = tx * d.ostride + d.ox[0];
                ovalid[g] = (c.ty0 + ty_first < d.tile_h) && pr < rows && ty < d.tile_h && tx < d.tile_w && oy < d.out_h && ox < d.out_w;
                offv[g] = (int64_t)oy * d.y_pitch + ox;
                nzv[g] = 0.f;
                if (do_ep && p.ep.noise && ovalid[g])
                    nzv[g] = p.ep.noise[((int64_t)(p.ep.noise_n > 1 ? n : 0) * d.out_h + oy) * d.out_w + ox] * ep_ns;
            }
#pragma unroll
            for (int m = 0; m < WM; ++m) {
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    float osv[8], bv[8];
#pragma unroll
                    for (int r8 = 0; r8 < 8; ++r8) {
                        const int r = hh * 8 + r8;
                        const int co = co0 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                        osv[r8] = (osc && co < d.cout) ? osc[co] : 1.0f;
                        bv[r8] = (do_ep && p.ep.bias && co < d.cout) ? p.ep.bias[co] : 0.f;
                    }
#pragma unroll
                    for (int g = 0; g < WN; ++g) {
                        float rv[8];
#pragma unroll
                        for (int r8 = 0; r8 < 8; ++r8) {
                            const int r = hh * 8 + r8;
                            const int co = co0 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                            rv[r8] = (do_ep && p.ep.residual && ovalid[g] && co < d.cout)
                                         ? p.ep.residual[(int64_t)n * d.y_batch + (int64_t)(choff + co) * y_plane + offv[g]] : 0.f;
                        }
#pragma unroll
                        for (int r8 = 0; r8 < 8; ++r8) {
                            const int r = hh * 8 + r8;
                            const int co = co0 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                            float v = acc[0][m][g][r] * osv[r8];
                            if (do_ep) {
                                v += nzv[g];
                                v += bv[r8];
                                if (p.ep.act == MGF_ACT_LRELU) v = v > 0.f ? v : v * p.ep.alpha;
                                else if (p.ep.act == MGF_ACT_RELU) v = v > 0.f ? v : 0.f;
                                v = v * p.ep.gain + rv[r8];
                            }
                            if (ovalid[g] && co < d.cout) yn[(int64_t)(choff + co) * y_plane + offv[g]] = v;
                        }
                    }
                }
            }
            return;
        }
#pragma unroll
        for (int g = 0; g < WN; ++g) {
            const int ty_first = (((wave * WN + g) * 32) * p.tw_magic) >> 16;
            if (c.ty0 + ty_first >= d.tile_h || ty_first >= rows) continue;   // whole pixel group below the image / the tile (wave-uniform)
            const int pix = (wave * WN + g) * 32 + l31;
            const int pr = (pix * p.tw_magic) >> 16;
            const int ty = c.ty0 + pr, tx = c.tx0 + (pix - pr * TW);
            const bool pvalid = pr < rows && ty < d.tile_h && tx < d.tile_w;
#pragma unroll
            for (int m = 0; m < WM; ++m) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = co0 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    if (co >= d.cout || !pvalid) continue;
                    const float os = osc ? osc[co] : 1.0f;
                    float* yc = yn + (int64_t)(choff + co) * y_plane;
                    // parity sets: q = 2*a + b -> row 2*ty + a, cols 2*tx + {0,1} written as a pair
#pragma unroll
                    for (int a2 = 0; a2 < 2; ++a2) {
                        const int oy = 2 * ty + a2, ox = 2 * tx;
                        if (oy >= d.out_h || ox >= d.out_w) continue;
                        const int64_t off = (int64_t)oy * d.y_pitch + ox;
                        const float v0 = acc[NG == 4 ? 2 * a2 : 0][m][g][r] * os;
                        const float v1 = acc[NG == 4 ? 2 * a2 + 1 : 0][m][g][r] * os;
                        if (ox + 1 < d.out_w) *reinterpret_cast<float2*>(yc + off) = make_float2(v0, v1);
                        else yc[off] = v0;
                    }
                }
            }
        }
    };

    if (PIPE) {
        // host guarantees cin % CK == 0 and chunk-aligned K slices: every chunk is full
        int slot, slot_step, slot_end, wbase;
        if (p.xcd_per > 0) {
            slot = (int)(blockIdx.x >> 3); slot_step = (int)(gridDim.x >> 3); wbase = (int)(blockIdx.x & 7) * p.xcd_per;
            slot_end = total_items - wbase < p.xcd_per ? total_items - wbase : p.xcd_per;
        } else { slot = (int)blockIdx.x; slot_step = (int)gridDim.x; wbase = 0; slot_end = total_items; }
        if (slot >= slot_end) return;
        TileCtx cur;
        decode(wbase + slot, cur);
        int b = 0;
        StageSet S0;
        constexpr bool XTILE = !AR && WN <= 2;       // (the three- and four-group shapes have no registers left for it: they spill)
        if (XTILE) {
            // Round 6: ONE chunk stream across the tile boundaries.  During a tile's LAST chunk -- where the staging registers used to idle --
            // the NEXT tile's first chunk is requested (its slot set-up included) and, behind the MFMAs, stored into the free staging buffer
            // like any other chunk, so the epilogue runs with no staging register live and the next tile starts on data that is already in
            // LDS (a tile used to start cold: slot set-up, one HBM round trip, an LDS store and a barrier with nothing to do); the loop over
            // a tile's other chunks has no condition left.  Same-box A/B at 32 samples (tools/conv_micro.py, us, before / after): transposed
            // convs 1384-1391 / 1365-1389, 2464-2474 / 2431, 2476-2480 / 2446-2449, 2598-2613 / 2579, 2998-3002 / 3003-3028 (32^2 .. 512^2
            // input); 3x3 at 32^2 1183-1192 / 1178-1179 -- 1 - 1.5 % on the middle layers, nothing on the last: the other resident workgroup
            // had covered most of the cold start.  (Held across the epilogue instead of stored in front of it, the prefetched chunk spilled:
            // +5 %; with the tile switch as a branch inside one chunk loop the whole loop lost 10 %.)
            setup_slots(cur);
            load_chunk(cur.c_begin);
            store_chunk(lds + b * buf_floats);
            __syncthreads();
            for (;;) {
                const int nch = (cur.c_end - cur.c_begin) / CKK;
                const int nslot = slot + slot_step;
                const bool have_next = nslot < slot_end;
                // every chunk but the last: the next chunk of this tile in flight behind the MFMAs -- no condition left in this loop
                for (int c = 0; c + 1 < nch; ++c) {
                    load_chunk(cur.c_begin + (c + 1) * CK);
                    mfma_chunk(lds + b * buf_floats);
                    store_chunk(lds + (b ^ 1) * buf_floats);
                    __syncthreads();
                    b ^= 1;
                }
                // the last chunk: the next tile's first chunk behind it
                if (have_next) {
                    TileCtx nx;                                                      // (decoded here and again behind the epilogue: nothing of it lives across either)
                    decode(wbase + nslot, nx);
                    setup_slots(nx);
                    load_chunk(nx.c_begin);
                }
                mfma_chunk(lds + b * buf_floats);
                if (have_next) store_chunk(lds + (b ^ 1) * buf_floats);
                __syncthreads();
                b ^= 1;
                epilogue(cur);                                                 // stores drain behind the next tile's work
                zero_acc();
                if (!have_next) break;
                slot = nslot;
                decode(wbase + slot, cur);
            }
        } else {
            for (;;) {
                // first chunk of this tile: global -> registers -> LDS (latency covered by the co-resident workgroup and by the
                // previous tile's stores, which are still draining)
                if (AR) { setup_slots_b(cur); load_chunk_b(S0, cur.c_begin); store_chunk_b(S0, lds + b * buf_floats); }
                else { setup_slots(cur); load_chunk(cur.c_begin); store_chunk(lds + b * buf_floats); }
                __syncthreads();
                const int nch = (cur.c_end - cur.c_begin) / CKK;
                for (int c = 0; c < nch; ++c) {
                    float* curb = lds + b * buf_floats;
                    float* nxtb = lds + (b ^ 1) * buf_floats;
                    const bool more = c + 1 < nch;
                    if (AR) {
                        if (more) load_chunk_b(S0, cur.c_begin + (c + 1) * CKK);
                        mfma_chunk_b(curb);
                        if (more) store_chunk_b(S0, nxtb);
                    } else {
                        if (more) load_chunk(cur.c_begin + (c + 1) * CK);
                        mfma_chunk(curb);
                        if (more) store_chunk(nxtb);
                    }
                    __syncthreads();
                    b ^= 1;
                }
                epilogue(cur);
                zero_acc();
                slot += slot_step;
                if (slot >= slot_end) break;
                decode(wbase + slot, cur);
            }
        }
    } else {
        const int w = p.xcd_per > 0 ? (int)(blockIdx.x & 7) * p.xcd_per + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
        if (w >= total_items || (p.xcd_per > 0 && (int)(blockIdx.x >> 3) >= p.xcd_per)) return;
        TileCtx cur;
        decode(w, cur);
        const int nchunks = (cur.c_end - cur.c_begin + CK - 1) / CK;
        for (int c = 0; c < nchunks; ++c) {
            __syncthreads();
            stage_direct(cur, cur.c_begin + c * CK, lds);
            __syncthreads();
            mfma_chunk(lds);
        }
        epilogue(cur);
    }
}

// Sum the K slices in index order, demodulate, run the epilogue, write y.  One lane per output element.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(ConvParams p) {
    const mgf_conv_desc& d = p.d;
    const int64_t per_n = (int64_t)d.cout * d.out_h * d.out_w;
    const int64_t total = per_n * d.n;
    const int64_t pplane = (int64_t)d.out_h * d.y_pitch;
    const int64_t slice = (int64_t)d.n * d.cout * pplane;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int ox = (int)(i % d.out_w);
        int64_t r = i / d.out_w;
        const int oy = (int)(r % d.out_h); r /= d.out_h;
        const int co = (int)(r % d.cout);
        const int n = (int)(r / d.cout);
        const int64_t poff = ((int64_t)n * d.cout + co) * pplane + (int64_t)oy * d.y_pitch + ox;
        float v = 0.f;
        // fixed summation order (deterministic); loads issued four at a time so their latencies overlap
        const float* pp = p.partial + poff;
        int s = 0;
        for (; s + 8 <= p.ksplit; s += 8) {
            float a[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) a[u] = pp[(int64_t)(s + u) * slice];
#pragma unroll
            for (int u = 0; u < 8; ++u) v += a[u];
        }
        for (; s + 4 <= p.ksplit; s += 4) {
            const float a0 = pp[(int64_t)s * slice], a1 = pp[(int64_t)(s + 1) * slice];
            const float a2 = pp[(int64_t)(s + 2) * slice], a3 = pp[(int64_t)(s + 3) * slice];
            v += a0; v += a1; v += a2; v += a3;
        }
        for (; s < p.ksplit; ++s) v += pp[(int64_t)s * slice];
        if (p.out_scale) v *= p.out_scale[(int64_t)n * d.out_scale_stride + co];
        const int64_t yoff = (int64_t)n * d.y_batch + (int64_t)(d.y_choff + co) * d.y_plane + (int64_t)oy * d.y_pitch + ox;
        if (p.has_ep) v = epi(p.ep, v, n, co, oy, ox, d.out_h, d.out_w, yoff);
        p.y[yoff] = v;
    }
}

// ---- optional per-launch instrumentation (mgf_conv_profile_begin/end): HIP events on the launch stream around the main
// kernel only (not the split-K reduce), so the durations line up with rocprofv3's per-kernel trace ----
struct ProfRec { hipEvent_t e0, e1; int wm, wn, mode, pipe, nt, ksplit; double flops, bytes; const char* name; };
bool g_prof_on = false;
std::vector<ProfRec> g_prof;

struct ProfScope {
    hipStream_t st; bool on;
    ProfScope(hipStream_t s, int wm, int wn, int mode, int pipe, int nt, const ConvParams& p) : st(s), on(g_prof_on) {
        if (!on) return;
        ProfRec r;
        r.name = nullptr;
        r.wm = wm; r.wn = wn; r.mode = mode; r.pipe = pipe; r.nt = nt; r.ksplit = p.ksplit;
        const mgf_conv_desc& d = p.d;
        // algorithmic FLOPs: conv = 2*taps*cin*cout per output pixel; the stride-2 transposed conv is
        // counted at its own cost, 2*9*cin*cout per INPUT pixel
        r.flops = mode == 1 ? 2.0 * 9 * d.cin * (double)d.cout * d.in_h * d.in_w * d.n
                            : 2.0 * d.ntaps * d.cin * (double)d.cout * d.tile_h * d.tile_w * d.n;
        // algorithmic HBM bytes: every input, output (the RGB image when ToRGB is fused: y is not written) and weight once
        r.bytes = 4.0 * ((double)d.n * d.cin * d.in_h * d.in_w + (double)d.ntaps * d.cin * d.cout +
                         (double)d.n * (d.rgb_out ? d.rgb_channels : d.cout) * d.out_h * d.out_w);
        (void)hipEventCreate(&r.e0); (void)hipEventCreate(&r.e1);
        (void)hipEventRecord(r.e0, st);
        g_prof.push_back(r);
    }
    ~ProfScope() { if (on) (void)hipEventRecord(g_prof.back().e1, st); }
};

// the bf16x3 instantiations (AR = 1): persistent, pipelined, 32-channel x 256-pixel tiles; 3x3 stride-1 and the transposed conv
template <int MODE>
int launch_conv_bf(const ConvParams& p_in, hipStream_t st) {
    ConvParams p = p_in;
    constexpr int WM = 1, WN = 2, CO_T = 32;
    const int chs = p.fh * p.fw;
    const size_t lds = 2 * ((size_t)16 * chs + (size_t)144 * CO_T) * sizeof(float);
    if (2 * chs > 768 || p.d.cin % CKB != 0 || lds > 80 * 1024) {
        mgf_set_error("conv_taps(bf16x3): footprint %d x %d / %d input channels not supported", p.fh, p.fw, p.d.cin);
        return MGF_EUNSUPPORTED;
    }
    const int64_t items = (int64_t)p.tiles_x * p.tiles_y * p.co_tiles * p.d.n * p.ksplit;
    const int64_t resident = (int64_t)MGF_NUM_CU * 2;
    dim3 grid((unsigned)(items < resident ? items : resident));
    p.xcd_per = 0;
    if (items >= 16) {
        p.xcd_per = (int)((items + 7) / 8);
        if (grid.x % 8 != 0) grid.x = (grid.x + 7) / 8 * 8;
    }
    auto kern1 = conv_taps_kernel<WM, WN, MODE, true, (MODE == 1 ? 10 : 9), 1>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)kern1, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        if (e != hipSuccess) { mgf_set_error("conv_taps(bf16x3): cannot raise dynamic LDS: %s", hipGetErrorString(e)); return MGF_ELAUNCH; }
        attr_set = true;
    }
    {
        ProfScope ps(st, WM, WN, MODE, 1, MODE == 1 ? 10 : 9, p);
        if (ps.on) g_prof.back().name = MODE == 1 ? "conv_taps_kernel<1, 2, 1, true, 10, 1>" : "conv_taps_kernel<1, 2, 0, true, 9, 1>";
        hipLaunchKernelGGL(kern1, grid, dim3(256), lds, st, p);
    }
    if (p.ksplit > 1) {
        const int64_t total = (int64_t)p.d.n * p.d.cout * p.d.out_h * p.d.out_w;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(mgf_stream_grid(total, 256, 2)), dim3(256), 0, st, p);
    }
    return MGF_OK;
}

template <int WM, int WN, int MODE>
int launch_conv(const ConvParams& p_in, hipStream_t st) {
    ConvParams p = p_in;
    constexpr int CO_T = 32 * WM;
    const size_t buf = ((size_t)CK * p.fh * p.fw + (size_t)p.d.ntaps * CK * CO_T) * sizeof(float);
    const bool pipe = (size_t)CK * p.fh * p.fw <= (size_t)Slots<WM, WN>::XS * 256 &&
                      (size_t)p.d.ntaps * CK * CO_T / 4 <= (size_t)Slots<WM, WN>::WS * 256 &&
                      256 / p.fw + 2 <= 2 * p.fh &&   // slot walk: at most two row wraps per 256-element step
                      p.d.cin % CK == 0;              // the persistent pipeline only moves full chunks
    const size_t lds = pipe ? 2 * ((size_t)Slots<WM, WN>::XS * 256 + (size_t)Slots<WM, WN>::WS * 1024) * sizeof(float) : buf;
    if (lds > 160 * 1024) { mgf_set_error("conv_taps: tile needs %zu bytes of LDS (> 160 KiB)", lds); return MGF_EUNSUPPORTED; }
    const int64_t items = (int64_t)p.tiles_x * p.tiles_y * p.co_tiles * p.d.n * p.ksplit;
    // persistent kernels: as many workgroups as the chip holds at once (2 per CU; 3 for the small-register variant)
    static const char* res_env = mgf_knob("MGF_RESIDENT");      // tuning hook (experiments only): workgroups per CU
    const int64_t resident = (int64_t)MGF_NUM_CU * (res_env ? atoi(res_env) : ((WM == 1 && WN == 2 && MODE == 0) ? 3 : 2));
    dim3 grid((unsigned)(pipe && (p.d.ntaps == 9 || ((p.d.ntaps == 1 || p.d.ntaps == 7 || p.d.ntaps == 3) && MODE == 0)) ? (items < resident ? items : resident) : items));
    // XCD-aware order needs a grid that is a multiple of the 8 XCDs (a grid of ceil(items/8)*8 workgroups otherwise)
    static const char* xcd_env = mgf_knob("MGF_XCD");            // tuning hook (experiments only): 0 disables the XCD-aware order
    p.xcd_per = 0;
    if (!(xcd_env && xcd_env[0] == '0') && items >= 16) {
        p.xcd_per = (int)((items + 7) / 8);
        if (grid.x % 8 != 0) grid.x = (grid.x + 7) / 8 * 8;
    }
    const int nt = p.d.ntaps;
    if (pipe && nt == 9 && MODE == 1 && WN == 2 && p.fixed_geo) {
        ProfScope ps(st, WM, WN, MODE, 1, 10, p);
        hipLaunchKernelGGL((conv_taps_kernel<WM, (MODE == 1 ? 2 : WN), MODE, true, (MODE == 1 ? 10 : 9)>), grid, dim3(256), lds, st, p);
    } else if (pipe && nt == 9 && MODE == 0 && WN == 1 && p.fixed_geo == 2) {
        ProfScope ps(st, WM, WN, MODE, 1, 11, p);
        hipLaunchKernelGGL((conv_taps_kernel<WM, (MODE == 0 ? 1 : WN), MODE, true, (MODE == 0 ? 11 : 9)>), grid, dim3(256), lds, st, p);
    } else if (pipe && nt == 9) {
        ProfScope ps(st, WM, WN, MODE, 1, 9, p);
        hipLaunchKernelGGL((conv_taps_kernel<WM, WN, MODE, true, 9>), grid, dim3(256), lds, st, p);
    } else if (pipe && (nt == 7 || nt == 3) && MODE == 0) {
        // 1x7 / 7x1 and 1x3 / 3x1 layers (InceptionResnetV1's Block17 / Block8): the pipelined, persistent form with the tap loop unrolled
        ProfScope ps(st, WM, WN, MODE, 1, nt, p);
        if (nt == 7) hipLaunchKernelGGL((conv_taps_kernel<WM, WN, MODE == 1 ? 0 : MODE, true, 7>), grid, dim3(256), lds, st, p);
        else hipLaunchKernelGGL((conv_taps_kernel<WM, WN, MODE == 1 ? 0 : MODE, true, 3>), grid, dim3(256), lds, st, p);
    } else if (pipe && nt == 1 && MODE == 0) {
        ProfScope ps(st, WM, WN, MODE, 1, 1, p);
        hipLaunchKernelGGL((conv_taps_kernel<WM, WN, MODE == 1 ? 0 : MODE, true, 1>), grid, dim3(256), lds, st, p);
    } else {
        ProfScope ps(st, WM, WN, MODE, 0, nt == 9 ? 9 : 0, p);
        // un-pipelined staging: large footprints (strided convs, 4x4 maps) and unusual tap counts
        auto kern9 = conv_taps_kernel<WM, WN, MODE, false, 9>;
        auto kern0 = conv_taps_kernel<WM, WN, MODE == 1 ? 0 : MODE, false, 0>;
        const size_t lds1 = buf;
        const void* fn = (nt == 9) ? (const void*)kern9 : (const void*)kern0;
        if (lds1 > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1);
            if (e != hipSuccess) { mgf_set_error("conv_taps: cannot raise dynamic LDS to %zu: %s", lds1, hipGetErrorString(e)); return MGF_ELAUNCH; }
        }
        if (nt == 9) hipLaunchKernelGGL(kern9, grid, dim3(256), lds1, st, p);
        else hipLaunchKernelGGL(kern0, grid, dim3(256), lds1, st, p);
    }
    if (p.ksplit > 1) {
        const int64_t total = (int64_t)p.d.n * p.d.cout * p.d.out_h * p.d.out_w;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(mgf_stream_grid(total, 256, 2)), dim3(256), 0, st, p);
    }
    return MGF_OK;
}

__global__ void pack_weights_kernel(float* wp, float* wsq, const float* w, int cout, int cin, int kh, int kw, int cout_pad,
                                    float gain, int flip) {
    const int T = kh * kw;
    const int64_t total = (int64_t)T * cin * cout_pad;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int co = (int)(i % cout_pad);
        const int64_t r = i / cout_pad;
        const int ci = (int)(r % cin);
        const int t = (int)(r / cin);
        float v = 0.f;
        if (co < cout) {
            int ky = t / kw, kx = t % kw;
            if (flip) { ky = kh - 1 - ky; kx = kw - 1 - kx; }
            v = w[(((int64_t)co * cin + ci) * kh + ky) * kw + kx] * gain;
        }
        wp[i] = v;
    }
    if (wsq) {
        const int64_t tot2 = (int64_t)cout * cin;
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < tot2; i += (int64_t)gridDim.x * blockDim.x) {
            float s = 0.f;
            for (int t = 0; t < T; ++t) { float v = w[i * T + t] * gain; s += v * v; }
            wsq[i] = s;
        }
    }
}

}  // namespace

extern "C" int mgf_conv_profile_begin(void) {
    for (auto& r : g_prof) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
    g_prof.clear();
    g_prof_on = true;
    return MGF_OK;
}

// the other convolution kernels of the library (csrc/wino.hip) report through the same instrumentation
void mgf_prof_external_begin(hipStream_t st, const char* name, double flops, double bytes) {
    if (!g_prof_on) return;
    ProfRec r;
    r.name = name; r.wm = r.wn = r.mode = r.pipe = r.nt = 0; r.ksplit = 1; r.flops = flops; r.bytes = bytes;
    (void)hipEventCreate(&r.e0); (void)hipEventCreate(&r.e1);
    (void)hipEventRecord(r.e0, st);
    g_prof.push_back(r);
}
void mgf_prof_external_end(hipStream_t st) {
    if (g_prof_on && !g_prof.empty()) (void)hipEventRecord(g_prof.back().e1, st);
}

extern "C" int mgf_conv_profile_end(mgf_conv_prof_rec* out, int32_t max_recs) {
    g_prof_on = false;
    int n = 0;
    for (auto& r : g_prof) {
        if (hipEventSynchronize(r.e1) != hipSuccess) { mgf_set_error("conv_profile_end: event sync failed"); return MGF_ELAUNCH; }
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, r.e0, r.e1);
        if (out && n < max_recs) {
            mgf_conv_prof_rec& o = out[n];
            if (r.name) snprintf(o.kernel, sizeof(o.kernel), "%s", r.name);
            else snprintf(o.kernel, sizeof(o.kernel), "conv_taps_kernel<%d, %d, %d, %s, %d, 0>", r.wm, r.wn, r.mode, r.pipe ? "true" : "false", r.nt);
            o.flops = r.flops; o.bytes = r.bytes; o.seconds = ms * 1e-3; o.ksplit = r.ksplit;
        }
        ++n;
        (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1);
    }
    g_prof.clear();
    return n;
}

// Last row and last column of the stride-2 transposed 3x3 conv output t[2h+1][2w+1] (t[2i+kh][2j+kw] += w[kh][kw] x[i][j]): the
// 4*in + 1 positions that do not belong to the in x in grid of 2x2 output quads the MFMA kernel tiles exactly.  The last column only
// sees the taps kw = 2 (input column w-1), the last row only kh = 2 (input row h-1): along its part, output position q = 2i gets
// W[k=0] x[i] + W[k=2] x[i-1] and q = 2i + 1 gets W[k=1] x[i] -- two small GEMMs over the input's border line.
// Round 3: on the matrix cores (the VALU version read 15 LDS words per 12 FMAs, multiplied half of its taps by zero and ran at a tenth of
// what the layer's flops need: 0.94 ms per iteration for 2 % of the transposed convs' work).  A workgroup owns 64 consecutive positions
// (32 even + 32 odd) x 64 output channels; K is walked in chunks of 32 input channels staged in LDS (weights coalesced along cout, the
// border line -- 33 input samples per channel -- with the style folded in).  Waves 0 / 1: the even positions, one 32-channel block each,
// k-step = (channel, tap 0 | tap 2 by lane half); wave 2: the odd positions, both channel blocks, k-step = channel pair; wave 3 only
// stages.  32 MFMAs per wave and chunk in all three.
constexpr int TB_POS = 64, TB_CK = 32, TB_LINE = TB_POS / 2 + 1, TB_LP = TB_LINE + 1;
__global__ __launch_bounds__(256) void tconv_border_kernel(float* t, const float* x, const float* wp, const float* in_scale,
                                                           const float* out_scale, int cin, int h, int w, int cout, int cout_pad,
                                                           int64_t pitch, int64_t plane, int64_t batch, int64_t os_stride, int col_groups) {
    __shared__ float Ws[3][TB_CK][64];
    __shared__ float Xl[TB_CK][TB_LP];                                  // Xl[ci][jj] = s[ci] x[ci][i0 - 1 + jj] along the border line, 0 outside it
    const int tid = threadIdx.x, n = blockIdx.z, co0 = blockIdx.y * 64;
    const int lane = tid & 63, l31 = lane & 31, half = lane >> 5, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool col = (int)blockIdx.x < col_groups;                      // column part (ox = 2w) or row part (oy = 2h)
    const int p0 = (col ? blockIdx.x : blockIdx.x - col_groups) * TB_POS;
    const int npos = col ? 2 * h + 1 : 2 * w;
    const int lim = col ? h : w;
    const int tap0 = col ? 2 : 6, tapstep = col ? 3 : 1;               // packed tap index of k: (kh = k, kw = 2) or (kh = 2, kw = k)
    const float* xn = x + (int64_t)n * cin * h * w;
    const int64_t hw = (int64_t)h * w;
    // staging role of this thread for the line: entries (ci, jj) = 32 x 33 = 1056 over 256 threads; the source offset (or -1) is fixed
    constexpr int XR = (TB_CK * TB_LINE + 255) / 256, WR = 3 * TB_CK * 64 / 256;
    int xsrc[XR];
#pragma unroll
    for (int r = 0; r < XR; ++r) {
        const int e = tid + 256 * r, jj = e % TB_LINE;
        const int i = p0 / 2 - 1 + jj;
        xsrc[r] = (e < TB_CK * TB_LINE && i >= 0 && i < lim) ? (col ? i * w + (w - 1) : (h - 1) * w + i) : -1;
    }
    float wreg[WR], xreg[XR];
    auto request = [&](int c0) {                                       // global -> registers for one K chunk (loads only)
#pragma unroll
        for (int r = 0; r < WR; ++r) {
            const int e = tid + 256 * r, co = e & 63, ci = (e >> 6) % TB_CK, k = e / (64 * TB_CK);
            const int cg = c0 + ci < cin ? c0 + ci : cin - 1, cc = co0 + co < cout_pad ? co0 + co : cout_pad - 1;
            wreg[r] = wp[((int64_t)(tap0 + k * tapstep) * cin + cg) * cout_pad + cc];
        }
#pragma unroll
        for (int r = 0; r < XR; ++r) {
            const int e = tid + 256 * r, ci = e / TB_LINE;
            const int cg = c0 + ci < cin ? c0 + ci : cin - 1;
            xreg[r] = xn[cg * hw + (xsrc[r] >= 0 ? xsrc[r] : 0)] * (in_scale ? in_scale[(int64_t)n * cin + cg] : 1.0f);
        }
    };
    f32x16 acc[2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
    request(0);
    for (int c0 = 0; c0 < cin; c0 += TB_CK) {
        __syncthreads();
#pragma unroll
        for (int r = 0; r < WR; ++r) {
            const int e = tid + 256 * r, co = e & 63, ci = (e >> 6) % TB_CK, k = e / (64 * TB_CK);
            Ws[k][ci][co] = c0 + ci < cin ? wreg[r] : 0.f;
        }
#pragma unroll
        for (int r = 0; r < XR; ++r) {
            const int e = tid + 256 * r, jj = e % TB_LINE, ci = e / TB_LINE;
            if (e < TB_CK * TB_LINE) Xl[ci][jj] = (c0 + ci < cin && xsrc[r] >= 0) ? xreg[r] : 0.f;
        }
        __syncthreads();
        if (c0 + TB_CK < cin) request(c0 + TB_CK);                      // next chunk's loads fly during the MFMAs below
        if (wv < 2) {                                                  // even positions q = p0 + 2 j: tap 0 reads x[i], tap 2 reads x[i - 1]
#pragma unroll 8
            for (int ci = 0; ci < TB_CK; ++ci)
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(Ws[2 * half][ci][wv * 32 + l31], Xl[ci][l31 + 1 - half], acc[0], 0, 0, 0);
        } else if (wv == 2) {                                          // odd positions q = p0 + 2 j + 1: tap 1 reads x[i]
#pragma unroll 8
            for (int cp = 0; cp < TB_CK / 2; ++cp) {
                const int ci = 2 * cp + half;
                const float bv = Xl[ci][l31 + 1];
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(Ws[1][ci][l31], bv, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(Ws[1][ci][32 + l31], bv, acc[1], 0, 0, 0);
            }
        }
    }
    if (wv > 2) return;
    const int q = p0 + 2 * l31 + (wv == 2 ? 1 : 0);
    if (q >= npos) return;
    const int oy = col ? q : 2 * h, ox = col ? 2 * w : q;
    float* tn = t + (int64_t)n * batch + (int64_t)oy * pitch + ox;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        if (m == 1 && wv != 2) break;
        const int cb = wv == 2 ? m : wv;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co0 + cb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (co < cout) tn[(int64_t)co * plane] = acc[m][r] * (out_scale ? out_scale[(int64_t)n * os_stride + co] : 1.0f);
        }
    }
}

extern "C" int mgf_tconv3x3s2_border_f32(float* t, const float* x, const float* wp, const float* in_scale, const float* out_scale,
                                         int32_t n, int32_t cin, int32_t h, int32_t w, int32_t cout, int32_t cout_pad, int64_t t_pitch,
                                         int64_t t_plane, int64_t t_batch, int64_t out_scale_stride, mgf_stream_t stream) {
    MGF_REQUIRE(t && x && wp && n >= 1 && n <= 65535 && cin >= 1 && h >= 1 && w >= 1 && cout >= 1 && cout_pad >= cout, MGF_EINVAL,
                "tconv3x3s2_border: bad arguments");
    MGF_REQUIRE(t_pitch >= 2 * w + 1 && t_plane >= (int64_t)(2 * h + 1) * t_pitch, MGF_EINVAL, "tconv3x3s2_border: output pitch/plane too small");
    const int col_groups = (int)mgf_cdiv(2 * h + 1, TB_POS), row_groups = (int)mgf_cdiv(2 * w, TB_POS);
    hipLaunchKernelGGL(tconv_border_kernel, dim3(col_groups + row_groups, (unsigned)mgf_cdiv(cout, 64), n), dim3(256), 0, (hipStream_t)stream,
                       t, x, wp, in_scale, out_scale, cin, h, w, cout, cout_pad, t_pitch, t_plane, t_batch, out_scale_stride, col_groups);
    MGF_CHECK_LAUNCH("tconv3x3s2_border");
    return MGF_OK;
}

extern "C" int mgf_pack_conv_weights(float* wp, float* wsq, const float* w, int32_t cout, int32_t cin, int32_t kh, int32_t kw,
                                     int32_t cout_pad, float gain, int32_t flip, mgf_stream_t stream) {
    MGF_REQUIRE(wp && w, MGF_EINVAL, "pack_conv_weights: null pointer");
    MGF_REQUIRE(cout >= 1 && cin >= 1 && kh >= 1 && kw >= 1 && kh * kw <= MGF_MAX_TAPS, MGF_EINVAL, "pack_conv_weights: bad shape");
    MGF_REQUIRE(cout_pad >= cout && cout_pad % 32 == 0, MGF_EINVAL, "pack_conv_weights: cout_pad must be a multiple of 32 >= cout");
    const int64_t total = (int64_t)kh * kw * cin * cout_pad;
    hipLaunchKernelGGL(pack_weights_kernel, dim3(mgf_stream_grid(total, 256, 4)), dim3(256), 0, (hipStream_t)stream, wp, wsq, w,
                       cout, cin, kh, kw, cout_pad, gain, flip);
    MGF_CHECK_LAUNCH("pack_conv_weights");
    return MGF_OK;
}

static int conv_taps_dispatch(float* y, const float* x, const float* wp, const void* wb, const float* in_scale, const float* out_scale,
                              const mgf_conv_desc* dd, const mgf_epilogue* ep, mgf_stream_t stream) {
    MGF_REQUIRE(x && (wp || wb) && dd && (y || dd->rgb_out), MGF_EINVAL, "conv_taps: null pointer");
    const mgf_conv_desc& d = *dd;
    MGF_REQUIRE(d.n >= 1 && d.cin >= 1 && d.cout >= 1 && d.in_h >= 1 && d.in_w >= 1, MGF_EINVAL, "conv_taps: bad shape");
    MGF_REQUIRE(d.cout_pad >= d.cout && d.cout_pad % 32 == 0, MGF_EINVAL, "conv_taps: cout_pad must be a multiple of 32 >= cout");
    MGF_REQUIRE(d.ntaps >= 1 && d.ntaps <= MGF_MAX_TAPS, MGF_EINVAL, "conv_taps: ntaps out of range");
    MGF_REQUIRE(d.ngroups == 1 || d.ngroups == 4, MGF_EUNSUPPORTED, "conv_taps: ngroups must be 1 or 4");
    MGF_REQUIRE(d.istride >= 1 && d.ostride >= 1 && d.tile_h >= 1 && d.tile_w >= 1, MGF_EINVAL, "conv_taps: bad strides/tile");
    MGF_REQUIRE((int64_t)d.cin * d.in_h * d.in_w * d.n <= INT32_MAX, MGF_ETOOBIG, "conv_taps: input too large");
    MGF_REQUIRE((int64_t)MGF_MAX_TAPS * d.cin * d.cout_pad < (1LL << 30), MGF_ETOOBIG, "conv_taps: weight image too large");
    // (the staging path addresses one sample's input and the weight image with 32-bit BYTE offsets)
    MGF_REQUIRE((int64_t)d.cin * d.in_h * d.in_w < (1LL << 30), MGF_ETOOBIG, "conv_taps: one sample's input must stay below 4 GiB");
    MGF_REQUIRE(d.y_pitch >= d.out_w && d.y_plane >= (int64_t)d.out_h * d.y_pitch, MGF_EINVAL, "conv_taps: output strides too small");
    if (d.rgb_out) {
        MGF_REQUIRE(d.rgb_w && d.rgb_channels >= 1 && d.rgb_channels <= 4, MGF_EINVAL, "conv_taps: fused projection needs rgb_w and 1..4 channels");
        MGF_REQUIRE(d.cout <= 32 && d.ngroups == 1 && (!ep || !ep->residual), MGF_EUNSUPPORTED,
                    "conv_taps: fused projection needs cout <= 32, a plain conv and no residual");
    }
    if (ep) MGF_REQUIRE(ep->act == 0 || ep->act == MGF_ACT_LINEAR || ep->act == MGF_ACT_LRELU || ep->act == MGF_ACT_RELU,
                        MGF_EUNSUPPORTED, "conv_taps: epilogue activation %d unsupported", ep->act);
    ConvParams p;
    p.y = y; p.x = x; p.wp = wp; p.wb = wb; p.in_scale = in_scale; p.out_scale = out_scale; p.d = d;
    p.has_ep = ep != nullptr;
    if (ep) { p.ep = *ep; if (p.ep.act == 0) p.ep.act = MGF_ACT_LINEAR; } else { p.ep = mgf_epilogue{}; p.ep.gain = 1.f; }
    int dy_min = d.dy[0], dy_max = d.dy[0], dx_min = d.dx[0], dx_max = d.dx[0];
    for (int t = 1; t < d.ntaps; ++t) {
        dy_min = d.dy[t] < dy_min ? d.dy[t] : dy_min; dy_max = d.dy[t] > dy_max ? d.dy[t] : dy_max;
        dx_min = d.dx[t] < dx_min ? d.dx[t] : dx_min; dx_max = d.dx[t] > dx_max ? d.dx[t] : dx_max;
    }
    p.dy_min = dy_min; p.dx_min = dx_min;
    int mode = 0;
    if (d.ngroups == 4) {
        MGF_REQUIRE(d.ntaps == 9 && d.ostride == 2 && d.istride == 1, MGF_EUNSUPPORTED, "conv_taps: 4-group mode is the 3x3 stride-2 transposed conv");
        for (int t = 0; t < 9; ++t) {
            MGF_REQUIRE(d.group[t] == tconv_group(t), MGF_EINVAL, "conv_taps: tap %d must belong to parity group %d", t, tconv_group(t));
        }
        MGF_REQUIRE(ep == nullptr, MGF_EUNSUPPORTED, "conv_taps: the transposed-conv mode has no fused epilogue (it feeds the FIR pass)");
        MGF_REQUIRE(d.y_pitch % 2 == 0 && d.y_plane % 2 == 0 && d.y_batch % 2 == 0 && ((uintptr_t)y % 8) == 0, MGF_EINVAL,
                    "conv_taps: transposed-conv output needs even pitch/plane/batch strides and 8-byte alignment");
        mode = 1;
    }
    // tile geometry
    int tw_log2 = 5;
    while (tw_log2 > 2 && (1 << (tw_log2 - 1)) >= d.tile_w) --tw_log2;
    int TW = 1 << tw_log2;
    // workgroup tile: wide channel tiles when there are >= 64 output channels, more pixels per wave otherwise
    int wm = 1, wn = 2;
    // maps of at most 128 positions (4x4, 8x8 and the 5x5 / 9x9 parity grids of the first transposed convs): 128-lane pixel tiles,
    // half the padded MFMA work of the 256-lane ones
    static const char* small_env = mgf_knob("MGF_SMALL_TILES");    // tuning hook (experiments only): 0 = always 256-lane tiles
    const bool small = (int64_t)d.tile_h * d.tile_w <= 128 && !(small_env && small_env[0] == '0');
    if (wb) {
        // bf16x3: 32-channel x 256-pixel tiles of 3x3 stride-1 convolutions and of the transposed conv on its canonical 8 x 32-quad tile
        MGF_REQUIRE(d.ntaps == 9 && d.istride == 1 && !small && TW == 32 && d.cin % CKB == 0, MGF_EUNSUPPORTED,
                    "conv_taps(bf16x3): needs 9 taps, stride 1, a map of more than 128 positions at least 32 wide and cin %% 16 == 0");
    }
    if (mode == 1 && small) wn = 1;
    if (mode == 0 && wb) { wm = 1; wn = 2; }
    else if (mode == 0) {
        if (d.cout_pad % 64 == 0 && d.cout > 32) { wm = 2; wn = small ? 1 : 2; }
        else if (small) { wm = 1; wn = 1; }
        // <= 32 output channels on a large map: 12-row tiles for 3x3 (measured 91 vs 87 TFLOP/s at 1024^2), 16-row tiles for 1x1
        else if ((int64_t)d.tile_h * d.tile_w >= 512 * 64) { wm = 1; wn = d.ntaps == 9 ? 3 : 4; }
        // stride-2 convolutions (the data gradient of the up-sampling layers, IResNet's down-sampling convs): the footprint of a tile is
        // four times that of a stride-1 tile and does not fit the pipeline's register slots, so these launches stage synchronously --
        // with 128-lane tiles (4 x 32 outputs, 9 x 65 footprint: 37 KB of LDS, four workgroups per CU instead of two) the workgroups
        // cover each other's staging phases
        static const char* s2_env = mgf_knob("MGF_S2_SMALL");          // tuning hook (experiments only): 0 keeps the 256-lane tile
        if (d.istride == 2 && !(s2_env && s2_env[0] == '0')) wn = 1;
        // tuning hook (experiments only): MGF_CONV_TILE=wm,wn forces the tile of MODE-0 launches
        static const char* force = mgf_knob("MGF_CONV_TILE");
        if (force && force[0] && force[1] == ',' && force[2]) {
            const int fm = force[0] - '0', fn = force[2] - '0';
            if ((fm == 2 && fn == 2 && d.cout_pad % 64 == 0) || (fm == 1 && (fn == 2 || fn == 3 || fn == 4))) { wm = fm; wn = fn; }
        }
    }
    const int PX = 128 * wn;
    int rows = PX / TW;
    if (mode == 1) {
        // The parity grids of the transposed conv are (in+1) wide -- 33, 65, 129 ...: one column more than a whole number of
        // 32-wide tiles.  Any width works for the lane -> (row, column) map, so take the one that needs the fewest tiles
        // (rows * tw may leave a few of the 256 pixel lanes idle).
        static const char* tw_env = mgf_knob("MGF_TCONV_TW");      // tuning hook (experiments only): 0 keeps 32-wide tiles
        // an odd width costs the predicated epilogue and some idle lanes: it has to save at least 10% of the tiles
        int64_t best = mgf_cdiv(d.tile_w, TW) * mgf_cdiv(d.tile_h, rows) * 9;       // in tenths of a tile
        for (int tw = 8; tw <= 64 && !(tw_env && tw_env[0] == '0') && !wb; ++tw) {
            const int r = PX / tw;
            const int fh_ = r + (dy_max - dy_min), fw_ = tw + (dx_max - dx_min);
            if (256 / fw_ + 2 > 2 * fh_ || (size_t)CK * fh_ * fw_ > (size_t)(wn == 1 ? Slots<1, 1>::XS : Slots<1, 2>::XS) * 256) continue;
            const int64_t tiles = mgf_cdiv(d.tile_w, tw) * mgf_cdiv(d.tile_h, r);
            if (tiles * 10 < best) { best = tiles * 10; TW = tw; rows = r; }
        }
    }
    if (mode == 0 && !wb && d.istride == 1 && TW == 32 && d.tile_w > 16) {
        // A tall tap pattern (InceptionResnetV1's 7x1 layers: 7 rows x 1 column) makes the footprint of the 8 x 32 tile -- 14 x 32 x 8 channels --
        // overflow the pipeline's staging slots, and the launch used to fall back to the synchronous kernel (mfma_busy 0.39 against 0.61 for
        // the 1x7 twin).  A 16 x 16 tile covers the same 256 pixels with a 22 x 16 footprint, which fits.
        const size_t cap = (size_t)(wn == 4 ? Slots<1, 4>::XS : (wn == 3 ? Slots<1, 3>::XS : (wn == 1 ? Slots<1, 1>::XS : Slots<1, 2>::XS))) * 256;
        auto fits = [&](int tw) {
            const int r = PX / tw, fh_ = r + (dy_max - dy_min), fw_ = tw + (dx_max - dx_min);
            return (size_t)CK * fh_ * fw_ <= cap && 256 / fw_ + 2 <= 2 * fh_;
        };
        if (!fits(32) && fits(16)) { TW = 16; rows = PX / 16; }
    }
    p.tw = TW; p.rows = rows; p.tw_magic = (65536 + TW - 1) / TW;
    p.tiles_x = (int)mgf_cdiv(d.tile_w, TW);
    p.tiles_y = (int)mgf_cdiv(d.tile_h, rows);
    p.fh = (rows - 1) * d.istride + (dy_max - dy_min) + 1;
    p.fw = (TW - 1) * d.istride + (dx_max - dx_min) + 1;
    p.fixed_geo = 0;
    if (mode == 1 && p.fh == 9 && p.fw == 33) {
        p.fixed_geo = 1;
        for (int t = 0; t < 9; ++t)
            if (d.dy[t] != ((t / 3) == 2 ? -1 : 0) || d.dx[t] != ((t % 3) == 2 ? -1 : 0)) p.fixed_geo = 0;
        static const char* fg_env = mgf_knob("MGF_TCONV_FIXED");       // tuning hook (experiments only): 0 = run-time geometry
        if (fg_env && fg_env[0] == '0') p.fixed_geo = 0;
    }
    if (mode == 0 && d.ntaps == 9 && d.istride == 2 && p.fh == 9 && p.fw == 65) {
        p.fixed_geo = 2;
        for (int t = 0; t < 9; ++t)
            if (d.dy[t] - dy_min != t / 3 || d.dx[t] - dx_min != t % 3) p.fixed_geo = 0;
        static const char* fg_env = mgf_knob("MGF_TCONV_FIXED");
        if (fg_env && fg_env[0] == '0') p.fixed_geo = 0;
    }
    p.co_tiles = d.cout_pad / (32 * wm);
    // split-K when the output tiling alone cannot fill the chip (>= 2 workgroups on each of 256 CUs wanted)
    const int64_t base_wgs = (int64_t)p.tiles_x * p.tiles_y * p.co_tiles * d.n;
    const int nchunks = (int)mgf_cdiv(d.cin, wb ? CKB : CK);
    int ksplit = 1;
    static const int splitk_below = [] { const char* e = mgf_knob("MGF_SPLITK_BELOW"); return e ? atoi(e) : 256; }();   // tuning hook (512 / 256 / 128: 99.6 / 100.4 / 101.1 single-target gradient iters/s, 542 / 548 / 545 literal)
    if (d.workspace && base_wgs < splitk_below && nchunks >= 4 && !d.rgb_out) {
        static const int splitk_target = [] { const char* e = mgf_knob("MGF_SPLITK_TARGET"); return e ? atoi(e) : 1024; }();      // tuning hook: workgroups wanted (256 / 512 / 768 / 1024: 157.8 / 158.7 / 157.5 / 157.0 single-target gradient iters/s, headline and config 3 unchanged: tools/splitk_target_ab.sh)
        ksplit = (int)mgf_cdiv(splitk_target, base_wgs);
        if (ksplit > nchunks / 2) ksplit = nchunks / 2;
        const int64_t slice = (int64_t)d.n * d.cout * d.out_h * d.y_pitch;
        while (ksplit > 1 && slice * ksplit > d.workspace_floats) --ksplit;
        if (ksplit < 1) ksplit = 1;
    }
    p.chunks_per_split = (int)mgf_cdiv(nchunks, ksplit);
    ksplit = (int)mgf_cdiv(nchunks, p.chunks_per_split);          // drop empty tail slices
    p.ksplit = ksplit;
    p.partial = d.workspace;
    p.off32_ok = ((int64_t)(d.y_choff + d.cout_pad) * d.y_plane < (1LL << 30)) && ((int64_t)d.cout_pad * d.out_h * d.y_pitch < (1LL << 30));
    MGF_REQUIRE((int64_t)p.tiles_x * p.tiles_y * p.co_tiles * d.n * ksplit <= INT32_MAX, MGF_ETOOBIG, "conv_taps: grid too large");
    hipStream_t st = (hipStream_t)stream;
    int rc;
    if (wb) {
        MGF_REQUIRE(mode == 0 || p.fixed_geo == 1, MGF_EUNSUPPORTED, "conv_taps(bf16x3): the transposed conv needs its canonical tile (9 x 33 footprint)");
        rc = mode == 1 ? launch_conv_bf<1>(p, st) : launch_conv_bf<0>(p, st);
    }
    else if (mode == 1) rc = wn == 1 ? launch_conv<1, 1, 1>(p, st) : launch_conv<1, 2, 1>(p, st);
    else if (wn == 1) rc = wm == 2 ? launch_conv<2, 1, 0>(p, st) : launch_conv<1, 1, 0>(p, st);
    else if (wm == 2) rc = launch_conv<2, 2, 0>(p, st);
    else if (wn == 4) rc = launch_conv<1, 4, 0>(p, st);
    else if (wn == 3) rc = launch_conv<1, 3, 0>(p, st);
    else rc = launch_conv<1, 2, 0>(p, st);
    if (rc != MGF_OK) return rc;
    MGF_CHECK_LAUNCH("conv_taps");
    return MGF_OK;
}

extern "C" int mgf_conv_taps_f32(float* y, const float* x, const float* wp, const float* in_scale, const float* out_scale,
                                 const mgf_conv_desc* dd, const mgf_epilogue* ep, mgf_stream_t stream) {
    MGF_REQUIRE(wp, MGF_EINVAL, "conv_taps: null pointer");
    return conv_taps_dispatch(y, x, wp, nullptr, in_scale, out_scale, dd, ep, stream);
}

extern "C" int mgf_conv_taps_bf16x3_f32(float* y, const float* x, const void* wb, const float* in_scale, const float* out_scale,
                                        const mgf_conv_desc* dd, const mgf_epilogue* ep, mgf_stream_t stream) {
    MGF_REQUIRE(wb && ((uintptr_t)wb % 16) == 0, MGF_EINVAL, "conv_taps(bf16x3): the split weights must be 16-byte aligned");
    return conv_taps_dispatch(y, x, nullptr, wb, in_scale, out_scale, dd, ep, stream);
}
