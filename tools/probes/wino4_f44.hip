// PROBE (round 3), not part of the library: measured slower than the form-3 kernels on every generator layer (profiles/r3_probe_wino4_f44.txt)
// and shelved here with its numbers.  To rebuild the experiment: add this file to morphganformer_amd/build.py SOURCES (it includes
// mgf_common.h from csrc/), declare mgf_conv3x3_winograd4_f32 in include/mgf.h / _lib.py with the argument list of mgf_conv3x3_winograd3_f32,
// and pack the weights on the host:
//     G = [[1/4,0,0],[-1/6,-1/6,-1/6],[-1/6,1/6,-1/6],[1/24,1/12,1/6],[1/24,-1/12,1/6],[0,0,1]]            (float64)
//     U = einsum("ai,ocij,bj->aboc", G, w * gain, G).reshape(6, 6, cout/32, 2, 16, cin/4, 4)             # a, b, cot, mb, m, c, k
//     u = U.permute(0, 5, 2, 6, 4, 1, 3).float().reshape(6, cin/4, cout/32, 64, 6, 2)                     # lane = 16 k + m
//
// Winograd F(4x4, 3x3): the modulated 3x3 / stride-1 / pad-1 convolution of training/networks.py:288-303 with 36 instead of 64
// (F(2x2,3x3), csrc/wino3.hip) or 144 (direct) multiply-accumulates per 4x4 output pixels and channel pair -- 0.5625 of form 3's matrix work.
// Contract: include/mgf.h (mgf_conv3x3_winograd4_f32); weights in the layout of conv.winograd4_weights (built once per checkpoint, in
// float64, from Lavin & Gray's G: U = G g G^T).
//
//   Y = A^T [ (G g G^T) (.) (B^T d B) ] A        d: 6x6 input patch, Y: 4x4 outputs
//   B^T = [ 4  0 -5  0  1  0 ]      A^T = [ 1  1  1  1  1  0 ]
//         [ 0 -4 -4  1  1  0 ]            [ 0  1 -1  2 -2  0 ]
//         [ 0  4 -4 -1  1  0 ]            [ 0  1  1  4  4  0 ]
//         [ 0 -2 -1  2  1  0 ]            [ 0  1 -1  8 -8  1 ]
//         [ 0  2 -1 -2  1  0 ]
//         [ 0  4  0 -5  0  1 ]
//
// Same idea as form 3 -- the transformed input never touches LDS -- on the six rows of the 6x6 transformed patch: a workgroup is SIX waves,
// wave a owns row a (positions (a, 0..5)), and a lane computes exactly the B-operand values of its own MFMAs.  The MFMA is
// v_mfma_f32_16x16x4_f32 (16 output channels x 16 tiles x 4 input channels, 32 cycles, the same FLOP rate as 32x32x2): lane l is tile
// l & 15 of the workgroup's 16 tiles (64 x 4 output pixels) and channel l >> 4 of the chunk of 4, so one lane handles ONE (tile, channel)
// pair per chunk: row a of B^T d  (4 multiply-adds per column on 3 - 4 patch rows, coefficients wave-uniform, the style folded into them)
// and then the row's own 6-point transform (14 operations): ~ 38 VALU operations for 12 MFMAs (6 positions x 2 blocks of 16 channels).
// Accumulators: 6 x 2 x 4 = 48 registers, so three waves per SIMD = two workgroups per CU.  Weights stream from L2 (48 bytes per lane and
// chunk, three 16-byte loads), the footprint (6 rows x 66 columns per channel) through a double-buffered LDS image as in form 3.
// Output: every wave reduces its row over the columns (A^T on the right, in registers), the six rows meet in LDS, waves 0..3 each finish
// one of the tile's four output rows: 16 tiles x 4 pixels = 256 contiguous bytes per channel row and store instruction.
#include "mgf_common.h"
#include <algorithm>
#include <cstdlib>

namespace {

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

struct Wino4Params {
    float* y;
    const float* x;
    const float* u;           // [6 rows a][cin / 4][cout / 32][64 lanes][6 columns b][2 blocks]
    const float* in_scale;    // [n][cin] or null
    const float* out_scale;   // [n or 1][cout] or null
    int n, cin, h, w, cout, os_stride;
    int tiles_x, tiles_y, co_tiles;
    int xcd_per;
    mgf_epilogue ep;
    int has_ep;
};

constexpr int W4CK = 4;                     // input channels per chunk = K of the MFMA
constexpr int W4FW = 68;                    // LDS row pitch of the footprint (66 columns used; 16-byte aligned rows)
constexpr int W4CH = 6 * W4FW;              // floats per channel
constexpr int W4RAW = W4CK * W4CH;          // floats per staging buffer (1632)
constexpr int W4XS = 5;                     // staging slots per lane: 4 channels x 6 x 66 = 1584 elements over 384 lanes
constexpr int W4NV = 32;                    // values per lane and exchange slot: 4 output columns x 8 channel registers
constexpr unsigned W4OOB = 0xFFFFFFF0u;

__global__ __launch_bounds__(384, 3) void wino4_conv_kernel(Wino4Params p) {
    constexpr int MB = 2;
    extern __shared__ float lds[];
    float* const raw0 = lds;
    float* const raw1 = lds + W4RAW + 64;                          // (each buffer is followed by a 64-float scratch row: surplus staging lanes park there)
    const int tid = threadIdx.x, lane = tid & 63;
    const int a = __builtin_amdgcn_readfirstlane(tid >> 6);       // wave = row of the transformed patch
    const int tn = lane & 15, kq = lane >> 4;                      // tile of the workgroup's 16, channel of the chunk's 4

    int b_ = blockIdx.x;
    if (p.xcd_per > 0) {
        b_ = (b_ & 7) * p.xcd_per + (b_ >> 3);
        if (b_ >= p.n * p.tiles_x * p.tiles_y * p.co_tiles) return;
    }
    const int cot = b_ % p.co_tiles; b_ /= p.co_tiles;
    const int ptx = b_ % p.tiles_x; b_ /= p.tiles_x;
    const int pty = b_ % p.tiles_y;
    const int n = b_ / p.tiles_y;
    const int co0 = cot * 32, oy0 = pty * 4, ox0 = ptx * 64;
    const int plane = p.h * p.w;
    const int nck = p.cin / W4CK;
    const float* xn = p.x + (int64_t)n * p.cin * plane;
    const float* sc = p.in_scale ? p.in_scale + (int64_t)n * p.cin : nullptr;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)xn, 0, p.cin * plane * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rnull = __builtin_amdgcn_make_buffer_rsrc((void*)p.u, 0, 0, 0x00020000);
    const int64_t ubytes = (int64_t)36 * p.cin * p.cout * 4;
    const __amdgpu_buffer_rsrc_t ru = __builtin_amdgcn_make_buffer_rsrc((void*)p.u, 0, (int)ubytes, 0x00020000);

    // ---- staging slots: element e = tid + 384 j of the chunk's [4][6][66] footprint -> global offset (channel included) / LDS index ----
    unsigned xoff[W4XS];
    int xlds[W4XS];
#pragma unroll
    for (int j = 0; j < W4XS; ++j) {
        const int e = tid + 384 * j;
        const int ch = e / 396, rem = e - ch * 396;
        const int r = rem / 66, q = rem - r * 66;
        const int iy = oy0 - 1 + r, ix = ox0 - 1 + q;
        const bool ok = e < 1584 && iy >= 0 && iy < p.h && ix >= 0 && ix < p.w;
        xoff[j] = ok ? (unsigned)(ch * plane + iy * p.w + ix) * 4u : W4OOB;
        xlds[j] = e < 1584 ? ch * W4CH + r * W4FW + q : W4RAW + (tid & 63);        // surplus lanes: the scratch row behind the buffer
    }
    auto load_x = [&](float (&dst)[W4XS], int c, bool live = true) {
        const __amdgpu_buffer_rsrc_t r = live ? rx : rnull;
#pragma unroll
        for (int j = 0; j < W4XS; ++j)
            dst[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, xoff[j], c * W4CK * plane * 4, 0));
    };
    auto park_x = [&](float* R, const float (&src)[W4XS]) {
#pragma unroll
        for (int j = 0; j < W4XS; ++j) R[xlds[j]] = src[j];
    };
    // A operands of lane l for (row a, chunk c, channel tile): 12 consecutive floats [b][mb]
    const unsigned aoff = (unsigned)lane * 48u;
    auto load_a = [&](v4f (&dst)[3], int c, bool live = true) {
        const int soff = ((a * nck + c) * p.co_tiles + cot) * (64 * 48);
        const __amdgpu_buffer_rsrc_t r = live ? ru : rnull;
#pragma unroll
        for (int i = 0; i < 3; ++i)
            dst[i] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(r, aoff + 16u * i, soff, 0));
    };
    auto load_s = [&](int c) -> float { return sc ? sc[c * W4CK + kq] : 1.f; };

    // row a of B^T: T[j] = sum_i cf[i] * P[rw[i]][j]   (wave-uniform rows and coefficients)
    int rw0, rw1, rw2, rw3;
    float cf0, cf1, cf2, cf3;
    switch (a) {
        case 0:  rw0 = 0; rw1 = 2; rw2 = 4; rw3 = 4; cf0 = 4.f;  cf1 = -5.f; cf2 = 1.f;  cf3 = 0.f; break;
        case 1:  rw0 = 1; rw1 = 2; rw2 = 3; rw3 = 4; cf0 = -4.f; cf1 = -4.f; cf2 = 1.f;  cf3 = 1.f; break;
        case 2:  rw0 = 1; rw1 = 2; rw2 = 3; rw3 = 4; cf0 = 4.f;  cf1 = -4.f; cf2 = -1.f; cf3 = 1.f; break;
        case 3:  rw0 = 1; rw1 = 2; rw2 = 3; rw3 = 4; cf0 = -2.f; cf1 = -1.f; cf2 = 2.f;  cf3 = 1.f; break;
        case 4:  rw0 = 1; rw1 = 2; rw2 = 3; rw3 = 4; cf0 = 2.f;  cf1 = -1.f; cf2 = -2.f; cf3 = 1.f; break;
        default: rw0 = 1; rw1 = 3; rw2 = 5; rw3 = 5; cf0 = 4.f;  cf1 = -5.f; cf2 = 1.f;  cf3 = 0.f; break;
    }
    const int tbase = kq * W4CH + 4 * tn;
    const int o0 = tbase + rw0 * W4FW, o1 = tbase + rw1 * W4FW, o2 = tbase + rw2 * W4FW, o3 = tbase + rw3 * W4FW;
    auto transform = [&](float (&V)[6], const float* R, float sv) {
        const float c0 = cf0 * sv, c1 = cf1 * sv, c2 = cf2 * sv, c3 = cf3 * sv;          // the style rides on the row coefficients
        const v4f p0 = *reinterpret_cast<const v4f*>(R + o0), p1 = *reinterpret_cast<const v4f*>(R + o1);
        const v4f p2 = *reinterpret_cast<const v4f*>(R + o2), p3 = *reinterpret_cast<const v4f*>(R + o3);
        const v2f q0 = *reinterpret_cast<const v2f*>(R + o0 + 4), q1 = *reinterpret_cast<const v2f*>(R + o1 + 4);
        const v2f q2 = *reinterpret_cast<const v2f*>(R + o2 + 4), q3 = *reinterpret_cast<const v2f*>(R + o3 + 4);
        const v4f t03 = c0 * p0 + c1 * p1 + c2 * p2 + c3 * p3;
        const v2f t45 = c0 * q0 + c1 * q1 + c2 * q2 + c3 * q3;
        const float T0 = t03.x, T1 = t03.y, T2 = t03.z, T3 = t03.w, T4 = t45.x, T5 = t45.y;
        const float sa = T4 + T3, sb = T2 + T1, sd = T4 - T3, se = T2 - T1, sf = T4 - T2, sg = T3 - T1;
        V[0] = 4.f * T0 + (T4 - 5.f * T2);
        V[1] = sa - 4.f * sb;
        V[2] = sd - 4.f * se;
        V[3] = sf + 2.f * sg;
        V[4] = sf - 2.f * sg;
        V[5] = 4.f * T1 + (T5 - 5.f * T3);
    };

    v4f acc[6][MB];
#pragma unroll
    for (int b = 0; b < 6; ++b)
#pragma unroll
        for (int m = 0; m < MB; ++m) acc[b][m] = v4f{0.f, 0.f, 0.f, 0.f};
    auto mfma_chunk = [&](const v4f (&A)[3], const float (&V)[6]) {
        const float af[12] = {A[0].x, A[0].y, A[0].z, A[0].w, A[1].x, A[1].y, A[1].z, A[1].w, A[2].x, A[2].y, A[2].z, A[2].w};
#pragma unroll
        for (int b = 0; b < 6; ++b)
#pragma unroll
            for (int m = 0; m < MB; ++m) acc[b][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[b * MB + m], V[b], acc[b][m], 0, 0, 0);
    };

    // ---- prologue ----
    const int last = nck - 1;
    auto chunk = [&](int i) { return i < last ? i : last; };
    v4f A0[3], A1[3];
    float V0[6], V1[6];
    float xr[W4XS];
    float sv_nxt;                                                  // style of the chunk transformed next
    {
        float xa[W4XS], xb[W4XS];
        load_x(xa, 0);
        load_a(A0, 0);
        load_x(xb, chunk(1));
        load_x(xr, chunk(2));
        const float s0 = load_s(0);
        sv_nxt = load_s(chunk(1));
        park_x(raw0, xa);
        park_x(raw1, xb);
        __syncthreads();
        transform(V0, raw0, s0);
        __syncthreads();                                           // body(0) parks chunk 2 over raw0
    }
    auto body = [&](int i, v4f (&Acur)[3], v4f (&Anxt)[3], float (&Vcur)[6], float (&Vnxt)[6], float* raw_nxt, float* raw_park) {
        load_a(Anxt, chunk(i + 1), i + 1 < nck);
        __builtin_amdgcn_sched_barrier(0);
        transform(Vnxt, raw_nxt, sv_nxt);
        mfma_chunk(Acur, Vcur);
        __builtin_amdgcn_sched_barrier(0);
        park_x(raw_park, xr);
        load_x(xr, chunk(i + 3), i + 3 < nck);
        sv_nxt = load_s(chunk(i + 2));
        __syncthreads();
    };
    for (int it = 0; it < nck; it += 2) {
        body(it, A0, A1, V0, V1, raw1, raw0);
        if (it + 1 < nck) body(it + 1, A1, A0, V1, V0, raw0, raw1);
    }

    // ---- output transform, columns: R[c] = sum_b M[b] A^T[c][b] per accumulator register ----
    float* const xch = lds;                                        // [6 rows][W4NV values][64 lanes]; the staging buffers are dead
    float Rv[W4NV];                                                // [c][m * 4 + r]
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float m0 = acc[0][m][r], m1 = acc[1][m][r], m2 = acc[2][m][r], m3 = acc[3][m][r], m4 = acc[4][m][r], m5 = acc[5][m][r];
            const float s1 = m1 + m2, d1 = m1 - m2, s2 = m3 + m4, d2 = m3 - m4;
            const int q = m * 4 + r;
            Rv[0 * 8 + q] = m0 + s1 + s2;
            Rv[1 * 8 + q] = d1 + 2.f * d2;
            Rv[2 * 8 + q] = s1 + 4.f * s2;
            Rv[3 * 8 + q] = d1 + 8.f * d2 + m5;
        }
    {
        float* dst = xch + a * (W4NV * 64) + lane;
#pragma unroll
        for (int v = 0; v < W4NV; ++v) dst[v * 64] = Rv[v];
    }
    __syncthreads();
    if (a >= 4) return;                                            // waves 0..3 finish output row a of the tile
    // rows: Y[r] = sum_a A^T[r][a] R[a]:  r0 = R0+R1+R2+R3+R4, r1 = (R1-R2) + 2 (R3-R4), r2 = (R1+R2) + 4 (R3+R4), r3 = (R1-R2) + 8 (R3-R4) + R5
    const float k34 = a == 0 ? 1.f : (a == 1 ? 2.f : (a == 2 ? 4.f : 8.f));
    const float s12 = (a & 1) ? -1.f : 1.f;                        // sign of R2 (and of R4 inside the bracket)
    const float k0 = a == 0 ? 1.f : 0.f, k5 = a == 3 ? 1.f : 0.f;
    const int oy = oy0 + a, ox = ox0 + 4 * tn;
    const bool ok_px = oy < p.h && ox < p.w;                       // w % 4 == 0: a tile is inside whenever its first column is
    const bool do_ep = p.has_ep != 0;
    const float slope = !do_ep ? 1.f : (p.ep.act == MGF_ACT_LRELU ? p.ep.alpha : (p.ep.act == MGF_ACT_RELU ? 0.f : 1.f));
    const float gain = do_ep ? p.ep.gain : 1.f;
    const float* osc = p.out_scale ? p.out_scale + (int64_t)n * p.os_stride : nullptr;
    v4f nz = {0.f, 0.f, 0.f, 0.f};
    if (do_ep && p.ep.noise && ok_px) {
        const float ns = p.ep.noise_strength ? *p.ep.noise_strength : 1.f;
        nz = *reinterpret_cast<const v4f*>(p.ep.noise + (int64_t)(p.ep.noise_n > 1 ? n : 0) * plane + (int64_t)oy * p.w + ox) * ns;
    }
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)(p.y + (int64_t)n * p.cout * plane), 0, p.cout * plane * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rres = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((do_ep && p.ep.residual) ? p.ep.residual + (int64_t)n * p.cout * plane : p.x), 0, (do_ep && p.ep.residual) ? p.cout * plane * 4 : 0, 0x00020000);
    const float* x0 = xch + 0 * (W4NV * 64) + lane;
    const float* x1 = xch + 1 * (W4NV * 64) + lane;
    const float* x2 = xch + 2 * (W4NV * 64) + lane;
    const float* x3 = xch + 3 * (W4NV * 64) + lane;
    const float* x4 = xch + 4 * (W4NV * 64) + lane;
    const float* x5 = xch + 5 * (W4NV * 64) + lane;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int co = co0 + 16 * (q >> 2) + 4 * kq + (q & 3);
        const float os = osc ? osc[co] : 1.f;
        const float bb = (do_ep && p.ep.bias) ? p.ep.bias[co] : 0.f;
        const unsigned voff = ok_px ? (unsigned)(co * plane + oy * p.w + ox) * 4u : W4OOB;
        const v4f rr = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rres, voff, 0, 0));
        v4f out;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int v = c * 8 + q;
            const float r0 = x0[v * 64], r1 = x1[v * 64], r2 = x2[v * 64], r3 = x3[v * 64], r4 = x4[v * 64], r5 = x5[v * 64];
            float t = (r1 + s12 * r2) + k34 * (r3 + s12 * r4) + k0 * r0 + k5 * r5;
            t *= os;
            t += nz[c];
            t += bb;
            t = t > 0.f ? t : t * slope;
            out[c] = t * gain + rr[c];
        }
        typedef unsigned v4u __attribute__((ext_vector_type(4)));
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, out), ry, voff, 0, 0);
    }
}

}  // namespace

extern "C" int mgf_conv3x3_winograd4_f32(float* y, const float* x, const float* u, const float* in_scale, const float* out_scale, int32_t n,
                                         int32_t cin, int32_t h, int32_t w, int32_t cout, int32_t out_scale_stride, const mgf_epilogue* ep,
                                         mgf_stream_t stream) {
    MGF_REQUIRE(y && x && u && n >= 1 && cin >= 1 && cout >= 1 && h >= 4 && w >= 4, MGF_EINVAL, "conv3x3_winograd4: bad arguments");
    MGF_REQUIRE(cin % W4CK == 0 && cout % 32 == 0 && h % 4 == 0 && w % 4 == 0, MGF_EUNSUPPORTED,
                "conv3x3_winograd4: cin %% 4, cout %% 32, h %% 4 and w %% 4 must be 0 (got %d, %d, %dx%d)", cin, cout, h, w);
    MGF_REQUIRE((int64_t)cin * h * w <= INT32_MAX / 4 && (int64_t)36 * cin * cout <= INT32_MAX / 4 && (int64_t)cout * h * w <= INT32_MAX / 4, MGF_ETOOBIG,
                "conv3x3_winograd4: one sample / the weight planes must stay below 2 GiB (32-bit buffer offsets)");
    MGF_REQUIRE(((uintptr_t)u % 16) == 0 && ((uintptr_t)y % 16) == 0, MGF_EINVAL, "conv3x3_winograd4: u and y must be 16-byte aligned");
    if (ep) {
        MGF_REQUIRE(ep->act == 0 || ep->act == MGF_ACT_LINEAR || ep->act == MGF_ACT_LRELU || ep->act == MGF_ACT_RELU, MGF_EUNSUPPORTED,
                    "conv3x3_winograd4: epilogue activation %d unsupported", ep->act);
        MGF_REQUIRE((!ep->residual || ((uintptr_t)ep->residual % 16) == 0) && (!ep->noise || ((uintptr_t)ep->noise % 16) == 0), MGF_EINVAL,
                    "conv3x3_winograd4: residual and noise must be 16-byte aligned");
    }
    Wino4Params p;
    p.y = y; p.x = x; p.u = u; p.in_scale = in_scale; p.out_scale = out_scale;
    p.n = n; p.cin = cin; p.h = h; p.w = w; p.cout = cout; p.os_stride = out_scale_stride;
    p.tiles_x = (int)mgf_cdiv(w, 64); p.tiles_y = h / 4; p.co_tiles = cout / 32;
    p.has_ep = ep != nullptr;
    if (ep) { p.ep = *ep; if (p.ep.act == 0) p.ep.act = MGF_ACT_LINEAR; } else { p.ep = mgf_epilogue{}; p.ep.gain = 1.f; }
    int64_t blocks = (int64_t)n * p.tiles_x * p.tiles_y * p.co_tiles;
    MGF_REQUIRE(blocks <= INT32_MAX - 8, MGF_ETOOBIG, "conv3x3_winograd4: too many workgroups");
    p.xcd_per = 0;
    if ((int64_t)36 * cin * cout * 4 <= (4 << 20) && blocks >= 16) {
        p.xcd_per = (int)((blocks + 7) / 8);
        blocks = (int64_t)p.xcd_per * 8;
    }
    // staging: 2 x (1632 floats + a scratch row for the surplus staging lanes); exchange: 6 x 32 x 64 floats over the same memory
    const size_t lds = std::max<size_t>((size_t)(2 * (W4RAW + 64)), (size_t)(6 * W4NV * 64)) * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)wino4_conv_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
        if (e != hipSuccess) { mgf_set_error("conv3x3_winograd4: cannot raise dynamic LDS: %s", hipGetErrorString(e)); return MGF_ELAUNCH; }
        attr_set = true;
    }
    mgf_prof_external_begin((hipStream_t)stream, "wino4_conv_kernel", 2.0 * 9 * cin * (double)cout * h * w * n,
                            4.0 * ((double)n * cin * h * w + 9.0 * cin * cout + (double)n * cout * h * w));
    hipLaunchKernelGGL(wino4_conv_kernel, dim3((unsigned)blocks), dim3(384), lds, (hipStream_t)stream, p);
    mgf_prof_external_end((hipStream_t)stream);
    MGF_CHECK_LAUNCH("conv3x3_winograd4");
    return MGF_OK;
}
