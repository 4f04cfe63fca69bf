// Landmark-Delaunay warp of a morph (1024_warp_morphs.py:78-113,163-210).  The reference walks the triangles
// of the averaged-landmark mesh and, per triangle, warps the bounding-box patch of the generated image with cv2.warpAffine (INTER_LINEAR,
// BORDER_REFLECT_101) and pastes it through a cv2.fillConvexPoly mask; later triangles overwrite earlier ones.  Here the polygon fill -- a
// few hundred scanlines of integer edge walking per mesh -- is done once on the host into a label map (drivers.warp_plan: which triangle
// wrote each pixel LAST), and ONE gather kernel does all the pixel work: every labelled pixel evaluates its triangle's warpAffine exactly as
// OpenCV's WarpAffineInvoker + remapBilinear<float> do (imgproc/src/imgwarp.cpp, 4.x):
//   * coordinates on the 1/1024 fixed-point grid: X = (cvRound((iM[1] y + iM[2]) 1024) + 16 + cvRound(iM[0] x 1024)) >> 5, i.e. source
//     positions rounded to 1/32 pixel (INTER_BITS = 5, AB_BITS = 10, round_delta = 16), in double without fused multiply-adds;
//   * the 32 x 32 bilinear weight table of initInterTab2D in float ((1 - fy)(1 - fx), (1 - fy) fx, fy (1 - fx), fy fx with f = i / 32.f),
//     the four products summed left to right in float (for the integer-valued image the script reads back from its PNG every step is exact);
//   * BORDER_REFLECT_101 at the border of the triangle's own source PATCH (the numpy slice img_G[r1.y : r1.y + r1.h, r1.x : r1.x + r1.w]),
//     not at the image border.
// Contract: include/mgf.h (mgf_cv_warp_triangles_f32).  OpenCV is absent offline: parity with the real library is unpinned; the oracle
// (oracle/warp_ref.py) is a literal transcription of the same published sources.
#include "mgf_common.h"

namespace {

struct WarpTri {
    double im[6];            // the INVERTED 2x3 matrix warpAffine works with (destination patch -> source patch)
    int32_t dx, dy;          // origin of the destination patch in the image (r.x, r.y)
    int32_t sx, sy, sw, sh;  // source patch: origin and size inside the image (already clipped like the numpy slice)
};

__device__ __forceinline__ int reflect101(int p, int n) {       // core borderInterpolate(p, len, BORDER_REFLECT_101)
    if ((unsigned)p < (unsigned)n) return p;
    if (n == 1) return 0;
    do { p = p < 0 ? -p : n - 1 - (p - n) - 1; } while ((unsigned)p >= (unsigned)n);
    return p;
}

__device__ __forceinline__ int cv_round(double v) { return (int)__double2ll_rn(v); }      // cvRound: round half to even

__global__ __launch_bounds__(256) void cv_warp_triangles_kernel(float* out, const float* src, const int32_t* label, const WarpTri* tris, int ntri, int c,
                                                                int h, int w, float background) {
    const int64_t total = (int64_t)h * w;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int t = label[i];
        if (t < 0 || t >= ntri) {                                                        // no triangle covers this pixel
            for (int ch = 0; ch < c; ++ch) out[(int64_t)ch * total + i] = background;
            continue;
        }
        const WarpTri T = tris[t];
        if (T.sw < 1 || T.sh < 1 || T.sx < 0 || T.sy < 0 || T.sx + T.sw > w || T.sy + T.sh > h) {      // a record that would read outside src
            for (int ch = 0; ch < c; ++ch) out[(int64_t)ch * total + i] = background;
            continue;
        }
        const int x = (int)(i % w) - T.dx, y = (int)(i / w) - T.dy;
        // WarpAffineInvoker: no contraction anywhere (the products and sums are separate roundings in OpenCV's baseline build)
        const int X0 = cv_round(__dmul_rn(__dadd_rn(__dmul_rn(T.im[1], (double)y), T.im[2]), 1024.0)) + 16;
        const int Y0 = cv_round(__dmul_rn(__dadd_rn(__dmul_rn(T.im[4], (double)y), T.im[5]), 1024.0)) + 16;
        const int X = (X0 + cv_round(__dmul_rn(__dmul_rn(T.im[0], (double)x), 1024.0))) >> 5;
        const int Y = (Y0 + cv_round(__dmul_rn(__dmul_rn(T.im[3], (double)x), 1024.0))) >> 5;
        int sx = X >> 5, sy = Y >> 5;
        sx = sx < -32768 ? -32768 : (sx > 32767 ? 32767 : sx);                           // saturate_cast<short>
        sy = sy < -32768 ? -32768 : (sy > 32767 ? 32767 : sy);
        const float scale = 1.f / 32.f;
        const float fx = __fmul_rn((float)(X & 31), scale), fy = __fmul_rn((float)(Y & 31), scale);
        const float ax = __fsub_rn(1.f, fx), ay = __fsub_rn(1.f, fy);
        const float w0 = __fmul_rn(ay, ax), w1 = __fmul_rn(ay, fx), w2 = __fmul_rn(fy, ax), w3 = __fmul_rn(fy, fx);
        int x0, x1, y0, y1;
        if ((unsigned)sx < (unsigned)(T.sw - 1) && (unsigned)sy < (unsigned)(T.sh - 1)) { x0 = sx; x1 = sx + 1; y0 = sy; y1 = sy + 1; }
        else { x0 = reflect101(sx, T.sw); x1 = reflect101(sx + 1, T.sw); y0 = reflect101(sy, T.sh); y1 = reflect101(sy + 1, T.sh); }
        const int64_t o00 = (int64_t)(T.sy + y0) * w + T.sx + x0, o01 = (int64_t)(T.sy + y0) * w + T.sx + x1;
        const int64_t o10 = (int64_t)(T.sy + y1) * w + T.sx + x0, o11 = (int64_t)(T.sy + y1) * w + T.sx + x1;
        for (int ch = 0; ch < c; ++ch) {
            const float* s = src + (int64_t)ch * total;
            const float v = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(s[o00], w0), __fmul_rn(s[o01], w1)), __fmul_rn(s[o10], w2)), __fmul_rn(s[o11], w3));
            out[(int64_t)ch * total + i] = v;
        }
    }
}

}  // namespace

extern "C" int64_t mgf_cv_warp_triangle_bytes(void) { return (int64_t)sizeof(WarpTri); }

extern "C" int mgf_cv_warp_triangles_f32(float* out, const float* src, const int32_t* label, const void* triangles, int32_t ntri, int32_t c,
                                         int32_t h, int32_t w, float background, mgf_stream_t stream) {
    MGF_REQUIRE(out && src && label && triangles && ntri >= 1 && c >= 1 && h >= 1 && w >= 1, MGF_EINVAL, "cv_warp_triangles: bad arguments");
    MGF_REQUIRE(out != src, MGF_EINVAL, "cv_warp_triangles: out must not alias src");
    MGF_REQUIRE(((uintptr_t)triangles % 8) == 0, MGF_EINVAL, "cv_warp_triangles: the triangle records must be 8-byte aligned");
    hipLaunchKernelGGL(cv_warp_triangles_kernel, dim3(mgf_stream_grid((int64_t)h * w, 256, 1)), dim3(256), 0, (hipStream_t)stream, out, src, label,
                       reinterpret_cast<const WarpTri*>(triangles), ntri, c, h, w, background);
    MGF_CHECK_LAUNCH("cv_warp_triangles");
    return MGF_OK;
}
