// Landmark-Delaunay warp of a morph (1024_warp_morphs.py:78-113,163-210): the reference walks the
// triangles of the averaged-landmark mesh and, per triangle, warps the bounding-box patch of the generated image with
// cv2.warpAffine (bilinear, BORDER_REFLECT_101) and pastes it through a cv2.fillConvexPoly mask.  Here ONE gather kernel does the
// whole image: every output pixel finds the last triangle of the list that covers it (later triangles overwrite earlier ones in
// the reference), maps its centre through that triangle's destination->source affine map and samples the source bilinearly.
// Contract: include/mgf.h (mgf_piecewise_affine_warp_f32).  OpenCV itself is absent offline, so two of its implementation details are
// NOT reproduced: the 1/32-pixel fixed-point coordinate grid of warpAffine and the anti-aliased (LINE_AA) mask edge.
#include "mgf_common.h"

namespace {

__device__ __forceinline__ int reflect101(int i, int n) {       // cv2.BORDER_REFLECT_101: gfedcb|abcdefgh|gfedcba
    if (n == 1) return 0;
    while (i < 0 || i >= n) i = i < 0 ? -i : 2 * (n - 1) - i;
    return i;
}

__global__ __launch_bounds__(256) void piecewise_affine_warp_kernel(float* out, const float* src, const int32_t* tri, const float* inv, int ntri,
                                                                    int c, int h, int w, float background) {
    extern __shared__ float sm[];                    // [ntri][6] affine rows + [ntri][6] integer polygon
    float* A = sm;
    int* P = reinterpret_cast<int*>(sm + (size_t)ntri * 6);
    for (int i = threadIdx.x; i < ntri * 6; i += 256) { A[i] = inv[i]; P[i] = tri[i]; }
    __syncthreads();
    const int64_t total = (int64_t)h * w;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int x = (int)(i % w), y = (int)(i / w);
        int hit = -1;
        for (int t = 0; t < ntri; ++t) {
            const int* q = P + t * 6;
            // inclusive point-in-triangle on the integer polygon (either orientation), like a filled convex polygon with its boundary
            const int64_t e0 = (int64_t)(q[2] - q[0]) * (y - q[1]) - (int64_t)(q[3] - q[1]) * (x - q[0]);
            const int64_t e1 = (int64_t)(q[4] - q[2]) * (y - q[3]) - (int64_t)(q[5] - q[3]) * (x - q[2]);
            const int64_t e2 = (int64_t)(q[0] - q[4]) * (y - q[5]) - (int64_t)(q[1] - q[5]) * (x - q[4]);
            if ((e0 >= 0 && e1 >= 0 && e2 >= 0) || (e0 <= 0 && e1 <= 0 && e2 <= 0)) hit = t;
        }
        if (hit < 0) {
            for (int ch = 0; ch < c; ++ch) out[(int64_t)ch * total + i] = background;
            continue;
        }
        const float* a = A + hit * 6;
        const float sx = a[0] * (float)x + a[1] * (float)y + a[2];
        const float sy = a[3] * (float)x + a[4] * (float)y + a[5];
        const float fx = floorf(sx), fy = floorf(sy);
        const float lx = sx - fx, ly = sy - fy;
        const int x0 = reflect101((int)fx, w), x1 = reflect101((int)fx + 1, w);
        const int y0 = reflect101((int)fy, h), y1 = reflect101((int)fy + 1, h);
        for (int ch = 0; ch < c; ++ch) {
            const float* s = src + (int64_t)ch * total;
            const float v = (1.f - ly) * ((1.f - lx) * s[(int64_t)y0 * w + x0] + lx * s[(int64_t)y0 * w + x1]) +
                            ly * ((1.f - lx) * s[(int64_t)y1 * w + x0] + lx * s[(int64_t)y1 * w + x1]);
            out[(int64_t)ch * total + i] = v;
        }
    }
}

}  // namespace

extern "C" int mgf_piecewise_affine_warp_f32(float* out, const float* src, const int32_t* tri_xy, const float* dst_to_src, int32_t ntri, int32_t c,
                                             int32_t h, int32_t w, float background, mgf_stream_t stream) {
    MGF_REQUIRE(out && src && tri_xy && dst_to_src && ntri >= 1 && c >= 1 && h >= 1 && w >= 1, MGF_EINVAL, "piecewise_affine_warp: bad arguments");
    MGF_REQUIRE(ntri <= 1024, MGF_EUNSUPPORTED, "piecewise_affine_warp: at most 1024 triangles (got %d)", ntri);
    MGF_REQUIRE(out != src, MGF_EINVAL, "piecewise_affine_warp: out must not alias src");
    const size_t lds = (size_t)ntri * 12 * sizeof(float);
    hipLaunchKernelGGL(piecewise_affine_warp_kernel, dim3(mgf_stream_grid((int64_t)h * w, 256, 1)), dim3(256), lds, (hipStream_t)stream, out, src,
                       tri_xy, dst_to_src, ntri, c, h, w, background);
    MGF_CHECK_LAUNCH("piecewise_affine_warp");
    return MGF_OK;
}
