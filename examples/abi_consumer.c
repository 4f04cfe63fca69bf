/*
 * A torch-free consumer of the C ABI (include/mgf.h): plain C99, the HIP runtime only for device memory.
 * What a maintainer's cgo / JNI / ctypes stub would do, as a program: allocate, call, compare with the
 * definitions restated on the host right here.
 *
 *   gcc -std=c99 -O1 -I include examples/abi_consumer.c -o abi_consumer -L morphganformer_amd -lmgf_hip -L /opt/rocm/lib -lamdhip64 -lm
 *   LD_LIBRARY_PATH=morphganformer_amd:/opt/rocm/lib ./abi_consumer
 *
 * Entry points exercised: mgf_bias_act (bias_act.cpp:24), mgf_upfirdn2d (upfirdn2d.cpp:8), mgf_mse_f32 (torch.nn.MSELoss of the drivers),
 * mgf_dssim_u8_f32 (`dssim`, 1024_example_SSIM.py:115-117), mgf_winograd2_weights_f32 + mgf_conv3x3_winograd3_f32 (modulated_conv2d,
 * training/networks.py:288-303: the FP32-MFMA kernel of the hot path).  Exit code 0 = every result within its stated tolerance.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "mgf.h"

/* the four HIP runtime calls this program needs (hip_runtime_api.h declares them the same way) */
extern int hipMalloc(void** ptr, size_t size);
extern int hipFree(void* ptr);
extern int hipMemcpy(void* dst, const void* src, size_t size, int kind);   /* 1 = host to device, 2 = device to host */
extern int hipDeviceSynchronize(void);

static int failures = 0;
#define CHECK(cond, ...) do { if (!(cond)) { printf("FAIL %s:%d: ", __FILE__, __LINE__); printf(__VA_ARGS__); printf("\n"); ++failures; } } while (0)
#define MGF(call) do { int rc_ = (call); if (rc_ != MGF_OK) { printf("FAIL %s -> %d (%s)\n", #call, rc_, mgf_last_error()); return 1; } } while (0)

static void* to_device(const void* host, size_t bytes) {
    void* d = NULL;
    if (hipMalloc(&d, bytes) != 0) { printf("hipMalloc failed\n"); exit(2); }
    if (host && hipMemcpy(d, host, bytes, 1) != 0) { printf("hipMemcpy failed\n"); exit(2); }
    return d;
}
static void to_host(void* host, const void* dev, size_t bytes) {
    if (hipDeviceSynchronize() != 0 || hipMemcpy(host, dev, bytes, 2) != 0) { printf("device error\n"); exit(2); }
}
static float frand(unsigned* s) { *s = *s * 1664525u + 1013904223u; return (float)((*s >> 8) & 0xFFFF) / 32768.0f - 1.0f; }

/* ---- bias_act: y = lrelu(x + b[channel]) * gain (bias_act.py:137-198 with act='lrelu', dim=1) ---- */
static int test_bias_act(void) {
    enum { N = 3, C = 5, HW = 7 };
    float x[N * C * HW], b[C], y[N * C * HW];
    unsigned s = 1;
    for (int i = 0; i < N * C * HW; ++i) x[i] = frand(&s);
    for (int i = 0; i < C; ++i) b[i] = frand(&s);
    float *dx = to_device(x, sizeof x), *db = to_device(b, sizeof b), *dy = to_device(NULL, sizeof y);
    const float alpha = 0.2f, gain = 1.41421356f;
    MGF(mgf_bias_act(dy, dx, db, NULL, NULL, NULL, MGF_F32, N * C * HW, HW, C, 0, MGF_ACT_LRELU, alpha, gain, -1.0f, NULL));
    to_host(y, dy, sizeof y);
    for (int i = 0; i < N * C * HW; ++i) {
        const float t = x[i] + b[(i / HW) % C];
        const float want = (t > 0 ? t : t * alpha) * gain;
        CHECK(fabsf(y[i] - want) <= 1e-6f * (1 + fabsf(want)), "bias_act[%d] = %g, want %g", i, y[i], want);
    }
    hipFree(dx); hipFree(db); hipFree(dy);
    return 0;
}

/* ---- upfirdn2d: zero-stuff by 2, pad (2, 1, 2, 1), correlate with the FLIPPED 4x4 filter, gain 4 (upfirdn2d.py:148-196) ---- */
static int test_upfirdn2d(void) {
    enum { H = 6, W = 5, F = 4, OH = 2 * H, OW = 2 * W };      /* (in * up + pad0 + pad1 - fsize + down) / down = 2 in */
    float x[2 * H * W], f[F * F], y[2 * OH * OW];
    const float taps[F] = {1, 3, 2, 1};                   /* not symmetric: the flip is part of what is checked */
    unsigned s = 7;
    for (int i = 0; i < 2 * H * W; ++i) x[i] = frand(&s);
    for (int i = 0; i < F; ++i) for (int j = 0; j < F; ++j) f[i * F + j] = taps[i] * taps[j] / 49.0f;
    float *dx = to_device(x, sizeof x), *df = to_device(f, sizeof f), *dy = to_device(NULL, sizeof y);
    MGF(mgf_upfirdn2d(dy, dx, df, MGF_F32, 1, 2, H, W, 2 * H * W, H * W, W, 1, OH, OW, 2 * OH * OW, OH * OW, OW, 1, F, F, 2, 2, 1, 1,
                      2, 1, 2, 1, 0, 4.0f, NULL, NULL));
    to_host(y, dy, sizeof y);
    for (int c = 0; c < 2; ++c)
        for (int oy = 0; oy < OH; ++oy)
            for (int ox = 0; ox < OW; ++ox) {
                double acc = 0;
                for (int fy = 0; fy < F; ++fy)
                    for (int fx = 0; fx < F; ++fx) {
                        const int uy = oy + fy - 2, ux = ox + fx - 2;          /* position in the zero-stuffed map (pad0 = 2) */
                        if (uy < 0 || ux < 0 || uy >= 2 * H || ux >= 2 * W || (uy & 1) || (ux & 1)) continue;
                        acc += (double)x[(c * H + uy / 2) * W + ux / 2] * f[(F - 1 - fy) * F + (F - 1 - fx)];
                    }
                const float want = (float)(acc * 4.0), got = y[(c * OH + oy) * OW + ox];
                CHECK(fabsf(got - want) <= 2e-6f * (1 + fabsf(want)), "upfirdn2d[%d,%d,%d] = %g, want %g", c, oy, ox, got, want);
            }
    hipFree(dx); hipFree(df); hipFree(dy);
    return 0;
}

/* ---- the pixel terms: MSE and DSSIM of two [3, 20, 24] images in [-1, 1] ---- */
static int quant(float v) { float q = rintf(v * 127.5f + 127.5f); return (int)(q < 0 ? 0 : (q > 255 ? 255 : q)); }
static int test_pixel_terms(void) {
    enum { C = 3, H = 20, W = 24, NUM = C * H * W };
    float a[2 * NUM], b[NUM], out[2];
    unsigned s = 3;
    for (int i = 0; i < NUM; ++i) { b[i] = 0.8f * sinf(0.05f * (float)i) ; a[i] = b[i] + 0.1f * frand(&s); a[NUM + i] = b[i]; }
    float *da = to_device(a, sizeof a), *db = to_device(b, sizeof b), *dout = to_device(NULL, sizeof out);
    float* red = to_device(NULL, 2 * (size_t)mgf_reduce_scratch_floats() * sizeof(float));
    MGF(mgf_mse_f32(dout, da, db, 2, NUM, 0, 1.0f, 0, red, NULL));
    to_host(out, dout, sizeof out);
    double m = 0;
    for (int i = 0; i < NUM; ++i) m += ((double)a[i] - b[i]) * ((double)a[i] - b[i]);
    CHECK(fabs(out[0] - m / NUM) <= 1e-5 * (m / NUM), "mse = %g, want %g", out[0], m / NUM);
    CHECK(out[1] == 0.0f, "mse of identical images = %g", out[1]);

    const int64_t sb = mgf_dssim_scratch_bytes(2, C, H, W);
    CHECK(sb > 0, "dssim scratch size %lld", (long long)sb);
    void* scratch = to_device(NULL, (size_t)sb);
    MGF(mgf_dssim_u8_f32(dout, da, db, 2, C, H, W, 0, 255.0f, 1.0f, 0, scratch, NULL));
    to_host(out, dout, sizeof out);
    const double c1 = 2.55 * 2.55, c2 = 7.65 * 7.65;
    double mean_c = 0;
    for (int c = 0; c < C; ++c) {
        double sum = 0;
        for (int i = 0; i + 7 <= H; ++i)
            for (int j = 0; j + 7 <= W; ++j) {
                double sx = 0, sy = 0, sxx = 0, syy = 0, sxy = 0;
                for (int r = 0; r < 7; ++r)
                    for (int q = 0; q < 7; ++q) {
                        const double xv = quant(a[(c * H + i + r) * W + j + q]), yv = quant(b[(c * H + i + r) * W + j + q]);
                        sx += xv; sy += yv; sxx += xv * xv; syy += yv * yv; sxy += xv * yv;
                    }
                const double ux = sx / 49, uy = sy / 49, k = 49.0 / 48.0;
                const double vx = k * (sxx / 49 - ux * ux), vy = k * (syy / 49 - uy * uy), vxy = k * (sxy / 49 - ux * uy);
                sum += (2 * ux * uy + c1) * (2 * vxy + c2) / ((ux * ux + uy * uy + c1) * (vx + vy + c2));
            }
        mean_c += sum / ((H - 6) * (W - 6));
    }
    const double want = (1 - mean_c / C) / 2;
    CHECK(fabs(out[0] - want) <= 2e-7, "dssim = %.9g, want %.9g", out[0], want);
    CHECK(out[1] == 0.0f, "dssim of identical images = %g", out[1]);
    /* argument checking happens on the host, before any launch: a map narrower than the window is refused with a message */
    CHECK(mgf_dssim_u8_f32(dout, da, db, 1, C, 6, W, 0, 255.0f, 1.0f, 0, scratch, NULL) == MGF_EINVAL && strlen(mgf_last_error()) > 0,
          "dssim accepted a 6-row image");
    hipFree(da); hipFree(db); hipFree(dout); hipFree(red); hipFree(scratch);
    return 0;
}

/* ---- the hot path's matrix kernel: modulated 3x3 convolution (modulated_conv2d, training/networks.py:288-303) through the Winograd form the
 * engine uses -- y[n, co] = lrelu(d[n, co] * sum_ci (s[n, ci] w[co, ci]) (*) x[n, ci] + bias[co]) * gain, pad 1 -- against direct loops ---- */
static int test_modulated_conv(void) {
    enum { N = 2, CI = 8, CO = 32, H = 32, W = 32 };
    static float x[N * CI * H * W], w[CO * CI * 9], s[N * CI], d[N * CO], bias[CO], y[N * CO * H * W];
    unsigned sd = 11;
    for (int i = 0; i < N * CI * H * W; ++i) x[i] = frand(&sd);
    for (int i = 0; i < CO * CI * 9; ++i) w[i] = frand(&sd);
    for (int i = 0; i < N * CI; ++i) s[i] = 1.0f + 0.5f * frand(&sd);
    for (int i = 0; i < CO; ++i) bias[i] = 0.1f * frand(&sd);
    const float wgain = 1.0f / sqrtf((float)(CI * 9));
    for (int n = 0; n < N; ++n)                       /* demodulation d = rsqrt(sum (w s)^2 + 1e-8) (:291-293) */
        for (int co = 0; co < CO; ++co) {
            double acc = 0;
            for (int ci = 0; ci < CI; ++ci)
                for (int t = 0; t < 9; ++t) { const double v = (double)w[(co * CI + ci) * 9 + t] * wgain * s[n * CI + ci]; acc += v * v; }
            d[n * CO + co] = (float)(1.0 / sqrt(acc + 1e-8));
        }
    float *dx = to_device(x, sizeof x), *dw = to_device(w, sizeof w), *ds = to_device(s, sizeof s), *dd = to_device(d, sizeof d);
    float *db = to_device(bias, sizeof bias), *dy = to_device(NULL, sizeof y), *du = to_device(NULL, 16 * CI * CO * sizeof(float));
    MGF(mgf_winograd2_weights_f32(du, dw, CO, CI, wgain, NULL));
    mgf_epilogue ep;
    memset(&ep, 0, sizeof ep);
    ep.bias = db; ep.act = MGF_ACT_LRELU; ep.alpha = 0.2f; ep.gain = 1.41421356f; ep.noise_n = 1;
    MGF(mgf_conv3x3_winograd3_f32(dy, dx, du, ds, dd, N, CI, H, W, CO, CO, &ep, NULL));
    to_host(y, dy, sizeof y);
    double worst = 0, scale = 0;
    for (int n = 0; n < N; ++n)
        for (int co = 0; co < CO; ++co)
            for (int oy = 0; oy < H; ++oy)
                for (int ox = 0; ox < W; ++ox) {
                    double acc = 0;
                    for (int ci = 0; ci < CI; ++ci)
                        for (int ky = 0; ky < 3; ++ky)
                            for (int kx = 0; kx < 3; ++kx) {
                                const int iy = oy + ky - 1, ix = ox + kx - 1;
                                if (iy < 0 || ix < 0 || iy >= H || ix >= W) continue;
                                acc += (double)w[(co * CI + ci) * 9 + ky * 3 + kx] * wgain * s[n * CI + ci] * x[((n * CI + ci) * H + iy) * W + ix];
                            }
                    double t = acc * d[n * CO + co] + bias[co];
                    t = (t > 0 ? t : 0.2 * t) * 1.41421356;
                    const double e = fabs(y[((n * CO + co) * H + oy) * W + ox] - t);
                    if (e > worst) worst = e;
                    if (fabs(t) > scale) scale = fabs(t);
                }
    CHECK(worst <= 1e-5 * scale, "modulated 3x3 conv: worst error %g of max |y| %g", worst, scale);
    hipFree(dx); hipFree(dw); hipFree(ds); hipFree(dd); hipFree(db); hipFree(dy); hipFree(du);
    return 0;
}

int main(void) {
    if (!mgf_device_ok()) { printf("abi_consumer: no usable gfx950 device\n"); return 3; }
    if (test_bias_act() || test_upfirdn2d() || test_pixel_terms() || test_modulated_conv()) return 1;
    if (failures) { printf("abi_consumer: %d check(s) failed\n", failures); return 1; }
    printf("abi_consumer: OK (mgf_bias_act, mgf_upfirdn2d, mgf_mse_f32, mgf_dssim_u8_f32, mgf_conv3x3_winograd3_f32; library version %d)\n", mgf_version());
    return 0;
}
