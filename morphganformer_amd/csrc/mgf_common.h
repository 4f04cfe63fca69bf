// Shared host/device helpers for the gfx950 kernels.  CDNA4 only: wave = 64 lanes, 256 CUs in 8 XCDs.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/mgf.h"

#define MGF_WAVE 64
#define MGF_NUM_CU 256
#define MGF_NUM_XCD 8

void mgf_set_error(const char* fmt, ...);

#define MGF_REQUIRE(cond, code, ...)            \
    do {                                         \
        if (!(cond)) {                           \
            mgf_set_error(__VA_ARGS__);          \
            return (code);                       \
        }                                        \
    } while (0)

#define MGF_CHECK_LAUNCH(name)                                                        \
    do {                                                                              \
        hipError_t e_ = hipGetLastError();                                            \
        if (e_ != hipSuccess) {                                                       \
            mgf_set_error("%s: launch failed: %s", name, hipGetErrorString(e_));      \
            return MGF_ELAUNCH;                                                       \
        }                                                                             \
    } while (0)

static inline int64_t mgf_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Dispatch knobs (tile shapes, split-K thresholds, kernel choices).  In the product library every knob IS its measured default:
// mgf_knob() returns nullptr at compile time and the `static const` it initialises folds to the constant -- nothing reads the environment.
// Only the experiment library of tools/build_exp.sh (-DMGF_TUNING_HOOKS, exp_build/, never the product's object cache) looks the name up,
// so that tools/*_micro.py can sweep a knob without a rebuild.
#ifdef MGF_TUNING_HOOKS
#include <stdlib.h>
static inline const char* mgf_knob(const char* name) { return getenv(name); }
#else
static inline constexpr const char* mgf_knob(const char*) { return nullptr; }
#endif

// Grid size for a grid-stride streaming kernel: enough workgroups to fill 256 CUs x 8, no more.
static inline int mgf_stream_grid(int64_t work_items, int block, int per_thread) {
    int64_t g = mgf_cdiv(work_items, (int64_t)block * per_thread);
    int64_t cap = (int64_t)MGF_NUM_CU * 8;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

// instrumentation hooks of mgf_conv_profile_begin/end for convolution kernels outside conv_taps.hip (no-ops unless profiling is on)
void mgf_prof_external_begin(hipStream_t st, const char* name, double flops, double bytes);
void mgf_prof_external_end(hipStream_t st);

#ifdef __HIPCC__
// Sum over the 64 lanes of a wave (DPP/bpermute butterflies emitted by the compiler for __shfl_xor).
template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
template <typename T>
__device__ __forceinline__ T wave_max(T v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { T u = __shfl_xor(v, o, 64); v = u > v ? u : v; }
    return v;
}
#endif
