// bias_act for gfx950: y = clamp(act(x + b) * gain) and its first/second derivative forms.
// Contract: include/mgf.h (mgf_bias_act); reference semantics: torch_utils/ops/bias_act.cu:15-139,
// bias_act.py:86-115.  HBM-bound streaming op: 16-byte vector loads/stores, grid-stride over
// 256 CUs x 8 workgroups, bias index computed once per vector when the bias period allows it.
#include "mgf_common.h"
#include <hip/hip_fp16.h>

namespace {

struct BAParams {
    void* y;
    const void* x;
    const void* b;
    const void* xref;
    const void* yref;
    const void* dy;
    int64_t numel;
    int64_t step_b;
    int64_t size_b;
    int grad;
    float alpha, gain, clamp;
};

template <typename T> struct Compute { typedef float type; };
template <> struct Compute<double> { typedef double type; };

// One activation = one functor with value / first-derivative / second-derivative forms.
// Derivatives are expressed through the saved, gain-free output yy (or the biased input xr for swish),
// exactly the quantities the reference plugin keeps.
template <int A, typename S> struct Act;

template <typename S> struct Act<MGF_ACT_LINEAR, S> {
    __device__ static S f(S x, S) { return x; }
    __device__ static S d1(S g, S, S, S) { return g; }
    __device__ static S d2(S, S, S, S) { return S(0); }
};
template <typename S> struct Act<MGF_ACT_RELU, S> {
    __device__ static S f(S x, S) { return x > S(0) ? x : S(0); }
    __device__ static S d1(S g, S yy, S, S) { return yy > S(0) ? g : S(0); }
    __device__ static S d2(S, S, S, S) { return S(0); }
};
template <typename S> struct Act<MGF_ACT_LRELU, S> {
    __device__ static S f(S x, S a) { return x > S(0) ? x : x * a; }
    __device__ static S d1(S g, S yy, S, S a) { return yy > S(0) ? g : g * a; }
    __device__ static S d2(S, S, S, S) { return S(0); }
};
template <typename S> struct Act<MGF_ACT_TANH, S> {
    // (the library tanh: the reference plugin's (e^x - e^-x) / (e^x + e^-x), bias_act.cu, cancels for small |x| -- 2e-5 of the value at
    // |x| ~ 1e-2 in float32 -- and the gate is the reference's torch implementation, bias_act.py:86-115)
    __device__ static S f(S x, S) { return tanh(x); }
    __device__ static S d1(S g, S yy, S, S) { return g * (S(1) - yy * yy); }
    __device__ static S d2(S g, S yy, S, S) { return g * (S(1) - yy * yy) * (S(-2) * yy); }
};
template <typename S> struct Act<MGF_ACT_SIGMOID, S> {
    __device__ static S f(S x, S) { return x < S(-80) ? S(0) : S(1) / (exp(-x) + S(1)); }
    __device__ static S d1(S g, S yy, S, S) { return g * yy * (S(1) - yy); }
    __device__ static S d2(S g, S yy, S, S) { return g * yy * (S(1) - yy) * (S(1) - S(2) * yy); }
};
template <typename S> struct Act<MGF_ACT_ELU, S> {
    __device__ static S f(S x, S) { return x >= S(0) ? x : expm1(x); }
    __device__ static S d1(S g, S yy, S, S) { return yy >= S(0) ? g : g * (yy + S(1)); }
    __device__ static S d2(S g, S yy, S, S) { return yy >= S(0) ? S(0) : g * (yy + S(1)); }
};
template <typename S> struct Act<MGF_ACT_SELU, S> {
    static constexpr double kScale = 1.0507009873554804934193349852946;
    static constexpr double kAlpha = 1.6732632423543772848170429916717;
    __device__ static S f(S x, S) { return x >= S(0) ? S(kScale) * x : S(kScale * kAlpha) * expm1(x); }
    __device__ static S d1(S g, S yy, S, S) { return yy >= S(0) ? g * S(kScale) : g * (yy + S(kScale * kAlpha)); }
    __device__ static S d2(S g, S yy, S, S) { return yy >= S(0) ? S(0) : g * (yy + S(kScale * kAlpha)); }
};
template <typename S> struct Act<MGF_ACT_SOFTPLUS, S> {
    __device__ static S f(S x, S) { return x > S(80) ? x : log1p(exp(x)); }
    __device__ static S d1(S g, S yy, S, S) { return g * (S(1) - exp(-yy)); }
    __device__ static S d2(S g, S yy, S, S) { S c = exp(-yy); return g * c * (S(1) - c); }
};
template <typename S> struct Act<MGF_ACT_SWISH, S> {
    __device__ static S f(S x, S) { return x < S(-80) ? S(0) : x / (exp(-x) + S(1)); }
    __device__ static S d1(S g, S, S xr, S) {
        if (xr > S(40)) return g;
        S c = exp(xr), d = c + S(1);
        return g * c * (xr + d) / (d * d);
    }
    __device__ static S d2(S g, S, S xr, S) {
        if (xr > S(40)) return S(0);
        S c = exp(xr), d = c + S(1);
        return g * c * (xr * (S(2) - d) + S(2) * d) / (d * d * d);
    }
};

// (element -> compute type: float for float / half, DOUBLE for double -- this used to return float for every T, which gave the float64 path float32 inputs)
template <typename T> __device__ __forceinline__ typename Compute<T>::type to_s(T v) { return (typename Compute<T>::type)v; }
template <> __device__ __forceinline__ float to_s<__half>(__half v) { return __half2float(v); }
template <typename T, typename S> __device__ __forceinline__ T from_s(S v) { return (T)v; }
template <> __device__ __forceinline__ __half from_s<__half, float>(float v) { return __float2half(v); }

template <typename T, typename S, int A>
__device__ __forceinline__ T ba_one(const BAParams& p, int64_t i, S bias) {
    const T* X = (const T*)p.x;
    S x = (S)to_s(X[i]);
    S alpha = (S)p.alpha, gain = (S)p.gain, clampv = (S)p.clamp;
    S y;
    if (p.grad == 0) {
        y = Act<A, S>::f(x + bias, alpha) * gain;
        if (clampv >= S(0)) y = y > clampv ? clampv : (y < -clampv ? -clampv : y);
    } else {
        S xr = p.xref ? (S)to_s(((const T*)p.xref)[i]) + bias : bias;
        S yr = p.yref ? (S)to_s(((const T*)p.yref)[i]) : S(0);
        S dyv = p.dy ? (S)to_s(((const T*)p.dy)[i]) : S(1);
        S yy = gain != S(0) ? yr / gain : S(0);
        if (A == MGF_ACT_SWISH) yr = Act<A, S>::f(xr, alpha) * gain;   // swish keeps x, not y: rebuild the clamp key
        y = (p.grad == 1 ? Act<A, S>::d1(x, yy, xr, alpha) : Act<A, S>::d2(x, yy, xr, alpha)) * gain * dyv;
        if (clampv >= S(0) && !(yr > -clampv && yr < clampv)) y = S(0);
    }
    return from_s<T, S>(y);
}

// VEC elements per thread per iteration; requires numel % VEC == 0 and (step_b % VEC == 0 or no bias).
template <typename T, int A, int VEC>
__global__ __launch_bounds__(256) void bias_act_kernel(BAParams p) {
    typedef typename Compute<T>::type S;
    struct alignas(sizeof(T) * VEC) Pack { T v[VEC]; };
    const int64_t nvec = p.numel / VEC;
    const T* B = (const T*)p.b;
    for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i0 = v * VEC;
        S bias = S(0);
        if (B) bias = (S)to_s(B[(i0 / p.step_b) % p.size_b]);
        Pack out;
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            out.v[j] = ba_one<T, S, A>(p, i0 + j, bias);
        }
        *reinterpret_cast<Pack*>((T*)p.y + i0) = out;
    }
}

template <typename T, int A>
int launch_act(const BAParams& p, hipStream_t st) {
    constexpr int VEC = 16 / sizeof(T);
    const bool aligned = ((uintptr_t)p.y % 16 == 0) && ((uintptr_t)p.x % 16 == 0) &&
                         (!p.xref || (uintptr_t)p.xref % 16 == 0) && (!p.yref || (uintptr_t)p.yref % 16 == 0) &&
                         (!p.dy || (uintptr_t)p.dy % 16 == 0);
    const bool vec_ok = aligned && (p.numel % VEC == 0) && (!p.b || p.step_b % VEC == 0);
    if (vec_ok) {
        int grid = mgf_stream_grid(p.numel / VEC, 256, 2);
        hipLaunchKernelGGL((bias_act_kernel<T, A, VEC>), dim3(grid), dim3(256), 0, st, p);
    } else {
        int grid = mgf_stream_grid(p.numel, 256, 4);
        hipLaunchKernelGGL((bias_act_kernel<T, A, 1>), dim3(grid), dim3(256), 0, st, p);
    }
    return 0;
}

template <typename T>
int launch_dtype(const BAParams& p, int act, hipStream_t st) {
    switch (act) {
        case MGF_ACT_LINEAR: return launch_act<T, MGF_ACT_LINEAR>(p, st);
        case MGF_ACT_RELU: return launch_act<T, MGF_ACT_RELU>(p, st);
        case MGF_ACT_LRELU: return launch_act<T, MGF_ACT_LRELU>(p, st);
        case MGF_ACT_TANH: return launch_act<T, MGF_ACT_TANH>(p, st);
        case MGF_ACT_SIGMOID: return launch_act<T, MGF_ACT_SIGMOID>(p, st);
        case MGF_ACT_ELU: return launch_act<T, MGF_ACT_ELU>(p, st);
        case MGF_ACT_SELU: return launch_act<T, MGF_ACT_SELU>(p, st);
        case MGF_ACT_SOFTPLUS: return launch_act<T, MGF_ACT_SOFTPLUS>(p, st);
        case MGF_ACT_SWISH: return launch_act<T, MGF_ACT_SWISH>(p, st);
    }
    return -1;
}

}  // namespace

extern "C" int mgf_bias_act(void* y, const void* x, const void* b, const void* xref, const void* yref, const void* dy,
                            int dtype, int64_t numel, int64_t step_b, int64_t size_b, int grad, int act, float alpha,
                            float gain, float clamp, mgf_stream_t stream) {
    MGF_REQUIRE(numel >= 0 && numel <= INT32_MAX, MGF_ETOOBIG, "bias_act: x is too large (%lld elements)", (long long)numel);
    MGF_REQUIRE(grad >= 0 && grad <= 2, MGF_EINVAL, "bias_act: grad must be 0, 1 or 2 (got %d)", grad);
    MGF_REQUIRE(act >= MGF_ACT_LINEAR && act <= MGF_ACT_SWISH, MGF_EINVAL, "bias_act: unknown activation id %d", act);
    MGF_REQUIRE(dtype == MGF_F32 || dtype == MGF_F64 || dtype == MGF_F16, MGF_EUNSUPPORTED, "bias_act: unsupported dtype %d", dtype);
    if (numel == 0) return MGF_OK;
    MGF_REQUIRE(x && y, MGF_EINVAL, "bias_act: x and y must be non-null");
    MGF_REQUIRE(!b || (step_b >= 1 && size_b >= 1), MGF_EINVAL, "bias_act: bias needs step_b >= 1 and size_b >= 1");
    BAParams p{y, x, b, xref, yref, dy, numel, b ? step_b : 1, b ? size_b : 1, grad, alpha, gain, clamp};
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MGF_F32) launch_dtype<float>(p, act, st);
    else if (dtype == MGF_F64) launch_dtype<double>(p, act, st);
    else launch_dtype<__half>(p, act, st);
    MGF_CHECK_LAUNCH("bias_act");
    return MGF_OK;
}
