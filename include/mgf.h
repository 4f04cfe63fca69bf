/*
 * mgf.h -- C ABI of the MI355X (gfx950) latent-projection / GANformer-synthesis engine.
 *
 * Drop-in boundary for the hot path of nz0001na/MorphGANformer.  Every entry point
 * takes plain device pointers + sizes + a HIP stream; no torch types.  All functions:
 *   - return 0 on success, a negative MGF_E* code on failure (mgf_last_error() gives the message,
 *     thread-local), never throw, never synchronise the device, never allocate device memory;
 *   - launch on the given stream (NULL = the null stream), outputs are caller-allocated;
 *   - validate shapes/extents on the host before launching (mirrors the TORCH_CHECKs of the plugins).
 *
 * Each declaration cites the reference interface it replaces (file:line under the reference tree).
 */
#ifndef MGF_H_
#define MGF_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* mgf_stream_t; /* hipStream_t */

enum { MGF_OK = 0, MGF_EINVAL = -1, MGF_EUNSUPPORTED = -2, MGF_ELAUNCH = -3, MGF_ETOOBIG = -4 };
enum { MGF_F32 = 0, MGF_F64 = 1, MGF_F16 = 2 };
/* OR-ed into mgf_upfirdn2d's `flip` argument by a caller that KNOWS the filter is an outer product fy (x) fx (every filter
 * upfirdn2d.setup_filter builds from a 1-D tap list is): lets the 4x4 blur take its separable kernel.  Without the hint the
 * general kernel runs; a wrong hint gives wrong results (the filter lives in device memory and is not inspected). */
enum { MGF_FILTER_SEPARABLE = 2 };
/* OR-ed into `flip` by the engine-internal caller (conv.upfirdn_into): tensors of more than INT32_MAX elements are accepted wherever the
 * dispatcher picks one of the tiled / streaming kernels (64-bit plane addressing); without it the plug-in contract of the reference holds
 * (upfirdn2d.cpp:14-15,28: numel <= INT_MAX) and such a call is refused with MGF_ETOOBIG. */
enum { MGF_FILTER_LARGE = 4 };
/* activation ids = the reference's cuda_idx (torch_utils/ops/bias_act.py:15-25) */
enum { MGF_ACT_LINEAR = 1, MGF_ACT_RELU = 2, MGF_ACT_LRELU = 3, MGF_ACT_TANH = 4, MGF_ACT_SIGMOID = 5,
       MGF_ACT_ELU = 6, MGF_ACT_SELU = 7, MGF_ACT_SOFTPLUS = 8, MGF_ACT_SWISH = 9,
       /* mgf_conv1x1_f32 only: linear in front of the residual add, ReLU behind it -- y = relu((acc + bias) * gain + residual), the residual
        * blocks of facenet_pytorch's InceptionResnetV1 (`out = relu(conv2d(cat) * scale + x)`) */
       MGF_ACT_RELU_POST = 10 };

const char* mgf_last_error(void);
int mgf_version(void);
/* 1 if a gfx950 device is present and usable */
int mgf_device_ok(void);

/* ------------------------------------------------------------------------------------------------
 * bias_act -- replaces bias_act_plugin.bias_act(x, b, xref, yref, dy, grad, dim, act, alpha, gain, clamp)
 * (torch_utils/ops/bias_act.cpp:24-82, kernel bias_act.cu:15-139).
 *   y[i] = F_grad( x[i], b[(i / step_b) % size_b], xref[i], yref[i], dy[i] )
 * grad 0: y = clamp(act(x+b)*gain); grad 1: first derivative form (x carries dy); grad 2: second derivative form.
 * NULL pointer = the reference's "empty tensor".  numel <= INT32_MAX (bias_act.cpp:32).
 */
int mgf_bias_act(void* y, const void* x, const void* b, const void* xref, const void* yref, const void* dy,
                 int dtype, int64_t numel, int64_t step_b, int64_t size_b,
                 int grad, int act, float alpha, float gain, float clamp, mgf_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * upfirdn2d -- replaces upfirdn2d_plugin.upfirdn2d(x, f, upx, upy, downx, downy, padx0, padx1, pady0, pady1,
 * flip, gain) (torch_utils/ops/upfirdn2d.cpp:8-86, kernels upfirdn2d.cu:21-333).
 * x: [n, c, in_h, in_w] with element strides (sn, sc, sh, sw) -> NCHW and channels_last both accepted;
 * f: float32 [fh, fw] contiguous; y: [n, c, out_h, out_w] with strides (yn, yc, yh, yw), where
 *   out = (in*up + pad0 + pad1 - fsize + down) / down          (upfirdn2d.cpp:24-25)
 * Optional fused epilogue (all NULL/0 = plain reference op), applied after the FIR and gain:
 *   y = lrelu_or_linear( y + noise[n % noise_n, oy, ox] * *noise_strength + bias[c] ) * ep_gain
 * used by the synthesis engine to fold SynthesisLayer steps 5-6 (training/networks.py:1036-1040) into the
 * post-transposed-conv blur.
 */
typedef struct mgf_epilogue {
    const float* bias;            /* [c] or NULL */
    const float* noise;           /* [noise_n, out_h, out_w] or NULL */
    const float* noise_strength;  /* device scalar or NULL (=1) */
    int32_t noise_n;              /* 1 = broadcast over batch */
    int32_t act;                  /* MGF_ACT_LINEAR / MGF_ACT_LRELU / MGF_ACT_RELU */
    float alpha;
    float gain;                   /* post-activation gain */
    const float* residual;        /* same shape/strides as y, or NULL: y += residual after the gain */
} mgf_epilogue;

int mgf_upfirdn2d(void* y, const void* x, const float* f, int dtype,
                  int32_t n, int32_t c, int32_t in_h, int32_t in_w,
                  int64_t sn, int64_t sc, int64_t sh, int64_t sw,
                  int32_t out_h, int32_t out_w, int64_t yn, int64_t yc, int64_t yh, int64_t yw,
                  int32_t fh, int32_t fw, int32_t upx, int32_t upy, int32_t downx, int32_t downy,
                  int32_t padx0, int32_t padx1, int32_t pady0, int32_t pady1, int32_t flip, float gain,
                  const mgf_epilogue* ep, mgf_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Tap-list convolution on FP32 MFMA (v_mfma_f32_32x32x2_f32) -- the engine's replacement for the cuDNN calls
 * behind conv2d_resample / modulated_conv2d (torch_utils/ops/conv2d_resample.py:21-46,99-139;
 * training/networks.py:253-328).
 *
 *   acc[n, g, co, ty, tx] = sum_{t in group g} sum_{ci} wp[t][ci][co] *
 *                           ( x[n, ci, ty*istride + dy[t], tx*istride + dx[t]] * in_scale[n, ci] )   (0 outside x)
 *   y[n, co, ty*ostride + oy[g], tx*ostride + ox[g]] = epilogue( acc * out_scale[n, co] )
 *
 * - a 3x3 'same' correlation is 9 taps in one group (dy,dx in {-1,0,1});
 * - the stride-2 transposed conv of the up=2 path (conv2d_resample.py:117-130) is 9 taps in 4 parity groups
 *   with ostride 2, computed at the transposed-conv's own FLOP count;
 * - in_scale = style modulation, out_scale = demodulation (networks.py:288-293) applied in registers instead of
 *   materialising per-sample weights.
 * wp is the packed weight image produced by mgf_pack_conv_weights.  x,y are dense NCHW float32; y has an explicit
 * row pitch / plane / batch stride (in elements) and a channel offset so that it can be a padded workspace or a
 * channel slice of a concat buffer.
 */
#define MGF_MAX_TAPS 9
typedef struct mgf_conv_desc {
    int32_t n, cin, in_h, in_w;          /* input  [n, cin, in_h, in_w] */
    int32_t cout;                        /* real output channels */
    int32_t cout_pad;                    /* packed weight channel count (multiple of 32) */
    int32_t tile_h, tile_w;              /* extent of the (ty,tx) iteration space */
    int32_t istride, ostride;
    int32_t ntaps, ngroups;
    int32_t dy[MGF_MAX_TAPS], dx[MGF_MAX_TAPS], group[MGF_MAX_TAPS];
    int32_t oy[4], ox[4];
    int32_t out_h, out_w;                /* valid output extent (stores outside are dropped) */
    int64_t y_pitch, y_plane, y_batch;   /* element strides of y */
    int32_t y_choff;                     /* channel offset into y */
    int32_t out_scale_stride;            /* elements between samples in out_scale (0 = shared) */
    float* workspace;                    /* optional split-K scratch (device); NULL = never split the K dimension */
    int64_t workspace_floats;            /* capacity of workspace in floats */
    /* Optional fused 1x1 projection of the conv result (ToRGBLayer, training/networks.py:1054-1065, folded into conv_last):
     * when rgb_out != NULL the kernel does not write y at all but
     *   rgb_out[n, c, oy, ox] = sum_co rgb_w[n, c, co] * epilogue(acc)[co] + rgb_bias[c]        c < rgb_channels <= 4
     * rgb_w carries the per-sample style modulation (W[c,co] * s[n,co]).  Needs cout <= 32, one K slice, dense rgb_out. */
    const float* rgb_w;                  /* [n, rgb_channels, cout] */
    const float* rgb_bias;               /* [rgb_channels] or NULL */
    float* rgb_out;                      /* [n, rgb_channels, out_h, out_w] */
    int32_t rgb_channels;
    int32_t pad_;
} mgf_conv_desc;

int mgf_conv_taps_f32(float* y, const float* x, const float* wp, const float* in_scale, const float* out_scale,
                      const mgf_conv_desc* d, const mgf_epilogue* ep, mgf_stream_t stream);
/* The same convolution in the OPT-IN "bf16x3" arithmetic (not the reference's float32: an engine mode of its own, never the default).
 * Every float32 operand is split into two bfloat16 terms a = a1 + a2 (a1 = bf16(a), a2 = bf16(a - a1): 16 significant bits) and a product
 * runs as three v_mfma_f32_32x32x16_bf16 (a2 b1 + a1 b2 + a1 b1) with float32 accumulation -- 3 x 32 matrix-pipe cycles per 16 input
 * channels instead of 8 x 64; worst error 4 - 5e-6 of max|y| on the generator's layer shapes (tools/probes/bf16x3_gemm.hip).
 * wb: the weights split once per checkpoint: bfloat16 [cin / 16][term 2][tap 9][lane half 2][cout_pad][8 channels], from the float32
 * image of mgf_pack_conv_weights (gain folded in).  The style multiplies the ACTIVATIONS where they are staged (w (s x) instead of
 * (w s) x).  Serves 3x3 stride-1 convolutions and the 4-group transposed conv on maps at least 32 wide with cin % 16 == 0;
 * MGF_EUNSUPPORTED otherwise.  Descriptor, epilogue and fused ToRGB projection as in mgf_conv_taps_f32. */
int mgf_conv_taps_bf16x3_f32(float* y, const float* x, const void* wb, const float* in_scale, const float* out_scale,
                             const mgf_conv_desc* d, const mgf_epilogue* ep, mgf_stream_t stream);

/* Per-launch instrumentation of mgf_conv_taps_f32 for roofline accounting: between _begin and _end every conv launch is
 * bracketed by HIP events on its launch stream (main kernel only, not the split-K reduce; do not use during graph capture).
 * _end waits for the events and fills up to max_recs records in launch order; returns the number of launches seen. */
typedef struct mgf_conv_prof_rec {
    char kernel[64];      /* demangled kernel name as rocprofv3 prints it, e.g. "conv_taps_kernel<2, 2, 0, true, 9>" */
    double flops;         /* algorithmic FLOPs of the launch (transposed conv counted per input pixel) */
    double seconds;       /* event-to-event duration */
    double bytes;         /* algorithmic HBM bytes of the launch: input + output (the RGB image when ToRGB is fused) + weights, each once */
    int32_t ksplit;       /* K slices used (1 = no split) */
    int32_t pad_;
} mgf_conv_prof_rec;
int mgf_conv_profile_begin(void);
int mgf_conv_profile_end(mgf_conv_prof_rec* out, int32_t max_recs);

/* Winograd F(2x2,3x3) form of the 3x3 / stride-1 / pad-1 correlation (same operands and result as the 9-tap mgf_conv_taps_f32 launch
 * behind modulated_conv2d, training/networks.py:288-303, with 2.25x fewer matrix operations; results differ from the direct form
 * by float32 rounding only).
 *   winograd_weights: u[xi][ci / 8][co][slot] = gain * (G g G^T)[xi] for w [cout, cin, 3, 3] (xi = 0..15; the 8 channels of a chunk
 *                     are stored in MFMA operand order, slot 4*(c%2) + (c%8)/2), once per checkpoint; 16*cin*cout floats
 *   conv3x3_winograd: y[n, co] = epilogue( out_scale[n, co] * sum_ci (in_scale[n, ci] * w[co, ci]) (*) x[n, ci] ), dense NCHW,
 *                     cin % 8 == 0, cout % 64 == 0, h and w even (form 1); in_scale / out_scale / ep may be NULL */
int mgf_winograd_weights_f32(float* u, const float* w, int32_t cout, int32_t cin, float gain, mgf_stream_t stream);
/* Second form of the same operation (two 4-wave workgroups per CU, 32 output channels x 16x16 outputs each, 4-channel chunks):
 * u[xi][ci / 4][co][slot] with the 4 channels of a chunk in MFMA operand order (slot 2*(c%2) + (c%4)/2); cin % 4 == 0, cout % 32 == 0 */
int mgf_winograd2_weights_f32(float* u, const float* w, int32_t cout, int32_t cin, float gain, mgf_stream_t stream);
/* form 2 writing a channel slice of a wider output (y [n, C_total, h, w] with y_batch = C_total*h*w elements between samples, channels
 * [y_choff, y_choff + cout)) -- a SqueezeNet Fire module's expand3x3 half of the concat buffer; odd map sides are accepted here and in
 * mgf_conv3x3_winograd2_f32 (element-wise stores of the partial last quads) */
int mgf_conv3x3_winograd2_slice_f32(float* y, const float* x, const float* u, const float* in_scale, const float* out_scale, int32_t n,
                                    int32_t cin, int32_t h, int32_t w, int32_t cout, int32_t out_scale_stride, int64_t y_batch,
                                    int32_t y_choff, const mgf_epilogue* ep, mgf_stream_t stream);
/* form 2 with the fused 1x1 projection of mgf_conv_desc.rgb_* (ToRGB folded into conv_last, training/networks.py:1054-1065): needs
 * cout == 32; writes only rgb_out[n, c, h, w] = sum_co rgb_w[n, c, co] * (out_scale[n, co] * conv)[co] + rgb_bias[c], c < rgb_channels <= 4 */
int mgf_conv3x3_winograd2_rgb_f32(float* rgb_out, const float* x, const float* u, const float* in_scale, const float* out_scale,
                                  const float* rgb_w, const float* rgb_bias, int32_t n, int32_t cin, int32_t h, int32_t w, int32_t cout,
                                  int32_t out_scale_stride, int32_t rgb_channels, mgf_stream_t stream);
int mgf_conv3x3_winograd2_f32(float* y, const float* x, const float* u, const float* in_scale, const float* out_scale, int32_t n,
                              int32_t cin, int32_t h, int32_t w, int32_t cout, int32_t out_scale_stride, const mgf_epilogue* ep,
                              mgf_stream_t stream);
/* Third form of the same operation (csrc/wino3.hip), the one the engine uses for 32x32 maps and larger: the four rows of the transformed
 * patch go to the four waves of a workgroup, a lane computes exactly the transformed-input values of its own MFMA operand slots (no
 * LDS round trip of the transformed input), the weight operands come straight from L2, the style multiplies the input footprint.
 * u in the layout of mgf_winograd2_weights_f32; cin % 4 == 0, cout % 32 == 0 (64 output channels per workgroup when cout % 64 == 0),
 * dense y [n, cout, h, w] (odd map sides are accepted here too, with element-wise stores); epilogue as mgf_conv_taps_f32 (noise and
 * residual 8-byte aligned on even maps).
 * _rgb: the fused 1x1 projection of mgf_conv3x3_winograd2_rgb_f32 (cout == 32, rgb_channels <= 3). */
int mgf_conv3x3_winograd3_f32(float* y, const float* x, const float* u, const float* in_scale, const float* out_scale, int32_t n,
                              int32_t cin, int32_t h, int32_t w, int32_t cout, int32_t out_scale_stride, const mgf_epilogue* ep,
                              mgf_stream_t stream);
/* form 3 writing a channel slice of a wider output (y [n, C_total, h, w] with y_batch = C_total*h*w elements between samples, channels
 * [y_choff, y_choff + cout); the residual, when given, has the same layout) and/or on ODD map sides (element-wise stores of the pixel pairs):
 * a SqueezeNet Fire module's expand3x3 half of the concat buffer on the 255 / 127 / 63 px LPIPS maps */
int mgf_conv3x3_winograd3_slice_f32(float* y, const float* x, const float* u, const float* in_scale, const float* out_scale, int32_t n,
                                    int32_t cin, int32_t h, int32_t w, int32_t cout, int32_t out_scale_stride, int64_t y_batch,
                                    int32_t y_choff, const mgf_epilogue* ep, mgf_stream_t stream);
/* 1x1 convolution without modulation -- the resnet skip projection of a SynthesisBlock (training/networks.py:1102-1105: Conv2dLayer with
 * kernel_size 1 -> conv2d_resample.py:99-103 -> F.conv2d) and the SqueezeNet Fire squeeze / expand1x1 layers of LPIPS
 * (lpips/pretrained_networks.py:7-44, torchvision Fire) -- as a register-operand MFMA GEMM (csrc/pointwise.hip: no LDS staging):
 *   y[n, y_choff + co, p] = epilogue( sum_ci w[ci][co] * in_scale[n, ci] * x[n, ci, p] ),  p in [0, hw)
 * (in_scale may be NULL; with it this is the modulated, un-demodulated 1x1 conv of ToRGBLayer, training/networks.py:1054-1065)
 * w is the [cin][cout_pad] image mgf_pack_conv_weights makes of a 1x1 kernel; y may be a channel slice of a wider buffer (y_batch elements
 * between samples, 0 = dense; the epilogue's residual then has the same layout); epilogue as mgf_conv_taps_f32 minus the noise input. */
int mgf_conv1x1_f32(float* y, const float* x, const float* w, const float* in_scale, int32_t n, int32_t cin, int32_t hw, int32_t cout,
                    int32_t cout_pad, int64_t y_batch, int32_t y_choff, const mgf_epilogue* ep, mgf_stream_t stream);
/* tuning / tests: pin the 32-channel blocks a wave of mgf_conv1x1_f32 carries (1 or 2; 0 = by layer shape) */
int mgf_conv1x1_force_shape(int32_t channel_blocks);

/* The narrow ends of the LPIPS(squeeze) stem outside the fused stem kernel (lpips/pretrained_networks.py:7-44, features.0), as
 * streaming VALU kernels (csrc/narrow_conv.hip).
 *   conv3x3s2_few_inputs:   y[n,cout,oh,ow] = act(bias + conv3x3 / stride 2 / no padding of x[n,cin<=4,in_h,in_w]), oh = (in_h-3)/2+1;
 *                           w in the torch layout [cout][cin][3][3]
 *   tconv3x3s2_few_outputs: t[n,co<=4, 2i+kh, 2j+kw] += w[kh,kw][ci][co] x[n,ci,i,j] over the whole [2h+1] x [2w+1] output (row pitch
 *                           `pitch`, even; the layout mgf_conv_taps_f32 + mgf_tconv3x3s2_border_f32 write); wp in
 *                           mgf_pack_conv_weights' [tap][cin][cout_pad] layout, cout_pad >= 4 */
int mgf_conv3x3s2_few_inputs_f32(float* y, const float* x, const float* w, const float* bias, int32_t n, int32_t cin, int32_t in_h,
                                 int32_t in_w, int32_t cout, int32_t relu, mgf_stream_t stream);
int mgf_tconv3x3s2_few_outputs_f32(float* y, const float* x, const float* wp, int32_t n, int32_t cin, int32_t h, int32_t w, int32_t cout,
                                   int32_t cout_pad, int32_t pitch, int64_t y_plane, int64_t y_batch, mgf_stream_t stream);
/* form 3 with the epilogue's residual given at HALF resolution: residual_low [n, cout, h/2, w/2] is up-sampled 2x inside the epilogue with the
 * [1,3,3,1] (x) [1,3,3,1] / 16 filter of upfirdn2d.upsample2d(x, f, up=2) (padding [2,1,2,1], gain 4; torch_utils/ops/upfirdn2d.py:300-336) and
 * added after the gain, i.e. y = act(...) * gain + upsample2d(residual_low).  This is the resnet skip branch of a SynthesisBlock
 * (training/networks.py:1157-1160, 245-250, conv2d_resample.py:105-108) without its full-resolution tensor.  ep->residual must be NULL. */
int mgf_conv3x3_winograd3_up2res_f32(float* y, const float* x, const float* u, const float* in_scale, const float* out_scale,
                                     const float* residual_low, int32_t n, int32_t cin, int32_t h, int32_t w, int32_t cout,
                                     int32_t out_scale_stride, const mgf_epilogue* ep, mgf_stream_t stream);
/* tuning / test hook: pin the workgroup shape of the form-3 launches (0 = automatic choice; 21 = 64 channels x 32 tiles, 12 = 32 x 64,
 * 11 = 32 x 32 with three workgroups per CU); a shape that does not fit the call (21 with cout % 64 != 0) falls back to the automatic one */
int mgf_winograd3_force_shape(int32_t shape);
int mgf_conv3x3_winograd3_rgb_f32(float* rgb_out, const float* x, const float* u, const float* in_scale, const float* out_scale,
                                  const float* rgb_w, const float* rgb_bias, int32_t n, int32_t cin, int32_t h, int32_t w, int32_t cout,
                                  int32_t out_scale_stride, int32_t rgb_channels, mgf_stream_t stream);
int mgf_conv3x3_winograd_f32(float* y, const float* x, const float* u, const float* in_scale, const float* out_scale, int32_t n,
                             int32_t cin, int32_t h, int32_t w, int32_t cout, int32_t out_scale_stride, const mgf_epilogue* ep,
                             mgf_stream_t stream);

/* Last row (oy = 2h) and last column (ox = 2w) of the stride-2 transposed 3x3 conv output t [n, cout, 2h+1, pitch] from the same
 * operands as the 4-group mode of mgf_conv_taps_f32 (x [n,cin,h,w], packed taps, in_scale [n,cin] | NULL, out_scale | NULL).
 * With it the MFMA launch can tile exactly the h x w grid of 2x2 output quads (desc.tile_h = h, tile_w = w) instead of (h+1) x (w+1). */
int mgf_tconv3x3s2_border_f32(float* t, const float* x, const float* wp, const float* in_scale, const float* out_scale, int32_t n,
                              int32_t cin, int32_t h, int32_t w, int32_t cout, int32_t cout_pad, int64_t t_pitch, int64_t t_plane,
                              int64_t t_batch, int64_t out_scale_stride, mgf_stream_t stream);

/* Repack [cout, cin, kh, kw] float32 weights (times `gain`) into the [tap][cin][cout_pad] image read by
 * mgf_conv_taps_f32; `flip` reverses kh,kw (true convolution).  Taps are emitted in (kh, kw) row-major order.
 * Also emits wsq[cout, cin] = sum_k (w*gain)^2 when wsq != NULL (demodulation table). Device pointers. */
int mgf_pack_conv_weights(float* wp, float* wsq, const float* w, int32_t cout, int32_t cin, int32_t kh, int32_t kw,
                          int32_t cout_pad, float gain, int32_t flip, mgf_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Styles + demodulation coefficients -- FullyConnectedLayer affine + the `d` of modulated_conv2d
 * (training/networks.py:131-150, 288-291, 1022, 1056-1059).  For sample n and job j, with
 * wg = ws[n * ws_stride_n + j.w_offset ...] (the global latent component of that layer's ws slot):
 *   s[n, ci] = ((sum_k wg[k] * aff_w[ci, k]) * aff_gain + aff_b[ci]) * style_gain
 *   d[n, co] = rsqrt( sum_ci wsq[co, ci] * s[n, ci]^2 + 1e-8 )                   (if wsq and d are non-NULL)
 * wsq comes from mgf_pack_conv_weights.  The _multi form takes a DEVICE array of jobs (one launch for all layers); max_cin = the
 * largest job cin if the caller knows it (lets one workgroup serve all n <= 32 samples and read each wsq table once), else 0.
 */
typedef struct mgf_style_job {
    const float* aff_w;   /* [cin, wdim] */
    const float* aff_b;   /* [cin] */
    const float* wsq;     /* [cout, cin] or NULL */
    float* s;             /* [n, cin] */
    float* d;             /* [n, cout] or NULL */
    int32_t cin, cout;
    int32_t w_offset;
    float aff_gain, style_gain;
} mgf_style_job;
/* Demodulation alone, for callers that hold the styles as a tensor (the operator-level modulated_conv2d, networks.py:253-328, whose
 * `styles` argument is computed by the caller):  d[n,co] = rsqrt(sum_ci wsq[co,ci] s[n,ci]^2 + 1e-8), wsq from mgf_pack_conv_weights;
 * and its adjoint  ds[n,ci] = -s[n,ci] sum_co dd[n,co] d[n,co]^3 wsq[co,ci]  (what autograd runs through networks.py:288-291). */
int mgf_demod_f32(float* d, const float* s, const float* wsq, int32_t n, int32_t cin, int32_t cout, mgf_stream_t stream);
int mgf_demod_bwd_f32(float* ds, const float* dd, const float* d, const float* s, const float* wsq, int32_t n, int32_t cin, int32_t cout,
                      mgf_stream_t stream);
int mgf_style_demod(const mgf_style_job* job, const float* ws, int64_t ws_stride_n, int32_t n, int32_t wdim, mgf_stream_t stream);
int mgf_style_demod_multi(const mgf_style_job* jobs_dev, int32_t njobs, const float* ws, int64_t ws_stride_n,
                          int32_t n, int32_t wdim, int32_t max_cin, mgf_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Duplex (image <- latents) attention of a SynthesisLayer, k-means/parametric-centroid form, integration "mul",
 * layer norm -- replaces TransformerLayer.forward + integrate + att_norm (training/networks.py:748-822,657-672,
 * 341-358) and the noise / bias_act tail of SynthesisLayer.forward (:1036-1040).
 *
 * Re-associated (exact in real arithmetic; checked against the oracle to 1e-3 relative):
 *   S[f,t]  = sum_c x[n,c,f] * wqc[c,t] + spos[f,t]           wqc [c,t], spos [f,t]: checkpoint constants (engine.py)
 *   P       = softmax_t(S)
 *   g[f,c]  = sum_t P[f,t] * vwb[n,c,t]                        vwb = (V Wm^T + bm + 1), per sample (mgf_attn_values)
 *   y[n,c,f]= epilogue( x[n,c,f] * rsqrt(mean_c x^2 + 1e-8) * g[f,c] )
 * probs (optional) receives P as [n, f, t]; argmax (optional) receives the per-pixel latent assignment [n, f] int32.
 */
int mgf_duplex_attention(float* y, const float* x, const float* wqc, const float* spos, const float* vwb,
                         int32_t n, int32_t c, int32_t f, int32_t t,
                         const mgf_epilogue* ep, int32_t ep_w, float* probs, int32_t* argmax, mgf_stream_t stream);

/* SynthesisNetwork.list2tensor (training/networks.py:1222-1242) for one layer: probs [n, side*side, t] (what mgf_duplex_attention
 * writes) replicated nearest-neighbour to out_res x out_res (upsample2d with the all-ones kernel) into slice `layer` of the
 * stacked tensor out [n, t, n_layers, 1, out_res, out_res].  Only needed with return_att=True; the projection loop discards it. */
int mgf_att_map_upsample_f32(float* out, const float* probs, int32_t n, int32_t side, int32_t t, int32_t out_res, int32_t layer,
                             int32_t n_layers, mgf_stream_t stream);

/* vwb[n,c,t] = sum_k ycomp[n,t,k] * wmv[c,k] + bmv[c]   (wmv = Wm*Wv folded, bmv = Wm*bv + bm + 1: checkpoint constants;
 * ycomp[n,t,:] = ws[n*ws_stride_n + t*ws_stride_t + w_offset ...], the local latent components).  t fastest in vwb. */
typedef struct mgf_attn_job {
    const float* wmv;     /* [c, wdim] */
    const float* bmv;     /* [c] */
    float* vwb;           /* [n, c, t] */
    int32_t c;
    int32_t w_offset;
} mgf_attn_job;
int mgf_attn_values(const mgf_attn_job* job, const float* ws, int64_t ws_stride_n, int64_t ws_stride_t, int32_t n, int32_t t,
                    int32_t wdim, mgf_stream_t stream);
int mgf_attn_values_multi(const mgf_attn_job* jobs_dev, int32_t njobs, const float* ws, int64_t ws_stride_n, int64_t ws_stride_t,
                          int32_t n, int32_t t, int32_t wdim, mgf_stream_t stream);

/* Per-layer noise maps of noise_mode="random" (networks.py:1016-1017: torch.randn per layer and call) and the per-sample ToRGB weights of
 * the fused conv_last epilogue (networks.py:1056-1063), so that the captured launch sequence holds hand-written kernels only.
 *   randn:       out[0:n] ~ N(0,1): Philox4x32-10 keyed by `seed`, Box-Muller; `state` = 16 zero-initialised device bytes {uint64 stream
 *                position, uint32 ticket, pad} that the launch advances itself -- a replayed hipGraph draws fresh numbers every time
 *   rgb_weights: out[n, c, co] = w[c, co] * s[n, co] */
int mgf_randn_f32(float* out, int64_t n, uint64_t seed, void* state, mgf_stream_t stream);
int mgf_rgb_weights_f32(float* out, const float* w, const float* s, int32_t n, int32_t c, int32_t cout, mgf_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Mapping network z -> w  (MappingNetwork.forward, training/networks.py:894-942 with MLP/ResnetLayer :154-221 and the
 * latent self-attention TransformerLayer :748-822).  One workgroup per sample.
 * params: packed float32 blob, layout documented in morphganformer_amd/engine.py (pack_mapping_params).
 * z: [n, k, dim] ; w: [n, k, dim].  dim == 32, k <= 33.
 */
int64_t mgf_mapping_param_floats(int32_t k, int32_t dim, int32_t n_res_layers);
int mgf_mapping_forward(float* w, const float* z, const float* params, int32_t n, int32_t k, int32_t dim,
                        int32_t n_res_layers, int32_t normalize_global, mgf_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Losses of the projection loop, batched over n independent candidates (one generator forward evaluates n steps of the
 * literal loop; n = 1 is the reference's one-image-per-step form).
 * mse:   out[i] (+)= scale * mean((a[i] - b[i])^2), a: [n, numel], b: rows b_batch_stride elements apart (0 = one shared
 *        target)                                      torch.nn.MSELoss, 1024_example_wing_loss_perceptual_sqz_MSE.py:176
 * wing:  out[i] = WingLoss(pred row (*pred_step + i, clamped to max_row when max_row >= 0), target) in f64   wing_loss.py:19-28
 *        (pred is a [rows, numel] table; pred_step NULL = rows 0..n-1)
 * lpips unit:  out = f / (|f|_channels + 1e-10), f: [n,c,hw]  (normalize_tensor, lpips/__init__.py / networks_basic.py:70-80);
 *        run once per target image on each of its taps
 * lpips layer: out[i] (+)= mean_hw( sum_c lin[c] * (f0/(|f0|+1e-10) - f1_unit)^2 ), f0: [n,c,hw] raw taps of the candidates,
 *        f1_unit: UNIT-NORMALISED taps of the target (from mgf_lpips_unit_f32; same arithmetic, so identical images give
 *        exactly 0), samples f1_batch_stride elements apart (0 = one shared target)    lpips/networks_basic.py:70-87
 *        at most 512 channels per tap
 * All reduce through a deterministic two-stage reduction (no float atomics) using `scratch`
 * (>= n * mgf_reduce_scratch_floats() floats).
 */
int64_t mgf_reduce_scratch_floats(void);
int mgf_mse_f32(float* out, const float* a, const float* b, int32_t n, int64_t numel, int64_t b_batch_stride, float scale,
                int32_t accumulate, float* scratch, mgf_stream_t stream);
/* dssim: out[i] (+)= scale * (1 - SSIM(u8(img[i]), u8(target))) / 2 as float32 -- `dssim`, 1024_example_SSIM.py:115-117 (the same function as
 *        lpips/__init__.py:54-55): skimage's compare_ssim(data_range, multichannel=True) with its defaults (7x7 uniform window, sample
 *        covariance, K1 0.01, K2 0.03, the window positions whole inside the image, mean over positions, then over channels).
 *        u8(x) = clip(rint(x * 127.5 + 127.5), 0, 255): the image the drivers save (misc.to_pil, misc.py:115-116).  img [n,c,h,w] in [-1, 1],
 *        target [c,h,w] (t_batch_stride 0) or one per sample; h, w >= 7; scratch: mgf_dssim_scratch_bytes(n,c,h,w) bytes, 8-byte aligned.
 *        The window sums are exact integers, everything above them float64 in a fixed order (bit-reproducible). */
int64_t mgf_dssim_scratch_bytes(int32_t n, int32_t c, int32_t h, int32_t w);
int mgf_dssim_u8_f32(float* out, const float* img, const float* target, int32_t n, int32_t c, int32_t h, int32_t w, int64_t t_batch_stride,
                     float data_range, float scale, int32_t accumulate, void* scratch, mgf_stream_t stream);
/* The LBP matching distance of 1024_example_LBP_percept.py:34-58,162-166 per candidate, in three steps (csrc/lbp.hip):
 *   lbp_gray224:  gray [n,224,224] u8 = cv2.resize(cv2.cvtColor(to_pil(img), COLOR_BGR2GRAY), (224, 224)) of img [n,3,h,w] in [-1, 1]: misc.to_pil's
 *                 rint quantisation (misc.py:115-116), OpenCV's 8-bit gray weights (1868, 9617, 4899, >> 14) with the FIRST channel in the blue
 *                 slot -- the scripts hand an RGB array to a BGR conversion (:48) -- or, true_rgb_order != 0, in the red slot (what
 *                 cv2.imread(IMREAD_GRAYSCALE) does to the target file, :41); INTER_LINEAR in OpenCV's 11-bit fixed point.
 *                 tables: int32 [2][224][4] {index 0, index 1, coefficient 0, coefficient 1} for the 224 columns, then the 224 rows
 *                 (half-pixel centres; host-side, drivers.cv_resize_tables)
 *   lbp_codes:    codes [n,224,224] u8 = skimage.feature.local_binary_pattern(gray, 24, 3, 'uniform') (values 0..25), float64 in skimage's
 *                 expression order; offsets: float64 [2][24] = the sample offsets {rp, cp} rounded to 5 decimals (host-side)
 *   lbp_distance: out[i] = 1 - dot(x_i, y) / (sqrt(dot(x_i, x_i)) * sqrt(dot(y, y))), x_i the code map of gray[i], y = target_codes
 *                 [224*224] u8; float64, the dots exact integers.  scratch: mgf_lbp_scratch_bytes(n) bytes.
 * OpenCV and scikit-image are not part of the reference tree: their published algorithms are restated. */
int64_t mgf_lbp_scratch_bytes(int32_t n);
int mgf_lbp_gray224_u8(uint8_t* gray, const float* img, const int32_t* tables, int32_t n, int32_t h, int32_t w, int32_t true_rgb_order,
                       mgf_stream_t stream);
int mgf_lbp_codes_u8(uint8_t* codes, const uint8_t* gray, const double* offsets, int32_t n, mgf_stream_t stream);
int mgf_lbp_distance_f64(double* out, const uint8_t* gray, const uint8_t* target_codes, const double* offsets, int32_t n, void* scratch,
                         mgf_stream_t stream);
int mgf_wing_loss_f64(double* out, const double* pred, const double* target, int32_t n, int64_t numel, double omega, double epsilon,
                      const int32_t* pred_step, int32_t max_row, mgf_stream_t stream);
/* AdaptiveWingLoss(omega=14, theta=0.5, epsilon=1, alpha=2.1) of adaptive_wing_loss.py:12-39, same row addressing as the wing loss */
int mgf_adaptive_wing_loss_f64(double* out, const double* pred, const double* target, int32_t n, int64_t numel, double omega, double theta,
                               double epsilon, double alpha, const int32_t* pred_step, int32_t max_row, mgf_stream_t stream);
int mgf_lpips_unit_f32(float* out, const float* f, int32_t n, int32_t c, int64_t hw, mgf_stream_t stream);
int mgf_lpips_layer_f32(float* out, const float* f0, const float* f1_unit, const float* lin, int32_t n, int32_t c, int64_t hw,
                        int64_t f1_batch_stride, int32_t accumulate, float* scratch, mgf_stream_t stream);
/* The LPIPS(squeeze) stem in one pass: features.0 (conv 3->64, 3x3, stride 2, no padding; w: [64,27] in (ci,kh,kw) order and
 * b: [64], both with the ScalingLayer folded in) -> ReLU (= LPIPS tap 0) -> MaxPool 3x3/2 ceil_mode, written to
 * pooled [n,64,ph,pw].  Tap 0 itself never goes to memory (lpips/pretrained_networks.py:6-56, networks_basic.py:64-92):
 *   reference mode (feat_out != NULL): feat_out [n,64,ch,cw] = tap0 / (|tap0|_channels + 1e-10), kept for the target image;
 *   distance mode  (feat_ref != NULL): out[i] (+)= mean_hw( sum_c lin[c] * (tap0/(|tap0|+1e-10) - feat_ref)^2 ) against ONE
 *                  shared reference map feat_ref [64,ch,cw]; scratch as for mgf_lpips_layer_f32.
 * ch = (h-3)/2+1, ph = ceil((ch-3)/2)+1 (torch's ceil_mode rule), same for the widths. */
int mgf_lpips_stem_f32(float* pooled, const float* x, const float* w, const float* b, float* feat_out, const float* feat_ref,
                       const float* lin, float* out, int32_t n, int32_t h, int32_t w_in, int32_t accumulate, float* scratch,
                       mgf_stream_t stream);
/* y = max over a ksize x ksize window (2 or 3), stride 2, floor mode: out = (in - ksize)/2 + 1
 * (torchvision vgg16 features[4,9,16,23]: MaxPool2d(2,2); alexnet features[2,5]: MaxPool2d(3,2)) */
int mgf_maxpool_s2_floor_f32(float* y, const float* x, int32_t nc, int32_t in_h, int32_t in_w, int32_t ksize, mgf_stream_t stream);
/* y = max over a 3x3 window, stride 2, ceil_mode (torchvision SqueezeNet1.1 features[2,5,8]) */
int mgf_maxpool3x3s2_ceil_f32(float* y, const float* x, int32_t nc, int32_t in_h, int32_t in_w, int32_t out_h, int32_t out_w,
                              mgf_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Literal projection step bookkeeping (1024_example_wing_loss_perceptual_sqz_MSE.py:156-157,179,186-189), on device so the
 * loop never synchronises with the host.  `batch` consecutive steps are handled per call:
 *   perturb: latent_n[j] = latent_in + eps[s] * sigma[s],  s = min(*step + j, steps_total - 1)   (eps: [steps, numel] injected
 *            randn stream, sigma: [steps]); two roundings, like torch
 *   select:  for j in 0..batch-1 (in step order, s = *step + j < steps_total):
 *              total = (double)p_loss[j] + lamda * w_loss[j] + (double)(beta * mse[j])
 *            -- torch's promotion of `p_loss + args.lamda * w_loss + args.beta * mse_loss` (:179): lamda is a python double times a
 *            float64 tensor, beta a python scalar times a float32 tensor (float32 product) --
 *            if total < *min_loss: min_loss=total, best_latent=latent_n[j], best_step=s.
 *            losses_out[s] = total (NaN when valid[s] == 0 = "no face", ...sqz_MSE.py:165-166; valid NULL = always valid).
 *            Finally *step = min(*step + batch, steps_total).  All state lives on the device -> graph-replayable.
 *            Improvement trail (optional; the drivers save a PNG of the scored image at every improvement, :186-195): when
 *            `take_slot` != NULL, take_slot[j] = the trail slot candidate j's image belongs in (-1 = not an improvement), and
 *            trail_steps/trail_losses[slot] record the step and the loss; *trail_count counts improvements.  Slots run 0, 1, ...,
 *            capacity-1; once full, further improvements overwrite the LAST slot (the best-so-far is always kept).
 *   keep_improvements: trail_imgs[take_slot[j]] = imgs[j] for every j with take_slot[j] >= 0 (imgs: [batch, numel], the scored
 *            images of this batch; trail_imgs: [capacity, numel]).
 */
int mgf_latent_perturb(float* latent_n, const float* latent_in, const float* eps, const float* sigma, const int32_t* step,
                       int32_t batch, int32_t steps_total, int64_t numel, mgf_stream_t stream);
/* projection_example_v2_percept.py:131-166: `copies` noisy copies of the latent per step, averaged before the generator sees them --
 * latent_n[j, i] = torch.mean over c of (latent_in[i] + eps[s, c, i] * sigma[s]), s = *step + j, eps [steps, copies, numel]; the sum in
 * torch's CPU order (blocks of 16 rows, then the tail; 1 <= copies <= 255) so that the kept latent equals the script's bit for bit. */
int mgf_latent_perturb_mean(float* latent_n, const float* latent_in, const float* eps, const float* sigma, const int32_t* step,
                            int32_t batch, int32_t steps_total, int64_t numel, int32_t copies, mgf_stream_t stream);
int mgf_select_best(double* min_loss, float* best_latent, int32_t* best_step, double* losses_out,
                    const float* latent_n, int64_t numel, const float* p_loss, const double* w_loss, const float* mse_loss,
                    double lamda, float beta, int32_t* step, const int32_t* valid, int32_t batch, int32_t steps_total,
                    int32_t* take_slot, int32_t* trail_count, int32_t trail_capacity, int32_t* trail_steps, double* trail_losses,
                    mgf_stream_t stream);
int mgf_keep_improvements(float* trail_imgs, const float* imgs, int64_t numel, const int32_t* take_slot, int32_t batch,
                          mgf_stream_t stream);
/* uint8 HWC image = clip(rint(x*127.5+127.5), 0, 255) from CHW float (misc.to_pil, misc.py:114-123) */
int mgf_to_uint8_hwc(uint8_t* out, const float* img, int32_t c, int32_t h, int32_t w, mgf_stream_t stream);
/* The gray uint8 image the drivers hand to dlib for every generated image (...sqz_MSE.py:159-163): cv2.normalize(img, None, 0, 255,
 * NORM_MINMAX, CV_8U) over the whole float image -- u8 = clip(rint((double(x) - min) * (255.0 / (max - min)))), 0 for a flat image -- then
 * cv2.cvtColor(COLOR_BGR2GRAY) applied to RGB-ordered data: gray = (c0 * 1868 + c1 * 9617 + c2 * 4899 + (1 << 13)) >> 14.
 * img [n,3,h,w] float32 planar (16-byte aligned; a batch needs h*w % 4 == 0) -> gray [n,h,w]; scratch: n * mgf_reference_gray_scratch_floats()
 * floats.  Per candidate its own min / max, like the per-image call of the driver.  1 MB instead of 12.6 MB per candidate crosses to the host. */
int64_t mgf_reference_gray_scratch_floats(void);
int mgf_reference_gray_u8(uint8_t* gray, const float* img, int32_t n, int32_t h, int32_t w, float* scratch, mgf_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Biometric branch: the IResNet embedder (backbones/iresnet.py:28-161).  Its convolutions are mgf_conv_taps_f32 launches
 * (eval-mode BatchNorm folded into out_scale/bias, the identity shortcut into the residual port); these are the rest.
 *   channel_affine_prelu: y = prelu_c(x * scale[c] + shift[c]) on [n,c,hw]; scale / shift / slope may each be NULL
 *                         (BatchNorm2d in front of a zero-padded conv, iresnet.py:47-48; nn.PReLU(planes), :50)
 *   linear:               y[s,o] = b[o] + sum_i w[o,i] x[s,i]   (nn.Linear, :99,158), n <= 16 rows, in_features % 4 == 0
 *   resize_bilinear:      F.interpolate(x, (out_h,out_w), mode="bilinear", align_corners=False) on nc planes
 */
int mgf_channel_affine_prelu_f32(float* y, const float* x, const float* scale, const float* shift, const float* slope,
                                 int32_t n, int32_t c, int64_t hw, mgf_stream_t stream);
int mgf_linear_f32(float* y, const float* x, const float* w, const float* b, int32_t n, int32_t in_features, int32_t out_features,
                   mgf_stream_t stream);
int mgf_resize_bilinear_f32(float* y, const float* x, int32_t nc, int32_t in_h, int32_t in_w, int32_t out_h, int32_t out_w,
                            mgf_stream_t stream);
/* Head of the FaceNet embedder the biometric driver calls (facenet_pytorch.InceptionResnetV1, 1024_example_FaceNet_percept.py:30-32,
 * 147-158; its convolutions are mgf_conv_taps_f32 / mgf_conv1x1_f32 / Winograd launches with eval-mode BatchNorm folded in):
 *   spatial_mean:  y[p] = mean(x[p, 0:hw])                       (nn.AdaptiveAvgPool2d(1)), nc planes
 *   l2_normalize:  y[r] = x[r] / max(||x[r]||_2, eps)            (F.normalize(x, p=2, dim=1)), n rows of d floats */
int mgf_spatial_mean_f32(float* y, const float* x, int32_t nc, int64_t hw, mgf_stream_t stream);
int mgf_l2_normalize_f32(float* y, const float* x, int32_t n, int32_t d, float eps, mgf_stream_t stream);
/* gradient mode of the same network (what autograd runs through the package's Block35 / Block17 / Block8 / Mixed_6a / Mixed_7a modules):
 *   l2_normalize_bwd:  dx = (dy - y <y, dy>) / max(||x||, eps), y = x / max(||x||, eps)
 *   spatial_mean_bwd:  dx[p, i] = dy[p] / hw
 *   relu_bwd_slice:    dx[n, c, p] = y[n, y_choff + c, p] > 0 ? dy[n, dy_choff + c, p] : 0 -- the ReLU backward of ONE branch of a concat
 *                      buffer: dy / y are channel slices of [n, dy_channels | y_channels, hw] tensors, dx is dense [n, c, hw]
 * its convolution gradients are mgf_conv_taps_f32 / mgf_conv1x1_f32 launches on transposed taps. */
int mgf_l2_normalize_bwd_f32(float* dx, const float* dy, const float* x, int32_t n, int32_t d, float eps, mgf_stream_t stream);
int mgf_spatial_mean_bwd_f32(float* dx, const float* dy, int32_t nc, int64_t hw, mgf_stream_t stream);
int mgf_relu_bwd_slice_f32(float* dx, const float* dy, int32_t dy_channels, int32_t dy_choff, const float* y, int32_t y_channels,
                           int32_t y_choff, int32_t n, int32_t c, int64_t hw, mgf_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Gradient mode: dLoss/dlatent through the synthesis network -- what torch autograd does for the reference when the loss is
 * differentiated with respect to the latent (the north-star reading of the projection loop; bias_act.py:137-198,
 * upfirdn2d.py:237-256, networks.py:253-328, 748-822).  Weights are constants, so a layer needs
 *   - the activation gradient: a forward mgf_conv_taps_f32 launch on channel-transposed taps (dgrad) and mgf_upfirdn2d with up/down
 *     swapped;
 *   - the style gradient of  c_o = d_o * sum_i s_i (W_oi * x_i):   dL/ds_i = <x_i, g_i> - s_i * sum_o <dc_o, c_o> d_o^2 wsq[o,i]
 *     with g_i = sum_o W_oi^T * (d_o dc_o) the un-modulated dgrad result;
 *   - the gradients of the duplex attention with respect to its input and its value table.
 * All per-channel reductions are deterministic two-stage sums: a launch writes `mgf_bwd_chunks(hw)` partials per (sample, channel),
 * the consumer (mgf_style_demod_bwd_multi) adds them in order.
 *
 * layer_act_bwd: for y = lrelu_alpha(c + noise * strength + bias) * gain (+ residual), given dy:
 *     dz = dy * gain * (y - residual > 0 ? 1 : alpha)                              (may alias dy)
 *     dot_part[n, ch, chunk] = sum dz * c   with c recovered from y by inverting the activation     (skipped when dot_part == NULL)
 * channel_dot:   dot_part[n, ch, chunk] = sum a * b
 * style_grad:    dot_part[n, ch, chunk] = sum x * g;   dx (+)= s[n, ch] * g      (s NULL = 1; accumulate != 0 adds to dx)
 */
int32_t mgf_bwd_chunks(int64_t hw);
int mgf_layer_act_bwd_f32(float* dz, float* dot_part, const float* dy, const float* y, const float* residual, const float* bias,
                          const float* noise, const float* noise_strength, int32_t noise_n, int32_t n, int32_t c, int64_t hw,
                          float alpha, float gain, mgf_stream_t stream);
int mgf_channel_dot_f32(float* dot_part, const float* a, const float* b, int32_t n, int32_t c, int64_t hw, mgf_stream_t stream);
int mgf_style_grad_f32(float* dot_part, float* dx, const float* x, const float* g, const float* s, int32_t n, int32_t c, int64_t hw,
                       int32_t accumulate, mgf_stream_t stream);
/* mgf_style_grad_f32 of a layer fused with mgf_layer_act_bwd_f32 of the layer BEFORE it, whose output y is this layer's input x (conv1 ->
 * conv0 of a SynthesisBlock; the earlier layer must have no residual): style_part = <x, g> per (n, c, chunk), dz = (s g) gain
 * (x > 0 ? 1 : alpha), dot_part (may be NULL) = <dz, c> as in mgf_layer_act_bwd_f32 -- bit-identical to the two calls in sequence,
 * without the intermediate s g ever reaching memory.  `residual` and `dx` (both or neither) cover an earlier layer WITH a residual
 * (conv_last -> conv1 of the last block): the activation is inverted on x - residual and s g is also written to dx, which the block's
 * skip branch reads. */
int mgf_style_grad_act_bwd_f32(float* style_part, float* dot_part, float* dz, float* dx, const float* x, const float* g, const float* s,
                               const float* residual, const float* residual_low, int32_t w, const float* bias, const float* noise,
                               const float* noise_strength, int32_t noise_n, int32_t n, int32_t c, int64_t hw, float alpha, float gain,
                               mgf_stream_t stream);
/* The same pair followed by the ADJOINT OF THE 4x4 RESAMPLE BLUR (upfirdn2d with pad 2, upfirdn2d.py:237-256) in one pass, for an earlier
 * layer that is an up-sampling layer without attention and without residual: dt [n, c, h + 1, w + 1] = upfirdn2d(dz, f, pad 2, gain
 * fir_gain, flip) -- the map the stride-2 data-gradient convolution reads; dz itself never reaches memory.  f: the 4x4 OUTER-PRODUCT filter
 * (device, as upfirdn2d.setup_filter builds it).  The dot-product partials come one per 64 x 64 output tile: style_part / dot_part are
 * [n, c, mgf_style_act_fir_tiles(h, w)] (dot_part may be NULL). */
int32_t mgf_style_act_fir_tiles(int32_t h, int32_t w);
int mgf_style_act_fir_bwd_f32(float* style_part, float* dot_part, float* dt, const float* x, const float* g, const float* s, const float* bias,
                              const float* noise, const float* noise_strength, int32_t noise_n, const float* f, int32_t flip, float fir_gain,
                              int32_t n, int32_t c, int32_t h, int32_t w, float alpha, float gain, mgf_stream_t stream);
/* residual_low (here and in mgf_layer_act_bwd_low_f32; instead of `residual`, maps of row length w): the residual BEFORE its 2x FIR
 * up-sampling, [n][c][h/2][w/2] -- the resnet skip branch as the form-3 Winograd epilogue consumes it in the forward
 * (mgf_conv3x3_winograd3_f32's residual_low); it is up-sampled here with that epilogue's arithmetic, so the full-resolution skip
 * tensor exists in neither pass. */
int mgf_layer_act_bwd_low_f32(float* dz, float* dot_part, const float* dy, const float* y, const float* residual, const float* residual_low,
                              int32_t w, const float* bias, const float* noise, const float* noise_strength, int32_t noise_n, int32_t n,
                              int32_t c, int64_t hw, float alpha, float gain, mgf_stream_t stream);
/* Backward of mgf_duplex_attention without its epilogue (apply mgf_layer_act_bwd_f32 first), same operands as the forward:
 *   dx[n,c,f]    gradient with respect to the attention input x
 *   dg[n,c,f]    = da * x * rsqrt(mean_c x^2 + 1e-8), scratch consumed by mgf_attn_values_grad (may be NULL)
 *   probs[n,f,t] the recomputed softmax (may be NULL)
 * attn_values_grad: dvwb[n,c,t] = sum_f dg[n,c,f] * probs[n,f,t], the gradient of the per-sample value table. */
int mgf_duplex_attention_bwd(float* dx, float* dg, float* probs, const float* da, const float* x, const float* wqc, const float* spos,
                             const float* vwb, int32_t n, int32_t c, int32_t f, int32_t t, mgf_stream_t stream);
int mgf_attn_values_grad(float* dvwb, const float* dg, const float* probs, int32_t n, int32_t c, int32_t f, int32_t t, mgf_stream_t stream);
/* the same result through a register-operand MFMA GEMM over pixel slices + a fixed-order reduce (csrc/backward.hip); `workspace` holds the
 * per-slice partial sums (mgf_attn_values_grad_workspace_floats(n, c) floats, 16-byte aligned).  Falls back to the kernel above when
 * t != 16, the pixel count is not a multiple of 8 or the workspace is missing / too small. */
int64_t mgf_attn_values_grad_workspace_floats(int32_t n, int32_t c);
int mgf_attn_values_grad_ws(float* dvwb, const float* dg, const float* probs, int32_t n, int32_t c, int32_t f, int32_t t, float* workspace,
                            int64_t workspace_floats, mgf_stream_t stream);
/* Latent side.  style_demod_bwd_multi: per job (= modulated layer) and sample, from the partial dots above,
 *   ds[i]  = sum_chunks ds_part[n,i,:] - s[n,i] * sum_o (sum_chunks dc_part[n,o,:]) d[n,o]^2 wsq[o,i]      (second term only with demod)
 *   dwg[n, job, k] = aff_gain * style_gain * sum_i ds[i] * aff_w[i, k]          gradient wrt the global latent component
 * attn_values_bwd_multi: dyc[n, job, t, k] = sum_c dvwb[n,c,t] * wmv[c,k]       gradient wrt the local latent components
 * latent_grad_gather:    dw[n, t < k-1, :] = scale * sum_jobs dyc,  dw[n, k-1, :] = scale * sum_jobs dwg      (w layout [n, k, wdim]) */
typedef struct mgf_style_bwd_job {
    const float* aff_w;    /* [cin, wdim] */
    const float* wsq;      /* [cout, cin] or NULL */
    const float* s;        /* [n, cin] */
    const float* d;        /* [n, cout] or NULL */
    const float* ds_part;  /* [n, cin, s_chunks] */
    const float* dc_part;  /* [n, cout, d_chunks] or NULL */
    int32_t cin, cout, s_chunks, d_chunks;
    float aff_gain, style_gain;
} mgf_style_bwd_job;
typedef struct mgf_attn_bwd_job {
    const float* wmv;      /* [c, wdim] */
    const float* dvwb;     /* [n, c, t] */
    int32_t c;
    int32_t pad_;
} mgf_attn_bwd_job;
int mgf_style_demod_bwd_multi(float* dwg, const mgf_style_bwd_job* jobs_dev, int32_t njobs, int32_t n, int32_t wdim, int32_t max_channels,
                              mgf_stream_t stream);
int mgf_attn_values_bwd_multi(float* dyc, const mgf_attn_bwd_job* jobs_dev, int32_t njobs, int32_t n, int32_t t, int32_t wdim,
                              mgf_stream_t stream);
/* mgf_style_demod_bwd_multi and mgf_attn_values_bwd_multi in ONE launch (grid: style jobs + attention jobs): the two do not depend on each
 * other and are single-workgroup latency chains at one sample */
int mgf_latent_bwd_multi(float* dwg, const mgf_style_bwd_job* style_jobs_dev, int32_t n_style_jobs, float* dyc,
                         const mgf_attn_bwd_job* attn_jobs_dev, int32_t n_attn_jobs, int32_t n, int32_t t, int32_t wdim, int32_t max_channels,
                         mgf_stream_t stream);
/* The per-attention-layer by-products of the backward pass that nothing on the critical path waits for -- the value gradient
 * dvwb[c][t] = sum_f dg[c][f] P[f][t] (mgf_attn_values_grad_ws) and the demodulation partials <dc, c> (mgf_channel_dot_f32) -- for ALL attention
 * layers in two launches at the end of the pass instead of three small launches per layer (33 -> 2 at 1024^2).  jobs_dev: njobs records of
 * mgf_attn_grad_job_bytes() bytes { const float* dg, probs, dc, cpre; float* part, dvwb, dc_part; int32 c, f, slices, nchunk, blk_grad, blk_dot,
 * blk_red, pad } (dc_part NULL: no demodulation; slices from mgf_attn_values_grad_slices, > 0 required; part: n * slices * c * 16 floats;
 * blk_*: the layer's first workgroup in the flat grids of grad_blocks / dot_blocks / reduce_blocks workgroups). */
int32_t mgf_attn_values_grad_slices(int32_t n, int32_t c, int32_t f, int32_t t);
int64_t mgf_attn_grad_job_bytes(void);
int mgf_attn_grad_multi(const void* jobs_dev, int32_t njobs, int32_t n, int32_t grad_blocks, int32_t dot_blocks, int32_t reduce_blocks,
                        mgf_stream_t stream);
int mgf_latent_grad_gather(float* dw, const float* dwg, int32_t n_style_jobs, const float* dyc, int32_t n_attn_jobs, int32_t n, int32_t k,
                           int32_t wdim, float scale, mgf_stream_t stream);
/* Backward of mgf_mapping_forward: dz[n,k,dim] from dw[n,k,dim].  The forward is recomputed; its per-layer activations go to
 * `scratch` (>= n * mgf_mapping_bwd_scratch_floats(k, dim, n_res_layers) floats). */
int64_t mgf_mapping_bwd_scratch_floats(int32_t k, int32_t dim, int32_t n_res_layers);
int mgf_mapping_backward(float* dz, const float* dw, const float* z, const float* params, float* scratch, int32_t n, int32_t k,
                         int32_t dim, int32_t n_res_layers, int32_t normalize_global, mgf_stream_t stream);
/* The same pair without the recomputation: mgf_mapping_forward_save is mgf_mapping_forward that also fills `scratch` (same size
 * and layout) with the activations; mgf_mapping_backward_saved reads them for the same z. */
int mgf_mapping_forward_save(float* w, const float* z, const float* params, float* scratch, int32_t n, int32_t k, int32_t dim,
                             int32_t n_res_layers, int32_t normalize_global, mgf_stream_t stream);
int mgf_mapping_backward_saved(float* dz, const float* dw, const float* z, const float* params, float* scratch, int32_t n, int32_t k,
                               int32_t dim, int32_t n_res_layers, int32_t normalize_global, mgf_stream_t stream);

/* Loss side of gradient mode (what autograd does through lpips/networks_basic.py:64-92 and torch.nn.MSELoss):
 *   lpips_layer_bwd:   df0 (+)= d/df0 [ scale * mean_hw sum_c lin[c] (f0/(|f0|+1e-10) - f1_unit)^2 ], operands as mgf_lpips_layer_f32;
 *                      a pixel whose channels are all zero gets a zero gradient (autograd: NaN)
 *   relu_bwd_split:    dz = (y > 0 ? dy : 0); channels [0, c_split) go to dz_a [n, c_split, hw], the rest to dz_b [n, c - c_split, hw]
 *                      (the two halves of a SqueezeNet Fire concat; c_split == c and dz_a == dy is the plain in-place ReLU backward)
 *   maxpool3x3s2_ceil_bwd: dx of mgf_maxpool3x3s2_ceil_f32; a window's gradient goes to its first maximum in row-major order (torch)
 *   mse_grad:          d (+)= scale * 2 (a - b) / numel, operands as mgf_mse_f32 */
int mgf_lpips_layer_bwd_f32(float* df0, const float* f0, const float* f1_unit, const float* lin, int32_t n, int32_t c, int64_t hw,
                            int64_t f1_batch_stride, float scale, int32_t accumulate, mgf_stream_t stream);
/* mgf_lpips_layer_bwd_f32 followed by mgf_relu_bwd_split_f32 on the same tap (every LPIPS tap is a ReLU output f0) in one pass:
 * dz = f0 > 0 ? dy + d/df0[...] : 0 with dy the gradient arriving from the layers behind the tap (NULL: none), split like
 * relu_bwd_split (dz_b NULL with c_split == c: no split; dz_a == dy: in place). */
int mgf_lpips_layer_bwd_relu_f32(float* dz_a, float* dz_b, const float* dy, const float* f0, const float* f1_unit, const float* lin,
                                 int32_t n, int32_t c, int32_t c_split, int64_t hw, int64_t f1_batch_stride, float scale,
                                 mgf_stream_t stream);
/* The pair that shares the per-pixel sums: mgf_lpips_layer_stats_f32 is mgf_lpips_layer_f32 that also writes stats [n][3][hw] =
 * (sum_c f0^2, sum_c lin f0^2, sum_c lin f1_unit f0) (NULL: none); mgf_lpips_layer_bwd_relu_stats_f32 reads them instead of sweeping
 * both feature maps for them (NULL: as mgf_lpips_layer_bwd_relu_f32). */
int mgf_lpips_layer_stats_f32(float* out, float* stats, const float* f0, const float* f1_unit, const float* lin, int32_t n, int32_t c,
                              int64_t hw, int64_t f1_batch_stride, int32_t accumulate, float* scratch, mgf_stream_t stream);
/* ... and the same tap WITHOUT its finish launch: the partial sums stay in `scratch` (one set of n * mgf_reduce_scratch_floats() floats per tap,
 * *nparts_out partials per sample; a HOST pointer), and mgf_lpips_finish_taps_f32 adds up to 8 taps to out in tap order -- the same sums in the
 * same order as one finish per tap, in one launch (nparts / scales are host arrays; scale = 1 / hw of the tap) */
int mgf_lpips_layer_defer_f32(float* scratch, float* stats, const float* f0, const float* f1_unit, const float* lin, int32_t n, int32_t c, int64_t hw,
                              int64_t f1_batch_stride, int32_t* nparts_out, mgf_stream_t stream);
int mgf_lpips_finish_taps_f32(float* out, const float* scratch, int64_t set_stride_floats, int32_t ntaps, const int32_t* nparts, const float* scales,
                              int32_t n, int32_t accumulate, mgf_stream_t stream);
int mgf_lpips_layer_bwd_relu_stats_f32(float* dz_a, float* dz_b, const float* dy, const float* f0, const float* f1_unit, const float* lin,
                                       const float* stats, int32_t n, int32_t c, int32_t c_split, int64_t hw, int64_t f1_batch_stride,
                                       float scale, mgf_stream_t stream);
int mgf_relu_bwd_split_f32(float* dz_a, float* dz_b, const float* dy, const float* y, int32_t n, int32_t c, int32_t c_split, int64_t hw,
                           mgf_stream_t stream);
/* The same gradient from tap indices stored by the forward (mgf_maxpool3x3s2_ceil_idx_f32: y as mgf_maxpool3x3s2_ceil_f32 plus, per
 * output, the row-major index 0..8 of the window's first maximum): neither the input map nor the window scan is needed. */
int mgf_maxpool3x3s2_ceil_idx_f32(float* y, uint8_t* idx, const float* x, int32_t nc, int32_t in_h, int32_t in_w, int32_t out_h,
                                  int32_t out_w, mgf_stream_t stream);
int mgf_maxpool3x3s2_ceil_bwd_idx_f32(float* dx, const float* dy, const uint8_t* idx, int32_t nc, int32_t in_h, int32_t in_w, int32_t out_h,
                                      int32_t out_w, mgf_stream_t stream);
int mgf_maxpool3x3s2_ceil_bwd_f32(float* dx, const float* dy, const float* x, int32_t nc, int32_t in_h, int32_t in_w, int32_t out_h,
                                  int32_t out_w, mgf_stream_t stream);
/* dx of mgf_maxpool_s2_floor_f32 (ksize 2 or 3), first-maximum rule */
int mgf_maxpool_s2_floor_bwd_f32(float* dx, const float* dy, const float* x, int32_t nc, int32_t in_h, int32_t in_w, int32_t ksize,
                                 mgf_stream_t stream);
int mgf_mse_grad_f32(float* d, const float* a, const float* b, int32_t n, int64_t numel, int64_t b_batch_stride, float scale,
                     int32_t accumulate, mgf_stream_t stream);

/* Gradient mode of the biometric branch (autograd through backbones/iresnet.py:46-58,145-160 and F.interpolate); its convolution
 * gradients are mgf_conv_taps_f32 launches on transposed taps, BatchNorm gradients are mgf_channel_affine_prelu_f32 with the scale only.
 *   prelu_bwd:           dx = dy * (y > 0 ? 1 : slope[c]) from the PReLU OUTPUT y -- needs positive slopes (sign(y) = sign(input))
 *   linear_bwd:          dx[s,i] = sum_o dy[s,o] w[o,i]                        (n <= 16 rows, n * out_features floats <= 64 KiB)
 *   resize_bilinear_bwd: dx += adjoint of mgf_resize_bilinear_f32 applied to dy; dx [nc,in_h,in_w] must be zeroed (or hold the sum to
 *                        add to) beforehand; deterministic when the map is shrunk by >= 2x (no two outputs share a source pixel) */
int mgf_prelu_bwd_f32(float* dx, const float* dy, const float* y, const float* slope, int32_t n, int32_t c, int64_t hw, mgf_stream_t stream);
int mgf_linear_bwd_f32(float* dx, const float* dy, const float* w, int32_t n, int32_t in_features, int32_t out_features, mgf_stream_t stream);
int mgf_resize_bilinear_bwd_f32(float* dx, const float* dy, int32_t nc, int32_t in_h, int32_t in_w, int32_t out_h, int32_t out_w,
                                mgf_stream_t stream);

/* Landmark-Delaunay warp post-process (1024_warp_morphs.py:78-113,163-210): per triangle of the mesh the script pastes
 * cv2.warpAffine(patch, INTER_LINEAR, BORDER_REFLECT_101) through a cv2.fillConvexPoly mask, later triangles over earlier ones.
 *   label [h,w] int32:  the index of the triangle that wrote each pixel LAST, -1 = none (drivers.warp_plan rasterises OpenCV's polygon fill
 *                       -- Bresenham outline + 16.16 scanline edges -- on the host: integer control flow over a few hundred scanlines);
 *   triangles:          ntri records of mgf_cv_warp_triangle_bytes() bytes, 8-byte aligned: { double im[6] -- the INVERTED 2x3 matrix
 *                       warpAffine works with; int32 dx, dy -- origin of the destination patch; int32 sx, sy, sw, sh -- the source patch };
 *   out [c,h,w]:        every labelled pixel = OpenCV's value -- coordinates on the 1/1024 fixed-point grid rounded to 1/32 pixel
 *                       (INTER_BITS 5, AB_BITS 10, round_delta 16), the 32 x 32 float weight table, four products summed left to right in float,
 *                       BORDER_REFLECT_101 at the source PATCH -- the others `background` (the script's imgMorph starts at 0).
 * OpenCV is absent offline: parity with the real library is unpinned; oracle/warp_ref.py transcribes the same published sources. */
int64_t mgf_cv_warp_triangle_bytes(void);
int mgf_cv_warp_triangles_f32(float* out, const float* src, const int32_t* label, const void* triangles, int32_t ntri, int32_t c, int32_t h,
                              int32_t w, float background, mgf_stream_t stream);

/* torch.optim.Adam.step() on the latent (1024_example_wing_loss_perceptual_sqz_MSE.py:146,181-184; defaults betas (0.9, 0.999),
 * eps 1e-8; weight_decay 1e-4 in 1024_example_MSE.py:117), device-resident: lr = lr_table[*step] (the get_lr schedule, :63-68),
 * nothing happens when valid[*step] == 0 (the "no face" `continue`, :165-166) or *step >= steps_total; *adam_t is the optimizer's own
 * step count (bias corrections 1 - beta^t in double, like torch). */
int mgf_adam_step_f32(float* param, float* exp_avg, float* exp_avg_sq, int32_t* adam_t, const float* grad, const float* lr_table,
                      const int32_t* step, const int32_t* valid, int64_t numel, int32_t steps_total, float beta1, float beta2, float eps,
                      float weight_decay, mgf_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MGF_H_ */
