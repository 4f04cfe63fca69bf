"""Host wrappers around the FP32-MFMA tap-list convolution (mgf_conv_taps_f32 / mgf_pack_conv_weights, include/mgf.h).

These are the building blocks shared by the reference-compatible operator API (torch_utils/ops/conv2d_resample.py) and
the synthesis engine (engine.py), which packs weights once per checkpoint instead of per call.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass

import torch

from . import _lib


def round_up(v: int, m: int) -> int:
    return (v + m - 1) // m * m


def profile_begin():
    """Start bracketing every conv launch with HIP events on its stream (csrc/conv_taps.hip ProfScope); eager mode only."""
    _lib.check(_lib.lib().mgf_conv_profile_begin(), "conv_profile_begin")


def profile_end(max_recs=4096):
    """-> list of (kernel_name, algorithmic_flops, seconds, ksplit, algorithmic_bytes) in launch order."""
    recs = (_lib.ConvProfRec * max_recs)()
    n = _lib.lib().mgf_conv_profile_end(recs, max_recs)
    if n < 0:
        _lib.check(n, "conv_profile_end")
    return [(recs[i].kernel.decode(), recs[i].flops, recs[i].seconds, recs[i].ksplit, recs[i].bytes) for i in range(min(n, max_recs))]


@dataclass
class PackedConv:
    wp: torch.Tensor            # [taps, cin, cout_pad]
    wsq: torch.Tensor | None    # [cout, cin]
    cout: int
    cin: int
    kh: int
    kw: int
    cout_pad: int


def pack_weights(w: torch.Tensor, gain: float = 1.0, flip: bool = False, want_wsq: bool = False) -> PackedConv:
    """[cout, cin, kh, kw] -> tap-major image (+ demodulation table).  flip=True reverses kh,kw (true convolution)."""
    _lib.require_gpu(w)
    w = w.contiguous().float()
    cout, cin, kh, kw = w.shape
    cout_pad = round_up(cout, 32)
    wp = torch.empty([kh * kw, cin, cout_pad], dtype=torch.float32, device=w.device)
    wsq = torch.empty([cout, cin], dtype=torch.float32, device=w.device) if want_wsq else None
    rc = _lib.lib().mgf_pack_conv_weights(wp.data_ptr(), _lib.ptr(wsq), w.data_ptr(), cout, cin, kh, kw, cout_pad,
                                          float(gain), int(flip), _lib.stream_ptr())
    _lib.check(rc, "pack_conv_weights")
    return PackedConv(wp, wsq, cout, cin, kh, kw, cout_pad)


def pack_weights_bf16x3(pc: PackedConv) -> torch.Tensor:
    """The weight operand of the opt-in "bf16x3" arithmetic (mgf_conv_taps_bf16x3_f32): the float32 tap-major image of a 3x3 layer split into
    two bfloat16 terms w = w1 + w2 (w1 = bf16(w), w2 = bf16(w - w1), round to nearest even) and laid out
    [cin / 16][term 2][tap 9][lane half 2][cout_pad][8 channels] -- once per checkpoint."""
    assert pc.kh * pc.kw == 9 and pc.cin % 16 == 0, (pc.kh, pc.kw, pc.cin)
    w = pc.wp                                                          # [9, cin, cout_pad]
    hi = w.to(torch.bfloat16)
    mid = (w - hi.float()).to(torch.bfloat16)
    t = torch.stack([hi, mid]).view(2, 9, pc.cin // 16, 2, 8, pc.cout_pad)
    return t.permute(2, 0, 1, 3, 5, 4).contiguous()                    # [cin/16][term][tap][half][cout_pad][8]


def _desc(n, cin, in_h, in_w, cout, cout_pad, tile_h, tile_w, istride, ostride, taps, groups, oy, ox, out_h, out_w,
          y_pitch, y_plane, y_batch, y_choff=0, out_scale_stride=0):
    d = _lib.ConvDesc()
    d.n, d.cin, d.in_h, d.in_w, d.cout, d.cout_pad = n, cin, in_h, in_w, cout, cout_pad
    d.tile_h, d.tile_w, d.istride, d.ostride = tile_h, tile_w, istride, ostride
    d.ntaps, d.ngroups = len(taps), (max(groups) + 1 if groups else 1)
    for i, (dy, dx) in enumerate(taps):
        d.dy[i], d.dx[i] = dy, dx
        d.group[i] = groups[i] if groups else 0
    for i in range(4):
        d.oy[i] = oy[i] if i < len(oy) else 0
        d.ox[i] = ox[i] if i < len(ox) else 0
    d.out_h, d.out_w, d.y_pitch, d.y_plane, d.y_batch = out_h, out_w, y_pitch, y_plane, y_batch
    d.y_choff, d.out_scale_stride = y_choff, out_scale_stride
    ws = _workspace(torch.cuda.current_device())
    d.workspace, d.workspace_floats = ws.data_ptr(), ws.numel()
    return d


# the <= 4-channel ends of the LPIPS stem on the streaming kernels of csrc/narrow_conv.hip (0: the MFMA tap-list kernel; experiments, tests)
NARROW_CONV = os.environ.get("MGF_NARROW_CONV", "1") != "0"
POINTWISE = os.environ.get("MGF_POINTWISE", "1") != "0"                   # tuning hook: 0 = 1x1 layers on the tap-list kernel

# Split-K scratch: one slab per (device, stream) -- launches on one stream are ordered, launches on different streams may overlap.
WORKSPACE_FLOATS = 16 << 20
_WS = {}


def _workspace(dev_index):
    """Split-K scratch of the CURRENT stream (launches on different streams may overlap, so each stream has its own slab)."""
    key = (dev_index, torch.cuda.current_stream(dev_index).cuda_stream)
    ws = _WS.get(key)
    if ws is None:
        ws = torch.empty(WORKSPACE_FLOATS, dtype=torch.float32, device=torch.device("cuda", dev_index))
        _WS[key] = ws
    return ws


def conv_forward(x, pc: PackedConv, stride=1, pad=(0, 0), in_scale=None, out_scale=None, epilogue=None, out=None,
                 out_choff=0, rgb=None, taps=None, ksize=None, bf=None):
    """Correlation with the packed taps: y[oy,ox] = sum w[kh,kw] x[oy*stride + kh - pad_y, ox*stride + kw - pad_x].

    `out` may be a larger [n, C_total, oh, ow] buffer; this conv then writes channels [out_choff, out_choff + cout).
    bf: pack_weights_bf16x3(pc) -> the launch runs in the opt-in bf16x3 arithmetic (3x3 stride 1 only; mgf_conv_taps_bf16x3_f32)."""
    _lib.require_gpu(x, pc.wp, in_scale, out_scale, out, bf)
    launch = _lib.lib().mgf_conv_taps_f32 if bf is None else _lib.lib().mgf_conv_taps_bf16x3_f32
    wptr = pc.wp.data_ptr() if bf is None else bf.data_ptr()
    assert x.dtype == torch.float32 and x.is_contiguous() and x.ndim == 4 and x.shape[1] == pc.cin
    n, cin, h, w = x.shape
    py, px = pad
    kh_, kw_ = ksize if ksize is not None else (pc.kh, pc.kw)      # ksize/taps: `pc` holds a SUBSET of a larger kernel's taps
    oh = (h + 2 * py - kh_) // stride + 1
    ow = (w + 2 * px - kw_) // stride + 1
    if taps is None:
        taps = [(kh - py, kw - px) for kh in range(pc.kh) for kw in range(pc.kw)]
    assert len(taps) == pc.kh * pc.kw
    if rgb is not None:
        # fused 1x1 projection (ToRGB folded into the conv): rgb = (rgb_w [n,c,cout], rgb_bias [c] | None, rgb_out [n,c,oh,ow]);
        # the conv result itself is not written
        rgb_w, rgb_b, rgb_out = rgb
        _lib.require_gpu(rgb_w, rgb_b, rgb_out)
        assert rgb_w.is_contiguous() and rgb_out.is_contiguous() and tuple(rgb_out.shape) == (n, rgb_w.shape[1], oh, ow)
        d = _desc(n, cin, h, w, pc.cout, pc.cout_pad, oh, ow, stride, 1, taps, None, [0], [0], oh, ow, ow, oh * ow,
                  pc.cout * oh * ow, 0, 0 if out_scale is None else out_scale.stride(0) if out_scale.ndim == 2 else 0)
        d.rgb_w, d.rgb_bias, d.rgb_out, d.rgb_channels = rgb_w.data_ptr(), _lib.ptr(rgb_b), rgb_out.data_ptr(), rgb_w.shape[1]
        rc = launch(None, x.data_ptr(), wptr, _lib.ptr(in_scale), _lib.ptr(out_scale),
                    C.byref(d), C.byref(epilogue) if epilogue is not None else None, _lib.stream_ptr())
        _lib.check(rc, "conv_taps(rgb)")
        return rgb_out
    if out is None:
        out = torch.empty([n, pc.cout, oh, ow], dtype=torch.float32, device=x.device)
    assert out.is_contiguous() and out.shape[0] == n and out.shape[2] == oh and out.shape[3] == ow
    if (POINTWISE and pc.kh == 1 and pc.kw == 1 and stride == 1 and pad == (0, 0) and out_scale is None
            and (in_scale is None or (in_scale.is_contiguous() and tuple(in_scale.shape) == (n, cin)))
            and taps == [(0, 0)] and (epilogue is None or not epilogue.noise)):
        # 1x1 layer without demodulation: the register-operand GEMM (csrc/pointwise.hip) instead of the LDS-staged tap-list kernel
        rc = _lib.lib().mgf_conv1x1_f32(out.data_ptr(), x.data_ptr(), pc.wp.data_ptr(), _lib.ptr(in_scale), n, cin, h * w, pc.cout, pc.cout_pad,
                                        out.shape[1] * oh * ow, out_choff, C.byref(epilogue) if epilogue is not None else None,
                                        _lib.stream_ptr())
        _lib.check(rc, "conv1x1")
        return out
    d = _desc(n, cin, h, w, pc.cout, pc.cout_pad, oh, ow, stride, 1, taps, None, [0], [0], oh, ow,
              ow, oh * ow, out.shape[1] * oh * ow, out_choff,
              0 if out_scale is None else out_scale.stride(0) if out_scale.ndim == 2 else 0)
    rc = launch(out.data_ptr(), x.data_ptr(), wptr, _lib.ptr(in_scale), _lib.ptr(out_scale),
                C.byref(d), C.byref(epilogue) if epilogue is not None else None, _lib.stream_ptr())
    _lib.check(rc, "conv_taps")
    return out


def transpose_packed(pc: PackedConv, flip: bool) -> PackedConv:
    """The packed taps of the data-gradient convolution: channels swapped ([tap][cout][cin_pad], gain kept), tap order reversed
    when `flip` (the gradient of a correlation is a true convolution with the same kernel)."""
    t = pc.wp[:, :, :pc.cout].permute(0, 2, 1)
    if flip:
        t = t.flip(0)
    cpad = round_up(pc.cin, 32)
    wp = torch.zeros([pc.kh * pc.kw, pc.cout, cpad], dtype=torch.float32, device=pc.wp.device)
    wp[:, :, :pc.cin] = t
    return PackedConv(wp.contiguous(), None, pc.cin, pc.cout, pc.kh, pc.kw, cpad)


def winograd_weights(w: torch.Tensor, gain: float = 1.0) -> torch.Tensor:
    """[cout, cin, 3, 3] -> the 16 transformed weight planes u [16, cin, cout] of mgf_conv3x3_winograd_f32 (once per checkpoint)."""
    _lib.require_gpu(w)
    w = w.contiguous().float()
    cout, cin, kh, kw = w.shape
    assert (kh, kw) == (3, 3)
    u = torch.empty([16, cin, cout], dtype=torch.float32, device=w.device)
    _lib.check(_lib.lib().mgf_winograd_weights_f32(u.data_ptr(), w.data_ptr(), cout, cin, float(gain), _lib.stream_ptr()), "winograd_weights")
    return u


def winograd2_weights(w: torch.Tensor, gain: float = 1.0) -> torch.Tensor:
    """Weight planes of the second kernel form (mgf_conv3x3_winograd2_f32): [16, cin / 4, cout, 4]."""
    _lib.require_gpu(w)
    w = w.contiguous().float()
    cout, cin, kh, kw = w.shape
    assert (kh, kw) == (3, 3) and cin % 4 == 0
    u = torch.empty([16, cin // 4, cout, 4], dtype=torch.float32, device=w.device)
    _lib.check(_lib.lib().mgf_winograd2_weights_f32(u.data_ptr(), w.data_ptr(), cout, cin, float(gain), _lib.stream_ptr()), "winograd2_weights")
    return u


# Which kernel serves the [16, cin/4, cout, 4] weight layout on even maps: form 3 (csrc/wino3.hip, the transformed input stays in
# registers) unless MGF_WINOGRAD_FORM=2 (tuning hook: form 2, csrc/wino.hip, which also takes odd maps and channel slices)
WINOGRAD_FORM = int(os.environ.get("MGF_WINOGRAD_FORM", "3"))
WINOGRAD3_MIN_WGS = int(os.environ.get("MGF_WINOGRAD3_MIN_WGS", "128"))      # tuning hook: fewest form-3 workgroups that still beat the split-K tap-list launch (one target, gradient mode: 512 / 256 / 128 / 64 -> 157.4 / 157.9 / 159.4 / 158.0 iters/s, tools/wino3_min_wgs_ab.sh)


def winograd3_ok(x, out, out_choff):
    """Form 3 serves every call of the [16, cin/4, cout, 4] weight layout (odd maps and channel slices included) unless pinned off."""
    return WINOGRAD_FORM == 3 and min(x.shape[2], x.shape[3]) >= 2


def winograd2_forward(x, u, in_scale=None, out_scale=None, epilogue=None, out=None, out_choff=0, residual_low=None):
    """3x3 / stride 1 / pad 1 correlation on weights in the [16, cin/4, cout, 4] layout.  Form 3 serves dense outputs on even maps;
    form 2 (and only it) may write channels [out_choff, out_choff + cout) of a wider `out` and takes odd map sides.
    residual_low [n, cout, h/2, w/2] (form 3 only): the epilogue adds upsample2d(residual_low, [1,3,3,1], up=2) -- the resnet skip branch
    without its full-resolution tensor (mgf_conv3x3_winograd3_up2res_f32)."""
    _lib.require_gpu(x, u, in_scale, out_scale, out, residual_low)
    if residual_low is not None:
        n, cin, h, w = x.shape
        cout = u.shape[2]
        assert winograd3_ok(x, out, out_choff) and epilogue is not None, "residual_low needs the form-3 kernel and an epilogue"
        assert residual_low.is_contiguous() and tuple(residual_low.shape) == (n, cout, h // 2, w // 2) and residual_low.dtype == torch.float32
        if out is None:
            out = torch.empty([n, cout, h, w], dtype=torch.float32, device=x.device)
        assert out.is_contiguous() and tuple(out.shape) == (n, cout, h, w)
        os_stride = 0 if out_scale is None else out_scale.stride(0) if out_scale.ndim == 2 else 0
        _lib.check(_lib.lib().mgf_conv3x3_winograd3_up2res_f32(out.data_ptr(), x.data_ptr(), u.data_ptr(), _lib.ptr(in_scale), _lib.ptr(out_scale),
                                                               residual_low.data_ptr(), n, cin, h, w, cout, os_stride, C.byref(epilogue),
                                                               _lib.stream_ptr()), "conv3x3_winograd3_up2res")
        return out
    if winograd3_ok(x, out, out_choff):
        assert x.dtype == torch.float32 and x.is_contiguous() and x.ndim == 4 and u.ndim == 4 and x.shape[1] == u.shape[1] * 4
        n, cin, h, w = x.shape
        cout = u.shape[2]
        if out is None:
            out = torch.empty([n, cout, h, w], dtype=torch.float32, device=x.device)
        assert out.is_contiguous() and out.shape[0] == n and tuple(out.shape[2:]) == (h, w) and out_choff + cout <= out.shape[1]
        os_stride = 0 if out_scale is None else out_scale.stride(0) if out_scale.ndim == 2 else 0
        rc = _lib.lib().mgf_conv3x3_winograd3_slice_f32(out.data_ptr(), x.data_ptr(), u.data_ptr(), _lib.ptr(in_scale), _lib.ptr(out_scale), n, cin, h,
                                                        w, cout, os_stride, out.shape[1] * h * w, out_choff,
                                                        C.byref(epilogue) if epilogue is not None else None, _lib.stream_ptr())
        _lib.check(rc, "conv3x3_winograd3")
        return out
    assert x.dtype == torch.float32 and x.is_contiguous() and x.ndim == 4 and u.ndim == 4 and x.shape[1] == u.shape[1] * 4
    n, cin, h, w = x.shape
    cout = u.shape[2]
    if out is None:
        out = torch.empty([n, cout, h, w], dtype=torch.float32, device=x.device)
    assert out.is_contiguous() and out.shape[0] == n and tuple(out.shape[2:]) == (h, w) and out_choff + cout <= out.shape[1]
    os_stride = 0 if out_scale is None else out_scale.stride(0) if out_scale.ndim == 2 else 0
    rc = _lib.lib().mgf_conv3x3_winograd2_slice_f32(out.data_ptr(), x.data_ptr(), u.data_ptr(), _lib.ptr(in_scale), _lib.ptr(out_scale), n, cin, h,
                                                    w, cout, os_stride, out.shape[1] * h * w, out_choff,
                                                    C.byref(epilogue) if epilogue is not None else None, _lib.stream_ptr())
    _lib.check(rc, "conv3x3_winograd2")
    return out


def winograd2_rgb_forward(x, u, rgb_w, rgb_bias, rgb_out, in_scale=None, out_scale=None):
    """conv_last + ToRGB in one Winograd launch (cout == 32): rgb_out [n,c,h,w] = rgb_w [n,c,cout] . (out_scale * conv3x3(x)) + rgb_bias."""
    _lib.require_gpu(x, u, rgb_w, rgb_bias, rgb_out, in_scale, out_scale)
    n, cin, h, w = x.shape
    cout = u.shape[2]
    assert x.is_contiguous() and u.ndim == 4 and u.shape[3] == 4 and rgb_w.is_contiguous() and rgb_out.is_contiguous()
    assert tuple(rgb_out.shape) == (n, rgb_w.shape[1], h, w) and rgb_w.shape[2] == cout
    os_stride = 0 if out_scale is None else out_scale.stride(0) if out_scale.ndim == 2 else 0
    if WINOGRAD_FORM == 3 and h % 2 == 0 and w % 2 == 0 and rgb_w.shape[1] <= 3:
        _lib.check(_lib.lib().mgf_conv3x3_winograd3_rgb_f32(rgb_out.data_ptr(), x.data_ptr(), u.data_ptr(), _lib.ptr(in_scale), _lib.ptr(out_scale),
                                                            rgb_w.data_ptr(), _lib.ptr(rgb_bias), n, cin, h, w, cout, os_stride, rgb_w.shape[1],
                                                            _lib.stream_ptr()), "conv3x3_winograd3_rgb")
        return rgb_out
    _lib.check(_lib.lib().mgf_conv3x3_winograd2_rgb_f32(rgb_out.data_ptr(), x.data_ptr(), u.data_ptr(), _lib.ptr(in_scale), _lib.ptr(out_scale),
                                                        rgb_w.data_ptr(), _lib.ptr(rgb_bias), n, cin, h, w, cout, os_stride, rgb_w.shape[1],
                                                        _lib.stream_ptr()), "conv3x3_winograd2_rgb")
    return rgb_out


def winograd_ok(cin, cout, h, w):
    """Shapes the Winograd kernels take (winograd_pack picks the form by map size): >= 16x16 maps with even sides; form 1 (16x16 maps)
    whole 64-channel output tiles and 8-channel chunks, form 2 whole 32-channel tiles and 4-channel chunks."""
    if h % 2 or w % 2 or min(h, w) < 16:
        return False
    return (cin % 8 == 0 and cout % 64 == 0) if min(h, w) <= 16 else (cin % 4 == 0 and cout % 32 == 0)


def winograd_fills_chip(n, cout, h, w):
    """The Winograd kernel has no split-K: one workgroup per (sample, 16x16 tile, 64 channels), one workgroup per CU.  Below two
    waves of workgroups (2 x 256) the tap-list kernel with its split-K path is faster (a single 1024^2 projection, n = 1)."""
    if min(h, w) <= 16:
        return n * -(-h // 16) * -(-w // 16) * (cout // 64) >= 512
    if WINOGRAD_FORM == 3:
        # form 3: 32 channels x (32 x 4 outputs) per workgroup, four workgroups per CU.  Half a wave of workgroups (512 of the 1024 the
        # chip holds: a single 64x64 x 512-channel image, gradient mode) is still faster than the split-K tap-list launch + its reduce
        return n * -(-h // 4) * -(-w // 32) * (cout // 32) >= WINOGRAD3_MIN_WGS
    return n * -(-h // 8) * -(-w // 32) * (cout // 32) >= 512


def winograd_pack(w: torch.Tensor, gain: float, res: int) -> torch.Tensor:
    """Transformed weights in the layout of the kernel form that is faster at this map size (tools/wino_micro.py): the
    one-workgroup-per-CU form on 16x16 maps, the two-workgroups-per-CU form above.  winograd_forward dispatches on the layout."""
    return winograd_weights(w, gain) if res <= 16 else winograd2_weights(w, gain)


def winograd_forward(x, u, in_scale=None, out_scale=None, epilogue=None, out=None, residual_low=None):
    """3x3 / stride 1 / pad 1 correlation through the Winograd F(2x2,3x3) kernels; same contract as conv_forward(pad=(1, 1)).
    u: [16, cin, cout] (winograd_weights) or [16, cin / 4, cout, 4] (winograd2_weights)."""
    if u.ndim == 4:
        return winograd2_forward(x, u, in_scale, out_scale, epilogue, out, residual_low=residual_low)
    assert residual_low is None
    _lib.require_gpu(x, u, in_scale, out_scale, out)
    assert x.dtype == torch.float32 and x.is_contiguous() and x.ndim == 4 and x.shape[1] == u.shape[1]
    n, cin, h, w = x.shape
    cout = u.shape[2]
    if out is None:
        out = torch.empty([n, cout, h, w], dtype=torch.float32, device=x.device)
    assert out.is_contiguous() and tuple(out.shape) == (n, cout, h, w)
    os_stride = 0 if out_scale is None else out_scale.stride(0) if out_scale.ndim == 2 else 0
    rc = _lib.lib().mgf_conv3x3_winograd_f32(out.data_ptr(), x.data_ptr(), u.data_ptr(), _lib.ptr(in_scale), _lib.ptr(out_scale), n, cin, h, w,
                                             cout, os_stride, C.byref(epilogue) if epilogue is not None else None, _lib.stream_ptr())
    _lib.check(rc, "conv3x3_winograd")
    return out


def conv_large_forward(x, w, bias, stride, pad, act="relu", out=None):
    """Convolution with more than 9 taps (AlexNet's 11x11 and 5x5): the tap list is cut into groups of <= 9, each group is one
    launch accumulating into `out` through the residual port (linear), bias + activation follow in one mgf_bias_act pass."""
    _lib.require_gpu(x, w, bias)
    cout, cin, kh, kw = w.shape
    n, _, h, wd = x.shape
    py, px = (pad, pad) if isinstance(pad, int) else pad
    oh, ow = (h + 2 * py - kh) // stride + 1, (wd + 2 * px - kw) // stride + 1
    if out is None:
        out = torch.empty([n, cout, oh, ow], dtype=torch.float32, device=x.device)
    taps = [(a, b) for a in range(kh) for b in range(kw)]
    wf = w.reshape(cout, cin, kh * kw)
    for g0 in range(0, len(taps), 9):
        grp = taps[g0:g0 + 9]
        pc = pack_weights(wf[:, :, g0:g0 + len(grp)].reshape(cout, cin, 1, len(grp)).contiguous())
        ep = None if g0 == 0 else _lib.make_epilogue(residual=out)
        conv_forward(x, pc, stride=stride, pad=(py, px), epilogue=ep, out=out, taps=[(a - py, b - px) for a, b in grp],
                     ksize=(kh, kw))
    if bias is None and act == "linear":
        return out
    act_code = {"linear": 1, "relu": 2}[act]
    _lib.check(_lib.lib().mgf_bias_act(out.data_ptr(), out.data_ptr(), _lib.ptr(bias), None, None, None, _lib.MGF_F32, out.numel(),
                                       oh * ow, cout, 0, act_code, 0.0, 1.0, -1.0, _lib.stream_ptr()), "bias_act")
    return out


def conv_large_dgrad(dy, w, pad, out=None):
    """Data gradient of a stride-1 convolution with more than 9 taps (AlexNet's 5x5): dx = correlation of dy with the spatially
    flipped, channel-transposed kernel, padding k - 1 - pad -- the same chained <= 9-tap launches as conv_large_forward."""
    cout, cin, kh, kw = w.shape
    wt = w.permute(1, 0, 2, 3).flip(2, 3).contiguous()             # [cin, cout, kh, kw]
    assert kh == kw and kh - 1 - pad >= 0
    return conv_large_forward(dy, wt, None, 1, kh - 1 - pad, act="linear", out=out)


def conv_strided_dgrad(dy, w, stride, pad, in_hw, out=None, in_scale=None, out_scale=None):
    """Data gradient of a strided convolution y = conv(x, w, stride, pad) with a large kernel (AlexNet's 11x11 / stride 4 stem):
    dx[s i + kh - pad] += w[co, ci, kh, kw] dy[co, i].  The stride^2 output phases are independent stride-1 correlations of dy with the
    sub-kernels w[.., r + s t, r' + s u] (<= 9 taps each for 11x11 / 4), run as tap-list launches into a phase-planar buffer that one
    strided copy interleaves into dx [n, cin, H, W].  pad: p or (py, px); in_scale [n, cout] / out_scale [n, cin]: per-sample channel scales of
    dy / dx inside the launches (the modulated form of torch_utils.ops.conv2d_resample)."""
    _lib.require_gpu(dy, w, in_scale, out_scale)
    cout, cin, kh, kw = w.shape
    pad_y, pad_x = (pad, pad) if isinstance(pad, int) else pad
    if -(-kh // stride) * -(-kw // stride) > _lib.MAX_TAPS:
        raise _lib.MgfError(f"conv_strided_dgrad: a {kh}x{kw} kernel at stride {stride} has more than {_lib.MAX_TAPS} taps per output phase")
    n, _, hy, wy = dy.shape
    H, W = in_hw
    s = stride
    hq, wq = -(-H // s), -(-W // s)
    if out is None:
        out = torch.empty([n, cin, H, W], dtype=torch.float32, device=dy.device)
    phase = torch.empty([n, cin, hq, wq], dtype=torch.float32, device=dy.device)

    def geom(diff):                 # (pad, ksize) that make conv_forward's output hq = hy + diff long (the taps are given explicitly)
        assert diff >= 0
        return ((diff + 1) // 2, 2) if diff % 2 else (diff // 2, 1)

    (py_, ky_), (px_, kx_) = geom(hq - hy), geom(wq - wy)
    key = (w.data_ptr(), tuple(w.shape), s, pad_y, pad_x)
    hit = _STRIDED_PLANS.get(key)
    # the entry holds a reference to `w` itself: the address cannot be recycled for other weights while the plan lives, and a hit is
    # only taken for the very same tensor object's storage (same version counter = not modified in place since)
    plan = hit[0] if hit is not None and hit[1] is w and hit[2] == w._version else None
    if plan is None:                # the phase sub-kernels are a checkpoint constant: packed once
        plan = []
        for a in range(s):
            ry, qy = (a + pad_y) % s, (a + pad_y) // s
            ts = [t for t in range(-(-kh // s)) if ry + s * t < kh]
            for b in range(s):
                rx, qx = (b + pad_x) % s, (b + pad_x) // s
                us = [u for u in range(-(-kw // s)) if rx + s * u < kw]
                if not ts or not us:                                 # a phase no tap reaches (kernel narrower than the stride): zeros
                    plan.append((a, b, None, None))
                    continue
                sub = torch.stack([w[:, :, ry + s * t, rx + s * u] for t in ts for u in us], dim=-1)        # [cout, cin, taps]
                pc = pack_weights(sub.permute(1, 0, 2).reshape(cin, cout, 1, len(ts) * len(us)).contiguous())
                plan.append((a, b, pc, [(qy - t, qx - u) for t in ts for u in us]))
        if len(_STRIDED_PLANS) >= 8:                                 # a handful of stems at most: drop the oldest plan and its packs
            _STRIDED_PLANS.pop(next(iter(_STRIDED_PLANS)))
        _STRIDED_PLANS[key] = (plan, w, w._version)
    for a, b, pc, taps in plan:
        if pc is None:
            out[:, :, a::s, b::s].zero_()
            continue
        conv_forward(dy, pc, pad=(py_, px_), taps=taps, ksize=(ky_, kx_), out=phase, in_scale=in_scale, out_scale=out_scale)
        dst = out[:, :, a::s, b::s]
        dst.copy_(phase[:, :, :dst.shape[2], :dst.shape[3]])
    return out


_STRIDED_PLANS = {}


TCONV_TAPS = [(-1 if kh == 2 else 0, -1 if kw == 2 else 0) for kh in range(3) for kw in range(3)]
TCONV_GROUPS = [(2 if kh == 1 else 0) + (1 if kw == 1 else 0) for kh in range(3) for kw in range(3)]


SPLIT_TCONV_BORDER = os.environ.get("MGF_TCONV_BORDER", "1") != "0"       # tuning hook: 0 = one launch over the (h+1) x (w+1) grid
TCONV_SPLIT_MIN = int(os.environ.get("MGF_TCONV_SPLIT_MIN", "16"))         # smallest map side that takes the split (tuning hook)


def tconv_pitch(w: int, align: int = 4) -> int:
    """Row pitch (floats) of the transposed conv's [2h+1, 2w+1] workspace: a multiple of 4 (aligned float2 / float4 accesses).  align=32
    starts every row on a 128-byte line: the 256-byte row segments a tile stores are then whole lines -- nothing for the float32 kernel,
    whose stores hide under matrix work, but 22 % of the 512 -> 1024 launch in the bf16x3 mode, whose stores do not (2310 -> 1806 us)."""
    return round_up(2 * w + 1, align)


def conv3x3s2_few_inputs(x, w, bias=None, relu=False, out=None):
    """3x3 / stride-2 / unpadded convolution of a map with <= 4 channels (+ bias, ReLU): the LPIPS stem outside the fused stem kernel.
    w: [cout, cin, 3, 3] (the torch layout, not a PackedConv)."""
    _lib.require_gpu(x, w, bias, out)
    cout, cin = w.shape[:2]
    assert tuple(w.shape[2:]) == (3, 3) and cin <= 4 and w.is_contiguous() and w.dtype == torch.float32
    assert x.dtype == torch.float32 and x.is_contiguous() and x.shape[1] == cin
    n, _, h, wd = x.shape
    oh, ow = (h - 3) // 2 + 1, (wd - 3) // 2 + 1
    if out is None:
        out = torch.empty([n, cout, oh, ow], dtype=torch.float32, device=x.device)
    assert out.is_contiguous() and tuple(out.shape) == (n, cout, oh, ow)
    rc = _lib.lib().mgf_conv3x3s2_few_inputs_f32(out.data_ptr(), x.data_ptr(), w.data_ptr(), _lib.ptr(bias), n, cin, h, wd, cout, int(relu),
                                                 _lib.stream_ptr())
    _lib.check(rc, "conv3x3s2_few_inputs")
    return out


def tconv3x3s2_forward(x, pc: PackedConv, in_scale=None, out_scale=None, out=None, bf=None):
    """Stride-2 3x3 transposed convolution t[2i+kh, 2j+kw] += w[kh,kw] x[i,j] -> view [n, cout, 2h+1, 2w+1] of a padded-pitch
    workspace (row pitch a multiple of 4 floats so the parity pairs are written as aligned float2)."""
    _lib.require_gpu(x, pc.wp, in_scale, out_scale, out)
    assert pc.kh == 3 and pc.kw == 3 and x.dtype == torch.float32 and x.is_contiguous()
    n, cin, h, w = x.shape
    oh, ow = 2 * h + 1, 2 * w + 1
    if out is None:
        out = torch.empty([n, pc.cout, oh, tconv_pitch(w)], dtype=torch.float32, device=x.device)
    pitch = out.shape[3]                               # (the caller's workspace decides: tconv_pitch(w) or a 128-byte aligned one)
    assert out.is_contiguous() and tuple(out.shape) == (n, pc.cout, oh, pitch) and pitch >= ow and pitch % 4 == 0
    # The MFMA launch tiles the h x w grid of 2x2 output quads exactly (rows/columns 0 .. 2h-1 / 2w-1); the last row and column
    # -- 4*in + 1 positions with at most two taps each -- come from a small border kernel.  Tiling (h+1) x (w+1) instead would
    # spend 7-20 % of the MFMA work on padding (33-wide parity grids over 32-wide tiles) -- and half of it on a 16 px map, whose 17 x 17
    # parity grid needs two 256-lane tiles where the 16 x 16 quads fill exactly one (727 -> 503 us for the 16 -> 32 layer at 25 samples).
    # Maps below 16 px keep the single launch (8 px: 212 vs 302 us with the split).
    if NARROW_CONV and pc.cout <= 4 and in_scale is None and out_scale is None:
        # <= 4 output channels (the LPIPS stem's data gradient): a stream over x, not a GEMM (csrc/narrow_conv.hip)
        rc = _lib.lib().mgf_tconv3x3s2_few_outputs_f32(out.data_ptr(), x.data_ptr(), pc.wp.data_ptr(), n, cin, h, w, pc.cout, pc.cout_pad,
                                                       pitch, oh * pitch, pc.cout * oh * pitch, _lib.stream_ptr())
        _lib.check(rc, "tconv3x3s2_few_outputs")
        return out[:, :, :, :ow]
    # (one or two images -- gradient mode at a single target -- keep the single launch up to 64 px: the border kernel's few workgroups then
    # cost more than the padded tiles, 6.76 -> 6.64 ms per gradient step; from four images on the split wins from 16 px, 686 vs 677 iters/s)
    split = SPLIT_TCONV_BORDER and min(h, w) >= (TCONV_SPLIT_MIN if n >= 4 else max(TCONV_SPLIT_MIN, 128))
    # bf: the main launch in the opt-in bf16x3 arithmetic (needs the split form: whole 32-wide tiles of quads); the border stays float32
    use_bf = bf is not None and split and w % 32 == 0 and cin % 16 == 0
    os_stride = 0 if out_scale is None else out_scale.stride(0) if out_scale.ndim == 2 else 0
    d = _desc(n, cin, h, w, pc.cout, pc.cout_pad, h if split else h + 1, w if split else w + 1, 1, 2, TCONV_TAPS, TCONV_GROUPS,
              [0, 0, 1, 1], [0, 1, 0, 1], oh, ow, pitch, oh * pitch, pc.cout * oh * pitch, 0, os_stride)
    if use_bf:
        rc = _lib.lib().mgf_conv_taps_bf16x3_f32(out.data_ptr(), x.data_ptr(), bf.data_ptr(), _lib.ptr(in_scale), _lib.ptr(out_scale),
                                                 C.byref(d), None, _lib.stream_ptr())
    else:
        rc = _lib.lib().mgf_conv_taps_f32(out.data_ptr(), x.data_ptr(), pc.wp.data_ptr(), _lib.ptr(in_scale), _lib.ptr(out_scale),
                                          C.byref(d), None, _lib.stream_ptr())
    _lib.check(rc, "conv_taps(tconv)")
    if split:
        rc = _lib.lib().mgf_tconv3x3s2_border_f32(out.data_ptr(), x.data_ptr(), pc.wp.data_ptr(), _lib.ptr(in_scale), _lib.ptr(out_scale),
                                                  n, cin, h, w, pc.cout, pc.cout_pad, pitch, oh * pitch, pc.cout * oh * pitch, os_stride,
                                                  _lib.stream_ptr())
        _lib.check(rc, "tconv3x3s2_border")
    return out[:, :, :, :ow]


def upfirdn_into(y, x, f2d, up=1, pad=(0, 0, 0, 0), gain=1.0, flip=False, epilogue=None, separable=False, down=1):
    """mgf_upfirdn2d on arbitrary-stride 4-D views (x may be the padded-pitch transposed-conv workspace).
    separable=True asserts that f2d is an outer product (every setup_filter([taps]) result is) -> MGF_FILTER_SEPARABLE hint."""
    _lib.require_gpu(x, y, f2d)
    n, c, h, w = x.shape
    fh, fw = f2d.shape
    px0, px1, py0, py1 = pad
    oh = (h * up + py0 + py1 - fh + down) // down
    ow = (w * up + px0 + px1 - fw + down) // down
    assert tuple(y.shape) == (n, c, oh, ow), (tuple(y.shape), (n, c, oh, ow))
    sx, sy = x.stride(), y.stride()
    rc = _lib.lib().mgf_upfirdn2d(y.data_ptr(), x.data_ptr(), f2d.data_ptr(), _lib.MGF_F32, n, c, h, w, sx[0], sx[1], sx[2],
                                  sx[3], oh, ow, sy[0], sy[1], sy[2], sy[3], fh, fw, up, up, down, down, px0, px1, py0, py1,
                                  int(flip) | (2 if separable else 0) | 4,      # 4 = MGF_FILTER_LARGE: the engine's own calls
                                  float(gain), C.byref(epilogue) if epilogue is not None else None,
                                  _lib.stream_ptr())
    _lib.check(rc, "upfirdn2d")
    return y
