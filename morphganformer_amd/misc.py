"""The output contract of the drivers under the reference's own names (misc.py:88-130): `misc.to_pil(img)` and
`misc.crop_max_rectangle(img, ratio)`, as `crop(misc.to_pil(img_gen_raw[0]), args.ratio).save(...)` uses them
(1024_example_wing_loss_perceptual_sqz_MSE.py:190-195, 1024_merge_morph_2.py:86-87, 1024_generate.py:38-40).

The float -> uint8 conversion (adjust_range to [0, 255] in float32, rint, clip: misc.py:103-124) runs on the device
(mgf_to_uint8_hwc, byte-exact against the reference's numpy arithmetic: tests/test_hip_drivers.py); a numpy image -- what the
drivers hold after `.cpu().numpy()` -- is uploaded first.  There is no host conversion path: without the HIP library this raises.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib


def crop_center(img, cw, ch):
    """misc.py:88-90."""
    w, h = img.size
    return img.crop((int((w - cw) // 2), int((h - ch) // 2), int((w + cw) // 2), int((h + ch) // 2)))


def crop_max_rectangle(img, ratio=1.0):
    """Largest centred rectangle of size (s, ratio * s), s = min(w, h / ratio); ratio=None keeps the image (misc.py:94-98)."""
    if ratio is None:
        return img
    s = min(img.size[0], img.size[1] / ratio)
    return crop_center(img, s, ratio * s)


def to_uint8(img, drange=(-1, 1)):
    """CHW (or HW) float image in `drange` -> uint8 HWC (HW) numpy array, the bytes misc.to_pil puts into the PIL image."""
    if tuple(float(v) for v in drange) != (-1.0, 1.0):
        raise ValueError(f"to_pil: only the drivers' drange [-1, 1] is built (got {list(drange)})")
    x = img if isinstance(img, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(np.asarray(img, dtype=np.float32)))
    if x.ndim == 4 and x.shape[0] == 1:
        x = x[0]
    assert x.ndim in (2, 3), tuple(x.shape)
    if x.ndim == 2:
        x = x.unsqueeze(0)
    x = x.to("cuda" if not x.is_cuda else x.device).contiguous().float()
    c, h, w = x.shape
    out = torch.empty([h, w, c], dtype=torch.uint8, device=x.device)
    _lib.check(_lib.lib().mgf_to_uint8_hwc(out.data_ptr(), x.data_ptr(), c, h, w, _lib.stream_ptr()), "to_uint8")
    out = out.cpu().numpy()
    return out[:, :, 0] if c == 1 else out                  # grayscale CHW => HW (misc.py:116-117)


def to_pil(img, drange=(-1, 1)):
    """misc.to_pil (misc.py:112-130): CHW float image (numpy, as the drivers pass it, or a device tensor) -> PIL image, mode L / RGB / RGBA
    by channel count."""
    from PIL import Image
    a = to_uint8(img, drange)
    fmt = "L" if a.ndim == 2 else {1: "L", 3: "RGB", 4: "RGBA"}.get(a.shape[-1], "L")
    return Image.fromarray(a, fmt)
