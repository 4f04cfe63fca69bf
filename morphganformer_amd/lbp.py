"""Host side of the LBP matching-distance objective (1024_example_LBP_percept.py:34-58): the tables the device kernels read and the
target's feature.  The per-candidate work -- to_pil quantisation, BGR2GRAY, the 224 x 224 resize, local_binary_pattern(24, 3, 'uniform'),
the cosine distance -- is csrc/lbp.hip (mgf_lbp_gray224_u8 / mgf_lbp_codes_u8 / mgf_lbp_distance_f64).

OpenCV and scikit-image are third-party packages outside the reference tree (and absent offline): what is restated here is their
published algorithm -- cv2.resize's INTER_LINEAR on uint8 (imgproc/resize.cpp: half-pixel centres, coefficients rounded to 11-bit
fixed point) and skimage/feature/_texture.pyx's sample offsets (rounded to 5 decimals).  UNPINNED against the real packages; one known
difference outside this restatement's reach: `cv2.imread(path, IMREAD_GRAYSCALE)` of a JPEG takes libjpeg's own luma plane, not the
BGR2GRAY fixed-point weights `target_feature` applies to decoded RGB pixels (identical for PNG / BMP targets)."""
import numpy as np
import torch

from . import _lib

SIDE, POINTS, RADIUS = 224, 24, 3          # settings of the script (:34-37; cv2.resize(.., (224, 224)), n_points = 8 * radius, 'uniform')


def resize_table(dst, src):
    """cv2.resize INTER_LINEAR, one axis: (index 0, index 1, coefficient 0, coefficient 1) per destination position, coefficients at scale
    2048 (saturate_cast<short>(c * INTER_RESIZE_COEF_SCALE), i.e. round-half-even of the float32 product)."""
    # resize.cpp: scale = 1. / ((double)dst / src); fx = (float)((dx + 0.5) * scale - 0.5); sx = cvFloor(fx); fx -= sx -- float32 BEFORE the
    # floor and the fraction (in float64 the index or an 11-bit coefficient can differ by one for some size ratios)
    f = ((np.arange(dst, dtype=np.float64) + 0.5) * (1.0 / (dst / src)) - 0.5).astype(np.float32)
    s0f = np.floor(f)
    s0 = s0f.astype(np.int64)
    fr = (f - s0f).astype(np.float32)
    lo = s0 < 0
    s0[lo], fr[lo] = 0, 0.0
    hi = s0 >= src - 1
    s0[hi], fr[hi] = src - 1, 0.0
    c1 = np.rint(fr * np.float32(2048)).astype(np.int64)
    c0 = np.rint((np.float32(1) - fr) * np.float32(2048)).astype(np.int64)
    return s0, np.minimum(s0 + 1, src - 1), c0, c1


def resize_tables(src_h, src_w, dst=SIDE):
    """int32 [2][dst][4]: the column table (over src_w), then the row table (over src_h) -- what mgf_lbp_gray224_u8 reads."""
    cols = np.stack(resize_table(dst, src_w), 1)
    rows = np.stack(resize_table(dst, src_h), 1)
    return np.ascontiguousarray(np.stack([cols, rows]).astype(np.int32))


def offsets(points=POINTS, radius=RADIUS):
    """float64 [2][P]: skimage's sample offsets rp = round(-R sin(2 pi p / P), 5), cp = round(R cos(2 pi p / P), 5) (_texture.pyx)."""
    a = 2 * np.pi * np.arange(points, dtype=np.double) / points
    return np.ascontiguousarray(np.stack([np.round(-radius * np.sin(a), 5), np.round(radius * np.cos(a), 5)]))


class LbpWorkspace:
    """Device tables + buffers for `n` images of h x w pixels."""

    def __init__(self, n, h, w, device):
        self.n, self.h, self.w = int(n), int(h), int(w)
        self.tab = torch.from_numpy(resize_tables(h, w)).to(device)
        self.off = torch.from_numpy(offsets()).to(device)
        self.gray = torch.empty(self.n, SIDE * SIDE, dtype=torch.uint8, device=device)
        self.scratch = torch.empty(int(_lib.lib().mgf_lbp_scratch_bytes(self.n)) // 8, dtype=torch.int64, device=device)

    def gray224(self, img, true_rgb_order=False):
        """img [n,3,h,w] float32 in [-1, 1] -> the uint8 224 x 224 gray images the feature is computed on."""
        _lib.require_gpu(img)
        n = img.shape[0]
        assert tuple(img.shape[1:]) == (3, self.h, self.w) and n <= self.n and img.dtype == torch.float32 and img.is_contiguous()
        _lib.check(_lib.lib().mgf_lbp_gray224_u8(self.gray.data_ptr(), img.data_ptr(), self.tab.data_ptr(), n, self.h, self.w,
                                                 1 if true_rgb_order else 0, _lib.stream_ptr()), "lbp_gray224")
        return self.gray[:n]

    def codes(self, gray):
        out = torch.empty_like(gray)
        _lib.check(_lib.lib().mgf_lbp_codes_u8(out.data_ptr(), gray.data_ptr(), self.off.data_ptr(), gray.shape[0], _lib.stream_ptr()), "lbp_codes")
        return out

    def distance_into(self, out_f64, img, target_codes):
        """out[i] = the script's LBPLoss(LBP_feature_im(to_pil(img[i])), target feature) (:47-58), float64."""
        g = self.gray224(img)
        _lib.check(_lib.lib().mgf_lbp_distance_f64(out_f64.data_ptr(), g.data_ptr(), target_codes.data_ptr(), self.off.data_ptr(),
                                                   g.shape[0], self.scratch.data_ptr(), _lib.stream_ptr()), "lbp_distance")
        return out_f64


def target_feature(image_u8_rgb, device="cuda"):
    """`LBP_feature(path)` (:40-45) of the target: cv2.imread(path, IMREAD_GRAYSCALE) -- the file's own pixels at the file's own size, true
    colour order (red weighs 4899) --, resized to 224 x 224, local_binary_pattern.  image_u8_rgb: [H, W, 3] uint8 as PIL reads the file.
    Returns the uint8 code map [224 * 224] on the device (the feature the candidates are compared with)."""
    a = np.asarray(image_u8_rgb)
    assert a.dtype == np.uint8 and a.ndim == 3 and a.shape[2] == 3
    h, w = a.shape[:2]
    ws = LbpWorkspace(1, h, w, device)
    x = torch.from_numpy(np.ascontiguousarray(a.transpose(2, 0, 1))).to(device).float().div_(127.5).sub_(1.0)[None].contiguous()   # to_pil's rint gives the pixels back
    return ws.codes(ws.gray224(x, true_rgb_order=True))[0].clone()
