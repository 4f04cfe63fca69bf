"""ORACLE (test infrastructure only) -- CPU restatement of the projection losses and loop.

* wing_loss_ref / adaptive_wing_loss_ref <- wing_loss.py:19-28, adaptive_wing_loss.py:13-44
* mse_ref                                <- torch.nn.MSELoss as used at 1024_example_wing_loss_perceptual_sqz_MSE.py:176
* get_lr_ref / noise_strength_ref        <- ...sqz_MSE.py:63-68,156
* latent_stats_ref                       <- ...sqz_MSE.py:251-255
* to_uint8_ref                           <- misc.py:114-123 (to_pil rounding contract)
* squeeze_features_ref / lpips_ref       <- lpips/networks_basic.py:64-101, lpips/__init__.py:44-46,
                                            lpips/pretrained_networks.py:6-56 (slice boundaries);
                                            backbone topology = torchvision squeezenet1_1.features
                                            (third-party, un-vendored: torchvision>=0.9.1, requirements.txt:2)
* projection_literal_ref                 <- ...sqz_MSE.py:131-208 (literal semantics: best-of-N noisy sampling,
                                            SURVEY.md section 0.1) with injected epsilon / landmark streams.

Pinned by tests/golden/loss_kats.npz (Wing/AWing/schedule KATs from the reference's own classes) and
tests/golden/loop_tiny.npz (reference Generator + reference WingLoss driven through the loop).
LPIPS: the DISTANCE half (ScalingLayer, normalize_tensor, squared difference, lin heads, spatial average, sum) is pinned by
tests/golden/lpips_dist.npz = the reference's own lpips/networks_basic.py PNetLin.forward / lpips.PerceptualLoss.forward run on
injected tap tensors and vendored lin weights.  The BACKBONE half with ImageNet weights stays UNPINNED (torchvision and its
weights are absent offline): topology restated, checked on seeded random weights against torch's own conv ops.
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F


def wing_loss_ref(pred, target, omega=10.0, epsilon=2.0):
    d = (target - pred).abs()
    c = omega - omega * math.log(1 + omega / epsilon)
    small = omega * torch.log(1 + d / epsilon)
    loss = torch.where(d < omega, small, d - c)
    return loss.sum() / d.numel()


def adaptive_wing_loss_ref(pred, target, omega=14.0, theta=0.5, epsilon=1.0, alpha=2.1):
    y = target
    d = (y - pred).abs()
    p = alpha - y
    l1 = omega * torch.log(1 + torch.pow(d / omega, p))
    tp = torch.pow(torch.as_tensor(theta / epsilon, dtype=y.dtype), p)
    a = omega * (1 / (1 + tp)) * p * torch.pow(torch.as_tensor(theta / epsilon, dtype=y.dtype), p - 1) * (1 / epsilon)
    c = theta * a - omega * torch.log(1 + tp)
    loss = torch.where(d < theta, l1, a * d - c)
    return loss.sum() / d.numel()


def mse_ref(a, b):
    return (a - b).square().mean()


def psnr_ref(p0, p1, peak=255.0):
    """1024_example_PSNR.py:113-114: `10*np.log10(peak**2/np.mean((1.*p0-1.*p1)**2))` on the flattened float32 images (the driver passes
    the [-1, 1] images and keeps the default peak of 255, :158)."""
    import numpy as np
    a, b = np.asarray(p0, dtype=np.float32).reshape(-1), np.asarray(p1, dtype=np.float32).reshape(-1)
    return 10 * np.log10(peak ** 2 / np.mean((1. * a - 1. * b) ** 2))


def psnr_script_ref(img_gen, imgs, peak=255.0):
    """The PSNR driver's loss exactly as 1024_example_PSNR.py:150-158 forms it, transcribed step by step from [1, C, H, W] tensors:
        img_gen_raw2 = img_gen.permute(0, 3, 1, 2);  pre = tensor2np(img_gen_raw2);  trg = tensor2np(imgs)
        psnr(pre.flatten(), trg.flatten())       with tensor2np(t) = t[0].numpy().transpose((1, 2, 0))  (:120-122)
    `pre` comes back in C-H-W order (the permute and the transpose cancel), `trg` is H-W-C: the flattened streams pair element i of one
    with element i of the other, i.e. different pixels and channels.  This function IS the script's arithmetic; whether the product
    follows it (`psnr_layout="script"`) or the aligned definition (`psnr_ref`) is ProjectionArgs' choice."""
    import numpy as np
    import torch
    tensor2np = lambda t: t[0].cpu().float().numpy().transpose((1, 2, 0))
    img_gen, imgs = torch.as_tensor(img_gen), torch.as_tensor(imgs)
    pre = tensor2np(img_gen.permute(0, 3, 1, 2))
    trg = tensor2np(imgs)
    assert pre.shape == tuple(img_gen.shape[1:]) and trg.shape == (imgs.shape[2], imgs.shape[3], imgs.shape[1])
    return psnr_ref(pre.flatten(), trg.flatten(), peak)


def to_u8_ref(img):
    """misc.py:115-116 (`to_pil`): adjust_range([-1, 1] -> [0, 255]) in float32, np.rint, clip, uint8 -- the image the drivers save."""
    import numpy as np
    x = np.asarray(img, dtype=np.float32)
    scale = (np.float32(255) - np.float32(0)) / (np.float32(1) - np.float32(-1))
    bias = np.float32(0) - np.float32(-1) * scale
    return np.rint(x * scale + bias).clip(0, 255).astype(np.uint8)


def ssim_ref(X, Y, data_range=255.0, win_size=7, K1=0.01, K2=0.03):
    """skimage.measure.compare_ssim(X, Y, data_range=..., multichannel=True) with its defaults, restated from the published source
    (scikit-image 0.14-0.16, skimage/measure/_structural_similarity.py; Wang, Bovik, Sheikh, Simoncelli 2004) -- scikit-image is NOT in this
    image, so this half is UNPINNED against the library itself and held by hand-computed known answers (tests/test_oracle.py) instead.
    X, Y: [H, W, C] arrays of one dtype.  Per channel: uniform_filter means / second moments over win_size x win_size, sample covariance
    (cov_norm = NP / (NP - 1)), S = (2 ux uy + C1)(2 vxy + C2) / ((ux^2 + uy^2 + C1)(vx + vy + C2)), mean of S over the positions whose
    window lies inside the image (crop by (win_size - 1) // 2); then the mean over channels."""
    import numpy as np
    from scipy.ndimage import uniform_filter
    X, Y = np.asarray(X), np.asarray(Y)
    assert X.shape == Y.shape and X.dtype == Y.dtype and X.ndim == 3
    if min(X.shape[:2]) < win_size:
        raise ValueError("win_size exceeds image extent")        # what compare_ssim raises (and what :158's flattened call runs into)
    per = []
    for ch in range(X.shape[-1]):
        x, y = X[..., ch].astype(np.float64), Y[..., ch].astype(np.float64)
        NP = win_size ** 2
        cov_norm = NP / (NP - 1)
        ux, uy = uniform_filter(x, size=win_size), uniform_filter(y, size=win_size)
        uxx, uyy, uxy = uniform_filter(x * x, size=win_size), uniform_filter(y * y, size=win_size), uniform_filter(x * y, size=win_size)
        vx, vy, vxy = cov_norm * (uxx - ux * ux), cov_norm * (uyy - uy * uy), cov_norm * (uxy - ux * uy)
        C1, C2 = (K1 * data_range) ** 2, (K2 * data_range) ** 2
        S = ((2 * ux * uy + C1) * (2 * vxy + C2)) / ((ux ** 2 + uy ** 2 + C1) * (vx + vy + C2))
        pad = (win_size - 1) // 2
        per.append(S[pad:S.shape[0] - pad, pad:S.shape[1] - pad].mean())
    return float(np.mean(per))


def dssim_ref(img, target, data_range=255.0):
    """`dssim` (1024_example_SSIM.py:115-117 = lpips/__init__.py:54-55): (1 - compare_ssim(p0, p1, data_range=255, multichannel=True)) / 2
    of the uint8 HWC images; img, target: [C, H, W] float images in [-1, 1], quantised like the saved image (to_u8_ref).  float32 like the
    script's torch.FloatTensor (:159).  The script itself hands compare_ssim FLATTENED float arrays (:158), which compare_ssim rejects
    (a 1-D array is narrower than the window): this is the function as defined, on the images the function is documented for."""
    import numpy as np
    a = to_u8_ref(img).transpose(1, 2, 0)
    b = to_u8_ref(target).transpose(1, 2, 0)
    return np.float32((1 - ssim_ref(a, b, data_range=data_range)) / 2.)


def cv_bgr2gray_u8_ref(img_u8, blue_first=True):
    """cv2.cvtColor(im, COLOR_BGR2GRAY) on uint8 as OpenCV's 8-bit path publishes it: (B 1868 + G 9617 + R 4899 + 2^13) >> 14.  The scripts
    hand it an RGB array (1024_example_LBP_percept.py:48), so with blue_first=True channel 0 -- red -- takes the blue weight;
    blue_first=False is what cv2.imread(IMREAD_GRAYSCALE) does to a file's true colours (:41).  OpenCV is absent: UNPINNED against it."""
    import numpy as np
    a = np.asarray(img_u8).astype(np.int64)
    c0, c2 = (a[..., 0], a[..., 2]) if blue_first else (a[..., 2], a[..., 0])
    return ((c0 * 1868 + a[..., 1] * 9617 + c2 * 4899 + (1 << 13)) >> 14).astype(np.uint8)


def cv_resize_linear_gray_ref(gray_u8, width, height):
    """cv2.resize(gray, (width, height)), INTER_LINEAR on uint8, pixel by pixel from OpenCV's published scheme (imgproc/resize.cpp): source
    coordinate (d + 0.5) * scale - 0.5, clamped at both ends, the coordinate cast to float32 BEFORE its floor and fraction are taken (as resize.cpp does), float32 weights rounded to 11-bit
    fixed point (cvRound: half to even), the
    horizontal pass in int32, the vertical pass (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2.  UNPINNED against OpenCV."""
    import numpy as np
    g = np.asarray(gray_u8)
    assert g.dtype == np.uint8 and g.ndim == 2
    ih, iw = g.shape

    def axis(dst, src):
        out = []
        for d in range(dst):
            # resize.cpp: scale_x = 1. / ((double)dsize.width / ssize.width); fx = (float)((dx + 0.5) * scale_x - 0.5); sx = cvFloor(fx); fx -= sx
            f = np.float32((d + 0.5) * (1.0 / (dst / src)) - 0.5)
            s = int(math.floor(f))
            fr = np.float32(f - np.float32(s))
            if s < 0:
                s, fr = 0, np.float32(0)
            if s >= src - 1:
                s, fr = src - 1, np.float32(0)
            out.append((s, min(s + 1, src - 1), int(np.rint((np.float32(1) - fr) * np.float32(2048))), int(np.rint(fr * np.float32(2048)))))
        return out

    xs, ys = axis(width, iw), axis(height, ih)
    out = np.zeros((height, width), np.uint8)
    gi = g.astype(np.int64)
    for oy, (y0, y1, b0, b1) in enumerate(ys):
        for ox, (x0, x1, a0, a1) in enumerate(xs):
            s0 = int(gi[y0, x0]) * a0 + int(gi[y0, x1]) * a1
            s1 = int(gi[y1, x0]) * a0 + int(gi[y1, x1]) * a1
            out[oy, ox] = min(255, max(0, (((b0 * (s0 >> 4)) >> 16) + ((b1 * (s1 >> 4)) >> 16) + 2) >> 2))
    return out


def lbp_uniform_ref(image, P=24, R=3):
    """skimage.feature.local_binary_pattern(image, P, R, 'uniform') restated from the published source (skimage/feature/_texture.pyx,
    _local_binary_pattern, and skimage/_shared/interpolation.pxd): the image as float64; sample offsets rp = round(-R sin(2 pi p / P), 5),
    cp = round(R cos(2 pi p / P), 5); texture[p] = bilinear_interpolation(r + rp, c + cp) with mode 'C' (0 outside the image) in exactly
    `top = (1 - dc) * tl + dc * tr; bottom = (1 - dc) * bl + dc * br; (1 - dr) * top + dr * bottom`; signed[p] = texture[p] - centre >= 0;
    changes = number of p < P - 1 with signed[p] != signed[p + 1] (not circular); code = sum(signed) if changes <= 2 else P + 1.
    Returns float64 like skimage.  scikit-image is absent: UNPINNED against it, held by hand-computed answers (tests/test_oracle_golden.py)."""
    import numpy as np
    img = np.ascontiguousarray(image, dtype=np.double)
    rows, cols = img.shape
    ang = 2 * np.pi * np.arange(P, dtype=np.double) / P
    rp, cp = np.round(-R * np.sin(ang), 5), np.round(R * np.cos(ang), 5)
    r = np.arange(rows, dtype=np.double)[:, None] + np.zeros((1, cols))
    c = np.arange(cols, dtype=np.double)[None, :] + np.zeros((rows, 1))

    def px(rr, cc):
        ok = (rr >= 0) & (rr < rows) & (cc >= 0) & (cc < cols)
        return np.where(ok, img[np.clip(rr, 0, rows - 1), np.clip(cc, 0, cols - 1)], 0.0)

    signed = np.zeros((P, rows, cols), np.int64)
    for p in range(P):
        y, x = r + rp[p], c + cp[p]
        minr, minc = np.floor(y).astype(np.int64), np.floor(x).astype(np.int64)
        maxr, maxc = np.ceil(y).astype(np.int64), np.ceil(x).astype(np.int64)
        dr, dc = y - minr, x - minc
        top = (1 - dc) * px(minr, minc) + dc * px(minr, maxc)
        bottom = (1 - dc) * px(maxr, minc) + dc * px(maxr, maxc)
        signed[p] = ((1 - dr) * top + dr * bottom) - img >= 0
    changes = (signed[:-1] != signed[1:]).sum(0)
    return np.where(changes <= 2, signed.sum(0), P + 1).astype(np.double)


def lbp_cosine_distance_ref(x, y):
    """`cosine_distance` (1024_example_LBP_percept.py:54-55) on the flattened float64 features."""
    import numpy as np
    x, y = np.asarray(x, np.double).reshape(-1), np.asarray(y, np.double).reshape(-1)
    return 1 - np.dot(x, y) / (np.sqrt(np.dot(x, x)) * np.sqrt(np.dot(y, y)))


def lbp_feature_im_ref(img):
    """`LBP_feature_im(np.asarray(misc.to_pil(img)))` (:47-52,162-164): img [3, H, W] float in [-1, 1] -> the saved uint8 RGB image -> BGR2GRAY
    (on RGB data) -> 224 x 224 -> local_binary_pattern(24, 3, 'uniform') -> flattened float64."""
    u8 = to_u8_ref(img).transpose(1, 2, 0)
    return lbp_uniform_ref(cv_resize_linear_gray_ref(cv_bgr2gray_u8_ref(u8, blue_first=True), 224, 224)).reshape(-1)


def lbp_feature_file_ref(image_u8_rgb):
    """`LBP_feature(path)` (:40-45): cv2.imread(path, IMREAD_GRAYSCALE) of the file's own pixels (true colour order), 224 x 224, the same feature."""
    return lbp_uniform_ref(cv_resize_linear_gray_ref(cv_bgr2gray_u8_ref(image_u8_rgb, blue_first=False), 224, 224)).reshape(-1)


def pool_above_ref(img, above=256):
    """projection_example_v1.py:148-155: an image taller than `above` is averaged over factor x factor blocks, factor = height // above."""
    import numpy as np
    x = np.asarray(img)
    batch, channel, height, width = x.shape
    if height > above:
        factor = height // above
        x = x.reshape(batch, channel, height // factor, factor, width // factor, factor).mean((3, 5))
    return x


def get_lr_ref(t, initial_lr, rampdown=0.25, rampup=0.05):
    ramp = min(1.0, (1.0 - t) / rampdown)
    ramp = 0.5 - 0.5 * math.cos(ramp * math.pi)
    ramp = ramp * min(1.0, t / rampup)
    return initial_lr * ramp


def noise_strength_ref(t, latent_std, noise=0.05, noise_ramp=0.75):
    """...sqz_MSE.py:156 with its dtypes: latent_std is a 0-d float32 tensor, so both multiplications are float32 tensor x python
    scalar ops (each rounds to float32); `.item()` returns that value as a python float."""
    return (torch.as_tensor(float(latent_std), dtype=torch.float32) * noise * max(0.0, 1.0 - t / noise_ramp) ** 2).item()


def latent_stats_ref(samples):
    """samples [N,k,D] -> (latent_mean [k,D], latent_std scalar) exactly as the drivers compute them."""
    mean = samples.mean(0)
    std = ((samples - mean).pow(2).sum() / samples.shape[0]) ** 0.5
    return mean, std


def to_uint8_ref(img_chw):
    """misc.to_pil arithmetic: rint(x*127.5 + 127.5) clipped to [0,255], CHW -> HWC."""
    a = np.asarray(img_chw, dtype=np.float32)
    scale = (np.float32(255) - np.float32(0)) / (np.float32(1) - np.float32(-1))
    bias = np.float32(0) - np.float32(-1) * scale
    a = a * scale + bias
    a = np.rint(a).clip(0, 255).astype(np.uint8)
    return a.transpose(1, 2, 0) if a.ndim == 3 else a


# --------------------------------------------------------------------------------------------
# LPIPS (squeeze)

SQUEEZE_FIRES = {3: (64, 16, 64), 4: (128, 16, 64), 6: (128, 32, 128), 7: (256, 32, 128),
                 9: (256, 48, 192), 10: (384, 48, 192), 11: (384, 64, 256), 12: (512, 64, 256)}
SQUEEZE_TAPS_AFTER = [1, 4, 7, 9, 10, 11, 12]       # feature index after which each of the 7 taps is read
SQUEEZE_POOLS = [2, 5, 8]
SQUEEZE_CHNS = [64, 128, 256, 384, 384, 512, 512]
LPIPS_SHIFT = (-0.030, -0.088, -0.188)
LPIPS_SCALE = (0.458, 0.448, 0.450)


def squeeze_backbone_random(seed=0):
    """Seeded He-scaled random SqueezeNet1.1 feature weights under torchvision's key names."""
    rng = np.random.Generator(np.random.PCG64(seed))
    sd = {}

    def conv(name, co, ci, k):
        sd[name + ".weight"] = torch.from_numpy((rng.standard_normal((co, ci, k, k)) * math.sqrt(2.0 / (ci * k * k))).astype(np.float32))
        sd[name + ".bias"] = torch.from_numpy((rng.standard_normal(co) * 0.05).astype(np.float32))

    conv("features.0", 64, 3, 3)
    for idx, (ci, sq, ex) in SQUEEZE_FIRES.items():
        conv(f"features.{idx}.squeeze", sq, ci, 1)
        conv(f"features.{idx}.expand1x1", ex, sq, 1)
        conv(f"features.{idx}.expand3x3", ex, sq, 3)
    return sd


def squeeze_features_ref(bb, x):
    """The 7 LPIPS taps of SqueezeNet1.1 on an already-scaled input."""
    taps = []
    h = F.relu(F.conv2d(x, bb["features.0.weight"], bb["features.0.bias"], stride=2))
    for idx in range(1, 13):
        if idx == 1:
            pass
        elif idx in SQUEEZE_POOLS:
            h = F.max_pool2d(h, kernel_size=3, stride=2, ceil_mode=True)
        else:
            p = f"features.{idx}"
            s = F.relu(F.conv2d(h, bb[p + ".squeeze.weight"], bb[p + ".squeeze.bias"]))
            e1 = F.relu(F.conv2d(s, bb[p + ".expand1x1.weight"], bb[p + ".expand1x1.bias"]))
            e3 = F.relu(F.conv2d(s, bb[p + ".expand3x3.weight"], bb[p + ".expand3x3.bias"], padding=1))
            h = torch.cat([e1, e3], 1)
        if idx in SQUEEZE_TAPS_AFTER:
            taps.append(h)
    return taps


# torchvision vgg16.features / alexnet.features (third-party, not vendored; topology restated from torchvision >= 0.9.1) with
# the LPIPS slice boundaries of lpips/pretrained_networks.py:58-135.  Rows: ("conv", features index, cin, cout, k, stride, pad)
# -- always followed by ReLU --, ("pool", k) = MaxPool2d(k, 2) floor mode, ("tap",) = LPIPS tap.
VGG_SPEC = ([("conv", 0, 3, 64, 3, 1, 1), ("conv", 2, 64, 64, 3, 1, 1), ("tap",), ("pool", 2),
             ("conv", 5, 64, 128, 3, 1, 1), ("conv", 7, 128, 128, 3, 1, 1), ("tap",), ("pool", 2),
             ("conv", 10, 128, 256, 3, 1, 1), ("conv", 12, 256, 256, 3, 1, 1), ("conv", 14, 256, 256, 3, 1, 1), ("tap",), ("pool", 2),
             ("conv", 17, 256, 512, 3, 1, 1), ("conv", 19, 512, 512, 3, 1, 1), ("conv", 21, 512, 512, 3, 1, 1), ("tap",), ("pool", 2),
             ("conv", 24, 512, 512, 3, 1, 1), ("conv", 26, 512, 512, 3, 1, 1), ("conv", 28, 512, 512, 3, 1, 1), ("tap",)])
ALEX_SPEC = ([("conv", 0, 3, 64, 11, 4, 2), ("tap",), ("pool", 3), ("conv", 3, 64, 192, 5, 1, 2), ("tap",), ("pool", 3),
              ("conv", 6, 192, 384, 3, 1, 1), ("tap",), ("conv", 8, 384, 256, 3, 1, 1), ("tap",), ("conv", 10, 256, 256, 3, 1, 1), ("tap",)])
NET_SPECS = {"vgg": VGG_SPEC, "alex": ALEX_SPEC}


def backbone_random(net, seed=0):
    """Seeded He-scaled feature weights under torchvision's key names (same generator as the package's random_backbone)."""
    if net == "squeeze":
        return squeeze_backbone_random(seed)
    rng = np.random.Generator(np.random.PCG64(seed))
    sd = {}
    for row in NET_SPECS[net]:
        if row[0] == "conv":
            _, idx, ci, co, k, _, _ = row
            sd[f"features.{idx}.weight"] = torch.from_numpy((rng.standard_normal((co, ci, k, k)) * math.sqrt(2.0 / (ci * k * k))).astype(np.float32))
            sd[f"features.{idx}.bias"] = torch.from_numpy((rng.standard_normal(co) * 0.05).astype(np.float32))
    return sd


def sequential_features_ref(net, bb, x):
    """The 5 LPIPS taps of vgg16 / alexnet on an already-scaled input."""
    taps, h = [], x
    for row in NET_SPECS[net]:
        if row[0] == "conv":
            _, idx, ci, co, k, s, p = row
            h = F.relu(F.conv2d(h, bb[f"features.{idx}.weight"], bb[f"features.{idx}.bias"], stride=s, padding=p))
        elif row[0] == "pool":
            h = F.max_pool2d(h, kernel_size=row[1], stride=2)
        else:
            taps.append(h)
    return taps


def scaling_layer_ref(x):
    """ScalingLayer.forward (networks_basic.py:94-101)."""
    return (x - torch.tensor(LPIPS_SHIFT).reshape(1, 3, 1, 1)) / torch.tensor(LPIPS_SCALE).reshape(1, 3, 1, 1)


def normalize_tensor_ref(f, eps=1e-10):
    """lpips.normalize_tensor (lpips/__init__.py:44-46)."""
    return f / (f.square().sum(1, keepdim=True).sqrt() + eps)


def lpips_distance_ref(taps0, taps1, lins, per_layer=False):
    """The distance half of PNetLin.forward (networks_basic.py:70-92, lpips=True, spatial=False): unit-normalise each tap over
    channels, squared difference, 1x1 lin head (dropout = identity in eval), spatial mean, sum over taps.
    Pinned by tests/golden/lpips_dist.npz (the reference's own PNetLin.forward on injected tap tensors)."""
    vals = []
    for a, b, lin in zip(taps0, taps1, lins):
        d = (normalize_tensor_ref(a) - normalize_tensor_ref(b)).square()
        vals.append((d * lin.reshape(1, -1, 1, 1)).sum(1, keepdim=True).mean([2, 3], keepdim=True))
    total = vals[0]
    for v in vals[1:]:
        total = total + v
    return (total, vals) if per_layer else total


def lpips_ref(bb, lins, img0, img1, per_layer=False, net="squeeze"):
    """PNetLin.forward (networks_basic.py:64-92), version 0.1, spatial=False.  lins: list of [C] tensors."""
    feats = squeeze_features_ref if net == "squeeze" else (lambda b_, x_: sequential_features_ref(net, b_, x_))
    return lpips_distance_ref(feats(bb, scaling_layer_ref(img0)), feats(bb, scaling_layer_ref(img1)), lins, per_layer)


# --------------------------------------------------------------------------------------------
# Projection loop, literal semantics

def mean_rows_torch_order_ref(rows):
    """`torch.mean(x, 1)` of a float32 [copies, m] block as torch's CPU kernel forms it (SumKernel.cpp: cascade_sum -> multi_row_sum for an
    outer reduction): rows in blocks of 16, each block summed sequentially from zero, the block sums summed sequentially, the remaining
    rows summed sequentially from zero, tail + blocks, then divided by the count -- every step one float32 rounding.  Restated in numpy so
    that the device kernel's order (mgf_latent_perturb_mean) can be held against it AND it against torch itself (1 <= copies <= 255).
    Holds for the columns torch reduces four vector registers at a time, i.e. all of them when m is a multiple of 32 (k D = 544 is;
    checked on this container's AVX-512 build); the remainder columns of other widths take another path (row_sum's four interleaved
    partial sums) that the product does not need and refuses (ProjectionEngine: numel % 32)."""
    import numpy as np
    x = np.asarray(rows, dtype=np.float32)
    n = x.shape[0]
    assert 1 <= n <= 255
    blocks = np.zeros(x.shape[1:], np.float32)
    run = np.zeros(x.shape[1:], np.float32)
    for c in range(n):
        run = (run + x[c]).astype(np.float32)
        if (c & 15) == 15:
            blocks = (blocks + run).astype(np.float32)
            run = np.zeros(x.shape[1:], np.float32)
    return ((run + blocks).astype(np.float32) / np.float32(n)).astype(np.float32)


def projection_literal_ref(gen_fn, loss_fn, latent_mean, latent_std, eps_stream, steps,
                           noise=0.05, noise_ramp=0.75, min_loss_init=100.0, total_steps=None, copies=1):
    """Best-of-N noisy sampling around latent_mean.

    gen_fn(latent [1,k,D]) -> image; loss_fn(step, image) -> python float or None (= 'no face', step skipped);
    eps_stream[i] is the injected randn_like draw of step i.  Returns (best_latent, best_step, best_loss, losses).
    `total_steps` (default = steps) is args.step of the schedule when only a prefix of the run is evaluated.
    copies > 1: projection_example_v2_percept.py:131-166 -- `latent_in` is the flattened start latent repeated `copies` times
    ([1, copies, k D], :133-140), eps_stream[i] is [1, copies, k, D], and the generator sees (and the loop keeps, :193) `torch.mean(latent_n, 1)`
    reshaped to [1, k, D] (:157-158).
    """
    total_steps = total_steps or steps
    latent_in = latent_mean[None].clone()
    if copies > 1:
        latent_in = latent_mean.reshape(1, -1).unsqueeze(1).repeat(1, copies, 1)
    best, best_step, min_loss = None, -1, float(min_loss_init)
    losses = []
    for i in range(steps):
        t = i / total_steps
        sigma = float(noise_strength_ref(t, float(latent_std), noise, noise_ramp))
        if copies > 1:
            latent_c = latent_in + eps_stream[i].reshape(1, copies, -1) * sigma
            latent_n = torch.mean(latent_c, 1).reshape([1, *latent_mean.shape])
        else:
            latent_n = latent_in + eps_stream[i] * sigma
        img = gen_fn(latent_n)
        val = loss_fn(i, img)
        losses.append(val)
        if val is None:
            continue
        if val < min_loss:
            min_loss, best, best_step = val, latent_n.clone(), i
    return best, best_step, min_loss, losses


# --------------------------------------------------------------------------------------------
# Projection loop, gradient mode (the loop the drivers set up at :143-184 with the detach at :158 removed)

def projection_gradient_ref(gen_fn, loss_fn, latent_mean, latent_std, eps_stream, steps, lr=0.01, rampdown=0.25, rampup=0.05,
                            noise=0.05, noise_ramp=0.75, min_loss_init=100.0, total_steps=None, weight_decay=0.0):
    """torch autograd + torch.optim.Adam through gen_fn / loss_fn.  loss_fn(step, image) -> scalar tensor or None ('no face':
    the step is skipped before optimizer.step()).  Returns (best_latent, best_step, best_loss, losses, trajectory) where
    trajectory[i] is latent_in after step i."""
    total_steps = total_steps or steps
    latent_in = latent_mean[None].clone().requires_grad_(True)
    opt = torch.optim.Adam([latent_in], lr=lr, weight_decay=weight_decay)
    best, best_step, min_loss = None, -1, float(min_loss_init)
    losses, traj = [], []
    for i in range(steps):
        t = i / total_steps
        opt.param_groups[0]["lr"] = get_lr_ref(t, lr, rampdown, rampup)
        sigma = float(noise_strength_ref(t, float(latent_std), noise, noise_ramp))
        latent_n = latent_in + eps_stream[i] * sigma
        val = loss_fn(i, gen_fn(latent_n))
        if val is None:
            losses.append(None)
            traj.append(latent_in.detach().clone())
            continue
        opt.zero_grad()
        val.backward()
        opt.step()
        num = float(val.detach())
        losses.append(num)
        traj.append(latent_in.detach().clone())
        if num < min_loss:
            min_loss, best, best_step = num, latent_n.detach().clone(), i
    return best, best_step, min_loss, losses, traj


# --------------------------------------------------------------------------------------------
# Landmark-Delaunay warp (1024_warp_morphs.py:78-113,163-210).  PARITY UNPINNED: the reference's arithmetic lives in OpenCV
# (cv2.getAffineTransform / warpAffine / fillConvexPoly, absent offline).  Restated from the documented semantics: per triangle of
# the destination mesh (in list order, later triangles overwrite), pixels of the integer polygon np.int32(dst triangle), boundary
# included, take the bilinear sample (BORDER_REFLECT_101) of the source at the affine pre-image of their coordinates.  Not
# restated: warpAffine's 1/32-pixel fixed-point coordinate grid, the LINE_AA blend on triangle edges, reflection at the patch border.

def piecewise_affine_warp_ref(src, tri_xy, dst_to_src, background=0.0):
    """src [C,H,W] float64 numpy; tri_xy [T,6] int; dst_to_src [T,6].  Pure numpy, one triangle at a time like the reference."""
    c, h, w = src.shape
    out = np.full((c, h, w), float(background))
    ys, xs = np.mgrid[0:h, 0:w]
    refl = lambda i, n: np.abs(((i + (n - 1)) % (2 * (n - 1))) - (n - 1)) if n > 1 else np.zeros_like(i)
    for q, a in zip(np.asarray(tri_xy, np.int64), np.asarray(dst_to_src, np.float64)):
        e0 = (q[2] - q[0]) * (ys - q[1]) - (q[3] - q[1]) * (xs - q[0])
        e1 = (q[4] - q[2]) * (ys - q[3]) - (q[5] - q[3]) * (xs - q[2])
        e2 = (q[0] - q[4]) * (ys - q[5]) - (q[1] - q[5]) * (xs - q[4])
        m = ((e0 >= 0) & (e1 >= 0) & (e2 >= 0)) | ((e0 <= 0) & (e1 <= 0) & (e2 <= 0))
        sx = a[0] * xs + a[1] * ys + a[2]
        sy = a[3] * xs + a[4] * ys + a[5]
        fx, fy = np.floor(sx), np.floor(sy)
        lx, ly = sx - fx, sy - fy
        x0, x1 = refl(fx.astype(np.int64), w), refl(fx.astype(np.int64) + 1, w)
        y0, y1 = refl(fy.astype(np.int64), h), refl(fy.astype(np.int64) + 1, h)
        val = (1 - ly) * ((1 - lx) * src[:, y0, x0] + lx * src[:, y0, x1]) + ly * ((1 - lx) * src[:, y1, x0] + lx * src[:, y1, x1])
        out[:, m] = val[:, m]
    return out
