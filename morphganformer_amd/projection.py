"""The latent-projection loop of the reference drivers (1024_example_wing_loss_perceptual_sqz_MSE.py:131-208 and its
MSE / percept / Wing siblings), device-resident on MI355X.

Semantics ("literal" mode = what the reference computes).  Every driver detaches the generator output
to numpy before any loss, so `latent_in` never receives a gradient and Adam is inert; the loop is a best-of-N noisy search
around `latent_mean`:

    for i in range(steps):
        t = i / steps
        sigma = latent_std * noise * max(0, 1 - t / noise_ramp) ** 2                  (:156)
        latent_n = latent_in + randn_like(latent_in) * sigma                          (:157, :71-73)
        img = G(latent_n, 0.7)[0]      # 0.7 lands in `c`; no truncation; noise_mode="random"   (:158)
        total = LPIPS(img, target) + lamda * Wing(landmarks(img), landmarks(target)) + beta * MSE(img, target)   (:173-179)
        if total < min_loss: min_loss, best = total, latent_n                         (:186-189)

One iteration is a fixed sequence of HIP kernel launches with all loop state (step counter, best-so-far, loss history)
on the device, captured once into a hipGraph and replayed: no PCIe copy and no host synchronisation per step (the
reference makes three crossings of 12.6 MB per step).  Landmark extraction is dlib on the host in the reference
(:159-170) -- a third-party detector that is not reproducible offline -- so landmarks enter as an injected [steps,68,2]
float64 table (and a `valid` table for the "no face found -> continue" branch, :165-166).

`GradientProjectionEngine` below is the gradient-descent reading of the north star (the loss is back-propagated into the
latent and Adam moves it) -- the loop the drivers set up (optimizer, lr schedule) but never get (the detach severs it).
"""
from __future__ import annotations

import math
import os
from dataclasses import dataclass

import numpy as np
import torch

from . import _lib


@dataclass
class ProjectionArgs:
    """argparse defaults of the reference driver (...sqz_MSE.py:224-245)."""
    step: int = 5000
    lamda: float = 0.01
    beta: float = 1.0
    lr: float = 0.01
    lr_rampup: float = 0.05
    lr_rampdown: float = 0.25
    noise: float = 0.05
    noise_ramp: float = 0.75
    truncation_psi: float = 0.7
    n_mean_latent: int = 10000
    ratio: float = 1.0
    percept_weight: float = 1.0     # coefficient of the LPIPS term: 1 in the Wing/LPIPS/MSE drivers, 0.5 in 1024_example_percept_MSE.py:147
    min_loss_init: float = 100.0
    # "mse": beta * MSE(img, target), the drivers' pixel term.  "psnr": the pixel term of 1024_example_PSNR.py:113-114,158 --
    # 10 log10(255^2 / mean((p0 - p1)^2)) on the [-1, 1] float images, MINIMISED like every other loss of these loops (:173-175 keeps
    # the candidate with the smallest value; the script's objective as written, not a claim that it is a sensible one).  AS WRITTEN also
    # covers the element order (`psnr_layout="script"`, the default): :150-153 permute the generated image and tensor2np it back into
    # C-H-W order, tensor2np the target into H-W-C order, and :158 compares the two FLATTENED arrays -- element i of the candidate's
    # CHW stream against element i of the target's HWC stream, i.e. different pixels and channels.  `psnr_layout="aligned"` is the PSNR the
    # function defines on corresponding pixels (a deliberate deviation from the script; ranks candidates by their MSE)
    # "dssim": `dssim` of 1024_example_SSIM.py:115-117 (= lpips/__init__.py:54-55), (1 - SSIM) / 2 with skimage's defaults, on the uint8
    # images (the generated image as the drivers save it, misc.to_pil) -- the function as defined; the script's own call site (:158) passes
    # flattened float arrays, which compare_ssim rejects
    # "lbp": the matching distance of 1024_example_LBP_percept.py:34-58,162-166 -- 1 - cos(LBP(24, 3, 'uniform') code map of the saved image at
    # 224 x 224, the target file's code map) in float64 -- in place of every other term (the script scores nothing else); needs
    # ProjectionEngine(lbp_target=lbp.target_feature(file pixels))
    pixel_term: str = "mse"
    psnr_layout: str = "script"
    # projection_example_v2_percept.py:131-166: the optimised latent holds `latent_copies` (18 there) copies of the start latent, every copy
    # receives its own noise each step and the generator sees their mean (`torch.mean(latent_n, 1)`, reproduced in torch's summation order:
    # the kept latent -- that mean -- is bit-exact); 1 = the other drivers' single latent.  Literal mode; eps is then [steps, 1, copies, k, D]
    latent_copies: int = 1
    # projection_example_v1.py:150-155: a generated image taller than `pool_above` pixels is block-averaged by height // pool_above before
    # the image-space losses (the target is then given at the pooled size, :84-92 resize it to 256); 0 = off (the 1024 drivers)
    pool_above: int = 0


def get_lr(t, initial_lr, rampdown=0.25, rampup=0.05):
    """Learning-rate schedule of the drivers (:63-68).  Inert in literal mode (the optimizer never sees a gradient)."""
    lr_ramp = min(1, (1 - t) / rampdown)
    lr_ramp = 0.5 - 0.5 * math.cos(lr_ramp * math.pi)
    lr_ramp = lr_ramp * min(1, t / rampup)
    return initial_lr * lr_ramp


def noise_schedule(steps, latent_std, noise, noise_ramp):
    """sigma_i for i in range(steps) as float32, with the driver's roundings (:156): `latent_std` is a 0-d float32 TENSOR there, so
    `latent_std * args.noise` rounds to float32, the multiplication by the python-float ramp factor rounds to float32 again, and
    `.item()` hands that value to latent_noise (:157, :71-73), where `noise * strength` multiplies a float32 tensor by it."""
    base = np.float32(latent_std) * np.float32(noise)                                   # float32 x float32 -> float32, like the tensor op
    return np.array([base * np.float32(max(0, 1 - (i / steps) / noise_ramp) ** 2) for i in range(steps)], dtype=np.float32)


def latent_stats(G, n_mean_latent=10000, device="cuda", generator=None):
    """latent_mean [k,D] and latent_std scalar from N(0,I) samples (:251-255)."""
    samples = torch.randn(n_mean_latent, *G.input_shape[1:], device=device, generator=generator)
    mean = samples.mean(0)
    std = ((samples - mean).pow(2).sum() / n_mean_latent) ** 0.5
    return mean, std


def _as_latent(latent_mean, shape):
    """latent_mean as a tensor of `shape` = (k, D) or (k, num_ws, D): a [k, D] mean is broadcast over the layer slots of a W+ latent."""
    t = latent_mean.detach().clone().float()
    if len(shape) == 3 and t.numel() == shape[0] * shape[2]:
        t = t.reshape(shape[0], 1, shape[2]).expand(*shape)
    return t.reshape(*shape).contiguous()


def mapping_only(G, z):
    """w [n, k, D] = the mapping network on z [n, k, D] for ANY n, without touching the generator's synthesis workspaces: the kernel
    writes into a buffer of its own (G.mapping / G._mapping_into size the whole synthesis workspace for the batch -- 1.6 GB per sample at
    1024^2 -- which a statistics pass over 10 000 samples must never do)."""
    _lib.require_gpu(z)
    cfg = G.cfg
    assert tuple(z.shape[1:]) == (cfg.k, cfg.z_dim), tuple(z.shape)
    z = z.contiguous().float()
    w = torch.empty(z.shape[0], cfg.k, cfg.w_dim, dtype=torch.float32, device=z.device)
    _lib.check(_lib.lib().mgf_mapping_forward(w.data_ptr(), z.data_ptr(), G.plan.mapping_blob.data_ptr(), z.shape[0], cfg.k, cfg.w_dim,
                                              cfg.mapping_layers // 2, int(cfg.normalize_global), _lib.stream_ptr()), "mapping_forward")
    return w


def latent_stats_w(G, n_mean_latent=10000, device="cuda", generator=None):
    """Statistics of the INTERMEDIATE latent for a W+ search (what StyleGAN2-style projectors use: w_avg and w_std of mapped samples):
    mean [k, D] and the scalar std of G.mapping(z)[:, :, 0] over z ~ N(0, I), with the drivers' formula (:251-255) applied to w."""
    z = torch.randn(n_mean_latent, *G.input_shape[1:], device=device, generator=generator)
    w = mapping_only(G, z)
    mean = w.mean(0)
    std = ((w - mean).pow(2).sum() / n_mean_latent) ** 0.5
    return mean, std


class ProjectionEngine:
    """One target image <-> one latent search, replayable as a hipGraph."""

    def __init__(self, G, target, latent_mean, latent_std, args: ProjectionArgs = None, percept=None, use_mse=True,
                 lm_target=None, lm_steps=None, lm_valid=None, eps=None, noise_mode="random", seed=0, use_graph=True, batch=1,
                 landmark_fn=None, biometric=None, gamma=1.0, wing_kind="wing", landmark_model=None, pipeline=False, keep_images=0,
                 latent_shape=None, landmark_input="float", lbp_target=None):
        """batch = number of consecutive loop steps evaluated per generator forward.  In literal mode the steps do not depend
        on each other (latent_in never changes), so evaluating `batch` candidates at once and examining them in step order
        gives exactly the sequential loop's result while the small 4x4..64x64 layers, the mapping network and the LPIPS tail
        get `batch` times more parallel work per launch.

        wing_kind: "wing" (WingLoss(10, 2), the default drivers) or "awing" (AdaptiveWingLoss(14, 0.5, 1, 2.1) on the same landmark
        tensors, 1024_example_wing_loss_adaptive.py:176 with lamda = 1e-5).

        pipeline: software-pipeline consecutive batches over two streams -- while the losses and the selection of batch i run
        on a side stream, the generator already synthesises batch i+1 (its own latent / image buffers and step counter).  The
        literal loop's steps are independent, and selection still happens in step order, so results are unchanged; one extra
        generator batch is in flight at any time, and the run's last launch sequence scores its batch with nothing beside it (the
        first generator batch and the last loss phase are the two un-overlapped ends: B generator forwards for B scored batches).

        keep_images: K > 0 keeps the SCORED image of every improvement on the device (the drivers write `{step:06d}_{loss:04f}.png`
        of exactly that image -- random per-layer noise included -- at every improvement, :186-195): a trail of up to K images with
        their steps and losses, filled inside the launch sequence (no host round trip); after K improvements the last slot is
        overwritten, so the best-so-far image is always there.  `improvements()` returns the trail, `save_improvements()` writes it
        under the reference's file names.

        biometric: optional `iresnet.BiometricLoss`; adds gamma * MSE(embed(img), embed(target)) to the objective (the
        FaceNet term of 1024_example_FaceNet_percept.py:147-158 on the vendored IResNet embedder).

        landmark_model: optional DEVICE callable `m(img [B,3,H,W]) -> (landmarks [B,68,2] float64, valid [B] int32)` -- the GPU
        landmark-regressor interface of SURVEY.md 8f row 3.  It runs inside the launch sequence (graph-capturable if the model
        is), so the loop keeps its no-host-round-trip property; results are scattered into the landmark table on the device.

        landmark_fn: optional host callback `f(img_hwc float32 numpy [H,W,3]) -> [68,2] array or None` standing where the
        drivers call dlib on every generated image (:159-170; `drivers.reference_gray_u8` reproduces their cv2 normalise +
        gray conversion).  With it the landmark table is filled step by step (None = "no face", the step is skipped) at
        the price of one device->host image copy and a host call per candidate, and graph replay is off.

        landmark_input: what `landmark_fn` is handed.  "float": the float32 [H,W,3] image (12.6 MB per candidate at 1024^2 crosses to the
        host, the caller converts it itself).  "gray_u8": the uint8 [H,W] gray image the drivers build for dlib (:159-163, cv2.normalize +
        BGR2GRAY on RGB data), made ON THE DEVICE (mgf_reference_gray_u8, bit-identical to drivers.reference_gray_u8) and copied into
        pinned host memory -- 1 MB per candidate -- and the launch sequence stays two captured hipGraphs with the host detour between
        them: {perturb, generator, gray image, device->host copy} | callbacks, one host->device copy of the landmark rows | {losses,
        selection}."""
        self.G, self.args = G, args or ProjectionArgs()
        self.batch = int(batch)
        assert self.batch >= 1
        a = self.args
        dev = G.device
        self.device = dev
        _lib.require_gpu(target, latent_mean)
        self.steps = a.step
        self.target = target.detach().float().clone(memory_format=torch.contiguous_format)     # the engine's OWN copy: retarget() rewrites it in place
        assert self.target.shape[0] == 1, "one target per engine (the drivers process images serially)"
        self.percept = percept
        self.use_lbp = a.pixel_term == "lbp"
        self.use_mse = use_mse and not self.use_lbp
        self.use_wing = lm_target is not None
        if self.use_lbp and (lbp_target is None or self.use_wing or percept is not None or biometric is not None):
            raise _lib.MgfError("projection: pixel_term='lbp' is the whole objective of 1024_example_LBP_percept.py -- pass lbp_target "
                                "(lbp.target_feature of the target file) and no landmark / LPIPS / biometric term")
        self.noise_mode = noise_mode
        k, D = G.cfg.k, G.cfg.z_dim
        # (k, D): the drivers' z latent.  GradientProjectionEngine(latent_space="w+") passes (k, num_ws, D)
        self.latent_shape = ls = tuple(latent_shape) if latent_shape is not None else (k, D)
        self.numel = int(np.prod(ls))
        self.latent_in = _as_latent(latent_mean, ls).reshape(1, *ls).contiguous().float()
        sig = noise_schedule(a.step, float(latent_std), a.noise, a.noise_ramp)
        self.sigma = torch.as_tensor(sig, device=dev)
        self.copies = int(a.latent_copies)
        if not 1 <= self.copies <= 255:
            raise ValueError(f"latent_copies must be 1 .. 255 (got {a.latent_copies})")
        if self.copies > 1 and self.numel % 32:
            # (torch reduces a width's last numel % 32 columns on another path with another summation order: the bit-exact mean is only
            # claimed for widths it reduces four vector registers at a time -- k D = 17 x 32 = 544 is one)
            raise ValueError(f"latent_copies > 1 needs a latent of a multiple of 32 elements (got {self.numel})")
        if eps is None:
            gen = torch.Generator(device=dev)
            gen.manual_seed(seed)
            eps = torch.randn(a.step, 1, *(((self.copies,) if self.copies > 1 else ()) + ls), device=dev, generator=gen)
        self.eps = eps.to(dev).contiguous().float()
        assert self.eps.shape[0] >= a.step and self.eps[0].numel() == self.numel * self.copies, \
            f"eps must be [steps, 1{', copies' if self.copies > 1 else ''}, *latent shape] (got {tuple(self.eps.shape)})"
        assert wing_kind in ("wing", "awing")
        self.wing_kind = wing_kind
        self.landmark_fn = landmark_fn
        self.landmark_model = landmark_model
        if landmark_model is not None:
            assert lm_target is not None and landmark_fn is None, "landmark_model needs lm_target and excludes landmark_fn"
            # + batch spare rows: the candidates of a ragged last batch that lie past the final step land there, not on real rows
            lm_steps = np.zeros((a.step + self.batch,) + tuple(np.shape(lm_target)), np.float64)
            lm_valid = np.zeros(a.step + self.batch, np.int32)
        assert landmark_input in ("float", "gray_u8"), landmark_input
        self.landmark_input = landmark_input
        self.callback_graphs = landmark_fn is not None and landmark_input == "gray_u8"
        if landmark_fn is not None:
            assert lm_target is not None, "landmark_fn needs the target image's landmarks (lm_target)"
            spare = self.batch if self.callback_graphs else 0          # (a ragged last batch's surplus candidates land on spare rows)
            lm_steps = np.zeros((a.step + spare,) + tuple(np.shape(lm_target)), np.float64)
            lm_valid = np.zeros(a.step + spare, np.int32)
            if not self.callback_graphs:
                use_graph = False
        if self.use_wing:
            self.lm_target = torch.as_tensor(lm_target, dtype=torch.float64, device=dev).contiguous()
            self.lm_steps = torch.as_tensor(lm_steps, dtype=torch.float64, device=dev).contiguous()
            assert self.lm_steps.shape[0] >= a.step and self.lm_steps.shape[1:] == self.lm_target.shape
        self.valid = None if lm_valid is None else torch.as_tensor(lm_valid, dtype=torch.int32, device=dev).contiguous()
        # device-resident loop state
        self.step_ctr = torch.zeros(1, dtype=torch.int32, device=dev)
        self.min_loss = torch.full([1], float(a.min_loss_init), dtype=torch.float64, device=dev)
        self.best_latent = torch.zeros(1, *ls, dtype=torch.float32, device=dev)
        self.best_step = torch.full([1], -1, dtype=torch.int32, device=dev)
        self.losses = torch.full([a.step], float("nan"), dtype=torch.float64, device=dev)
        B = self.batch
        self.latent_n = torch.empty(B, *ls, dtype=torch.float32, device=dev)
        self.p_loss = torch.zeros(B, dtype=torch.float32, device=dev)
        self.mse_loss = torch.zeros(B, dtype=torch.float32, device=dev)
        self.w_loss = torch.zeros(B, dtype=torch.float64, device=dev)
        self.scratch = torch.empty(B * int(_lib.lib().mgf_reduce_scratch_floats()), dtype=torch.float32, device=dev)
        self._arange = torch.arange(B, dtype=torch.int64, device=dev)
        assert a.pixel_term in ("mse", "psnr", "dssim", "lbp"), a.pixel_term
        assert a.psnr_layout in ("script", "aligned"), a.psnr_layout
        # what the pixel term reads as "the target": the image itself, or -- the PSNR script's element order -- its H-W-C stream under the
        # candidate's C-H-W indexing (one permuted copy, refreshed by retarget(); the kernel is the same aligned sum)
        self.pix_target = self._script_order(self.target) if (a.pixel_term == "psnr" and a.psnr_layout == "script") else self.target
        if self.use_lbp:
            from . import lbp
            r = G.cfg.img_resolution
            self.lbp_ws = lbp.LbpWorkspace(B, r, r, dev)
            self.lbp_codes = torch.as_tensor(lbp_target, dtype=torch.uint8, device=dev).reshape(-1).contiguous()
            assert self.lbp_codes.numel() == lbp.SIDE * lbp.SIDE, "lbp_target: the 224 x 224 code map of lbp.target_feature"
        self._init_pool(B)
        if a.pixel_term == "dssim" and use_mse:
            c, h, w = self.target.shape[-3:]
            nbytes = int(_lib.lib().mgf_dssim_scratch_bytes(B, c, h, w))
            if nbytes <= 0:
                raise _lib.MgfError(f"projection: pixel_term='dssim' needs images of at least 7x7 pixels, got {h}x{w}")
            self.dssim_scratch = torch.empty(nbytes // 8, dtype=torch.float64, device=dev)
        if self.percept is not None:
            self.percept.set_target(self.target)
        self.biometric, self.gamma = biometric, float(gamma)
        if biometric is not None:
            biometric.set_target(self.target)
        self.keep_images = int(keep_images)
        if self.keep_images > 0:
            per = G.cfg.img_channels * G.cfg.img_resolution ** 2
            self.trail_imgs = torch.empty(self.keep_images, per, dtype=torch.float32, device=dev)
            self.trail_steps = torch.full([self.keep_images], -1, dtype=torch.int32, device=dev)
            self.trail_losses = torch.zeros(self.keep_images, dtype=torch.float64, device=dev)
            self.trail_count = torch.zeros(1, dtype=torch.int32, device=dev)
            self.take_slot = torch.full([B], -1, dtype=torch.int32, device=dev)
            if self.keep_images < B:
                import warnings
                warnings.warn(f"ProjectionEngine: keep_images={self.keep_images} < batch={B}: one launch sequence can improve more often than "
                              "there are trail slots, and then only the latest improvement of the overflow keeps its image (the reference "
                              "writes a PNG at every improvement); use keep_images >= batch", stacklevel=2)
        self._spilled, self._seq_since_spill, self._trail_lost = [], 0, 0
        self.use_graph = use_graph
        self.graph = None
        if self.callback_graphs:
            c, r = G.cfg.img_channels, G.cfg.img_resolution
            assert c == 3, "the gray conversion is defined on 3-channel images"
            lshape = tuple(self.lm_target.shape)
            self.gray_dev = torch.empty(B, r, r, dtype=torch.uint8, device=dev)
            self.gray_host = torch.empty(B, r, r, dtype=torch.uint8).pin_memory()
            self.gray_scratch = torch.empty(B * int(_lib.lib().mgf_reference_gray_scratch_floats()), dtype=torch.float32, device=dev)
            self.lm_stage_host = torch.zeros(B, *lshape, dtype=torch.float64).pin_memory()
            self.ok_stage_host = torch.zeros(B, dtype=torch.int32).pin_memory()
            self.lm_stage = torch.zeros(B, *lshape, dtype=torch.float64, device=dev)
            self.ok_stage = torch.zeros(B, dtype=torch.int32, device=dev)
            self.cb_graphs = None
            self._host_step = 0                     # host mirror of step_ctr (the callbacks need the step numbers without a device read)
        self.pipeline = bool(pipeline) and landmark_fn is None
        if self.pipeline:
            if (G.n, G.lean) != (B, True):
                G._alloc(B, True)
            self.latent_ns = [self.latent_n, torch.empty_like(self.latent_n)]
            self.imgs = [G.img, torch.empty_like(G.img)]
            self.gen_ctr = torch.zeros(1, dtype=torch.int32, device=dev)          # step counter of the generator side
            self.loss_stream = torch.cuda.Stream(device=dev)
            self.graphs = [None, None]
            self.prime_graph = None
            self.tail_graphs = [None, None]                                       # losses only: the run's LAST launch sequence has no next batch to synthesise
            self._parity = 0
            self._primed = False
            self._seq_launched = 0                                                # host count of launch sequences since the last rewind

    def _init_pool(self, B):
        """projection_example_v1.py:150-155: images above `pool_above` pixels are block-averaged by height // pool_above in front of the
        image-space losses; the target comes at the pooled size."""
        a, G = self.args, self.G
        r = G.cfg.img_resolution
        self.pool_factor = r // a.pool_above if (a.pool_above and r > a.pool_above) else 1
        want = r // self.pool_factor
        if tuple(self.target.shape[-2:]) != (want, want):
            raise _lib.MgfError(f"projection: the target is {tuple(self.target.shape[-2:])}, the image-space losses see {want}x{want} "
                                f"(generator {r}x{r}, pool_above={a.pool_above})")
        if self.pool_factor > 1:
            f = self.pool_factor
            # the block mean as the library's own upfirdn2d: an f x f box filter of weight 1 / f^2, down-sampling by f, no padding
            self.pool_box = torch.full([f, f], 1.0 / (f * f), dtype=torch.float32, device=self.device)
            self.pooled = torch.empty(B, G.cfg.img_channels, want, want, dtype=torch.float32, device=self.device)

    def _pooled(self, img):
        if self.pool_factor == 1:
            return img
        from . import conv as cv
        return cv.upfirdn_into(self.pooled[:img.shape[0]], img, self.pool_box, up=1, down=self.pool_factor, pad=(0, 0, 0, 0), gain=1.0)

    # ------------------------------------------------------------------ one iteration
    def _iteration(self):
        """`batch` consecutive steps of the loop: perturb -> generator -> losses -> in-order best-so-far selection."""
        img = self._gen_phase(self.latent_n, self.step_ctr)
        self._loss_phase(img, self.latent_n)
        return img

    def _gen_phase(self, latent_n, ctr):
        L, st, a, B = _lib.lib(), _lib.stream_ptr(), self.args, self.batch
        if self.copies > 1:       # the v2 driver's averaged copies (projection_example_v2_percept.py:147-159)
            _lib.check(L.mgf_latent_perturb_mean(latent_n.data_ptr(), self.latent_in.data_ptr(), self.eps.data_ptr(), self.sigma.data_ptr(),
                                                 ctr.data_ptr(), B, self.steps, self.numel, self.copies, st), "latent_perturb_mean")
        else:
            _lib.check(L.mgf_latent_perturb(latent_n.data_ptr(), self.latent_in.data_ptr(), self.eps.data_ptr(),
                                            self.sigma.data_ptr(), ctr.data_ptr(), B, self.steps, self.numel, st), "latent_perturb")
        # psi lands in `c`.  lean: the literal loop has no backward pass -- layer outputs share arenas block after block
        return self.G.forward_workspace(latent_n, a.truncation_psi, noise_mode=self.noise_mode, lean=True)[0]

    def _loss_phase(self, img, latent_n):
        L, st, a, B = _lib.lib(), _lib.stream_ptr(), self.args, self.batch
        full_img, img = img, self._pooled(img)              # (the improvement trail keeps the image as generated)
        if self.percept is not None:
            self.percept.distance_into(self.p_loss, img)
            if a.percept_weight != 1.0:
                self.p_loss.mul_(float(a.percept_weight))
        if self.biometric is not None:      # rides in the p_loss slot: p_loss = LPIPS + gamma * embedding MSE
            self.biometric.distance_into(self.p_loss, img, scale=self.gamma, accumulate=self.percept is not None)
        if self.use_mse and a.pixel_term == "dssim":
            c, h, w = img.shape[1:]
            _lib.check(L.mgf_dssim_u8_f32(self.mse_loss.data_ptr(), img.data_ptr(), self.target.data_ptr(), B, c, h, w, 0, 255.0, 1.0, 0,
                                          self.dssim_scratch.data_ptr(), st), "dssim")
        elif self.use_mse:
            per = img.numel() // B
            _lib.check(L.mgf_mse_f32(self.mse_loss.data_ptr(), img.data_ptr(), self.pix_target.data_ptr(), B, per, 0, 1.0, 0,
                                     self.scratch.data_ptr(), st), "mse")
            if a.pixel_term == "psnr":
                # 10 * np.log10(peak ** 2 / np.mean(d ** 2)) with peak = 255., in float32 like the script's numpy (1024_example_PSNR.py:113-114)
                self.mse_loss.reciprocal_().mul_(65025.0).log10_().mul_(10.0)
        self._landmarks(full_img)
        if self.use_lbp:        # rides in the float64 slot (the script compares float64 distances, :166-172), coefficient 1
            self.lbp_ws.distance_into(self.w_loss, full_img, self.lbp_codes)
        if self.use_wing and self.wing_kind == "wing":
            _lib.check(L.mgf_wing_loss_f64(self.w_loss.data_ptr(), self.lm_steps.data_ptr(), self.lm_target.data_ptr(), B,
                                           self.lm_target.numel(), 10.0, 2.0, self.step_ctr.data_ptr(), self.lm_steps.shape[0] - 1, st),
                       "wing_loss")
        elif self.use_wing:
            _lib.check(L.mgf_adaptive_wing_loss_f64(self.w_loss.data_ptr(), self.lm_steps.data_ptr(), self.lm_target.data_ptr(), B,
                                                    self.lm_target.numel(), 14.0, 0.5, 1.0, 2.1, self.step_ctr.data_ptr(),
                                                    self.lm_steps.shape[0] - 1, st), "adaptive_wing_loss")
        keep = self.keep_images > 0
        _lib.check(L.mgf_select_best(self.min_loss.data_ptr(), self.best_latent.data_ptr(), self.best_step.data_ptr(),
                                     self.losses.data_ptr(), latent_n.data_ptr(), self.numel,
                                     _lib.ptr(self.p_loss if (self.percept is not None or self.biometric is not None) else None),
                                     _lib.ptr(self.w_loss if (self.use_wing or self.use_lbp) else None),
                                     _lib.ptr(self.mse_loss if self.use_mse else None), 1.0 if self.use_lbp else float(a.lamda), float(a.beta),
                                     self.step_ctr.data_ptr(), _lib.ptr(self.valid), B, self.steps,
                                     _lib.ptr(self.take_slot if keep else None), _lib.ptr(self.trail_count if keep else None),
                                     self.keep_images, _lib.ptr(self.trail_steps if keep else None),
                                     _lib.ptr(self.trail_losses if keep else None), st), "select_best")
        if keep:
            _lib.check(L.mgf_keep_improvements(self.trail_imgs.data_ptr(), full_img.data_ptr(), self.trail_imgs.shape[1],
                                               self.take_slot.data_ptr(), B, st), "keep_improvements")

    def _landmarks(self, img):
        """Fill this batch's rows of the landmark / valid tables when a detector is attached (host callback or device model)."""
        B = self.batch
        if self.landmark_fn is not None and self.callback_graphs:
            # the callbacks ran on the host between the two launch sequences (_run_callback); their rows wait in the device staging buffers
            idx = self.step_ctr.long() + self._arange
            self.lm_steps.index_copy_(0, idx, self.lm_stage)
            self.valid.index_copy_(0, idx, self.ok_stage)
        elif self.landmark_fn is not None:
            self._detect_landmarks(img)
        if self.landmark_model is not None:
            lm, ok = self.landmark_model(img)
            idx = self.step_ctr.long() + self._arange                                    # rows of this batch's steps, on the device
            self.lm_steps.index_copy_(0, idx, lm.to(torch.float64).reshape(B, *self.lm_target.shape))
            self.valid.index_copy_(0, idx, ok.to(torch.int32).reshape(B))

    # ------------------------------------------------------------------ pipelined mode
    def _pipe_gen(self, p):
        """Generator side of batch parity p (own latent / image buffer, own step counter)."""
        self.G.img = self.imgs[p]
        img = self._gen_phase(self.latent_ns[p], self.gen_ctr)
        self.gen_ctr.add_(self.batch).clamp_(max=self.steps)
        return img

    def _pipe_step(self, p):
        """Steady state: losses + selection of the batch in buffers p on the loss stream, the generator of the next batch in buffers
        p ^ 1 on the current stream; joined at the end."""
        main = torch.cuda.current_stream(self.device)
        self.loss_stream.wait_stream(main)
        with torch.cuda.stream(self.loss_stream):
            self._loss_phase(self.imgs[p], self.latent_ns[p])
        self._pipe_gen(p ^ 1)
        main.wait_stream(self.loss_stream)

    def _pipe_tail(self, p):
        """The run's last launch sequence: losses + selection of the batch in buffers p, nothing left to synthesise beside them."""
        self._loss_phase(self.imgs[p], self.latent_ns[p])

    def _pipe_capture(self):
        tensors = self._state() + (self.gen_ctr,)
        state = [t.clone() for t in tensors]
        s = torch.cuda.Stream(device=self.device)
        s.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(s):
            self._pipe_step(0)
            self._pipe_step(1)
        torch.cuda.current_stream(self.device).wait_stream(s)
        torch.cuda.synchronize(self.device)
        for p in (0, 1):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self._pipe_step(p)
            self.graphs[p] = g
            t = torch.cuda.CUDAGraph()
            with torch.cuda.graph(t, pool=g.pool()):
                self._pipe_tail(p)
            self.tail_graphs[p] = t
        # the first batch of a run (nothing to score beside it yet) as a graph of its own: a re-targeted engine starts every run with it
        self.prime_graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.prime_graph, pool=self.graphs[0].pool()):
            self._pipe_gen(0)
        for dst, src in zip(tensors, state):
            dst.copy_(src)
        self._pin_workspace()

    def _run_pipelined(self, n):
        if self.use_graph and self.graphs[0] is None:
            self._pipe_capture()                                          # (its warm-up overwrites both image buffers: capture first)
        if not self._primed:
            if self.use_graph and self._parity == 0:
                self.prime_graph.replay()                                 # the first batch has no losses to overlap with
            else:
                self._pipe_gen(self._parity)
            self._primed = True
        last = -(-self.steps // self.batch) - 1                          # index of the sequence that scores the run's last candidates
        for _ in range(n):
            self._before_sequence()
            final = self._seq_launched >= last                            # (and anything a caller launches past the end: nothing left to score)
            if self.use_graph:
                (self.tail_graphs if final else self.graphs)[self._parity].replay()
            elif final:
                self._pipe_tail(self._parity)
            else:
                self._pipe_step(self._parity)
            self._parity ^= 1
            self._seq_launched += 1

    # ------------------------------------------------------------------ callback mode on the gray uint8 image
    def _cb_phase_a(self):
        """perturb -> generator -> the drivers' gray uint8 image of every candidate -> pinned host memory (all asynchronous)."""
        img = self._gen_phase(self.latent_n, self.step_ctr)
        n, _, h, w = img.shape
        _lib.check(_lib.lib().mgf_reference_gray_u8(self.gray_dev.data_ptr(), img.data_ptr(), n, h, w, self.gray_scratch.data_ptr(),
                                                    _lib.stream_ptr()), "reference_gray_u8")
        self.gray_host.copy_(self.gray_dev, non_blocking=True)
        return img

    def _cb_phase_b(self):
        self._loss_phase(self.G.img, self.latent_n)

    def _cb_capture(self):
        state = [t.clone() for t in self._state()]
        s = torch.cuda.Stream(device=self.device)
        s.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(s):
            self._cb_phase_a()
            self._cb_phase_b()
        torch.cuda.current_stream(self.device).wait_stream(s)
        torch.cuda.synchronize(self.device)
        ga, gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        with torch.cuda.graph(ga):
            self._cb_phase_a()
        with torch.cuda.graph(gb, pool=ga.pool()):
            self._cb_phase_b()
        for dst, src in zip(self._state(), state):
            dst.copy_(src)
        if self.use_wing:
            self.lm_steps.zero_()
            self.valid.zero_()
        self.cb_graphs = (ga, gb)
        self._pin_workspace()

    def _run_callback(self, n):
        """`n` launch sequences of the callback mode: phase A, one stream synchronisation, the host detector on every candidate's gray image
        (in step order; None = "no face", the step is skipped, :165-166), ONE host->device copy of the batch's landmark rows, phase B."""
        if self.use_graph and self.cb_graphs is None:
            self._cb_capture()
        B, st = self.batch, torch.cuda.current_stream(self.device)
        for _ in range(n):
            self._before_sequence()
            if self.cb_graphs is not None:
                self.cb_graphs[0].replay()
            else:
                self._cb_phase_a()
            st.synchronize()
            s0 = self._host_step
            gray = self.gray_host.numpy()
            self.ok_stage_host.zero_()
            for j in range(min(B, self.steps - s0)):
                lm = self.landmark_fn(gray[j])
                if lm is None:
                    continue
                self.lm_stage_host[j].copy_(torch.as_tensor(np.asarray(lm, dtype=np.float64).reshape(self.lm_target.shape)))
                self.ok_stage_host[j] = 1
            self.lm_stage.copy_(self.lm_stage_host, non_blocking=True)
            self.ok_stage.copy_(self.ok_stage_host, non_blocking=True)
            if self.cb_graphs is not None:
                self.cb_graphs[1].replay()
            else:
                self._cb_phase_b()
            self._host_step = min(s0 + B, self.steps)

    def _detect_landmarks(self, img):
        """Host detour of the callback mode: hand every candidate image of this batch to `landmark_fn`, in step order."""
        s0 = int(self.step_ctr.item())                                     # host sync (the reference syncs three times per step)
        host = img.permute(0, 2, 3, 1).contiguous().cpu().numpy()         # [B,H,W,3] float32, what the drivers build at :159-161
        for j in range(min(self.batch, self.steps - s0)):
            lm = self.landmark_fn(host[j])
            if lm is None:
                continue
            self.lm_steps[s0 + j].copy_(torch.as_tensor(np.asarray(lm, dtype=np.float64).reshape(self.lm_target.shape)))
            self.valid[s0 + j] = 1

    def _state(self):
        """Device tensors that make up the loop state (saved and restored around the capture warm-up)."""
        st = (self.step_ctr, self.min_loss, self.best_latent, self.best_step, self.losses)
        if self.keep_images > 0:
            st += (self.trail_count, self.trail_steps, self.trail_losses)
        return st

    def _spill_trail(self):
        """Move the filled trail slots to host memory and empty the device trail (between launch sequences; one host sync).  `run()`
        calls this whenever the NEXT launch sequence could overflow the K slots -- a sequence improves at most `batch` times -- so with
        keep_images >= batch every improvement of a run of any length keeps its scored image, like the reference's PNG per improvement
        (...sqz_MSE.py:186-195), for K x 12.6 MB of device memory instead of one slot per step."""
        torch.cuda.synchronize(self.device)
        total = int(self.trail_count.item())
        n = min(total, self.keep_images)
        self._trail_lost += total - n
        if n:
            steps, losses = self.trail_steps[:n].cpu().numpy(), self.trail_losses[:n].cpu().numpy()
            imgs = self.trail_imgs[:n].cpu()
            c, r = self.G.cfg.img_channels, self.G.cfg.img_resolution
            self._spilled += [(int(steps[i]), float(losses[i]), imgs[i].view(c, r, r)) for i in range(n)]
        self.trail_count.zero_()
        self.trail_steps.fill_(-1)
        self._seq_since_spill = 0

    def _before_sequence(self):
        """Trail bookkeeping in front of every launch sequence (see _spill_trail)."""
        if self.keep_images > 0:
            if self._seq_since_spill > 0 and (self._seq_since_spill + 1) * self.batch > self.keep_images:
                self._spill_trail()
            self._seq_since_spill += 1

    def improvements(self):
        """The improvement trail of a keep_images=K engine: list of (step, loss, image [C,H,W]) in the order the improvements
        happened -- the (step, min_loss, img_gen_raw) triples the drivers turn into PNG files (:186-195).  Images still resident in the
        device trail are device tensors, those already spilled to the host (see _spill_trail) CPU tensors.  With keep_images >= batch
        the list is complete; otherwise a launch sequence with more than K improvements kept its first K-1 and its latest (a warning
        says how many images were lost)."""
        assert self.keep_images > 0, "construct the engine with keep_images=K"
        torch.cuda.synchronize(self.device)
        total = int(self.trail_count.item())
        n = min(total, self.keep_images)
        lost = self._trail_lost + total - n
        if lost:
            import warnings
            warnings.warn(f"ProjectionEngine: {lost} improvement image(s) were overwritten: keep_images={self.keep_images} < batch={self.batch} "
                          "(the reference writes one PNG per improvement)", stacklevel=2)
        c, r = self.G.cfg.img_channels, self.G.cfg.img_resolution
        steps, losses = self.trail_steps.cpu().numpy(), self.trail_losses.cpu().numpy()
        return list(self._spilled) + [(int(steps[i]), float(losses[i]), self.trail_imgs[i].view(c, r, r)) for i in range(n)]

    def save_improvements(self, output_dir, ratio=1.0):
        """Write the trail as the drivers do: `{output_dir}/{step:06d}_{loss:04f}.png` of crop(to_pil(img), ratio) (:190-195)."""
        from .drivers import save_image
        paths = []
        for step, loss, img in self.improvements():
            paths.append(save_image(self.G, img.unsqueeze(0).to(self.device), os.path.join(output_dir, "{:06d}_{:04f}.png".format(step, loss)), ratio))
        return paths

    def _capture(self):
        # warm-up on a side stream (allocations, lazy init) before capture, then restore the loop state
        state = [t.clone() for t in self._state()]
        s = torch.cuda.Stream(device=self.device)
        s.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(s):
            self._iteration()
        torch.cuda.current_stream(self.device).wait_stream(s)
        torch.cuda.synchronize(self.device)
        # graph_debug: keep the hipGraph_t behind the executable graph, so that its nodes can be counted afterwards (bench.py)
        keep = bool(getattr(self, "graph_debug", False))
        g = torch.cuda.CUDAGraph(keep_graph=True) if keep else torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self._iteration()
        if keep:
            g.instantiate()
        for dst, src in zip(self._state(), state):
            dst.copy_(src)
        self.graph = g
        self._pin_workspace()

    def _pin_workspace(self):
        """The captured graph references the generator's workspace of this batch size: keep it alive as long as this engine lives."""
        import weakref
        G = self.G
        weakref.finalize(self, G.unpin, G.pin())

    def run(self, steps=None):
        """Advance the loop by `steps` iterations (default: all remaining), `batch` of them per launch sequence."""
        done = int(self.step_ctr.item()) if steps is None else None
        n = (self.steps - done) if steps is None else steps
        n = (n + self.batch - 1) // self.batch
        if self.pipeline:
            self._run_pipelined(n)
            return self
        if self.callback_graphs:
            self._run_callback(n)
            return self
        if self.use_graph and self.graph is None:
            self._capture()
        for _ in range(n):
            self._before_sequence()
            if self.graph is not None:
                self.graph.replay()
            else:
                self._iteration()
        return self

    @staticmethod
    def _script_order(target):
        """[1, C, H, W] tensor whose C-H-W stream is the target's H-W-C stream (`tensor2np(imgs).flatten()`, 1024_example_PSNR.py:122,153,158)."""
        return target.permute(0, 2, 3, 1).contiguous().view(target.shape)

    def retarget(self, target, lm_target=None, lm_steps=None, lm_valid=None, eps=None, seed=None, latent_mean=None, latent_std=None,
                 lbp_target=None):
        """Point this engine at ANOTHER target image and rewind the loop, keeping everything that was expensive to set up: the captured
        hipGraph(s), the generator workspace, the LPIPS / embedder workspaces, the loss scratch.  Only data changes, in place, in the
        buffers the graph already references: the target image and its cached LPIPS taps / embedding, the landmark tables, the noise
        stream (`eps` given, or redrawn from `seed`), optionally the start latent and the noise schedule, and the loop state (step
        counter, best-so-far, loss history, improvement trail).  A run after retarget() equals the run of a freshly constructed engine
        on the same inputs bit for bit (tests/test_hip_drivers.py).  This is what the reference's serial per-image loop amortises by
        keeping G / percept / latent statistics outside `projection()` (projection_example_v2_percept_morph.py:311-355); BASELINE
        configs 3 and 5 are batches of targets."""
        a, dev = self.args, self.device
        _lib.require_gpu(target)
        assert tuple(target.shape) == tuple(self.target.shape), (tuple(target.shape), tuple(self.target.shape))
        torch.cuda.synchronize(dev)
        self.target.copy_(target)
        if self.pix_target is not self.target:
            self.pix_target.copy_(self._script_order(self.target))
        if self.percept is not None:
            self.percept.set_target(self.target)                  # same shapes: rewrites the cached taps in place
        if self.biometric is not None:
            self.biometric.set_target(self.target)
        if self.use_lbp:
            assert lbp_target is not None, "this engine scores the LBP distance: pass the new target's code map (lbp.target_feature)"
            self.lbp_codes.copy_(torch.as_tensor(lbp_target, dtype=torch.uint8).reshape(-1))
        if self.use_wing:
            assert lm_target is not None, "this engine has a Wing term: pass the new target's landmarks"
            self.lm_target.copy_(torch.as_tensor(lm_target, dtype=torch.float64).reshape(self.lm_target.shape))
            if self.landmark_fn is not None or self.landmark_model is not None:
                self.lm_steps.zero_()
                self.valid.zero_()
            else:
                assert lm_steps is not None, "pass the per-step landmark table of the new target"
                t = torch.as_tensor(lm_steps, dtype=torch.float64)
                assert t.shape[0] >= a.step and tuple(t.shape[1:]) == tuple(self.lm_target.shape)
                self.lm_steps[:a.step].copy_(t[:a.step])
                if self.valid is not None:
                    self.valid.copy_(torch.ones(self.valid.shape, dtype=torch.int32) if lm_valid is None
                                     else torch.as_tensor(lm_valid, dtype=torch.int32).reshape(self.valid.shape))
                else:
                    assert lm_valid is None, "the engine was built without a `valid` table; construct it with lm_valid to use one"
        if eps is not None:
            assert eps.shape[0] >= a.step
            self.eps[:a.step].copy_(eps[:a.step].reshape(a.step, *self.eps.shape[1:]))
        elif seed is not None:
            gen = torch.Generator(device=dev)
            gen.manual_seed(seed)
            self.eps.copy_(torch.randn(self.eps.shape, device=dev, generator=gen))       # same draw as the constructor's
        if latent_mean is not None:
            self.latent_in.copy_(latent_mean.detach().reshape(self.latent_in.shape))
        if latent_std is not None:
            self.sigma.copy_(torch.as_tensor(noise_schedule(a.step, float(latent_std), a.noise, a.noise_ramp)))
        self.rewind()
        return self

    def rewind(self):
        """Loop state back to step 0 (graph, workspaces and inputs untouched)."""
        a = self.args
        self.step_ctr.zero_()
        self.min_loss.fill_(float(a.min_loss_init))
        self.best_latent.zero_()
        self.best_step.fill_(-1)
        self.losses.fill_(float("nan"))
        if self.keep_images > 0:
            self.trail_count.zero_()
            self.trail_steps.fill_(-1)
            self.trail_losses.zero_()
        self._spilled, self._seq_since_spill, self._trail_lost = [], 0, 0
        if self.callback_graphs:
            self._host_step = 0
        if self.pipeline:
            self.gen_ctr.zero_()
            self._parity, self._primed, self._seq_launched = 0, False, 0
        return self

    def result(self):
        """(best_latent [1,k,D] cpu, best_step int, best_loss float, losses [steps] float64 numpy).  Raises like the
        reference (`latent_path[-1]` on an empty list, :208) when no step ever improved on min_loss_init."""
        torch.cuda.synchronize(self.device)
        bs = int(self.best_step.item())
        if bs < 0:
            raise IndexError("projection: no iteration improved on the initial min_loss (reference: latent_path[-1] on an empty list)")
        return self.best_latent.cpu().clone(), bs, float(self.min_loss.item()), self.losses.cpu().numpy()


class GradientProjectionEngine(ProjectionEngine):
    """Gradient mode: the same loop with the loss back-propagated into the latent.

        latent_n = latent_in + eps_i * sigma_i;  img = G(latent_n)
        total = percept_weight * LPIPS + lamda * Wing + beta * MSE [+ gamma * embedding MSE]     (Wing's landmarks come from a
                                                                                                 detector: no gradient)
        latent_in <- Adam(lr_i = get_lr(i / steps)).step(d total / d latent_in);  keep (latent_n, i) if total < min_loss

    i.e. the drivers' loop (...sqz_MSE.py:143-189) with the `.cpu().detach().numpy()` at :158 removed.  One candidate per target
    and step (the steps now depend on each other); forward, losses, backward (grad.GeneratorGrad, LPIPS / embedder backward),
    Adam and the best-of bookkeeping are one device-resident launch sequence, replayed as a hipGraph.  The oracle is torch
    autograd + torch.optim.Adam through the CPU restatement (oracle/loss_ref.py: projection_gradient_ref).

    `target` may hold B > 1 images: B independent projections (own latent, Adam state, noise stream, landmarks, best-so-far)
    advance in lockstep through ONE generator forward/backward per step -- the batch that lifts the 4x4..64x64 layers off
    their batch-1 latency floor (BASELINE configs 3 and 5 are batches of independent targets).  lm_target is then [B,68,2],
    lm_steps [B,steps,68,2], lm_valid [B,steps], eps [steps,B,k,D]; `result()` returns per-target lists."""

    def __init__(self, G, target, latent_mean, latent_std, args: ProjectionArgs = None, percept=None, use_mse=True, lm_target=None,
                 lm_steps=None, lm_valid=None, eps=None, noise_mode="random", seed=0, use_graph=True, landmark_fn=None, biometric=None,
                 gamma=1.0, wing_kind="wing", landmark_model=None, betas=(0.9, 0.999), adam_eps=1e-8, weight_decay=0.0,
                 latent_space="z", **ignored):
        """latent_space: "z" -- the drivers' parameter, the gradient runs on through the mapping network -- or "w+": the parameter is the
        per-layer intermediate latent ws [k, num_ws, D] itself (north_star: "backprops into the k-component latent W+"; layer `slot` reads
        ws[:, slot], networks.py:1252-1253), perturbed, descended by Adam and kept best-of exactly like z.  latent_mean is then a w-space
        start -- [k, D] (broadcast over the slots, e.g. latent_stats_w's mean) or [k, num_ws, D] -- and latent_std a w-space scale.  The
        reference has no such driver (its "W+" averages 18 copies of z, projection_example_v2_percept.py:133-162); the oracle is torch autograd
        + Adam on ws through the CPU restatement (tests/test_hip_gradient.py)."""
        from .grad import GeneratorGrad
        assert latent_space in ("z", "w+"), latent_space
        self.latent_space = latent_space
        ls = (G.cfg.k, G.cfg.num_ws, G.cfg.w_dim) if latent_space == "w+" else (G.cfg.k, G.cfg.z_dim)
        B = int(target.shape[0])
        if B == 1:
            super().__init__(G, target, latent_mean, latent_std, args, percept=percept, use_mse=use_mse, lm_target=lm_target,
                             lm_steps=lm_steps, lm_valid=lm_valid, eps=eps, noise_mode=noise_mode, seed=seed, use_graph=use_graph, batch=1,
                             landmark_fn=landmark_fn, biometric=biometric, gamma=gamma, wing_kind=wing_kind,
                             landmark_model=landmark_model, pipeline=False, latent_shape=ls)
            self.lm_tables = [self.lm_steps] if self.use_wing else None
        else:
            self._init_multi(G, target, latent_mean, latent_std, args, percept, use_mse, lm_target, lm_steps, lm_valid, eps, noise_mode,
                             seed, use_graph, biometric, gamma, wing_kind, ls)
            assert landmark_fn is None and landmark_model is None, "landmark detectors are wired for one target per engine"
        self.targets = B
        a, dev = self.args, self.device
        if a.latent_copies != 1:
            raise _lib.MgfError("GradientProjectionEngine: latent_copies is the literal-mode v2 driver's averaged latent (projection_example_v2_percept.py); "
                                "gradient mode optimises one latent")
        if a.pixel_term != "mse" or a.pool_above:
            raise _lib.MgfError("GradientProjectionEngine: pixel_term='psnr' / 'dssim' / 'lbp' / pool_above are literal-mode objectives (the PSNR / v1 drivers "
                                "sever the gradient like every other driver; only the Wing / LPIPS / MSE / biometric terms have backward passes)")
        if biometric is not None and hasattr(biometric.embedder, "keep_activations"):
            biometric.embedder.keep_activations = True          # (the FaceNet embedder re-uses its buffers block after block otherwise)
        self.gg = GeneratorGrad(G)
        self.betas, self.adam_eps, self.weight_decay = betas, float(adam_eps), float(weight_decay)
        self.lr_table = torch.as_tensor(np.array([get_lr(i / a.step, a.lr, a.lr_rampdown, a.lr_rampup) for i in range(a.step)],
                                                 dtype=np.float32), device=dev)
        self.exp_avg = torch.zeros_like(self.latent_in)
        self.exp_avg_sq = torch.zeros_like(self.latent_in)
        self.adam_t = torch.zeros(B, dtype=torch.int32, device=dev)
        self.dimg = torch.zeros_like(self.target)

    def _init_multi(self, G, target, latent_mean, latent_std, args, percept, use_mse, lm_target, lm_steps, lm_valid, eps, noise_mode,
                    seed, use_graph, biometric, gamma, wing_kind, ls):
        """Loop state of B lockstep projections (the single-target layout of ProjectionEngine with a leading B axis)."""
        self.G, self.args = G, args or ProjectionArgs()
        a, dev = self.args, G.device
        B = int(target.shape[0])
        self.device, self.batch, self.steps = dev, 1, a.step
        _lib.require_gpu(target, latent_mean)
        self.target = target.detach().float().clone(memory_format=torch.contiguous_format)     # the engine's OWN copy: retarget() rewrites it in place
        self.percept, self.use_mse, self.use_wing, self.noise_mode = percept, use_mse, lm_target is not None, noise_mode
        self.latent_shape = ls
        self.numel = int(np.prod(ls))
        lm0 = latent_mean.detach().float()
        lm0 = lm0.reshape(B, *ls) if lm0.numel() == B * self.numel else _as_latent(lm0, ls).reshape(1, *ls).expand(B, *ls)
        self.latent_in = lm0.contiguous().clone()
        self.sigma = torch.as_tensor(noise_schedule(a.step, float(latent_std), a.noise, a.noise_ramp), device=dev)
        if eps is None:
            gen = torch.Generator(device=dev)
            gen.manual_seed(seed)
            eps = torch.randn(a.step, B, *ls, device=dev, generator=gen)
        self.eps = eps.to(dev).contiguous().float()
        assert tuple(self.eps.shape) == (self.eps.shape[0], B, *ls) and self.eps.shape[0] >= a.step
        assert wing_kind in ("wing", "awing")
        self.wing_kind, self.landmark_fn, self.landmark_model = wing_kind, None, None
        self.landmark_input, self.callback_graphs = "float", False
        if self.use_wing:
            self.lm_target = torch.as_tensor(lm_target, dtype=torch.float64, device=dev).contiguous()           # [B,68,2]
            self.lm_steps = torch.as_tensor(lm_steps, dtype=torch.float64, device=dev).contiguous()             # [B,steps,68,2]
            assert self.lm_target.shape[0] == B and self.lm_steps.shape[:2] == (B, self.lm_steps.shape[1]) and self.lm_steps.shape[1] >= a.step
            self.lm_tables = [self.lm_steps[j] for j in range(B)]
        else:
            self.lm_tables = None
        self.valid = None if lm_valid is None else torch.as_tensor(lm_valid, dtype=torch.int32, device=dev).reshape(B, -1).contiguous()
        self.step_ctr = torch.zeros(B, dtype=torch.int32, device=dev)              # one copy per target (select advances its own)
        self.min_loss = torch.full([B], float(a.min_loss_init), dtype=torch.float64, device=dev)
        self.best_latent = torch.zeros(B, *ls, dtype=torch.float32, device=dev)
        self.best_step = torch.full([B], -1, dtype=torch.int32, device=dev)
        self.losses = torch.full([B, a.step], float("nan"), dtype=torch.float64, device=dev)
        self.latent_n = torch.empty(B, *ls, dtype=torch.float32, device=dev)
        self.p_loss = torch.zeros(B, dtype=torch.float32, device=dev)
        self.mse_loss = torch.zeros(B, dtype=torch.float32, device=dev)
        self.w_loss = torch.zeros(B, dtype=torch.float64, device=dev)
        self.scratch = torch.empty(B * int(_lib.lib().mgf_reduce_scratch_floats()), dtype=torch.float32, device=dev)
        if percept is not None:
            percept.set_target(self.target)
        self.biometric, self.gamma = biometric, float(gamma)
        if biometric is not None:
            biometric.set_target(self.target)                 # one embedding per target
        self.use_graph, self.graph, self.pipeline, self.keep_images = use_graph, None, False, 0
        self.pool_factor = 1

    def _state(self):
        return super()._state() + (self.latent_in, self.exp_avg, self.exp_avg_sq, self.adam_t)

    def _iteration(self):
        L, st, a, B = _lib.lib(), _lib.stream_ptr(), self.args, self.targets
        # B targets: one flat parameter of B * numel floats, one noise row [B * numel] per step
        _lib.check(L.mgf_latent_perturb(self.latent_n.data_ptr(), self.latent_in.data_ptr(), self.eps.data_ptr(), self.sigma.data_ptr(),
                                        self.step_ctr.data_ptr(), 1, self.steps, B * self.numel, st), "latent_perturb")
        if self.latent_space == "w+":
            img = self.gg.forward(ws=self.latent_n, noise_mode=self.noise_mode)   # [B, k, num_ws, D]: every layer reads its own slot
        else:
            img = self.gg.forward(self.latent_n, noise_mode=self.noise_mode)      # psi lands in `c` in the drivers: no truncation
        if B == 1:
            self._landmarks(img)                                                  # before Adam: a "no face" step must not move the latent
        per = img.numel() // B
        tstride = per if B > 1 else 0
        if self.use_mse:
            _lib.check(L.mgf_mse_grad_f32(self.dimg.data_ptr(), img.data_ptr(), self.target.data_ptr(), B, per, tstride, float(a.beta), 0,
                                          st), "mse_grad")
        else:
            self.dimg.zero_()
        if self.percept is not None:
            self.percept.distance_into(self.p_loss, img, keep_taps=True)
            self.percept.grad_into(self.dimg, scale=float(a.percept_weight), accumulate=True)
            if a.percept_weight != 1.0:
                self.p_loss.mul_(float(a.percept_weight))
        if self.biometric is not None:      # rides in the p_loss slot, like the literal loop
            self.biometric.distance_into(self.p_loss, img, scale=self.gamma, accumulate=self.percept is not None)
            self.biometric.grad_into(self.dimg, scale=self.gamma, accumulate=True)
        if self.use_mse:
            _lib.check(L.mgf_mse_f32(self.mse_loss.data_ptr(), img.data_ptr(), self.target.data_ptr(), B, per, tstride, 1.0, 0,
                                     self.scratch.data_ptr(), st), "mse")
        dz = self.gg.backward_ws(self.dimg) if self.latent_space == "w+" else self.gg.backward(self.dimg)
        has_p = self.percept is not None or self.biometric is not None
        for j in range(B):                   # per-target optimizer step, Wing term and best-of bookkeeping (tiny launches)
            ctr = self.step_ctr[j:]
            valid = None if self.valid is None else (self.valid[j] if B > 1 else self.valid)
            _lib.check(L.mgf_adam_step_f32(self.latent_in[j:].data_ptr(), self.exp_avg[j:].data_ptr(), self.exp_avg_sq[j:].data_ptr(),
                                           self.adam_t[j:].data_ptr(), dz[j:].data_ptr(), self.lr_table.data_ptr(), ctr.data_ptr(),
                                           _lib.ptr(valid), self.numel, self.steps, float(self.betas[0]), float(self.betas[1]),
                                           self.adam_eps, self.weight_decay, st), "adam_step")
            if self.use_wing:
                tab, tgt = self.lm_tables[j], (self.lm_target[j] if B > 1 else self.lm_target)
                if self.wing_kind == "wing":
                    _lib.check(L.mgf_wing_loss_f64(self.w_loss[j:].data_ptr(), tab.data_ptr(), tgt.data_ptr(), 1, tgt.numel(), 10.0, 2.0,
                                                   ctr.data_ptr(), tab.shape[0] - 1, st), "wing_loss")
                else:
                    _lib.check(L.mgf_adaptive_wing_loss_f64(self.w_loss[j:].data_ptr(), tab.data_ptr(), tgt.data_ptr(), 1, tgt.numel(),
                                                            14.0, 0.5, 1.0, 2.1, ctr.data_ptr(), tab.shape[0] - 1, st), "adaptive_wing_loss")
            _lib.check(L.mgf_select_best(self.min_loss[j:].data_ptr(), self.best_latent[j:].data_ptr(), self.best_step[j:].data_ptr(),
                                         self.losses.reshape(B, -1)[j].data_ptr(), self.latent_n[j:].data_ptr(), self.numel,
                                         _lib.ptr(self.p_loss[j:] if has_p else None), _lib.ptr(self.w_loss[j:] if self.use_wing else None),
                                         _lib.ptr(self.mse_loss[j:] if self.use_mse else None), float(a.lamda), float(a.beta),
                                         ctr.data_ptr(), _lib.ptr(valid), 1, self.steps, None, None, 0, None, None, st), "select_best")
        return img

    def run(self, steps=None):
        done = int(self.step_ctr[0].item()) if steps is None else None
        n = (self.steps - done) if steps is None else steps
        if self.use_graph and self.graph is None:
            self._capture()
        for _ in range(n):
            if self.graph is not None:
                self.graph.replay()
            else:
                self._iteration()
        return self

    def result(self):
        """One target: like ProjectionEngine.result().  B targets: (best_latents [B,k,D] cpu, best_steps [B], best_losses [B],
        losses [B,steps]) as numpy / tensors; a target that never improved on min_loss_init raises like the reference."""
        if self.targets == 1:
            return super().result()
        torch.cuda.synchronize(self.device)
        bs = self.best_step.cpu().numpy()
        if (bs < 0).any():
            raise IndexError(f"projection: targets {np.nonzero(bs < 0)[0].tolist()} never improved on the initial min_loss")
        return self.best_latent.cpu().clone(), bs, self.min_loss.cpu().numpy(), self.losses.cpu().numpy()


def save_best_png(G, latent, path, ratio=1.0, noise_mode="const"):
    """Write the image of `latent` as the drivers do (misc.to_pil + crop_max_rectangle, misc.py:94-130; :194-195)."""
    from . import misc
    latent = latent.to(G.device)
    if latent.ndim == 4:                       # a W+ result [1, k, num_ws, D] (GradientProjectionEngine(latent_space="w+"))
        img = G.forward_workspace(ws=latent, noise_mode=noise_mode)[0]
    else:
        img = G.forward_workspace(latent, None, noise_mode=noise_mode)[0]
    im = misc.crop_max_rectangle(misc.to_pil(img), ratio)
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    im.save(path)
    return path


def synthetic_landmarks(steps, res, seed):
    """Injected stand-in for the dlib detector: fixed [68,2] grid in the central half + seeded jitter."""
    rng = np.random.Generator(np.random.PCG64(seed))
    target = rng.integers(res // 4, 3 * res // 4, size=(68, 2)).astype(np.float64)
    per_step = target[None] + rng.integers(-max(res // 64, 1), max(res // 64, 1) + 1, size=(steps, 68, 2)).astype(np.float64)
    return target, per_step
