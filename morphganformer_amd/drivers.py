"""The callers on either side of the projection loop, re-stated as functions:

  image_transform   target image -> [1,3,S,S] in [-1,1]      (1024_example_wing_loss_perceptual_sqz_MSE.py:89-108)
  save_latent_mat / load_latent_mat    the `.mat` latent exchange format, key 'w'   (1024_merge_morph_2.py:73-92)
  generate_images   z ~ N(0,1) -> G(z, psi) -> PNG           (1024_generate.py:19-41)
  merge_morph       (1-a) w1 + a w2 -> G(., psi) -> JPG+.mat (1024_merge_morph_2.py:83-92; the reference hard-codes a = 0.5,
                    BASELINE config 4 sweeps 11 values)
  project_image     latent statistics + one ProjectionEngine run + best-of PNG / .mat   (:135-208, :246-268)
  second_stage      a projection initialised from an earlier result (edit_MSE.py pattern, BASELINE config 5)
  warp_morph        landmark-Delaunay warp of a morph onto the averaged landmarks (1024_warp_morphs.py:78-113,163-210)

Every image is produced by the HIP generator (`engine.Generator`); there is no CPU path here.  Landmark detection (dlib) is a
closed third-party CPU dependency: landmarks are passed in by the caller (projection.synthetic_landmarks stands in offline).
"""
from __future__ import annotations

import os

import numpy as np
import torch

from . import _lib
from .projection import GradientProjectionEngine, ProjectionArgs, ProjectionEngine, latent_stats, latent_stats_w


# ----------------------------------------------------------------------------------------------------------------- image I/O
def image_transform(src, size=1024, device="cuda"):
    """Resize(size) (shorter side, bilinear, PIL semantics) -> CenterCrop(size) -> ToTensor -> Normalize(0.5, 0.5).
    `src`: path or PIL image.  Returns float32 [1,3,size,size] on `device`."""
    from PIL import Image
    im = Image.open(src) if not isinstance(src, Image.Image) else src
    im = im.convert("RGB")
    w, h = im.size
    if (w <= h and w != size) or (h < w and h != size):
        if w <= h:
            nw, nh = size, int(size * h / w)
        else:
            nw, nh = int(size * w / h), size
        im = im.resize((nw, nh), Image.BILINEAR)
        w, h = im.size
    if w < size or h < size:            # CenterCrop pads small images with zeros
        pl, pt = max((size - w) // 2, 0), max((size - h) // 2, 0)
        canvas = Image.new("RGB", (max(w, size), max(h, size)))
        canvas.paste(im, (pl, pt))
        im, (w, h) = canvas, canvas.size
    top, left = int(round((h - size) / 2.0)), int(round((w - size) / 2.0))
    im = im.crop((left, top, left + size, top + size))
    x = torch.from_numpy(np.asarray(im, dtype=np.uint8).copy()).permute(2, 0, 1).float().div(255)
    return ((x - 0.5) / 0.5).unsqueeze(0).to(device)


def reference_gray_u8(img_hwc):
    """The gray uint8 image the drivers hand to dlib (:161-163): cv2.normalize(img, None, 0, 255, NORM_MINMAX, CV_8U) over
    the whole float image, then cv2.cvtColor(..., COLOR_BGR2GRAY) applied to RGB-ordered data (so channel 0 gets the blue
    weight) -- restated in numpy with OpenCV's 8-bit fixed-point coefficients (B 1868, G 9617, R 4899, >> 14)."""
    x = np.asarray(img_hwc, dtype=np.float32)
    lo, hi = float(x.min()), float(x.max())
    scale = 255.0 / (hi - lo) if hi > lo else 0.0
    u8 = np.clip(np.rint((x.astype(np.float64) - lo) * scale), 0, 255).astype(np.uint8)
    c = u8.astype(np.uint32)
    return ((c[..., 0] * 1868 + c[..., 1] * 9617 + c[..., 2] * 4899 + (1 << 13)) >> 14).astype(np.uint8)


def to_uint8_image(G, img):
    """[1,C,H,W] float32 device image in [-1,1] -> uint8 HWC numpy (misc.to_pil's rint+clip, misc.py:114-130) on the device."""
    c, h, w = img.shape[1:]
    out = torch.empty([h, w, c], dtype=torch.uint8, device=img.device)
    _lib.check(_lib.lib().mgf_to_uint8_hwc(out.data_ptr(), img.contiguous().data_ptr(), c, h, w, _lib.stream_ptr()), "to_uint8")
    return out.cpu().numpy()


def _crop_max_rectangle(im, ratio):
    if ratio is None:
        return im
    w, h = im.size
    s = min(w, h / ratio)
    cw, ch = s, ratio * s
    return im.crop((int((w - cw) // 2), int((h - ch) // 2), int((w + cw) // 2), int((h + ch) // 2)))


def save_image(G, img, path, ratio=1.0):
    from PIL import Image
    im = _crop_max_rectangle(Image.fromarray(to_uint8_image(G, img), "RGB"), ratio)
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    im.save(path)
    return path


def save_latent_mat(path, w):
    """`{'w': float32 [1,k,D]}` in MATLAB v5 format, like sio.savemat in the drivers (:201-206)."""
    import scipy.io as sio
    w = np.asarray(w.detach().cpu() if isinstance(w, torch.Tensor) else w, dtype=np.float32)
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    sio.savemat(path, {"w": w})
    return path


def load_latent_mat(path):
    import scipy.io as sio
    w = np.asarray(sio.loadmat(path)["w"], dtype=np.float32)
    if w.ndim != 3:
        raise ValueError(f"{path}: 'w' has shape {w.shape}, expected [1,k,D]")
    return w


# ----------------------------------------------------------------------------------------------------------------- drivers
def generate_images(G, images_num=32, truncation_psi=0.7, output_dir=None, ratio=1.0, seed=None, noise_mode="random"):
    """1024_generate.py:31-41.  Returns the list of z tensors used (and writes sample_%06d.png when `output_dir` is given)."""
    gen = None
    if seed is not None:
        gen = torch.Generator(device=G.device)
        gen.manual_seed(seed)
    zs = []
    for i in range(images_num):
        z = torch.randn([1, G.cfg.k, G.cfg.z_dim], device=G.device, generator=gen)
        img = G(z, truncation_psi=truncation_psi, noise_mode=noise_mode)[0]
        if output_dir is not None:
            save_image(G, img, os.path.join(output_dir, f"sample_{i:06d}.png"), ratio)
        zs.append(z.cpu())
    return zs


def merge_morph(G, w1, w2, alphas=(0.5,), truncation_psi=0.7, noise_mode="random", out_prefix=None, ratio=1.0):
    """Linear latent morphs `dw = (1-a) w1 + a w2` rendered with G(dw, psi) (1024_merge_morph_2.py:83-92).
    w1/w2: numpy or tensors [1,k,D] (what the `.mat` files hold).  Returns (latents [A,1,k,D] numpy, images [A,3,H,W] device).
    The blend is done in numpy float32 exactly like the reference (`0.5 * w1 + 0.5 * w2` on loadmat arrays)."""
    a1 = np.asarray(w1.detach().cpu() if isinstance(w1, torch.Tensor) else w1, dtype=np.float32)
    a2 = np.asarray(w2.detach().cpu() if isinstance(w2, torch.Tensor) else w2, dtype=np.float32)
    lat, imgs = [], []
    for a in alphas:
        if a == 0.5:
            dw = 0.5 * a1 + 0.5 * a2
        else:
            dw = np.float32(1.0 - a) * a1 + np.float32(a) * a2
        img = G(torch.from_numpy(dw).to(G.device), truncation_psi, noise_mode=noise_mode)[0]
        if out_prefix is not None:
            tag = f"{out_prefix}_a{a:.2f}"
            save_image(G, img, tag + ".jpg", ratio)
            save_latent_mat(tag + ".mat", dw)
        lat.append(dw)
        imgs.append(img[0].clone())
    return np.stack(lat), torch.stack(imgs)


DEFAULT_BATCH = 32       # loop steps per generator forward in literal mode: the configuration bench.py times (1.6 GB of activations per step at 1024^2;
                         # measured 20 .. 64: 32 is the fastest, 25 -- the round-2 figure -- 1.5 - 3 % behind)


def project_image(G, target, lm_target, lm_steps, args: ProjectionArgs = None, percept=None, latent_mean=None, latent_std=None,
                  eps=None, out_prefix=None, batch=DEFAULT_BATCH, use_graph=True, noise_mode="random", use_mse=True, seed=None,
                  landmark_fn=None, mode="literal", weight_decay=0.0, path_to_gen=None, keep_images=64, engine=None,
                  return_engine=False, latent_space="z"):
    """One full `projection(...)` call (:135-208).  `target`: [1,3,S,S] from image_transform; `lm_target` [68,2] and either
    `lm_steps` [steps,68,2] (injected landmark detections) or `landmark_fn` (host detector called on every generated image,
    see ProjectionEngine).  mode="literal" is the loop as the reference executes it (best-of-N noisy sampling, `batch` steps per
    forward -- 32 by default, the benchmarked configuration; the result does not depend on it); mode="gradient" back-propagates the
    loss into the latent and lets Adam move it (GradientProjectionEngine; one candidate per step; weight_decay=1e-4 is the
    1024_example_MSE.py:117 optimizer; latent_space="w+" optimises the per-layer intermediate latent [k, num_ws, D] instead of z -- the
    statistics are then taken in w space, projection.latent_stats_w -- and `w` comes back as [1, k, num_ws, D]).  Returns dict(w, step,
    loss, losses).

    Outputs, like the drivers: with `path_to_gen` the SCORED image of every improvement -- the candidate as it was generated and
    ranked, its random per-layer noise included -- is written as `{path_to_gen}/{step:06d}_{loss:04f}.png` (:190-195; literal mode:
    the images stay on the device during the run, `keep_images` slots that are spilled to the host between launch sequences, and are
    written afterwards; gradient mode: the best latent's rendering under that name).  `out_prefix` adds the latent as
    `{out_prefix}.mat` (key 'w', :201-206 of the morph drivers) and, when no `path_to_gen` trail is written, the best latent's
    rendering as `{out_prefix}.png`.

    engine: a ProjectionEngine from an earlier call with the same generator, objective, step count and batch (return_engine=True
    hands it out) -- it is re-targeted in place (`ProjectionEngine.retarget`), which keeps its captured hipGraph and workspaces; this is
    how `project_many` walks a list of targets."""
    args = args or ProjectionArgs()
    if mode not in ("literal", "gradient"):
        raise ValueError(f"mode must be 'literal' or 'gradient' (got {mode!r})")
    if engine is not None and mode != "literal":
        raise ValueError("engine= (re-targeting) is for literal mode")
    if latent_space not in ("z", "w+") or (latent_space == "w+" and mode != "gradient"):
        raise ValueError("latent_space must be 'z', or 'w+' together with mode='gradient'")
    if latent_mean is None or latent_std is None:
        gen = None
        if seed is not None:
            gen = torch.Generator(device=G.device)
            gen.manual_seed(seed)
        stats = latent_stats_w if latent_space == "w+" else latent_stats
        latent_mean, latent_std = stats(G, args.n_mean_latent, G.device, generator=gen)
    keep = max(int(keep_images), int(batch)) if path_to_gen is not None and mode == "literal" else 0
    if engine is not None:
        if engine.G is not G or engine.batch != batch or engine.steps != args.step or engine.keep_images != keep:
            raise ValueError("engine= was built for another generator / batch / step count / trail size")
        # the objective is baked into the captured launch sequence: a re-targeted engine must score exactly what a fresh one would
        diff = [name for name, ok in (("landmarks (Wing term)", engine.use_wing == (lm_target is not None)),
                                      ("percept", engine.percept is percept), ("use_mse", engine.use_mse == bool(use_mse)),
                                      ("noise_mode", engine.noise_mode == noise_mode), ("args", engine.args == args),
                                      ("landmark_fn", engine.landmark_fn is landmark_fn)) if not ok]
        if diff:
            raise ValueError("engine= was built for another objective: " + ", ".join(diff) + " differ(s); build a fresh engine")
        eng = engine.retarget(target, lm_target=lm_target, lm_steps=lm_steps, eps=eps, seed=seed if eps is None else None,
                              latent_mean=latent_mean, latent_std=float(latent_std))
    elif mode == "gradient":
        eng = GradientProjectionEngine(G, target, latent_mean, float(latent_std), args, weight_decay=weight_decay, percept=percept,
                                       lm_target=lm_target, lm_steps=lm_steps, eps=eps, noise_mode=noise_mode, use_graph=use_graph,
                                       use_mse=use_mse, landmark_fn=landmark_fn, seed=0 if seed is None else seed, latent_space=latent_space)
    else:
        eng = ProjectionEngine(G, target, latent_mean, float(latent_std), args, percept=percept, lm_target=lm_target,
                               lm_steps=lm_steps, eps=eps, noise_mode=noise_mode, use_graph=use_graph, batch=batch, use_mse=use_mse,
                               landmark_fn=landmark_fn, keep_images=keep, seed=0 if seed is None else seed)
    w, step, loss, losses = eng.run().result()
    out = {"w": w, "step": step, "loss": loss, "losses": losses}
    if out_prefix is not None:
        save_latent_mat(f"{out_prefix}.mat", w)
        if path_to_gen is None:                      # no improvement trail asked for: still leave an image of the result beside the latent
            from .projection import save_best_png
            out["image"] = save_best_png(G, w, f"{out_prefix}.png", args.ratio)
    if path_to_gen is not None:
        if mode == "literal":
            out["images"] = eng.save_improvements(path_to_gen, args.ratio)
        else:
            from .projection import save_best_png
            out["images"] = [save_best_png(G, w, os.path.join(path_to_gen, "{:06d}_{:04f}.png".format(step, loss)), args.ratio)]
    if return_engine:
        out["engine"] = eng
    return out


def project_many(G, targets, landmarks=None, dynamic=False, lockstep=1, **kw):
    """Pair-level sharding of BASELINE configs 3/5: rank r projects `targets[r::world]` (or, with dynamic=True, whatever the
    shared `distributed.WorkQueue` hands it), then ONE all_gather returns every item's {latent, loss, step} to every rank.
    targets: list of [1,3,S,S] device tensors (or image paths); landmarks: optional list of (lm_target, lm_steps) per item.
    lockstep > 1 (gradient mode, static sharding): a rank advances that many of its items through one generator
    forward/backward per step (GradientProjectionEngine with B targets) instead of one after the other.
    In literal mode the rank builds ONE engine (latent statistics, LPIPS workspaces, hipGraph capture) for its first item and
    re-targets it for the others, as the reference keeps G / percept / latent statistics outside its per-image loop
    (projection_example_v2_percept_morph.py:311-355).
    Returns dict(latents [N,k,D], losses [N], steps [N], items [N]) ordered by item id."""
    import torch.distributed as dist
    from .distributed import gather_many, pack_result, run_sharded, shard_items, unpack_results
    on = dist.is_available() and dist.is_initialized()
    rank, world = (dist.get_rank(), dist.get_world_size()) if on else (0, 1)
    load = lambda t: t if isinstance(t, torch.Tensor) else image_transform(t, size=G.img_resolution, device=G.device)
    w_plus = kw.get("latent_space", "z") == "w+"
    lshape = (G.cfg.k, G.cfg.num_ws, G.cfg.w_dim) if w_plus else (G.cfg.k, G.cfg.z_dim)      # a W+ result is [k, num_ws, D] per item
    width = int(np.prod(lshape)) + 3
    if w_plus and kw.get("mode") != "gradient":
        raise ValueError("latent_space='w+' needs mode='gradient'")
    if lockstep > 1:
        if dynamic or kw.get("mode") != "gradient":
            raise ValueError("lockstep groups need mode='gradient' and static sharding")
        recs = []
        mine = shard_items(len(targets), rank, world)
        for g0 in range(0, len(mine), lockstep):
            ids = mine[g0:g0 + lockstep]
            res = _project_group(G, [load(targets[i]) for i in ids], [landmarks[i] for i in ids] if landmarks is not None else None, **kw)
            recs += [pack_result(res["w"][j:j + 1].to(G.device), float(res["loss"][j]), int(res["step"][j]), item=i) for j, i in enumerate(ids)]
        rows = torch.stack(recs) if recs else torch.empty([0, width], dtype=torch.float64, device=G.device)
        return unpack_results(gather_many(rows, -(-len(targets) // world)), lshape)
    reuse = kw.get("mode", "literal") == "literal" and kw.get("landmark_fn") is None and kw.get("eps") is None
    if reuse and (kw.get("latent_mean") is None or kw.get("latent_std") is None):
        a = kw.get("args") or ProjectionArgs()
        gen = None
        if kw.get("seed") is not None:
            gen = torch.Generator(device=G.device)
            gen.manual_seed(kw["seed"])
        kw["latent_mean"], kw["latent_std"] = latent_stats(G, a.n_mean_latent, G.device, generator=gen)     # once per rank, not per item
    state = {"eng": None}

    def work(i):
        lm_t, lm_s = landmarks[i] if landmarks is not None else (None, None)
        eng = state["eng"]
        if eng is not None and eng.use_wing != (lm_t is not None):
            eng = None                     # an item with / without landmarks after one without / with: another objective, a fresh engine
        r = project_image(G, load(targets[i]), lm_t, lm_s, engine=eng, return_engine=reuse, **kw)
        state["eng"] = r.get("engine")
        return pack_result(r["w"].to(G.device), r["loss"], r["step"], item=i)

    rows, _mine = run_sharded(len(targets), work, width, G.device, dynamic=dynamic)
    return unpack_results(rows, lshape)


def _project_group(G, targets, landmarks, args: ProjectionArgs = None, percept=None, latent_mean=None, latent_std=None, eps=None,
                   use_graph=True, noise_mode="random", use_mse=True, seed=None, weight_decay=0.0, mode="gradient", latent_space="z",
                   **unused):
    """B targets through one lockstep GradientProjectionEngine; returns dict(w [B,k,D] (W+: [B,k,num_ws,D]), step [B], loss [B],
    losses [B,steps])."""
    args = args or ProjectionArgs()
    if unused:
        raise TypeError(f"project_many(lockstep=...): unsupported arguments {sorted(unused)}")
    if latent_mean is None or latent_std is None:
        gen = None
        if seed is not None:
            gen = torch.Generator(device=G.device)
            gen.manual_seed(seed)
        latent_mean, latent_std = (latent_stats_w if latent_space == "w+" else latent_stats)(G, args.n_mean_latent, G.device, generator=gen)
    tg = torch.cat([t.reshape(1, *t.shape[-3:]) for t in targets]).contiguous()
    lm_t = lm_s = None
    if landmarks is not None:
        lm_t, lm_s = np.stack([np.asarray(l[0]) for l in landmarks]), np.stack([np.asarray(l[1]) for l in landmarks])
    if len(targets) == 1 and lm_t is not None:
        lm_t, lm_s = lm_t[0], lm_s[0]
    eng = GradientProjectionEngine(G, tg, latent_mean, float(latent_std), args, weight_decay=weight_decay, percept=percept,
                                   lm_target=lm_t, lm_steps=lm_s, eps=eps, noise_mode=noise_mode, use_graph=use_graph, use_mse=use_mse,
                                   seed=0 if seed is None else seed, latent_space=latent_space)
    w, step, loss, losses = eng.run().result()
    if len(targets) == 1:
        return {"w": w, "step": np.array([step]), "loss": np.array([loss]), "losses": losses[None]}
    return {"w": w, "step": step, "loss": loss, "losses": losses}


# the 12 frame points the reference adds to the 68 landmarks before triangulating (1024_warp_morphs.py:130-132)
WARP_EXTRA_POINTS = [[0, 0], [0, 341], [0, 682], [0, 1023], [341, 0], [682, 0], [1023, 0], [1023, 341], [1023, 682], [1023, 1023],
                     [341, 1023], [682, 1023]]


def warp_mesh(points_src, points_dst):
    """Host side of the Delaunay warp (1024_warp_morphs.py:163-201): triangulate the DESTINATION points with scipy (as the reference
    does), and per triangle return the integer polygon np.int32(dst triangle) that cv2.fillConvexPoly would fill and the affine map
    destination -> source (the inverse of cv2.getAffineTransform(srcTri, dstTri)), solved in float64.
    Returns (tri_xy int32 [T,6], dst_to_src float32 [T,6], simplices [T,3])."""
    from scipy.spatial import Delaunay
    ps, pd = np.asarray(points_src, np.float64), np.asarray(points_dst, np.float64)
    simplices = Delaunay(pd).simplices
    tri_xy = np.zeros((len(simplices), 6), np.int32)
    maps = np.zeros((len(simplices), 6), np.float64)
    for t, idx in enumerate(simplices):
        d, s_ = pd[idx], ps[idx]
        tri_xy[t] = np.int32(d).reshape(-1)                               # truncation, like np.int32(tRect) (:101)
        a = np.concatenate([d, np.ones((3, 1))], axis=1)                  # [x y 1] @ M^T = src
        maps[t] = np.linalg.solve(a, s_).T.reshape(-1)
    return tri_xy, maps.astype(np.float32), simplices


def warp_morph(img, points_G, points_avg, background=0.0):
    """Warp the generated morph so that its landmarks `points_G` land on the averaged landmarks `points_avg` (both [P,2] in pixel
    coordinates, frame points included), the post-process of 1024_warp_morphs.py:163-210 as one gather kernel.
    img: [1,C,H,W] or [C,H,W] float32 device tensor in ANY value range (the reference works on the 0..255 float BGR image it reads
    back from the PNG); returns a tensor of the same shape."""
    _lib.require_gpu(img)
    x = img.reshape(-1, *img.shape[-2:]).contiguous().float()
    c, h, w = x.shape
    tri_xy, maps, _ = warp_mesh(points_G, points_avg)
    dev = x.device
    t_d = torch.as_tensor(tri_xy, device=dev).contiguous()
    m_d = torch.as_tensor(maps, device=dev).contiguous()
    out = torch.empty_like(x)
    _lib.check(_lib.lib().mgf_piecewise_affine_warp_f32(out.data_ptr(), x.data_ptr(), t_d.data_ptr(), m_d.data_ptr(), len(tri_xy), c, h, w,
                                                        float(background), _lib.stream_ptr()), "piecewise_affine_warp")
    return out.reshape(img.shape)


def second_stage(G, target, w_init, latent_std, lm_target, lm_steps, **kw):
    """A second projection whose noisy candidates are drawn around an earlier result instead of the latent mean
    (edit_MSE.py: `latent_in = w1` pattern; BASELINE config 5)."""
    w0 = torch.as_tensor(np.asarray(w_init.detach().cpu() if isinstance(w_init, torch.Tensor) else w_init, dtype=np.float32))
    return project_image(G, target, lm_target, lm_steps, latent_mean=w0.reshape(w0.shape[-2:]).to(G.device),
                         latent_std=latent_std, **kw)
