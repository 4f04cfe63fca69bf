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

from . import _lib, misc
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
    return misc.to_uint8(img)


_crop_max_rectangle = misc.crop_max_rectangle        # (the reference's name lives in morphganformer_amd.misc)


def save_image(G, img, path, ratio=1.0):
    """`crop(misc.to_pil(img[0]), ratio).save(path)` of the drivers (...sqz_MSE.py:190-195)."""
    im = misc.crop_max_rectangle(misc.to_pil(img), ratio)
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
    if w.ndim not in (3, 4):
        raise ValueError(f"{path}: 'w' has shape {w.shape}, expected [1,k,D] (or a W+ latent [1,k,num_ws,D])")
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


def render_latent(G, w, truncation_psi=0.7, noise_mode="random"):
    """The image of a projected latent as the morph drivers render it: `G(w, truncation_psi)` for a z latent [n,k,D] (the 0.7 lands in `c`,
    SURVEY 0.2) and `G(ws=w)` for a W+ latent [n,k,num_ws,D] (gradient mode, latent_space="w+": every layer reads its own slot;
    truncation acts in the mapping network only, networks.py:935-941, so there is none to apply).  -> [n,3,R,R], the caller's own tensor."""
    w = (w if isinstance(w, torch.Tensor) else torch.from_numpy(np.asarray(w, dtype=np.float32))).to(G.device)
    if w.ndim == 4:
        return G(ws=w, noise_mode=noise_mode)[0]
    return G(w, truncation_psi, noise_mode=noise_mode)[0]


def merge_morph(G, w1, w2, alphas=(0.5,), truncation_psi=0.7, noise_mode="random", out_prefix=None, ratio=1.0, batched=False):
    """Linear latent morphs `dw = (1-a) w1 + a w2` rendered with G(dw, psi) (1024_merge_morph_2.py:83-92).
    w1/w2: numpy or tensors [1,k,D] (what the `.mat` files hold), or two W+ results [1,k,num_ws,D] -- blended slot by slot and rendered
    through `G(ws=...)` (render_latent).  Returns (latents [A,1,k,D] numpy, images [A,3,H,W] device).
    The blend is done in numpy float32 exactly like the reference (`0.5 * w1 + 0.5 * w2` on loadmat arrays).
    batched=True renders the whole sweep with ONE generator forward of len(alphas) images (BASELINE config 4's 11-alpha sweep as one
    batch-11 forward) instead of one forward per alpha like the script; the images agree to float32 rounding (other kernel shapes)."""
    a1 = np.asarray(w1.detach().cpu() if isinstance(w1, torch.Tensor) else w1, dtype=np.float32)
    a2 = np.asarray(w2.detach().cpu() if isinstance(w2, torch.Tensor) else w2, dtype=np.float32)
    if a1.shape != a2.shape:
        raise ValueError(f"merge_morph: the two latents differ in shape: {a1.shape} vs {a2.shape}")
    blend = lambda a: (0.5 * a1 + 0.5 * a2) if a == 0.5 else (np.float32(1.0 - a) * a1 + np.float32(a) * a2)
    lat, imgs = [], []
    sweep = render_latent(G, np.concatenate([blend(a) for a in alphas]), truncation_psi, noise_mode) if batched and len(alphas) > 1 else None
    for j, a in enumerate(alphas):
        dw = blend(a)
        img = sweep[j:j + 1] if sweep is not None else render_latent(G, dw, truncation_psi, noise_mode)
        if out_prefix is not None:
            tag = f"{out_prefix}_a{a:.2f}"
            save_image(G, img, tag + ".jpg", ratio)
            save_latent_mat(tag + ".mat", dw)
        lat.append(dw)
        imgs.append(img[0].clone())
    return np.stack(lat), torch.stack(imgs)


def merge_morph_tree(G, src_path, dst_path, truncation_psi=0.7, ratio=1.0, noise_mode="random"):
    """The directory walk of 1024_merge_morph_2.py:53-92: `src_path/<id>/<name>/` holds the projected latents (`.mat`, beside their `.png`s) of
    one bona fide image; every id folder has two such name folders (entries with a dot are skipped, :61-63; the first two are used, :64-65), and
    every latent of the first is morphed with every latent of the second -- `0.5 * w1 + 0.5 * w2`, rendered with `G(W, truncation_psi)` -- into
    `dst_path/<id>/<stem1>+<stem2>.jpg / .mat` (:77-92); pairs whose image already exists are skipped (:80).  Returns the list of written stems."""
    done = []
    for ident in sorted(os.listdir(src_path)):
        id_dir = os.path.join(src_path, ident)
        if not os.path.isdir(id_dir):
            continue
        names = [nm for nm in sorted(os.listdir(id_dir)) if len(nm.split(".")) == 1]
        if len(names) < 2:
            raise ValueError(f"{id_dir}: needs the latent folders of two bona fide images, found {names}")
        dst_id = os.path.join(dst_path, ident)
        os.makedirs(dst_id, exist_ok=True)
        mats = [[f for f in sorted(os.listdir(os.path.join(id_dir, nm))) if f.endswith(".mat")] for nm in names[:2]]
        for im1 in mats[0]:
            w1 = load_latent_mat(os.path.join(id_dir, names[0], im1))
            for im2 in mats[1]:
                stem = im1[:-4] + "+" + im2[:-4]
                if os.path.exists(os.path.join(dst_id, stem + ".jpg")):
                    continue
                w2 = load_latent_mat(os.path.join(id_dir, names[1], im2))
                lat, imgs = merge_morph(G, w1, w2, (0.5,), truncation_psi, noise_mode=noise_mode)
                save_image(G, imgs[0:1], os.path.join(dst_id, stem + ".jpg"), ratio)
                save_latent_mat(os.path.join(dst_id, stem + ".mat"), lat[0])
                done.append(os.path.join(ident, stem))
    return done


def merge_files(src_path, dst_path):
    """1024_merge_files.py:20-45, the step in front of the morph walk above ("combine all bonafides from different results"): a results tree
    `src_path/<version>/<variant>/<id>/<name>/<file>` is folded over its variants into `dst_path/<version>/<id>/<name>/<file>` -- the per-image
    folders of every variant of a version land side by side under one id; entries of an id folder that carry a dot are skipped (:36), files
    are copied with `shutil.copy` (a later variant's file of the same name overwrites an earlier one, as in the script, which walks
    `os.listdir` order; here sorted, so the result does not depend on the file system).  Host-side file bookkeeping: no tensor is touched.
    Returns the list of destination files."""
    import shutil
    out = []
    for version in sorted(os.listdir(src_path)):
        v_dir = os.path.join(src_path, version)
        if not os.path.isdir(v_dir):
            continue
        os.makedirs(os.path.join(dst_path, version), exist_ok=True)          # (:23-25: made even when the version holds no variant)
        for variant in sorted(os.listdir(v_dir)):
            for ident in sorted(os.listdir(os.path.join(v_dir, variant))):
                id_dir = os.path.join(v_dir, variant, ident)
                for name in sorted(os.listdir(id_dir)):
                    if len(name.split(".")) > 1:
                        continue
                    dst_fold = os.path.join(dst_path, version, ident, name)
                    os.makedirs(dst_fold, exist_ok=True)
                    for img in sorted(os.listdir(os.path.join(id_dir, name))):
                        out.append(shutil.copy(os.path.join(id_dir, name, img), os.path.join(dst_fold, img)))
    return out


DEFAULT_BATCH = 32       # loop steps per generator forward in literal mode: the configuration bench.py times (1.6 GB of activations per step at 1024^2;
                         # measured 20 .. 64: 32 is the fastest, 25 -- the round-2 figure -- 1.5 - 3 % behind)


def project_image(G, target, lm_target, lm_steps, args: ProjectionArgs = None, percept=None, latent_mean=None, latent_std=None,
                  eps=None, out_prefix=None, batch=DEFAULT_BATCH, use_graph=True, noise_mode="random", use_mse=True, seed=None,
                  landmark_fn=None, mode="literal", weight_decay=0.0, path_to_gen=None, keep_images=64, engine=None,
                  return_engine=False, latent_space="z", landmark_input="float", biometric=None, gamma=1.0, lbp_target=None, pipeline=None):
    """One full `projection(...)` call (:135-208).  `target`: [1,3,S,S] from image_transform; `lm_target` [68,2] and either
    `lm_steps` [steps,68,2] (injected landmark detections) or `landmark_fn` (host detector called on every generated image,
    see ProjectionEngine; landmark_input="gray_u8" hands it the drivers' gray uint8 image, built on the device).  mode="literal" is the loop as the reference executes it (best-of-N noisy sampling, `batch` steps per
    forward -- 32 by default, the benchmarked configuration; the result does not depend on it); mode="gradient" back-propagates the
    loss into the latent and lets Adam move it (GradientProjectionEngine; one candidate per step; weight_decay=1e-4 is the
    1024_example_MSE.py:117 optimizer; latent_space="w+" optimises the per-layer intermediate latent [k, num_ws, D] instead of z -- the
    statistics are then taken in w space, projection.latent_stats_w -- and `w` comes back as [1, k, num_ws, D]).  Returns dict(w, step,
    loss, losses).

    biometric / gamma: an iresnet.BiometricLoss (embedder "facenet" or "iresnetNN") adds gamma * MSE(embed(img), embed(target)) to the objective
    (BASELINE config 3; 1024_example_FaceNet_percept.py:147-158).  lbp_target: with args.pixel_term="lbp" the target FILE's LBP code map
    (lbp.target_feature; 1024_example_LBP_percept.py:140).

    Outputs, like the drivers: with `path_to_gen` the SCORED image of every improvement -- the candidate as it was generated and
    ranked, its random per-layer noise included -- is written as `{path_to_gen}/{step:06d}_{loss:04f}.png` (:190-195; literal mode:
    the images stay on the device during the run, `keep_images` slots that are spilled to the host between launch sequences, and are
    written afterwards; gradient mode: the best latent's rendering under that name).  `out_prefix` adds the latent as
    `{out_prefix}.mat` (key 'w', :201-206 of the morph drivers) and, when no `path_to_gen` trail is written, the best latent's
    rendering as `{out_prefix}.png`.

    pipeline: literal mode without a host landmark callback -- the losses and the selection of one batch of candidates run on a side stream while the
    generator already synthesises the next batch (ProjectionEngine(pipeline=True): same result, bit for bit; one more image batch of memory).  None
    (default) = where it was measured to pay: LPIPS(squeeze) without an embedder (+1.9 % on a 1000-step projection at 1024^2 and 32 candidates per
    forward; LPIPS(vgg) and the FaceNet term, whose own matrix work then competes with the generator's, lose 1 - 1.5 %, and with no perceptual term
    there is nothing to overlap, -0.5 %: those stay on one stream).

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
    if pipeline is None:
        pipeline = mode == "literal" and landmark_fn is None and biometric is None and getattr(percept, "net", None) == "squeeze"
    keep = max(int(keep_images), int(batch)) if path_to_gen is not None and mode == "literal" else 0
    if engine is not None:
        if (engine.G is not G or engine.batch != batch or engine.steps != args.step or engine.keep_images != keep or
                engine.pipeline != (bool(pipeline) and landmark_fn is None)):
            raise ValueError("engine= was built for another generator / batch / step count / trail size / pipeline mode")
        # the objective is baked into the captured launch sequence: a re-targeted engine must score exactly what a fresh one would
        diff = [name for name, ok in (("landmarks (Wing term)", engine.use_wing == (lm_target is not None)),
                                      ("percept", engine.percept is percept), ("use_mse", engine.use_mse == bool(use_mse)),
                                      ("noise_mode", engine.noise_mode == noise_mode), ("args", engine.args == args),
                                      ("landmark_fn", engine.landmark_fn is landmark_fn), ("biometric", engine.biometric is biometric),
                                      ("gamma", biometric is None or engine.gamma == float(gamma))) if not ok]
        if diff:
            raise ValueError("engine= was built for another objective: " + ", ".join(diff) + " differ(s); build a fresh engine")
        eng = engine.retarget(target, lm_target=lm_target, lm_steps=lm_steps, eps=eps, seed=seed if eps is None else None,
                              latent_mean=latent_mean, latent_std=float(latent_std), lbp_target=lbp_target)
    elif mode == "gradient":
        eng = GradientProjectionEngine(G, target, latent_mean, float(latent_std), args, weight_decay=weight_decay, percept=percept,
                                       lm_target=lm_target, lm_steps=lm_steps, eps=eps, noise_mode=noise_mode, use_graph=use_graph,
                                       use_mse=use_mse, landmark_fn=landmark_fn, seed=0 if seed is None else seed, latent_space=latent_space,
                                       biometric=biometric, gamma=gamma)
    else:
        eng = ProjectionEngine(G, target, latent_mean, float(latent_std), args, percept=percept, lm_target=lm_target,
                               lm_steps=lm_steps, eps=eps, noise_mode=noise_mode, use_graph=use_graph, batch=batch, use_mse=use_mse,
                               landmark_fn=landmark_fn, keep_images=keep, seed=0 if seed is None else seed, landmark_input=landmark_input,
                               biometric=biometric, gamma=gamma, lbp_target=lbp_target, pipeline=pipeline)
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
    Returns dict(latents [N,k,D], losses [N], steps [N], items [N]) ordered by item id, plus `mine`: the items this rank worked on."""
    import torch.distributed as dist
    from .distributed import gather_many, pack_result, run_sharded, shard_items, unpack_results
    on = dist.is_available() and dist.is_initialized()
    rank, world = (dist.get_rank(), dist.get_world_size()) if on else (0, 1)
    # an item is a device tensor, an image path, or a callable that produces the tensor when the item is taken (a rank only ever touches its own)
    pa = getattr(kw.get("args"), "pool_above", 0) or 0           # (projection_example_v1.py: the target at the size the losses see)
    tsize = G.img_resolution // (G.img_resolution // pa) if (pa and G.img_resolution > pa) else G.img_resolution
    load = lambda t: t if isinstance(t, torch.Tensor) else (t() if callable(t) else image_transform(t, size=tsize, device=G.device))
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
        out = unpack_results(gather_many(rows, -(-len(targets) // world)), lshape)
        out["mine"] = list(mine)
        return out
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

    rows, mine = run_sharded(len(targets), work, width, G.device, dynamic=dynamic)
    out = unpack_results(rows, lshape)
    out["mine"] = mine                                       # the items THIS rank projected, in the order it took them
    return out


def read_pair_csv(path, threshold=0.5):
    """The pair list of projection_example_v2_percept_morph.py:337-343: rows `img1, img2, similarity`; the header row (`img1`) and rows below the
    similarity threshold are skipped.  -> [(img1, img2), ...]"""
    import csv
    pairs = []
    with open(path, "r", newline="") as fh:
        for row in csv.reader(fh):
            if not row or row[0] == "img1":
                continue
            if float(row[2]) < threshold:
                continue
            pairs.append((row[0], row[1]))
    return pairs


def morph_pairs(G, pairs, src_dir, dst_raw, dst_morph, landmarks=None, truncation_psi=0.7, ratio=1.0, noise_mode="random", **project_kw):
    """BASELINE config 3's outer loop (projection_example_v2_percept_morph.py:330-365): for every pair of bona fide images, project both into the
    latent space, render them (`<a>_<b>_A.png`, `_B.png` under dst_raw) and their 0.5 / 0.5 latent morph (`<a>_<b>.png` under dst_morph); pairs
    whose morph exists are skipped (:352).  The 2 x len(pairs) projections are independent: `project_many` shards them over the ranks of the
    process group (static or, with dynamic=True, through the work queue) and gathers every latent to every rank; the renderings are then
    dealt `pairs[rank::world]`.  pairs: [(img1, img2)] file names under src_dir (read_pair_csv); landmarks: optional {file name: (lm_target,
    lm_steps)}; project_kw: project_image's arguments (args, percept, biometric, gamma, batch, seed, mode, dynamic, ...).
    Returns dict(pairs (the ones worked on), latents [2P,k,D] (W+: [2P,k,num_ws,D]), losses, steps, written (this rank's morph paths))."""
    import torch.distributed as dist
    from .distributed import shard_items
    stem = lambda f: f.split(".")[0]                          # (`img1.split('.')[0]`, :347)
    on = dist.is_available() and dist.is_initialized()
    rank, world = (dist.get_rank(), dist.get_world_size()) if on else (0, 1)
    # ONE rank looks at dst_morph and tells the others: the static shards and the size of the result gather are derived from this list, so
    # every rank must hold the same one (node-local scratch or a lagging network file system would otherwise let them disagree -- and an
    # empty list on some ranks only would let those skip the collective the others wait in)
    todo = [(a, b) for a, b in pairs if not os.path.exists(os.path.join(dst_morph, f"{stem(a)}_{stem(b)}.png"))] if rank == 0 else None
    if on and world > 1:
        box = [todo]
        dist.broadcast_object_list(box, src=0)
        todo = [tuple(pr) for pr in box[0]]
    if not todo:
        return {"pairs": [], "latents": None, "losses": None, "steps": None, "written": []}
    files = [f for pr in todo for f in pr]
    lms = None if landmarks is None else [landmarks[f] for f in files]
    res = project_many(G, [os.path.join(src_dir, f) for f in files], landmarks=lms, **project_kw)
    written = []
    for pi in shard_items(len(todo), rank, world):
        a, b = todo[pi]
        name = f"{stem(a)}_{stem(b)}"
        w1, w2 = res["latents"][2 * pi:2 * pi + 1], res["latents"][2 * pi + 1:2 * pi + 2]
        for w, tag in ((w1, "_A"), (w2, "_B")):                 # (a W+ result [1,k,num_ws,D] renders through G(ws=...): render_latent)
            save_image(G, render_latent(G, w, truncation_psi, noise_mode), os.path.join(dst_raw, name + tag + ".png"), ratio)
        _, imgs = merge_morph(G, w1, w2, (0.5,), truncation_psi, noise_mode=noise_mode)
        save_image(G, imgs[0:1], os.path.join(dst_morph, name + ".png"), ratio)
        written.append(os.path.join(dst_morph, name + ".png"))
    return {"pairs": todo, "latents": res["latents"], "losses": res["losses"], "steps": res["steps"], "written": written}


def _project_group(G, targets, landmarks, args: ProjectionArgs = None, percept=None, latent_mean=None, latent_std=None, eps=None,
                   use_graph=True, noise_mode="random", use_mse=True, seed=None, weight_decay=0.0, mode="gradient", latent_space="z",
                   biometric=None, gamma=1.0, batch=None, pipeline=None, **unused):
    """B targets through one lockstep GradientProjectionEngine; returns dict(w [B,k,D] (W+: [B,k,num_ws,D]), step [B], loss [B],
    losses [B,steps])."""
    args = args or ProjectionArgs()
    if unused or pipeline:                          # (`batch` is literal mode's steps per forward: gradient mode evaluates one candidate per step; `pipeline` is literal mode's too)
        unused = dict(unused, **({"pipeline": pipeline} if pipeline else {}))
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
                                   seed=0 if seed is None else seed, latent_space=latent_space, biometric=biometric, gamma=gamma)
    w, step, loss, losses = eng.run().result()
    if len(targets) == 1:
        return {"w": w, "step": np.array([step]), "loss": np.array([loss]), "losses": losses[None]}
    return {"w": w, "step": step, "loss": loss, "losses": losses}


# the 12 frame points the reference adds to the 68 landmarks before triangulating (1024_warp_morphs.py:130-132)
WARP_EXTRA_POINTS = [[0, 0], [0, 341], [0, 682], [0, 1023], [341, 0], [682, 0], [1023, 0], [1023, 341], [1023, 682], [1023, 1023],
                     [341, 1023], [682, 1023]]


# ---- host set-up of the warp: the OpenCV calls of morphTriangle that are integer / 6x6 control work, restated from OpenCV's published sources
# (4.x: imgproc/src/shapedescr.cpp pointSetBoundingRect, imgwarp.cpp getAffineTransform / warpAffine, core LUImpl, drawing.cpp fillConvexPoly /
# FillConvexPoly / Line / LineIterator) in closed form; the per-pixel work is the device kernel's (csrc/warp.hip)
_XY_SHIFT = 16


def _cv_bounding_rect(tri):
    """cv2.boundingRect(np.float32([tri])): floor of the float32 extremes, both ends included."""
    p = np.asarray(tri, np.float32)
    lo, hi = np.floor(p.min(0)).astype(np.int64), np.floor(p.max(0)).astype(np.int64)
    return int(lo[0]), int(lo[1]), int(hi[0] - lo[0] + 1), int(hi[1] - lo[1] + 1)


def _cv_affine_inverse(src_tri, dst_tri):
    """cv2.getAffineTransform(np.float32(src), np.float32(dst)) -- the 6x6 system solved by Gaussian elimination with partial pivoting in
    double (core LUImpl), whole rows at a time -- followed by warpAffine's inversion of the matrix (no WARP_INVERSE_MAP).  -> iM [6]."""
    s, d = np.asarray(src_tri, np.float32).astype(np.float64), np.asarray(dst_tri, np.float32).astype(np.float64)
    a = np.zeros((6, 7))
    a[0::2, 0:2], a[0::2, 2], a[1::2, 3:5], a[1::2, 5] = s, 1.0, s, 1.0
    a[0::2, 6], a[1::2, 6] = d[:, 0], d[:, 1]
    singular = False
    for i in range(6):
        k = i + int(np.argmax(np.abs(a[i:, i])))              # first maximum, like the `>` scan
        if abs(a[k, i]) < np.finfo(np.float64).eps * 100:     # LU64f's threshold: cv::solve then returns false and zeroes X,
            singular = True                                   # getAffineTransform does not look at the result -- a collinear source
            break                                             # triangle gets the all-zero matrix, and warpAffine's inverse of it is zero
        if k != i:
            a[[i, k], i:] = a[[k, i], i:]
        alpha = a[i + 1:, i] * (-1.0 / a[i, i])
        a[i + 1:, i + 1:] += alpha[:, None] * a[i, i + 1:][None, :]
    x = np.zeros(6)
    for i in range(5, -1, -1):
        if singular:
            break
        acc = a[i, 6]
        for k in range(i + 1, 6):
            acc -= a[i, k] * x[k]
        x[i] = acc / a[i, i]
    m = x.copy()
    det = m[0] * m[4] - m[1] * m[3]
    det = 1.0 / det if det != 0 else 0.0
    a11, a22 = m[4] * det, m[0] * det
    m[0], m[1], m[3], m[4] = a11, m[1] * -det, m[3] * -det, a22
    b1 = -m[0] * m[2] - m[1] * m[5]
    b2 = -m[3] * m[2] - m[4] * m[5]
    m[2], m[5] = b1, b2
    return m


def _cv_line8(p, q):
    """Pixels of Line(img, p, q, color, 8): LineIterator(connectivity 8, left_to_right) in closed form -- step k of the major axis has made
    (2 dy k + dx - 1) // (2 dx) minor steps (the iterator's error term steps when the ideal minor coordinate exceeds the current one by MORE
    than one half)."""
    (x1, y1), (x2, y2) = (int(p[0]), int(p[1])), (int(q[0]), int(q[1]))
    if x2 < x1:
        x1, y1, x2, y2 = x2, y2, x1, y1
    dx, dy = x2 - x1, y2 - y1
    ys = -1 if dy < 0 else 1
    dy = abs(dy)
    steep = dy > dx
    major, minor = (dy, dx) if steep else (dx, dy)
    k = np.arange(major + 1, dtype=np.int64)
    m = (2 * minor * k + major - 1) // (2 * major) if major > 0 else np.zeros(1, np.int64)
    return (x1 + m, y1 + ys * k) if steep else (x1 + k, y1 + ys * m)


def _cv_fill_convex_poly(height, width, pts):
    """bool mask [height, width] of the pixels cv2.fillConvexPoly(mask, np.int32(pts), color, lineType=16, shift=0) writes on a non-8U image
    (LINE_AA falls back to 8-connected there): the Bresenham outline plus FillConvexPoly's scanline spans.  The two edge walkers advance by a
    constant 16.16 step between vertex rows, so every run of rows between two vertex events is filled at once."""
    v = [(int(a), int(b)) for a, b in pts]
    n = len(v)
    mask = np.zeros((height, width), bool)
    for i in range(n):
        xs, ys = _cv_line8(v[i - 1], v[i])
        mask[ys, xs] = True
    ymin, ymax = min(p[1] for p in v), max(p[1] for p in v)
    imin = min(range(n), key=lambda i: (v[i][1], i))
    if n < 3 or max(p[0] for p in v) < 0 or ymax < 0 or min(p[0] for p in v) >= width or ymin >= height:
        return mask
    ymax = min(ymax, height - 1)
    ed = [[imin, 1, -(1 << _XY_SHIFT), 0, ymin], [imin, n - 1, -(1 << _XY_SHIFT), 0, ymin]]         # idx, di, x, dx, ye
    edges, y = n, ymin
    cols = np.arange(width)
    while y <= ymax:
        for e in ed:
            if y >= e[4]:
                idx0, idx = e[0], (e[0] + e[1]) % n
                while True:
                    edges -= 1
                    if edges < 0:
                        break
                    ty = v[idx][1]
                    if ty > y:
                        xs, xe = v[idx0][0] << _XY_SHIFT, v[idx][0] << _XY_SHIFT
                        num, den = (xe - xs) * 2 + (ty - y), 2 * (ty - y)
                        e[0], e[2], e[3], e[4] = idx, xs, (abs(num) // den) * (1 if num >= 0 else -1), ty      # C division: toward zero
                        break
                    idx0, idx = idx, (idx + e[1]) % n
        if edges < 0:
            break
        y_next = min(ed[0][4], ed[1][4], ymax + 1)              # rows [y, y_next): both walkers keep their step
        y_next = max(y_next, y + 1)
        k = np.arange(y_next - y, dtype=np.int64)
        xa, xb = ed[0][2] + k * ed[0][3], ed[1][2] + k * ed[1][3]
        lo, hi = np.minimum(xa, xb), np.maximum(xa, xb)
        xx1, xx2 = (lo + (1 << 15)) >> _XY_SHIFT, (hi + (1 << 15)) >> _XY_SHIFT
        rows = np.arange(y, y_next)
        ok = (rows >= 0) & (xx2 >= 0) & (xx1 < width)
        span = (cols[None, :] >= np.maximum(xx1, 0)[:, None]) & (cols[None, :] <= np.minimum(xx2, width - 1)[:, None]) & ok[:, None]
        mask[rows[ok]] |= span[ok]
        steps = y_next - y
        ed[0][2] += steps * ed[0][3]
        ed[1][2] += steps * ed[1][3]
        y = y_next
    return mask


def warp_plan(points_src, points_dst, height, width):
    """Host side of the Delaunay warp (1024_warp_morphs.py:163-203): triangulate the DESTINATION points with scipy (as the script does) and,
    per triangle in list order, do what morphTriangle does before any pixel is touched -- the two bounding rectangles, the fillConvexPoly
    mask of np.int32(tRect), cv2.getAffineTransform(np.float32(t1Rect), np.float32(tRect)) inverted as warpAffine inverts it.
    Returns (label int32 [height, width]: the triangle that writes each pixel LAST, -1 = none; records: one per triangle -- iM [6] float64,
    destination patch origin, source patch origin and size; simplices)."""
    from scipy.spatial import Delaunay
    ps, pd = np.asarray(points_src, np.float64), np.asarray(points_dst, np.float64)
    simplices = Delaunay(pd).simplices
    label = np.full((height, width), -1, np.int32)
    recs = []
    for t, idx in enumerate(simplices):
        tg, ta = ps[idx], pd[idx]
        r1, r = _cv_bounding_rect(tg), _cv_bounding_rect(ta)
        if r[0] < 0 or r[1] < 0 or r[0] + r[2] > width or r[1] + r[3] > height or r1[0] < 0 or r1[1] < 0:
            raise ValueError("warp: mesh points must lie inside the image (the script's numpy slices would not line up otherwise)")
        t_rect, t1_rect = ta - np.array(r[:2], np.float64), tg - np.array(r1[:2], np.float64)
        sw, sh = min(r1[0] + r1[2], width) - r1[0], min(r1[1] + r1[3], height) - r1[1]       # img_G[r1.y : r1.y + r1.h, ...]: numpy clips the slice
        if sw < 1 or sh < 1:
            raise ValueError("warp: a source triangle lies outside the image")
        m = _cv_fill_convex_poly(r[3], r[2], np.int32(t_rect))
        view = label[r[1]:r[1] + r[3], r[0]:r[0] + r[2]]
        view[m] = t
        recs.append((_cv_affine_inverse(np.float32(t1_rect), np.float32(t_rect)), r[0], r[1], r1[0], r1[1], sw, sh))
    return label, recs, simplices


def _pack_warp_records(recs):
    """The device records of mgf_cv_warp_triangles_f32: { double im[6]; int32 dx, dy, sx, sy, sw, sh }."""
    size = int(_lib.lib().mgf_cv_warp_triangle_bytes())
    dt = np.dtype([("im", np.float64, 6), ("geo", np.int32, 6)])
    assert dt.itemsize == size, (dt.itemsize, size)
    arr = np.zeros(len(recs), dt)
    for i, (im, *geo) in enumerate(recs):
        arr["im"][i], arr["geo"][i] = im, geo
    return arr


def warp_morph(img, points_G, points_avg, background=0.0):
    """Warp the generated morph so that its landmarks `points_G` land on the averaged landmarks `points_avg` (both [P,2] in pixel
    coordinates, frame points included): the post-process of 1024_warp_morphs.py:163-210, every pixel computed as OpenCV computes it
    (fixed-point 1/32-pixel coordinates, float weight table, BORDER_REFLECT_101 at the triangle's source patch, fillConvexPoly's pixel set).
    img: [1,C,H,W] or [C,H,W] float32 device tensor in ANY value range -- the script works on the 0..255 float BGR image it reads back from
    the PNG, for which every step is exact; returns imgMorph, a tensor of the same shape (`warp_morph_u8` gives the bytes the script writes)."""
    _lib.require_gpu(img)
    x = img.reshape(-1, *img.shape[-2:]).contiguous().float()
    c, h, w = x.shape
    label, recs, _ = warp_plan(points_G, points_avg, h, w)
    dev = x.device
    lab_d = torch.as_tensor(label, device=dev).contiguous()
    rec_d = torch.as_tensor(_pack_warp_records(recs).view(np.uint8).reshape(-1), device=dev).contiguous()
    out = torch.empty_like(x)
    _lib.check(_lib.lib().mgf_cv_warp_triangles_f32(out.data_ptr(), x.data_ptr(), lab_d.data_ptr(), rec_d.data_ptr(), len(recs), c, h, w,
                                                    float(background), _lib.stream_ptr()), "cv_warp_triangles")
    return out.reshape(img.shape)


def warp_morph_u8(img_u8_hwc, points_G, points_avg):
    """uint8 [H,W,C] in -> uint8 [H,W,C] out, what the script does between cv2.imread and cv2.imwrite (:186-207): np.float32(image), the
    triangle loop, np.uint8(imgMorph) (truncation)."""
    a = np.asarray(img_u8_hwc)
    assert a.dtype == np.uint8 and a.ndim == 3
    x = torch.from_numpy(np.ascontiguousarray(a.transpose(2, 0, 1))).to("cuda").float()
    out = warp_morph(x, points_G, points_avg)
    return out.to(torch.uint8).permute(1, 2, 0).contiguous().cpu().numpy()           # values in [0, 255]: the cast truncates like np.uint8


def frame_points(size: int):
    """The 12 frame points of 1024_warp_morphs.py:130-132 for a `size` x `size` image (0, 341, 682, 1023 at 1024)."""
    q = [(size - 1) * i // 3 for i in range(4)]
    return [[q[0], q[0]], [q[0], q[1]], [q[0], q[2]], [q[0], q[3]], [q[1], q[0]], [q[2], q[0]], [q[3], q[0]], [q[3], q[1]], [q[3], q[2]],
            [q[3], q[3]], [q[1], q[3]], [q[2], q[3]]]


def gray_u8_device(img):
    """[n,3,H,W] float32 device images -> [n,H,W] uint8 numpy: the gray image `get_landmarks_G` hands to dlib (1024_warp_morphs.py:61-66,
    the same two cv2 calls as the projection drivers), made on the device (mgf_reference_gray_u8)."""
    _lib.require_gpu(img)
    x = img.contiguous().float()
    n, c, h, w = x.shape
    assert c == 3
    out = torch.empty(n, h, w, dtype=torch.uint8, device=x.device)
    scratch = torch.empty(n * int(_lib.lib().mgf_reference_gray_scratch_floats()), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().mgf_reference_gray_u8(out.data_ptr(), x.data_ptr(), n, h, w, scratch.data_ptr(), _lib.stream_ptr()), "reference_gray_u8")
    return out.cpu().numpy()


def warp_morphs(G, w1, w2, landmark1, landmark2, landmark_G=None, landmark_fn=None, out_dir=None, truncation_psi=0.7, noise_mode="random"):
    """The whole of 1024_warp_morphs.py's main (:128-210): the 0.5 / 0.5 latent morph rendered with `G(W, truncation_psi)` (:153-156),
    the averaged landmarks of the two bona fide images + the 12 frame points triangulated (scipy Delaunay, :160-164), the morph's own
    landmarks (:166-168: `get_landmarks_G` -- here `landmark_G` [68,2], or `landmark_fn(gray uint8 [H,W])` called on the drivers' gray image
    of the morph), then every triangle of the morph warped onto the averaged mesh (:186-200) and written as bytes (:203-206).
    landmark1 / landmark2: [68,2] detections on the two source images (dlib is the caller's; `get_landmarks_img`, :46-56).
    Writes morph_G.png and Morph_final.png under `out_dir` when given.  Returns dict(latent [1,k,D], morph uint8 [H,W,3] (what the script
    reads back from morph_G.png), warped uint8 [H,W,3] (Morph_final.png), points_G, points_avg)."""
    lat, imgs = merge_morph(G, w1, w2, (0.5,), truncation_psi, noise_mode=noise_mode)
    img = imgs[0:1]
    size = int(img.shape[-1])
    morph_u8 = to_uint8_image(G, img)                       # the bytes misc.to_pil writes and cv2.imread reads back (channel order does not matter: per-channel work)
    if landmark_G is None:
        if landmark_fn is None:
            raise ValueError("warp_morphs: pass the morph's landmarks (landmark_G) or a detector (landmark_fn)")
        landmark_G = landmark_fn(gray_u8_device(img)[0])
        if landmark_G is None:                              # (the script would fail on `None.detach()` two lines later)
            raise _lib.MgfError("warp_morphs: the detector found no face in the morph")
    l1, l2, lg = (np.asarray(v.detach().cpu() if isinstance(v, torch.Tensor) else v, dtype=np.float64).reshape(-1, 2) for v in (landmark1, landmark2, landmark_G))
    if not (l1.shape == l2.shape == lg.shape):
        raise ValueError(f"warp_morphs: landmark sets differ in shape: {l1.shape}, {l2.shape}, {lg.shape}")
    extra = np.asarray(frame_points(size), dtype=np.float64)
    points_avg = np.concatenate([(l1 + l2) / 2, extra])    # torch.div(landmark1.add(landmark2), 2) on DoubleTensors
    points_G = np.concatenate([lg, extra])
    warped = warp_morph_u8(morph_u8, points_G, points_avg)
    if out_dir is not None:
        from PIL import Image
        os.makedirs(out_dir, exist_ok=True)
        Image.fromarray(morph_u8, "RGB").save(os.path.join(out_dir, "morph_G.png"))
        Image.fromarray(warped, "RGB").save(os.path.join(out_dir, "Morph_final.png"))
    return {"latent": lat[0], "morph": morph_u8, "warped": warped, "points_G": points_G, "points_avg": points_avg}


def cv_resize_linear_u8(img_u8_hwc, width, height):
    """cv2.resize(I, (width, height)) of a uint8 image with the default INTER_LINEAR, restated from OpenCV's published sources (imgproc
    resize.cpp: half-pixel centres, coefficients rounded to 11-bit fixed point, HResizeLinear in int32, VResizeLinear<uchar>'s
    `(((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2`).  Host-side input formatting of one small image, like
    image_transform; cv2 is absent offline, so bit-parity with it is UNPINNED."""
    a = np.asarray(img_u8_hwc)
    assert a.dtype == np.uint8 and a.ndim == 3
    ih, iw = a.shape[:2]

    from .lbp import resize_table as table

    x0, x1, a0, a1 = table(width, iw)
    y0, y1, b0, b1 = table(height, ih)
    src = a.astype(np.int64)
    hrow = src[:, x0] * a0[None, :, None] + src[:, x1] * a1[None, :, None]             # [ih, width, c], scale 2048
    s0, s1 = hrow[y0], hrow[y1]
    out = (((b0[:, None, None] * (s0 >> 4)) >> 16) + ((b1[:, None, None] * (s1 >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def facenet_feature(image_u8_hwc, embedder):
    """extract_FaceNet.py:30-40: `cv2.resize(I, (224, 224))`, clamp to [0, 255], (x - 127.5) / 128, InceptionResnetV1 -> the flattened
    512-d embedding.  embedder: facenet.InceptionResnetV1Embedder (the HIP network; its weights are the caller's)."""
    small = cv_resize_linear_u8(image_u8_hwc, 224, 224)
    x = torch.from_numpy(small.astype(np.float32)).clamp_(0, 255).sub_(127.5).div_(128.0).permute(2, 0, 1)[None].contiguous()
    emb = embedder(x.to(embedder.device if hasattr(embedder, "device") else "cuda"))
    return emb.detach().cpu().numpy().reshape(-1)


def second_stage(G, target, w_init, latent_std, lm_target, lm_steps, **kw):
    """A second projection whose noisy candidates are drawn around an earlier result instead of the latent mean
    (edit_MSE.py: `latent_in = w1` pattern; BASELINE config 5)."""
    w0 = torch.as_tensor(np.asarray(w_init.detach().cpu() if isinstance(w_init, torch.Tensor) else w_init, dtype=np.float32))
    return project_image(G, target, lm_target, lm_steps, latent_mean=w0.reshape(w0.shape[-2:]).to(G.device),
                         latent_std=latent_std, **kw)
