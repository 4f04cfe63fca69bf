"""Command-line front ends with the reference scripts' arguments and defaults:

    python -m morphganformer_amd.cli generate --model net.pkl --output-dir images --images-num 32 --truncation-psi 0.7
                                              (1024_generate.py:44-54)
    python -m morphganformer_amd.cli project  --model net.pkl --image face.png --landmarks lm.npz --path_to_gen out/
                                              (1024_example_wing_loss_perceptual_sqz_MSE.py:222-268; argparse names kept)
    python -m morphganformer_amd.cli morph    --model net.pkl --w1 a.mat --w2 b.mat --alphas 0,0.1,...,1 --out out/a+b
                                              (1024_merge_morph_2.py:25-92)

`--gpus` pins the visible device like the scripts' CUDA_VISIBLE_DEVICES line.  The reference detects landmarks with dlib on the
target and on every generated image; dlib is a closed third-party dependency, so `project` takes them from `--landmarks`
(an .npz with `target` [68,2] and `steps` [>=step,68,2], e.g. written by a detector run elsewhere) or, without it, runs the
MSE(+LPIPS) objective only (the 1024_example_MSE.py / ..._percept.py variants).
"""
from __future__ import annotations

import argparse
import os
import sys


def _loop_arguments(p):
    """The projection loop's arguments, shared by `project` and `morph-pairs` (1024_example_wing_loss_perceptual_sqz_MSE.py:222-245)."""
    p.add_argument("--model", type=str, default="models/ffhq-snapshot-1024_v2.pkl")
    p.add_argument("--gpus", type=str, default="0")
    p.add_argument("--size", type=int, default=1024)
    p.add_argument("--n_mean_latent", type=int, default=10000)
    p.add_argument("--step", type=int, default=5000)
    p.add_argument("--lamda", type=float, default=0.01)
    p.add_argument("--beta", type=float, default=1)
    p.add_argument("--lr_rampup", type=float, default=0.05)
    p.add_argument("--lr_rampdown", type=float, default=0.25)
    p.add_argument("--lr", type=float, default=0.01)
    p.add_argument("--noise", type=float, default=0.05)
    p.add_argument("--noise_ramp", type=float, default=0.75)
    p.add_argument("--ratio", type=float, default=1.0)
    p.add_argument("--truncation_psi", type=float, default=0.7)
    p.add_argument("--noise_regularize", type=float, default=1e5)       # accepted and unused, like the reference
    p.add_argument("--w_plus", action="store_true",
                   help="with --mode gradient: optimise the per-layer latent W+ [k, num_ws, D] instead of z (the reference accepts the flag "
                        "and never reads it; in literal mode it stays unused here too)")
    p.add_argument("--percept_weight", type=float, default=1.0, help="coefficient of the LPIPS term (0.5 with --beta 0.5 = 1024_example_percept_MSE.py)")
    p.add_argument("--pixel-term", choices=["mse", "psnr", "dssim", "lbp"], default="mse",
                   help="psnr = the pixel term of 1024_example_PSNR.py (10 log10(255^2 / MSE), minimised like the script does, and -- see --psnr-layout -- "
                        "with the script's element order; use with --no-lpips); "
                        "dssim = (1 - SSIM) / 2 of the uint8 images (1024_example_SSIM.py's `dssim`); lbp = the LBP matching distance of "
                        "1024_example_LBP_percept.py, the whole objective of that script (use with --no-lpips; literal mode)")
    p.add_argument("--psnr-layout", choices=["script", "aligned"], default="script",
                   help="script = 1024_example_PSNR.py:150-158 as written: the candidate's C-H-W stream against the target's H-W-C stream (different pixels "
                        "are paired); aligned = the PSNR of corresponding pixels (a deviation from the script)")
    p.add_argument("--latent-copies", type=int, default=1,
                   help="projection_example_v2_percept.py:131-166: this many noisy copies of the latent per step, averaged (torch.mean order) before the "
                        "generator (18 there, with --min-loss-init 1.0 --net vgg --no-mse --pool-above 256); literal mode")
    p.add_argument("--min-loss-init", type=float, default=100.0, help="the loop's starting min_loss (100 in the 1024 drivers, 1.0 in the v2 drivers)")
    p.add_argument("--pool-above", type=int, default=0,
                   help="projection_example_v1.py:150-155: block-average generated images taller than this (256 there) by height // N before the "
                        "image-space losses; the target image is then transformed to that size")
    p.add_argument("--net", type=str, default="squeeze", choices=["squeeze", "vgg", "alex"], help="LPIPS backbone")
    p.add_argument("--no-lpips", action="store_true", help="MSE(+Wing) only, the 1024_example_MSE.py objective")
    p.add_argument("--lpips-backbone", type=str, default=None, metavar="STATE_DICT",
                   help="torchvision feature weights of --net (.pth state dict or .npz with `features.N...` keys) -- what the reference "
                        "fetches with pretrained=True; required unless --no-lpips or --lpips-random-backbone")
    p.add_argument("--lpips-random-backbone", action="store_true",
                   help="score with SEEDED RANDOM backbone features (smoke runs only: the LPIPS term is then not a perceptual distance)")
    p.add_argument("--batch", type=int, default=32,
                   help="loop steps evaluated per generator forward in literal mode (same result; 32 = the benchmarked configuration, 51 GB of "
                        "activations at 1024^2)")
    p.add_argument("--pipeline", type=int, default=-1, choices=[-1, 0, 1],
                   help="literal mode: score one batch of candidates on a side stream while the generator synthesises the next (same result); -1 = where it pays "
                        "(LPIPS(squeeze) without an embedder: +1.9 %% iterations/s), 0 = never, 1 = always")
    p.add_argument("--keep-images", type=int, default=64,
                   help="device slots for the scored image of every improvement (spilled to the host between launch sequences, so every "
                        "improvement gets its PNG like the reference; raised to --batch if smaller)")
    p.add_argument("--mode", type=str, default="literal", choices=["literal", "gradient"],
                   help="literal = the loop as the reference executes it (best-of-N noisy sampling); gradient = back-propagate the loss "
                        "into the latent and let Adam move it")
    p.add_argument("--seed", type=int, default=None)
    p.add_argument("--biometric", type=str, default="none", choices=["none", "facenet", "iresnet18", "iresnet34", "iresnet50", "iresnet100"],
                   help="add gamma * MSE(embed(img), embed(target)): facenet = InceptionResnetV1 on the un-resized image, the term "
                        "1024_example_FaceNet_percept.py:147-158 scores with (alone: --no-lpips --no-mse); iresnetNN = the vendored ArcFace network")
    p.add_argument("--gamma", type=float, default=1.0, help="coefficient of the biometric term")
    p.add_argument("--biometric-weights", type=str, default=None, metavar="STATE_DICT",
                   help="the embedder's state dict (.pth / .npz; facenet_pytorch's vggface2 weights, an insightface iresnet checkpoint) -- what the "
                        "reference fetches by name; required with --biometric unless --biometric-random")
    p.add_argument("--biometric-random", action="store_true", help="seeded random embedder weights (smoke runs only)")
    p.add_argument("--no-mse", action="store_true", help="drop the MSE term (beta * MSE)")


def build_parser():
    ap = argparse.ArgumentParser(prog="morphganformer_amd", description="MI355X latent-projection / GANformer drivers")
    sub = ap.add_subparsers(dest="cmd", required=True)

    g = sub.add_parser("generate", help="Generate images using a pretrained network pickle")
    g.add_argument("--model", type=str, required=True)
    g.add_argument("--gpus", type=str, default="0")
    g.add_argument("--output-dir", type=str, default="images")
    g.add_argument("--images-num", type=int, default=32)
    g.add_argument("--truncation-psi", type=float, default=0.7)
    g.add_argument("--ratio", type=float, default=1.0)
    g.add_argument("--seed", type=int, default=None)

    p = sub.add_parser("project", help="Project one face image into the latent space")
    p.add_argument("--image", type=str, required=True)
    p.add_argument("--landmarks", type=str, default=None)
    p.add_argument("--path_to_gen", type=str, default="images/projection/")
    _loop_arguments(p)

    q = sub.add_parser("morph-pairs", help="Project both images of every CSV pair and render their latent morph "
                                           "(projection_example_v2_percept_morph.py:330-365; BASELINE config 3's outer loop)")
    q.add_argument("--csv", type=str, required=True, help="rows `img1,img2,similarity`; the header row and rows below --threshold are skipped")
    q.add_argument("--threshold", type=float, default=0.5)
    q.add_argument("--src", type=str, required=True, help="directory of the bona fide images")
    q.add_argument("--dst-raw", type=str, required=True, help="<a>_<b>_A.png / _B.png: the two projections")
    q.add_argument("--dst-morph", type=str, required=True, help="<a>_<b>.png: the 0.5 / 0.5 latent morph")
    q.add_argument("--dynamic", action="store_true", help="ranks pull images from a shared work queue instead of images[rank::world]")
    _loop_arguments(q)

    m = sub.add_parser("morph", help="Render linear morphs of two projected latents")
    m.add_argument("--model", type=str, required=True)
    m.add_argument("--w1", type=str, required=True)
    m.add_argument("--w2", type=str, required=True)
    m.add_argument("--alphas", type=str, default="0.5")
    m.add_argument("--out", type=str, required=True, help="output prefix: <out>_a0.50.jpg / .mat")
    m.add_argument("--gpus", type=str, default="0")
    m.add_argument("--ratio", type=float, default=1.0)
    m.add_argument("--truncation_psi", type=float, default=0.7)
    mf = sub.add_parser("merge-files", help="Fold a results tree <src>/<version>/<variant>/<id>/<name>/ over its variants (1024_merge_files.py); no model, no GPU")
    mf.add_argument("--src", type=str, required=True)
    mf.add_argument("--dst", type=str, required=True)
    t = sub.add_parser("morph-tree", help="Morph every latent pair of every id folder (the directory walk of 1024_merge_morph_2.py)")
    t.add_argument("--model", type=str, required=True)
    t.add_argument("--src", type=str, required=True, help="<src>/<id>/<name>/*.mat: two name folders per id")
    t.add_argument("--dst", type=str, required=True, help="<dst>/<id>/<stem1>+<stem2>.jpg / .mat")
    t.add_argument("--gpus", type=str, default="0")
    t.add_argument("--ratio", type=float, default=1.0)
    t.add_argument("--truncation_psi", type=float, default=0.7)
    w = sub.add_parser("warp", help="Morph two projected latents and warp the morph onto the averaged landmarks (1024_warp_morphs.py)")
    w.add_argument("--model", type=str, required=True)
    w.add_argument("--w1", type=str, required=True, help=".mat latent of the first bona fide image")
    w.add_argument("--w2", type=str, required=True)
    w.add_argument("--landmarks", type=str, required=True,
                   help=".npz with `lm1`, `lm2` [68,2] (detections on the two source images) and `lm_G` [68,2] (on the morph)")
    w.add_argument("--out", type=str, required=True, help="output directory: morph_G.png, Morph_final.png")
    w.add_argument("--gpus", type=str, default="0")
    w.add_argument("--truncation_psi", type=float, default=0.7)
    e = sub.add_parser("extract-facenet", help="The FaceNet feature of image files (`facenet_feature`, extract_FaceNet.py:30-40): 224x224 "
                                               "cv2-style resize, (x - 127.5) / 128, InceptionResnetV1 -> 512 numbers per image")
    e.add_argument("images", nargs="+", help="image files")
    e.add_argument("--out", type=str, required=True, help=".mat (scipy.io.savemat: `names`, `features` [N, 512]) or .npy ([N, 512])")
    e.add_argument("--biometric-weights", type=str, default=None, metavar="STATE_DICT",
                   help="facenet_pytorch's InceptionResnetV1 state dict (.pth / .npz) -- what the reference fetches with pretrained='vggface2'")
    e.add_argument("--biometric-random", action="store_true", help="seeded random embedder weights (smoke runs only)")
    e.add_argument("--gpus", type=str, default="0")
    return ap


def _extract_facenet(a):
    import numpy as np
    import torch
    from PIL import Image
    from . import drivers
    from .facenet import InceptionResnetV1Embedder, random_state
    if a.biometric_weights is None and not a.biometric_random:
        raise SystemExit("extract-facenet: the embedder's weights are needed: --biometric-weights <state dict> (or --biometric-random)")
    if a.biometric_weights:
        state = dict(np.load(a.biometric_weights)) if a.biometric_weights.endswith(".npz") else \
            {k: v.numpy() for k, v in torch.load(a.biometric_weights, map_location="cpu", weights_only=True).items()}
    else:
        print("WARNING: seeded random InceptionResnetV1 weights (--biometric-random); the output is not a face embedding")
        state = random_state(0)
    net = InceptionResnetV1Embedder(state, n=1, device="cuda")
    feats = [drivers.facenet_feature(np.asarray(Image.open(f).convert("RGB")), net) for f in a.images]
    feats = np.stack(feats).astype(np.float32)
    if a.out.endswith(".npy"):
        np.save(a.out, feats)
    else:
        import scipy.io as sio
        sio.savemat(a.out, {"names": np.array(a.images, dtype=object), "features": feats})
    print(f"{len(a.images)} feature(s) -> {a.out}")
    return 0


def main(argv=None):
    a = build_parser().parse_args(argv)
    if a.cmd == "merge-files":                              # file bookkeeping only (1024_merge_files.py): no generator, no GPU
        from . import drivers
        files = drivers.merge_files(a.src, a.dst)
        print(f"copied {len(files)} files")
        return 0
    launched = int(os.environ.get("WORLD_SIZE", "1")) > 1 and "LOCAL_RANK" in os.environ
    # read by the HSA runtime when it starts, i.e. at the first GPU call (loader.load_network below): set here, before torch is imported,
    # like bench.py does -- the host driver only supports dmabuf IPC, RCCL fails without it
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if not launched:
        os.environ.setdefault("CUDA_VISIBLE_DEVICES", a.gpus)
    import numpy as np
    import torch
    if launched:                                              # one process per GPU (torch.distributed.run): rank r works on device LOCAL_RANK
        torch.cuda.set_device(int(os.environ["LOCAL_RANK"]))
    if a.cmd == "extract-facenet":                          # no generator involved
        return _extract_facenet(a)
    from . import drivers, loader
    from .projection import ProjectionArgs

    print("Loading networks...")
    G = loader.load_network(a.model, device="cuda")["Gs"]
    if a.cmd == "generate":
        print("Generate and save images...")
        drivers.generate_images(G, a.images_num, a.truncation_psi, a.output_dir, a.ratio, seed=a.seed)
        return 0
    if a.cmd == "morph-tree":
        for stem in drivers.merge_morph_tree(G, a.src, a.dst, a.truncation_psi, a.ratio):
            print(stem)
        return 0
    if a.cmd == "warp":
        lm = np.load(a.landmarks)
        drivers.warp_morphs(G, drivers.load_latent_mat(a.w1), drivers.load_latent_mat(a.w2), lm["lm1"], lm["lm2"], landmark_G=lm["lm_G"],
                            out_dir=a.out, truncation_psi=a.truncation_psi)
        print("done")
        return 0
    if a.cmd == "morph":
        alphas = [float(v) for v in a.alphas.split(",")]
        drivers.merge_morph(G, drivers.load_latent_mat(a.w1), drivers.load_latent_mat(a.w2), alphas, a.truncation_psi,
                            out_prefix=a.out, ratio=a.ratio)
        return 0
    # project
    from .lpips import PerceptualLoss
    args = ProjectionArgs(step=a.step, lamda=a.lamda, beta=a.beta, lr=a.lr, lr_rampup=a.lr_rampup, lr_rampdown=a.lr_rampdown,
                          noise=a.noise, noise_ramp=a.noise_ramp, truncation_psi=a.truncation_psi, n_mean_latent=a.n_mean_latent,
                          ratio=a.ratio, percept_weight=a.percept_weight, pixel_term=a.pixel_term, psnr_layout=a.psnr_layout, pool_above=a.pool_above,
                          latent_copies=a.latent_copies, min_loss_init=a.min_loss_init)
    percept = None
    if not a.no_lpips:
        if a.lpips_backbone is None and not a.lpips_random_backbone:
            raise SystemExit(f"{a.cmd}: the LPIPS term needs the torchvision backbone weights: --lpips-backbone <state dict> "
                             "(or --no-lpips / --lpips-random-backbone)")
        from .lpips import load_backbone_state
        state = load_backbone_state(a.lpips_backbone) if a.lpips_backbone else None
        if state is None:
            print("WARNING: LPIPS runs on seeded random backbone weights (--lpips-random-backbone); the term is not a perceptual distance")
        percept = PerceptualLoss(model="net-lin", net=a.net, use_gpu=True, device=G.device, backbone_state=state,
                                 allow_random_backbone=state is None)
    biometric = None
    if a.biometric != "none":
        if a.biometric_weights is None and not a.biometric_random:
            raise SystemExit(f"{a.cmd}: --biometric {a.biometric} needs the embedder's weights: --biometric-weights <state dict> (or --biometric-random)")
        from .iresnet import BiometricLoss
        state = None
        if a.biometric_weights:
            if a.biometric_weights.endswith(".npz"):
                state = dict(np.load(a.biometric_weights))
            else:
                state = torch.load(a.biometric_weights, map_location="cpu", weights_only=True)
                state = state.get("state_dict", state)
        else:
            print(f"WARNING: the biometric term runs on seeded random {a.biometric} weights (--biometric-random); it is not a face embedding")
        biometric = BiometricLoss(a.biometric, state=state, n=a.batch if a.mode == "literal" else 1, device=G.device)
    if percept is None and a.no_mse and biometric is None and not getattr(a, "landmarks", None) and a.pixel_term != "lbp":
        raise SystemExit(f"{a.cmd}: every term of the objective is switched off")
    if a.pixel_term == "lbp" and (a.cmd != "project" or a.mode != "literal"):
        raise SystemExit("--pixel-term lbp is the objective of the single-image literal loop (project --mode literal)")
    space = "w+" if (a.w_plus and a.mode == "gradient") else "z"
    if a.cmd == "morph-pairs":
        # one process per GPU under `python -m torch.distributed.run --nproc-per-node N -m morphganformer_amd.cli morph-pairs ...`: the
        # 2 x pairs projections are sharded over the ranks, one all_gather returns the latents, the renderings are dealt pairs[rank::world]
        import torch.distributed as dist
        if int(os.environ.get("WORLD_SIZE", "1")) > 1 and not dist.is_initialized():
            from .distributed import init_process_group
            init_process_group("nccl", device_id=torch.device("cuda", torch.cuda.current_device()))
        kw = dict(args=args, percept=percept, batch=a.batch, seed=a.seed, mode=a.mode, latent_space=space, keep_images=a.keep_images, pipeline=None if a.pipeline < 0 else (bool(a.pipeline) and a.mode == "literal"),
                  biometric=biometric, gamma=a.gamma, use_mse=not a.no_mse)
        if a.mode == "literal":
            kw["dynamic"] = a.dynamic
        res = drivers.morph_pairs(G, drivers.read_pair_csv(a.csv, a.threshold), a.src, a.dst_raw, a.dst_morph,
                                  truncation_psi=a.truncation_psi, ratio=a.ratio, **kw)
        for path in res["written"]:
            print(path)
        if dist.is_initialized():
            dist.barrier()
            dist.destroy_process_group()
        return 0
    tsize = a.size // (a.size // a.pool_above) if (a.pool_above and a.size > a.pool_above) else a.size       # the size the image-space losses see
    target = drivers.image_transform(a.image, size=tsize, device=G.device)
    lm_t = lm_s = None
    if a.landmarks:
        lm = np.load(a.landmarks)
        lm_t, lm_s = lm["target"], lm["steps"]
        if lm_s.shape[0] < a.step:
            raise SystemExit(f"--landmarks holds {lm_s.shape[0]} steps, --step is {a.step}")
    stem = os.path.splitext(os.path.basename(a.image))[0]
    lbp_target = None
    if a.pixel_term == "lbp":          # LBP_feature(path): the file's own pixels (1024_example_LBP_percept.py:40-45,140); min_distance starts at 1 (:151)
        from PIL import Image
        from . import lbp
        lbp_target = lbp.target_feature(np.asarray(Image.open(a.image).convert("RGB")), G.device)
        args.min_loss_init = 1.0
    res = drivers.project_image(G, target, lm_t, lm_s, args=args, percept=percept, batch=a.batch, seed=a.seed,
                                out_prefix=os.path.join(a.path_to_gen, stem), mode=a.mode, path_to_gen=a.path_to_gen,
                                keep_images=a.keep_images, latent_space=space, biometric=biometric, gamma=a.gamma, use_mse=not a.no_mse, pipeline=None if a.pipeline < 0 else (bool(a.pipeline) and a.mode == "literal"),
                                lbp_target=lbp_target)
    print(f"best step {res['step']}  loss {res['loss']:.6f}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
