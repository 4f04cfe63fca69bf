"""GPU parity for the callers either side of the projection loop and the remaining BASELINE configs:

 * configs[0]  256^2, 50 steps, MSE only            vs tests/golden/loop_config0_256.npz (a run of the REFERENCE modules)
 * config 4    11 linear morphs through G(dw, psi)  vs tests/golden/morph_tiny.npz (REFERENCE Generator)
 * config 5    second-stage projection initialised from an earlier result: vs the CPU oracle loop
 * loader.load_network on a reference-layout pickle -> HIP Generator, generate/project drivers' file outputs
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _tiny_G(seed=0, max_batch=1):
    from morphganformer_amd.engine import Generator
    from morphganformer_amd.synth_weights import TINY, make_state_dict
    return Generator(make_state_dict(TINY, seed=seed), TINY, "cuda", max_batch=max_batch)


@pytest.mark.parametrize("batch", [1, 5])
def test_config0_256_mse_only_matches_reference_run(golden, batch):
    from morphganformer_amd.engine import Generator
    from morphganformer_amd.projection import ProjectionArgs, ProjectionEngine
    from morphganformer_amd.synth_weights import SMALL256, make_state_dict
    g = golden("loop_config0_256.npz")
    G = Generator(make_state_dict(SMALL256, seed=0), SMALL256, "cuda", max_batch=batch)
    target = torch.from_numpy(g["target_u8"]).float().div(255).sub(0.5).div(0.5)[None].cuda()
    eng = ProjectionEngine(G, target, torch.from_numpy(g["latent_mean"]).cuda(), float(g["latent_std"]), ProjectionArgs(step=50),
                           percept=None, use_mse=True, eps=torch.from_numpy(g["eps"]).cuda(), noise_mode="const", batch=batch)
    lat, bstep, bloss, losses = eng.run().result()
    assert bstep == int(g["best_step"])
    assert np.array_equal(lat.numpy(), g["best_latent"])                      # bit-exact under injected noise
    assert np.abs(losses - g["losses"]).max() < 1e-3 * np.abs(g["losses"]).max()
    assert abs(bloss - float(g["best_loss"])) < 1e-3 * float(g["best_loss"])


def test_merge_morph_matches_reference_renderings(golden, tmp_path):
    from morphganformer_amd import drivers
    g = golden("morph_tiny.npz")
    G = _tiny_G()
    lat, imgs = drivers.merge_morph(G, g["w1"], torch.from_numpy(g["w2"]), alphas=[float(a) for a in g["alphas"]],
                                    truncation_psi=0.7, noise_mode="const", out_prefix=str(tmp_path / "m" / "a+b"))
    assert np.array_equal(lat, g["latents"])                                   # the blend itself: bit-exact
    err = (imgs.cpu().numpy() - g["images"])
    assert np.abs(err).max() < 2e-4 * np.abs(g["images"]).max()
    # psi given by keyword (1024_generate.py:35) really truncates; positionally (merge_morph / the loop) it lands in `c`
    img_kw = G(torch.from_numpy(g["w1"]).cuda(), truncation_psi=0.7, noise_mode="const")[0].cpu().numpy()
    assert np.abs(img_kw - g["img_w1_psi07"]).max() < 2e-4 * np.abs(g["img_w1_psi07"]).max()
    assert np.abs(img_kw - g["images"][0]).max() > 1e-2
    # files: 11 jpg + 11 mat, the .mat holds the blended latent under 'w'
    files = sorted(os.listdir(tmp_path / "m"))
    assert len([f for f in files if f.endswith(".jpg")]) == 11 and len([f for f in files if f.endswith(".mat")]) == 11
    assert np.array_equal(drivers.load_latent_mat(str(tmp_path / "m" / "a+b_a0.50.mat")), g["latents"][5])


def test_second_stage_projection_vs_oracle(golden):
    """Config 5: candidates are drawn around an earlier projection's result instead of latent_mean."""
    from morphganformer_amd import drivers
    from morphganformer_amd.projection import ProjectionArgs
    from morphganformer_amd.synth_weights import TINY, make_state_dict
    from oracle.generator_ref import generator_ref, to_torch_state
    from oracle.loss_ref import mse_ref, projection_literal_ref
    g = golden("loop_tiny.npz")
    G = _tiny_G()
    steps = 12
    w1 = g["best_latent"]                                                      # stage-1 result of the reference run
    eps = torch.from_numpy(np.random.Generator(np.random.PCG64(8)).standard_normal((steps, 1, TINY.k, TINY.z_dim)).astype(np.float32))
    target = torch.from_numpy(g["target"])
    res = drivers.second_stage(G, target.cuda(), w1, float(g["latent_std"]), None, None, args=ProjectionArgs(step=steps),
                               eps=eps.cuda(), noise_mode="const", batch=4)
    tsd = to_torch_state(make_state_dict(TINY, seed=0))
    ref = projection_literal_ref(lambda z: generator_ref(tsd, z, TINY, "const"), lambda i, img: float(mse_ref(img, target)),
                                 torch.from_numpy(w1[0]), float(g["latent_std"]), eps, steps)
    assert res["step"] == ref[1]
    assert torch.equal(res["w"], ref[0])
    assert np.allclose(res["losses"], np.array(ref[3]), rtol=1e-3)


def test_load_network_builds_the_hip_generator(tmp_path):
    from morphganformer_amd import loader
    from test_host_and_abi import _tiny_snapshot
    p = str(tmp_path / "net.pkl")
    _tiny_snapshot(p, seed=3)
    nets = loader.load_network(p, device="cuda")
    G = nets["Gs"]
    assert isinstance(nets["D"], loader.PersistentStub)                       # never instantiated on this path
    from morphganformer_amd.synth_weights import TINY, synthetic_latents
    z = torch.from_numpy(synthetic_latents(TINY, 1, seed=4)).cuda()
    a = G(z, None, noise_mode="const")[0]
    b = _tiny_G(seed=3)(z, None, noise_mode="const")[0]
    assert torch.equal(a, b)
    assert G.input_shape[1:] == [TINY.k, TINY.z_dim] and G.img_resolution == 64


def test_generate_and_project_drivers_write_what_the_reference_writes(tmp_path):
    from PIL import Image
    from morphganformer_amd import drivers
    from morphganformer_amd.projection import ProjectionArgs, synthetic_landmarks
    G = _tiny_G()
    zs = drivers.generate_images(G, images_num=3, truncation_psi=0.7, output_dir=str(tmp_path / "gen"), seed=1, noise_mode="const")
    assert sorted(os.listdir(tmp_path / "gen")) == [f"sample_{i:06d}.png" for i in range(3)]
    im = np.asarray(Image.open(tmp_path / "gen" / "sample_000001.png"))
    assert im.shape == (64, 64, 3) and im.dtype == np.uint8
    ref = G(zs[1].cuda(), truncation_psi=0.7, noise_mode="const")[0][0].cpu().numpy().transpose(1, 2, 0)
    assert np.array_equal(im, np.rint(ref * 127.5 + 127.5).clip(0, 255).astype(np.uint8))      # misc.to_pil's rounding
    # project: target image file -> transform -> loop -> .mat + best-of PNG
    Image.fromarray(im).resize((80, 72)).save(tmp_path / "face.png")
    target = drivers.image_transform(str(tmp_path / "face.png"), size=64)
    assert target.shape == (1, 3, 64, 64) and target.is_cuda
    steps = 8
    lm_t, lm_s = synthetic_landmarks(steps, 64, 2)
    out = drivers.project_image(G, target, lm_t, lm_s, args=ProjectionArgs(step=steps, n_mean_latent=500), seed=0,
                                out_prefix=str(tmp_path / "proj" / "face"), path_to_gen=str(tmp_path / "proj"), batch=3, noise_mode="random")
    assert out["w"].shape == (1, 17, 32) and 0 <= out["step"] < steps and out["loss"] == np.nanmin(out["losses"])
    assert np.array_equal(drivers.load_latent_mat(str(tmp_path / "proj" / "face.mat")), out["w"].numpy())
    # the drivers write `{step:06d}_{loss:04f}.png` of the scored image at EVERY improvement (...sqz_MSE.py:186-195): one file per
    # running-minimum step of the loss history, named with that step and loss
    hist, best, want = out["losses"], 100.0, []
    for i, v in enumerate(hist):
        if v < best:
            best = v
            want.append("{:06d}_{:04f}.png".format(i, v))
    pngs = sorted(f for f in os.listdir(tmp_path / "proj") if f.endswith(".png"))
    assert pngs == sorted(want) and want[-1].startswith(f"{out['step']:06d}_")
    assert np.asarray(Image.open(tmp_path / "proj" / want[-1])).shape == (64, 64, 3)


def test_improvement_trail_keeps_the_scored_images(golden):
    """keep_images: the image kept for every improvement is the candidate that was scored (bit-equal to a fresh const-noise rendering of
    that step's latent), in improvement order; with fewer slots than improvements the last slot holds the final best."""
    from morphganformer_amd.projection import ProjectionArgs, ProjectionEngine
    g = golden("loop_tiny.npz")
    steps = 50
    G = _tiny_G()

    def make(keep, batch):
        return ProjectionEngine(G, torch.from_numpy(g["target"]).cuda(), torch.from_numpy(g["latent_mean"]).cuda(), float(g["latent_std"]),
                                ProjectionArgs(step=steps), use_mse=True, lm_target=g["lm_target"], lm_steps=g["lm_steps"],
                                eps=torch.from_numpy(g["eps"]).cuda(), noise_mode="const", batch=batch, keep_images=keep)
    want, best = [], 100.0
    for i, v in enumerate(g["losses"]):
        if v < best:
            best = v
            want.append(i)
    assert len(want) >= 3 and want[-1] == int(g["best_step"])
    for batch in (1, 7):
        eng = make(16, batch)
        lat, bstep, bloss, losses = eng.run().result()
        trail = eng.improvements()
        assert [t[0] for t in trail] == want and trail[-1][1] == bloss
        for step, loss, img in trail:
            assert loss == losses[step]
            fresh = G(torch.from_numpy(g["latents_n"][step]).cuda(), None, noise_mode="const")[0]
            # (a fresh single-image rendering may take other kernel paths than the batch -- split-K, tile shapes: float32 rounding only)
            assert float((img.cuda() - fresh[0]).abs().max()) < 1e-5 * float(fresh.abs().max())   # (spilled trail images live on the host)
    # two slots, more improvements: first improvement in slot 0, the final best in slot 1 (also when several improve in one batch)
    eng = make(2, 50)
    eng.run()
    trail = eng.improvements()
    assert [t[0] for t in trail] == [want[0], want[-1]]
    fresh = G(torch.from_numpy(g["latents_n"][want[-1]]).cuda(), None, noise_mode="const")[0][0]
    assert float((trail[1][2].cuda() - fresh).abs().max()) < 1e-5 * float(fresh.abs().max())


def test_project_image_pipeline_option_equals_the_plain_loop(tmp_path):
    """drivers.project_image(pipeline=True): the losses of one batch of candidates on a side stream beside the generator of the next, the run's last
    launch sequence scoring alone -- the same latent, step, loss history and the same improvement PNGs (names and bytes) as the one-stream loop, also
    through a re-targeted engine, with a step count the batch does not divide and the improvement trail spilled on the way; and an engine built for one
    mode is refused for the other."""
    import glob
    from morphganformer_amd import drivers
    from morphganformer_amd.lpips import PerceptualLoss
    from morphganformer_amd.projection import ProjectionArgs, synthetic_landmarks
    from morphganformer_amd.synth_weights import TINY, synthetic_latents
    G = _tiny_G()
    P = PerceptualLoss(net="squeeze", allow_random_backbone=True)
    steps, batch = 37, 5
    zs = torch.from_numpy(synthetic_latents(TINY, 2, seed=77)).cuda()
    targets = [G(zs[j:j + 1], None, noise_mode="const")[0].clamp(-1, 1).clone() for j in range(2)]
    lms = [synthetic_landmarks(steps, 64, 600 + j) for j in range(2)]
    torch.manual_seed(9)
    latent_mean = torch.randn(TINY.k, TINY.z_dim, device="cuda")
    runs = {}
    for pipe in (False, True):
        eng, res = None, []
        for j in range(2):
            d = tmp_path / f"pipe{int(pipe)}_{j}"
            r = drivers.project_image(G, targets[j], lms[j][0], lms[j][1], args=ProjectionArgs(step=steps, min_loss_init=1e9), percept=P, batch=batch,
                                      seed=40 + j, latent_mean=latent_mean, latent_std=1.5, path_to_gen=str(d), keep_images=6, engine=eng,
                                      return_engine=True, pipeline=pipe, noise_mode="const")      # (random per-layer noise is drawn in launch order, which the two modes' set-up passes advance differently)
            eng = r["engine"]
            files = {os.path.basename(f): open(f, "rb").read() for f in sorted(glob.glob(str(d / "*.png")))}
            res.append((r["w"].clone(), r["step"], r["loss"], np.array(r["losses"]), files))
        assert eng.pipeline == pipe
        runs[pipe] = (res, eng)
    for (w0, s0, l0, h0, f0), (w1, s1, l1, h1, f1) in zip(runs[False][0], runs[True][0]):
        assert torch.equal(w0, w1) and s0 == s1 and l0 == l1 and np.array_equal(h0, h1)
        assert len(f0) >= 1 and f0 == f1
    with pytest.raises(ValueError, match="pipeline"):
        drivers.project_image(G, targets[0], lms[0][0], lms[0][1], args=ProjectionArgs(step=steps, min_loss_init=1e9), percept=P, batch=batch, seed=40,
                              latent_mean=latent_mean, latent_std=1.5, path_to_gen=str(tmp_path / "x"), keep_images=6, engine=runs[False][1], pipeline=True,
                              noise_mode="const")


def test_retargeted_engine_equals_fresh_engines(golden):
    """ProjectionEngine.retarget: one engine (one captured hipGraph, one set of workspaces) walked over three targets gives, for every
    target, the run of a freshly constructed engine bit for bit -- best latent, best step, loss history, improvement trail -- with the
    full objective (LPIPS + Wing + MSE), graph replay, several steps per forward and the trail spilled to the host on the way."""
    from morphganformer_amd.lpips import PerceptualLoss
    from morphganformer_amd.projection import ProjectionArgs, ProjectionEngine, synthetic_landmarks
    from morphganformer_amd.synth_weights import TINY, synthetic_latents
    g = golden("loop_tiny.npz")
    G = _tiny_G()
    steps, batch, keep = 40, 4, 8
    mean, std = torch.from_numpy(g["latent_mean"]).cuda(), float(g["latent_std"])
    targets = [G(torch.from_numpy(synthetic_latents(TINY, 1, 3000 + j)).cuda(), None, noise_mode="const")[0].clamp(-1, 1) for j in range(3)]
    lms = [synthetic_landmarks(steps, 64, 20 + j) for j in range(3)]

    def fresh(j):
        P = PerceptualLoss(net="squeeze", allow_random_backbone=True)
        return ProjectionEngine(G, targets[j], mean, std, ProjectionArgs(step=steps), percept=P, use_mse=True, lm_target=lms[j][0],
                                lm_steps=lms[j][1], noise_mode="const", seed=100 + j, batch=batch, keep_images=keep, use_graph=True)

    def outcome(eng):
        lat, bstep, bloss, hist = eng.run().result()
        trail = [(s_, l_, im.cpu().clone()) for s_, l_, im in eng.improvements()]
        return lat, bstep, bloss, hist, trail

    want = [outcome(fresh(j)) for j in range(3)]
    eng = fresh(0)
    graph = None
    for j in range(3):
        if j:
            eng.retarget(targets[j], lm_target=lms[j][0], lm_steps=lms[j][1], seed=100 + j)
        lat, bstep, bloss, hist, trail = outcome(eng)
        if graph is None:
            graph = eng.graph
        assert eng.graph is graph and graph is not None                     # the captured graph is reused, never re-captured
        w = want[j]
        assert bstep == w[1] and bloss == w[2] and torch.equal(lat, w[0]) and np.array_equal(hist, w[3]), j
        assert [(t[0], t[1]) for t in trail] == [(t[0], t[1]) for t in w[4]] and len(trail) >= 2
        assert all(torch.equal(a[2], b[2]) for a, b in zip(trail, w[4])), j
    assert not np.array_equal(want[0][3], want[1][3])                         # the targets really differ
    # the trail of a long run is complete although it has fewer device slots than improvements: it spills between launch sequences
    hist, best, impr = want[0][3], 100.0, []
    for i, v in enumerate(hist):
        if v < best:
            best = v
            impr.append(i)
    assert [t[0] for t in want[0][4]] == impr


def test_landmark_callback_mode_equals_injected_table(golden):
    """A host detector called on every generated image (the drivers' dlib step) gives the same run as the table it produces."""
    from morphganformer_amd.projection import ProjectionArgs, ProjectionEngine
    g = golden("loop_tiny.npz")
    steps = 10
    seen = []

    def detector(img_hwc):                     # deterministic stand-in: landmarks from image statistics, "no face" now and then
        assert img_hwc.shape == (64, 64, 3) and img_hwc.dtype == np.float32
        k = len(seen)
        seen.append(float(img_hwc.mean()))
        if k in (2, 5):
            return None
        return g["lm_target"] + np.round(40 * img_hwc[:34, :4, 0].astype(np.float64)).reshape(68, 2)

    def make(**kw):
        return ProjectionEngine(_tiny_G(), torch.from_numpy(g["target"]).cuda(), torch.from_numpy(g["latent_mean"]).cuda(),
                                float(g["latent_std"]), ProjectionArgs(step=steps), use_mse=True, lm_target=g["lm_target"],
                                eps=torch.from_numpy(g["eps"][:steps]).cuda(), noise_mode="const", **kw)
    e1 = make(landmark_fn=detector, batch=4)
    lat1, st1, loss1, hist1 = e1.run().result()
    assert len(seen) == steps and e1.graph is None
    table, valid = e1.lm_steps.cpu().numpy(), e1.valid.cpu().numpy()
    assert valid.tolist() == [1, 1, 0, 1, 1, 0, 1, 1, 1, 1]
    lat2, st2, loss2, hist2 = make(lm_steps=table, lm_valid=valid, batch=1).run().result()
    assert st1 == st2 and torch.equal(lat1, lat2) and np.isnan(hist1[2]) and np.isnan(hist1[5])
    assert np.allclose(np.nan_to_num(hist1), np.nan_to_num(hist2), rtol=1e-6)


@pytest.mark.parametrize("shape", [(1, 64, 64), (3, 96, 132), (2, 1024, 1024), (1, 5, 7)])
def test_reference_gray_u8_on_the_device_is_bit_exact(shape):
    """mgf_reference_gray_u8 -- cv2.normalize(NORM_MINMAX, CV_8U) + BGR2GRAY on RGB data, per candidate (...sqz_MSE.py:159-163) -- against
    drivers.reference_gray_u8 (the numpy statement with its hand-checked KAT, tests/test_host_and_abi.py): every byte equal; a flat image -> 0."""
    from morphganformer_amd import _lib, drivers
    n, h, w = shape
    torch.manual_seed(h * w)
    img = torch.randn(n, 3, h, w) * torch.tensor([0.5, 1.0, 2.0][:n] if n <= 3 else 1.0).reshape(-1, 1, 1, 1)
    img[0, :, 0, 0] = 0.0
    if n > 1:
        img[1] = 0.25                                                     # a flat candidate: max == min
    x = img.cuda().contiguous()
    gray = torch.empty(n, h, w, dtype=torch.uint8, device="cuda")
    scratch = torch.empty(n * int(_lib.lib().mgf_reference_gray_scratch_floats()), device="cuda")
    _lib.check(_lib.lib().mgf_reference_gray_u8(gray.data_ptr(), x.data_ptr(), n, h, w, scratch.data_ptr(), _lib.stream_ptr()))
    for i in range(n):
        want = drivers.reference_gray_u8(img[i].permute(1, 2, 0).numpy())
        assert np.array_equal(gray[i].cpu().numpy(), want), i
    # the KAT of the numpy statement, through the kernel
    kat = torch.tensor([[-1.0, -1.0, -1.0], [1.0, 1.0, 1.0], [1.0, -1.0, 0.0], [0.0, 0.0, 0.0]]).t().reshape(1, 3, 2, 2).contiguous().cuda()
    g4 = torch.empty(1, 2, 2, dtype=torch.uint8, device="cuda")
    _lib.check(_lib.lib().mgf_reference_gray_u8(g4.data_ptr(), kat.data_ptr(), 1, 2, 2, scratch.data_ptr(), _lib.stream_ptr()))
    assert g4.cpu().reshape(-1).tolist()[:3] == [0, 255, 67]


@pytest.mark.parametrize("use_graph", [True, False])
def test_landmark_callback_on_gray_u8_equals_injected_table(golden, use_graph):
    """landmark_input="gray_u8": the detector gets the drivers' gray uint8 image, built on the device and copied to pinned host memory
    between two captured launch sequences; the run equals the run on the table the callbacks produced, ragged last batch included, and
    a re-targeted engine starts again at step 0."""
    from morphganformer_amd import drivers
    from morphganformer_amd.projection import ProjectionArgs, ProjectionEngine
    g = golden("loop_tiny.npz")
    steps = 10
    seen = []
    G = _tiny_G()

    def detector(gray):
        assert gray.shape == (64, 64) and gray.dtype == np.uint8
        k = len(seen)
        seen.append(gray.copy())
        if k % 10 in (2, 5):
            return None
        return g["lm_target"] + (gray[:34, :4].astype(np.float64) / 8).round().reshape(68, 2)

    def make(**kw):
        return ProjectionEngine(G, torch.from_numpy(g["target"]).cuda(), torch.from_numpy(g["latent_mean"]).cuda(),
                                float(g["latent_std"]), ProjectionArgs(step=steps), use_mse=True, lm_target=g["lm_target"],
                                eps=torch.from_numpy(g["eps"][:steps]).cuda(), noise_mode="const", **kw)
    e1 = make(landmark_fn=detector, landmark_input="gray_u8", batch=4, use_graph=use_graph)
    lat1, st1, loss1, hist1 = e1.run().result()
    assert len(seen) == steps and (e1.cb_graphs is not None) == use_graph
    table, valid = e1.lm_steps[:steps].cpu().numpy(), e1.valid[:steps].cpu().numpy()
    assert valid.tolist() == [1, 1, 0, 1, 1, 0, 1, 1, 1, 1]
    lat2, st2, loss2, hist2 = make(lm_steps=table, lm_valid=valid, batch=1).run().result()
    assert st1 == st2 and torch.equal(lat1, lat2) and np.isnan(hist1[2]) and np.isnan(hist1[5])
    assert np.allclose(np.nan_to_num(hist1), np.nan_to_num(hist2), rtol=1e-6)
    # what the callback saw IS the numpy statement applied to the generated image of that step
    sigma0 = np.float32(np.float32(float(g["latent_std"])) * np.float32(0.05))
    img0 = G((torch.from_numpy(g["latent_mean"])[None] + torch.from_numpy(g["eps"][0]) * float(sigma0)).cuda(), None, noise_mode="const")[0]
    assert np.array_equal(seen[0], drivers.reference_gray_u8(img0[0].permute(1, 2, 0).cpu().numpy()))
    # re-target: same target again -> the same run, callbacks called afresh
    e1.retarget(torch.from_numpy(g["target"]).cuda(), lm_target=g["lm_target"], eps=torch.from_numpy(g["eps"][:steps]).cuda())
    lat3, st3, loss3, hist3 = e1.run().result()
    assert len(seen) == 2 * steps and st3 == st1 and torch.equal(lat3, lat1) and loss3 == loss1


def test_cli_generate_project_morph(tmp_path):
    from morphganformer_amd import cli, drivers
    from morphganformer_amd.projection import synthetic_landmarks
    from test_host_and_abi import _tiny_snapshot
    from PIL import Image
    pkl = str(tmp_path / "net.pkl")
    _tiny_snapshot(pkl, seed=3)
    assert cli.main(["generate", "--model", pkl, "--output-dir", str(tmp_path / "g"), "--images-num", "2", "--seed", "1"]) == 0
    assert sorted(os.listdir(tmp_path / "g")) == ["sample_000000.png", "sample_000001.png"]
    lm_t, lm_s = synthetic_landmarks(6, 64, 1)
    np.savez(tmp_path / "lm.npz", target=lm_t, steps=lm_s)
    import re
    for name in ("a", "b"):
        Image.open(tmp_path / "g" / ("sample_000000.png" if name == "a" else "sample_000001.png")).save(tmp_path / f"{name}.png")
        argv = ["project", "--model", pkl, "--image", str(tmp_path / f"{name}.png"), "--landmarks", str(tmp_path / "lm.npz"),
                "--path_to_gen", str(tmp_path / "p" / name), "--size", "64", "--step", "6", "--n_mean_latent", "200", "--batch", "4", "--seed", "0"]
        with pytest.raises(SystemExit, match="lpips-backbone"):      # the LPIPS term needs real backbone weights unless opted out
            cli.main(argv)
        assert cli.main(argv + ["--lpips-random-backbone"]) == 0
        files = sorted(os.listdir(tmp_path / "p" / name))
        pngs = [f for f in files if f.endswith(".png")]
        assert [f for f in files if f.endswith(".mat")] == [f"{name}.mat"] and len(pngs) >= 1
        assert all(re.fullmatch(r"\d{6}_\d+\.\d{6}\.png", f) for f in pngs), pngs        # {:06d}_{:04f}.png (...sqz_MSE.py:193)
    # a state-dict file of torchvision-keyed weights is accepted in place of the random backbone
    from morphganformer_amd.lpips import random_squeeze_backbone
    np.savez(tmp_path / "sq.npz", **random_squeeze_backbone(0))
    assert cli.main(["project", "--model", pkl, "--image", str(tmp_path / "a.png"), "--path_to_gen", str(tmp_path / "p" / "c"), "--size", "64",
                     "--step", "4", "--n_mean_latent", "200", "--batch", "2", "--seed", "0", "--lpips-backbone", str(tmp_path / "sq.npz")]) == 0
    assert cli.main(["morph", "--model", pkl, "--w1", str(tmp_path / "p" / "a" / "a.mat"), "--w2", str(tmp_path / "p" / "b" / "b.mat"),
                     "--alphas", "0,0.5,1", "--out", str(tmp_path / "m" / "a+b")]) == 0
    assert len(os.listdir(tmp_path / "m")) == 6
    w = drivers.load_latent_mat(str(tmp_path / "m" / "a+b_a0.50.mat"))
    assert np.array_equal(w, 0.5 * drivers.load_latent_mat(str(tmp_path / "p" / "a" / "a.mat")) + 0.5 * drivers.load_latent_mat(str(tmp_path / "p" / "b" / "b.mat")))


def test_project_many_shards_and_gathers(golden):
    """Pair-level sharding entry point (one rank here: every item, ordered by id; the N > 1 gather is covered by the gloo test)."""
    from morphganformer_amd import drivers
    from morphganformer_amd.projection import ProjectionArgs
    g = golden("loop_tiny.npz")
    G = _tiny_G()
    t = torch.from_numpy(g["target"]).cuda()
    targets = [t, (t * 0.9).contiguous(), (t * 0.8).contiguous()]
    kw = dict(args=ProjectionArgs(step=6, n_mean_latent=300), seed=0, batch=3, noise_mode="const")
    res = drivers.project_many(G, targets, **kw)
    assert res["items"].tolist() == [0, 1, 2] and tuple(res["latents"].shape) == (3, 17, 32)
    single = drivers.project_image(G, targets[1], None, None, **kw)
    assert torch.equal(res["latents"][1].cpu(), single["w"][0]) and int(res["steps"][1]) == single["step"]
    dyn = drivers.project_many(G, targets, dynamic=True, **kw)
    assert torch.equal(dyn["latents"], res["latents"]) and torch.equal(dyn["losses"], res["losses"]) and torch.equal(dyn["steps"], res["steps"])
    # a re-targeted engine rewrites ITS copy of the target, never the caller's tensor (it used to alias item 0 and leave the last item's
    # pixels in it: a second pass over the same list then scored item 0 against the wrong image)
    assert torch.equal(targets[0], torch.from_numpy(g["target"]).cuda())


def test_project_many_under_an_rccl_process_group_of_one_rank(golden):
    """First contact with RCCL happens HERE, not in the driver's scaling run: in this process (no re-exec, no child after GPU
    initialisation) a world-size-1 "nccl" process group is created through `distributed.init_process_group` (which makes the TCPStore, hands
    it to torch and keeps it for the work queues), `drivers.project_many(dynamic=True)` runs its three tiny targets through the queue and
    the ragged float64 `all_gather_into_tensor` on DEVICE tensors, `gather_results` gathers one record, and the group is destroyed.  The
    N > 1 control flow is the gloo tests' (tests/test_host_and_abi.py); the job's shape is projection_example_v2_percept_morph.py:329-365."""
    import socket
    import torch.distributed as dist
    from morphganformer_amd import distributed, drivers
    from morphganformer_amd.projection import ProjectionArgs
    assert not dist.is_initialized()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    g = golden("loop_tiny.npz")
    G = _tiny_G()
    t = torch.from_numpy(g["target"]).cuda()
    targets = [t, (t * 0.9).contiguous(), (t * 0.8).contiguous()]
    kw = dict(args=ProjectionArgs(step=6, n_mean_latent=300), seed=0, batch=3, noise_mode="const")
    ref = drivers.project_many(G, targets, **kw)                          # no process group: the plain loop
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    before = distributed.get_store()
    try:
        store = distributed.init_process_group("nccl", rank=0, world_size=1, host="127.0.0.1", port=port,
                                               device_id=torch.device("cuda", torch.cuda.current_device()))
        assert dist.is_initialized() and dist.get_backend() == "nccl" and dist.get_world_size() == 1
        qs = distributed.get_store()                               # the queues' view of the same store (own prefix, restart count in it)
        assert qs is not None and qs.add("mgf_test/counter", 1) == 1 and qs.add("mgf_test/counter", 1) == 2 and store.num_keys() >= 1
        ids = torch.empty(1, dtype=torch.int64, device="cuda")
        dist.all_gather_into_tensor(ids, torch.tensor([7], dtype=torch.int64, device="cuda"))       # one raw RCCL collective
        assert ids.tolist() == [7]
        dyn = drivers.project_many(G, targets, dynamic=True, **kw)
        assert dyn["items"].tolist() == [0, 1, 2] and dyn["mine"] == [0, 1, 2] and dyn["latents"].is_cuda
        assert torch.equal(dyn["latents"], ref["latents"]) and torch.equal(dyn["steps"], ref["steps"]) and torch.equal(dyn["losses"], ref["losses"])
        sta = drivers.project_many(G, targets, **kw)
        assert torch.equal(sta["latents"], ref["latents"])
        one = distributed.gather_results(ref["latents"][1:2].cuda(), float(ref["losses"][1]), int(ref["steps"][1]), item=1)
        assert torch.equal(one["latents"][0], ref["latents"][1]) and one["items"].tolist() == [1] and float(one["losses"][0]) == float(ref["losses"][1])
        dist.barrier()
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()
        distributed._STORE = before
    assert not dist.is_initialized()


def test_bench_force_dist_runs_the_rccl_path():
    """`bench.py --force-dist` in a CHILD process (never an exec from this one): the N > 1 code path of the benchmark with one rank -- the TCPStore
    hand-over through distributed.init_process_group("nccl", device_id=...), the warm-up all_gather, the barriers around the timed region, the
    per-rank time gather and the timed result gather (`gather_results` on device tensors), destroy -- and the JSON line says so (`rccl_ranks` 1,
    `ranks.gather_ms` measured).  What the driver's scaling run exercises first, minus the other ranks."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--force-dist", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--gradient-steps", "0",
           "--targets", "0", "--objectives", "0", "--landmark-callback", "none", "--config4", "0", "--config5-targets", "0", "--bf16x3-leg", "0"]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout.decode()[-2000:]                      # RCCL's banner went to stderr: stdout carries exactly the JSON line
    d = json.loads(lines[0])
    assert d["rccl_ranks"] == 1 and d["n_gpus"] == 1 and d["value"] > 0
    assert d["ranks"]["gather_ms"] is not None and len(d["ranks"]["per_rank_iters_per_s"]) == 1
    assert d["roofline"]["frac"] > 0.3


def test_device_landmark_model_in_the_graph(golden):
    """GPU landmark-regressor interface: a device callable produces the landmarks of every candidate inside the captured launch
    sequence; the run equals the one driven by the table it produced (batch 3 over 8 steps: ragged last batch)."""
    from morphganformer_amd.projection import ProjectionArgs, ProjectionEngine
    g = golden("loop_tiny.npz")
    steps = 8
    base = torch.from_numpy(g["lm_target"]).cuda()

    def model(img):                          # toy regressor: landmarks move with pooled image statistics; a "no face" rule
        pooled = torch.nn.functional.adaptive_avg_pool2d(img[:, :1], (68, 2)).reshape(img.shape[0], 68, 2).double()
        lm = base[None] + torch.round(60 * pooled)
        ok = (img.mean(dim=(1, 2, 3)) > -10).to(torch.int32)
        return lm, ok

    def make(**kw):
        return ProjectionEngine(_tiny_G(), torch.from_numpy(g["target"]).cuda(), torch.from_numpy(g["latent_mean"]).cuda(),
                                float(g["latent_std"]), ProjectionArgs(step=steps), use_mse=True, lm_target=g["lm_target"],
                                eps=torch.from_numpy(g["eps"][:steps]).cuda(), noise_mode="const", **kw)
    e1 = make(landmark_model=model, batch=3, use_graph=True)
    lat1, st1, loss1, hist1 = e1.run().result()
    assert e1.graph is not None and e1.valid.cpu().tolist()[:steps] == [1] * steps
    lat2, st2, loss2, hist2 = make(lm_steps=e1.lm_steps.cpu().numpy()[:steps], batch=1).run().result()
    assert st1 == st2 and torch.equal(lat1, lat2) and np.allclose(hist1, hist2, rtol=1e-6)
    assert float(e1.lm_steps[:steps].std()) > 0


def test_project_image_gradient_mode_beats_literal_sampling(golden, tmp_path):
    """configs[0] shape (256^2, MSE only, the 1024_example_MSE.py optimizer incl. weight_decay=1e-4) through the driver in both
    modes on the same noise stream: descending the gradient must end below the best of the literal loop's noisy samples."""
    from morphganformer_amd import drivers
    from morphganformer_amd.engine import Generator
    from morphganformer_amd.projection import ProjectionArgs
    from morphganformer_amd.synth_weights import SMALL256, make_state_dict
    g = golden("loop_config0_256.npz")
    G = Generator(make_state_dict(SMALL256, seed=0), SMALL256, "cuda", max_batch=1)
    target = torch.from_numpy(g["target_u8"]).float().div(255).sub(0.5).div(0.5)[None].cuda()
    kw = dict(args=ProjectionArgs(step=50, lr=0.05), percept=None, latent_mean=torch.from_numpy(g["latent_mean"]).cuda(),
              latent_std=float(g["latent_std"]), eps=torch.from_numpy(g["eps"]).cuda(), noise_mode="const")
    lit = drivers.project_image(G, target, None, None, batch=5, **kw)
    grad = drivers.project_image(G, target, None, None, mode="gradient", weight_decay=1e-4, out_prefix=str(tmp_path / "g"), **kw)
    assert lit["step"] == int(g["best_step"])
    assert grad["loss"] < 0.9 * lit["loss"], (grad["loss"], lit["loss"])
    assert np.isfinite(grad["losses"]).all() and os.path.exists(tmp_path / "g.mat")


def test_project_many_gradient_lockstep_groups(golden):
    """project_many(mode='gradient', lockstep=2): a rank's items go through one generator pass per step two at a time; every item
    still gets its own record, and an item's result does not depend on which group it travelled in (same noise stream)."""
    from morphganformer_amd import drivers
    from morphganformer_amd.projection import ProjectionArgs
    g = golden("loop_tiny.npz")
    G = _tiny_G()
    t = torch.from_numpy(g["target"]).cuda()
    targets = [t, (t * 0.9).contiguous(), (t * 0.8).contiguous()]
    torch.manual_seed(3)
    mean, std = torch.randn(17, 32, device="cuda"), 1.0
    eps2 = torch.randn(6, 2, 17, 32, device="cuda")
    kw = dict(args=ProjectionArgs(step=6, lr=0.05, lr_rampup=0.3), latent_mean=mean, latent_std=std, noise_mode="const", mode="gradient")
    res = drivers.project_many(G, targets[:2], lockstep=2, eps=eps2, **kw)
    assert res["items"].tolist() == [0, 1] and tuple(res["latents"].shape) == (2, 17, 32)
    for j in range(2):
        single = drivers.project_image(G, targets[j], None, None, eps=eps2[:, j:j + 1].contiguous(), **kw)
        assert int(res["steps"][j]) == single["step"]
        assert abs(float(res["losses"][j]) - single["loss"]) < 1e-4 * abs(single["loss"])
    odd = drivers.project_many(G, targets, lockstep=2, seed=1, **kw)             # 3 items: a group of two and a group of one
    assert odd["items"].tolist() == [0, 1, 2]
    with pytest.raises(ValueError):
        drivers.project_many(G, targets, lockstep=2, args=ProjectionArgs(step=6))


def test_wplus_driver_statistics_never_size_a_synthesis_workspace(golden):
    """project_image(mode='gradient', latent_space='w+') WITHOUT latent_mean: the w-space statistics pass (projection.latent_stats_w over
    n_mean_latent mapped samples) runs the mapping network alone -- no generator workspace larger than the engine's batch is ever created
    (round 3 sized the synthesis workspace for 2000 samples: 3.3 TB at 1024^2) -- and equals G.mapping on the same z; project_many carries
    W+ records ([k, num_ws, D] per item) through the gather, one by one and in lockstep groups."""
    from morphganformer_amd import drivers
    from morphganformer_amd.projection import ProjectionArgs, latent_stats_w, mapping_only
    g = golden("loop_tiny.npz")
    G = _tiny_G()
    cfg = G.cfg
    t = torch.from_numpy(g["target"]).cuda()
    z = torch.randn(7, cfg.k, cfg.z_dim, device="cuda")
    assert torch.equal(mapping_only(G, z), G.mapping(z)[:, :, 0])
    G2 = _tiny_G()
    gen = torch.Generator(device="cuda"); gen.manual_seed(4)
    mean, std = latent_stats_w(G2, 4000, "cuda", gen)
    assert tuple(mean.shape) == (cfg.k, cfg.w_dim) and float(std) > 0 and max(k[0] for k in G2._workspaces) == 1
    args = ProjectionArgs(step=5, lr=0.05, lr_rampup=0.3, n_mean_latent=3000)
    r = drivers.project_image(G2, t, None, None, args=args, mode="gradient", latent_space="w+", noise_mode="const", seed=2)
    assert tuple(r["w"].shape) == (1, cfg.k, cfg.num_ws, cfg.w_dim) and np.isfinite(r["losses"]).all()
    assert max(k[0] for k in G2._workspaces) == 1, sorted(G2._workspaces)
    kw = dict(args=args, mode="gradient", latent_space="w+", noise_mode="const", seed=2)
    many = drivers.project_many(G2, [t, (t * 0.9).contiguous()], **kw)
    assert tuple(many["latents"].shape) == (2, cfg.k, cfg.num_ws, cfg.w_dim)
    assert torch.equal(many["latents"][0].cpu(), r["w"][0])
    grp = drivers.project_many(G2, [t, (t * 0.9).contiguous()], lockstep=2, **kw)
    assert tuple(grp["latents"].shape) == (2, cfg.k, cfg.num_ws, cfg.w_dim) and max(k[0] for k in G2._workspaces) == 2
    with pytest.raises(ValueError):
        drivers.project_many(G2, [t], latent_space="w+", args=args)


def test_engine_reuse_checks_the_objective(golden):
    """project_image(engine=...) re-targets a captured launch sequence: an engine built for another objective (no Wing term, another
    percept / noise mode / ProjectionArgs) is refused instead of silently scoring the first item's objective; project_many builds a
    fresh engine when an item with landmarks follows one without."""
    from morphganformer_amd import drivers
    from morphganformer_amd.projection import ProjectionArgs
    g = golden("loop_tiny.npz")
    G = _tiny_G()
    t = torch.from_numpy(g["target"]).cuda()
    steps = 6
    args = ProjectionArgs(step=steps)
    kw = dict(args=args, percept=None, latent_mean=torch.from_numpy(g["latent_mean"]).cuda(), latent_std=float(g["latent_std"]),
              noise_mode="const", batch=2, seed=1)
    first = drivers.project_image(G, t, None, None, return_engine=True, **kw)
    eng = first["engine"]
    with pytest.raises(ValueError, match="Wing"):
        drivers.project_image(G, t, g["lm_target"], g["lm_steps"][:steps], engine=eng, **kw)
    with pytest.raises(ValueError, match="args"):
        drivers.project_image(G, t, None, None, engine=eng, **dict(kw, args=ProjectionArgs(step=steps, beta=0.5)))
    with pytest.raises(ValueError, match="noise_mode"):
        drivers.project_image(G, t, None, None, engine=eng, **dict(kw, noise_mode="none"))
    again = drivers.project_image(G, t, None, None, engine=eng, **kw)
    assert again["step"] == first["step"] and again["loss"] == first["loss"]
    # mixed items: without landmarks, then with -- the second item's Wing term must be in ITS total
    lms = [(None, None), (g["lm_target"], g["lm_steps"][:steps])]
    many = drivers.project_many(G, [t, t], landmarks=lms, **kw)
    alone = drivers.project_image(G, t, *lms[1], **kw)
    assert float(many["losses"][0]) == first["loss"] and float(many["losses"][1]) == alone["loss"] and alone["loss"] != first["loss"]


def test_gradient_entry_points_reject_bad_arguments():
    """The new C-ABI entry points validate on the host and report through mgf_last_error (no launch on bad input)."""
    from morphganformer_amd import _lib
    L = _lib.lib()
    x = torch.zeros(64, device="cuda")
    with pytest.raises(_lib.MgfError, match="alpha and gain"):
        _lib.check(L.mgf_layer_act_bwd_f32(x.data_ptr(), None, x.data_ptr(), x.data_ptr(), None, None, None, None, 1, 1, 1, 64, 0.0, 1.0, None))
    with pytest.raises(_lib.MgfError, match="latent components"):
        _lib.check(L.mgf_duplex_attention_bwd(x.data_ptr(), None, None, x.data_ptr(), x.data_ptr(), x.data_ptr(), x.data_ptr(), x.data_ptr(),
                                              1, 2, 2, 17, None))
    with pytest.raises(_lib.MgfError, match="wdim must divide 256"):
        _lib.check(L.mgf_style_demod_bwd_multi(x.data_ptr(), x.data_ptr(), 1, 1, 48, 64, None))
    with pytest.raises(_lib.MgfError, match="bad split"):
        _lib.check(L.mgf_relu_bwd_split_f32(x.data_ptr(), None, x.data_ptr(), x.data_ptr(), 1, 4, 2, 16, None))
    with pytest.raises(_lib.MgfError, match="output extent"):
        _lib.check(L.mgf_maxpool3x3s2_ceil_bwd_f32(x.data_ptr(), x.data_ptr(), x.data_ptr(), 1, 8, 8, 5, 4, None))
    with pytest.raises(_lib.MgfError, match="null pointer"):
        _lib.check(L.mgf_adam_step_f32(x.data_ptr(), x.data_ptr(), x.data_ptr(), None, x.data_ptr(), x.data_ptr(), x.data_ptr(), None, 4, 2,
                                       0.9, 0.999, 1e-8, 0.0, None))
    with pytest.raises(_lib.MgfError, match="at most 16 rows"):
        _lib.check(L.mgf_linear_bwd_f32(x.data_ptr(), x.data_ptr(), x.data_ptr(), 17, 4, 4, None))


def _warp_points(rng, side, n_inner, jitter):
    edge = side - 1
    third = [0, edge // 3, 2 * edge // 3, edge]
    frame = np.array([[0, v] for v in third] + [[v, 0] for v in third[1:]] + [[edge, v] for v in third[1:]] + [[v, edge] for v in third[1:3]], np.float64)
    lm1 = rng.integers(side // 8, side - side // 8, (n_inner, 2)).astype(np.float64)
    lm2 = lm1 + rng.integers(-jitter, jitter + 1, (n_inner, 2))
    inner_avg = (lm1 + lm2) / 2                                          # averaged integer detections: halves, like torch.div(l1 + l2, 2)
    inner_G = lm1 + rng.integers(-jitter, jitter + 1, (n_inner, 2))      # dlib detections on the morph: integers
    return np.concatenate([inner_G, frame]), np.concatenate([inner_avg, frame])


@pytest.mark.parametrize("side,n_inner,jitter", [(96, 20, 4), (1024, 68, 12)])
def test_landmark_delaunay_warp_is_byte_exact_vs_the_opencv_restatement(side, n_inner, jitter):
    """1024_warp_morphs.py:78-113,163-210 -- per triangle cv2.warpAffine(INTER_LINEAR, BORDER_REFLECT_101) of the
    bounding-box patch pasted through cv2.fillConvexPoly -- as a host label map + one gather kernel, against oracle/warp_ref.py's literal
    transcription of the OpenCV sources.  On the uint8 image the script reads back from its PNG every step is exact: imgMorph float values and
    the written bytes are IDENTICAL (integer work: the bar is bit-exact).  Parity with the real OpenCV is unpinned (cv2 absent offline)."""
    from morphganformer_amd import drivers
    from oracle import warp_ref as W
    rng = np.random.Generator(np.random.PCG64(side))
    yy, xx = np.mgrid[0:side, 0:side].astype(np.float64)
    img_u8 = np.stack([127 + 100 * np.sin(xx / (7 + c) + c) * np.cos(yy / 9 - c) + rng.integers(-20, 20, (side, side)) for c in range(3)],
                      axis=-1).clip(0, 255).astype(np.uint8)
    p_G, p_avg = _warp_points(rng, side, n_inner, jitter)
    label, recs, simp = drivers.warp_plan(p_G, p_avg, side, side)
    ref_fn = W.warp_morph_ref if side <= 128 else W.warp_morph_ref_rows
    ref = ref_fn(img_u8.astype(np.float32), [tuple(p) for p in p_G], [tuple(p) for p in p_avg], simp)
    src = torch.from_numpy(img_u8.transpose(2, 0, 1).copy()).float().cuda()
    out = drivers.warp_morph(src, p_G, p_avg)
    assert tuple(out.shape) == (3, side, side)
    assert np.array_equal(out.cpu().numpy().transpose(1, 2, 0), ref)                       # imgMorph, float, bit for bit
    got_u8 = drivers.warp_morph_u8(img_u8, p_G, p_avg)
    assert got_u8.dtype == np.uint8 and np.array_equal(got_u8, np.uint8(ref))              # the bytes cv2.imwrite gets
    assert (label >= 0).all() and len(recs) == len(simp)                                   # the frame points span the image: every pixel is written
    assert np.abs(got_u8.astype(np.int32) - img_u8.astype(np.int32)).max() > 5             # ... and it did move something
    # a float image in another value range takes the same path (products no longer exact: one rounding per operation, same order)
    f = (src / 127.5 - 1.0).contiguous()
    ref_f = ref_fn(f.cpu().numpy().transpose(1, 2, 0), [tuple(p) for p in p_G], [tuple(p) for p in p_avg], simp)
    assert np.array_equal(drivers.warp_morph(f, p_G, p_avg, background=0.0).cpu().numpy().transpose(1, 2, 0), ref_f)
    assert len(drivers.WARP_EXTRA_POINTS) == 12


def test_warp_morphs_driver_mirrors_the_script(tmp_path):
    """1024_warp_morphs.py:128-210 end to end on the tiny generator: 0.5 / 0.5 latent morph -> bytes -> landmarks of the two sources averaged and
    triangulated with the frame points -> the morph's own landmarks (injected, or a detector called on the drivers' gray uint8 image) -> every
    triangle warped -> Morph_final.png.  Against the literal OpenCV transcription applied to the same bytes: byte-exact."""
    from scipy.spatial import Delaunay
    from morphganformer_amd import cli, drivers
    from morphganformer_amd.engine import Generator
    from morphganformer_amd.synth_weights import TINY, make_state_dict
    from oracle import warp_ref as W
    from test_host_and_abi import _tiny_snapshot
    from PIL import Image
    assert drivers.frame_points(1024) == drivers.WARP_EXTRA_POINTS
    G = Generator(make_state_dict(TINY, seed=0), TINY, "cuda", max_batch=1)
    S = TINY.img_resolution
    rng = np.random.Generator(np.random.PCG64(77))
    w1, w2 = (rng.standard_normal((1, TINY.k, TINY.z_dim)).astype(np.float32) for _ in range(2))
    base = np.stack([rng.integers(8, S - 8, 68), rng.integers(8, S - 8, 68)], axis=1)
    base = np.unique(base, axis=0)
    lm1, lm2 = base + rng.integers(-2, 3, base.shape), base + rng.integers(-2, 3, base.shape)
    lm_G = base + rng.integers(-3, 4, base.shape)
    r = drivers.warp_morphs(G, w1, w2, lm1, lm2, landmark_G=lm_G, out_dir=str(tmp_path / "o"), noise_mode="const")
    assert np.array_equal(r["latent"], 0.5 * w1 + 0.5 * w2)
    img = G(torch.from_numpy(r["latent"]).cuda(), 0.7, noise_mode="const")[0]            # (the script's call: the 0.7 lands on `c`)
    want_u8 = np.clip(np.rint((img[0].permute(1, 2, 0).cpu().numpy().astype(np.float64) + 1) * 127.5), 0, 255).astype(np.uint8)
    assert np.array_equal(r["morph"], want_u8)
    extra = np.asarray(drivers.frame_points(S), np.float64)
    p_avg = np.concatenate([(lm1.astype(np.float64) + lm2) / 2, extra])
    p_G = np.concatenate([lm_G.astype(np.float64), extra])
    assert np.array_equal(r["points_avg"], p_avg) and np.array_equal(r["points_G"], p_G)
    simp = Delaunay(p_avg).simplices.tolist()
    ref = W.warp_morph_ref(r["morph"].astype(np.float32), [tuple(p) for p in p_G], [tuple(p) for p in p_avg], simp)
    assert np.array_equal(r["warped"], np.uint8(ref))
    assert np.array_equal(np.asarray(Image.open(tmp_path / "o" / "Morph_final.png")), r["warped"])
    assert np.array_equal(np.asarray(Image.open(tmp_path / "o" / "morph_G.png")), r["morph"])
    # the detector variant: it sees the drivers' gray uint8 image of the morph (get_landmarks_G, :61-66)
    seen = []

    def detector(gray):
        seen.append(gray.copy())
        return lm_G

    r2 = drivers.warp_morphs(G, w1, w2, lm1, lm2, landmark_fn=detector, noise_mode="const")
    assert np.array_equal(r2["warped"], r["warped"])
    assert np.array_equal(seen[0], drivers.reference_gray_u8(img[0].permute(1, 2, 0).cpu().numpy()))
    with pytest.raises(_lib_error()):
        drivers.warp_morphs(G, w1, w2, lm1, lm2, landmark_fn=lambda gray: None, noise_mode="const")
    with pytest.raises(ValueError):
        drivers.warp_morphs(G, w1, w2, lm1, lm2[:10], landmark_G=lm_G, noise_mode="const")
    # the command line
    pkl = str(tmp_path / "net.pkl")
    _tiny_snapshot(pkl, seed=3)
    drivers.save_latent_mat(str(tmp_path / "a.mat"), w1)
    drivers.save_latent_mat(str(tmp_path / "b.mat"), w2)
    np.savez(tmp_path / "lm.npz", lm1=lm1, lm2=lm2, lm_G=lm_G)
    assert cli.main(["warp", "--model", pkl, "--w1", str(tmp_path / "a.mat"), "--w2", str(tmp_path / "b.mat"), "--landmarks", str(tmp_path / "lm.npz"),
                     "--out", str(tmp_path / "w")]) == 0
    assert sorted(os.listdir(tmp_path / "w")) == ["Morph_final.png", "morph_G.png"]
    assert np.asarray(Image.open(tmp_path / "w" / "Morph_final.png")).shape == (S, S, 3)


def _lib_error():
    from morphganformer_amd import _lib
    return _lib.MgfError


def test_merge_morph_tree_walks_the_directories_like_the_script(tmp_path):
    """1024_merge_morph_2.py:53-92: ids -> the two latent folders -> every pair of `.mat` files -> `<stem1>+<stem2>.jpg / .mat`; existing
    images are skipped; the blended latent is `0.5 * w1 + 0.5 * w2` on the loadmat arrays."""
    from morphganformer_amd import cli, drivers
    from morphganformer_amd.engine import Generator
    from morphganformer_amd.synth_weights import TINY, make_state_dict
    from test_host_and_abi import _tiny_snapshot
    from PIL import Image
    G = Generator(make_state_dict(TINY, seed=0), TINY, "cuda", max_batch=1)
    rng = np.random.Generator(np.random.PCG64(5))
    src, lat = tmp_path / "src", {}
    for ident in ("0001", "0002"):
        for name, files in (("personA", ["000003_0.812345", "000007_0.701234"]), ("personB", ["000001_0.912345"])):
            os.makedirs(src / ident / name)
            for f in files:
                lat[(ident, name, f)] = rng.standard_normal((1, TINY.k, TINY.z_dim)).astype(np.float32)
                drivers.save_latent_mat(str(src / ident / name / (f + ".mat")), lat[(ident, name, f)])
                Image.new("RGB", (8, 8)).save(src / ident / name / (f + ".png"))          # the projections' images lie beside the latents
        (src / ident / "notes.txt").write_text("x")                                        # entries with a dot are not latent folders
    done = drivers.merge_morph_tree(G, str(src), str(tmp_path / "dst"), noise_mode="const")
    assert sorted(done) == [os.path.join(i, s) for i in ("0001", "0002") for s in ("000003_0.812345+000001_0.912345", "000007_0.701234+000001_0.912345")]
    for ident in ("0001", "0002"):
        assert sorted(os.listdir(tmp_path / "dst" / ident)) == sorted(s + e for s in ("000003_0.812345+000001_0.912345", "000007_0.701234+000001_0.912345")
                                                                       for e in (".jpg", ".mat"))
        w = drivers.load_latent_mat(str(tmp_path / "dst" / ident / "000003_0.812345+000001_0.912345.mat"))
        assert np.array_equal(w, 0.5 * lat[(ident, "personA", "000003_0.812345")] + 0.5 * lat[(ident, "personB", "000001_0.912345")])
        assert Image.open(tmp_path / "dst" / ident / "000003_0.812345+000001_0.912345.jpg").size == (TINY.img_resolution, TINY.img_resolution)
    assert drivers.merge_morph_tree(G, str(src), str(tmp_path / "dst"), noise_mode="const") == []      # everything exists: nothing is redone
    pkl = str(tmp_path / "net.pkl")
    _tiny_snapshot(pkl, seed=3)
    assert cli.main(["morph-tree", "--model", pkl, "--src", str(src), "--dst", str(tmp_path / "dst2")]) == 0
    assert len(os.listdir(tmp_path / "dst2" / "0002")) == 4


def test_morph_pairs_projects_renders_and_skips_like_the_script(tmp_path):
    """projection_example_v2_percept_morph.py:330-365 (BASELINE config 3's outer loop): CSV rows -> both images projected -> `_A.png`, `_B.png`,
    the 0.5 / 0.5 morph; header and low-similarity rows skipped, existing morphs skipped; the morph IS G(0.5 w1 + 0.5 w2) of the returned latents."""
    from morphganformer_amd import drivers
    from morphganformer_amd.engine import Generator
    from morphganformer_amd.projection import ProjectionArgs
    from morphganformer_amd.synth_weights import TINY, make_state_dict
    from PIL import Image
    G = Generator(make_state_dict(TINY, seed=0), TINY, "cuda", max_batch=1)
    drivers.generate_images(G, 3, output_dir=str(tmp_path / "src"), seed=4, noise_mode="const")
    for i, nm in enumerate(("anna.png", "ben.png", "cleo.png")):
        os.rename(tmp_path / "src" / f"sample_{i:06d}.png", tmp_path / "src" / nm)
    (tmp_path / "pairs.csv").write_text("img1,img2,simi\nanna.png,ben.png,0.81\nanna.png,cleo.png,0.31\nben.png,cleo.png,0.5\n")
    pairs = drivers.read_pair_csv(str(tmp_path / "pairs.csv"))
    assert pairs == [("anna.png", "ben.png"), ("ben.png", "cleo.png")]                      # header and the 0.31 row are gone (`re < 0.5`)
    kw = dict(args=ProjectionArgs(step=8, n_mean_latent=200), percept=None, batch=4, seed=2, noise_mode="const")
    r = drivers.morph_pairs(G, pairs, str(tmp_path / "src"), str(tmp_path / "raw"), str(tmp_path / "morph"), **kw)
    assert sorted(os.listdir(tmp_path / "morph")) == ["anna_ben.png", "ben_cleo.png"]
    assert sorted(os.listdir(tmp_path / "raw")) == ["anna_ben_A.png", "anna_ben_B.png", "ben_cleo_A.png", "ben_cleo_B.png"]
    assert tuple(r["latents"].shape) == (4, TINY.k, TINY.z_dim) and len(r["written"]) == 2
    w = 0.5 * r["latents"][0:1].cpu().numpy() + 0.5 * r["latents"][1:2].cpu().numpy()
    want = drivers.to_uint8_image(G, G(torch.from_numpy(w).cuda(), 0.7, noise_mode="const")[0])
    assert np.array_equal(np.asarray(Image.open(tmp_path / "morph" / "anna_ben.png")), want)
    want_b = drivers.to_uint8_image(G, G(r["latents"][1:2].cuda(), 0.7, noise_mode="const")[0])
    assert np.array_equal(np.asarray(Image.open(tmp_path / "raw" / "anna_ben_B.png")), want_b)
    # everything exists: nothing is projected again; one morph removed: only that pair is redone
    assert drivers.morph_pairs(G, pairs, str(tmp_path / "src"), str(tmp_path / "raw"), str(tmp_path / "morph"), **kw)["pairs"] == []
    os.remove(tmp_path / "morph" / "ben_cleo.png")
    r2 = drivers.morph_pairs(G, pairs, str(tmp_path / "src"), str(tmp_path / "raw"), str(tmp_path / "morph"), **kw)
    assert r2["pairs"] == [("ben.png", "cleo.png")] and tuple(r2["latents"].shape) == (2, TINY.k, TINY.z_dim)
    assert torch.equal(r2["latents"], r["latents"][2:4])                                  # same seed, same targets: the same projections


def test_morph_pairs_gradient_mode_w_plus(tmp_path):
    """morph_pairs(mode="gradient", latent_space="w+") used to run every projection and then die in the render step (a 4-D
    latent passed on as z).  W+ results [1,k,num_ws,D] render through G(ws=...), blend slot by slot, and round-trip through the .mat format."""
    from morphganformer_amd import drivers
    from morphganformer_amd.engine import Generator
    from morphganformer_amd.projection import ProjectionArgs
    from morphganformer_amd.synth_weights import TINY, make_state_dict
    from PIL import Image
    G = Generator(make_state_dict(TINY, seed=0), TINY, "cuda", max_batch=1)
    drivers.generate_images(G, 2, output_dir=str(tmp_path / "src"), seed=4, noise_mode="const")
    pairs = [("sample_000000.png", "sample_000001.png")]
    kw = dict(args=ProjectionArgs(step=6, n_mean_latent=200, lr=0.05, min_loss_init=1e30), percept=None, seed=2, noise_mode="const", mode="gradient",
              latent_space="w+")
    r = drivers.morph_pairs(G, pairs, str(tmp_path / "src"), str(tmp_path / "raw"), str(tmp_path / "morph"), **kw)
    assert tuple(r["latents"].shape) == (2, TINY.k, TINY.num_ws, TINY.w_dim)
    assert sorted(os.listdir(tmp_path / "raw")) == ["sample_000000_sample_000001_A.png", "sample_000000_sample_000001_B.png"]
    w1, w2 = r["latents"][0:1], r["latents"][1:2]
    want = drivers.to_uint8_image(G, G(ws=(0.5 * w1 + 0.5 * w2).cuda(), noise_mode="const")[0])
    assert np.array_equal(np.asarray(Image.open(tmp_path / "morph" / "sample_000000_sample_000001.png")), want)
    want_a = drivers.to_uint8_image(G, G(ws=w1.cuda(), noise_mode="const")[0])
    assert np.array_equal(np.asarray(Image.open(tmp_path / "raw" / "sample_000000_sample_000001_A.png")), want_a)
    lat, imgs = drivers.merge_morph(G, w1, w2, (0.0, 0.5), noise_mode="const", out_prefix=str(tmp_path / "m" / "ab"))
    assert lat.shape == (2, 1, TINY.k, TINY.num_ws, TINY.w_dim) and torch.equal(imgs[0], G(ws=w1.cuda(), noise_mode="const")[0][0])
    assert np.array_equal(drivers.load_latent_mat(str(tmp_path / "m" / "ab_a0.50.mat")), lat[1])
    with pytest.raises(ValueError, match="differ in shape"):
        drivers.merge_morph(G, w1, w2[:, :, 0], (0.5,))


def test_misc_to_pil_and_crop_under_the_reference_names():
    """morphganformer_amd.misc.to_pil / crop_max_rectangle (misc.py:94-130) -- the calls `crop(misc.to_pil(img_gen_raw[0]), ratio)` of the drivers
    -- on a numpy CHW image as the drivers pass it and on a device tensor: the bytes equal the reference's float32 adjust_range + rint + clip."""
    from morphganformer_amd import misc
    rng = np.random.Generator(np.random.PCG64(5))
    img = (rng.standard_normal((3, 37, 52)) * 0.8).astype(np.float32)
    img[0, 0, :6] = [-1.0, 1.0, -1.00001, 1.00001, 0.0, -0.0039215689]       # range ends, just outside, a .5 tie candidate
    scale = (np.float32(255) - np.float32(0)) / (np.float32(1) - np.float32(-1))
    bias = np.float32(0) - np.float32(-1) * scale
    want = np.rint(img.transpose(1, 2, 0) * scale + bias).clip(0, 255).astype(np.uint8)     # misc.py:103-124
    im = misc.to_pil(img)
    assert im.mode == "RGB" and im.size == (52, 37) and np.array_equal(np.asarray(im), want)
    assert np.array_equal(np.asarray(misc.to_pil(torch.from_numpy(img).cuda())), want)
    gray = misc.to_pil(img[:1])
    assert gray.mode == "L" and np.array_equal(np.asarray(gray), want[:, :, 0])
    assert misc.crop_max_rectangle(im, None) is im
    assert misc.crop_max_rectangle(im, 1.0).size == (37, 37) and misc.crop_max_rectangle(im, 0.5).size == (52, 26)
    with pytest.raises(ValueError, match="drange"):
        misc.to_pil(img, drange=[0, 1])


def test_cli_morph_pairs(tmp_path):
    from morphganformer_amd import cli
    from test_host_and_abi import _tiny_snapshot
    pkl = str(tmp_path / "net.pkl")
    _tiny_snapshot(pkl, seed=3)
    assert cli.main(["generate", "--model", pkl, "--output-dir", str(tmp_path / "src"), "--images-num", "2", "--seed", "1"]) == 0
    (tmp_path / "pairs.csv").write_text("img1,img2,simi\nsample_000000.png,sample_000001.png,0.9\n")
    argv = ["morph-pairs", "--model", pkl, "--csv", str(tmp_path / "pairs.csv"), "--src", str(tmp_path / "src"), "--dst-raw", str(tmp_path / "raw"),
            "--dst-morph", str(tmp_path / "morph"), "--size", "64", "--step", "6", "--n_mean_latent", "200", "--batch", "3", "--seed", "0"]
    with pytest.raises(SystemExit, match="lpips-backbone"):
        cli.main(argv)
    assert cli.main(argv + ["--no-lpips"]) == 0
    assert os.listdir(tmp_path / "morph") == ["sample_000000_sample_000001.png"]
    assert sorted(os.listdir(tmp_path / "raw")) == ["sample_000000_sample_000001_A.png", "sample_000000_sample_000001_B.png"]
    assert cli.main(argv + ["--no-lpips", "--mode", "gradient"]) == 0              # everything exists: nothing to do
    # the biometric term from the command line (1024_example_FaceNet_percept.py's objective is this term alone: --no-lpips --no-mse)
    proj = ["project", "--model", pkl, "--image", str(tmp_path / "src" / "sample_000000.png"), "--path_to_gen", str(tmp_path / "b"), "--size", "64",
            "--step", "4", "--n_mean_latent", "200", "--batch", "2", "--seed", "0", "--no-lpips"]
    with pytest.raises(SystemExit, match="biometric-weights"):
        cli.main(proj + ["--biometric", "iresnet18"])
    assert cli.main(proj + ["--biometric", "iresnet18", "--biometric-random", "--gamma", "1e-12"]) == 0
    assert cli.main(proj + ["--biometric", "iresnet18", "--biometric-random", "--gamma", "1e-12", "--no-mse", "--path_to_gen", str(tmp_path / "b2")]) == 0
    with pytest.raises(SystemExit, match="switched off"):
        cli.main(proj + ["--no-mse"])


@pytest.mark.gpu
def test_facenet_feature_and_its_cli_verb(tmp_path):
    """`facenet_feature` (extract_FaceNet.py:30-40): cv2.resize to 224 x 224, clamp, (x - 127.5) / 128, InceptionResnetV1, flatten -- the driver
    against the oracle's network on the same resized input (seeded weights: parity with facenet_pytorch itself is unpinned), and the
    `extract-facenet` verb writing the same numbers for two files."""
    from PIL import Image
    import scipy.io as sio
    from morphganformer_amd import cli, drivers
    from morphganformer_amd.facenet import InceptionResnetV1Embedder, random_state
    from oracle.embed_ref import inception_resnet_v1_ref
    rng = np.random.default_rng(3)
    yy, xx = np.mgrid[0:300, 0:280].astype(np.float32)
    imgs = []
    for k in range(2):
        base = np.stack([np.sin(xx / (17 + k)) + np.cos(yy / 23), np.sin((xx + yy) / 31), np.cos(xx / 11) * np.sin(yy / (13 + k))], -1)
        imgs.append(np.clip(127.5 + 60 * base + rng.normal(0, 6, base.shape), 0, 255).astype(np.uint8))
    sd = random_state(0)
    net = InceptionResnetV1Embedder(sd, n=1)
    tsd = {k: torch.from_numpy(v) for k, v in sd.items()}
    feats = []
    for im in imgs:
        got = drivers.facenet_feature(im, net)
        small = drivers.cv_resize_linear_u8(im, 224, 224)
        x = ((torch.from_numpy(small.astype(np.float32)) - 127.5) / 128.0).permute(2, 0, 1)[None]
        with torch.no_grad():
            want = inception_resnet_v1_ref(tsd, x).numpy().reshape(-1)
        assert got.shape == (512,) and np.abs(got - want).max() < 1e-3 * np.abs(want).max()
        feats.append(got)
    assert np.abs(feats[0] - feats[1]).max() > 1e-4                 # two different faces, two different features
    files = []
    for k, im in enumerate(imgs):
        files.append(str(tmp_path / f"f{k}.png"))
        Image.fromarray(im, "RGB").save(files[-1])
    with pytest.raises(SystemExit):
        cli.main(["extract-facenet", *files, "--out", str(tmp_path / "f.mat")])
    assert cli.main(["extract-facenet", *files, "--out", str(tmp_path / "f.mat"), "--biometric-random"]) == 0
    m = sio.loadmat(str(tmp_path / "f.mat"))
    assert m["features"].shape == (2, 512) and np.abs(m["features"] - np.stack(feats)).max() < 1e-6
    assert cli.main(["extract-facenet", files[1], "--out", str(tmp_path / "f.npy"), "--biometric-random"]) == 0
    assert np.abs(np.load(str(tmp_path / "f.npy"))[0] - feats[1]).max() < 1e-6


@pytest.mark.gpu
def test_cli_project_with_the_lbp_objective(tmp_path):
    """`project --pixel-term lbp --no-lpips --no-mse`: 1024_example_LBP_percept.py as a command -- the target file's LBP feature is taken from the
    file's own pixels, the loop keeps the smallest distance, the latent and the improvement images are written under the scripts' names."""
    from morphganformer_amd import cli, drivers
    from test_host_and_abi import _tiny_snapshot
    pkl = str(tmp_path / "net.pkl")
    _tiny_snapshot(pkl, seed=3)
    assert cli.main(["generate", "--model", pkl, "--output-dir", str(tmp_path / "g"), "--images-num", "1", "--seed", "2"]) == 0
    img = str(tmp_path / "g" / "sample_000000.png")
    argv = ["project", "--model", pkl, "--image", img, "--path_to_gen", str(tmp_path / "p"), "--size", "64", "--step", "6", "--n_mean_latent", "200",
            "--batch", "4", "--seed", "0", "--no-lpips", "--no-mse", "--pixel-term", "lbp"]
    assert cli.main(argv) == 0
    files = sorted(os.listdir(tmp_path / "p"))
    assert "sample_000000.mat" in files and any(f.endswith(".png") for f in files)
    assert drivers.load_latent_mat(str(tmp_path / "p" / "sample_000000.mat")).shape[-2:] == (17, 32)
    with pytest.raises(SystemExit):
        cli.main(argv + ["--mode", "gradient"])
