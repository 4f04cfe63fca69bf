"""GPU parity of the HIP generator engine against (a) the reference module's own outputs on seeded synthetic weights
(tests/golden/gen_tiny.npz, gen_full1024.npz -- produced by oracle/make_golden.py from /root/reference) and (b) the CPU
oracle on fresh seeded inputs.  Gate (BASELINE.json north_star): generated pixels within 1e-3 of the fp32 CPU path,
relative to max|img|; per-pixel latent assignment (argmax over the 16 components) exact wherever the reference's top-2
probabilities are separated by more than float32 re-association noise.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

PIX_TOL = 1e-3


def rel(a, b):
    a = a.detach().double().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / np.abs(b).max())


@pytest.fixture(scope="module")
def tiny():
    from morphganformer_amd.engine import Generator
    from morphganformer_amd.synth_weights import TINY, make_state_dict
    sd = make_state_dict(TINY, seed=0)
    return Generator(sd, TINY, "cuda", max_batch=2), sd, TINY


def test_mapping_matches_reference(tiny, golden):
    G, sd, cfg = tiny
    g = golden("gen_tiny.npz")
    ws = G(torch.from_numpy(g["z"]).cuda(), None, subnet="mapping")
    assert tuple(ws.shape) == (2, cfg.k, cfg.num_ws, cfg.w_dim)
    assert rel(ws[:, :, 0], g["ws"]) < 1e-5


@pytest.mark.parametrize("mode", ["const", "none", "inject"])
def test_tiny_generator_matches_reference(tiny, golden, mode):
    G, sd, cfg = tiny
    g = golden("gen_tiny.npz")
    z = torch.from_numpy(g["z"]).cuda()
    noises = None
    if mode == "inject":
        noises = {k[len("noise_"):]: torch.from_numpy(g[k]).cuda() for k in g.files if k.startswith("noise_")}
    G.taps = {}
    img = G(z, None, noise_mode=mode, noises=noises)[0]
    taps, G.taps = G.taps, None
    assert tuple(img.shape) == (2, 3, 64, 64)
    if mode == "const":
        for res in cfg.block_resolutions:
            assert rel(taps[f"synthesis.b{res}"], g[f"tap_b{res}"]) < PIX_TOL, res
    assert rel(img, g["img_" + mode]) < PIX_TOL


def test_tiny_attention_assignments(tiny, golden):
    G, sd, cfg = tiny
    g = golden("gen_tiny.npz")
    z = torch.from_numpy(g["z"]).cuda()
    img, att = G(z, None, noise_mode="const", return_att=True, att_format="maps")
    for key in ("b4.conv1", "b16.conv0", "b64.conv1"):
        probs, argmax = att["synthesis." + key]
        ref = g["probs_" + key]                      # [n, F, T]
        assert np.abs(probs.cpu().numpy() - ref).max() < 1e-4, key
        top2 = np.sort(ref, axis=-1)[..., -2:]
        decided = (top2[..., 1] - top2[..., 0]) > 1e-4
        assert decided.mean() > 0.99
        assert np.array_equal(argmax.cpu().numpy()[decided], ref.argmax(-1)[decided]), key


def test_generator_to_refuses_to_be_ignored(tiny):
    """`G = load_network(...)["Gs"].to(device)` (...sqz_MSE.py:249): the device the generator lives on is accepted in every spelling, any other
    device, dtype or a request for trainable weights RAISES instead of being silently ignored."""
    from morphganformer_amd import _lib
    G, sd, cfg = tiny
    cur = torch.cuda.current_device()
    for dev_ in ("cuda", f"cuda:{cur}", torch.device("cuda", cur), cur):
        assert G.to(dev_) is G
    assert G.to(device=f"cuda:{cur}", dtype=torch.float32) is G and G.to(torch.float32) is G and G.to(torch.zeros(1, device="cuda")) is G
    assert G.eval() is G and G.requires_grad_(False) is G and G.train(False) is G
    for bad in ("cpu", f"cuda:{cur + 3}", torch.device("cuda", cur + 1), cur + 1):
        with pytest.raises(_lib.MgfError, match="cannot move"):
            G.to(bad)
    with pytest.raises(_lib.MgfError, match="float32"):
        G.to(torch.float16)
    with pytest.raises(_lib.MgfError):
        G.requires_grad_(True)
    with pytest.raises(_lib.MgfError):
        G.train()


def test_tiny_vs_oracle_fresh_latents(tiny):
    from oracle.generator_ref import generator_ref, to_torch_state
    G, sd, cfg = tiny
    tsd = to_torch_state(sd)
    torch.manual_seed(123)
    z = torch.randn(2, cfg.k, cfg.z_dim)
    ref = generator_ref(tsd, z, cfg, "const")
    img = G(z.cuda(), None, noise_mode="const")[0]
    assert rel(img, ref.numpy()) < PIX_TOL
    # batch of 1 re-allocates the workspace and must agree with the batched run
    img1 = G(z[:1].cuda(), None, noise_mode="const")[0].clone()      # the engine reuses its output buffer
    assert rel(img1, ref[:1].numpy()) < PIX_TOL
    # random noise mode is reproducible under a fixed torch seed and differs from const
    torch.manual_seed(7); a = G(z[:1].cuda(), 0.7)[0].clone()
    torch.manual_seed(7); b = G(z[:1].cuda(), 0.7)[0].clone()
    assert torch.equal(a, b) and not torch.equal(a, img1)


def test_full_1024_matches_reference_samples(golden):
    """BASELINE.json full size: 1024^2, k=17, channel_base 32768: 4096 sampled pixels, per-block statistics and a
    16x down-sampled image, all taken from the reference module on CPU."""
    from morphganformer_amd.engine import Generator
    from morphganformer_amd.synth_weights import FULL1024, make_state_dict
    g = golden("gen_full1024.npz")
    G = Generator(make_state_dict(FULL1024, seed=0), FULL1024, "cuda", max_batch=1)
    G.taps = {}
    img = G(torch.from_numpy(g["z"]).cuda(), None, noise_mode="const")[0]
    taps, G.taps = G.taps, None
    assert tuple(img.shape) == (1, 3, 1024, 1024)
    amax = float(g["img_absmax"])
    pix = img.reshape(-1)[torch.from_numpy(g["idx"]).cuda()].cpu().numpy()
    assert np.abs(pix - g["pixels"]).max() / amax < PIX_TOL
    ds = torch.nn.functional.avg_pool2d(img, 16).cpu().numpy()
    assert np.abs(ds - g["img_ds"]).max() / amax < PIX_TOL
    for res, m, r in zip(g["block_res"], g["block_mean"], g["block_rms"]):
        t = taps[f"synthesis.b{int(res)}"].double()
        assert abs(float(t.mean()) - m) < 1e-3 * r, res
        assert abs(float(t.square().mean().sqrt()) - r) < 1e-3 * r, res
    assert abs(float(img.double().mean()) - float(g["img_mean"])) < 1e-3 * float(g["img_rms"])
    # production schedule (no taps requested): conv_last + ToRGB fused into one kernel -- same pixels
    ref_img = img.clone()
    assert G.fuse_torgb
    img2 = G(torch.from_numpy(g["z"]).cuda(), None, noise_mode="const")[0]
    pix2 = img2.reshape(-1)[torch.from_numpy(g["idx"]).cuda()].cpu().numpy()
    assert np.abs(pix2 - g["pixels"]).max() / amax < PIX_TOL
    assert float((img2 - ref_img).abs().max()) / amax < 1e-5


def _check_att_full(att, g, slot, sample):
    for key in g["layers"]:
        key = str(key)
        probs, argmax = att["synthesis." + key]
        F = g["argmax_" + key].shape[1]
        assert tuple(argmax.shape[1:]) == (F,), key
        dec = np.unpackbits(g["decided_" + key][slot])[:F].astype(bool)
        assert dec.mean() > 0.99, key
        am = argmax[sample].cpu().numpy()
        assert np.array_equal(am[dec], g["argmax_" + key][slot][dec]), (key, int((am[dec] != g["argmax_" + key][slot][dec]).sum()))
        pr = probs[sample].cpu().numpy()
        assert np.abs(pr[g["rows_" + key]] - g["probs_" + key][slot]).max() < 1e-4, key
        # the handed-out argmax is the argmax of the handed-out probabilities wherever those are decided
        top2 = np.sort(pr, axis=-1)[:, -2:]
        own = (top2[:, 1] - top2[:, 0]) > 1e-6
        assert np.array_equal(am[own], pr.argmax(-1)[own]), key


def test_full_1024_attention_assignments_batch1(golden):
    """SURVEY 8d's integer gate at FULL size: the per-pixel argmax latent assignment of every one of the 1024^2 model's 11 attention
    layers (F up to 16 384, C = 512 / 256) against the reference module's own (att_full1024.npz, networks.py:505-524,776-792) --
    exact on every pixel the reference decides by more than 1e-4, probabilities <= 1e-4 on 256 sampled rows -- at batch 1."""
    from morphganformer_amd.engine import Generator
    from morphganformer_amd.synth_weights import FULL1024, make_state_dict
    g = golden("att_full1024.npz")
    G = Generator(make_state_dict(FULL1024, seed=0), FULL1024, "cuda", max_batch=1)
    for slot in range(2):
        img, att = G(torch.from_numpy(g["z"][slot:slot + 1]).cuda(), None, noise_mode="const", return_att=True, att_format="maps")
        assert len(att) == 11
        _check_att_full(att, g, slot, 0)


def test_full_1024_attention_assignments_bench_dispatch(golden):
    """The same gate in the dispatch bench.py times: drivers.DEFAULT_BATCH (32) candidates per forward (the wide attention launches, the
    batched conv kernels in front of them).  Candidates 0 / 31 carry the fixture's first latent, 1 / 30 its second, the rest are other
    latents; the 738 MB-per-image stacked tensor is not built (att_format="maps")."""
    from morphganformer_amd.drivers import DEFAULT_BATCH
    from morphganformer_amd.engine import Generator
    from morphganformer_amd.synth_weights import FULL1024, make_state_dict, synthetic_latents
    g = golden("att_full1024.npz")
    B = DEFAULT_BATCH
    G = Generator(make_state_dict(FULL1024, seed=0), FULL1024, "cuda", max_batch=B)
    z = torch.from_numpy(synthetic_latents(FULL1024, B, seed=4000)).clone()
    z[0], z[B - 1], z[1], z[B - 2] = (torch.from_numpy(g["z"][i]) for i in (0, 0, 1, 1))
    img, att = G.forward_workspace(z.cuda(), None, noise_mode="const", return_att=True, att_format="maps")
    for slot, sample in ((0, 0), (0, B - 1), (1, 1), (1, B - 2)):
        _check_att_full(att, g, slot, sample)


def test_full_1024_bench_batch_equals_batch1(golden):
    """The dispatch bench.py times: drivers.DEFAULT_BATCH (32) candidates per forward at 1024^2 (the persistent Winograd form on the 1024^2 layers, 100 000-workgroup grids in XCD-contiguous order, the
    `winograd_fills_chip` branches, the half-resolution skip fused into conv1's epilogue, 3.4 GB activation tensors behind 32-bit
    buffer offsets).  The reference's modulated conv treats the batch as groups (networks.py:300-303), so candidate j of a batched
    forward must equal a batch-1 forward of the same latent and the same per-layer noise: checked on EVERY pixel of all candidates
    (<= 3e-5 of max|img| through the 19 layers: the two batch sizes take different kernels -- split-K tap lists and the one-shot Winograd form, which
    scales the INPUT by the style, at batch 1; the persistent form, which scales the WEIGHTS like networks.py:288-291, at full batch -- i.e.
    float32 re-association noise, measured 1.2e-5 at worst), and candidate 0
    under noise_mode="const" against the reference module's own 1024^2 output (gen_full1024.npz, 1e-3)."""
    from morphganformer_amd.drivers import DEFAULT_BATCH
    from morphganformer_amd.engine import Generator
    from morphganformer_amd.synth_weights import FULL1024, make_state_dict, synthetic_latents
    g = golden("gen_full1024.npz")
    cfg, B = FULL1024, DEFAULT_BATCH
    G = Generator(make_state_dict(cfg, seed=0), cfg, "cuda", max_batch=B)
    idx = torch.from_numpy(g["idx"]).cuda()
    amax = float(g["img_absmax"])
    z = torch.from_numpy(synthetic_latents(cfg, B, seed=4242)).cuda()
    z[0] = torch.from_numpy(g["z"]).cuda()[0]
    # (1) candidate 0 of a full-batch forward, const noise, against the reference's pixels
    img = G.forward_workspace(z, None, noise_mode="const")[0]
    assert tuple(img.shape) == (B, 3, 1024, 1024)
    pix = img[0].reshape(-1)[idx].cpu().numpy()
    assert np.abs(pix - g["pixels"]).max() / amax < PIX_TOL
    ds = torch.nn.functional.avg_pool2d(img[:1], 16).cpu().numpy()
    assert np.abs(ds - g["img_ds"]).max() / amax < PIX_TOL
    img_const0 = img[0].clone()                                       # (img is the workspace buffer: the next forward overwrites it)
    # (2) injected per-layer noise, distinct per candidate: one batched forward == B batch-1 forwards, every pixel
    gen = torch.Generator(device="cuda"); gen.manual_seed(11)
    noises = {lp.name: torch.randn(B, lp.res * lp.res, device="cuda", generator=gen) for lp in G.plan.layers if lp.noise_strength is not None}
    img25 = G.forward_workspace(z, None, noise_mode="inject", noises=noises)[0].clone()
    assert not torch.equal(img25[0], img_const0)                      # the injected maps really are used
    worst = 0.0
    for j in range(B):
        one = G.forward_workspace(z[j:j + 1].contiguous(), None, noise_mode="inject",
                                  noises={k: v[j:j + 1].contiguous() for k, v in noises.items()})[0]
        m = float(img25[j].abs().max())
        err = float((img25[j] - one[0]).abs().max()) / m
        worst = max(worst, err)
        assert err < 3e-5, (j, err)
        assert float((img25[j].reshape(-1)[idx] - one[0].reshape(-1)[idx]).abs().max()) / m < 3e-5
    # distinct latents give distinct images (a stuck sample index would pass the equality above only for j = 0)
    assert float((img25[1] - img25[2]).abs().max()) > 1e-3 * amax
    print(f"batch-{B} vs batch-1 at 1024^2: worst relative pixel difference {worst:.2e}")


def test_list2tensor_matches_reference(tiny, golden):
    """SynthesisNetwork.list2tensor (networks.py:1222-1242): stacked, nearest-neighbour upsampled attention maps of return_att=True."""
    G, sd, cfg = tiny
    g = golden("att_tiny.npz")
    img, t = G(torch.from_numpy(g["z"]).cuda(), None, noise_mode="const", return_att=True)       # the reference's return contract
    assert tuple(t.shape) == tuple(g["shape"])
    _img, att = G(torch.from_numpy(g["z"]).cuda(), None, noise_mode="const", return_att=True, att_format="maps")
    assert torch.equal(G.list2tensor(att), t)
    assert np.abs(t[:, :, :, 0, 3::8, 5::8].cpu().numpy() - g["att_sub"]).max() < 1e-4
    # layer 0 is the 4x4 map: constant over 16x16 pixel blocks
    blk = t[:, :, 0, 0].reshape(2, cfg.k - 1, 4, 16, 4, 16)
    assert torch.equal(blk, blk[:, :, :, :1, :, :1].expand_as(blk))
    assert abs(float(t.sum(1).mean()) - 1.0) < 1e-5          # probabilities over the latent components


def test_generator_forward_boundary_arguments_vs_reference(tiny, golden):
    """Generator.forward(ws=per-layer latents), truncation_psi + truncation_cutoff, return_ws, subnet="synthesis" and the stacked
    return_att tensor (networks.py:1304-1331, 935-941, 1252-1253) against the reference module (tests/golden/wplus_tiny.npz); plus
    the ownership contract: results of two calls do not alias, and a call with another batch size leaves earlier workspaces intact."""
    from morphganformer_amd import _lib
    G, sd, cfg = tiny
    g = golden("wplus_tiny.npz")
    z, ws = torch.from_numpy(g["z"]).cuda(), torch.from_numpy(g["ws"]).cuda()
    img, att = G(ws=ws, noise_mode="const", return_att=True)
    assert rel(img, g["img_ws"]) < PIX_TOL
    assert tuple(att.shape) == tuple(g["att_shape"]) and np.abs(att[:, :, :, 0, 3::8, 5::8].cpu().numpy() - g["att_sub"]).max() < 1e-4
    assert rel(G(ws=ws, noise_mode="const", subnet="synthesis"), g["img_synthesis_subnet"]) < PIX_TOL
    # a broadcast ws is the ordinary path: same pixels as z
    gt = golden("gen_tiny.npz")
    wsb = G(torch.from_numpy(gt["z"]).cuda(), None, subnet="mapping")
    assert rel(G(ws=wsb, noise_mode="const")[0], gt["img_const"]) < PIX_TOL
    for psi, cut, kimg, kws in ((0.6, 5, "img_cut", "ws_cut"), (0.6, None, "img_psi", "ws_psi")):
        im, wso = G(z, None, truncation_psi=psi, truncation_cutoff=cut, noise_mode="const", return_ws=True)
        assert tuple(wso.shape) == g[kws].shape and rel(wso, g[kws]) < 1e-5, kws
        assert rel(im, g[kimg]) < PIX_TOL, kimg
        assert rel(G.mapping(z, psi, cut), g[kws]) < 1e-5
    with pytest.raises(_lib.MgfError):
        G(ws=ws[:, :, :3], noise_mode="const")                   # wrong number of layer slots: refuse, never guess
    # ownership: the reference returns fresh tensors
    a = G(z[:1], None, noise_mode="const")[0]
    keep = a.clone()
    b = G(z[1:], None, noise_mode="const")[0]
    assert a.data_ptr() != b.data_ptr() and torch.equal(a, keep) and not torch.equal(a, b)
    # workspaces are per batch size and survive calls with other sizes (hipGraphs of the engines hold their pointers)
    p2 = G.forward_workspace(z, None, noise_mode="const")[0].data_ptr()
    G(z[:1], None, noise_mode="const")
    assert G.forward_workspace(z, None, noise_mode="const")[0].data_ptr() == p2


def test_graph_replay_survives_other_batch_sizes(golden):
    """eng.run(k); G(w) with another batch size (the drivers render a preview at every improvement); eng.run(k) -- the captured graph
    keeps running on its own workspace and the run equals an undisturbed one."""
    from morphganformer_amd.engine import Generator
    from morphganformer_amd.projection import ProjectionArgs, ProjectionEngine
    from morphganformer_amd.synth_weights import TINY, make_state_dict
    g = golden("loop_tiny.npz")
    res = []
    for disturb in (False, True):
        G = Generator(make_state_dict(TINY, seed=0), TINY, "cuda", max_batch=1)
        eng = ProjectionEngine(G, torch.from_numpy(g["target"]).cuda(), torch.from_numpy(g["latent_mean"]).cuda(), float(g["latent_std"]),
                               ProjectionArgs(step=48), use_mse=True, lm_target=g["lm_target"], lm_steps=g["lm_steps"],
                               eps=torch.from_numpy(g["eps"]).cuda(), noise_mode="const", batch=4, use_graph=True)
        eng.run(24)
        if disturb:
            junk = [G(torch.randn(n, TINY.k, TINY.z_dim, device="cuda"), None, noise_mode="const")[0] for n in (1, 7, 2)]
            assert G._pins.get((4, True)) == 1            # the engine pinned the LEAN workspace of its batch size
        eng.run(24)
        res.append(eng.result())
    assert res[0][1] == res[1][1] and torch.equal(res[0][0], res[1][0]) and np.array_equal(res[0][3], res[1][3])


def test_lean_workspace_equals_full_workspace_and_is_a_third_of_it():
    """forward_workspace(lean=True) -- what the literal loop runs on: layer outputs are views of seven arenas re-used block after block --
    gives the image of the full workspace bit for bit (same kernels, same operands, other addresses), at 1024^2 and on the tiny config
    (whose last block does NOT fuse the skip up-sampling: the lean layout keeps that block's full-resolution skip tensor), for a third of
    the bytes; requests that hand out intermediate tensors (taps, return_att) fall back to the full flavour."""
    from morphganformer_amd.engine import Generator
    from morphganformer_amd.synth_weights import FULL1024, TINY, make_state_dict, synthetic_latents
    for cfg, n in ((TINY, 3), (FULL1024, 2)):
        G = Generator(make_state_dict(cfg, seed=0), cfg, "cuda", max_batch=1)
        z = torch.from_numpy(synthetic_latents(cfg, n, 77)).cuda()
        full = G.forward_workspace(z, None, noise_mode="const")[0].clone()
        lean = G.forward_workspace(z, None, noise_mode="const", lean=True)[0].clone()
        assert torch.equal(full, lean) and G.lean and set(G._workspaces) >= {(n, False), (n, True)}
        assert G._workspace_bytes(n, True) < 0.4 * G._workspace_bytes(n, False) or cfg is TINY
        again = G.forward_workspace(z, None, noise_mode="const", lean=True, return_att=True, att_format="maps")
        assert not G.lean and torch.equal(again[0], full)
        G.taps = {}
        G.forward_workspace(z, None, noise_mode="const", lean=True)
        assert not G.lean and len(G.taps) > 4
        G.taps = None
    assert G._workspace_bytes(32, True) < 17 * 2 ** 30 < 40 * 2 ** 30 < G._workspace_bytes(32, False)       # 32 steps per forward: 15 vs 45 GiB
