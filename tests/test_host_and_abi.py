"""CPU: host-side logic, the C-ABI surface (the library loads and exports every symbol include/mgf.h declares -- no
compute call is made without a GPU), argument validation that happens before any launch, and the world-size-2 result
gather over gloo."""
import ctypes
import math
import os
import re
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from morphganformer_amd import _lib, build
    build.build()
    return _lib.lib()


def test_header_symbols_exported(lib):
    from morphganformer_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "mgf.h")).read()
    declared = sorted(set(re.findall(r"\b(mgf_[a-z0-9_]+)\s*\(", hdr)))
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/mgf.h but not exported by libmgf_hip.so"
    assert sorted(_lib.EXPORTED_SYMBOLS) == declared, set(_lib.EXPORTED_SYMBOLS) ^ set(declared)
    # every declaration cites the reference interface it replaces
    assert hdr.count(".py:") + hdr.count(".cpp:") + hdr.count(".cu:") >= 15


def test_abi_struct_layouts_match_header():
    from morphganformer_amd import _lib
    assert ctypes.sizeof(_lib.Epilogue) == 48
    assert ctypes.sizeof(_lib.StyleJob) == 64
    assert ctypes.sizeof(_lib.AttnJob) == 32
    d = _lib.ConvDesc()
    assert ctypes.sizeof(d) == 4 * 12 + 4 * 27 + 4 * 8 + 4 * 2 + 4 + 24 + 8 or ctypes.sizeof(d) % 8 == 0
    assert _lib.ConvDesc.y_pitch.offset % 8 == 0


def test_abi_struct_layouts_match_a_c_compiler(tmp_path):
    """include/mgf.h is plain C: compile a probe with gcc and compare every struct's size and field offsets with ctypes."""
    from morphganformer_amd import _lib
    structs = {"mgf_epilogue": _lib.Epilogue, "mgf_conv_desc": _lib.ConvDesc, "mgf_conv_prof_rec": _lib.ConvProfRec,
               "mgf_style_job": _lib.StyleJob, "mgf_attn_job": _lib.AttnJob}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "mgf.h"', "int main(void) {"]
    for cname, cls in structs.items():
        lines.append(f'printf("{cname} %zu\\n", sizeof({cname}));')
        for fname, *_ in cls._fields_:
            lines.append(f'printf("{cname}.{fname} %zu\\n", offsetof({cname}, {fname}));')
    lines.append("return 0; }")
    src = tmp_path / "probe.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "probe"
    inc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", inc, str(src), "-o", str(exe)], check=True)
    got = dict(l.split() for l in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    for cname, cls in structs.items():
        assert int(got[cname]) == ctypes.sizeof(cls), cname
        for fname, *_ in cls._fields_:
            assert int(got[f"{cname}.{fname}"]) == getattr(cls, fname).offset, (cname, fname)


def test_host_side_validation_without_gpu(lib):
    """Errors raised before any kernel launch (mirrors the TORCH_CHECKs of the plugins) are reachable on CPU."""
    assert lib.mgf_version() >= 100
    assert lib.mgf_reduce_scratch_floats() >= 256
    assert lib.mgf_mapping_param_floats(17, 32, 4) == 4 * (2 * 1024 + 64) + 1024 + 32 + 4 * (6 * 1024 + 2 * 16 * 32 + 128) + 1024 + 32
    rc = lib.mgf_bias_act(None, None, None, None, None, None, 0, 5, 1, 1, 7, 3, 0.2, 1.0, -1.0, None)
    assert rc == -1 and b"grad" in lib.mgf_last_error()
    rc = lib.mgf_bias_act(None, None, None, None, None, None, 0, 2 ** 31, 1, 1, 0, 3, 0.2, 1.0, -1.0, None)
    assert rc == -4 and b"too large" in lib.mgf_last_error()
    rc = lib.mgf_bias_act(None, None, None, None, None, None, 0, 5, 1, 1, 0, 42, 0.2, 1.0, -1.0, None)
    assert rc == -1 and b"activation" in lib.mgf_last_error()
    assert lib.mgf_bias_act(None, None, None, None, None, None, 0, 0, 1, 1, 0, 3, 0.2, 1.0, -1.0, None) == 0     # empty tensor is a no-op
    rc = lib.mgf_upfirdn2d(None, None, None, 0, 1, 1, 2, 2, 4, 4, 2, 1, 9, 9, 1, 1, 1, 1, 5, 5, 1, 1, 1, 1, 0, 0, 0, 0, 0, 1.0, None, None)
    assert rc == -1 and b"empty" in lib.mgf_last_error()
    rc = lib.mgf_mapping_forward(1, 1, 1, 1, 17, 64, 4, 1, None)
    assert rc == -2 and b"latent width" in lib.mgf_last_error()


def test_ops_refuse_cpu_tensors_and_missing_library(monkeypatch):
    from morphganformer_amd import _lib
    from morphganformer_amd.torch_utils.ops import bias_act, upfirdn2d
    with pytest.raises(_lib.MgfError, match="no CPU fallback"):
        bias_act.bias_act(torch.zeros(2, 3), None)
    with pytest.raises(_lib.MgfError, match="no CPU fallback"):
        upfirdn2d.upfirdn2d(torch.zeros(1, 1, 4, 4), torch.ones(2, 2))
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libmgf_hip.so")
    with pytest.raises(_lib.MgfError, match="is missing"):
        _lib.lib()


def test_product_package_never_imports_oracle():
    pkg = os.path.join(ROOT, "morphganformer_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                for m in re.finditer(r"^\s*(from|import)\s+oracle\b.*$", src, re.M):
                    line_start = src.rfind("\n", 0, m.start()) + 1
                    func = src.rfind("\ndef ", 0, m.start())
                    assert "def smoke_projection" in src[func:m.start()], f"{f}: product code imports the oracle: {m.group(0)}"


def test_config_tables_and_synth_weights():
    from morphganformer_amd.synth_weights import FULL1024, TINY, make_state_dict, sinusoidal_grid
    assert FULL1024.num_ws == 19 and FULL1024.block_resolutions == [4, 8, 16, 32, 64, 128, 256, 512, 1024]
    assert [FULL1024.channels(r) for r in FULL1024.block_resolutions] == [512, 512, 512, 512, 512, 256, 128, 64, 32]
    assert [r for r in FULL1024.block_resolutions if FULL1024.has_attention(r)] == [4, 8, 16, 32, 64, 128]
    rows = FULL1024.layer_table()
    assert len(rows) == 19 and [r[5] for r in rows] == list(range(19)) and sum(1 for r in rows if r[6]) == 11
    # algorithmic FLOPs of the convs, SURVEY.md 8a row P5 / 8d: 172.7 GFLOP
    gf = sum(2 * (9 if nm != "torgb" else 1) * ci * co * ((res // up) ** 2) for res, nm, ci, co, up, *_ in rows) / 1e9
    gf += sum(2 * FULL1024.channels(r // 2) * FULL1024.channels(r) * (r // 2) ** 2 for r in FULL1024.block_resolutions[1:]) / 1e9
    assert abs(gf - 172.7) < 0.5 and abs(FULL1024.conv_gflop() - gf) < 1e-9
    a, b = make_state_dict(TINY, 0), make_state_dict(TINY, 0)
    assert list(a) == list(b) and all(np.array_equal(a[k], b[k]) for k in a)
    assert not np.array_equal(a["pos"], make_state_dict(TINY, 1)["pos"])
    n_params = sum(v.size for k, v in make_state_dict(FULL1024, 0).items()
                   if not k.endswith(("resample_kernel", "noise_const", "grid_pos", "w_avg")))
    assert n_params == 30916948                                     # generator parameter count
    g = sinusoidal_grid(8, 32)
    assert g.shape == (8, 8, 32) and np.allclose(g[3, :, :8], g[5, :, :8]) and np.allclose(g[:, 2, 16:24], g[:, 2, 16:24])


def test_schedule_and_args_mirror_reference_defaults():
    from morphganformer_amd.projection import ProjectionArgs, get_lr, noise_schedule, synthetic_landmarks
    from oracle.loss_ref import get_lr_ref, noise_strength_ref
    a = ProjectionArgs()
    assert (a.step, a.lamda, a.beta, a.lr, a.noise, a.noise_ramp, a.lr_rampup, a.lr_rampdown) == (5000, 0.01, 1.0, 0.01, 0.05, 0.75, 0.05, 0.25)
    for t in (0, 0.01, 0.05, 0.3, 0.76, 0.99):
        assert get_lr(t, 0.01) == get_lr_ref(t, 0.01)
    sig = noise_schedule(40, 23.3, 0.05, 0.75)
    assert all(sig[i] == noise_strength_ref(i / 40, 23.3) for i in range(40)) and sig[30] == 0.0 and sig[0] == 23.3 * 0.05
    t, s = synthetic_landmarks(5, 1024, 1)
    assert t.shape == (68, 2) and s.shape == (5, 68, 2) and np.abs(s - t[None]).max() <= 16 and t.min() >= 256 and t.max() < 768


def test_mapping_blob_layout(lib):
    from morphganformer_amd.engine import pack_mapping_params
    from morphganformer_amd.synth_weights import TINY, make_state_dict
    sd = make_state_dict(TINY, 0)
    blob = pack_mapping_params(sd, TINY)
    assert blob.dtype == np.float32 and blob.size == lib.mgf_mapping_param_floats(TINY.k, TINY.w_dim, 4)
    # first matrix = global_mlp.l0.fc0 with lrmul and He gain folded
    w = sd["mapping.global_mlp.l0.fc0.weight"].astype(np.float64) * (0.01 / math.sqrt(32))
    assert np.allclose(blob[:1024].reshape(32, 32), w, rtol=1e-6)
    assert np.allclose(blob[1024:1056], sd["mapping.global_mlp.l0.fc0.bias"] * 0.01, rtol=1e-6)


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


GLOO_WORKER = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from morphganformer_amd.distributed import shard_items, pack_result, gather_results, gather_many
dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{sys.argv[2]}", rank=int(sys.argv[3]), world_size=2)
rank = dist.get_rank()
mine = shard_items(5, rank, 2)
assert mine == ([0, 2, 4] if rank == 0 else [1, 3])
torch.manual_seed(rank)
lat = torch.randn(1, 17, 32)
res = gather_results(lat, 0.5 + rank, 10 + rank, item=rank)
assert res["latents"].shape == (2, 17, 32) and res["steps"].tolist() == [10, 11] and res["losses"].tolist() == [0.5, 1.5]
assert torch.equal(res["latents"][rank], lat[0])            # f32 -> f64 -> f32 round trip is exact
recs = torch.stack([pack_result(torch.full((1, 17, 32), float(i)), float(i), i, item=i) for i in mine])
allr = gather_many(recs, 3)
assert allr.shape[0] == 5 and allr[:, -1].tolist() == [0, 1, 2, 3, 4] and allr[:, 0].tolist() == [0, 1, 2, 3, 4]
dist.barrier(); dist.destroy_process_group()
print("ok", rank)
"""


def test_result_gather_gloo_world2(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(GLOO_WORKER)
    port = _free_port()
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, str(port), str(r)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for r in range(2)]
    outs = [p.communicate(timeout=120)[0].decode() for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"ok {r}" in o, o


# ----------------------------------------------------------------------------------------------------------------- callers
def _tiny_snapshot(path, seed=3):
    from morphganformer_amd import loader
    from morphganformer_amd.synth_weights import TINY, make_state_dict
    sd = make_state_dict(TINY, seed=seed)
    kw = dict(z_dim=TINY.z_dim, c_dim=0, w_dim=TINY.w_dim, k=TINY.k, img_resolution=TINY.img_resolution, img_channels=3,
              synthesis_kwargs=dict(channel_base=TINY.channel_base, channel_max=TINY.channel_max, end_res=TINY.attn_max_log2res),
              mapping_kwargs=dict(num_layers=TINY.mapping_layers))
    loader.save_snapshot_like_reference(path, {"G": sd, "D": {"b4.fc.weight": np.zeros((2, 3), np.float32)}, "Gs": sd},
                                        {"G": "Generator", "D": "Discriminator", "Gs": "Generator"}, {"G": kw, "Gs": kw})
    return sd, kw


def test_network_pickle_reader_is_inert_and_complete(tmp_path):
    """loader.load_network's file format (persistence.py:110-118,171-194) without exec: stubs, state_dict, config."""
    import pickle
    from morphganformer_amd import loader
    from morphganformer_amd.synth_weights import TINY
    p = str(tmp_path / "net.pkl")
    sd, kw = _tiny_snapshot(p)
    stubs = loader.load_network_stubs(p)
    assert set(stubs) == {"G", "D", "Gs"} and stubs["D"].class_name == "Discriminator"
    got = stubs["Gs"].state_dict()
    assert set(got) == set(sd)
    for k in sd:
        assert np.array_equal(np.asarray(sd[k]), got[k].numpy()), k
    assert loader.config_from_stub(stubs["Gs"]) == TINY
    # anything outside the allow-list is refused before it can run

    class Evil:
        def __reduce__(self):
            return (os.system, ("echo pwned > %s" % (tmp_path / "pwned"),))
    with open(p, "wb") as f:
        pickle.dump({"G": Evil(), "D": Evil(), "Gs": Evil()}, f)
    with pytest.raises(pickle.UnpicklingError):
        loader.load_network_stubs(p)
    assert not (tmp_path / "pwned").exists()
    # a snapshot without the three networks is an error, like the reference's KeyError on ["Gs"]
    with open(p, "wb") as f:
        pickle.dump({"G": 1}, f)
    with pytest.raises(pickle.UnpicklingError):
        loader.load_network_stubs(p)


@pytest.mark.skipif(not os.path.isdir("/root/reference/torch_utils"), reason="reference tree not present (GPU box)")
def test_network_pickle_reader_on_a_real_reference_pickle(tmp_path):
    """Pickle the REFERENCE Generator with the REFERENCE persistence machinery, read it back with the inert reader."""
    import pickle
    from morphganformer_amd import loader
    from morphganformer_amd.synth_weights import TINY, make_state_dict
    from oracle.make_golden import build_reference_generator, import_reference
    ref = import_reference()
    sd = make_state_dict(TINY, seed=5)
    G = build_reference_generator(ref, TINY, sd)
    p = str(tmp_path / "ref.pkl")
    with open(p, "wb") as f:
        pickle.dump(dict(G=G, D=G, Gs=G), f)
    assert b"class Generator" in open(p, "rb").read()           # module source is embedded ... and never executed below
    stubs = loader.load_network_stubs(p)
    got = stubs["Gs"].state_dict()
    want = G.state_dict()
    assert set(got) == set(want)
    for k, v in want.items():
        assert torch.equal(v, got[k]), k
    cfg = loader.config_from_stub(stubs["Gs"])
    assert cfg == TINY


def test_driver_host_helpers(tmp_path):
    from PIL import Image
    from morphganformer_amd import drivers
    # image_transform: shorter side -> size (bilinear), centre crop, [-1,1]
    rng = np.random.Generator(np.random.PCG64(1))
    im = Image.fromarray(rng.integers(0, 256, size=(40, 64, 3), dtype=np.uint8), "RGB")
    x = drivers.image_transform(im, size=32, device="cpu")
    assert x.shape == (1, 3, 32, 32) and x.dtype == torch.float32 and -1 <= float(x.min()) and float(x.max()) <= 1
    want = im.resize((int(32 * 64 / 40), 32), Image.BILINEAR)
    left = int(round((want.size[0] - 32) / 2.0))
    want = np.asarray(want.crop((left, 0, left + 32, 32)), dtype=np.float32) / 255
    assert np.array_equal(x[0].permute(1, 2, 0).numpy(), ((torch.from_numpy(want) - 0.5) / 0.5).numpy())
    same = drivers.image_transform(Image.fromarray(np.full((32, 32, 3), 255, np.uint8)), size=32, device="cpu")
    assert float(same.min()) == 1.0
    # .mat latent exchange: key 'w', float32 [1,k,D]
    w = rng.standard_normal((1, 17, 32)).astype(np.float32)
    p = drivers.save_latent_mat(str(tmp_path / "a" / "x.mat"), torch.from_numpy(w))
    assert np.array_equal(drivers.load_latent_mat(p), w)
    import scipy.io as sio
    sio.savemat(str(tmp_path / "bad.mat"), {"w": np.zeros(4, np.float32)})
    with pytest.raises(ValueError):
        drivers.load_latent_mat(str(tmp_path / "bad.mat"))


def test_cli_arguments_mirror_the_reference_scripts():
    """argparse names/defaults of 1024_generate.py:44-54 and 1024_example_wing_loss_perceptual_sqz_MSE.py:222-245."""
    from morphganformer_amd.cli import build_parser
    ap = build_parser()
    g = ap.parse_args(["generate", "--model", "m.pkl"])
    assert (g.gpus, g.output_dir, g.images_num, g.truncation_psi, g.ratio) == ("0", "images", 32, 0.7, 1.0)
    p = ap.parse_args(["project", "--image", "x.png"])
    assert p.model == "models/ffhq-snapshot-1024_v2.pkl"
    assert (p.size, p.n_mean_latent, p.step, p.lamda, p.beta) == (1024, 10000, 5000, 0.01, 1)
    assert (p.lr_rampup, p.lr_rampdown, p.lr, p.noise, p.noise_ramp, p.ratio, p.truncation_psi) == (0.05, 0.25, 0.01, 0.05, 0.75, 1.0, 0.7)
    assert p.noise_regularize == 1e5 and p.w_plus is False
    m = ap.parse_args(["morph", "--model", "m.pkl", "--w1", "a.mat", "--w2", "b.mat", "--out", "o"])
    assert m.alphas == "0.5" and m.truncation_psi == 0.7


def test_reference_gray_conversion_kat():
    """cv2.normalize(NORM_MINMAX, CV_8U) + COLOR_BGR2GRAY on RGB-ordered data (...sqz_MSE.py:161-163), hand-checked values."""
    from morphganformer_amd.drivers import reference_gray_u8
    img = np.zeros((1, 3, 3), np.float32)
    img[0, 0] = (-1.0, -1.0, -1.0)          # global minimum -> (0,0,0)   -> 0
    img[0, 1] = (1.0, 1.0, 1.0)             # global maximum -> (255,..)  -> 255
    img[0, 2] = (1.0, -1.0, 0.0)            # (255, 0, 128) -> (255*1868 + 128*4899 + 8192) >> 14 = 67
    assert reference_gray_u8(img).tolist() == [[0, 255, 67]]
    assert reference_gray_u8(np.full((2, 2, 3), 0.3, np.float32)).tolist() == [[0, 0], [0, 0]]      # flat image: max == min
