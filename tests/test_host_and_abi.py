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


def test_build_cache_keys_objects_on_flags_compiler_and_source(tmp_path):
    """morphganformer_amd/build.py names every object after a hash of the compiler banner, the full flag list, the source and the shared
    headers: an object compiled with a timing-ablation macro (or by another compiler) has ANOTHER name and can never be linked into the
    product; the experiment build keeps its objects in exp_build/_obj and never reads the product's cache."""
    import inspect
    import os
    from morphganformer_amd import build as B
    src = os.path.join(B.CSRC, "conv_taps.hip")
    base = B.object_key(src, B.FLAGS)
    assert base == B.object_key(src, list(B.FLAGS)) and len(base) == 16
    assert B.object_key(src, B.FLAGS[:-2] + ["-DMGF_EXP=1"] + B.FLAGS[-2:]) != base
    assert B.object_key(os.path.join(B.CSRC, "wino3.hip"), B.FLAGS) != base
    saved = B._compiler_id
    try:
        B._compiler_id = (saved or B.compiler_id()) + " (another build)"
        assert B.object_key(src, B.FLAGS) != base
    finally:
        B._compiler_id = saved
    # the library on disk was linked from exactly the objects the current sources hash to
    if os.path.exists(B.LIB + ".objs"):
        want = [f"{s}.{B.object_key(os.path.join(B.CSRC, s), B.FLAGS)}.o" for s in B.SOURCES]
        assert B._linked_from(B.LIB) == want
    code = inspect.getsource(B.build_experiment)
    assert "exp_build" in code and "OBJ" not in code.replace("_obj", "")
    with open(os.path.join(B.ROOT, "tools", "build_exp.sh")) as f:
        sh = f.read()
    assert "cp -u" not in sh.split("set -e")[1] and "csrc/_obj" not in sh


def test_product_package_never_imports_oracle():
    pkg = os.path.join(ROOT, "morphganformer_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                m = re.search(r"^\s*(from|import)\s+oracle\b.*$", src, re.M)
                assert m is None, f"{f}: product code imports the oracle: {m.group(0)}"
                assert "importlib" not in src or "oracle" not in src, f"{f}: product code may reach the oracle through importlib"


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
    assert sig.dtype == np.float32 and all(float(sig[i]) == noise_strength_ref(i / 40, 23.3) for i in range(40)) and sig[30] == 0.0
    assert sig[0] == np.float32(23.3) * np.float32(0.05)            # float32 tensor x python scalar, not the float64 product
    for steps, ls in ((5000, 23.3), (1000, 7.77)):                  # long schedules against the tensor expression of the driver (:156)
        lt = torch.tensor(ls, dtype=torch.float32)
        want = [(lt * 0.05 * max(0, 1 - (i / steps) / 0.75) ** 2).item() for i in range(steps)]
        assert noise_schedule(steps, ls, 0.05, 0.75).astype(np.float64).tolist() == want
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
from morphganformer_amd.distributed import shard_items, pack_result, gather_results, gather_many, WorkQueue, init_process_group
init_process_group("gloo", rank=int(sys.argv[3]), world_size=2, host="127.0.0.1", port=int(sys.argv[2]))
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
# dynamic work queue: the two ranks together take every item exactly once, whatever the interleaving
import time
taken = []
for i in WorkQueue(7):
    taken.append(i)
    time.sleep(0.01 * (1 + 2 * rank))
cnt = torch.zeros(7)
cnt[taken] = 1
dist.all_reduce(cnt)
assert cnt.tolist() == [1.0] * 7 and len(taken) >= 1
# a second and a third queue in the same process group start from zero again (two-stage runs), also under another name
for n_items, kw in ((5, {}), (4, {"name": "stage2"})):
    dist.barrier()
    taken = list(WorkQueue(n_items, **kw))
    cnt = torch.zeros(n_items)
    cnt[taken] = 1
    dist.all_reduce(cnt)
    assert cnt.tolist() == [1.0] * n_items, (n_items, cnt.tolist())
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


GLOO_WORKER_RAGGED = r"""
import os, sys, time, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from morphganformer_amd.distributed import pack_result, run_sharded, unpack_results, make_store, set_store
world = int(sys.argv[4])
# a caller with a process group of its own: it creates the store, gives it to torch and hands it to the module (set_store)
store = make_store(int(sys.argv[3]), world, "127.0.0.1", int(sys.argv[2]))
dist.init_process_group("gloo", store=store, rank=int(sys.argv[3]), world_size=world)
set_store(store)
rank = dist.get_rank()
N, K, D = 11, 17, 32
# ragged per-item cost, as when "no face found" skips most steps of some targets (...sqz_MSE.py:165-166): item i costs cost[i]
cost = [0.02, 0.30, 0.02, 0.02, 0.25, 0.02, 0.02, 0.02, 0.20, 0.02, 0.02]
def work(i):
    time.sleep(cost[i] * (1.0 + 0.5 * rank))                      # the ranks also run at different speeds
    return pack_result(torch.full((1, K, D), float(i)), 10.0 + i, 100 + i, item=i)
for dynamic in (True, False, True):                                 # (the second dynamic pass: a fresh queue epoch in the same group)
    dist.barrier()
    rows, mine = run_sharded(N, work, K * D + 3, "cpu", dynamic=dynamic)
    res = unpack_results(rows, (K, D))
    assert res["items"].tolist() == list(range(N)), res["items"].tolist()                 # every item exactly once, on every rank
    assert res["steps"].tolist() == [100 + i for i in range(N)] and res["losses"].tolist() == [10.0 + i for i in range(N)]
    assert all(float(res["latents"][i, 0, 0]) == float(i) for i in range(N))
    cnt = torch.zeros(N); cnt[mine] = 1
    dist.all_reduce(cnt)
    assert cnt.tolist() == [1.0] * N
    if not dynamic:
        assert mine == list(range(rank, N, world))
    else:
        assert len(mine) >= 1                                       # nobody starves; who takes what depends on the interleaving
dist.barrier(); dist.destroy_process_group()
print("ok", rank)
"""


def test_ragged_work_queue_gloo_world3(tmp_path):
    """project_many's multi-GPU skeleton (distributed.run_sharded: dynamic WorkQueue / static shards + the ragged result gather) at
    world size 3 with ragged per-item cost and ranks of different speed: every item is projected exactly once and every rank ends up
    with all results in item order."""
    script = tmp_path / "worker_ragged.py"
    script.write_text(GLOO_WORKER_RAGGED)
    port = _free_port()
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, str(port), str(r), "3"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for r in range(3)]
    outs = [p.communicate(timeout=180)[0].decode() for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"ok {r}" in o, o


def test_bench_self_launches_n_ranks():
    """`python bench.py --gpus 2` without a launcher starts two ranks itself (torch.distributed.run as a child of a parent that never
    imports torch), hands rank 0's JSON line through and fails when a rank fails -- rehearsed on CPU with the gloo dry-run leg."""
    import json
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--selftest-launch"]
    ok = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert ok.returncode == 0, ok.stderr.decode()[-2000:]
    lines = [ln for ln in ok.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0])["n_gpus"] == 2 and json.loads(lines[0])["ranks"] == [0, 1]
    bad = subprocess.run(cmd, env=dict(env, MGF_SELFTEST_FAIL_RANK="1"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert bad.returncode != 0
    # --workload config3: the sharded control flow (dynamic work queue over 16 * N items, weak + strong pass, per-rank statistics, one gather)
    # with a stub projection, three ranks of different speed
    c3 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--selftest-launch", "--workload", "config3",
                         "--config3-targets", "7", "--config3-steps", "10"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert c3.returncode == 0, c3.stderr.decode()[-2000:]
    line = json.loads([ln for ln in c3.stdout.decode().splitlines() if ln.startswith("{")][0])
    assert line["weak"]["targets"] == 21 and line["strong"]["targets"] == 7 and line["weak"]["steps_per_target"] == 10
    assert sum(line["weak"]["per_rank_targets"]) == 21 and sum(line["strong"]["per_rank_targets"]) == 7 and len(line["weak"]["per_rank_busy_s"]) == 3
    assert line["weak"]["per_rank_targets"][1] < line["weak"]["per_rank_targets"][0]          # the slow rank drew fewer items from the queue
    assert line["weak"]["iters_per_s"] > 0 and line["weak"]["rank_busy_max_s"] >= line["weak"]["rank_busy_min_s"]
    # the parent decides to launch before anything GPU-related is imported
    src = open(os.path.join(ROOT, "bench.py")).read()
    head = src[:src.index("def main():")]
    assert not re.search(r"^(import|from) (torch|numpy)", head, re.M)


def test_bench_refuses_more_ranks_than_visible_gpus():
    """`bench.py --gpus N` with N above the number of visible devices exits non-zero with a message BEFORE anything is launched or any
    GPU is touched -- as the launcher parent and as a rank the driver's own torch.distributed.run started (which would otherwise sit in
    the rendezvous or fail inside set_device)."""
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert r.returncode != 0 and b"--gpus 8 but 0 GPU(s) are visible" in r.stderr and not r.stdout.strip(), r.stderr.decode()[-2000:]
    assert b"self-launch" not in r.stderr                                   # nothing was started
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=dict(env, WORLD_SIZE="2", RANK="1", LOCAL_RANK="1",
                                                                                                MASTER_ADDR="127.0.0.1", MASTER_PORT="29999"),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert r.returncode != 0 and b"GPU(s) are visible" in r.stderr, r.stderr.decode()[-2000:]


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
    # the pair list of projection_example_v2_percept_morph.py:337-343: header and rows below the similarity threshold are skipped
    (tmp_path / "p.csv").write_text("img1,img2,simi\na.png,b.png,0.5\na.png,c.png,0.49\nc.png,b.png,0.93\n")
    assert drivers.read_pair_csv(str(tmp_path / "p.csv")) == [("a.png", "b.png"), ("c.png", "b.png")]
    assert drivers.read_pair_csv(str(tmp_path / "p.csv"), threshold=0.9) == [("c.png", "b.png")]
    assert drivers.frame_points(1024) == drivers.WARP_EXTRA_POINTS and drivers.frame_points(64)[3] == [0, 63]


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
    assert p.mode == "literal" and ap.parse_args(["project", "--image", "x.png", "--mode", "gradient"]).mode == "gradient"
    m = ap.parse_args(["morph", "--model", "m.pkl", "--w1", "a.mat", "--w2", "b.mat", "--out", "o"])
    assert m.alphas == "0.5" and m.truncation_psi == 0.7
    q = ap.parse_args(["morph-pairs", "--csv", "p.csv", "--src", "s", "--dst-raw", "r", "--dst-morph", "m"])
    assert (q.threshold, q.step, q.truncation_psi, q.model, q.dynamic) == (0.5, 5000, 0.7, "models/ffhq-snapshot-1024_v2.pkl", False)
    w = ap.parse_args(["warp", "--model", "m.pkl", "--w1", "a.mat", "--w2", "b.mat", "--landmarks", "lm.npz", "--out", "o"])
    assert w.truncation_psi == 0.7 and w.gpus == "0"                 # 1024_warp_morphs.py:127 (`truncation_psi = 0.7`)


def test_reference_gray_conversion_kat():
    """cv2.normalize(NORM_MINMAX, CV_8U) + COLOR_BGR2GRAY on RGB-ordered data (...sqz_MSE.py:161-163), hand-checked values."""
    from morphganformer_amd.drivers import reference_gray_u8
    img = np.zeros((1, 3, 3), np.float32)
    img[0, 0] = (-1.0, -1.0, -1.0)          # global minimum -> (0,0,0)   -> 0
    img[0, 1] = (1.0, 1.0, 1.0)             # global maximum -> (255,..)  -> 255
    img[0, 2] = (1.0, -1.0, 0.0)            # (255, 0, 128) -> (255*1868 + 128*4899 + 8192) >> 14 = 67
    assert reference_gray_u8(img).tolist() == [[0, 255, 67]]
    assert reference_gray_u8(np.full((2, 2, 3), 0.3, np.float32)).tolist() == [[0, 0], [0, 0]]      # flat image: max == min


# ------------------------------------------------------------------------------------------------- legacy TensorFlow snapshots
def _tf_stub_from_state(sd, cfg):
    """Build the nested TF-network structure of a legacy snapshot from a PyTorch-layout state_dict (test helper: the inverse
    of loader.convert_tf_generator's renaming / transposes / flips / +1)."""
    from morphganformer_amd import loader
    top, mapping, synthesis = [], [], []
    qkv = {"to_queries": "query", "to_keys": "key", "to_values": "value"}

    def att(prefix, rest, v):
        mod, _, kind = rest.rpartition(".")
        if rest == "centroids":
            return prefix + "toasgn_init", v
        if rest == "att_weight":
            return prefix + "iter_0/st_weights", v
        what = qkv.get(mod) or {"from_pos_map": "from_pos", "to_pos_map": "to_pos", "modulation": "out"}[mod]
        return prefix + f"{kind}_{what}", (v.T if kind == "weight" else v)

    for k, v in sd.items():
        v = np.asarray(v)
        if k.endswith(("resample_kernel", "grid_pos")):
            continue
        if k == "pos":
            top.append(("ltnt_emb/emb", v)); continue
        if k == "mapping.w_avg":
            top.append(("dlatent_avg", v)); continue
        p = k.split(".")
        if p[0] == "mapping":
            g = "global/" if p[1] == "global_mlp" else ""
            if p[2] == "out_layer":
                mapping.append((f"{g}Dense3/{p[3]}", v.T if p[3] == "weight" else v))
            elif p[2].startswith("l"):
                mapping.append((f"{g}Dense{p[2][1:]}_{p[3][2:]}/{p[4]}", v.T if p[4] == "weight" else v))
            else:
                mapping.append(att(f"AttLayer_{p[2][2:]}/", ".".join(p[3:]), v))
            continue
        r = int(p[1][1:])
        base = f"{r}x{r}/"
        if p[2] == "const":
            synthesis.append((base + "Const/const", v[None])); continue
        if p[2] == "skip":
            synthesis.append((base + "Skip/weight", v.transpose(2, 3, 1, 0)[::-1, ::-1])); continue
        if p[2] in ("conv0", "conv1"):
            lay = "Conv0_up" if p[2] == "conv0" else ("Conv" if r == 4 else "Conv1")
            rest = ".".join(p[3:])
            if rest == "weight":
                w = v.transpose(2, 3, 1, 0)
                synthesis.append((base + lay + "/weight", w[::-1, ::-1] if p[2] == "conv0" else w))
            elif rest == "biasAct.bias":
                synthesis.append((base + lay + "/bias", v))
            elif rest == "noise_strength":
                synthesis.append((base + lay + "/noise_strength", v))
            elif rest == "noise_const":
                synthesis.append((f"noise{int(math.log2(r)) * 2 - 5 + int(p[2][4])}", v[None, None]))
            elif rest == "affine.weight":
                synthesis.append((base + lay + "/mod_weight", v.T))
            elif rest == "affine.bias":
                synthesis.append((base + lay + "/mod_bias", v - 1))
            else:
                synthesis.append(att(base + lay + "/AttLayer_l2n/", rest[len("transformer."):], v))
            continue
        rest = ".".join(p[3:])
        tf = {("torgb", "weight"): "ToRGB/weight", ("torgb", "biasAct.bias"): "ToRGB/bias", ("torgb", "affine.weight"): "ToRGB/mod_weight",
              ("torgb", "affine.bias"): "ToRGB/mod_bias", ("conv_last", "weight"): "ToRGB/extraLayer/weight",
              ("conv_last", "affine.weight"): "ToRGB/extraLayer/mod_weight", ("conv_last", "affine.bias"): "ToRGB/extraLayer/mod_bias"}[(p[2], rest)]
        if rest == "weight":
            v = v.transpose(2, 3, 1, 0)
        elif rest == "affine.weight":
            v = v.T
        elif rest == "affine.bias":
            v = v - 1
        synthesis.append((base + tf, v))
    kw = dict(latent_size=cfg.z_dim, label_size=0, dlatent_size=cfg.w_dim, components_num=cfg.k - 1, resolution=cfg.img_resolution,
              num_channels=3, mapping_layersnum=cfg.mapping_layers, mapping_lrmul=cfg.mapping_lrmul, mapping_resnet=True,
              mapping_ltnt2ltnt=True, transformer=True, num_heads=1, use_pos=True, fmap_base=cfg.channel_base // 2, fmap_max=cfg.channel_max,
              architecture="resnet", local_noise=True, style=True, start_res=0, end_res=cfg.attn_max_log2res, integration="mul",
              norm="layer", kmeans=True, kmeans_iters=1, pos_type="sinus", pos_init="uniform", pos_directions_num=2)
    S = loader.TFNetworkStub
    return S(version=5, static_kwargs=kw, variables=top,
             components={"mapping": S(version=5, static_kwargs={}, variables=mapping, components={}),
                         "synthesis": S(version=5, static_kwargs={}, variables=synthesis, components={})})


def _mini256():
    from morphganformer_amd.synth_weights import GeneratorConfig
    return GeneratorConfig(img_resolution=256, channel_base=2048, channel_max=32, attn_max_log2res=6, normalize_global=False)


def test_legacy_tf_snapshot_conversion_roundtrip(tmp_path):
    """loader.convert_tf_generator (rules of loader.py:91-247): names, transposes, kernel flips, the +1 on style biases, noise
    indexing -- against a TF-layout snapshot synthesised from a known state_dict, and through the pickle reader."""
    import pickle
    from morphganformer_amd import loader
    from morphganformer_amd.synth_weights import make_state_dict
    cfg = _mini256()
    sd = make_state_dict(cfg, seed=7)
    stub = _tf_stub_from_state(sd, cfg)
    names = dict(loader.collect_tf_params(stub))
    assert "synthesis/4x4/Conv/mod_bias" in names and "synthesis/noise0" in names and "mapping/global/Dense3/weight" in names
    assert names["synthesis/8x8/Conv0_up/weight"].shape == (3, 3, 32, 32)                        # HWIO
    got, gcfg = loader.convert_tf_generator(stub)
    assert gcfg == cfg and list(got) == list(sd)
    for k in sd:
        assert np.array_equal(got[k], np.asarray(sd[k], dtype=np.float32)), k
    # a wrong architecture is refused, not silently mis-loaded
    bad = _tf_stub_from_state(sd, cfg)
    bad["static_kwargs"]["architecture"] = "skip"
    with pytest.raises(NotImplementedError):
        loader.convert_tf_generator(bad)
    # through the file format: a 3-tuple of dnnlib.tflib.network.Network objects
    import sys, types
    mod = types.ModuleType("dnnlib.tflib.network")

    class Network:
        def __init__(self, st):
            self.__dict__.update(st)

        def __getstate__(self):
            return dict(self.__dict__)
    Network.__module__, Network.__qualname__ = "dnnlib.tflib.network", "Network"
    mod.Network = Network
    saved = {k: sys.modules.get(k) for k in ("dnnlib", "dnnlib.tflib", "dnnlib.tflib.network")}
    sys.modules.update({"dnnlib": types.ModuleType("dnnlib"), "dnnlib.tflib": types.ModuleType("dnnlib.tflib"), "dnnlib.tflib.network": mod})
    try:
        def to_net(s_):
            return Network(dict(version=s_["version"], static_kwargs=s_["static_kwargs"], variables=s_["variables"],
                                components={k: to_net(v) for k, v in s_["components"].items()}))
        with open(tmp_path / "tf.pkl", "wb") as f:
            pickle.dump((to_net(stub), to_net(stub), to_net(stub)), f)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    nets = loader.load_network_stubs(str(tmp_path / "tf.pkl"))
    assert set(nets) == {"G", "D", "Gs"} and isinstance(nets["Gs"], loader.TFNetworkStub)
    got2, _ = loader.convert_tf_generator(nets["Gs"])
    assert all(np.array_equal(got2[k], got[k]) for k in got)


@pytest.mark.skipif(not os.path.isdir("/root/reference/torch_utils"), reason="reference tree not present (GPU box)")
def test_legacy_tf_conversion_matches_the_reference_converter():
    """The same synthetic TF snapshot through the REFERENCE's convert_tf_generator (loader.py:91-247)."""
    from morphganformer_amd import loader
    from morphganformer_amd.synth_weights import make_state_dict
    from oracle.make_golden import import_reference
    import_reference()
    import importlib
    ref_loader = importlib.import_module("loader")
    assert ref_loader.__file__.startswith("/root/reference")
    cfg = _mini256()
    sd = make_state_dict(cfg, seed=7)
    stub = _tf_stub_from_state(sd, cfg)

    def to_ref(s_):
        return ref_loader._TFNetworkStub(version=s_["version"], static_kwargs=dict(s_["static_kwargs"]), variables=list(s_["variables"]),
                                         components={k: to_ref(v) for k, v in s_["components"].items()})
    G = ref_loader.convert_tf_generator(to_ref(stub))
    want = {k: v.detach().numpy() for k, v in G.state_dict().items()}
    got, gcfg = loader.convert_tf_generator(stub)
    assert gcfg == cfg
    for k, v in got.items():
        if k.endswith("grid_pos"):
            assert np.allclose(v, want[k], atol=1e-6), k
        else:
            assert np.array_equal(v, want[k]), k
    assert set(want) - set(got) == set() or all("num_batches" in k for k in set(want) - set(got))


def test_merge_files_folds_the_variants_like_the_script(tmp_path):
    """1024_merge_files.py:20-45: <src>/<version>/<variant>/<id>/<name>/<file> -> <dst>/<version>/<id>/<name>/<file>; dotted entries of an id folder are
    skipped; the CLI verb needs neither a model nor a GPU."""
    from morphganformer_amd import cli, drivers
    src, dst = tmp_path / "src", tmp_path / "dst"
    layout = {("v1", "hog", "0001", "a"): ["a_000010_0.1.png", "a.mat"], ("v1", "dlib", "0001", "b"): ["b.mat"],
              ("v1", "dlib", "0002", "c"): ["c.png"], ("v2", "hog", "0001", "a"): ["x.png"]}
    for (ver, var, ident, name), files in layout.items():
        d = src / ver / var / ident / name
        d.mkdir(parents=True)
        for f in files:
            (d / f).write_text(f"{ver}/{var}/{ident}/{name}/{f}")
    (src / "v1" / "hog" / "0001" / "notes.txt").write_text("skipped: carries a dot")
    (src / "v3").mkdir()
    out = drivers.merge_files(str(src), str(dst))
    assert len(out) == 5
    assert sorted(os.listdir(dst / "v1" / "0001")) == ["a", "b"] and os.listdir(dst / "v1" / "0002") == ["c"] and os.listdir(dst / "v3") == []
    assert (dst / "v1" / "0001" / "b" / "b.mat").read_text() == "v1/dlib/0001/b/b.mat" and (dst / "v2" / "0001" / "a" / "x.png").exists()
    assert not (dst / "v1" / "0001" / "notes.txt").exists()
    assert cli.main(["merge-files", "--src", str(src), "--dst", str(tmp_path / "dst2")]) == 0
    assert sorted(os.listdir(tmp_path / "dst2")) == ["v1", "v2", "v3"]


def test_bench_fixed_batch_and_torch_free_launcher():
    """bench.py evaluates a FIXED 32 loop iterations per generator forward (every run -- the driver's, the rocprofv3 trace, the PMC passes --
    launches the same shapes); a bench step is one such launch sequence and exactly --steps of them are timed (no extension of the timed
    region); parsing its arguments / deciding to self-launch must not import torch (the launcher parent stays off the GPU)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, importlib.util; sys.argv = ['bench.py', '--steps', '97']; "
            f"spec = importlib.util.spec_from_file_location('bench_mod', {os.path.join(root, 'bench.py')!r}); "
            "mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod); a = mod.parse(); "
            "assert a.batch == 32 and a.steps == 97 and a.warmup == 5 and a.gpus == 1 and not hasattr(a, 'min_seconds'); "
            "assert 'torch' not in sys.modules and 'numpy' not in sys.modules; print('ok')")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.strip() == "ok", r.stderr[-2000:]


def test_facenet_topology_parameter_count_and_work():
    """The InceptionResnetV1 tables (facenet.py) against the one public known answer that needs no weights: the package's model summary
    counts 27 910 327 parameters with the 8631-way vggface2 classifier head, i.e. 23 482 624 without it (`classify=False`, what
    1024_example_FaceNet_percept.py:30-32 builds).  Plus the work the bench prices the config-3 term with."""
    from morphganformer_amd import facenet as F
    st = F.random_state(0)
    buffers = sum(v.size for k, v in st.items() if "running" in k)
    params = sum(v.size for v in st.values()) - buffers
    assert params == 27_910_327 - (512 * 8631 + 8631) == 23_482_624
    assert len(F.layer_table()) == sum(1 for k in st if k.endswith("conv.weight")) == 111              # BasicConv2d layers (conv + BN + ReLU)
    assert len(F.residual_table()) == sum(1 for k in st if k.endswith("conv2d.weight")) == 21          # the blocks' closing 1x1 convs
    assert abs(F.conv_gflop(160, 160) - 2.833) < 1e-3 and abs(F.conv_gflop(1024, 1024) - 164.62) < 1e-2


def test_lpips_backbone_topologies_against_published_parameter_counts():
    """The three torchvision `.features` stacks LPIPS cuts its taps from (pretrained_networks.py:6-135), restated from the published
    architectures -- weight-free known answers: squeezenet1_1 has 1 235 496 parameters, 513 000 of them in its classifier conv; AlexNet's five
    convolutions 2 469 696; VGG-16's thirteen 14 714 688."""
    from morphganformer_amd.lpips import random_backbone, random_squeeze_backbone
    assert sum(v.size for v in random_squeeze_backbone(0).values()) == 1_235_496 - (512 * 1000 + 1000)
    assert sum(v.size for v in random_backbone("alex", 0).values()) == 2_469_696
    assert sum(v.size for v in random_backbone("vgg", 0).values()) == 14_714_688


def test_cv_resize_linear_u8_known_answers_by_hand():
    """drivers.cv_resize_linear_u8 restates cv2.resize's INTER_LINEAR on uint8 (cv2 is absent: unpinned against the library): answers that
    follow from the published fixed-point scheme -- the same size is the identity; halving an even image is the rounded mean of each 2 x 2
    block ((a + b + c + d + 2) >> 2); doubling [0, 255] gives the half-pixel-centred ramp 0, 64, 191, 255 (63.75 / 191.25 rounded)."""
    from morphganformer_amd.drivers import cv_resize_linear_u8
    rng = np.random.default_rng(2)
    a = rng.integers(0, 256, (10, 14, 3)).astype(np.uint8)
    assert (cv_resize_linear_u8(a, 14, 10) == a).all()
    half = cv_resize_linear_u8(a, 7, 5)
    blocks = a.astype(np.int64).reshape(5, 2, 7, 2, 3).sum((1, 3))
    assert (half == ((blocks + 2) >> 2)).all()
    ramp = cv_resize_linear_u8(np.array([[[0], [255]]], np.uint8), 4, 1)
    assert ramp.reshape(-1).tolist() == [0, 64, 191, 255]
    col = cv_resize_linear_u8(np.array([[[0]], [[255]]], np.uint8), 1, 4)
    assert col.reshape(-1).tolist() == [0, 64, 191, 255]


def test_c_consumer_builds_and_links_against_the_header_and_library(tmp_path):
    """examples/abi_consumer.c compiles as C99 with -Wall -Werror against include/mgf.h and links against the built library; without a GPU it
    stops at mgf_device_ok() with exit code 3 (tests/test_hip_ops.py runs it on the device)."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None or not os.path.exists("/opt/rocm/lib/libamdhip64.so"):
        pytest.skip("gcc / the HIP runtime are not installed")
    libdir = os.path.join(ROOT, "morphganformer_amd")
    exe = str(tmp_path / "abi_consumer")
    subprocess.run(["gcc", "-std=c99", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "abi_consumer.c"),
                    "-o", exe, "-L", libdir, "-lmgf_hip", "-L", "/opt/rocm/lib", "-lamdhip64", "-lm"], check=True)
    import torch
    if not torch.cuda.is_available():
        env = dict(os.environ, LD_LIBRARY_PATH=libdir + ":/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
        r = subprocess.run([exe], env=env, capture_output=True, text=True, timeout=120)
        assert r.returncode == 3 and "no usable gfx950 device" in r.stdout, (r.returncode, r.stdout, r.stderr)
