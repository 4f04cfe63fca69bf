"""Benchmark of the hot path: latent-projection iterations/sec @1024^2, k=17 (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W           (N > 1 without a launcher: bench.py starts the N ranks itself, as
                                                             children of a parent that never touches the GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Workload (config 2 of BASELINE.json = configs[1]): one 1024x1024 synthetic target per GPU, literal-mode projection
iteration = perturb latent -> GANformer generator forward (noise_mode="random", like the drivers) -> LPIPS(squeeze)
+ lamda*Wing(injected landmarks) + beta*MSE -> best-so-far selection, all resident on the device and replayed as a hipGraph.
In literal mode the loop's iterations do not depend on each other (the latent never receives a gradient), so the
engine evaluates `--batch` (32) consecutive iterations per generator forward and examines them in order: the result (best step,
best latent, loss history) is the sequential loop's, bit for bit (tests/test_hip_projection.py), and every iteration's full work --
its own noise draws, forward, three losses, selection -- is inside the timed region.

A bench STEP is one pass of the hot path over one batch: one launch sequence of `--batch` loop iterations (one hipGraph replay,
0.044 s; with `--pipeline 1`, the default, the replay scores batch i on a side stream while the generator synthesises batch i + 1 -- K steps are still
K generator forwards and K loss phases, the first generator batch is primed in the warm-up: ProjectionEngine(pipeline=True), same result).  W untimed steps, then EXACTLY K timed steps; `steps` = K, `ms_per_step` = one launch sequence, `value` = loop iterations per
second = K * batch / elapsed (`iters`, `iters_per_step`, `ms_per_iter` spell that out).  `--batch` is fixed, not derived from K, so the
driver's run, the rocprofv3 trace and the PMC passes in profiles/ all launch the same kernels on the same shapes.
Weights are seeded synthetic tensors (no checkpoint exists offline); inputs are resident in HBM before the timed region.
N > 1: one independent target per rank (pair-level sharding, no data-path collective) -> "weak" scaling; the only collective is the
result gather after the timed region.  `--workload config3` is the list-of-targets shape of BASELINE config 3 (see its help).

Rank 0 prints ONE JSON line.  Extra objects (rank 0, N = 1): "roofline" (the MFMA conv kernel with the largest total time: executed
FLOPs / event-measured launch time vs the dense FP32-MFMA peak, PMC traffic and mfma_busy from the committed passes),
"generator_forward", "gradient_mode" (the loss back-propagated into the latent, Adam; one target and --gradient-lockstep targets),
"many_targets" (projections/s set-up included), "objectives" (the LPIPS(vgg) loop and config 3's four-term loop with the FaceNet
embedder), "landmark_callback" (a host detector in the loop), "cpu_baseline" (the CPU oracle's port of the same iteration on the host
cores) -- all reported beside the metric, never in `value`.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

np = torch = None          # imported in main(), AFTER the launcher decision: the parent of a self-launched N-rank run must not touch the GPU

FP32_MFMA_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
_T0 = time.perf_counter()


def log(msg):
    """Progress on stderr (the JSON line on stdout stays alone)."""
    print(f"[bench +{time.perf_counter() - _T0:7.1f}s] {msg}", file=sys.stderr, flush=True)


def host_cores():
    try:
        return len(os.sched_getaffinity(0))
    except AttributeError:
        return os.cpu_count() or 1


def cpu_model():
    """The host CPU's model string (SURVEY 8d asks for model and core count beside the CPU baseline)."""
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or platform.machine() or "unknown"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20,
                    help="timed STEPS: one step = one pass of the hot path over one batch = one launch sequence of --batch loop iterations "
                         "(0.045 s at 1024^2); exactly this many are timed")
    ap.add_argument("--warmup", type=int, default=5, help="untimed steps before the timed region (graph capture included)")
    ap.add_argument("--res", type=int, default=1024, help="debug only; the reported config is 1024")
    ap.add_argument("--batch", type=int, default=32,
                    help="loop steps evaluated per generator forward (exact in literal mode).  Fixed (not derived from --steps) so that every "
                         "run -- the driver's, the rocprofv3 trace, the PMC passes in profiles/ -- launches the same kernels on the same shapes")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--arith", choices=["f32", "bf16x3"], default="f32",
                    help="arithmetic of the generator's convolutions in the TIMED workload.  f32 (default, the headline): exact float32, the "
                         "reference's.  bf16x3: the opt-in mode (three bf16 matrix instructions per product, float32 accumulation) -- the line "
                         "is then labelled with it (`dtype`, `config.arithmetic`) and is NOT the headline metric")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-iters", type=int, default=3, help="timed iterations of the CPU oracle's port (+1 warm-up), about 3 s each on 64 threads")
    ap.add_argument("--biometric", type=int, default=0, metavar="DEPTH",
                    help="add the IResNet-DEPTH embedding-MSE term (BASELINE config 3's full objective); 0 = the config-2 objective")
    ap.add_argument("--lpips-net", choices=["squeeze", "vgg", "alex"], default="squeeze",
                    help="LPIPS backbone of the timed workload: squeeze = configs[1] (...sqz_MSE.py:258-260, the headline); vgg = the net "
                         "1024_example_percept_MSE.py:142-147 scores with; alex = 1024_example_percept_improved.py")
    ap.add_argument("--objectives", type=int, default=1,
                    help="1 = the extra `objectives` legs (rank 0, N=1): the same literal loop with LPIPS(vgg) and with BASELINE config 3's four-term "
                         "objective Wing + FaceNet (InceptionResnetV1 on the un-resized 1024^2 image) + LPIPS(squeeze) + MSE -- iters/s, the "
                         "dominant conv kernel and its executed fraction of the FP32-MFMA peak; 0 = skip")
    ap.add_argument("--objective-batch", type=int, default=16, help="loop steps per generator forward in the `objectives` legs")
    ap.add_argument("--workload", choices=["configs1", "config3"], default="configs1",
                    help="configs1 = the headline (configs[1], one target per rank, weak scaling).  config3 = BASELINE config 3's shape: a LIST of "
                         "independent targets with the four-term objective Wing + FaceNet + LPIPS(squeeze) + MSE, sharded over the ranks by "
                         "drivers.project_many through the dynamic work queue, one result gather at the end -- a weak pass (--config3-targets per "
                         "rank) and a strong pass (--config3-targets in all), per-rank item counts and busy times in the line")
    ap.add_argument("--config3-targets", type=int, default=16, help="targets per rank (weak pass) / in all (strong pass) of --workload config3")
    ap.add_argument("--config3-steps", type=int, default=128, help="loop steps per target of --workload config3 (the drivers run 1000+)")
    ap.add_argument("--pipeline", type=int, default=1,
                    help="1 (default since round 6) = the losses and the selection of batch i run on a side stream while the generator synthesises batch "
                         "i+1 (ProjectionEngine(pipeline=True): the same result bit for bit, +1.5-2 %% iters/s; the roofline leg and a rocprofv3 trace of "
                         "this command see the same two-stream schedule, so their per-kernel durations still agree with each other); 0 = one stream")
    ap.add_argument("--gradient-steps", type=int, default=20,
                    help="steps of the extra gradient-mode leg (loss back-propagated into the latent, Adam; rank 0, N=1 only); 0 = skip")
    ap.add_argument("--gradient-lockstep", type=str, default="8,16,32",
                    help="targets advanced in lockstep in the second half of the gradient-mode leg: a comma-separated list, one sub-leg each (the first "
                         "is reported as `lockstep`, all of them under `lockstep_sweep`); 0 = skip")
    ap.add_argument("--objective-batches", type=str, default="16,32",
                    help="candidates per forward of the config-3 objective leg: the first is `objectives.config3`, the others `objectives.config3_bNN`")
    ap.add_argument("--bf16x3-leg", type=int, default=1,
                    help="1 = the `bf16x3_mode` leg (rank 0, N=1): the headline's loop on a generator in the opt-in bf16x3 arithmetic, reported "
                         "beside the metric with its pixel error against the float32 engine; 0 = skip")
    ap.add_argument("--config4", type=int, default=1,
                    help="1 = the `config4` leg (rank 0, N=1): BASELINE config 4 -- two --target-steps-step projections (one re-targeted engine) and the "
                         "11-alpha sweep of their latents as ONE batch-11 generator forward: projections/s and sweep ms; 0 = skip")
    ap.add_argument("--config5-targets", type=int, default=32,
                    help="targets of the `config5` leg (rank 0, N=1): BASELINE config 5's batch of second-stage projections started from a stage-1 "
                         "latent (edit_MSE.py:229-231), MSE objective like the script, through drivers.project_many on one GPU; 0 = skip")
    ap.add_argument("--config5-steps", type=int, default=500, help="loop steps per second-stage projection of the config5 leg (stated in the line)")
    ap.add_argument("--targets", type=int, default=4,
                    help="targets of the many-target leg (BASELINE configs 3 and 5 are batches of targets): drivers.project_image walked over "
                         "this many 1024^2 targets with ONE engine re-targeted in place, timed end to end INCLUDING the set-up (latent statistics, "
                         "LPIPS workspaces, graph capture); rank 0, N=1 only; 0 = skip")
    ap.add_argument("--target-steps", type=int, default=1000, help="loop steps per target in the many-target leg (config 2: 1000)")
    ap.add_argument("--landmark-callback", choices=["none", "stub"], default="stub",
                    help="extra leg (rank 0, N=1): the same loop with a HOST landmark detector called on every generated image, as the drivers call "
                         "dlib (...sqz_MSE.py:159-170) -- the detector is a no-op stub (dlib is closed / absent), so the figure is what the engine's "
                         "host detour costs: the gray uint8 image built on the device, 1 MB per candidate to pinned host memory, one host->device "
                         "copy of the landmark rows, two hipGraphs per launch sequence")
    ap.add_argument("--force-dist", action="store_true", help="initialise the RCCL process group even for one rank (exercises the N>1 code path)")
    ap.add_argument("--force-launch", action="store_true", help="self-launch through torch.distributed.run even for --gpus 1 (exercises the launcher)")
    ap.add_argument("--selftest-launch", action="store_true", help=argparse.SUPPRESS)      # CPU/gloo dry run of the launcher (tests/)
    return ap.parse_args()


def _late_imports():
    """numpy / torch enter the module only on the worker side (main() after the launcher decision, or a tool importing build())."""
    global np, torch
    if torch is None:
        import numpy
        import torch as _torch
        np, torch = numpy, _torch


def build(cfg, device, rank, steps_total, use_graph, batch, biometric=0, pipeline=False, lpips_net="squeeze", arith="f32"):
    _late_imports()
    from morphganformer_amd.engine import Generator
    from morphganformer_amd.lpips import PerceptualLoss
    from morphganformer_amd.projection import ProjectionArgs, ProjectionEngine, latent_stats, synthetic_landmarks
    from morphganformer_amd.synth_weights import make_state_dict, synthetic_latents
    sd = make_state_dict(cfg, seed=0)
    G = Generator(sd, cfg, device, max_batch=1, arith=arith)
    G.fuse_torgb = True
    z_t = torch.from_numpy(synthetic_latents(cfg, 1, seed=1000 + rank)).to(device)
    target = G(z_t, None, noise_mode="const")[0].clamp(-1, 1).clone()
    gen = torch.Generator(device=device)
    gen.manual_seed(0)
    latent_mean, latent_std = latent_stats(G, 10000, device, gen)
    percept = PerceptualLoss(model="net-lin", net=lpips_net, use_gpu=True, device=device, allow_random_backbone=True)
    lm_t, lm_s = synthetic_landmarks(steps_total, cfg.img_resolution, seed=7 + rank)
    # (seeded random embedder weights give embedding distances far above the drivers' min_loss start of 100)
    args = ProjectionArgs(step=steps_total, min_loss_init=1e30 if biometric else 100.0)
    bio = None
    if biometric:
        from morphganformer_amd.iresnet import BiometricLoss, IResNetEmbedder
        bio = BiometricLoss(IResNetEmbedder(None, depth=biometric, n=batch, device=device, seed=0))
    eng = ProjectionEngine(G, target, latent_mean, latent_std, args, percept=percept, use_mse=True, lm_target=lm_t,
                           lm_steps=lm_s, noise_mode="random", seed=100 + rank, use_graph=use_graph, batch=batch,
                           biometric=bio, gamma=1e-6, pipeline=pipeline)
    return sd, G, percept, eng, target, latent_mean, float(latent_std), (lm_t, lm_s)


def roofline_leg(eng, iters=3, pmc_tag=""):
    """Eager (un-graphed) iterations of the same step with every MFMA conv launch bracketed by HIP events on the launch
    stream, inside the library (mgf_conv_profile_begin/end): the main kernel only, so durations match rocprofv3's trace."""
    from morphganformer_amd import conv as cv
    tensors = [eng.step_ctr, eng.min_loss, eng.best_latent, eng.best_step, eng.losses] + ([eng.gen_ctr] if eng.pipeline else [])
    state = [t.clone() for t in tensors]
    # same schedule as the timed region: in pipelined mode the losses of one batch run on the side stream next to the generator
    step = (lambda k: eng._pipe_step(k & 1)) if eng.pipeline else (lambda k: eng._iteration())
    step(0)
    torch.cuda.synchronize()
    cv.profile_begin()
    for k in range(iters):
        step(k + 1)
    torch.cuda.synchronize()
    prof = cv.profile_end()
    for dst, src in zip(tensors, state):
        dst.copy_(src)
    agg = {}
    for kernel, flops, secs, ksplit, nbytes in prof:
        a = agg.setdefault(kernel, [0.0, 0.0, 0, 0.0])
        a[0] += flops
        a[1] += secs
        a[2] += 1
        a[3] += nbytes
    dom = max(agg, key=lambda k_: agg[k_][1])
    flops, secs, launches, nbytes = agg[dom]
    achieved = flops / secs / 1e12
    per_kernel = {k_: {"launches_per_iter": v[2] // iters, "avg_us": round(v[1] / v[2] * 1e6, 2), "tflops": round(v[0] / v[1] / 1e12, 2)}
                  for k_, v in agg.items()}
    total_conv_s = sum(v[1] for v in agg.values()) / iters
    # `achieved` / `frac` = what the matrix cores EXECUTE (<= 1 by construction): a Winograd F(2x2,3x3) kernel issues 16/36 of the direct
    # 3x3 form's FLOPs as MFMA work, so its executed rate is 4/9 of the algorithmic one.  The SURVEY 8(d) figure -- algorithmic FLOPs of
    # the direct form / launch time, which exceeds the peak for a Winograd kernel -- is reported beside it as algorithmic_achieved /
    # algorithmic_frac (a fraction of peak above 1 is not a roofline fraction).
    ratio = 16 / 36 if dom.startswith("wino") else 1.0
    executed = achieved * ratio
    extra = {"algorithmic_achieved": round(achieved, 2), "algorithmic_frac": round(achieved / FP32_MFMA_PEAK_TFLOPS, 4),
             "executed_over_algorithmic": round(ratio, 4)}
    if dom.startswith("wino"):
        extra["note"] = ("Winograd F(2x2,3x3): achieved/frac = MFMA FLOPs the kernel issues (4/9 of the direct form's) / launch time vs the dense "
                         "FP32-MFMA peak; algorithmic_* count the direct form's FLOPs (SURVEY 8d) and may exceed 1; cf. mfma_busy (SQ counter)")
    per_kernel = {k_: dict(v, executed_frac=round(v["tflops"] * (16 / 36 if k_.startswith("wino") else 1.0) / FP32_MFMA_PEAK_TFLOPS, 4))
                  for k_, v in per_kernel.items()}
    return {"bound": "mfma", "kernel": dom, "achieved": round(executed, 2), "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(executed / FP32_MFMA_PEAK_TFLOPS, 4), **extra, **pmc_fields(dom, eng, pmc_tag, launched=tuple(agg)),
            "algorithmic_bytes_per_launch": round(nbytes / launches),
            "avg_launch_us": round(secs / launches * 1e6, 2), "launches_per_iter": launches // iters,
            "algorithmic_gflop_per_launch": round(flops / launches / 1e9, 3),
            "all_conv_kernels": per_kernel, "conv_ms_per_iter": round(total_conv_s * 1e3, 3)}


def generator_leg(eng, iters=3):
    """The 1024^2 generator forward alone (mapping + synthesis of `batch` candidates, noise_mode="random"), HIP-event timed on
    the launch stream: north-star target ">= 40% of the MFMA roofline on the generator forward"."""
    G, a = eng.G, eng.args
    G.forward_workspace(eng.latent_n, a.truncation_psi, noise_mode="random", lean=True)       # (the workspace flavour the loop itself runs on)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        G.forward_workspace(eng.latent_n, a.truncation_psi, noise_mode="random", lean=True)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters / eng.batch
    gf = G.cfg.conv_gflop()
    return {"gflop_per_image": round(gf, 1), "ms_per_image": round(ms, 4), "tflops": round(gf / ms, 2),
            "frac_of_fp32_mfma_peak": round(gf / ms / FP32_MFMA_PEAK_TFLOPS, 4), "images_per_forward": eng.batch}


def graph_kernel_nodes(graph):
    """Kernel nodes of a captured torch.cuda.CUDAGraph(keep_graph=True) = kernel launches per replay (None when they cannot be counted):
    hipGraphGetNodes / hipGraphNodeGetType on the graph torch kept."""
    if graph is None:
        return None
    try:
        import ctypes
        hip = ctypes.CDLL("libamdhip64.so")
        handle = ctypes.c_void_p(graph.raw_cuda_graph())
        n = ctypes.c_size_t(0)
        if hip.hipGraphGetNodes(handle, None, ctypes.byref(n)) != 0 or n.value == 0:
            return None
        nodes = (ctypes.c_void_p * n.value)()
        if hip.hipGraphGetNodes(handle, nodes, ctypes.byref(n)) != 0:
            return None
        kernels = 0
        for node in nodes:
            t = ctypes.c_int(-1)
            if hip.hipGraphNodeGetType(ctypes.c_void_p(node), ctypes.byref(t)) == 0 and t.value == 0:          # hipGraphNodeTypeKernel
                kernels += 1
        return kernels or None
    except Exception:          # noqa: BLE001 -- a count beside the metric, never a reason to fail
        return None


def gradient_roofline(ge):
    """One eager forward + LPIPS + backward of a gradient-mode engine with every MFMA conv launch (forward and dgrad) bracketed by
    HIP events inside the library, as in roofline_leg: the dominant kernel, its algorithmic TFLOP/s and fraction of the FP32-MFMA peak."""
    from morphganformer_amd import conv as cv
    cv.profile_begin()
    img = ge.gg.forward(ge.latent_n, noise_mode="random")
    ge.percept.distance_into(ge.p_loss, img, keep_taps=True)
    ge.percept.grad_into(ge.dimg, scale=1.0, accumulate=False)
    ge.gg.backward(ge.dimg)
    torch.cuda.synchronize()
    agg = {}
    for kernel, flops, secs, ksplit, nbytes in cv.profile_end():
        a = agg.setdefault(kernel, [0.0, 0.0, 0])
        a[0] += flops
        a[1] += secs
        a[2] += 1
    dom = max(agg, key=lambda k_: agg[k_][1])
    flops, secs, launches = agg[dom]
    achieved = flops / secs / 1e12
    conv_s = sum(v[1] for v in agg.values())
    ratio = 16 / 36 if dom.startswith("wino") else 1.0
    return {"bound": "mfma", "kernel": dom, "achieved": round(achieved * ratio, 2), "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(achieved * ratio / FP32_MFMA_PEAK_TFLOPS, 4),
            "algorithmic_achieved": round(achieved, 2), "algorithmic_frac": round(achieved / FP32_MFMA_PEAK_TFLOPS, 4),
            "launches_per_step": launches, "avg_launch_us": round(secs / launches * 1e6, 2),
            "conv_ms_per_step": round(conv_s * 1e3, 3),
            "all_convs_tflops": round(sum(v[0] for v in agg.values()) / conv_s / 1e12, 2)}


def lockstep_leg(sd, cfg, device, eng, steps, total, B, conv_gf):
    """B independent targets advanced in lockstep through one generator forward/backward per step."""
    from morphganformer_amd.engine import Generator
    from morphganformer_amd.lpips import PerceptualLoss
    from morphganformer_amd.projection import GradientProjectionEngine, ProjectionArgs, synthetic_landmarks
    from morphganformer_amd.synth_weights import synthetic_latents
    torch.cuda.reset_peak_memory_stats(device)
    GB = Generator(sd, cfg, device, max_batch=B)
    zt = torch.from_numpy(synthetic_latents(cfg, B, seed=2000)).to(device)
    tg = torch.cat([GB(zt[j:j + 1], None, noise_mode="const")[0].clamp(-1, 1) for j in range(B)]).contiguous()
    lm = [synthetic_landmarks(total, cfg.img_resolution, seed=50 + j) for j in range(B)]
    gb = GradientProjectionEngine(GB, tg, eng.latent_in[0], 1.0, ProjectionArgs(step=total),
                                  percept=PerceptualLoss(model="net-lin", net="squeeze", use_gpu=True, device=device, allow_random_backbone=True), use_mse=True,
                                  lm_target=np.stack([l[0] for l in lm]), lm_steps=np.stack([l[1] for l in lm]), noise_mode="random",
                                  seed=6, use_graph=True)
    gb.sigma.copy_(eng.sigma[:1].expand(total))
    gb.run(4)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    gb.run(steps)
    torch.cuda.synchronize()
    dtb = time.perf_counter() - t0
    out = {"targets": B, "value": round(B * steps / dtb, 2), "unit": "iters/s (all targets)", "ms_per_step": round(dtb / steps * 1e3, 3),
           "step_tflops": round(2 * conv_gf * B * steps / dtb / 1e3, 2), "hbm_gib": round(torch.cuda.max_memory_allocated(device) / 2 ** 30, 1),
           "roofline": gradient_roofline(gb)}
    del gb, GB, tg
    torch.cuda.empty_cache()
    return out


def gradient_leg(sd, cfg, device, eng, steps, lockstep=(8,)):
    """Extra (not the headline metric): the same objective with the loss back-propagated into the latent and Adam moving it
    (projection.GradientProjectionEngine) -- one candidate per step, forward + LPIPS + backward + Adam as one hipGraph."""
    from morphganformer_amd.engine import Generator
    from morphganformer_amd.projection import GradientProjectionEngine, ProjectionArgs
    G1 = Generator(sd, cfg, device, max_batch=1)
    total = steps + 8
    ge = GradientProjectionEngine(G1, eng.target, eng.latent_in[0], 1.0, ProjectionArgs(step=total), percept=eng.percept, use_mse=True,
                                  lm_target=eng.lm_target.cpu().numpy(), lm_steps=eng.lm_steps[:total].cpu().numpy(), noise_mode="random",
                                  seed=5, use_graph=True)
    ge.sigma.copy_(eng.sigma[:1].expand(total))                    # the run's initial noise level at every step of this short leg
    ge.graph_debug = True                                          # (only so that the launches per step can be counted afterwards)
    ge.run(4)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ge.run(steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # phase split, eager, HIP events on the current stream
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    img = ge.gg.forward(ge.latent_n, noise_mode="random")
    ev[0].record()
    img = ge.gg.forward(ge.latent_n, noise_mode="random")
    ev[1].record()
    ge.percept.distance_into(ge.p_loss, img, keep_taps=True)
    ge.percept.grad_into(ge.dimg, scale=1.0, accumulate=False)
    ev[2].record()
    ge.gg.backward(ge.dimg)
    ev[3].record()
    torch.cuda.synchronize()
    conv_gf = cfg.conv_gflop()
    roof = gradient_roofline(ge)
    launches = graph_kernel_nodes(ge.graph)
    del ge
    torch.cuda.empty_cache()
    sweep = []
    for B in lockstep:
        if B > 1:
            try:
                sweep.append(lockstep_leg(sd, cfg, device, eng, steps, total, B, conv_gf))
                log(f"gradient lockstep {B}: {sweep[-1]['value']} iters/s")
            except Exception as exc:        # noqa: BLE001 -- reported in the line instead
                sweep.append({"targets": B, "error": f"{type(exc).__name__}: {exc}"})
    lock = sweep[0] if sweep else None
    return {"value": round(steps / dt, 2), "unit": "iters/s", "steps": steps, "ms_per_step": round(dt / steps * 1e3, 3),
            "candidates_per_step": 1, "launches_per_step": launches, "lockstep": lock,
            "lockstep_sweep": [{k: v for k, v in r.items() if k != "roofline"} for r in sweep], "generator_forward_ms": round(ev[0].elapsed_time(ev[1]), 3),
            "lpips_forward_backward_ms": round(ev[1].elapsed_time(ev[2]), 3), "generator_backward_ms": round(ev[2].elapsed_time(ev[3]), 3),
            "conv_gflop_forward_plus_dgrad": round(2 * conv_gf, 1),
            "generator_forward_frac_of_fp32_mfma_peak": round(conv_gf / ev[0].elapsed_time(ev[1]) / FP32_MFMA_PEAK_TFLOPS, 4),
            "step_tflops": round(2 * conv_gf * steps / dt / 1e3, 2), "roofline": roof,
            "note": "loss back-propagated into the latent (grad.GeneratorGrad + LPIPS backward + Adam), hipGraph replay; "
                    "the reference loop severs this gradient, so the headline metric stays the literal loop"}


def many_targets_leg(cfg, device, G, percept, batch, n_targets, steps, pipeline=False):
    """Projections per second over a list of targets, set-up included (missing #3): the reference's serial per-image
    loop (projection_example_v2_percept_morph.py:329-365) through drivers.project_image, the first call building the engine (latent
    statistics over 10 000 samples, LPIPS target taps, hipGraph capture), the others re-targeting it in place."""
    from morphganformer_amd import drivers
    from morphganformer_amd.projection import ProjectionArgs, synthetic_landmarks
    from morphganformer_amd.synth_weights import synthetic_latents
    zs = torch.from_numpy(synthetic_latents(cfg, n_targets, seed=5000)).to(device)
    targets = [G(zs[j:j + 1], None, noise_mode="const")[0].clamp(-1, 1) for j in range(n_targets)]
    lms = [synthetic_landmarks(steps, cfg.img_resolution, seed=300 + j) for j in range(n_targets)]
    args = ProjectionArgs(step=steps)
    torch.cuda.synchronize()
    times, eng, res = [], None, []
    t_all = time.perf_counter()
    for j in range(n_targets):
        t0 = time.perf_counter()
        r = drivers.project_image(G, targets[j], lms[j][0], lms[j][1], args=args, percept=percept, batch=batch, seed=40 + j,
                                  engine=eng, return_engine=True, pipeline=pipeline)
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
        eng = r["engine"]
        res.append((r["step"], round(r["loss"], 6)))
    total = time.perf_counter() - t_all
    steady = float(np.mean(times[1:])) if n_targets > 1 else times[0]
    rup = -(-steps // batch) * batch
    # the literal projection as the driver runs it: `steps` loop steps = ceil(steps / batch) launch sequences, the last one ragged (the graph
    # still evaluates `batch` candidates; the steps past `steps` are masked out of the selection) -- so the rate below counts LOOP STEPS, and
    # every re-targeted repeat is listed: the spread between them is the resolution of this box's numbers
    each = times[1:] if n_targets > 1 else times
    rates = [steps / t for t in each]
    return {"targets": n_targets, "steps_per_target": steps, "value": round(n_targets / total, 4), "unit": "projections/s (set-up included)",
            "total_s": round(total, 3), "first_projection_s": round(times[0], 3), "retargeted_projection_s": round(steady, 4),
            "setup_s": round(times[0] - steady, 3) if n_targets > 1 else None,
            "retargeted_iters_per_s": round(steps / steady, 2),
            "retargeted_repeats": {"projection_s": [round(t, 4) for t in each], "iters_per_s": [round(r, 2) for r in rates],
                                   "min": round(min(rates), 2), "max": round(max(rates), 2),
                                   "spread_pct": round(100.0 * (max(rates) - min(rates)) / (sum(rates) / len(rates)), 2),
                                   "launch_sequences": -(-steps // batch), "candidates_evaluated": rup,
                                   "note": f"each repeat = one literal {steps}-step projection of configs[1] through drivers.project_image on the re-targeted "
                                           f"engine (host set-up of the run, {-(-steps // batch)} graph replays incl. the ragged last one, result read-back); "
                                           "iters/s = loop steps / wall time"},
            "best": res,
            "note": "first call = engine set-up (latent statistics, LPIPS workspaces + target taps, hipGraph capture) + the run; the others "
                    "re-target that engine in place (ProjectionEngine.retarget) and replay its graph"}


def config4_leg(cfg, device, G, percept, batch, steps, latent_mean, latent_std, pipeline=False):
    """BASELINE config 4 (1024_merge_morph_2.py:83-92 after two `projection()` calls): two literal-mode projections of `steps` steps with
    configs[1]'s objective through ONE engine (built for the first target, re-targeted for the second), then the 11-alpha sweep
    `(1-a) w1 + a w2`, a = 0, 0.1 .. 1, rendered as ONE batch-11 generator forward.  Timed end to end, engine set-up included."""
    from morphganformer_amd import drivers
    from morphganformer_amd.projection import ProjectionArgs, synthetic_landmarks
    from morphganformer_amd.synth_weights import synthetic_latents
    zs = torch.from_numpy(synthetic_latents(cfg, 2, seed=6000)).to(device)
    targets = [G(zs[j:j + 1], None, noise_mode="const")[0].clamp(-1, 1) for j in range(2)]
    lms = [synthetic_landmarks(steps, cfg.img_resolution, seed=600 + j) for j in range(2)]
    alphas = [round(0.1 * i, 1) for i in range(11)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng, ws, tp = None, [], []
    for j in range(2):
        t1 = time.perf_counter()
        r = drivers.project_image(G, targets[j], lms[j][0], lms[j][1], args=ProjectionArgs(step=steps), percept=percept, batch=batch, seed=60 + j,
                                  latent_mean=latent_mean, latent_std=latent_std, engine=eng, return_engine=True, pipeline=pipeline)
        torch.cuda.synchronize()
        tp.append(time.perf_counter() - t1)
        eng = r["engine"]
        ws.append(r["w"])
    t2 = time.perf_counter()
    lat, imgs = drivers.merge_morph(G, ws[0], ws[1], alphas, truncation_psi=0.7, noise_mode="random", batched=True)
    torch.cuda.synchronize()
    first_sweep = time.perf_counter() - t2
    total = time.perf_counter() - t0
    t3 = time.perf_counter()
    drivers.merge_morph(G, ws[0], ws[1], alphas, truncation_psi=0.7, noise_mode="random", batched=True)       # (workspace of 11 exists now)
    torch.cuda.synchronize()
    sweep = time.perf_counter() - t3
    assert tuple(imgs.shape) == (11, cfg.img_channels, cfg.img_resolution, cfg.img_resolution) and bool(torch.isfinite(imgs).all())
    del eng
    return {"value": round(2 / (tp[0] + tp[1]), 4), "unit": "projections/s (engine set-up included)", "steps_per_projection": steps,
            "projection_s": [round(t, 3) for t in tp], "iters_per_s_retargeted": round(-(-steps // batch) * batch / tp[1], 2),
            "sweep_alphas": 11, "sweep_ms": round(sweep * 1e3, 3), "first_sweep_ms": round(first_sweep * 1e3, 3),
            "sweep_images_per_s": round(11 / sweep, 1), "total_s": round(total, 3),
            "note": "two projections + the 11-alpha sweep as one batch-11 forward (drivers.merge_morph(batched=True)); first_sweep_ms includes "
                    "allocating the batch-11 workspace"}


def config5_leg(cfg, device, G, batch, n_targets, steps, latent_std, pipeline=False):
    """BASELINE config 5 on one GPU: `n_targets` second-stage projections (edit_MSE.py:229-231 -- candidates drawn around a stage-1 latent
    instead of the latent mean; MSE objective and no landmarks, like the script) through drivers.project_many: ONE engine, re-targeted per
    item, results gathered by item id.  Timed end to end after a one-item warm-up (set-up is config 4's and many_targets' figure)."""
    from morphganformer_amd import drivers
    from morphganformer_amd.projection import ProjectionArgs
    from morphganformer_amd.synth_weights import synthetic_latents
    zs = torch.from_numpy(synthetic_latents(cfg, n_targets + 1, seed=8000)).to(device)
    w1 = zs[n_targets] * 0.5                              # the stage-1 result every second stage starts from (`w = w1.reshape([17, 32])`)
    targets = [(lambda j=j: G(zs[j:j + 1], None, noise_mode="const")[0].clamp(-1, 1)) for j in range(n_targets)]
    kw = dict(args=ProjectionArgs(step=steps), percept=None, batch=batch, latent_mean=w1, latent_std=float(latent_std), seed=9, pipeline=pipeline)
    drivers.project_many(G, targets[:1], **kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = drivers.project_many(G, targets, **kw)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert [int(v) for v in res["items"]] == list(range(n_targets))
    rup = -(-steps // batch) * batch
    return {"value": round(n_targets / dt, 4), "unit": "projections/s", "targets": n_targets, "steps_per_target": steps, "seconds": round(dt, 3),
            "iters_per_s": round(n_targets * rup / dt, 2), "objective": "MSE (edit_MSE.py:143)", "steps_per_forward": batch,
            "note": "second-stage projections started from one stage-1 latent, drivers.project_many on one GPU (BASELINE config 5 shards 32 "
                    "of them over 4 GPUs: the same call under torch.distributed)"}


def bf16x3_leg(sd, cfg, device, G32, target, latent_mean, latent_std, batch, min_seconds=0.6):
    """The OPT-IN bf16x3 arithmetic (Generator(arith="bf16x3"): transposed convs and the 3x3 layers below 256^2 as three bf16 matrix
    instructions per product, float32 accumulation) in configs[1]'s literal loop: iters/s beside the headline -- never IN it: the headline's
    arithmetic is the reference's float32 -- and the worst pixel difference against the float32 engine on the same latents and noise."""
    from morphganformer_amd.engine import Generator
    from morphganformer_amd.lpips import PerceptualLoss
    from morphganformer_amd.projection import ProjectionArgs, ProjectionEngine, synthetic_landmarks
    torch.cuda.reset_peak_memory_stats(device)
    Gb = Generator(sd, cfg, device, max_batch=1, arith="bf16x3")
    steps_total = 256 * batch
    percept = PerceptualLoss(model="net-lin", net="squeeze", use_gpu=True, device=device, allow_random_backbone=True)
    lm_t, lm_s = synthetic_landmarks(steps_total, cfg.img_resolution, seed=17)
    eng = ProjectionEngine(Gb, target, latent_mean, latent_std, ProjectionArgs(step=steps_total), percept=percept, use_mse=True, lm_target=lm_t,
                           lm_steps=lm_s, noise_mode="random", seed=21, use_graph=True, batch=batch)
    eng.run(batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.run(batch)
    torch.cuda.synchronize()
    per = time.perf_counter() - t0
    n_seq = max(2, min(200, int(min_seconds / per) + 1))
    t0 = time.perf_counter()
    eng.run(n_seq * batch)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    roof = roofline_leg(eng, iters=1)
    # the same 4 latents and the same (constant) noise through both engines
    z = torch.randn(4, cfg.k, cfg.z_dim, device=device, generator=torch.Generator(device=device).manual_seed(9))
    a = G32(z, None, noise_mode="const")[0]
    b = Gb(z, None, noise_mode="const")[0]
    err = float((a - b).abs().max() / a.abs().max())
    out = {"value": round(n_seq * batch / dt, 2), "unit": "iters/s", "steps": n_seq * batch, "ms_per_step": round(dt / (n_seq * batch) * 1e3, 4),
           "steps_per_forward": batch, "arithmetic": "bf16x3: f32 operands split into 2 bf16 terms, 3 v_mfma_f32_32x32x16_bf16 per product, f32 accumulate "
                                                     "(transposed convs >= 32^2 input, 3x3 layers of the 64^2 / 128^2 blocks); everything else float32",
           "max_pixel_error_vs_f32_engine": float(f"{err:.3e}"), "dominant_kernel": roof["kernel"], "avg_launch_us": roof["avg_launch_us"],
           "conv_ms_per_iter": roof["conv_ms_per_iter"],
           "bf16x3_kernels": {k: v for k, v in roof["all_conv_kernels"].items() if k.endswith(", 1>") and k.count(",") == 5},
           "note": "opt-in engine mode beside the metric; the headline `value` / `dtype` stay exact float32 (the reference's arithmetic)"}
    del eng, percept, Gb
    torch.cuda.empty_cache()
    return out


def objective_leg(cfg, device, G, target, latent_mean, latent_std, batch, lpips_net="squeeze", facenet=False, min_seconds=0.6, pipeline=False):
    """One more objective of the north star through the SAME literal loop, timed like the headline (hipGraph replay, whole launch
    sequences, >= min_seconds): iters/s, HBM in use, and the dominant MFMA conv kernel of its iteration with its executed fraction."""
    from morphganformer_amd.lpips import PerceptualLoss
    from morphganformer_amd.projection import ProjectionArgs, ProjectionEngine, synthetic_landmarks
    torch.cuda.reset_peak_memory_stats(device)
    steps_total = 256 * batch
    percept = PerceptualLoss(model="net-lin", net=lpips_net, use_gpu=True, device=device, allow_random_backbone=True)
    bio = None
    if facenet:
        from morphganformer_amd.iresnet import BiometricLoss
        bio = BiometricLoss("facenet", n=batch, device=device, seed=0)
    lm_t, lm_s = synthetic_landmarks(steps_total, cfg.img_resolution, seed=17)
    eng = ProjectionEngine(G, target, latent_mean, latent_std, ProjectionArgs(step=steps_total), percept=percept, use_mse=True, lm_target=lm_t,
                           lm_steps=lm_s, noise_mode="random", seed=21, use_graph=True, batch=batch, biometric=bio, gamma=1.0, pipeline=pipeline)
    eng.run(batch)                                   # capture + one replay
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.run(batch)
    torch.cuda.synchronize()
    per = time.perf_counter() - t0
    n_seq = max(2, min(200, int(min_seconds / per) + 1))
    t0 = time.perf_counter()
    eng.run(n_seq * batch)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    roof = roofline_leg(eng, iters=1, pmc_tag="vgg_" if lpips_net == "vgg" else "config3_")
    out = {"value": round(n_seq * batch / dt, 2), "unit": "iters/s", "steps": n_seq * batch, "ms_per_step": round(dt / (n_seq * batch) * 1e3, 4),
           "steps_per_forward": batch, "hbm_gib": round(torch.cuda.max_memory_allocated(device) / 2 ** 30, 1),
           "dominant_kernel": roof["kernel"], "executed_tflops": roof["achieved"], "executed_frac": roof["frac"],
           "algorithmic_tflops": roof["algorithmic_achieved"], "avg_launch_us": roof["avg_launch_us"],
           "launches_per_iter": roof["launches_per_iter"], "conv_ms_per_iter": roof["conv_ms_per_iter"],
           "mfma_busy": roof.get("mfma_busy"), "mfma_busy_source": roof.get("mfma_busy_source")}
    del eng, percept, bio
    torch.cuda.empty_cache()
    return out


def landmark_callback_leg(cfg, device, G, percept, target, latent_mean, latent_std, batch, headline):
    """The loop with a host detector in it (ProjectionEngine(landmark_fn=, landmark_input="gray_u8")): iters/s with a no-op detector."""
    from morphganformer_amd.projection import ProjectionArgs, ProjectionEngine, synthetic_landmarks
    steps = 64 * batch
    lm_t, _ = synthetic_landmarks(1, cfg.img_resolution, seed=7)
    calls = [0]

    def stub(gray):                                    # stands where detector(gray, 1) + predictor(gray, rect) stand; touches the image once
        calls[0] += 1
        return lm_t + float(gray[0, 0] & 1)

    eng = ProjectionEngine(G, target, latent_mean, latent_std, ProjectionArgs(step=steps), percept=percept, use_mse=True, lm_target=lm_t,
                           noise_mode="random", seed=11, use_graph=True, batch=batch, landmark_fn=stub, landmark_input="gray_u8")
    eng.run(2 * batch)
    torch.cuda.synchronize()
    timed = 24 * batch
    t0 = time.perf_counter()
    eng.run(timed)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    eng.result()
    v = timed / dt
    return {"value": round(v, 2), "unit": "iters/s", "steps": timed, "ms_per_step": round(dt / timed * 1e3, 4), "detector": "no-op stub",
            "callbacks": calls[0], "input": "gray uint8 [1024,1024] per candidate, built on the device (mgf_reference_gray_u8), pinned host memory",
            "fraction_of_table_mode": round(v / headline, 4),
            "note": "a real detector's own time comes on top (dlib: tens of ms per 1024^2 image on one core)"}


def pmc_tables(tag=""):
    """The committed counter summaries of this workload (profiles/, newest round first): HBM-side bytes per launch and MFMA-pipe
    utilisation per kernel.  Counters cannot be read from inside the process (rocprofv3 --pmc is a separate run, and gpurun forbids
    mixing it with tracing), so bench.py reports the figures of the committed passes and says which file they come from."""
    out = {"traffic": None, "mfma": None}
    for rnd in ("r6", "r5", "r4", "r3", "r2", "r1"):                  # `tag` selects the passes of another workload ("vgg_": the LPIPS(vgg) loop)
        path = os.path.join(ROOT, "profiles", f"{rnd}_{tag}pmc_traffic.json")
        if out["traffic"] is None and os.path.exists(path):
            with open(path) as fh:
                out["traffic"] = (f"profiles/{rnd}_{tag}pmc_traffic.json", json.load(fh))
        path = os.path.join(ROOT, "profiles", f"{rnd}_{tag}pmc_mfma.json")
        if out["mfma"] is None and os.path.exists(path):
            with open(path) as fh:
                out["mfma"] = (f"profiles/{rnd}_{tag}pmc_mfma.json", json.load(fh))
    return out


def pmc_fields(kernel, eng, tag="", launched=()):
    """`traffic` = HBM-side bytes per launch of `kernel` (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes over this same
    workload, corrected as MI355X_MICROARCH.md prescribes: FETCH_SIZE doubled on gfx950, WRITE_SIZE as read; profiles/README.md).
    The passes are taken at `steps_per_forward` candidates per launch; for another count the bytes are scaled by the ratio (every
    operand of these kernels except the weights -- < 1 % of the bytes -- is per candidate) and the line says so.
    `mfma_busy` = SQ_VALU_MFMA_BUSY_CYCLES / (cycles x SIMDs) of the same kernel from the committed MFMA pass."""
    t = pmc_tables(tag)
    out = {"traffic": None}
    if eng.G.cfg.img_resolution != 1024:
        return out
    # The counters are look-ups of committed passes, not measurements of this run: they are only reported while the passes still describe
    # this build -- every MFMA conv kernel the leg just launched must have a row in the table (a renamed, re-templated or new kernel means
    # the passes are stale: re-take them with tools/refresh_profiles.sh).
    for which in ("traffic", "mfma"):
        if t[which] is not None and launched:
            src, table = t[which]
            known = lambda k_: k_ in table or (k_.endswith(">") and any(r.startswith(k_[:-1] + ", ") for r in table))
            missing = sorted(k_ for k_ in launched if not known(k_))
            if missing:
                out["pmc_stale"] = f"{src} has no row for {missing}: counters not reported (re-take the PMC passes)"
                return out
    if t["traffic"] is not None:
        src, table = t["traffic"]
        rec = table.get(kernel)
        b0 = table.get("_meta", {}).get("steps_per_forward")
        if rec is not None and b0:
            scale = eng.batch / b0
            out["traffic"] = round(rec["hbm_bytes"] * scale)
            out["traffic_unit"] = f"bytes/launch (PMC: 2*FETCH_SIZE + WRITE_SIZE, {src}, collected at {b0} steps per forward" + \
                                  ("" if scale == 1 else f", scaled x{scale:.3f} to {eng.batch}") + ")"
    if t["mfma"] is not None:
        src, table = t["mfma"]
        rec = table.get(kernel)
        if rec is None and kernel.endswith(">"):
            # the engine's record names a kernel by its leading template arguments (`pw_conv_kernel<1, 4>`); the counter table has one row per
            # full instantiation (`..., 1, true>`, `..., 1, false>`): busy cycles over cycles of all of them = the time-weighted mean
            rows = [v for k, v in table.items() if k.startswith(kernel[:-1] + ", ") and isinstance(v, dict) and "total_ms" in v]
            if rows:
                rec = {"mfma_busy": round(sum(v["mfma_busy"] * v["total_ms"] for v in rows) / sum(v["total_ms"] for v in rows), 4)}
        if rec is not None:
            out["mfma_busy"] = rec["mfma_busy"]
            out["mfma_busy_source"] = f"SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 x 1024 SIMDs), {src}"
            if "clock_ghz" in rec:
                # the shader clock the kernel actually ran at (GRBM_GUI_ACTIVE / 8 / duration): under FP32-MFMA load the part settles near
                # 2.1 GHz, below the 2.4 GHz the 157.3 TFLOP/s peak is quoted at -- frac ~= mfma_busy x clock / 2.4
                out["clock_ghz"] = rec["clock_ghz"]
    return out


def cpu_baseline_leg(sd, cfg, target, latent_mean, latent_std, lms, iters):
    """The oracle's port of one iteration (what the reference computes per step, both LPIPS branches recomputed)."""
    from oracle.generator_ref import generator_ref, to_torch_state
    from oracle.loss_ref import lpips_ref, mse_ref, squeeze_backbone_random, wing_loss_ref
    cores = min(host_cores(), 64)
    torch.set_num_threads(cores)
    log(f"cpu_baseline: {cores} threads")
    tsd = to_torch_state(sd)
    bb = squeeze_backbone_random(0)
    lin = np.load(os.path.join(ROOT, "morphganformer_amd", "weights", "lpips_lin_squeeze.npz"))
    lins = [torch.from_numpy(lin[f"lin{i}"]) for i in range(7)]
    tgt = target.cpu()
    lm_t, lm_s = lms
    rng = np.random.Generator(np.random.PCG64(1))
    times = []
    with torch.no_grad():
        for i in range(iters + 1):
            t0 = time.perf_counter()
            z = latent_mean.cpu()[None] + torch.from_numpy(rng.standard_normal((1, cfg.k, cfg.z_dim)).astype(np.float32)) * (latent_std * 0.05)
            noises = {}
            for res in cfg.block_resolutions:
                for name in (["conv0"] if res > 4 else []) + ["conv1"]:
                    noises[f"synthesis.b{res}.{name}"] = torch.randn(1, res, res)
            img = generator_ref(tsd, z, cfg, "inject", noises)
            total = float(lpips_ref(bb, lins, img, tgt).sum()) + 0.01 * float(wing_loss_ref(torch.from_numpy(lm_s[i]), torch.from_numpy(lm_t))) \
                + float(mse_ref(img, tgt))
            times.append(time.perf_counter() - t0)
            log(f"cpu_baseline iteration {i}: {times[-1]:.2f} s (loss {total:.4f})")
    per = float(np.mean(times[1:]))
    return {"value": round(1.0 / per, 4), "unit": "iters/s", "cores": cores, "cpu_model": cpu_model(), "host_cores": host_cores(), "kind": "port",
            "s_per_iter": [round(t, 3) for t in times[1:]],
            "sample": f"{iters} timed iterations (+1 warm-up) of the same 1024^2 Wing+LPIPS(squeeze)+MSE step, torch-CPU fp32 oracle, "
                      f"{cores} threads; {per:.2f} s/iter"}


def self_launch(a):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves -- `python -m torch.distributed.run` as a CHILD
    process, from a parent that has made no GPU call (nothing here imports torch; re-exec'ing a process that initialised HIP is what
    this pool forbids) -- pass rank 0's JSON line through, and fail if any rank failed."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", MGF_BENCH_SELF_LAUNCHED="1")
    argv = [x for x in sys.argv[1:] if x != "--force-launch"]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + argv
    log(f"self-launch: {' '.join(cmd)}")
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = []
    for line in proc.stdout:                       # rank 0 prints exactly one line on stdout; everything else goes to stderr
        lines.append(line)
        sys.stdout.write(line)
        sys.stdout.flush()
    rc = proc.wait()
    if rc != 0:
        log(f"self-launch: torch.distributed.run exited with {rc}")
        return rc
    if not any(ln.lstrip().startswith("{") for ln in lines):
        log("self-launch: no JSON line came back from rank 0")
        return 1
    return 0


def sharded_passes(run_many, world, rank, per_rank, steps_per_item, barrier, sync, allgather_f64):
    """The control flow of --workload config3, free of GPU work (tests rehearse it under gloo with a stub `run_many`): a WEAK pass over
    per_rank * world items and a STRONG pass over per_rank items in all, each = barrier, run_many(n_items) (shards the items over the ranks,
    works on them, gathers every record to every rank), barrier; every rank's own busy time and item count are all-gathered afterwards.
    run_many(n) -> dict with `items` (all item ids, ordered) and `mine` (this rank's).  Returns the two result objects (rank-identical)."""
    out = {}
    for name, n_items in (("weak", per_rank * world), ("strong", per_rank)):
        barrier()
        sync()
        t0 = time.perf_counter()
        res = run_many(n_items)
        sync()
        busy = time.perf_counter() - t0
        barrier()
        wall = time.perf_counter() - t0
        items = [int(v) for v in res["items"]]
        assert items == list(range(n_items)), f"{name} pass: the gather must return every item exactly once, in order"
        stats = allgather_f64([busy, float(len(res["mine"])), wall])                    # [world][3]
        walls = [r[2] for r in stats]
        total = max(walls)
        counts = [int(r[1]) for r in stats]
        assert sum(counts) == n_items, (counts, n_items)
        out[name] = {"targets": n_items, "steps_per_target": steps_per_item, "seconds": round(total, 4),
                     "projections_per_s": round(n_items / total, 4), "iters_per_s": round(n_items * steps_per_item / total, 2),
                     "per_rank_targets": counts, "per_rank_busy_s": [round(r[0], 4) for r in stats],
                     "rank_busy_min_s": round(min(r[0] for r in stats), 4), "rank_busy_max_s": round(max(r[0] for r in stats), 4)}
    return out


def config3_workload(a, cfg, device, rank, world, dist):
    """--workload config3 on the GPUs: every rank builds the generator, LPIPS(squeeze), the FaceNet embedder once; items are synthetic
    targets G(z_j) rendered when a rank takes them; drivers.project_many(dynamic=True) walks them with ONE re-targeted engine per rank."""
    from morphganformer_amd import drivers
    from morphganformer_amd.distributed import gather_results
    from morphganformer_amd.engine import Generator
    from morphganformer_amd.iresnet import BiometricLoss
    from morphganformer_amd.lpips import PerceptualLoss
    from morphganformer_amd.projection import ProjectionArgs, latent_stats, synthetic_landmarks
    from morphganformer_amd.synth_weights import make_state_dict, synthetic_latents
    steps, per_rank, batch = a.config3_steps, a.config3_targets, a.objective_batch
    G = Generator(make_state_dict(cfg, seed=0), cfg, device, max_batch=1)
    percept = PerceptualLoss(model="net-lin", net="squeeze", use_gpu=True, device=device, allow_random_backbone=True)
    bio = BiometricLoss("facenet", n=batch, device=device, seed=0)
    gen = torch.Generator(device=device)
    gen.manual_seed(0)
    latent_mean, latent_std = latent_stats(G, 10000, device, gen)
    n_max = per_rank * world
    zs = torch.from_numpy(synthetic_latents(cfg, n_max, seed=7000)).to(device)
    targets = [(lambda j=j: G(zs[j:j + 1], None, noise_mode="const")[0].clamp(-1, 1)) for j in range(n_max)]
    lms = [synthetic_landmarks(steps, cfg.img_resolution, seed=900 + j) for j in range(n_max)]
    kw = dict(args=ProjectionArgs(step=steps), percept=percept, biometric=bio, gamma=1.0, batch=batch, latent_mean=latent_mean,
              latent_std=float(latent_std), seed=3, dynamic=True, use_graph=not a.no_graph)
    run_many = lambda n: drivers.project_many(G, targets[:n], landmarks=lms[:n], **kw)
    run_many(min(2, n_max))                                   # warm-up: engine set-up and graph capture are not what the passes compare
    barrier = dist.barrier if dist is not None else (lambda: None)

    def allgather(vals):
        t = torch.tensor(vals, dtype=torch.float64, device=device)
        if dist is None:
            return [t.tolist()]
        out = torch.empty(world * len(vals), dtype=torch.float64, device=device)
        dist.all_gather_into_tensor(out, t)
        return out.view(world, len(vals)).tolist()

    passes = sharded_passes(run_many, world, rank, per_rank, steps, barrier, torch.cuda.synchronize, allgather)
    gather_ms = None
    if dist is not None:
        lat = torch.zeros(1, cfg.k, cfg.z_dim, device=device)
        gather_results(lat, 0.0, 0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        gather_results(lat, 0.0, 0)
        torch.cuda.synchronize()
        gather_ms = round((time.perf_counter() - t0) * 1e3, 3)
    w = passes["weak"]
    return {"metric": "latent-projection iters/sec @1024^2, k=17 latents", "value": w["iters_per_s"], "unit": "iters/s", "n_gpus": world,
            "steps": w["targets"] * steps, "warmup": 2 * steps, "ms_per_step": round(w["seconds"] / (w["targets"] * steps) * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "hbm_gib": round(torch.cuda.max_memory_allocated(device) / 2 ** 30, 1), "gather_ms": gather_ms,
            "config": {"workload": f"config 3: {per_rank} independent {a.res}x{a.res} targets per GPU ({w['targets']} in all), Wing + FaceNet "
                                   "(InceptionResnetV1, un-resized image) + LPIPS(squeeze) + MSE literal-mode projection, "
                                   f"{steps} steps per target, pair-sharded by drivers.project_many through the dynamic work queue, one result "
                                   "gather; seeded synthetic weights / targets / landmarks", "k": cfg.k, "z_dim": cfg.z_dim,
                       "parallelism": f"pair-sharded x{world} (work queue)", "steps_per_forward": batch},
            "weak": w, "strong": passes["strong"]}


def launch_selftest(a):
    """CPU dry run of the launcher path (tests/): every rank joins a gloo group, the ranks all_gather their ids, rank 0 prints a
    line shaped like the real one.  MGF_SELFTEST_FAIL_RANK makes that rank exit non-zero (the parent must notice)."""
    import torch as th
    import torch.distributed as dist
    from morphganformer_amd.distributed import init_process_group
    init_process_group("gloo")                            # (keeps the store for the work queue: no private torch API)
    rank, world = dist.get_rank(), dist.get_world_size()
    if os.environ.get("MGF_SELFTEST_FAIL_RANK") == str(rank):
        sys.exit(3)
    got = [th.zeros(1, dtype=th.int64) for _ in range(world)]
    dist.all_gather(got, th.tensor([rank]))
    dist.barrier()
    line = {"metric": "launcher selftest", "n_gpus": world, "rccl_ranks": world, "ranks": [int(t) for t in got]}
    if a.workload == "config3":
        # the control flow of --workload config3 with a stub in place of the projection: distributed.run_sharded over the dynamic work
        # queue, ragged per-item cost, slower odd ranks; the same sharded_passes() the GPU path runs
        from morphganformer_amd.distributed import pack_result, run_sharded, unpack_results

        def work(i):
            time.sleep(0.01 * (1 + i % 3) * (1 + 2 * (rank % 2)))       # (odd ranks three times slower: far outside scheduling jitter on a loaded host)
            return pack_result(th.full([1, 2, 3], float(i)), 0.5 * i, i, item=i)

        def run_many(n):
            rows, mine = run_sharded(n, work, 2 * 3 + 3, th.device("cpu"), dynamic=True)
            res = unpack_results(rows, (2, 3))
            assert all(float(res["latents"][j, 0, 0]) == j for j in range(n))
            res["mine"] = mine
            return res

        def allgather(vals):
            out = [th.zeros(len(vals), dtype=th.float64) for _ in range(world)]
            dist.all_gather(out, th.tensor(vals, dtype=th.float64))
            return [o.tolist() for o in out]

        line.update(sharded_passes(run_many, world, rank, a.config3_targets, a.config3_steps, dist.barrier, lambda: None, allgather))
    if rank == 0:
        print(json.dumps(line), flush=True)
    dist.destroy_process_group()
    return 0


def visible_gpus():
    """How many GPUs this process may use, WITHOUT initialising one: torch.cuda.device_count() reads the driver's device list (it honours
    HIP_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES) and makes no HIP context on this image, so the launcher parent stays exec-safe."""
    import torch as _t
    return int(_t.cuda.device_count())


def main():
    a = parse()
    launched = "WORLD_SIZE" in os.environ and "RANK" in os.environ
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # (before anything can bring the HSA runtime up: RCCL needs dmabuf IPC on this pool)
    if not a.selftest_launch:
        have = visible_gpus()
        if a.gpus < 1 or a.gpus > have:
            # said once, by the launcher parent or by every rank the driver's own launcher started, before any rank touches a GPU or
            # waits in a rendezvous for ranks that can never come up
            raise SystemExit(f"bench.py: --gpus {a.gpus} but {have} GPU(s) are visible to this process "
                             f"(HIP_VISIBLE_DEVICES={os.environ.get('HIP_VISIBLE_DEVICES', '<unset>')}): nothing was launched")
    if not launched and (a.gpus > 1 or a.force_launch):
        return self_launch(a)
    if a.selftest_launch:
        return launch_selftest(a)
    _late_imports()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but the launcher started {world} ranks")
    rccl_ranks = None
    if world > 1 or a.force_dist or os.environ.get("MGF_BENCH_SELF_LAUNCHED"):
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if not launched:                                  # --force-dist without a launcher: a process group of one rank
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ.setdefault("MASTER_PORT", str(sk.getsockname()[1]))
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
            os.environ.setdefault("LOCAL_RANK", "0")
        torch.cuda.set_device(local_rank)
    else:
        dist = None
        torch.cuda.set_device(0)
    device = torch.device("cuda", local_rank if world > 1 else 0)
    if dist is not None:
        # RCCL prints a version banner on STDOUT when its first communicator comes up; stdout must carry exactly one JSON line, so file
        # descriptor 1 points at stderr while the group and its first collective are set up
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            from morphganformer_amd.distributed import init_process_group
            init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            # one RCCL collective before anything is timed: every rank contributes its rank id; the count that comes back is what the line reports
            ids = torch.empty(world, dtype=torch.int64, device=device)
            dist.all_gather_into_tensor(ids, torch.tensor([rank], dtype=torch.int64, device=device))
            assert ids.tolist() == list(range(world)), ids.tolist()
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved, 1)
            os.close(saved)
        rccl_ranks = dist.get_world_size()
    from morphganformer_amd.synth_weights import GeneratorConfig
    cfg = GeneratorConfig(img_resolution=a.res)
    if a.workload == "config3":
        line = config3_workload(a, cfg, device, rank, world, dist)
        line["rccl_ranks"] = rccl_ranks
        if rank == 0:
            print(json.dumps(line), flush=True)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return 0
    assert a.batch >= 1 and a.steps >= 1 and a.warmup >= 0
    # A bench STEP is one pass of the hot path over one batch: one launch sequence = `--batch` consecutive loop iterations of the
    # projection (perturb, generator forward, three losses, in-order selection -- each iteration's full work).  W untimed steps, then
    # EXACTLY K timed steps; `value` stays loop iterations per second (K * batch / elapsed).
    seqs, warm = a.steps, a.warmup
    steps = seqs * a.batch                             # loop iterations inside the timed region
    sd, G, percept, eng, target, latent_mean, latent_std, lms = build(cfg, device, rank, (seqs + warm + 4) * a.batch, not a.no_graph, a.batch,
                                                                      a.biometric, bool(a.pipeline), a.lpips_net, a.arith)

    log(f"built generator/LPIPS/engine on {device}; warm-up {warm} steps of {a.batch} loop iterations (+ graph capture)")
    eng.run(max(warm, 1) * a.batch)                    # (graph capture needs one launch sequence even with --warmup 0)
    torch.cuda.synchronize()
    log(f"warm-up done; timing {seqs} steps = {steps} loop iterations")
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.run(steps)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    log(f"timed {seqs} steps ({steps} loop iterations) in {elapsed:.3f} s ({a.batch} iterations per forward, {torch.cuda.max_memory_allocated(device) / 2 ** 30:.1f} GiB of HBM in use)")
    own_elapsed = elapsed
    rank_stats = {"per_rank_iters_per_s": [round(steps / own_elapsed, 3)], "gather_ms": None}
    if dist is not None:
        # every rank's own elapsed time (straggler visibility: the line's value uses the slowest), then the max for the metric
        tl = torch.empty(world, dtype=torch.float64, device=device)
        dist.all_gather_into_tensor(tl, torch.tensor([own_elapsed], dtype=torch.float64, device=device))
        per_rank = [steps / float(v) for v in tl.tolist()]
        elapsed = float(tl.max().item())
        # result gather (the only collective of the path): {latent, best loss, best step} per rank -- timed (second call: the first
        # one may include communicator set-up for this message size)
        from morphganformer_amd.distributed import gather_results
        lat, bstep, bloss, _ = eng.result()
        lat_d = lat.to(device)
        gathered = gather_results(lat_d, bloss, bstep)
        torch.cuda.synchronize()
        tg = time.perf_counter()
        gathered = gather_results(lat_d, bloss, bstep)
        torch.cuda.synchronize()
        rank_stats = {"per_rank_iters_per_s": [round(v, 3) for v in per_rank], "gather_ms": round((time.perf_counter() - tg) * 1e3, 3)}
        assert gathered["latents"].shape[0] == world
    else:
        eng.result()
    pr = rank_stats["per_rank_iters_per_s"]
    rank_stats.update(rank_min=min(pr), rank_max=max(pr), rank_mean=round(sum(pr) / len(pr), 3))

    out = {
        "metric": "latent-projection iters/sec @1024^2, k=17 latents", "value": round(world * steps / elapsed, 3),
        "unit": "iters/s", "n_gpus": world, "steps": seqs, "warmup": warm,
        "ms_per_step": round(elapsed / seqs * 1e3, 4), "iters_per_step": a.batch, "iters": steps, "ms_per_iter": round(elapsed / steps * 1e3, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32" if a.arith == "f32" else "bf16x3 (f32 operands split into 2 bf16 terms, 3 bf16 MFMAs per product, f32 accumulate): NOT the headline arithmetic",
        "data": "synthetic", "rccl_ranks": rccl_ranks, "timed_seconds": round(elapsed, 4), "ranks": rank_stats,
        "hbm_gib": round(torch.cuda.max_memory_allocated(device) / 2 ** 30, 1),
        "config": {"workload": f"configs[1]: single {a.res}x{a.res} face per GPU, Wing+LPIPS({a.lpips_net})+MSE literal-mode projection step, "
                               "noise_mode=random, seeded synthetic weights/targets/landmarks"
                               + (f" + IResNet-{a.biometric} embedding MSE (config 3 objective)" if a.biometric else ""), "k": cfg.k, "z_dim": cfg.z_dim,
                   "targets_per_gpu": 1, "parallelism": f"pair-sharded x{world}", "graph_replay": not a.no_graph,
                   "loss_generator_overlap": bool(a.pipeline), "steps_per_forward": a.batch,
                   "lpips_backbone": f"seeded random {a.lpips_net} weights (torchvision's are a remote fetch) + the reference's vendored lin heads",
                   "timed_region": f"{seqs} steps = {seqs} hipGraph replays of {a.batch} loop iterations each = {steps} iterations"},
    }
    if rank == 0:
        out["roofline"] = roofline_leg(eng)
        out["generator_forward"] = generator_leg(eng)
        log(f"roofline leg done: {out['roofline']['kernel']} {out['roofline']['achieved']} TFLOP/s executed")
        if world == 1 and a.gradient_steps > 0 and not a.biometric:
            try:            # an extra beside the metric: never let it take the JSON line down
                out["gradient_mode"] = gradient_leg(sd, cfg, device, eng, a.gradient_steps, [int(v) for v in str(a.gradient_lockstep).split(",") if int(v) > 1])
                log(f"gradient-mode leg done: {out['gradient_mode']['value']} iters/s")
            except Exception as exc:        # noqa: BLE001 -- reported in the line instead
                out["gradient_mode"] = {"error": f"{type(exc).__name__}: {exc}"}
                log(f"gradient-mode leg failed: {exc}")
        if world == 1 and a.targets > 0 and not a.biometric and a.res == 1024:
            try:
                out["many_targets"] = many_targets_leg(cfg, device, G, percept, a.batch, a.targets, a.target_steps, None if a.pipeline else False)
                log(f"many-target leg done: {out['many_targets']['value']} projections/s")
            except Exception as exc:        # noqa: BLE001 -- reported in the line instead
                out["many_targets"] = {"error": f"{type(exc).__name__}: {exc}"}
                log(f"many-target leg failed: {exc}")
        if world == 1 and a.config4 and not a.biometric and a.res == 1024:
            try:
                out["config4"] = config4_leg(cfg, device, G, percept, a.batch, a.target_steps, latent_mean, latent_std, None if a.pipeline else False)
                log(f"config4 leg done: {out['config4']['value']} projections/s, sweep {out['config4']['sweep_ms']} ms")
            except Exception as exc:        # noqa: BLE001 -- reported in the line instead
                out["config4"] = {"error": f"{type(exc).__name__}: {exc}"}
                log(f"config4 leg failed: {exc}")
        if world == 1 and a.config5_targets > 0 and not a.biometric and a.res == 1024:
            try:
                out["config5"] = config5_leg(cfg, device, G, a.batch, a.config5_targets, a.config5_steps, latent_std, None if a.pipeline else False)   # (the drivers' own policy: MSE-only objective -> one stream)
                log(f"config5 leg done: {out['config5']['value']} projections/s")
            except Exception as exc:        # noqa: BLE001 -- reported in the line instead
                out["config5"] = {"error": f"{type(exc).__name__}: {exc}"}
                log(f"config5 leg failed: {exc}")
        if world == 1 and a.objectives and not a.biometric and a.res == 1024 and a.lpips_net == "squeeze":
            out["objectives"] = {}
            c3 = ("Wing + FaceNet embedding MSE (InceptionResnetV1, un-resized 1024^2 image) + LPIPS(squeeze) + MSE: BASELINE config 3's objective "
                  "(1024_example_FaceNet_percept.py:147-158 + ...sqz_MSE.py:171-179)")
            obs = [int(v) for v in str(a.objective_batches).split(",")] if a.objective_batches else [a.objective_batch]
            legs = [("lpips_vgg", obs[0], dict(lpips_net="vgg"), "Wing + LPIPS(vgg) + MSE (1024_example_percept_MSE.py:142-147's backbone in configs[1]'s loop)"),
                    ("config3", obs[0], dict(facenet=True), c3)] + [(f"config3_b{b}", b, dict(facenet=True), c3) for b in obs[1:]]
            for name, ob, kw, what in legs:
                try:
                    out["objectives"][name] = dict(objective_leg(cfg, device, G, target, latent_mean, latent_std, ob, **kw), objective=what)   # (one stream: the heavy loss phases lose 1 - 1.5 % to the pipeline, tools/pipeline_legs_ab.sh)
                    log(f"objective leg {name}: {out['objectives'][name]['value']} iters/s")
                except Exception as exc:        # noqa: BLE001 -- reported in the line instead
                    out["objectives"][name] = {"error": f"{type(exc).__name__}: {exc}"}
                    log(f"objective leg {name} failed: {exc}")
        if world == 1 and a.bf16x3_leg and a.arith == "f32" and not a.biometric and a.res == 1024 and a.lpips_net == "squeeze":
            try:
                out["bf16x3_mode"] = bf16x3_leg(sd, cfg, device, G, target, latent_mean, latent_std, a.batch)
                log(f"bf16x3 leg done: {out['bf16x3_mode']['value']} iters/s, pixel error {out['bf16x3_mode']['max_pixel_error_vs_f32_engine']}")
            except Exception as exc:        # noqa: BLE001 -- reported in the line instead
                out["bf16x3_mode"] = {"error": f"{type(exc).__name__}: {exc}"}
                log(f"bf16x3 leg failed: {exc}")
        if world == 1 and a.landmark_callback == "stub" and not a.biometric and a.res == 1024:
            try:
                out["landmark_callback"] = landmark_callback_leg(cfg, device, G, percept, target, latent_mean, latent_std, a.batch, out["value"])
                log(f"landmark-callback leg done: {out['landmark_callback']['value']} iters/s")
            except Exception as exc:        # noqa: BLE001 -- reported in the line instead
                out["landmark_callback"] = {"error": f"{type(exc).__name__}: {exc}"}
                log(f"landmark-callback leg failed: {exc}")
        if world == 1 and not a.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline_leg(sd, cfg, target, latent_mean, latent_std, lms, a.cpu_iters)
            except Exception as exc:        # noqa: BLE001 -- the measured GPU line is still printed, with the failure recorded
                out["cpu_baseline"] = {"value": None, "error": f"{type(exc).__name__}: {exc}"}
                log(f"cpu baseline leg failed: {exc}")
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
