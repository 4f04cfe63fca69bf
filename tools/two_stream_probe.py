"""Experiment: two independent projection engines replayed on two HIP streams (does cross-stream concurrency fill the
memory/epilogue bubbles of the persistent conv kernels?).   python tools/two_stream_probe.py [batch] [launches] [nstreams]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from morphganformer_amd.synth_weights import GeneratorConfig

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 8
launches = int(sys.argv[2]) if len(sys.argv) > 2 else 8
nstreams = int(sys.argv[3]) if len(sys.argv) > 3 else 2
cfg = GeneratorConfig(img_resolution=1024)
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
engs, streams = [], []
for i in range(nstreams):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        _, G, percept, eng, *_ = bench.build(cfg, dev, i, batch * (launches + 4), True, batch)
        eng.run(batch * 2)
    engs.append(eng)
    streams.append(s)
torch.cuda.synchronize()
for mode in ("sequential", "concurrent"):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(launches // 2):
        for eng, s in zip(engs, streams):
            with torch.cuda.stream(s if mode == "concurrent" else streams[0]):
                eng.graph.replay()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    steps = (launches // 2) * nstreams * batch
    print(f"{mode}: {steps} steps in {dt * 1e3:.1f} ms -> {steps / dt:.1f} it/s (batch {batch}, {nstreams} engines, MGF_RESIDENT={os.environ.get('MGF_RESIDENT', 'default')})", flush=True)
