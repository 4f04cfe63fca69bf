"""Would two half-size candidate batches in flight beat one full batch?  Two independent engines (own generator, own LPIPS, batch B / 2, one-stream graphs) replayed
on two streams, staggered by half an iteration, against one engine at batch B with and without the loss / generator pipeline:   python tools/two_engine_probe.py [B]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from morphganformer_amd.synth_weights import FULL1024

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda", 0)
bench._late_imports()


def rate_single(batch, pipeline, seqs=12):
    sd, G, P, eng, *_ = bench.build(FULL1024, dev, 0, 4096 * batch, True, batch, pipeline=pipeline)
    eng.run(2 * batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.run(seqs * batch)
    torch.cuda.synchronize()
    return seqs * batch / (time.perf_counter() - t0)


def rate_two(batch, seqs=12):
    engs = [bench.build(FULL1024, dev, r, 4096 * batch, True, batch, pipeline=False)[3] for r in range(2)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
    for e, s in zip(engs, streams):
        with torch.cuda.stream(s):
            e.run(2 * batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(seqs):
        for e, s in zip(engs, streams):
            with torch.cuda.stream(s):
                e.run(batch)
    torch.cuda.synchronize()
    return 2 * seqs * batch / (time.perf_counter() - t0)


print(f"one engine, batch {B}, one stream:    {rate_single(B, False):7.1f} iters/s", flush=True)
print(f"one engine, batch {B}, pipelined:     {rate_single(B, True):7.1f} iters/s", flush=True)
print(f"two engines, batch {B // 2} each, two streams: {rate_two(B // 2):7.1f} iters/s", flush=True)
print(f"two engines, batch {B} each, two streams: {rate_two(B):7.1f} iters/s", flush=True)
