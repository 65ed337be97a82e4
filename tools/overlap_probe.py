"""Probe: does running two half-batches on two independent contexts / streams (phase-shifted, never joined) beat one
full batch?  The extractor alternates VALU-bound kernels (FAST, blur, descriptors) with latency-bound ones (pyramid
chain, quadtree); two free-running streams can fill each other's idle issue slots."""
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
from orb_slam3_modified_amd import ORBextractor, synth
from orb_slam3_modified_amd.replay import ReplayEngine

B, H, W = 256, 480, 640
dev = torch.device("cuda", 0)
host = synth.make_stream(64, H, W, synth.DEFAULT_SEED)
frames = torch.from_numpy(host[np.arange(B) % 64]).to(dev)

def run(nsplit, steps=30, warm=4, stagger=True):
    per = B // nsplit
    exs = [ORBextractor(1000, 1.2, 8, 20, 7) for _ in range(nsplit)]
    engs = [ReplayEngine(exs[i], frames[i * per:(i + 1) * per], lapping=(0, 1000), gather=False) for i in range(nsplit)]
    for _ in range(warm):
        for e in engs: e.step()
    for e in engs: e.drain()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if stagger and nsplit > 1:
        engs[0].step()   # engine 0 runs one half-batch ahead: phases interleave from then on
    for s in range(steps):
        for i, e in enumerate(engs):
            if stagger and nsplit > 1 and i == 0 and s == steps - 1:
                continue
            e.step()
    for e in engs: e.drain()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    feats = sum(int(e.counts(0)[:, 0].sum()) for e in engs)
    return dt * 1e3, feats / (dt * 1e3)

for nsplit, stagger in ((1, False), (2, False), (2, True), (4, False), (4, True)):
    ms, fpm = run(nsplit, stagger=stagger)
    print(f"contexts {nsplit} stagger {int(stagger)}: {ms:.4f} ms per {B} frames, {fpm/1e3:.1f} k features/ms", flush=True)
