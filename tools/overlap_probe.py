"""Probe: lanes as half-batches of every step vs lanes as alternating whole steps (two full-batch pipelines in flight)."""
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

def tune(ex):
    for n, v in (("fork_blur", 0), ("fork_fast0", 1), ("fork_qt", 1)):
        ex.set_option(n, v)

def run_alt(nctx, steps=40, warm=6, tuned=True):
    exs = [ORBextractor(1000, 1.2, 8, 20, 7) for _ in range(nctx)]
    if tuned:
        [tune(e) for e in exs]
    engs = [ReplayEngine(e, frames, lapping=(0, 1000), gather=False) for e in exs]
    for s in range(warm):
        engs[s % nctx].step()
    for e in engs: e.drain()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(steps):
        engs[s % nctx].step()
    for e in engs: e.drain()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3

def run_lanes(lanes, steps=40, warm=6):
    ex = ORBextractor(1000, 1.2, 8, 20, 7)
    eng = ReplayEngine(ex, frames, lapping=(0, 1000), gather=False, lanes=lanes)
    for _ in range(warm): eng.step()
    eng.drain(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps): eng.step()
    eng.drain(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3

for rep in range(2):
    print(f"lanes=2 (half batches)        : {run_lanes(2):.4f} ms/step", flush=True)
    print(f"alternating whole steps, 2 ctx: {run_alt(2):.4f} ms/step", flush=True)
    print(f"alternating whole steps, 3 ctx: {run_alt(3):.4f} ms/step", flush=True)
    print(f"alternating, 2 ctx, untuned   : {run_alt(2, tuned=False):.4f} ms/step", flush=True)
