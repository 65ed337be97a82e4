"""Multi-process (world_size 2, gloo, CPU) tests of the N > 1 path: camera-stream sharding, the fixed-size feature
block layout, and the all-gather exchange of batch-replay mode on host buffers.  On the GPU node the
exchange is ncclAllGather called by liborbx (orbx_replay_*; tests/test_gpu_replay.py) and the layout checked here is the one
orbx_replay_layout reports (ReplayEngine asserts they agree); the extraction itself needs no collective."""
import os
import socket

import numpy as np
import pytest

from orb_slam3_modified_amd import KP_DTYPE
from orb_slam3_modified_amd.replay import BlockLayout, gather_blocks_cpu, shard_streams, unpack_block


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_block(rank, layout):
    """Deterministic per-rank content: frame f of rank r holds n = 5 + 3*r + f keypoints."""
    blk = np.zeros(layout.nbytes, np.uint8)
    kps = blk[:layout.kps_bytes].view(KP_DTYPE).reshape(layout.frames, layout.cap)
    desc = blk[layout.desc_off:layout.desc_off + layout.desc_bytes].reshape(layout.frames, layout.cap, 32)
    counts = blk[layout.counts_off:layout.counts_off + layout.counts_bytes].view(np.int32).reshape(layout.frames, 2)
    for f in range(layout.frames):
        n = 5 + 3 * rank + f
        counts[f] = (n, rank)
        kps["x"][f, :n] = np.arange(n) + 100 * rank
        kps["octave"][f, :n] = f
        desc[f, :n] = (np.arange(n)[:, None] + rank + f) % 256
    return blk


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        layout = BlockLayout(frames=3, cap=40)
        outs = gather_blocks_cpu(_fake_block(rank, layout), layout)
        ok = len(outs) == world
        for r in range(world):
            frames = unpack_block(outs[r], layout)
            for f, (mono, kps, desc) in enumerate(frames):
                n = 5 + 3 * r + f
                ok &= mono == r and len(kps) == n and desc.shape == (n, 32)
                ok &= bool((kps["x"] == np.arange(n) + 100 * r).all()) and bool((kps["octave"] == f).all())
                ok &= bool((desc == (np.arange(n)[:, None] + r + f) % 256).all())
        mine = shard_streams(8, world, rank)
        q.put((rank, ok, mine))
    finally:
        dist.destroy_process_group()


def test_all_gather_of_feature_blocks_world2():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res)
    streams = sorted(s for _, _, mine in res for s in mine)
    assert streams == list(range(8))                      # every camera stream on exactly one rank
    by_rank = dict((r, m) for r, _, m in res)
    assert by_rank[0] == [0, 2, 4, 6] and by_rank[1] == [1, 3, 5, 7]   # stream c -> GPU c mod G


def test_block_layout_is_regular_and_aligned():
    lo = BlockLayout(frames=256, cap=1024)
    assert lo.desc_off % 256 == 0 and lo.counts_off % 256 == 0 and lo.nbytes % 256 == 0
    assert lo.desc_off >= lo.kps_bytes and lo.counts_off >= lo.desc_off + lo.desc_bytes
    assert lo.nbytes == lo.counts_off + ((lo.counts_bytes + 255) // 256) * 256


@pytest.mark.parametrize("n,world", [(8, 1), (8, 2), (8, 4), (8, 8), (5, 3)])
def test_shard_streams_partition(n, world):
    got = sorted(s for r in range(world) for s in shard_streams(n, world, r))
    assert got == list(range(n))
