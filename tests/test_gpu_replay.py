"""Batch-replay engine on the GPU: the RCCL all-gather path (1-rank NCCL group: same code, same stream ordering as
N > 1) returns exactly the per-rank feature block, and the blocks decode to the oracle's keypoints/descriptors."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("lanes", [1, 2])
def test_replay_engine_gather_matches_block_and_oracle(lanes):
    import torch
    import torch.distributed as dist
    from oracle import pyoracle as po
    from orb_slam3_modified_amd import ORBextractor, synth
    from orb_slam3_modified_amd.replay import ReplayEngine, unpack_block
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        host = synth.make_stream(4)
        nfr = 32 if lanes == 1 else 96            # lanes need >= 32 frames each
        frames = torch.from_numpy(host[np.arange(nfr) % 4]).to(dev)
        ex = ORBextractor(1000, 1.2, 8, 20, 7, device_id=0)
        eng = ReplayEngine(ex, frames, lapping=(0, 1000), gather=True, lanes=lanes)
        assert eng.gather and len(eng.lane_ranges) == lanes
        last = 0
        for _ in range(5):
            last = eng.step()
        eng.drain()
        torch.cuda.synchronize()
        blk = eng.blocks[last].cpu().numpy()
        gat = eng.gathered[last].cpu().numpy()
        assert np.array_equal(blk, gat[:len(blk)])
        res = unpack_block(gat[:eng.layout.nbytes], eng.layout)
        ora = po.OracleExtractor(1000, 1.2, 8, 20, 7)
        for f in (0, 1, 2, 3, 17, 31, nfr // 2 - 1, nfr // 2, nfr - 1):
            okps, odesc, omono = ora.extract(host[f % 4], (0, 1000))
            mono, kps, desc = res[f]
            assert mono == omono and kps.tobytes() == okps.tobytes() and np.array_equal(desc, odesc)
    finally:
        dist.destroy_process_group()
