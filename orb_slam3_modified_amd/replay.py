"""Batch-replay driver: camera streams sharded one per GPU, optional RCCL all-gather of the per-frame
feature blocks (BASELINE.json config 5).

One process per GPU (`torch.distributed`, backend "nccl" = RCCL on ROCm; "gloo" for the CPU unit tests of the
sharding / packing logic).  Frames are independent units, so extraction itself needs no collective; the only
exchange is the all-gather that gives every rank all cameras' descriptors, and it is issued asynchronously so it
overlaps the next batch's kernels (double-buffered feature blocks).

Feature block layout per rank and step (one contiguous uint8 buffer, fixed size so the gather is regular):
    [B][cap] orbx_keypoint (28 B) | [B][cap][32] descriptor bytes | [B][2] int32 (n, monoIndex)
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Tuple

import numpy as np

KP_BYTES = 28


def _up256(n: int) -> int:
    return (n + 255) // 256 * 256


@dataclass
class BlockLayout:
    frames: int
    cap: int

    @property
    def kps_bytes(self) -> int: return self.frames * self.cap * KP_BYTES
    @property
    def desc_bytes(self) -> int: return self.frames * self.cap * 32
    @property
    def counts_bytes(self) -> int: return self.frames * 2 * 4
    @property
    def desc_off(self) -> int: return _up256(self.kps_bytes)
    @property
    def counts_off(self) -> int: return self.desc_off + _up256(self.desc_bytes)
    @property
    def nbytes(self) -> int: return self.counts_off + _up256(self.counts_bytes)


def shard_streams(n_streams: int, world_size: int, rank: int) -> List[int]:
    """Camera stream c runs on GPU c mod G."""
    return [c for c in range(n_streams) if c % world_size == rank]


def unpack_block(block: np.ndarray, layout: BlockLayout):
    """Host view of one rank's feature block -> list of (monoIndex, keypoints, descriptors) per frame."""
    from ._lib import KP_DTYPE
    b = np.ascontiguousarray(block).view(np.uint8).reshape(-1)
    kps = b[:layout.kps_bytes].view(KP_DTYPE).reshape(layout.frames, layout.cap)
    desc = b[layout.desc_off:layout.desc_off + layout.desc_bytes].reshape(layout.frames, layout.cap, 32)
    counts = b[layout.counts_off:layout.counts_off + layout.counts_bytes].view(np.int32).reshape(layout.frames, 2)
    return [(int(counts[f, 1]), kps[f, :counts[f, 0]].copy(), desc[f, :counts[f, 0]].copy()) for f in range(layout.frames)]


class ReplayEngine:
    """Per-rank replay loop over device-resident frames with an overlapped all-gather of feature blocks.

    lanes > 1 splits the batch over that many extractor contexts, each on its own free-running stream.  The hot path
    alternates issue-bound kernels (FAST, blur, descriptors) with latency-bound ones (pyramid chain, quadtree); lanes that
    are never joined per step drift out of phase and fill each other's idle issue slots (measured: 2 lanes +7.6 % on
    256 x 640x480, 4 lanes less).  Results are identical: frames are independent and each lane writes its own rows of
    the step's feature block."""

    def __init__(self, extractor, frames_dev, lapping=(0, 1000), gather: bool = True, process_group=None, lanes: int = 1,
                 gather_what: str = "blocks"):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        # gather_what: "blocks" (default: `gathered` holds whole feature blocks, indexable with BlockLayout offsets) or "descriptors"
        # (north_star's exchange, what bench.py asks for: descriptor rows + counts only — read it through gathered_view())
        assert gather_what in ("descriptors", "blocks")
        self.gather_what = gather_what
        self.ex = extractor
        # frames_dev: torch uint8 [B, H, W] on this rank's GPU, or a list of such batches (same shape) that the steps rotate
        # through — step k processes batch k mod len
        self.frame_sets = list(frames_dev) if isinstance(frames_dev, (list, tuple)) else [frames_dev]
        frames_dev = self.frame_sets[0]
        assert all(f.shape == frames_dev.shape and f.stride() == frames_dev.stride() for f in self.frame_sets)
        self.frames = frames_dev
        self.B, self.H, self.W = frames_dev.shape
        self.lap = lapping
        self.layout = BlockLayout(self.B, extractor.capacity)
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.gather = gather and dist.is_initialized()   # a 1-rank group still exercises the collective path (tests)
        self.pg = process_group
        dev = frames_dev.device
        self.blocks = [torch.zeros(self.layout.nbytes, dtype=torch.uint8, device=dev) for _ in range(2)]
        # what every rank receives from every other rank per step: the descriptor rows + the per-frame counts (north_star: "RCCL
        # all-gather of descriptors"; [B][cap][32] + [B][2] int32 — the tail of the block, contiguous), or the whole block with the
        # 28-byte keypoints as well (SURVEY.md §8(e)'s block).  Fixed size, so the collective is regular.
        self.send_off = self.layout.desc_off if gather_what == "descriptors" else 0
        self.send_bytes = self.layout.nbytes - self.send_off
        self.gathered = [torch.zeros(self.send_bytes * self.world, dtype=torch.uint8, device=dev) if self.gather else None
                         for _ in range(2)]
        self.pending = [None, None]
        # transport: RCCL ("nccl") moves device buffers asynchronously on the gather stream; any other backend (gloo: the CPU suite and the
        # two-ranks-on-one-GPU test, where RCCL refuses a shared device) stages through pinned host buffers, synchronously
        self.device_collective = self.gather and dist.get_backend(process_group) == "nccl"
        if self.gather and not self.device_collective:
            pin = dev.type == "cuda"
            self.h_send = torch.zeros(self.send_bytes, dtype=torch.uint8, pin_memory=pin)
            self.h_recv = torch.zeros(self.send_bytes * self.world, dtype=torch.uint8, pin_memory=pin)
        self.gather_events = []   # (start, end) timing events of the collectives since reset_gather_timing()
        self.step_idx = 0
        # Explicit (non-default) streams carry the kernels AND order the collective behind them: the default stream's
        # handle is NULL, which the C ABI reads as "use the context's own stream" — invisible to torch/RCCL.
        lanes = max(1, min(int(lanes), self.B // 32 if self.B >= 64 else 1))
        per = (self.B + lanes - 1) // lanes
        self.lane_ranges = [(j * per, min(self.B, (j + 1) * per)) for j in range(lanes) if j * per < self.B]
        self.exs = [extractor] + [extractor.clone() for _ in self.lane_ranges[1:]]
        if len(self.lane_ranges) > 1:
            # with a second lane filling the idle issue slots, the in-lane forks that pay are different from the single-lane
            # ones.  Measured on all eight combinations, 2 lanes x 128 frames, three repetitions (round 2, after FAST reached
            # full residency): blur forked behind FAST + level-0 FAST beside the pyramid chain + the quadtree as one launch
            # = 0.985 ms per step against 0.999 for round 1's choice (blur in line, quadtree levels split) and 1.03 with no
            # fork at all; ORBX_* environment variables still win
            import os
            for ex in self.exs:
                for name, env, val in (("fork_blur", "ORBX_FORK_BLUR", 1), ("fork_fast0", "ORBX_FORK_FAST0", 1), ("fork_qt", "ORBX_FORK_QT", 0)):
                    if env not in os.environ:
                        ex.set_option(name, val)
        if dev.type == "cuda":   # buffers of every lane now, not inside the first (possibly timed) step
            for ex, (f0, f1) in zip(self.exs, self.lane_ranges):
                ex.reserve(self.H, self.W, f1 - f0)
        cuda = dev.type == "cuda"
        self.streams = [torch.cuda.Stream(device=dev) if cuda else None for _ in self.lane_ranges]
        self.stream = self.streams[0]
        # the collective ALWAYS runs on its own stream behind every lane of the step: on a lane's stream step k + 1's kernels would queue
        # behind step k's collective and the overlap would be gone
        self.gstream = torch.cuda.Stream(device=dev) if cuda else None
        self.lane_done = [[torch.cuda.Event() for _ in self.lane_ranges] for _ in range(2)] if cuda else None

    def step(self):
        """One pass of the hot path over this rank's batch (+ async all-gather of the resulting block)."""
        torch = self.torch
        i = self.step_idx & 1
        blk = self.blocks[i]
        base = blk.data_ptr()
        lo = self.layout
        pend = self.pending[i]
        frames = self.frame_sets[self.step_idx % len(self.frame_sets)]
        for j, (f0, f1) in enumerate(self.lane_ranges):
            with torch.cuda.stream(self.streams[j]):
                if pend is not None:  # the gather that last read this buffer must be done before a lane overwrites it
                    pend.wait()       # (makes this lane's stream wait for the collective)
                fr = frames[f0:f1]
                self.exs[j].extract_batch_device(fr.data_ptr(), f1 - f0, self.H, self.W, frames.stride(1), frames.stride(0),
                                                 base + f0 * lo.cap * KP_BYTES, base + lo.desc_off + f0 * lo.cap * 32,
                                                 base + lo.counts_off + f0 * 8, self.lap, self.streams[j].cuda_stream)
                if self.gather:
                    self.lane_done[i][j].record(self.streams[j])
        self.pending[i] = None
        if self.gather:  # enqueued behind the kernels of this step, overlaps the next step's kernels
            send = blk[self.send_off:]
            with torch.cuda.stream(self.gstream):
                for ev in self.lane_done[i]:
                    self.gstream.wait_event(ev)
                if self.device_collective:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(self.gstream)
                    self.pending[i] = self.dist.all_gather_into_tensor(self.gathered[i], send, group=self.pg, async_op=True)
                    self.pending[i].wait()   # orders the gather stream behind the collective (no host wait)
                    e1.record(self.gstream)
                    if len(self.gather_events) < 4096:
                        self.gather_events.append((e0, e1))
                else:
                    self.h_send.copy_(send, non_blocking=True)
                    self.gstream.synchronize()
                    self._host_all_gather()
                    self.gathered[i].copy_(self.h_recv, non_blocking=True)
        self.step_idx += 1
        return i

    def _host_all_gather(self):
        dist = self.dist
        try:
            dist.all_gather_into_tensor(self.h_recv, self.h_send, group=self.pg)
        except (RuntimeError, AttributeError, NotImplementedError):   # a backend without the flat form
            parts = list(self.h_recv.view(self.world, self.send_bytes).unbind(0))
            dist.all_gather(parts, self.h_send, group=self.pg)

    def reset_gather_timing(self):
        self.gather_events = []

    def gather_ms(self):
        """Average device time of one step's collective (HIP events on the gather stream) since reset_gather_timing(); None without one."""
        if not self.gather_events:
            return None
        self.drain()
        return sum(a.elapsed_time(b) for a, b in self.gather_events) / len(self.gather_events)

    def gathered_view(self, i: int, rank: int):
        """Rank `rank`'s contribution inside gathered buffer i, as (descriptor rows [B][cap][32], counts [B][2]) device views
        (gather_what == "blocks": the whole block as uint8)."""
        lo = self.layout
        part = self.gathered[i][rank * self.send_bytes:(rank + 1) * self.send_bytes]
        if self.gather_what == "blocks":
            return part
        desc = part[:lo.desc_bytes].view(self.B, lo.cap, 32)
        c0 = lo.counts_off - lo.desc_off
        counts = part[c0:c0 + lo.counts_bytes].view(self.torch.int32).reshape(self.B, 2)
        return desc, counts

    def drain(self):
        if self.gstream is not None:
            with self.torch.cuda.stream(self.gstream):
                for i in (0, 1):
                    if self.pending[i] is not None:
                        self.pending[i].wait()
                        self.pending[i] = None
        for st in self.streams + [self.gstream]:
            if st is not None:
                st.synchronize()

    def counts(self, i: int):
        lo = self.layout
        return self.blocks[i][lo.counts_off:lo.counts_off + lo.counts_bytes].view(self.torch.int32).reshape(self.B, 2)


def gather_blocks_cpu(block: np.ndarray, layout: BlockLayout, process_group=None) -> Optional[List[np.ndarray]]:
    """The same exchange on host tensors (gloo): used by the CPU multi-process tests of the packing / sharding."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(process_group)
    t = torch.from_numpy(np.ascontiguousarray(block).view(np.uint8).reshape(-1).copy())
    outs = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(outs, t, group=process_group)
    return [o.numpy() for o in outs]
