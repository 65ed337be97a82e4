"""Batch replay: camera streams sharded one per GPU, one all-gather of the per-step feature blocks (BASELINE.json
config 5).  The engine itself is C++ behind the C ABI (include/orbx.h: orbx_replay_*, csrc/orbx_replay.hip): lanes on their own streams,
double-buffered blocks, ncclAllGather called directly on a gather stream.  This module is the ctypes mirror of it plus the host-side
pieces a Python job adds: which rank am I and how do the 128 bytes of the ncclUniqueId reach the other ranks (torch.distributed when the
job was launched by it — process launch and control plane only — or explicit arguments), and, for a process group without RCCL between its
ranks (gloo: the CPU suite, two ranks sharing one GPU), a host all-gather handed to the engine as a callback.

Frames are independent units, so extraction itself needs no collective; the only exchange is the all-gather that gives every rank all
cameras' descriptors, queued behind the step's kernels so that it overlaps the next step's.

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
    """Per-rank replay loop over device-resident frames with an overlapped all-gather of feature blocks (ctypes mirror of orbx_replay).

    lanes > 1: that many extractor contexts, each on its own free-running stream.  alternate (the default): the lanes take WHOLE steps in turn, so
    every launch covers the whole batch while two steps are in flight (+3.5 % at 256 frames per step); alternate=False: every lane works on its
    share of every step.  The hot path
    alternates issue-bound kernels (FAST, blur, descriptors) with latency-bound ones (pyramid chain, quadtree); lanes that
    are never joined per step drift out of phase and fill each other's idle issue slots (measured: 2 lanes +7.6 % on
    256 x 640x480, 4 lanes less).  Results are identical: frames are independent and each lane writes its own rows of
    the step's feature block.

    frames_dev: one batch or a list of batches the steps rotate through (step k takes batch k mod len); a batch is anything with
    .data_ptr() / .shape == (B, H, W) / .stride() in elements of one byte (a torch uint8 tensor on this rank's GPU) or a tuple
    (device_address, B, H, W, row_stride, frame_stride).
    gather: True -> one exchange per step.  Who the other ranks are: rank / world / unique_id arguments (any launcher), else the
    torch.distributed group this process is in — backend "nccl": rank 0 makes the ncclUniqueId (orbx_replay_unique_id) and broadcasts it, the
    collective is RCCL called by liborbx; any other backend (gloo): the engine stages the block through pinned memory and calls back into
    dist.all_gather — else a one-rank RCCL group (the self-gather)."""

    def __init__(self, extractor, frames_dev, lapping=(0, 1000), gather: bool = True, process_group=None, lanes: int = 1,
                 gather_what: str = "blocks", rank: Optional[int] = None, world: Optional[int] = None, unique_id: Optional[bytes] = None,
                 alternate: Optional[bool] = None):
        import ctypes as C
        import sys
        from . import _lib
        self._C, self._lib = C, _lib
        self._L = L = _lib.lib()
        assert gather_what in ("descriptors", "blocks")
        self.gather_what = gather_what
        self.ex = extractor
        sets = list(frames_dev) if isinstance(frames_dev, list) or (isinstance(frames_dev, tuple) and not isinstance(frames_dev[0], int)) else [frames_dev]
        self._keep = sets                     # the caller's buffers stay alive as long as the engine steps over them
        self.frame_sets = [self._describe(f) for f in sets]
        assert all(f[1:] == self.frame_sets[0][1:] for f in self.frame_sets), "every batch of the rotation must have the same shape and strides"
        _, self.B, self.H, self.W, self._row_stride, self._frame_stride = self.frame_sets[0]
        self.lap = (int(lapping[0]), int(lapping[1]))
        # ---- who are the other ranks
        dist = None
        if rank is None and world is None and "torch" in sys.modules:
            import torch.distributed as _dist
            if _dist.is_available() and _dist.is_initialized():
                dist = _dist
        self.pg = process_group
        if dist is not None:
            rank, world = dist.get_rank(process_group), dist.get_world_size(process_group)
        self.rank, self.world = int(rank or 0), int(world or 1)
        host_cb = None
        use_rccl = bool(gather) and (dist is None or dist.get_backend(process_group) == "nccl")
        if gather and dist is not None and not use_rccl:
            import numpy as _np
            import torch

            def _exchange(_user, send, recv, nbytes):   # orbx_host_exchange_fn over the job's (non-RCCL) process group
                try:
                    s = torch.from_numpy(_np.ctypeslib.as_array((C.c_uint8 * nbytes).from_address(send)))
                    r = torch.from_numpy(_np.ctypeslib.as_array((C.c_uint8 * (nbytes * self.world)).from_address(recv)))
                    try:
                        dist.all_gather_into_tensor(r, s, group=process_group)
                    except (RuntimeError, AttributeError, NotImplementedError):   # a backend without the flat form
                        dist.all_gather(list(r.view(self.world, nbytes).unbind(0)), s, group=process_group)
                    return 0
                except Exception as e:   # noqa: BLE001 — must not unwind through the C frame
                    sys.stderr.write(f"[orbx replay] host all-gather failed: {e}\n")
                    return 1
            host_cb = L.HOST_EXCHANGE_FN(_exchange)
        self._host_cb = host_cb               # keeps the trampoline alive
        if gather and self.world > 1 and unique_id is None and host_cb is None and dist is None:
            raise ValueError("ReplayEngine: world > 1 with gather needs a unique_id (orbx_replay_unique_id on one rank) or a torch.distributed group")
        # ---- lanes: contexts of the caller's extractor's parameters (the first lane IS the caller's context)
        lanes = max(1, min(int(lanes), self.B // 32 if self.B >= 64 else 1))
        per = (self.B + lanes - 1) // lanes
        nl = len([j for j in range(lanes) if j * per < self.B])
        self.exs = [extractor] + [extractor.clone() for _ in range(nl - 1)]
        if alternate is not None:   # lane schedule: True = the lanes take whole steps in turn (the default for >= 2 lanes), False = every lane its share of every step
            extractor.set_option("replay_alternate", 1 if alternate else 0)
        arr = (C.c_void_p * nl)(*[e._ctx for e in self.exs])
        h = C.c_void_p()
        what = 0 if not gather else (1 if gather_what == "descriptors" else 2)
        # ---- two halves (include/orbx.h): prepare = everything this rank can fail at on its own; then every rank learns over the job's control
        # plane whether ALL ranks are ready, and only then does anybody enter ncclCommInitRank (a rank whose peer never arrives waits for ever)
        rc = L.orbx_replay_prepare(C.byref(h), arr, nl, self.B, self.H, self.W, what, self.rank, self.world, 1 if (use_rccl and host_cb is None) else 0,
                                   C.cast(host_cb, C.c_void_p) if host_cb is not None else None, None)
        local_error = None if rc == 0 else f"orbx_replay_prepare failed ({rc}): {L.orbx_last_error(self.exs[0]._ctx).decode()}"
        if gather and dist is not None and self.world > 1:
            box = [None]
            if use_rccl and self.rank == 0 and local_error is None and unique_id is None:
                buf = (C.c_uint8 * 128)()
                rc = L.orbx_replay_unique_id(buf)
                if rc == 0:
                    box[0] = bytes(buf)
                else:
                    local_error = f"orbx_replay_unique_id failed ({rc}): {L.orbx_replay_rccl_info().decode()}"
            states = [None] * self.world
            dist.all_gather_object(states, local_error, group=process_group)     # every rank's verdict on its own half
            bad = [(r_, e_) for r_, e_ in enumerate(states) if e_ is not None]
            if bad:
                if h.value:
                    L.orbx_replay_destroy(h)
                raise _lib.OrbxError(-3, "ReplayEngine: no rank enters the exchange because " +
                                     "; ".join(f"rank {r_}: {e_}" for r_, e_ in bad))
            if use_rccl and unique_id is None:   # RCCL between the ranks, called by liborbx: the id travels over the job's own control plane
                dist.broadcast_object_list(box, src=dist.get_global_rank(process_group, 0) if process_group is not None else 0, group=process_group)
                unique_id = box[0]
        elif local_error is not None:
            raise _lib.OrbxError(rc, local_error)
        if gather and host_cb is None:
            uid = (C.c_uint8 * 128).from_buffer_copy(unique_id) if unique_id is not None else None
            rc = L.orbx_replay_connect(h, uid)
            if rc != 0:
                msg = L.orbx_replay_last_error(h).decode()
                L.orbx_replay_destroy(h)
                raise _lib.OrbxError(rc, msg)
        self._h = h
        fr, cap, nb, do, co, so, sb, nlan = C.c_int(), C.c_int(), C.c_size_t(), C.c_size_t(), C.c_size_t(), C.c_size_t(), C.c_size_t(), C.c_int()
        self._check(L.orbx_replay_layout(h, C.byref(fr), C.byref(cap), C.byref(nb), C.byref(do), C.byref(co), C.byref(so), C.byref(sb), C.byref(nlan)))
        self.layout = BlockLayout(self.B, cap.value)
        assert (self.layout.nbytes, self.layout.desc_off, self.layout.counts_off) == (nb.value, do.value, co.value), "BlockLayout != orbx_replay_layout"
        self.send_off, self.send_bytes = so.value, sb.value
        self.lane_ranges = []
        for j in range(nlan.value):
            f0, f1 = C.c_int(), C.c_int()
            self._check(L.orbx_replay_lane_range(h, j, C.byref(f0), C.byref(f1)))
            self.lane_ranges.append((f0.value, f1.value))
        self._gather_cfg = bool(gather)
        self._gather = bool(gather)
        self.alternate = nlan.value >= 2 and all(rg == (0, self.B) for rg in self.lane_ranges)      # the lanes take whole steps in turn
        self.device_collective = bool(gather) and host_cb is None      # RCCL, asynchronous on the gather stream
        self.transport = L.orbx_replay_transport(h).decode()
        self.step_idx = 0

    @staticmethod
    def _describe(f):
        if isinstance(f, tuple):
            ptr, B, H, W, rs, fs = f
            return (int(ptr), int(B), int(H), int(W), int(rs), int(fs))
        B, H, W = (int(v) for v in f.shape)
        st = f.stride() if callable(getattr(f, "stride", None)) else f.strides
        assert int(st[2]) == 1, "pixels of a row must be contiguous"
        return (int(f.data_ptr()), B, H, W, int(st[1]), int(st[0]))

    def _check(self, rc):
        if rc < 0:
            raise self._lib.OrbxError(rc, self._L.orbx_replay_last_error(self._h).decode())
        return rc

    # ---- the exchange can be switched off / on between steps (bench.py measures the same loop both ways)
    @property
    def gather(self) -> bool:
        return self._gather

    @gather.setter
    def gather(self, on: bool):
        if on and not self._gather_cfg:
            raise ValueError("this engine was created without an exchange")
        self._check(self._L.orbx_replay_set_gather(self._h, 1 if on else 0))
        self._gather = bool(on)

    def step(self) -> int:
        """One pass of the hot path over this rank's batch (+ the asynchronous all-gather of the resulting block).  Returns the buffer index."""
        ptr = self.frame_sets[self.step_idx % len(self.frame_sets)][0]
        i = self._check(self._L.orbx_replay_step(self._h, ptr, self._row_stride, self._frame_stride, self.lap[0], self.lap[1]))
        self.step_idx += 1
        return i

    def drain(self):
        self._check(self._L.orbx_replay_drain(self._h))

    # ---- ordering a consumer on its own stream against ONE step's exchange (no drain): `stream` = a hipStream_t as an integer
    # (torch.cuda.Stream().cuda_stream), 0 / None = the legacy stream is NOT accepted by the library's rule — pass a real stream
    def wait_gathered(self, i: int, stream: int):
        self._check(self._L.orbx_replay_wait_gathered(self._h, i, self._C.c_void_p(int(stream))))

    def release_gathered(self, i: int, stream: int):
        self._check(self._L.orbx_replay_release_gathered(self._h, i, self._C.c_void_p(int(stream))))

    def wait_gathered_host(self, i: int, timeout_ms: int = -1) -> bool:
        """True when the last exchange into gathered buffer i has completed, False after timeout_ms (a peer that left?)."""
        rc = self._L.orbx_replay_wait_gathered_host(self._h, i, int(timeout_ms))
        if rc == -6:
            return False
        self._check(rc)
        return True

    # ---- failure containment: a rank whose lanes failed keeps taking part in every exchange with a poisoned block (all counts -1)
    @property
    def failed(self) -> int:
        return int(self._L.orbx_replay_failed(self._h))

    def abort(self):
        self._check(self._L.orbx_replay_abort(self._h))
        self._gather = False

    def reset_gather_timing(self):
        self._check(self._L.orbx_replay_gather_ms(self._h, None, None, 1))

    def gather_ms(self):
        """Average device time of one step's collective (HIP events on the gather stream) since reset_gather_timing(); None without one."""
        C = self._C
        ms, n = C.c_double(), C.c_longlong()
        self._check(self._L.orbx_replay_gather_ms(self._h, C.byref(ms), C.byref(n), 0))
        return ms.value if n.value > 0 else None

    # ---- buffers: device addresses for callers with their own device code, host copies (numpy) for everybody else
    def block_ptr(self, i: int) -> int:
        p = self._C.c_void_p()
        self._check(self._L.orbx_replay_block(self._h, i, self._C.byref(p)))
        return int(p.value)

    def gathered_ptr(self, i: int, rank: int) -> int:
        p = self._C.c_void_p()
        self._check(self._L.orbx_replay_gathered(self._h, i, rank, self._C.byref(p)))
        return int(p.value)

    def block_host(self, i: int) -> np.ndarray:
        """This rank's feature block i as host bytes (waits for everything in flight); unpack_block() reads it."""
        out = np.empty(self.layout.nbytes, np.uint8)
        self._check(self._L.orbx_replay_read(self._h, 0, i, self._lib.ptr(out), 0, out.nbytes))
        return out

    def write_block(self, i: int, data: np.ndarray):
        data = np.ascontiguousarray(data, np.uint8).reshape(-1)
        assert data.nbytes == self.layout.nbytes
        self._check(self._L.orbx_replay_write_block(self._h, i, self._lib.ptr(data), 0, data.nbytes))

    def gathered_host(self, i: int) -> np.ndarray:
        """Gathered buffer i, [world][send_bytes] host bytes."""
        out = np.empty(self.send_bytes * self.world, np.uint8)
        self._check(self._L.orbx_replay_read(self._h, 1, i, self._lib.ptr(out), 0, out.nbytes))
        return out.reshape(self.world, self.send_bytes)

    def gathered_view(self, i: int, rank: int):
        """Rank `rank`'s contribution inside gathered buffer i, on the host: (descriptor rows [B][cap][32], counts [B][2]) for
        gather_what == "descriptors", the whole block as uint8 for "blocks"."""
        lo = self.layout
        part = np.empty(self.send_bytes, np.uint8)
        self._check(self._L.orbx_replay_read(self._h, 1, i, self._lib.ptr(part), rank * self.send_bytes, part.nbytes))
        if self.gather_what == "blocks":
            return part
        desc = part[:lo.desc_bytes].reshape(self.B, lo.cap, 32)
        c0 = lo.counts_off - lo.desc_off
        counts = part[c0:c0 + lo.counts_bytes].view(np.int32).reshape(self.B, 2)
        return desc, counts

    def counts(self, i: int) -> np.ndarray:
        """[B][2] int32 {n keypoints, monoIndex} of block i (host copy)."""
        lo = self.layout
        out = np.empty(lo.counts_bytes, np.uint8)
        self._check(self._L.orbx_replay_read(self._h, 0, i, self._lib.ptr(out), lo.counts_off, out.nbytes))
        return out.view(np.int32).reshape(self.B, 2)

    def close(self):
        h, self._h = getattr(self, "_h", None), None
        if h is not None and h.value:
            self._L.orbx_replay_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:   # noqa: BLE001
            pass


def gather_blocks_cpu(block: np.ndarray, layout: BlockLayout, process_group=None) -> Optional[List[np.ndarray]]:
    """The same exchange on host tensors (gloo): used by the CPU multi-process tests of the packing / sharding."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(process_group)
    t = torch.from_numpy(np.ascontiguousarray(block).view(np.uint8).reshape(-1).copy())
    outs = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(outs, t, group=process_group)
    return [o.numpy() for o in outs]
