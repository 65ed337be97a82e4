"""Batch-replay engine on the GPU (the C ABI's orbx_replay_* through its ctypes mirror, and once from a plain C++ host): the RCCL
all-gather path (one-rank group: same code, same stream ordering as N > 1) returns exactly the per-rank feature block, and the blocks
decode to the oracle's keypoints / descriptors."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("lanes,what", [(1, "blocks"), (2, "blocks"), (2, "descriptors")])
def test_replay_engine_gather_matches_block_and_oracle(lanes, what):
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
        eng = ReplayEngine(ex, frames, lapping=(0, 1000), gather=True, lanes=lanes, gather_what=what)
        assert eng.gather and eng.device_collective and len(eng.lane_ranges) == lanes
        last = 0
        for _ in range(5):
            last = eng.step()
        eng.drain()
        torch.cuda.synchronize()
        assert eng.gather_ms() is not None and eng.gather_ms() > 0
        assert "ncclAllGather" in eng.transport and "rccl" in eng.transport, eng.transport
        blk = eng.block_host(last)
        gat = eng.gathered_host(last).reshape(-1)
        assert len(gat) == eng.send_bytes and np.array_equal(blk[eng.send_off:], gat)
        if what == "descriptors":   # descriptor rows + counts travel; the keypoints stay on their rank
            desc, counts = eng.gathered_view(last, 0)
            lo = eng.layout
            assert np.array_equal(desc.reshape(-1), blk[lo.desc_off:lo.desc_off + lo.desc_bytes])
            assert np.array_equal(counts, eng.counts(last))
            gat = blk
        res = unpack_block(gat[:eng.layout.nbytes], eng.layout)
        ora = po.OracleExtractor(1000, 1.2, 8, 20, 7)
        for f in (0, 1, 2, 3, 17, 31, nfr // 2 - 1, nfr // 2, nfr - 1):
            okps, odesc, omono = ora.extract(host[f % 4], (0, 1000))
            mono, kps, desc = res[f]
            assert mono == omono and kps.tobytes() == okps.tobytes() and np.array_equal(desc, odesc)
    finally:
        dist.destroy_process_group()


def test_replay_engine_without_a_process_group_self_gather_and_toggle():
    """No torch.distributed anywhere: gather=True alone makes the one-rank RCCL group inside liborbx (ncclGetUniqueId + ncclCommInitRank(1)).
    The exchange can be switched off and on between steps; an engine created without one refuses to switch it on."""
    import torch
    from orb_slam3_modified_amd import ORBextractor, OrbxError, synth
    from orb_slam3_modified_amd.replay import ReplayEngine
    dev = torch.device("cuda", 0)
    host = synth.make_stream(4)
    frames = torch.from_numpy(host[np.arange(64) % 4]).to(dev)
    ex = ORBextractor(1000, 1.2, 8, 20, 7, device_id=0)
    eng = ReplayEngine(ex, frames, lapping=(0, 1000), gather=True, lanes=2, gather_what="descriptors")
    assert eng.world == 1 and eng.rank == 0 and eng.device_collective and len(eng.lane_ranges) == 2
    for _ in range(3):
        i = eng.step()
    blk = eng.block_host(i)
    assert np.array_equal(eng.gathered_host(i)[0], blk[eng.send_off:])
    assert eng.gather_ms() > 0
    eng.reset_gather_timing()
    eng.gather = False
    keep = eng.gathered_host(i ^ 1).copy()
    j = eng.step()                       # no collective: the gathered buffer of this parity keeps what it had
    assert j == (i ^ 1) and eng.gather_ms() is None and np.array_equal(eng.gathered_host(j), keep)
    eng.gather = True
    k = eng.step()
    assert np.array_equal(eng.gathered_host(k)[0], eng.block_host(k)[eng.send_off:]) and eng.gather_ms() > 0
    eng.close()
    plain = ReplayEngine(ex, frames, lapping=(0, 1000), gather=False, lanes=1)
    assert plain.transport == "none"
    with pytest.raises(ValueError):
        plain.gather = True
    with pytest.raises(OrbxError):
        plain.gathered_host(0)
    plain.close()


def test_replay_from_a_plain_cpp_host(tmp_path):
    """include/orbx.h alone, no Python, no torch: tests/support/replay_host.cpp (the loop INTEGRATION.md section 8 shows) creates two lanes and a
    one-rank engine, steps over device-resident frames with the RCCL self-gather, and prints a digest of block and gathered buffer per step;
    the digests must equal the ones the ctypes mirror produces for the same frames."""
    import subprocess
    import torch
    from orb_slam3_modified_amd import ORBextractor, synth
    from orb_slam3_modified_amd.replay import ReplayEngine

    def digest(b):   # replay_host.cpp's: sum of 8-byte words w_k * (2k + 1) modulo 2^64
        b = np.ascontiguousarray(b, np.uint8).reshape(-1)
        b = np.concatenate([b, np.zeros((-len(b)) % 8, np.uint8)]).view("<u8")
        with np.errstate(over="ignore"):
            return int((b * (2 * np.arange(len(b), dtype=np.uint64) + np.uint64(1))).sum(dtype=np.uint64))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "replay_host")
    pkg = os.path.join(root, "orb_slam3_modified_amd")
    subprocess.check_call(["hipcc", "-O2", "-std=c++17", "-I", os.path.join(root, "include"), os.path.join(root, "tests", "support", "replay_host.cpp"), "-o", exe,
                           "-L", pkg, "-lorbx", "-Wl,-rpath," + pkg])
    host = synth.make_stream(4)
    nfr, steps = 64, 4
    frames_host = np.ascontiguousarray(host[np.arange(nfr) % 4])
    raw = str(tmp_path / "frames.u8")
    frames_host.tofile(raw)
    out = subprocess.check_output([exe, raw, str(nfr), "480", "640", str(steps)], timeout=600).decode().strip().splitlines()
    while out and not out[0].startswith("transport "):      # RCCL prints its version banner to stdout when the communicator is made
        out.pop(0)
    assert out and out[0].startswith("transport ncclAllGather"), out[:3]
    dev = torch.device("cuda", 0)
    eng = ReplayEngine(ORBextractor(1000, 1.2, 8, 20, 7, device_id=0), torch.from_numpy(frames_host).to(dev), lapping=(0, 1000), gather=True, lanes=2,
                       gather_what="blocks")
    want = []
    for s in range(steps):
        i = eng.step()
        b = eng.block_host(i)
        want.append(f"step {s} buffer {i} block {digest(b):016x} gathered {digest(eng.gathered_host(i)):016x} keypoints {int(eng.counts(i)[:, 0].sum())}")
    assert out[1:1 + steps] == want, (out, want)
    assert out[1 + steps].startswith("gather_ms ") and float(out[1 + steps].split()[1]) > 0
    assert out[2 + steps] == f"consumer stream without drain: {steps} of {steps} copies equal the gathered buffer", out[2 + steps]
    assert out[-1] == "ok"


def _check_every_frame(block, layout, host, ora, lap):
    from orb_slam3_modified_amd.replay import unpack_block
    res = unpack_block(block, layout)
    cache = {}
    for f in range(layout.frames):
        key = f % len(host)
        if key not in cache:
            cache[key] = ora.extract(host[key], lap)
        okps, odesc, omono = cache[key]
        mono, kps, desc = res[f]
        assert mono == omono and kps.tobytes() == okps.tobytes() and np.array_equal(desc, odesc), f"frame {f}"


def test_the_configuration_bench_times_is_bit_exact_on_every_frame():
    """bench.py's replay: 2 lanes x 130 frames (past the `small batch` cut-off of the in-lane forks, orbx_extractor.hip), the lanes'
    options fork_fast0 = 1 / fork_blur = 1 / fork_qt = 0 as ReplayEngine sets them, rotating batches, several steps in flight —
    EVERY frame of the last two steps against the oracle."""
    import torch
    from oracle import pyoracle as po
    from orb_slam3_modified_amd import ORBextractor, synth
    from orb_slam3_modified_amd.replay import ReplayEngine
    dev = torch.device("cuda", 0)
    host = synth.make_stream(24)
    nfr = 260
    sets_host = [host[(np.arange(nfr) + 7 * k) % 24] for k in range(3)]
    sets = [torch.from_numpy(h).to(dev) for h in sets_host]
    ex = ORBextractor(1000, 1.2, 8, 20, 7, device_id=0)
    eng = ReplayEngine(ex, sets, lapping=(0, 1000), gather=False, lanes=2, alternate=False)     # the SPLIT schedule: every lane its share of every step
    assert len(eng.lane_ranges) == 2 and eng.lane_ranges[0][1] - eng.lane_ranges[0][0] == 130 and not eng.alternate
    ora = po.OracleExtractor(1000, 1.2, 8, 20, 7)
    last = 0
    for _ in range(7):
        last = eng.step()
    eng.drain()
    torch.cuda.synchronize()
    for back in (0, 1):     # the last step and the one before it (the other block buffer)
        k = (eng.step_idx - 1 - back) % 3
        blk = eng.block_host(last ^ back)
        cache_host = sets_host[k]
        from orb_slam3_modified_amd.replay import unpack_block
        res = unpack_block(blk, eng.layout)
        memo = {}
        for f in range(nfr):
            key = int((f + 7 * k) % 24)
            if key not in memo:
                memo[key] = ora.extract(host[key], (0, 1000))
            okps, odesc, omono = memo[key]
            mono, kps, desc = res[f]
            assert mono == omono and kps.tobytes() == okps.tobytes() and np.array_equal(desc, odesc), (back, f)
        assert cache_host.shape[0] == nfr


@pytest.mark.parametrize("opts", [dict(fork_fast0=1, fork_blur=0, fork_qt=0), dict(fork_fast0=0, fork_blur=1, fork_qt=0),
                                  dict(fork_fast0=0, fork_blur=0, fork_qt=1), dict(fork_fast0=1, fork_blur=1, fork_qt=1),
                                  dict(fork_fast0=1, fork_blur=0, fork_qt=1),
                                  dict(fork_fast0=0, fork_blur=0, fork_qt=0)])
def test_each_stream_fork_alone_is_bit_exact_on_every_frame(opts):
    """The in-context stream forks (level-0 FAST beside the pyramid chain, blur behind FAST, quadtree level groups) one at a
    time on ONE context with a batch large enough for them to engage: results never depend on the launch shape."""
    import torch
    from oracle import pyoracle as po
    from orb_slam3_modified_amd import ORBextractor, synth
    from orb_slam3_modified_amd.replay import BlockLayout
    dev = torch.device("cuda", 0)
    host = synth.make_stream(12)
    nfr = 132
    frames = torch.from_numpy(host[np.arange(nfr) % 12]).to(dev)
    ex = ORBextractor(1000, 1.2, 8, 20, 7, device_id=0)
    for k, v in opts.items():
        ex.set_option(k, v)
    lo = BlockLayout(nfr, ex.capacity)
    blk = torch.zeros(lo.nbytes, dtype=torch.uint8, device=dev)
    st = torch.cuda.Stream(device=dev)
    for _ in range(3):
        ex.extract_batch_device(frames.data_ptr(), nfr, 480, 640, frames.stride(1), frames.stride(0), blk.data_ptr(), blk.data_ptr() + lo.desc_off,
                                blk.data_ptr() + lo.counts_off, (0, 1000), st.cuda_stream)
    st.synchronize()
    _check_every_frame(blk.cpu().numpy(), lo, host, po.OracleExtractor(1000, 1.2, 8, 20, 7), (0, 1000))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _world2_worker(rank, world, port, what, q):
    """One rank of a 2-rank job; both ranks share cuda:0 (RCCL refuses a shared device, so the exchange takes ReplayEngine's host-staged
    transport over gloo — everything else is the engine bench.py runs: lanes, rotating batches, double-buffered blocks, the gather
    ordered behind the lanes of its step)."""
    import hashlib
    import torch
    import torch.distributed as dist
    from oracle import pyoracle as po
    from orb_slam3_modified_amd import ORBextractor, synth
    from orb_slam3_modified_amd.replay import ReplayEngine, shard_streams, unpack_block
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(0)
        B, nsets, steps = 64, 3, 6
        cams = shard_streams(world, world, rank)           # stream c -> GPU c mod G: one camera per rank here
        assert cams == [rank]
        host = {r: [synth.make_stream(B, 480, 640, synth.DEFAULT_SEED + 1000 * r + 101 * k) for k in range(nsets)] for r in range(world)}
        sets = [torch.from_numpy(h).to(dev) for h in host[rank]]
        ex = ORBextractor(1000, 1.2, 8, 20, 7, device_id=0)
        eng = ReplayEngine(ex, sets, lapping=(0, 1000), gather=True, lanes=2, gather_what=what)
        assert eng.gather and not eng.device_collective and eng.world == world and len(eng.lane_ranges) == 2 and "host all-gather" in eng.transport
        ora = po.OracleExtractor(1000, 1.2, 8, 20, 7)
        lo = eng.layout
        ok, notes = True, []
        for step in range(steps):
            i = eng.step()
            if step < steps - 2:
                continue                                   # let the double buffers be reused before anything is checked
            eng.drain()
            torch.cuda.synchronize()
            k = step % nsets
            mine = eng.block_host(i)
            gat = eng.gathered_host(i)
            ok &= bool(np.array_equal(gat[rank], mine[eng.send_off:]))          # my own contribution, every frame, bit for bit
            # both ranks must hold the same gathered buffer
            digests = [None] * world
            dist.all_gather_object(digests, hashlib.sha256(gat.tobytes()).hexdigest())
            ok &= digests[0] == digests[1]
            for r in range(world):                         # every rank's part against the oracle on that rank's frames of this step
                for f in (0, B // 2 - 1, B // 2, B - 1):
                    okps, odesc, omono = ora.extract(host[r][k][f], (0, 1000))
                    if what == "blocks":
                        mono, kps, desc = unpack_block(gat[r], lo)[f]
                        good = mono == omono and kps.tobytes() == okps.tobytes() and np.array_equal(desc, odesc)
                    else:
                        desc_all, counts = eng.gathered_view(i, r)
                        n, mono = (int(v) for v in counts[f])
                        good = n == len(okps) and mono == omono and np.array_equal(desc_all[f, :n], odesc)
                    if not good:
                        notes.append((step, r, f))
                    ok &= good
        q.put((rank, bool(ok), notes))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("what", ["descriptors", "blocks"])
def test_replay_engine_world2_every_rank_holds_both_ranks_results(what):
    """The N > 1 path with a real ReplayEngine on every rank: two ranks, two different camera streams, several steps
    so that both block buffers and both gathered buffers are reused; every rank's gathered buffer must hold BOTH ranks' results,
    bit-exact against the oracle, and be identical on both ranks."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_world2_worker, args=(r, 2, port, what, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert sorted(r for r, _, _ in res) == [0, 1]
    assert all(ok for _, ok, _ in res), res


def _two_ranks_one_device_worker(rank, idfile, q):
    import time
    import torch
    from orb_slam3_modified_amd import ORBextractor, OrbxError, _lib, synth
    from orb_slam3_modified_amd.replay import ReplayEngine
    import ctypes as C
    L = _lib.lib()
    torch.cuda.set_device(0)
    if rank == 0:
        buf = (C.c_uint8 * 128)()
        assert L.orbx_replay_unique_id(buf) == 0
        with open(idfile + ".tmp", "wb") as f:
            f.write(bytes(buf))
        os.replace(idfile + ".tmp", idfile)
    else:
        for _ in range(600):
            if os.path.exists(idfile):
                break
            time.sleep(0.1)
    uid = open(idfile, "rb").read()
    frames = torch.from_numpy(synth.make_stream(2)[np.arange(32) % 2]).to("cuda:0")
    try:
        eng = ReplayEngine(ORBextractor(1000, 1.2, 8, 20, 7, device_id=0), frames, gather=True, lanes=1, gather_what="descriptors", rank=rank, world=2, unique_id=uid)
        i = eng.step(); eng.drain()
        q.put((rank, "created", eng.transport, int(eng.counts(i)[:, 0].sum())))
    except OrbxError as e:
        q.put((rank, "refused", str(e), 0))


def test_two_ranks_bootstrap_through_the_c_abi_is_refused_loudly_on_one_device(tmp_path):
    """The multi-rank bootstrap of the C ABI as far as a one-GPU box can take it: rank 0 makes the ncclUniqueId (orbx_replay_unique_id), the 128
    bytes reach rank 1 through a FILE (no torch.distributed anywhere: the host's own control plane), both ranks call orbx_replay_create(world = 2) —
    i.e. ncclCommInitRank finds its peer over RCCL's own bootstrap — and RCCL refuses the shared device ("Duplicate GPU detected",
    profiles/rccl_two_ranks_one_gpu_r4.txt).  What must hold: both ranks meet (no hang), both get ORBX_E_DEVICE with RCCL's message in
    orbx_last_error, nothing crashes.  On a box with two GPUs the same exchange would create the communicator."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    idfile = str(tmp_path / "nccl_id.bin")
    procs = [ctx.Process(target=_two_ranks_one_device_worker, args=(r, idfile, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert [r[0] for r in res] == [0, 1]
    for rank, what, msg, _ in res:
        assert what == "refused" and "ncclCommInitRank" in msg, res


def test_replay_c_abi_argument_checks_and_uneven_lanes():
    """orbx_replay_create refuses what it cannot run (no lanes, fewer frames than lanes, the same context twice, contexts of different
    parameters, more than one rank without a way to reach the others, both transports at once) with ORBX_E_INVALID and no engine; three lanes
    over 100 frames (34 + 34 + 32) give the oracle's bytes on every frame."""
    import ctypes as C
    import torch
    from oracle import pyoracle as po
    from orb_slam3_modified_amd import ORBextractor, _lib, synth
    from orb_slam3_modified_amd.replay import ReplayEngine, unpack_block
    L = _lib.lib()
    a, b = ORBextractor(1000, 1.2, 8, 20, 7, device_id=0), ORBextractor(1000, 1.2, 8, 20, 7, device_id=0)
    other = ORBextractor(500, 1.2, 6, 20, 7, device_id=0)
    h = C.c_void_p()

    def create(ctxs, frames, what=0, rank=0, world=1, uid=None, cb=None):
        arr = (C.c_void_p * max(len(ctxs), 1))(*[c._ctx for c in ctxs])
        return L.orbx_replay_create(C.byref(h), arr, len(ctxs), frames, 480, 640, what, rank, world, uid, cb, None)

    uid = (C.c_uint8 * 128)()
    cb = L.HOST_EXCHANGE_FN(lambda u, s, r, n: 0)
    for args in (([], 64), ([a, b], 1), ([a, a], 64), ([a, other], 64), ([a], 64, 3), ([a], 64, 1, 2, 2), ([a], 64, 1, 0, 2),
                 ([a], 64, 1, 0, 2, uid, C.cast(cb, C.c_void_p)), ([a], 0), ([a], 70000)):
        assert create(*args) == -1 and not h.value, args
    host = synth.make_stream(5)
    nfr = 100
    frames = torch.from_numpy(host[np.arange(nfr) % 5]).to("cuda:0")
    eng = ReplayEngine(a, frames, lapping=(0, 1000), gather=True, lanes=3, gather_what="blocks", alternate=False)
    assert eng.lane_ranges == [(0, 34), (34, 68), (68, 100)]
    for _ in range(3):
        i = eng.step()
    blk = eng.block_host(i)
    assert np.array_equal(eng.gathered_host(i)[0], blk)
    res = unpack_block(blk, eng.layout)
    ora = po.OracleExtractor(1000, 1.2, 8, 20, 7)
    want = [ora.extract(host[k], (0, 1000)) for k in range(5)]
    for f in range(nfr):
        okps, odesc, omono = want[f % 5]
        mono, kps, desc = res[f]
        assert mono == omono and kps.tobytes() == okps.tobytes() and np.array_equal(desc, odesc), f
    eng.close()


# ---------------------------------------------------------------------------------------------------------------------------------------------
# Round 6: the N > 1 path hardened for the run this pool cannot do

def test_a_consumer_stream_is_ordered_against_one_steps_exchange_without_a_drain():
    """orbx_replay_wait_gathered / _release_gathered: a consumer on ITS OWN stream reads gathered buffer i of step k while the engine runs on — no
    drain anywhere in the loop.  The consumer's copy of every step must equal that step's block (the steps rotate through different batches, so a
    copy taken too early — the previous occupant of the buffer — or too late — overwritten by step k + 2 — shows)."""
    import torch
    from orb_slam3_modified_amd import ORBextractor, synth
    from orb_slam3_modified_amd.replay import ReplayEngine
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    B, nsets, steps = 64, 3, 9
    host = [synth.make_stream(B, 480, 640, synth.DEFAULT_SEED + 7 * k) for k in range(nsets)]
    sets = [torch.from_numpy(h).to(dev) for h in host]
    ex = ORBextractor(1000, 1.2, 8, 20, 7, device_id=0)
    eng = ReplayEngine(ex, sets, lapping=(0, 1000), gather=True, lanes=2, gather_what="descriptors")
    with pytest.raises(Exception):
        eng.wait_gathered(0, torch.cuda.Stream().cuda_stream)            # nothing has been queued into the buffer yet
    cons = torch.cuda.Stream()
    nb = eng.send_bytes
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
    copies = [torch.empty(nb, dtype=torch.uint8, device=dev) for _ in range(steps)]
    for k in range(steps):
        i = eng.step()
        eng.wait_gathered(i, cons.cuda_stream)                            # device-side: the consumer's next work waits for step k's collective
        assert hip.hipMemcpyAsync(copies[k].data_ptr(), eng.gathered_ptr(i, 0), nb, 3, cons.cuda_stream) == 0   # 3 = device to device
        eng.release_gathered(i, cons.cuda_stream)                         # the collective of step k + 2 into buffer i waits for this point
    cons.synchronize()
    eng.drain()
    # what every step must have produced: the same batches through a second engine, one step at a time
    ex2 = ORBextractor(1000, 1.2, 8, 20, 7, device_id=0)
    ref = ReplayEngine(ex2, sets, lapping=(0, 1000), gather=False, lanes=1)
    want = []
    for k in range(nsets):
        i = ref.step()
        want.append(ref.block_host(i)[eng.send_off:].copy())
    lo = eng.layout
    c0 = lo.counts_off - lo.desc_off

    def valid(part):   # counts + the descriptor rows they cover (rows beyond a frame's count keep whatever an earlier step left in that buffer)
        counts = part[c0:c0 + lo.counts_bytes].view(np.int32).reshape(B, 2)
        desc = part[:lo.desc_bytes].reshape(B, lo.cap, 32)
        return counts, [desc[f, :counts[f, 0]] for f in range(B)]
    for k in range(steps):
        gc, gd = valid(copies[k].cpu().numpy())
        wc, wd = valid(want[k % nsets])
        assert np.array_equal(gc, wc) and gc[:, 0].min() > 500, f"the consumer's copy of step {k} does not hold that step's counts"
        assert all(np.array_equal(a, b) for a, b in zip(gd, wd)), f"the consumer's copy of step {k} does not hold that step's descriptor rows"
    assert len({want[k][c0:c0 + lo.counts_bytes].tobytes() for k in range(nsets)}) == nsets      # the batches do differ: a stale buffer would show
    assert eng.wait_gathered_host(0, 1000) and eng.wait_gathered_host(1, 1000)
    eng.close(); ref.close()


def test_lanes_must_compute_the_same_reference_build_and_get_their_fork_options_back():
    import ctypes as C
    from orb_slam3_modified_amd import ORBextractor, _lib
    L = _lib.lib()
    a, b = ORBextractor(1000, 1.2, 8, 20, 7, device_id=0), ORBextractor(1000, 1.2, 8, 20, 7, device_id=0)
    b.set_cpu_profile("opencv-4.4", 3)                                   # lane 1 would follow another OpenCV release than lane 0
    arr = (C.c_void_p * 2)(a._ctx, b._ctx)
    h = C.c_void_p()
    assert L.orbx_replay_create(C.byref(h), arr, 2, 64, 480, 640, 0, 0, 1, None, None, None) == -1
    assert b"CPU-path profile" in L.orbx_last_error(a._ctx)
    c = ORBextractor(1000, 1.2, 8, 25, 7, device_id=0)                   # another iniThFAST
    arr = (C.c_void_p * 2)(a._ctx, c._ctx)
    assert L.orbx_replay_create(C.byref(h), arr, 2, 64, 480, 640, 0, 0, 1, None, None, None) == -1
    assert b"FAST thresholds" in L.orbx_last_error(a._ctx)
    # the engine changes the lanes' fork options for its launch shape; the contexts are the caller's and get them back
    d = ORBextractor(1000, 1.2, 8, 20, 7, device_id=0)
    before = [L.orbx_get_option(e._ctx, n) for e in (a, d) for n in (b"fork_blur", b"fork_fast0", b"fork_qt")]
    arr = (C.c_void_p * 2)(a._ctx, d._ctx)
    assert L.orbx_replay_create(C.byref(h), arr, 2, 64, 480, 640, 0, 0, 1, None, None, None) == 0
    during = [L.orbx_get_option(e._ctx, n) for e in (a, d) for n in (b"fork_blur", b"fork_fast0", b"fork_qt")]
    L.orbx_replay_destroy(h)
    after = [L.orbx_get_option(e._ctx, n) for e in (a, d) for n in (b"fork_blur", b"fork_fast0", b"fork_qt")]
    assert during == [1, 1, 0, 1, 1, 0] and after == before == [1, 0, 1, 1, 0, 1], (before, during, after)


def _failing_rank_worker(rank, world, port, q):
    """world ranks on cuda:0 over the host transport; rank 1's lane fails at step 2 (and, the failure being sticky, at every later step).  Nobody may hang: the failing rank keeps taking part with a
    poisoned block, the healthy ranks see counts of -1 in its part from that step on."""
    import torch
    import torch.distributed as dist
    from orb_slam3_modified_amd import ORBextractor, OrbxError, synth
    from orb_slam3_modified_amd.replay import ReplayEngine
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(0)
        B, steps, bad_rank, bad_step = 32, 5, 1, 2
        frames = torch.from_numpy(synth.make_stream(B, 240, 320, synth.DEFAULT_SEED + rank)).to(dev)
        ex = ORBextractor(500, 1.2, 6, 20, 7, device_id=0)
        eng = ReplayEngine(ex, frames, lapping=(0, 1000), gather=True, lanes=1, gather_what="descriptors")
        errors, seen = [], []
        for step in range(steps):
            if rank == bad_rank and step == bad_step:
                eng._row_stride = 1       # a row stride smaller than the image: this rank's lane refuses the step (a real lane error, no test hook)
            try:
                i = eng.step()
            except OrbxError as e:
                errors.append((step, str(e)))
                i = step & 1
            eng.drain()
            parts = [eng.gathered_view(i, r)[1] for r in range(world)]          # [B][2] counts of every rank
            seen.append([int(p[:, 0].min()) for p in parts])
        q.put((rank, errors, seen, eng.failed))
        eng.close()
    finally:
        dist.destroy_process_group()


def test_a_failing_rank_poisons_its_block_and_nobody_hangs():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    world = 2
    procs = [ctx.Process(target=_failing_rank_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {r: (e, s, f) for r, e, s, f in [q.get(timeout=300) for _ in procs]}
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    # the healthy rank never got an error and saw rank 1's counts turn to -1 at step 2 and stay there; its own stay real
    e0, s0, f0 = res[0]
    assert e0 == [] and f0 == 0
    assert [s[1] for s in s0] == [s0[0][1], s0[1][1], -1, -1, -1] and s0[0][1] > 50 and all(s[0] > 50 for s in s0), s0
    # the failing rank got the error at step 2 and at every later step, and still holds the healthy rank's real results
    e1, s1, f1 = res[1]
    assert [st for st, _ in e1] == [2, 3, 4] and "step 2, lane 0" in e1[0][1] and "bad batch arguments" in e1[0][1] and f1 == -1, e1
    assert all(s[0] > 50 for s in s1) and [s[1] for s in s1][2:] == [-1, -1, -1], s1


def _eight_ranks_worker(rank, world, port, q):
    import hashlib
    import torch
    import torch.distributed as dist
    from orb_slam3_modified_amd import ORBextractor, synth
    from orb_slam3_modified_amd.replay import ReplayEngine, shard_streams
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(0)
        B, steps = 32, 3
        assert shard_streams(8, world, rank) == [rank]                             # S-8cam: camera c on GPU c mod 8
        host = synth.make_stream(B, 480, 640, synth.DEFAULT_SEED + 1000 * rank)
        frames = torch.from_numpy(host).to(dev)
        ex = ORBextractor(1000, 1.2, 8, 20, 7, device_id=0)
        eng = ReplayEngine(ex, frames, lapping=(0, 1000), gather=True, lanes=1, gather_what="descriptors")
        for _ in range(steps):
            i = eng.step()
        eng.drain()
        gat = eng.gathered_host(i)                                                 # [8][send_bytes]
        mine = eng.block_host(i)[eng.send_off:]
        counts = [eng.gathered_view(i, r)[1] for r in range(world)]
        q.put((rank, hashlib.sha256(gat.tobytes()).hexdigest(), bool(np.array_equal(gat[rank], mine)), hashlib.sha256(mine.tobytes()).hexdigest(),
               [hashlib.sha256(gat[r].tobytes()).hexdigest() for r in range(world)], [int(c[:, 0].sum()) for c in counts]))
        eng.close()
    finally:
        dist.destroy_process_group()


def test_eight_ranks_on_one_device_every_rank_holds_all_eight_blocks():
    """BASELINE config 5's shape — eight camera streams, eight ranks, one exchange per step — with everything but the wire: eight processes
    share cuda:0 (RCCL refuses that, so the exchange is the engine's host transport over gloo).  Every rank must end up with all eight ranks'
    descriptor blocks, identical everywhere, each part equal to what its owner computed."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    world = 8
    procs = [ctx.Process(target=_eight_ranks_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=900) for _ in procs)
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert [r[0] for r in res] == list(range(world))
    assert len({r[1] for r in res}) == 1, "the gathered buffers differ between ranks"
    assert all(r[2] for r in res)
    own = [r[3] for r in res]
    assert len(set(own)) == world                                                  # eight different cameras
    for r in res:
        assert r[4] == own and all(n > 20000 for n in r[5]), r[5]                  # every part = its owner's block; ~1000 features x 32 frames each


def test_abort_leaves_the_group_and_the_engine_goes_on_without_the_exchange():
    """orbx_replay_abort (ncclCommAbort: no hand-shake with ranks that may be gone): a one-rank RCCL engine steps with its self-gather, aborts,
    and keeps extracting — same blocks as before — with the exchange off; the bounded host wait reports a finished exchange at once."""
    import torch
    from orb_slam3_modified_amd import ORBextractor, synth
    from orb_slam3_modified_amd.replay import ReplayEngine
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    frames = torch.from_numpy(synth.make_stream(32, 240, 320)).to(dev)
    eng = ReplayEngine(ORBextractor(500, 1.2, 6, 20, 7, device_id=0), frames, lapping=(0, 1000), gather=True, lanes=1, gather_what="descriptors")
    assert eng.failed == 0
    i = eng.step()
    assert eng.wait_gathered_host(i, 5000)
    before = eng.block_host(i).copy()
    assert np.array_equal(eng.gathered_host(i).reshape(-1), before[eng.send_off:])
    eng.abort()
    assert not eng.gather and "[aborted]" in eng._L.orbx_replay_transport(eng._h).decode()
    for _ in range(3):
        j = eng.step()
    assert np.array_equal(eng.block_host(j), before) and eng.failed == 0        # the same frames: the same block, no exchange needed
    eng.close()


@pytest.mark.parametrize("lanes", [2, 3])
def test_the_alternate_lane_schedule_is_bit_exact_on_every_frame_with_steps_in_flight(lanes):
    """The default schedule since round 6: the lanes take WHOLE steps in turn (step k on lane k mod L, all frames), so two steps are in flight on
    free-running streams and every launch covers the whole batch.  Rotating batches, the exchange on, nothing drained until the end — EVERY frame
    of the last two steps (both blocks) against the oracle, each gathered buffer equal to its block; with three lanes the block of step k is written
    by another lane than the one that wrote it two steps before (the cross-lane wait)."""
    import torch
    from oracle import pyoracle as po
    from orb_slam3_modified_amd import ORBextractor, synth
    from orb_slam3_modified_amd.replay import ReplayEngine, unpack_block
    dev = torch.device("cuda", 0)
    host = synth.make_stream(24)
    nfr, nsets, steps = 96, 3, 9
    src = [[(f + 5 * k) % 24 for f in range(nfr)] for k in range(nsets)]            # which image frame f of batch k is
    sets = [torch.from_numpy(np.ascontiguousarray(host[src[k]])).to(dev) for k in range(nsets)]
    eng = ReplayEngine(ORBextractor(1000, 1.2, 8, 20, 7, device_id=0), sets, lapping=(0, 1000), gather=True, lanes=lanes, gather_what="blocks")
    assert eng.alternate and eng.lane_ranges == [(0, nfr)] * lanes
    ora = po.OracleExtractor(1000, 1.2, 8, 20, 7)
    want = [ora.extract(host[m], (0, 1000)) for m in range(24)]
    kept = [eng.step() for _ in range(steps)]
    assert kept == [s_ & 1 for s_ in range(steps)]
    for s_ in (steps - 2, steps - 1):
        i, k = kept[s_], s_ % nsets
        blk = eng.block_host(i)
        assert np.array_equal(eng.gathered_host(i)[0], blk)
        res = unpack_block(blk, eng.layout)
        for f in range(nfr):
            okps, odesc, omono = want[src[k][f]]
            mono, kps, desc = res[f]
            assert mono == omono and kps.tobytes() == okps.tobytes() and np.array_equal(desc, odesc), (lanes, s_, f)
    eng.close()
