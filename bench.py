#!/usr/bin/env python3
"""ORB front-end throughput on MI355X: features/ms and frames/s of the extractor hot path.

    python bench.py --gpus N --steps K --warmup W

Started without WORLD_SIZE and with N > 1, the script re-executes itself under `torch.distributed.run` with one rank per
visible GPU (min(N, visible); the line reports the world size RCCL actually saw).  Started by a launcher (RANK / LOCAL_RANK /
WORLD_SIZE set) it is one rank of that job.

A "step" is one pass of the hot path (pyramid -> FAST cells -> quadtree -> orientation -> rBRIEF) over one batch of B synthetic
640x480 frames per GPU that are already resident in HBM; results stay in HBM.  Workload = BASELINE.json configs[1] shape
(S-EuRoC-640: 640x480, 8 levels, 1000 features) replayed in batches; the steps rotate through several DISTINCT batches (more
input than the 256 MB Infinity Cache holds); every rank replays its own camera stream (seed + 1000*rank), i.e. weak scaling,
and for N > 1 the per-step feature blocks are all-gathered over RCCL asynchronously (batch-replay mode).
After the timed region frames of the last step are compared with the CPU oracle ("verified_frames"); a mismatch fails the run.
At N = 1 the same line also carries: the end-to-end latency of ORBextractor::operator() through the C++ adapter (host buffers),
the streamed front-end (operator() + BoW transform + the two per-frame guided searches as one loop, next to the same loop on the
reference-compiled CPU code), a secondary measurement of BASELINE config 4 (1024x1024, 2000 features) with its own roofline, and the
CPU baseline (the reference's own src/ORBextractor.cc compiled where it lies + this repository's port).
--streams 8 selects SURVEY 8(e)'s curve (the same 8 camera streams on G GPUs); N > 1 lines report the all-gather's device time.
Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import shutil
import socket
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


def level_pixels(rows, cols, nlevels=8, sf=1.2):
    s, out = np.float32(1.0), []
    for l in range(nlevels):
        if l:
            s = np.float32(np.float64(s) * np.float64(np.float32(sf)))
        inv = np.float32(1.0) / s
        out.append(int(np.rint(np.float32(cols) * inv)) * int(np.rint(np.float32(rows) * inv)))
    return out


def algorithmic_bytes(rows, cols, n_keypoints_per_frame, n_candidates_per_frame):
    """SURVEY.md §8(d).  Returns (fused-ideal bytes/frame, per-kernel staged bytes/frame)."""
    px = level_pixels(rows, cols)
    P, P0 = sum(px), px[0]
    fused = P0 + (P - P0) + 2 * P + 60 * n_keypoints_per_frame
    staged = {
        "k_resize(pyramid chain)": (P - px[-1]) + (P - P0),          # read levels 0..6, write levels 1..7
        "k_fast_cells": P + 4 * n_candidates_per_frame,               # one read of the pyramid + packed candidates out
        "k_quadtree": 4 * n_candidates_per_frame * 2,                 # gather candidates + final read (points stay in L2)
        "k_assemble": 8 * n_keypoints_per_frame,
        "k_blur7": 2 * P,                                              # read every level once, write its blurred copy
        "k_describe": (961 + 1369 + 60) * n_keypoints_per_frame,       # 31x31 patch + taps within 37x37 + 60 B out (with the Gaussian inside — k_describe_blur,
                                                                       # no k_blur7 — it is the 43x43 raw window + 60 B, and the 2 P of the blur row go away)
    }
    return fused, staged


def usable_cores():
    """CPU cores this process may actually use: the affinity mask and the cgroup CPU quota, not the host's core count (a
    container on a 256-thread host is typically limited to a few cores' worth of time: threads beyond the quota only throttle)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]           # cgroup v2
        if q != "max":
            quota = float(q) / float(per)
    except (OSError, ValueError):
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())       # cgroup v1
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    if quota:
        n = max(1, min(n, int(quota)))
    return n, (os.cpu_count() or 1), quota


def cpu_reference(frames, nfeatures, budget_s):
    """The reference's OWN src/ORBextractor.cc (oracle/_ref/libref_orbextractor.so: compiled where it lies by oracle/ref_fragments.mk,
    -O2, portable flags — the binary travels to the GPU box) timed on the host cores: one ORBextractor instance per thread, frames
    dealt round-robin, ctypes releases the GIL for the 10 ms a call takes.  Its five OpenCV primitives are the oracle's scalar
    restatements (no OpenCV exists in this image).  None when the library was not built."""
    import threading
    from oracle import pyoracle as po
    if not po.ref_extractor_available():
        return None
    ncores, host_threads, quota = usable_cores()
    fr = np.ascontiguousarray(frames)

    def run(threads, seconds):
        exs = [po.RefExtractor(nfeatures, 1.2, 8, 20, 7) for _ in range(threads)]
        done = [[0, 0] for _ in range(threads)]
        exs[0].extract(fr[0], (0, 1000))   # first-touch outside the clock
        t_end = time.perf_counter() + seconds

        def work(j):
            i = j
            while time.perf_counter() < t_end:
                kps, _, _ = exs[j].extract(fr[i % len(fr)], (0, 1000))
                done[j][0] += 1; done[j][1] += len(kps)
                i += threads
        t0 = time.perf_counter()
        th = [threading.Thread(target=work, args=(j,)) for j in range(threads)]
        for t in th: t.start()
        for t in th: t.join()
        dt = time.perf_counter() - t0
        return sum(d[1] for d in done) / (dt * 1e3), sum(d[0] for d in done), dt

    v1, k1, dt1 = run(1, budget_s * 0.4)
    va, ka, dta = run(ncores, budget_s * 0.6)
    return {"value": round(va, 3), "unit": "features/ms", "cores": ncores, "kind": "reference",
            "what": "the reference's src/ORBextractor.cc compiled where it lies (oracle/_ref/libref_orbextractor.so, g++ -O2 -ffp-contract=off, "
                    "portable); the five OpenCV primitives it calls (resize, FAST, GaussianBlur, copyMakeBorder, fastAtan2) are scalar "
                    "restatements — real OpenCV SIMD kernels are faster per core",
            "value_1core": round(v1, 3), "ms_per_frame_1core": round(dt1 * 1e3 / max(k1, 1), 3), "scaling_efficiency": round(va / (v1 * ncores), 3),
            "host_hw_threads": host_threads, "cgroup_cpu_quota": quota,
            "sample": f"{ka} frames of the same {fr.shape[2]}x{fr.shape[1]} stream on {ncores} threads ({dta:.1f} s), one ORBextractor per thread; "
                      f"1-core figure from {k1} frames ({dt1:.1f} s)"}


def cpu_port(frames, nfeatures, budget_s=12.0):
    """The oracle (CPU restatement of src/ORBextractor.cc, pinned to the reference's own file by tests/test_ref_fragments.py)
    timed on the host cores: a REPORTED baseline.  The timing copy is built here with -O3 -march=native -ffp-contract=off
    (SURVEY.md §8(d)); the frame-parallel leg is a std::thread pool inside the library (orbo_extract_many)."""
    from oracle import pyoracle as po
    ncores, host_threads, quota = usable_cores()
    po.build()
    libpath, flags = os.path.join(ROOT, "oracle", "liborb_oracle.so"), "portable build (-O3, no -march=native: native build failed)"
    tmp = tempfile.mkdtemp(prefix="orbx_native_")
    try:
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "native", f"OUT={tmp}"], stdout=subprocess.DEVNULL,
                              stderr=subprocess.DEVNULL)
        libpath, flags = os.path.join(tmp, "liborb_oracle_native.so"), "-O3 -march=native -ffp-contract=off"
    except Exception:
        pass
    L = C.CDLL(libpath)
    L.orbo_extract_many.restype = C.c_double
    L.orbo_extract_many.argtypes = [C.c_int, C.c_float, C.c_int, C.c_int, C.c_int, C.c_void_p] + [C.c_int] * 6 + [C.c_double, C.POINTER(C.c_longlong),
                                                                                                              C.POINTER(C.c_longlong)]
    fr = np.ascontiguousarray(frames)
    n, rows, cols = fr.shape

    def run(threads, seconds):
        f, k = C.c_longlong(0), C.c_longlong(0)
        dt = L.orbo_extract_many(nfeatures, 1.2, 8, 20, 7, fr.ctypes.data, n, rows, cols, 0, 1000, threads, seconds, C.byref(f), C.byref(k))
        return f.value / (dt * 1e3), k.value, dt

    v1, k1, dt1 = run(1, budget_s * 0.4)
    va, ka, dta = run(ncores, budget_s * 0.6)
    return {"value": round(va, 3), "unit": "features/ms", "cores": ncores, "kind": "port", "value_1core": round(v1, 3),
            "ms_per_frame_1core": round(dt1 * 1e3 / max(k1, 1), 3), "scaling_efficiency": round(va / (v1 * ncores), 3), "build": flags,
            "host_hw_threads": host_threads, "cgroup_cpu_quota": quota,
            "sample": f"{ka} frames of the same {cols}x{rows} stream on {ncores} std::threads ({dta:.1f} s); 1-core figure from {k1} frames "
                      f"({dt1:.1f} s); CPU path = this repo's restatement of src/ORBextractor.cc, checked bit for bit against the reference's "
                      "own file compiled over a container shim; its five OpenCV primitives are scalar restatements (real OpenCV SIMD "
                      "FAST / blur / resize is typically faster)"}


def cpu_baseline(frames, nfeatures, budget_s=24.0):
    """north_star: "the reference src/ORBextractor.cc timed on the host cores (core count stated) in the same run".  The reference's
    own file where its build exists (kind "reference"); this repository's restatement (kind "port", -O3 -march=native) beside it."""
    ref = None
    try:
        ref = cpu_reference(frames, nfeatures, budget_s * 0.5)
    except Exception as e:   # noqa: BLE001 — the reported baseline must not cost the bench line
        ref = None
        err = str(e)[:200]
    port = cpu_port(frames, nfeatures, budget_s * (0.5 if ref else 1.0))
    if ref is None:
        return port
    ref["port"] = port
    return ref


def streamed_frontend(host_frames, nfeatures, voc_descriptors, nframes=48):
    """BASELINE config 3's stand-in: the per-frame sequence Tracking runs — ORBextractor::operator() (src/Frame.cc:311,418-425) ->
    ORBVocabulary::transform (Frame::ComputeBoW, :738-745) -> SearchByProjection(Cur, Last) (src/Tracking.cc:2889) ->
    SearchByProjection(F, local points) (:3416) — as ONE loop over a stream, one frame at a time, host buffers in and out, through the
    C++ adapters on tests/support/ref_world objects (tools/streamed_frontend.cpp, built here), next to the SAME loop on the
    reference's own CPU code (oracle/_ref/ref_streamed_frontend, prebuilt) with the digest of all results compared."""
    from tests import world_util as wu
    tmp = tempfile.mkdtemp(prefix="orbx_front_")
    try:
        fr = np.ascontiguousarray(host_frames[:nframes])
        n, rows, cols = fr.shape
        raw, vocp = os.path.join(tmp, "frames.raw"), os.path.join(tmp, "voc.txt")
        fr.tofile(raw)
        from tests.vocab_util import make_vocabulary
        voc_info = make_vocabulary(vocp, np.concatenate(voc_descriptors), 10, 5, seed=9)
        exe = wu.build_frontend("orbx", tmp)
        out = wu.run_frontend(exe, raw, rows, cols, n, nfeatures, vocp, passes=3, timeout=300)
        out["vocabulary"] = f"synthetic k=10 L=5 ({voc_info['nodes']} nodes, {voc_info['words']} words) over the stream's own descriptors; levelsup 4"
        out["what"] = ("per frame, one at a time, host buffers: operator() + transform + SearchByProjection(Cur, Last, 15) + "
                       "SearchByProjection(F, ~2000 local points, th 1) through include/ORBextractor.h / ORBVocabulary.h / ORBmatcher.h; "
                       "ms_per_frame = the four calls, the host parts of Frame::Frame / isInFrustum are listed separately")
        if os.path.exists(wu.REF_FRONTEND_EXE):
            nref = min(n, 24)
            ref = wu.run_frontend(wu.REF_FRONTEND_EXE, raw, rows, cols, nref, nfeatures, vocp, passes=1, timeout=300)
            chk = wu.run_frontend(exe, raw, rows, cols, nref, nfeatures, vocp, passes=1, timeout=300) if nref != n else out
            # the comparison is made on a run of EQUAL length on both sides (the CPU loop is too slow for the timed run's frame count):
            # both digests of that comparison are printed, next to the frame count they cover
            ref["identical_results"] = bool(ref["results_digest"] == chk["results_digest"])
            ref["compared"] = {"frames": int(nref), "digest_cpu": ref["results_digest"], "digest_gpu_same_frames": chk["results_digest"]}
            out["results_digest_frames"] = int(out.get("frames_timed", n))
            ref["cores"] = 1
            out["cpu"] = ref
            if not ref["identical_results"]:
                raise SystemExit("bench.py: the streamed front-end on the GPU and the reference-compiled CPU loop DISAGREE — no number reported")
        # the same sequence AS A STREAM (tools/streamed_frontend.cpp --frames: forth and back through the images, a ring of 8 frames): 512 frames back to
        # back, and 160 frames paced by the 20 Hz time stamps of EuRoC MH_01 (Examples/Monocular/mono_euroc.cc:150-160: the GPU idles 49.6 of every
        # 50 ms) — every call's p50 / p90 / p99 / max.  The full 3 682-frame runs with the digest against the reference: profiles/config3_full_r6.txt
        try:
            stamps = wu.mh01_stamps(tmp)
            free = wu.run_frontend(exe, raw, rows, cols, n, nfeatures, vocp, passes=1, timeout=300, frames=512)
            paced = wu.run_frontend(exe, raw, rows, cols, n, nfeatures, vocp, passes=1, timeout=300, frames=160, stamps=stamps, pace=1)
            out["stream"] = {"back_to_back": {"frames": free["frames_timed"], "percentiles": free["percentiles"]},
                             "paced_20hz": {"frames": paced["frames_timed"], "percentiles": paced["percentiles"], "wall_s": paced["stream"]["wall_s"]},
                             "paced_over_back_to_back_p50": round(paced["percentiles"]["four_calls_ms"]["p50"] / free["percentiles"]["four_calls_ms"]["p50"], 3)}
        except Exception as e:   # noqa: BLE001
            out["stream"] = {"error": str(e)[:200]}
        return out
    except SystemExit:
        raise
    except Exception as e:   # noqa: BLE001
        return {"error": str(e)[:300]}


def frame_constructor():
    """Frame::Frame of the reference's UNMODIFIED src/Frame.cc (stereo: two extractions on two threads + ComputeStereoMatches on the host mirror of
    mvImagePyramid + the grid; monocular: extraction + grid), 752x480, compiled over the drop-in headers + liborbx.so (oracle/_ref/dropin_frame_world)
    and over the reference's own extractor (oracle/_ref/ref_frame_world) — both prebuilt by oracle/ref_fragments.mk where /root/reference exists;
    their results are compared by tests/test_frame_world.py, here they are timed."""
    ref_dir = os.path.join(ROOT, "oracle", "_ref")
    voc = os.path.join(ROOT, "tests", "golden", "voc_k5_L3.txt")
    out = {}
    tmp = tempfile.mkdtemp(prefix="orbx_frame_")
    try:
        for key, exe, n in (("gpu", "dropin_frame_world", 200), ("gpu_stereo_patch", "dropin_frame_world_stereo", 200), ("cpu", "ref_frame_world", 12)):
            path = os.path.join(ref_dir, exe)
            if not os.path.exists(path):
                out[key] = {"error": f"oracle/_ref/{exe} not built"}
                continue
            r = subprocess.run([path, voc, os.path.join(tmp, "o.txt"), str(-n)], capture_output=True, text=True, timeout=300)
            if r.returncode != 0:
                out[key] = {"error": (r.stderr or r.stdout).strip()[-200:]}
                continue
            out[key] = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        out["what"] = ("the reference's own src/Frame.cc constructors, 752x480: gpu = compiled over include/ORBextractor.h + liborbx.so, cpu = over the "
                       "reference's src/ORBextractor.cc (OpenCV primitives restated, 2 threads in the stereo constructor as in the reference); "
                       "ComputeStereoMatches runs on the host in both (Frame.cc untouched); gpu_stereo_patch = src/Frame.cc with "
                       "integration/Frame_stereo.patch (4 lines) and -DORBX_DEVICE_STEREO: the association on the device pyramids, no host pyramid")
        if "stereo_frame_ms" in out.get("gpu_stereo_patch", {}) and "gpu" in out:
            out["gpu"]["stereo_patched_ms"] = out["gpu_stereo_patch"]["stereo_frame_ms"]
        return out
    except Exception as e:   # noqa: BLE001
        return {"error": str(e)[:300]}


def e2e_operator(host_frames, nfeatures):
    """ORBextractor::operator() per frame through the C++ adapter, host buffers (tools/e2e_operator.cpp built here)."""
    pkg = os.path.join(ROOT, "orb_slam3_modified_amd")
    tmp = tempfile.mkdtemp(prefix="orbx_e2e_")
    exe, raw = os.path.join(tmp, "e2e_operator"), os.path.join(tmp, "frames.raw")
    try:
        subprocess.check_call(["g++", "-O2", "-std=c++14", "-DORBX_FORCE_CV_COMPAT", "-I", os.path.join(ROOT, "include"),
                               os.path.join(ROOT, "tools", "e2e_operator.cpp"), "-o", exe, "-L", pkg, "-lorbx", "-Wl,-rpath," + pkg,
                               "-Wl,--allow-shlib-undefined"], stderr=subprocess.DEVNULL)
        fr = np.ascontiguousarray(host_frames[:32])
        fr.tofile(raw)
        r = subprocess.run([exe, raw, str(fr.shape[1]), str(fr.shape[2]), str(len(fr)), str(nfeatures), "30"], capture_output=True, text=True,
                           timeout=120)
        if r.returncode != 0:
            return {"error": (r.stderr or r.stdout).strip()[:200]}
        out = json.loads(r.stdout.strip().splitlines()[-1])
        out["what"] = ("ORB_SLAM3::ORBextractor::operator() through include/ORBextractor.h, one frame per call, host buffers: image in, "
                       "kernels, keypoints + descriptors + the host mirror of mvImagePyramid out (PCIe-inclusive, latency-bound; never the "
                       "headline value); ms_per_frame_without_host_pyramid = the same with SetKeepHostPyramid(false), the monocular setting; "
                       "device_ms_per_frame = HIP events around the replayed graph (upload kernel, the pyramid in one launch, FAST + blur, quadtree + "
                       "assembly, descriptors[, pyramid mirror]): the kernels' share of a call, the rest is the host (image into the pinned buffer, "
                       "graph launch, wake-up)")
        return out
    except Exception as e:   # noqa: BLE001 — a missing compiler must not fail the benchmark line
        return {"error": str(e)[:200]}


def natural_leg(args, dev, local_rank, sync_all):
    """The same batch replay on REAL texture: the two 640x480 natural crops of tests/golden/natural_crops.npz (a photograph and a 60 %
    saturated screenshot, cut from the images the reference ships) tiled to a 256-frame batch — every copy shifted cyclically by its own
    (dx, dy), so that no two frames are equal.  The synthetic stream is 25 % flat by construction; this leg says what the kernels do on
    natural statistics: step time, features/ms, and k_fast_cells alone."""
    import torch
    from orb_slam3_modified_amd import ORBextractor
    from orb_slam3_modified_amd.replay import ReplayEngine
    try:
        nat = np.load(os.path.join(ROOT, "tests", "golden", "natural_crops.npz"))
        crops = [np.ascontiguousarray(nat[k]) for k in ("result_640x480_img", "pineapple_640x480_img")]
        B, steps = 256, 12
        host = np.stack([np.roll(crops[i % 2], (7 * (i // 2) % 480, 13 * (i // 2) % 640), (0, 1)) for i in range(B)])
        frames = torch.from_numpy(host).to(dev)
        exn = ORBextractor(args.nfeatures, 1.2, 8, 20, 7, device_id=local_rank)
        exn.set_cpu_profile(args.profile, args.fma_build)
        eng = ReplayEngine(exn, frames, lapping=(0, 1000), gather=False, lanes=args.lanes)
        dt, dmm, last = timed_median(eng, steps, 3, sync_all)
        c = eng.counts(last)
        v = verify_block(eng, last, host, [0, 1, B - 1], args.nfeatures, (0, 1000), args.variant)
        _, nkp, _, roof = kernel_roofline(exn, eng, frames, B, 480, 640, c, 1, steps, dt, nprof=3)
        return {"workload": f"natural crops (photograph + saturated screenshot, 640x480) tiled to {B} cyclically shifted frames, nfeatures {args.nfeatures}",
                "value": round(float(c[:, 0].sum()) * steps / (dt * 1e3), 1), "unit": "features/ms", "ms_per_step": round(dt / steps * 1e3, 4),
                "repeats": 3, "ms_per_step_min_max": [round(v_ / steps * 1e3, 4) for v_ in dmm],
                "features_per_frame": round(nkp, 1), "verified_frames": v, "kernels_ms_per_launch": roof["kernels_ms_per_launch"],
                "k_fast_cells_frac_of_hbm_peak": roof["frac"] if roof["kernel"].startswith("k_fast_cells") else None}
    except SystemExit:
        raise
    except Exception as e:   # noqa: BLE001
        return {"error": str(e)[:300]}


def verify_block(eng, block_index, host_frames, frame_ids, nfeatures, lap, variant=(0, 0, 0, 0, 0)):
    """Frames of one finished step against the CPU oracle (under the same CPU-path variant as the timed contexts), bit for bit.  Returns the
    number verified; raises on mismatch."""
    from oracle import pyoracle as po
    from orb_slam3_modified_amd.replay import unpack_block
    res = unpack_block(eng.block_host(block_index), eng.layout)
    with po.opencv_variant(*variant):
        ora = po.OracleExtractor(nfeatures, 1.2, 8, 20, 7)
        for f in frame_ids:
            okps, odesc, omono = ora.extract(host_frames[f], lap)
            mono, kps, desc = res[f]
            if not (mono == omono and kps.tobytes() == okps.tobytes() and np.array_equal(desc, odesc)):
                raise SystemExit(f"bench.py: frame {f} of the last timed step differs from the CPU oracle "
                                 f"({len(kps)} vs {len(okps)} keypoints) — the measured path is WRONG, no number reported")
    return len(frame_ids)


def fast_cells_per_frame(rows, cols, nlevels=8, sf=1.2):
    """Number of 35-px FAST cells of one frame (src/ORBextractor.cc:789-803: the border box is 16 px inside the image on every side)."""
    s, n = np.float32(1.0), 0
    for l in range(nlevels):
        if l:
            s = np.float32(np.float64(s) * np.float64(np.float32(sf)))
        inv = np.float32(1.0) / s
        w, h = int(np.rint(np.float32(cols) * inv)), int(np.rint(np.float32(rows) * inv))
        n += ((w - 32) // 35) * ((h - 32) // 35)
    return n


def rocprof_row(kernel, B, rows, cols, khash, ncalls=8):
    """The committed rocprofv3 --kernel-trace summary of this command (profiles/<tag>_kernel_stats.md, written by tools/final_refresh.sh): the
    average duration of `kernel`'s whole-batch launches in the ISOLATED roofline passes, when the file is stamped with the kernel sources of THIS
    build.  The roofline's own time is the HIP-event one measured in this run; this is the cross-check the contract asks for, with the file it
    comes from.  k_fast_cells runs as two launches (residency groups) whose workgroup counts sum to (cells per frame) x B.  The summary has one row
    per (launch shape, stream): since the replay lanes take whole steps in turn their overlapped launches have the same shapes as the isolated
    passes, but they come from other streams — the rows taken are those of ONE stream whose call counts equal the number of isolated passes."""
    import glob
    import itertools
    import re
    name = kernel.split("(")[0]
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_kernel_stats.md")), key=os.path.getmtime, reverse=True):
        try:
            txt = open(path).read()
        except OSError:
            continue
        m = re.search(r"kernel sources ([0-9a-f]{16})", txt)
        if not m or m.group(1) != khash:
            continue
        found = []   # (grid workgroups, calls, avg us, min us, max us, queue)
        for line in txt.splitlines():
            mm = re.match(r"\|[^|]*\b" + re.escape(name) + r"\b[^|]*\[grid (\d+)x1x1 wg\](?: \[(?:queue|stream) (\d+)\])?\s*\|\s*(\d+)\s*\|\s*[\d.]+\s*\|\s*([\d.]+)\s*\|\s*([\d.]+)\s*\|\s*([\d.]+)", line)
            if mm:
                found.append((int(mm.group(1)), int(mm.group(3)), float(mm.group(4)), float(mm.group(5)), float(mm.group(6)), mm.group(2)))
        if not name.startswith("k_fast_cells"):
            continue
        want = fast_cells_per_frame(rows, cols) * B
        best = None
        for q in sorted({r[5] for r in found}, key=lambda x: (x is None, x)):
            rows_q = [r for r in found if r[5] == q]
            for k in (1, 2, 3):
                for combo in itertools.combinations(rows_q, k):
                    if sum(r[0] for r in combo) == want:
                        exact = all(r[1] == ncalls for r in combo)
                        cand = (0 if exact else 1, sum(r[1] for r in combo), combo, q)
                        if best is None or cand[:2] < best[:2]:
                            best = cand
        if best:
            combo, q = best[2], best[3]
            out = {"file": os.path.relpath(path, ROOT), "rocprof_avg_ms": round(sum(r[2] for r in combo) / 1e3, 4), "calls": [r[1] for r in combo],
                   "grids": [r[0] for r in combo], "min_ms": round(sum(r[3] for r in combo) / 1e3, 4), "max_ms": round(sum(r[4] for r in combo) / 1e3, 4)}
            if q is not None:
                out["stream"] = int(q)
            return out
    return None


def kernel_roofline(ex, eng, frames, B, H, W, counts, world, steps, dt, nprof=8):
    """HIP-event time of every kernel (passes after the timed region: whole per-GPU batch on one context, kernels back to back)
    -> roofline object of the dominant kernel."""
    import torch
    # this context's buffers are re-allocated for the whole per-GPU batch (its lane used half of it) WITHOUT launching anything: every
    # whole-batch launch a profiler sees under this command is one of the timed passes below
    ex.reserve(H, W, B)
    torch.cuda.synchronize()
    out = eng.block_ptr(0)
    lo = eng.layout
    # one untimed pass over B - 8 frames: the first launches on freshly allocated buffers are slow (k_quadtree: 1.7 ms once instead of 0.1), and a
    # pass of ANOTHER frame count has other grid sizes than the timed ones, so a kernel trace of this command keeps it apart from them
    if B > 16:
        ex.extract_batch_device(frames.data_ptr(), B - 8, H, W, frames.stride(1), frames.stride(0), out, out + lo.desc_off, out + lo.counts_off,
                                (0, 1000), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
    ex.profile_enable(True)
    for _ in range(nprof):
        ex.extract_batch_device(frames.data_ptr(), B, H, W, frames.stride(1), frames.stride(0), out, out + lo.desc_off, out + lo.counts_off,
                                (0, 1000), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    prof = ex.profile_read()
    ex.profile_enable(False)
    ncand = sum(len(ex.debug_level_points(l, 0, frame=0)[0]) for l in range(8))
    nkp = float(counts[:, 0].mean())
    fused, staged = algorithmic_bytes(H, W, nkp, float(ncand))
    per_kernel = {k: (ms / max(n, 1)) for k, (ms, n) in prof.items() if n}
    dom = max(per_kernel, key=per_kernel.get)
    dom_ms = per_kernel[dom]
    dom_bytes = staged[dom] * B
    achieved = dom_bytes / (dom_ms * 1e-3) / 1e9
    # the bandwidth a plain device copy reaches on this GPU, as a second denominator next to the 8 TB/s of the spec
    copy_gbs = None
    try:
        nb = 1 << 29
        src = torch.empty(nb, dtype=torch.uint8, device=frames.device); dst = torch.empty_like(src); src.zero_()
        torch.cuda.synchronize()
        cs = torch.cuda.Stream()   # an explicit stream: handle 0 would mean "the context's own stream" to the C ABI
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for it in range(4):
            if it == 1: e0.record(cs)
            ex.debug_calib_copy(src.data_ptr(), dst.data_ptr(), nb, 16, cs.cuda_stream)
        e1.record(cs); torch.cuda.synchronize()
        copy_gbs = 2.0 * nb * 3 / (e0.elapsed_time(e1) * 1e-3) / 1e9
        del src, dst
    except Exception:   # noqa: BLE001 — a diagnostic, never a reason to lose the bench line
        copy_gbs = None
    return dom, nkp, fused, {
        "bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None,
        "device_copy_GBs": None if copy_gbs is None else round(copy_gbs, 1),
        "frac_of_device_copy": None if not copy_gbs else round(achieved / copy_gbs, 5),
        "algorithmic_bytes_per_launch": int(dom_bytes), "avg_launch_ms": round(dom_ms, 4), "frames_per_launch": B,
        "avg_launch_ms_source": f"HIP events around the kernel on its own stream, {nprof} whole-batch passes in this run (orbx_profile_read)",
        "launch_conditions": "whole per-GPU batch in one launch, kernels back to back on one context and its own queue (passes after the timed region; the one "
                             "untimed warm-up pass runs B - 8 frames, i.e. other grid sizes; the replay lanes' launches of the timed region have the same "
                             "shapes but overlap each other and come from other streams: a kernel trace of this command tells them apart by stream)",
        "pipeline_fused_ideal_bytes_per_frame": int(fused),
        "pipeline_frac": round(fused * (B * world * steps / dt) / 1e9 / (HBM_PEAK_GBS * world), 5),
        "kernels_ms_per_launch": {k: round(v, 4) for k, v in per_kernel.items()}}


def median_of(xs):
    xs = sorted(xs)
    return xs[len(xs) // 2] if len(xs) % 2 else 0.5 * (xs[len(xs) // 2 - 1] + xs[len(xs) // 2])


def exchange_legs(eng, steps, repeats, sync_all, reduce_max=None):
    """The same K-step timed loop WITH and WITHOUT the per-step collective, `repeats` times each, alternating (so that a drift of the box hits
    both alike) -> medians, extremes and the collective's own device time.  reduce_max: all-reduce(MAX) of a list over the ranks (N > 1)."""
    with_g, without_g, g_all = [], [], []
    for _ in range(repeats):
        for on, acc in ((True, with_g), (False, without_g)):
            eng.gather = on
            dt, _ = timed_replay(eng, steps, 2, sync_all)     # (resets the collective's timing after its warm-up steps)
            acc.append(dt)
            if on:
                g_all.append(eng.gather_ms() or 0.0)
    eng.gather = True
    g_ms = median_of(g_all)
    if reduce_max is not None:
        with_g, without_g = reduce_max(with_g), reduce_max(without_g)
        g_ms = reduce_max([g_ms])[0]
    k = 1e3 / steps
    mw, mo = median_of(with_g) * k, median_of(without_g) * k
    spread = max((max(v) - min(v)) / median_of(v) for v in (with_g, without_g))
    return {"gather_ms": None if not g_ms else round(g_ms, 4), "step_ms_with_gather": round(mw, 4), "step_ms_without_gather": round(mo, 4),
            "step_ms_with_gather_min_max": [round(min(with_g) * k, 4), round(max(with_g) * k, 4)],
            "step_ms_without_gather_min_max": [round(min(without_g) * k, 4), round(max(without_g) * k, 4)],
            "exposed_ms_per_step": round(mw - mo, 4),
            "exposed_note": "median(with) - median(without) over alternating repeats, signed: a value inside +-spread is 'not measurable', not 'zero'",
            "spread_frac": round(spread, 4), "repeats": repeats, "steps": steps}


def self_gather_exchange(ex, frame_sets, args, step_ms_plain, sync_all):
    """N = 1: what the per-step collective costs this GPU even alone — a ONE-rank RCCL group made by liborbx itself (orbx_replay_create without a
    unique id), the same engine with the all-gather switched on, the same timed loop.  No link carries anything (one rank), so gather_ms is
    RCCL's own launch + copy kernel on this GPU and the with / without difference is what those kernels take from the extractor's (VALU-bound)
    kernels: the two numbers DESIGN.md section 6's estimate for G = 8 starts from.  Runs after the timed region of `value`."""
    from orb_slam3_modified_amd.replay import ReplayEngine
    try:
        eng = ReplayEngine(ex.clone(), frame_sets, lapping=(0, 1000), gather=True, lanes=args.lanes, gather_what=args.gather, rank=0, world=1)
        steps = max(10, min(args.steps, 30))
        timed_replay(eng, steps, 3, sync_all)      # warm: the first collective builds RCCL's channels
        out = {"collective": f"ncclAllGather({args.gather}) called by liborbx (orbx_replay_step) in a ONE-rank group (self-gather: no link traffic), one per "
                             "step, async on its own stream, double-buffered", "transport": eng.transport, "bytes_per_rank_per_step": int(eng.send_bytes),
               "bytes_received_per_rank_per_step": 0}
        out.update(exchange_legs(eng, steps, max(3, args.repeats // 2 + 1), sync_all))
        out["headline_step_ms"] = round(step_ms_plain, 4)
        eng.close()
        return out
    except Exception as e:   # noqa: BLE001 — a diagnostic leg must not cost the bench line
        return {"error": str(e)[:300]}


class stdout_to_stderr:
    """RCCL prints its version banner to the C stdout when a communicator is created; the driver reads ONE JSON line from stdout."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        sys.stdout.flush()
        try:   # RCCL printed through libc's buffered stdout (a pipe is fully buffered): push it out while fd 1 still points at stderr
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:   # noqa: BLE001
            pass
        os.dup2(self.saved, 1)
        os.close(self.saved)
        return False


def timed_median(eng, steps, warmup, sync_all, repeats=3):
    """timed_replay `repeats` times (warm-up before the first only) -> (median seconds, [min, max] seconds, buffer index of the last step)."""
    runs = [timed_replay(eng, steps, warmup if r == 0 else 0, sync_all) for r in range(repeats)]
    ds = sorted(d for d, _ in runs)
    return median_of(ds), [ds[0], ds[-1]], runs[-1][1]


def timed_replay(eng, steps, warmup, sync_all):
    for _ in range(warmup):
        eng.step()
    eng.drain()
    sync_all()
    eng.reset_gather_timing()   # the first collective creates the communicator (hundreds of ms): not part of a step's gather time
    t0 = time.perf_counter()
    last = 0
    for _ in range(steps):
        last = eng.step()
    eng.drain()
    sync_all()
    return time.perf_counter() - t0, last


def matcher_roofline(host_frames, nfeatures, voc_descriptors):
    """the matcher and bag-of-words kernels next to THEIR ceilings, HIP-event timed in this run, on the stream each kernel runs on.
      k_knn2      all-pairs Hamming + two best (cv::BFMatcher.knnMatch(k=2), src/Frame.cc:1144): pairs/s against the integer VALU bound — 256 bits of
                  XOR + popcount are 16 lane-operations per pair (8 v_xor_b32 + 8 v_bcnt_u32_b32 with accumulate) at the 4-cycle wave-instruction
                  rate of profiles/valu_ceiling_r2.txt: 1024 SIMDs x 2.4 GHz / 4 x 64 lanes / 16 = 2.46e12 pairs/s
      k_window    the per-frame guided search (SearchByProjection's device pass over a resident target): algorithmic bytes
                  32 (Q + T) + 4 nnz + 12 Q against HBM — a single ~10 us launch, latency-bound by construction
      k_bow_descend  the vocabulary tree descent: N L k 32 gathered bytes per call against the L2 rate (34.5 TB/s: the tree is L2-resident)"""
    import torch
    from orb_slam3_modified_amd import ORBextractor, ORBmatcher, ORBVocabulary, _lib
    from orb_slam3_modified_amd._lib import ptr
    from tests.vocab_util import make_vocabulary
    out = {}
    try:
        L = _lib.lib()
        dev = torch.device("cuda", torch.cuda.current_device())
        ex = ORBextractor(nfeatures, 1.2, 8, 20, 7, device_id=dev.index)
        fr = [ex(f, None, (0, 1000)) for f in host_frames[:2]]
        (k0, d0h), (k1, d1h) = (fr[0][1], fr[0][2]), (fr[1][1], fr[1][2])
        side = torch.cuda.Stream()
        st = side.cuda_stream

        def ev_time(fn, reps):   # torch events on the stream the kernel is launched on
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(side)
            for _ in range(reps):
                fn()
            b.record(side)
            torch.cuda.synchronize()
            return a.elapsed_time(b) / reps * 1e-3
        valu_pairs_peak = 1024 * 2.4e9 / 4 * 64 / 16
        knn = {}
        for name, q, t in (("1000x1000", torch.from_numpy(d0h).to(dev), torch.from_numpy(d1h).to(dev)),
                           ("8192x8192", *(torch.randint(0, 256, (8192, 32), dtype=torch.uint8, device=dev) for _ in range(2)))):
            nq, nt = int(q.shape[0]), int(t.shape[0])
            idx = torch.zeros(nq * 2, dtype=torch.int32, device=dev); dist = torch.zeros(nq * 2, dtype=torch.int32, device=dev)
            sec = ev_time(lambda: L.orbx_knn2_allpairs_device(ex._ctx, ptr(q.data_ptr()), nq, ptr(t.data_ptr()), nt, ptr(idx.data_ptr()), ptr(dist.data_ptr()), ptr(st)),
                          40 if nq < 4000 else 8)
            ach = nq * nt / sec
            knn[name] = {"queries": nq, "train": nt, "us_per_call": round(sec * 1e6, 2), "achieved": round(ach / 1e9, 2), "peak": round(valu_pairs_peak / 1e9, 1),
                         "unit": "Gpairs/s", "frac": round(ach / valu_pairs_peak, 4)}
        out["k_knn2"] = dict(knn, bound="valu-int32", peak_derivation="16 lane-ops per 256-bit pair (8 xor + 8 popcount-accumulate) at one wave-instruction per 4 cycles: "
                             "1024 SIMDs x 2.4 GHz / 4 x 64 / 16 (profiles/valu_ceiling_r2.txt measures 0.59-0.60 T wave-instr/s for v_bcnt_u32_b32)")
        # ---- k_window: frame 0's keypoints as map points projected into frame 1 (th = 15 as TrackWithMotionModel, src/Tracking.cc:2889)
        m = ORBmatcher(ex, 0.9, True)
        grid = dict(min_x=0.0, min_y=0.0, inv_w=64.0 / float(host_frames.shape[2]), inv_h=48.0 / float(host_frames.shape[1]), cell_start=None, cell_idx=None)
        T = m.Target(k1, d1h, grid)
        lvl = k0["octave"].astype(np.int32)
        qx, qy = (k0["x"] + np.float32(1.5)).astype(np.float32), (k0["y"] + np.float32(0.5)).astype(np.float32)
        qr = (np.float32(15.0) * np.float32(1.2) ** lvl).astype(np.float32)
        ex.set_option("window_timing", 1)
        us, nnz = [], 0
        for i in range(60):
            r = T.search(qx, qy, qr, lvl - 1, lvl + 1, d0h, want_lists=True)
            nnz = int(r["row_ptr"][-1])
            if i >= 10:
                us.append(float(L.orbx_last_window_device_us(ex._ctx)))
        ex.set_option("window_timing", 0)
        T.close()
        sec = float(np.median(us)) * 1e-6
        Q, Tn = len(k0), len(k1)
        wbytes = 32 * (Q + Tn) + 4 * nnz + 12 * Q
        out["k_window"] = {"queries": Q, "target_keypoints": Tn, "candidates": nnz, "us_per_call": round(sec * 1e6, 2), "algorithmic_bytes": wbytes,
                           "achieved": round(wbytes / sec / 1e9, 2), "peak": 8000.0, "unit": "GB/s", "bound": "hbm", "frac": round(wbytes / sec / 8e12, 6),
                           "note": "one launch of ~10 us over 100 KB: launch- and latency-bound; the call's wall time (pack + launch + poll) is in streamed_frontend"}
        # ---- BoW descent
        tmp = tempfile.mkdtemp(prefix="orbx_mr_")
        try:
            vp = os.path.join(tmp, "voc.txt")
            info = make_vocabulary(vp, np.concatenate(list(voc_descriptors) + [d0h, d1h]), 10, 4, seed=9)
            voc = ORBVocabulary(ex)
            assert voc.loadFromTextFile(vp)
            dq = torch.from_numpy(d0h).to(dev)
            n = int(dq.shape[0])
            dw = torch.zeros(n, dtype=torch.int32, device=dev); dwt = torch.zeros(n, dtype=torch.float64, device=dev); dn = torch.zeros(n, dtype=torch.int32, device=dev)
            sec = ev_time(lambda: L.orbx_bow_transform_device(voc._voc, ptr(dq.data_ptr()), n, 4, ptr(dw.data_ptr()), ptr(dwt.data_ptr()), ptr(dn.data_ptr()), ptr(st)), 40)
            k_, L_ = 10, 4
            gb = n * L_ * k_ * 32
            out["k_bow_descend"] = {"features": n, "vocabulary": {"k": k_, "L": L_},
                                    "us_per_call": round(sec * 1e6, 2), "gathered_bytes": gb, "achieved": round(gb / sec / 1e9, 2), "peak": 34500.0, "unit": "GB/s",
                                    "bound": "l2", "frac": round(gb / sec / 34.5e12, 6),
                                    "note": "L dependent gather rounds of k child descriptors per feature on a 1000-feature call: latency-bound, not bandwidth-bound"}
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
        return out
    except SystemExit:
        raise
    except Exception as e:   # noqa: BLE001
        out["error"] = f"{type(e).__name__}: {e}"[:300]
        return out


def fixed_streams_leg(args, ex, dev, rank, world, cdev, sync_all, reduce_max, want_gather, nstreams=8):
    """SURVEY 8(e)'s curve in the SAME run as the weak line: a FIXED set of `nstreams` camera streams (S-8cam), stream c on GPU c mod G,
    --batch frames per stream and step — total work per step the same for every G (strong scaling), so the driver's N = 1, 2, 4, 8 runs give the
    survey's frames/s curve from this record while `value` stays the contract's weak line.  Same engine, lanes, exchange; one batch set (at
    2048 frames per step the level-0 input is 630 MB: nothing of it survives in the Infinity Cache between steps anyway)."""
    import torch
    import torch.distributed as dist
    from orb_slam3_modified_amd import synth
    from orb_slam3_modified_amd.replay import ReplayEngine, shard_streams
    if nstreams % world:
        return {"skipped": f"{nstreams} streams do not divide over {world} ranks evenly"}
    cams = shard_streams(nstreams, world, rank)
    Bs = args.batch * len(cams)
    host = np.concatenate([synth.make_stream(args.batch, args.rows, args.cols, synth.DEFAULT_SEED + 1000 * c) for c in cams])
    frames = torch.from_numpy(host).to(dev)
    exs = ex.clone()
    err = None
    try:
        with stdout_to_stderr():
            eng = ReplayEngine(exs, frames, lapping=(0, 1000), gather=want_gather, lanes=args.lanes, gather_what=args.gather)
    except Exception as e:   # noqa: BLE001 — every rank fails together here (replay.py agrees over the control plane before anybody connects)
        err = f"{type(e).__name__}: {e}"[:300]
        eng = ReplayEngine(exs, frames, lapping=(0, 1000), gather=False, lanes=args.lanes, gather_what=args.gather, rank=rank, world=world)
    steps = max(3, min(args.steps, (args.steps * 256 + Bs - 1) // Bs))   # about the weak line's number of frames per repeat
    repeats = 3
    dts, last = [], 0
    for r in range(repeats):
        d, last = timed_replay(eng, steps, 2 if r == 0 else 0, sync_all)
        dts.append(d)
    dts = reduce_max(dts)
    dt = sorted(dts)[repeats // 2]
    feats = torch.tensor([int(eng.counts(last)[:, 0].sum())], dtype=torch.int64, device=cdev)
    if world > 1:
        dist.all_reduce(feats, op=dist.ReduceOp.SUM)
    gms = eng.gather_ms() if eng.gather else None
    out = {"what": f"SURVEY 8(e): the same {nstreams} camera streams for every G, stream c -> GPU c mod G, {args.batch} frames per stream and step",
           "scaling": "strong", "streams": nstreams, "n_gpus": world, "frames_per_step_total": Bs * world, "frames_per_step_per_gpu": Bs,
           "steps": steps, "repeats": repeats, "ms_per_step": round(dt / steps * 1e3, 4),
           "ms_per_step_min_max": [round(min(dts) / steps * 1e3, 4), round(max(dts) / steps * 1e3, 4)],
           "frames_per_s": round(Bs * world * steps / dt, 1), "value": round(int(feats.item()) * steps / (dt * 1e3), 1), "unit": "features/ms",
           "exchange": ({"transport": eng.transport, "bytes_per_rank_per_step": int(eng.send_bytes), "gather_ms": (round(gms, 4) if gms else None)}
                        if eng.gather else ({"error": err} if err else "none"))}
    eng.close()
    del eng, frames
    torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=None, help="frames per step per camera stream (default 256, also with --streams: at G = 8 every GPU "
                                                            "then still steps over 256 frames = two 128-frame lanes; 64-frame steps lose 20 %% to short launches)")
    ap.add_argument("--batches", type=int, default=None, help="distinct batches the steps rotate through (default 4; 2 with --streams)")
    ap.add_argument("--rows", type=int, default=480)
    ap.add_argument("--cols", type=int, default=640)
    ap.add_argument("--nfeatures", type=int, default=1000)
    ap.add_argument("--no-gather", action="store_true", help="N>1: skip the batch-replay RCCL all-gather")
    ap.add_argument("--gather", choices=["descriptors", "blocks"], default="descriptors",
                    help="what the per-step all-gather moves: descriptor rows + counts (north_star) or whole feature blocks incl. keypoints")
    ap.add_argument("--streams", type=int, default=0,
                    help="SURVEY 8(e)'s curve: a FIXED set of S camera streams (S-8cam: --streams 8) sharded stream c -> GPU c mod G, "
                         "--batch frames per stream and step (strong scaling).  0 (default): one stream per GPU (weak scaling)")
    ap.add_argument("--no-fixed-streams", action="store_true", help="skip the `strong` record (the fixed-8-stream curve of SURVEY 8(e)) that the default weak "
                                                                     "line carries beside it")
    ap.add_argument("--no-frontend", action="store_true", help="skip the streamed front-end leg (operator() + BoW + two guided searches per frame)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the config-4 (1024x1024, 2000 features) leg and the operator() leg")
    ap.add_argument("--no-verify", action="store_true", help="timing experiments only: skip the oracle check of the last step")
    ap.add_argument("--cpu-budget", type=float, default=24.0, help="seconds of CPU work for the cpu_baseline leg")
    ap.add_argument("--lanes", type=int, default=int(os.environ.get("ORBX_LANES", "2")),
                    help="extractor contexts per GPU, each on its own free-running stream over 1/lanes of the batch")
    ap.add_argument("--repeats", type=int, default=9, help="the --steps-long timed loop is run this many times (each bracketed by barrier + synchronize): "
                                                          "ms_per_step / value are the MEDIAN repeat's, min / max / spread are reported beside them")
    ap.add_argument("--profile", default="opencv>=4.5.1", help="which CPU path the output must equal (orbx_set_cpu_profile; INTEGRATION.md section 6): "
                                                              "opencv>=4.5.1 | opencv-4.4 | opencv-4.4-sse | opencv-4.4-avx512 | opencv-4.4-scalar | opencv-3.2")
    ap.add_argument("--fma-build", type=int, default=0, help="bit 0: the reference built with -march=native on an FMA machine; bit 1: OpenCV's AVX2 fastAtan2")
    ap.add_argument("--control-backend", choices=["nccl", "gloo"], default="nccl",
                    help="torch.distributed backend of the job's CONTROL plane (barrier, MAX of times, the ncclUniqueId's broadcast).  nccl (default): the "
                         "exchange is RCCL called by liborbx.  gloo: the exchange takes the engine's host transport — with --share-gpu this is how the N > 1 code "
                         "path of this script is exercised on a one-GPU box (tests/test_bench_contract.py); never a configuration to quote")
    ap.add_argument("--share-gpu", action="store_true", help="testing only: every rank uses cuda:0 (RCCL refuses a shared device: needs --control-backend gloo)")
    args = ap.parse_args()
    args.repeats = max(1, args.repeats) | 1      # odd: the median is a measured repeat
    if args.batch is None:
        args.batch = 256   # per camera stream; SURVEY 8(e)'s B = 64 leaves a GPU two 32-frame lanes at G = 8 (measured: -20 %)
    if args.batches is None:
        args.batches = 2 if args.streams > 0 else 4

    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    requested = max(1, args.gpus)
    if "WORLD_SIZE" not in os.environ and requested > 1:
        n = min(requested, torch.cuda.device_count())
        if n > 1:   # become the launcher: one rank per GPU over RCCL
            s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
            os.execv(sys.executable, [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr",
                                      "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:])
    rank = int(os.environ.get("RANK", "0"))
    local_rank = 0 if args.share_gpu else int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.share_gpu and args.control_backend == "nccl" and world > 1:
        raise SystemExit("bench.py: --share-gpu needs --control-backend gloo (RCCL refuses two ranks on one device)")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        with stdout_to_stderr():   # the communicator (and RCCL's banner) now, not inside the first timed collective
            if args.control_backend == "nccl":
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
            else:
                dist.init_process_group("gloo", rank=rank, world_size=world)
            dist.barrier()
        if dist.get_world_size() != world:
            raise SystemExit(f"bench.py: WORLD_SIZE {world} but the process group has {dist.get_world_size()} ranks")
    dev = torch.device("cuda", local_rank)
    cdev = dev if args.control_backend == "nccl" else torch.device("cpu")   # where the control plane's little tensors live

    from orb_slam3_modified_amd import ORBextractor, synth
    from orb_slam3_modified_amd.replay import ReplayEngine

    from orb_slam3_modified_amd.replay import shard_streams
    H, W = args.rows, args.cols
    # S-8cam: camera c = S-EuRoC-640 stream with seed + 1000*c (+ 101*k for batch k): `batches` x B frames per camera, every frame
    # distinct (4 x 256 x 300 KB = 315 MB of level-0 input, more than the 256 MB Infinity Cache).  One generator run per batch:
    # a single longer stream would wander into the scene's flat quarter and lose the configuration's ~1000 features per frame.
    # Default (weak scaling): camera `rank` on GPU `rank`.  --streams S (strong scaling): the same S cameras for every G,
    # camera c on GPU c mod G, so a rank steps through (its cameras) x --batch frames.
    nsets = max(1, args.batches)
    if args.streams > 0:
        if args.streams % world:
            raise SystemExit(f"bench.py: --streams {args.streams} does not divide over {world} ranks evenly")
        cams = shard_streams(args.streams, world, rank)
    else:
        cams = [rank]
    B = args.batch * len(cams)                       # frames per step on this rank
    host_frames = np.concatenate([synth.make_stream(args.batch, H, W, synth.DEFAULT_SEED + 1000 * c + 101 * k) for k in range(nsets) for c in cams])
    frame_sets = [torch.from_numpy(host_frames[k * B:(k + 1) * B]).to(dev) for k in range(nsets)]
    ex = ORBextractor(args.nfeatures, 1.2, 8, 20, 7, device_id=local_rank)
    ex.set_cpu_profile(args.profile, args.fma_build)     # the default arguments name what orbx_create gives anyway; clones inherit it
    profile_text, profile_vals = ex.cpu_profile()
    args.variant = tuple(profile_vals.values())          # the oracle checks under the same five values
    exchange_error = None
    want_gather = world > 1 and not args.no_gather
    try:   # N > 1: rank 0 makes the ncclUniqueId, the job's control plane broadcasts it, liborbx calls ncclCommInitRank / ncclAllGather itself
        with stdout_to_stderr():
            eng = ReplayEngine(ex, frame_sets, lapping=(0, 1000), gather=want_gather, lanes=args.lanes, gather_what=args.gather)
    except Exception as e:   # noqa: BLE001 — the sharded extraction needs no collective: measure it, and say loudly that the exchange did not come up
        if not want_gather:
            raise
        exchange_error = f"{type(e).__name__}: {e}"[:300]
        eng = None
    if world > 1:
        flag = torch.tensor([1 if eng is None else 0], dtype=torch.int32, device=cdev)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        if int(flag.item()) and want_gather:     # one rank without a communicator: nobody may enter the collective
            if eng is not None:
                eng.close()
            exchange_error = exchange_error or "another rank could not create its RCCL communicator"
            eng = ReplayEngine(ex, frame_sets, lapping=(0, 1000), gather=False, lanes=args.lanes, gather_what=args.gather, rank=rank, world=world)

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def reduce_max(vals):
        t = torch.tensor(list(vals), dtype=torch.float64, device=cdev)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return [float(v) for v in t.tolist()]

    # ---- the timed region, `repeats` times: W untimed warm-up steps once, then every repeat = EXACTLY K steps bracketed by barrier + synchronize on
    # both sides (the contract's unit), MAX over ranks per repeat; the line reports the MEDIAN repeat (a measured one: repeats is odd)
    dts, lasts, starts = [], [], []
    for r in range(args.repeats):
        starts.append(eng.step_idx + (args.warmup if r == 0 else 0))
        dt_r, last_r = timed_replay(eng, args.steps, args.warmup if r == 0 else 0, sync_all)
        dts.append(dt_r); lasts.append(last_r)
    dts = reduce_max(dts)
    order = sorted(range(args.repeats), key=lambda r: dts[r])
    rmed = order[args.repeats // 2]
    dt, last = dts[rmed], lasts[-1]
    # the exchange: device time of one step's collective (HIP events on the gather stream), and the same timed loop with / without it,
    # alternating repeats -> how much of the collective the next step's kernels hide
    exchange = None
    if eng.gather:
        exchange = {"collective": f"ncclAllGather({args.gather}) called by liborbx (orbx_replay_step), one per step, async on its own stream, double-buffered",
                    "transport": eng.transport, "bytes_per_rank_per_step": int(eng.send_bytes), "bytes_received_per_rank_per_step": int(eng.send_bytes * (world - 1))}
        exchange.update(exchange_legs(eng, args.steps, max(3, args.repeats // 2 + 1), sync_all, reduce_max))
        g = exchange.get("gather_ms")
        exchange["overlap_frac"] = round(1.0 - min(1.0, max(0.0, exchange["exposed_ms_per_step"]) / g), 4) if g else None
        last = eng.step(); eng.drain(); sync_all()   # a fresh gathered step so that `last` below refers to real data again
    elif exchange_error:
        exchange = {"error": exchange_error, "note": "the RCCL exchange did not come up: the line measures the sharded extraction WITHOUT the per-step all-gather"}

    counts = eng.counts(last)
    # ---- the measured path must be the right path: frames of the last timed step against the CPU oracle (every rank its own)
    last_set = (eng.step_idx - 1) % nsets
    last_block = eng.block_host(last)
    # keypoints of every batch of the rotation (one untimed step each): timed step s processed batch s mod nsets
    per_set = [0] * nsets
    for _ in range(nsets):
        k = eng.step_idx % nsets
        i = eng.step()
        per_set[k] = int(eng.counts(i)[:, 0].sum())
    eng.write_block(last, last_block)
    feats_of = lambda r: sum(per_set[(starts[r] + s_) % nsets] for s_ in range(args.steps))   # noqa: E731
    total_feats = torch.tensor([feats_of(rmed)], dtype=torch.int64, device=cdev)
    if world > 1:
        dist.all_reduce(total_feats, op=dist.ReduceOp.SUM)
    total_feats = int(total_feats.item())
    total_frames = B * world * args.steps
    value = total_feats / (dt * 1e3)

    lane_edges = sorted({0, B - 1, B // 2 - 1, B // 2} | {f for (f0, f1) in eng.lane_ranges for f in (f0, f1 - 1)} | {B // 3})
    verified = 0 if args.no_verify else verify_block(eng, last, host_frames[last_set * B:(last_set + 1) * B], lane_edges, args.nfeatures, (0, 1000), args.variant)
    vt = torch.tensor([verified], dtype=torch.int64, device=cdev)
    if world > 1:
        dist.all_reduce(vt, op=dist.ReduceOp.SUM)

    strong = None
    if args.streams == 0 and not args.no_fixed_streams:   # every rank takes part
        strong = fixed_streams_leg(args, ex, dev, rank, world, cdev, sync_all, reduce_max, want_gather and exchange_error is None)

    if rank == 0:
        dom, nkp, fused, roof = kernel_roofline(ex, eng, frame_sets[0], B, H, W, counts, world, args.steps, dt)
        # HBM traffic and issue-side counters of the dominant kernel come from committed rocprofv3 --pmc passes (tools/pmc_traffic.py,
        # tools/pmc_sq.py): they are only reported when the files are stamped with the hash of the kernel sources this run was built from
        from orb_slam3_modified_amd.build import kernels_hash
        khash = kernels_hash()
        roof["kernels_hash"] = khash
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc):
            try:
                j = json.load(open(pmc))
                st = j.get("stamp", {})
                if j.get("batch") == B and j.get("rows") == H and j.get("cols") == W:
                    ent = j.get("kernels", {}).get(dom.split("(")[0])
                    if ent and st.get("kernels_hash") == khash:
                        # FETCH_SIZE + WRITE_SIZE from separate passes, corrected with the known-traffic calibration copy
                        roof["traffic"] = int(ent["hbm_bytes_per_launch"])
                        roof["traffic_source"] = (f"profiles/pmc_traffic.json: builder-run rocprofv3 --pmc passes of tools/pmc_traffic.py on this workload, commit "
                                                  f"{st.get('commit')} of {st.get('date')}, kernel sources {khash} = this build; NOT measured in this run")
                    elif ent:
                        roof["traffic_source"] = (f"profiles/pmc_traffic.json is STALE: measured on kernel sources {st.get('kernels_hash')} (commit {st.get('commit')}), "
                                                  f"this build is {khash}: traffic not reported")
            except Exception:
                pass
        sqp = os.path.join(ROOT, "profiles", "pmc_sq.json")
        if os.path.exists(sqp):
            try:
                j = json.load(open(sqp))
                st = j.get("stamp", {})
                issue = j.get("derived", {}).get(dom.split("(")[0])
                if issue and st.get("kernels_hash") == khash:
                    roof["issue_limits_pmc"] = {k: round(float(v), 4) for k, v in issue.items()}
                    # valu_busy prices every VALU instruction at 4 cycles; the kernel's own mix holds 2-cycle instructions (profiles/isa_weighted_r3.md:
                    # weighted price 0.85 of that): the fraction of the ISA-weighted issue ceiling
                    if dom.startswith("k_fast_cells") and "valu_busy" in issue:
                        roof["issue_limits_pmc"]["valu_busy_isa_weighted"] = round(float(issue["valu_busy"]) * 0.85, 4)
                    roof["issue_limits_pmc"]["source"] = (f"profiles/pmc_sq.json: builder-run rocprofv3 --pmc pass (tools/pmc_sq.py), commit {st.get('commit')} of "
                                                          f"{st.get('date')}, kernel sources {khash} = this build; NOT measured in this run")
                elif issue:
                    roof["issue_limits_pmc"] = {"source": f"profiles/pmc_sq.json is STALE (kernel sources {st.get('kernels_hash')}, this build {khash}): not reported"}
            except Exception:
                pass
        step_ms = dt / args.steps * 1e3
        rr = rocprof_row(dom, B, H, W, khash) if dom.startswith("k_fast_cells") else None
        if rr:
            roof["rocprof"] = dict(rr, note="committed rocprofv3 --kernel-trace summary of this command on THIS build's kernel sources (builder-run, another box); "
                                            "the roofline's own time is avg_launch_ms, measured in this run")
            roof["rocprof"]["hip_events_over_rocprof"] = round(roof["avg_launch_ms"] / rr["rocprof_avg_ms"], 4) if rr["rocprof_avg_ms"] else None
        k_ms = 1e3 / args.steps
        result = {
            "metric": "ORB features/ms (+ frames/s), 640x480 8-level pyramid, 1000 features/frame",
            "value": round(value, 1), "unit": "features/ms", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(step_ms, 4), "higher_is_better": True, "scaling": "strong" if args.streams > 0 else "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "frames_per_s": round(total_frames / dt, 1),
            "timing": {"repeats": args.repeats, "statistic": "ms_per_step / value / frames_per_s are the MEDIAN repeat's; every repeat = `steps` steps bracketed by "
                                                          "barrier + torch.cuda.synchronize() on both sides, MAX over ranks",
                       "ms_per_step_min": round(min(dts) * k_ms, 4), "ms_per_step_max": round(max(dts) * k_ms, 4),
                       "ms_per_step_all": [round(v * k_ms, 4) for v in dts], "spread_frac": round((max(dts) - min(dts)) / dt, 4),
                       "timed_region_s": round(sum(dts), 4), "median_repeat_s": round(dt, 5)},
            "verified_frames": int(vt.item()),
            "config": {"workload": f"S-EuRoC-640 batch replay: {B} frames/step/GPU of {W}x{H} u8, 8 levels sf 1.2, "
                                   f"nfeatures {args.nfeatures}, iniTh 20 minTh 7, mono lapping [0,1000]; frames resident in HBM, "
                                   f"results left in HBM; the steps rotate through {nsets} distinct batches ({nsets * B} distinct frames per GPU)",
                       "frames_per_step_per_gpu": B, "distinct_batches": nsets, "features_per_frame": round(float(nkp), 1),
                       "cpu_path_profile": {"active": profile_text, "options": profile_vals,
                                            "meaning": "which build of the reference CPU path the timed contexts (and the oracle that verified them) compute, bit for "
                                                       "bit: INTEGRATION.md section 6; the default is what orbx_create gives (OpenCV >= 4.5.1 blur, unfused fastAtan2 and "
                                                       "pattern rotation); the reference names OpenCV 4.4.0 / 3.2.0 and builds -march=native: --profile opencv-4.4 --fma-build 3"},
                       "cv_primitives": "recalled",   # cv::resize / FAST / GaussianBlur / fastAtan2 as the oracle restates them: no OpenCV exists where this was built and
                                                       # verified; tools/opencv_pin/run.sh is the maintainer's one-command pin against a real one
                       "exchange": (f"ncclAllGather({args.gather}) by liborbx, async/overlapped" if eng.gather else "none"),
                       "lanes_per_gpu": len(eng.lane_ranges), "requested_gpus": requested,
                       "lane_schedule": ("alternate: the lanes take whole steps in turn (step k on lane k mod L over all frames; two steps in flight)" if eng.alternate
                                         else "split: every lane works on its share of every step"),
                       "streams": (args.streams if args.streams > 0 else world), "frames_per_stream_per_step": args.batch,
                       "scaling_mode": (f"strong: the same {args.streams} camera streams for every G, stream c -> GPU c mod G (SURVEY 8(e))" if args.streams > 0
                                        else "weak: one camera stream per GPU, per-GPU work fixed; SURVEY 8(e)'s fixed-8-stream curve is the `strong` record of this line"),
                       "parallelism": (f"{args.streams} camera streams over {world} GPUs" if args.streams > 0 else f"one camera stream per GPU x{world}")},
            "roofline": roof,
        }
        if exchange is None and world == 1 and not args.no_gather:
            with stdout_to_stderr():
                exchange = self_gather_exchange(ex, frame_sets, args, step_ms, sync_all)
        if exchange is not None:
            exchange["ranks"] = world
            result["exchange"] = exchange
        if strong is not None:
            result["strong"] = strong
        if world == 1 and not args.no_frontend:
            try:
                fex = ORBextractor(args.nfeatures, 1.2, 8, 20, 7, device_id=local_rank)
                fex.set_cpu_profile(args.profile, args.fma_build)
                vdesc = [fex(host_frames[t], None, (0, 1000))[2] for t in range(0, 48, 8)]
                del fex
                result["streamed_frontend"] = streamed_frontend(host_frames, args.nfeatures, vdesc)
            except SystemExit:
                raise
            except Exception as e:   # noqa: BLE001
                result["streamed_frontend"] = {"error": str(e)[:300]}
            result["frame_constructor"] = frame_constructor()
            result["matcher_roofline"] = matcher_roofline(host_frames, args.nfeatures, [])
        if world == 1 and not args.no_secondary:
            result["end_to_end_operator"] = e2e_operator(host_frames, args.nfeatures)
            # ---- BASELINE config 4: TUM-VI shape, 1024x1024, 2000 features (large-image configuration)
            B4, H4, W4, NF4, steps4 = 64, 1024, 1024, 2000, 12
            host4 = synth.make_stream(B4, H4, W4, synth.DEFAULT_SEED + 77)
            frames4 = torch.from_numpy(host4).to(dev)
            ex4 = ORBextractor(NF4, 1.2, 8, 20, 7, device_id=local_rank)
            ex4.set_cpu_profile(args.profile, args.fma_build)
            eng4 = ReplayEngine(ex4, frames4, lapping=(0, 1000), gather=False, lanes=args.lanes)
            dt4, dmm4, last4 = timed_median(eng4, steps4, 3, sync_all)
            c4 = eng4.counts(last4)
            v4 = verify_block(eng4, last4, host4, [0, B4 // 2, B4 - 1], NF4, (0, 1000), args.variant)
            _, nkp4, _, roof4 = kernel_roofline(ex4, eng4, frames4, B4, H4, W4, c4, 1, steps4, dt4, nprof=3)
            result["secondary"] = {"workload": f"S-TUMVI-1024 batch replay (BASELINE config 4): {B4} frames/step of {W4}x{H4} u8, nfeatures {NF4}",
                                   "value": round(float(c4[:, 0].sum()) * steps4 / (dt4 * 1e3), 1), "unit": "features/ms",
                                   "frames_per_s": round(B4 * steps4 / dt4, 1), "ms_per_step": round(dt4 / steps4 * 1e3, 4), "steps": steps4,
                                   "repeats": 3, "ms_per_step_min_max": [round(v_ / steps4 * 1e3, 4) for v_ in dmm4],
                                   "features_per_frame": round(nkp4, 1), "verified_frames": v4, "roofline": roof4}
            eng4.close()
            del eng4, ex4, frames4
            result["secondary_natural"] = natural_leg(args, dev, local_rank, sync_all)
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(host_frames[:64], args.nfeatures, args.cpu_budget)
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
    if dist.is_initialized():
        with stdout_to_stderr():
            dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
