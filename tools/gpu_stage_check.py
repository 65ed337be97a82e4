#!/usr/bin/env python3
"""Stage-by-stage GPU-vs-oracle diagnosis (development tool; the pass/fail gate is tests/ -m gpu).

Runs the HIP extractor and the CPU oracle on the same synthetic frames and reports, per pyramid level, the
first stage that differs: pyramid bytes, FAST candidate list, quadtree keypoints, final keypoints, descriptors.
Writes a report to gpurun_out/stage_check.txt.
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import pyoracle as po  # noqa: E402
from orb_slam3_modified_amd import ORBextractor, ORBmatcher, ORBVocabulary, synth  # noqa: E402
from orb_slam3_modified_amd.vocabulary import write_text_vocabulary  # noqa: E402

OUT = []


def say(*a):
    s = " ".join(str(x) for x in a)
    print(s, flush=True)
    OUT.append(s)


def check_frame(gpu, ora, img, lap, tag):
    ok = True
    kps_o, desc_o, mono_o = ora.extract(img, lap)
    mono_g, kps_g, desc_g = gpu(img, None, lap)
    for l in range(gpu.nlevels):
        a, b = gpu.pyramid_level(l), ora.level(l)
        if a.shape != b.shape or not np.array_equal(a, b):
            nd = int((a != b).sum()) if a.shape == b.shape else -1
            say(f"  [{tag}] L{l} PYRAMID differs: shapes {a.shape} {b.shape} ndiff={nd}")
            if a.shape == b.shape:
                ys, xs = np.nonzero(a != b)
                say("     first diffs:", [(int(y), int(x), int(a[y, x]), int(b[y, x])) for y, x in list(zip(ys, xs))[:6]])
            ok = False
            continue
        gx, gy, gs = gpu.debug_level_points(l, 0)
        c = ora.level_keypoints(l, 0)
        ox, oy, osc = c["x"].astype(np.int32), c["y"].astype(np.int32), c["response"].astype(np.int32)
        if len(gx) != len(ox) or not (np.array_equal(gx, ox) and np.array_equal(gy, oy) and np.array_equal(gs, osc)):
            say(f"  [{tag}] L{l} CANDIDATES differ: gpu {len(gx)} oracle {len(ox)}")
            sg = set(zip(gx.tolist(), gy.tolist(), gs.tolist())); so = set(zip(ox.tolist(), oy.tolist(), osc.tolist()))
            say(f"     as sets: only-gpu {len(sg - so)} only-oracle {len(so - sg)}; e.g. {sorted(sg - so)[:4]} | {sorted(so - sg)[:4]}")
            n = min(len(gx), len(ox))
            d = np.nonzero((gx[:n] != ox[:n]) | (gy[:n] != oy[:n]) | (gs[:n] != osc[:n]))[0]
            if len(d):
                i = int(d[0])
                say(f"     first order diff at {i}: gpu {(gx[i], gy[i], gs[i])} oracle {(ox[i], oy[i], osc[i])}")
            ok = False
            continue
        gx, gy, gs = gpu.debug_level_points(l, 1)
        k = ora.level_keypoints(l, 1)
        ox, oy, osc = k["x"].astype(np.int32), k["y"].astype(np.int32), k["response"].astype(np.int32)
        if len(gx) != len(ox) or not (np.array_equal(gx, ox) and np.array_equal(gy, oy) and np.array_equal(gs, osc)):
            say(f"  [{tag}] L{l} QUADTREE differs: gpu {len(gx)} oracle {len(ox)} (candidates {len(c)})")
            sg = set(zip(gx.tolist(), gy.tolist())); so = set(zip(ox.tolist(), oy.tolist()))
            say(f"     as sets: only-gpu {len(sg - so)} only-oracle {len(so - sg)}")
            n = min(len(gx), len(ox))
            d = np.nonzero((gx[:n] != ox[:n]) | (gy[:n] != oy[:n]))[0]
            if len(d):
                say(f"     first order diff at {int(d[0])}")
            ok = False
    if mono_g != mono_o or len(kps_g) != len(kps_o):
        say(f"  [{tag}] COUNT differs: gpu n={len(kps_g)} mono={mono_g} oracle n={len(kps_o)} mono={mono_o}")
        ok = False
    else:
        for f in ("x", "y", "size", "angle", "response", "octave", "class_id"):
            if not np.array_equal(kps_g[f].view(np.int32), kps_o[f].view(np.int32)):
                d = np.nonzero(kps_g[f].view(np.int32) != kps_o[f].view(np.int32))[0]
                say(f"  [{tag}] KEYPOINT field {f} differs at {len(d)} of {len(kps_g)}; e.g. idx {int(d[0])}: "
                    f"gpu {kps_g[f][d[0]]!r} oracle {kps_o[f][d[0]]!r}")
                ok = False
        if not np.array_equal(desc_g, desc_o):
            rows = np.nonzero((desc_g != desc_o).any(axis=1))[0]
            bits = int(np.unpackbits(desc_g ^ desc_o).sum())
            say(f"  [{tag}] DESCRIPTORS differ: {len(rows)} rows of {len(desc_g)}, {bits} bits; first row {int(rows[0])}")
            ok = False
    return ok, len(kps_o)


def main():
    t0 = time.time()
    all_ok = True
    configs = [
        ("A 640x480 nf1000", 480, 640, 1000, (0, 1000)),
        ("native 752x480", 480, 752, 1000, (0, 1000)),
        ("native 600x350", 350, 600, 1000, (0, 1000)),
        ("B 1024x1024 nf2000", 1024, 1024, 2000, (0, 1000)),
        ("ini 640x480 nf5000", 480, 640, 5000, (0, 1000)),
        ("stereo-lap 640x480", 480, 640, 1000, (0, 0)),
        ("fisheye-lap 640x480", 480, 640, 1200, (200, 400)),
    ]
    for name, h, w, nf, lap in configs:
        frames = synth.make_stream(2, h, w)
        gpu = ORBextractor(nf, 1.2, 8, 20, 7)
        ora = po.OracleExtractor(nf, 1.2, 8, 20, 7)
        for t, img in enumerate(frames):
            ok, n = check_frame(gpu, ora, img, lap, f"{name} f{t}")
            say(f"{'OK  ' if ok else 'FAIL'} {name} frame {t}: {n} keypoints")
            all_ok &= ok
        # batch path must equal the single-frame path
        res = gpu.extract_batch(frames, lap)
        for t, img in enumerate(frames):
            kps_o, desc_o, mono_o = ora.extract(img, lap)
            mono_g, kps_g, desc_g = res[t]
            same = mono_g == mono_o and len(kps_g) == len(kps_o) and kps_g.tobytes() == kps_o.tobytes() and np.array_equal(desc_g, desc_o)
            if not same:
                say(f"FAIL {name} batch frame {t}")
                all_ok = False
        gpu.close()
    # flat / degenerate images
    gpu = ORBextractor(1000, 1.2, 8, 20, 7)
    ora = po.OracleExtractor(1000, 1.2, 8, 20, 7)
    for name, img in (("constant", np.full((480, 640), 77, np.uint8)),
                      ("noise", np.random.default_rng(5).integers(0, 256, (480, 640)).astype(np.uint8)),
                      ("checker", ((np.indices((480, 640)).sum(0) // 8) % 2 * 200 + 20).astype(np.uint8))):
        ok, n = check_frame(gpu, ora, img, (0, 1000), name)
        say(f"{'OK  ' if ok else 'FAIL'} {name}: {n} keypoints")
        all_ok &= ok
    # matcher
    frames = synth.make_stream(2)
    _, k0, d0 = gpu(frames[0]); _, k1, d1 = gpu(frames[1])
    m = ORBmatcher(gpu)
    rng = np.random.default_rng(3)
    nq = len(d0)
    lens = rng.integers(0, 40, nq)
    rp = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    cand = rng.integers(0, len(d1), rp[-1]).astype(np.int32)
    for lw in (False, True):
        g = m.nn_csr(d0, d1, rp, cand, lw, want_dist=True)
        o = po.nn_csr(d0, d1, rp, cand, lw)
        same = all(np.array_equal(a, b) for a, b in zip(g, o))
        say(f"{'OK  ' if same else 'FAIL'} nn_csr last_wins={lw}")
        all_ok &= same
    gi, gd = m.knn2(d0, d1); oi, od = po.knn2(d0, d1)
    same = np.array_equal(gi, oi) and np.array_equal(gd, od)
    say(f"{'OK  ' if same else 'FAIL'} knn2 {len(d0)}x{len(d1)}")
    all_ok &= same
    # bag of words on a synthetic k=10, L=3 vocabulary
    from tests.vocab_util import make_vocabulary  # noqa: E402
    voc_path = "/tmp/orbx_voc_test.txt"
    make_vocabulary(voc_path, np.concatenate([d0, d1]), k=10, L=3, seed=7)
    gv = ORBVocabulary(gpu); assert gv.loadFromTextFile(voc_path)
    ov = po.OracleVocabulary(voc_path)
    (gi_, gvv), gfv = gv.transform(d0, 2)
    (oi_, ovv), ofv = ov.transform(d0, 2)
    same = np.array_equal(gi_, oi_) and gvv.tobytes() == ovv.tobytes() and gfv == ofv
    say(f"{'OK  ' if same else 'FAIL'} bow transform ({len(gi_)} words, {len(gfv)} nodes)")
    all_ok &= same
    b1 = gv.transform(d1, 2)[0]
    s_g, s_o = gv.score((gi_, gvv), b1), po.score_l1((oi_, ovv), ov.transform(d1, 2)[0])
    sb = gv.score_batch((gi_, gvv), [b1, (gi_, gvv)])
    same = s_g == s_o and sb[0] == s_o and sb[1] == po.score_l1((oi_, ovv), (oi_, ovv))
    say(f"{'OK  ' if same else 'FAIL'} bow score {s_g!r} vs {s_o!r}; batch {sb.tolist()}")
    all_ok &= same
    # quick timing
    B = 64
    frames = synth.make_stream(B)
    gpu.extract_batch(frames[:4], (0, 1000))
    t1 = time.time(); res = gpu.extract_batch(frames, (0, 1000)); dt = time.time() - t1
    nk = sum(len(r[1]) for r in res)
    say(f"host batch of {B}: {dt * 1e3:.1f} ms incl. H2D/D2H -> {nk / dt / 1e3:.1f} features/ms")
    gpu.profile_enable(True)
    gpu.extract_batch(frames, (0, 1000))
    prof = gpu.profile_read()
    gpu.profile_enable(False)
    for k, (ms, n) in prof.items():
        if n:
            say(f"   {k:28s} {ms:8.3f} ms / {n} launches")
    say(f"ALL {'OK' if all_ok else 'FAILED'}  ({time.time() - t0:.1f}s)")
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/stage_check.txt", "w") as f:
        f.write("\n".join(OUT) + "\n")
    return 0 if all_ok else 1


if __name__ == "__main__":
    sys.exit(main())
