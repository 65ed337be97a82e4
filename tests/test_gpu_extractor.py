"""Parity tests proper: the HIP extractor (through the C ABI) against the CPU oracle, bit for bit.

Keypoint fields are compared as raw 32-bit patterns (x, y, size, angle, response, octave, class_id), descriptors
byte for byte, the return value (monoIndex) and the output ORDER included.  `response` is the
integer FAST score (the reference never computes a Harris score), so the north-star's 1e-4
tolerance collapses to exact equality; the only float fields (angle, scaled x/y) are required bit-exact too.
"""
import os

import numpy as np
import pytest

from oracle import pyoracle as po
from orb_slam3_modified_amd import ORBextractor, OrbxError, synth

pytestmark = pytest.mark.gpu


def assert_same(gpu_res, ora_res, tag=""):
    mono_g, kps_g, desc_g = gpu_res
    kps_o, desc_o, mono_o = ora_res
    assert mono_g == mono_o, (tag, mono_g, mono_o)
    assert len(kps_g) == len(kps_o), (tag, len(kps_g), len(kps_o))
    for f in kps_g.dtype.names:
        a, b = kps_g[f].view(np.int32), kps_o[f].view(np.int32)
        assert np.array_equal(a, b), (tag, f, int((a != b).sum()))
    assert np.array_equal(desc_g, desc_o), (tag, "descriptor bits", int(np.unpackbits(desc_g ^ desc_o).sum()))


CONFIGS = [
    # (rows, cols, nfeatures, lapping, nframes)
    (480, 640, 1000, (0, 1000), 3),     # BASELINE configs[1]: single 640x480 frame, bit-exact check (mono call)
    (480, 752, 1000, (0, 1000), 2),     # EuRoC native: two quadtree roots
    (350, 600, 1000, (0, 1000), 2),     # EuRoC.yaml resized shape
    (512, 512, 1500, (0, 1000), 1),     # TUM-VI.yaml shape
    (480, 640, 5000, (0, 1000), 1),     # mpIniORBextractor (5 x nFeatures)
    (480, 640, 1000, (0, 0), 1),        # stereo / RGB-D call: everything ascending, returns N
    (480, 640, 1200, (200, 400), 1),    # fisheye lapping columns: both output branches
]


@pytest.mark.parametrize("rows,cols,nf,lap,nframes", CONFIGS)
def test_extract_bit_exact(rows, cols, nf, lap, nframes):
    frames = synth.make_stream(nframes, rows, cols)
    gpu = ORBextractor(nf, 1.2, 8, 20, 7)
    ora = po.OracleExtractor(nf, 1.2, 8, 20, 7)
    for t, img in enumerate(frames):
        assert_same(gpu(img, None, lap), ora.extract(img, lap), f"{cols}x{rows} nf{nf} f{t}")


def test_config4_tumvi_1024_both_output_branches():
    img = synth.make_stream(1, 1024, 1024)[0]
    gpu = ORBextractor(2000, 1.2, 8, 20, 7)
    res = gpu(img, None, (0, 1000))
    assert_same(res, po.OracleExtractor(2000, 1.2, 8, 20, 7).extract(img, (0, 1000)), "1024^2")
    mono, kps, _ = res
    assert 0 < mono < len(kps) and (kps["x"][:mono] > 1000).all()


def test_stage_parity_pyramid_candidates_quadtree():
    img = synth.make_stream(1)[0]
    gpu = ORBextractor(1000, 1.2, 8, 20, 7)
    ora = po.OracleExtractor(1000, 1.2, 8, 20, 7)
    gpu(img, None, (0, 1000)); ora.extract(img, (0, 1000))
    pyr = gpu.mvImagePyramid
    for l in range(8):
        assert np.array_equal(pyr[l], ora.level(l)), f"pyramid level {l}"
        gx, gy, gs = gpu.debug_level_points(l, 0)
        c = ora.level_keypoints(l, 0)
        assert np.array_equal(gx, c["x"].astype(np.int32)) and np.array_equal(gy, c["y"].astype(np.int32))
        assert np.array_equal(gs, c["response"].astype(np.int32)), f"FAST candidates level {l}"
        gx, gy, gs = gpu.debug_level_points(l, 1)
        k = ora.level_keypoints(l, 1)
        assert np.array_equal(gx, k["x"].astype(np.int32)) and np.array_equal(gy, k["y"].astype(np.int32)), f"quadtree level {l}"


@pytest.mark.parametrize("rows,cols", [(480, 640), (350, 600), (134, 179), (480, 752)])
def test_stage_parity_blurred_planes_full_frame_incl_borders(rows, cols):
    """Every pixel of every blurred level (reflect-101 borders, partial last tiles) against cv::GaussianBlur's restatement:
    the descriptors only sample >= 1 px inside, so the end-to-end test alone would not see a border mistake."""
    frames = synth.make_stream(2, rows, cols)
    gpu = ORBextractor(1000, 1.2, 4 if rows < 200 else 8, 20, 7)
    gpu.extract_batch(frames, (0, 1000))
    for f in range(2):
        for l in range(gpu.nlevels):
            assert np.array_equal(gpu.debug_blur_level(l, frame=f), po.gaussian_blur7(gpu.pyramid_level(l, frame=f))), (rows, cols, f, l)


@pytest.mark.parametrize("name", ["constant", "noise", "checker", "gradient", "saturated"])
def test_degenerate_images(name):
    rng = np.random.default_rng(5)
    img = {"constant": np.full((480, 640), 77, np.uint8),
           "noise": rng.integers(0, 256, (480, 640)).astype(np.uint8),
           "checker": ((np.indices((480, 640)).sum(0) // 8) % 2 * 200 + 20).astype(np.uint8),
           "gradient": np.tile(np.linspace(0, 255, 640).astype(np.uint8), (480, 1)),
           "saturated": np.where(rng.random((480, 640)) < 0.5, 0, 255).astype(np.uint8)}[name]
    gpu = ORBextractor(1000, 1.2, 8, 20, 7)
    assert_same(gpu(img, None, (0, 1000)), po.OracleExtractor(1000, 1.2, 8, 20, 7).extract(img, (0, 1000)), name)


@pytest.mark.parametrize("nlevels,sf,ini,mn", [(1, 1.2, 20, 7), (4, 1.5, 30, 10), (8, 1.2, 7, 7), (6, 1.1, 20, 0)])
def test_other_parameters(nlevels, sf, ini, mn):
    img = synth.make_stream(1, 360, 480)[0]
    gpu = ORBextractor(800, sf, nlevels, ini, mn)
    assert_same(gpu(img, None, (0, 1000)), po.OracleExtractor(800, sf, nlevels, ini, mn).extract(img, (0, 1000)),
                f"L{nlevels} sf{sf}")
    t = po.OracleExtractor(800, sf, nlevels, ini, mn).tables()
    assert np.array_equal(gpu.GetScaleFactors(), t["scale"]) and np.array_equal(gpu.features_per_level(), t["quota"])
    assert np.array_equal(gpu.GetInverseScaleSigmaSquares(), t["inv_sigma2"])


@pytest.mark.parametrize("ini,mn", [(20, 7), (12, 12), (7, 20), (35, 3), (20, 0), (0, 0)])
@pytest.mark.parametrize("kind", ["synth", "natural", "sparse"])
def test_the_two_threshold_cell_loop_on_batches_and_single_frames(ini, mn, kind):
    """src/ORBextractor.cc:826-850: cv::FAST at iniThFAST, and at minThFAST only where that left the cell empty.  The batch kernel runs the two
    passes literally (round 6: stages B-D at iniTh first), the fused single-frame launch one pass at minTh with the threshold chosen afterwards:
    both against the oracle — also with equal thresholds, with minTh ABOVE iniTh (the second run then finds nothing new: the result is iniTh's),
    with threshold 0, on natural texture (most cells hold an iniTh corner) and on an image where most cells need the second pass."""
    if kind == "synth":
        imgs = synth.make_stream(5, 480, 640, 77)
    elif kind == "natural":
        nat = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "natural_crops.npz"))
        base = [np.ascontiguousarray(nat[k]) for k in ("result_640x480_img", "pineapple_640x480_img")]
        imgs = np.stack([np.roll(base[i % 2], (11 * i, 17 * i), (0, 1)) for i in range(5)])
    else:   # low contrast: almost every corner lies between the two thresholds
        imgs = (synth.make_stream(5, 480, 640, 78).astype(np.float32) * 0.12 + 100).astype(np.uint8)
    ora = po.OracleExtractor(1000, 1.2, 8, ini, mn)
    gpu = ORBextractor(1000, 1.2, 8, ini, mn)
    want = [ora.extract(f, (0, 1000)) for f in imgs]
    for passes in (2, 1):          # "fast_passes": 2 = the batch default, 1 = one pass at minTh (what the single-frame launch always does)
        gpu.set_option("fast_passes", passes)
        for i, r in enumerate(gpu.extract_batch(imgs, (0, 1000))):
            assert_same(r, want[i], f"{kind} th {ini}/{mn} batch frame {i} passes {passes}")
    assert_same(gpu(imgs[2], None, (0, 1000)), want[2], f"{kind} th {ini}/{mn} single frame")


@pytest.mark.parametrize("qt_points,fused", [(512, 1), (512, 0), (384, 1), (256, 1), (1024, 1), (2048, 0)])
def test_quadtree_point_capacities_and_fused_passes(qt_points, fused):
    """The three homes of a level's candidates — LDS point buffers, the overflow form (points in HBM, node indices for twice the capacity in
    LDS, same thread-per-point passes) and the wave-per-node fallback beyond that — and the fused first passes on / off give the same
    keypoints: a small LDS capacity pushes the synthetic frames' levels (770 ... 250 candidates) through all of them; single frame,
    batch, and a natural crop with 2 275 corners on level 0."""
    frames = synth.make_stream(3)
    nat = np.load("tests/golden/natural_crops.npz")["pineapple_640x480_img"]
    gpu = ORBextractor(1000, 1.2, 8, 20, 7)
    gpu.set_option("qt_points", qt_points)
    gpu.set_option("qt_fused", fused)
    ora = po.OracleExtractor(1000, 1.2, 8, 20, 7)
    want = [ora.extract(f, (0, 1000)) for f in frames]
    for f, wnt in zip(frames, want):
        assert_same(gpu(f, None, (0, 1000)), wnt, f"single qt_points {qt_points}")
    assert_same(gpu(nat, None, (0, 1000)), ora.extract(nat, (0, 1000)), f"natural qt_points {qt_points}")
    batch = np.stack(list(frames) * 3 + [frames[0]])      # 10 frames: the batch launch shapes (big / small level groups)
    res = gpu.extract_batch(batch, (0, 1000))
    for i, r in enumerate(res):
        assert_same(r, want[i % 3] if i < 9 else want[0], f"batch frame {i} qt_points {qt_points}")
    res = gpu.extract_batch(np.stack([nat, frames[1], nat, frames[2], nat, nat]), (0, 1000))
    wn = ora.extract(nat, (0, 1000))
    for i, r in enumerate(res):
        assert_same(r, wn if i in (0, 2, 4, 5) else want[1 if i == 1 else 2], f"mixed batch frame {i} qt_points {qt_points}")


@pytest.mark.parametrize("qt_points", [256, 512, 1024])
def test_quadtree_point_capacities_with_two_and_three_roots(qt_points):
    """The same three homes of a level's candidates on images whose levels start from two (752 x 480) and three (1241 x 376) root nodes —
    the fused passes do not apply there, the overflow form and the fallback do."""
    for rows, cols, nf in ((480, 752, 1200), (376, 1241, 2000)):
        frames = synth.make_stream(2, rows, cols, 77 + cols)
        gpu = ORBextractor(nf, 1.2, 8, 20, 7)
        gpu.set_option("qt_points", qt_points)
        ora = po.OracleExtractor(nf, 1.2, 8, 20, 7)
        want = [ora.extract(f, (0, 1000)) for f in frames]
        assert_same(gpu(frames[0], None, (0, 1000)), want[0], f"{cols}x{rows} single qt_points {qt_points}")
        for i, r in enumerate(gpu.extract_batch(np.stack([frames[0], frames[1], frames[1], frames[0], frames[1]]), (0, 1000))):
            assert_same(r, want[(0, 1, 1, 0, 1)[i]], f"{cols}x{rows} batch frame {i} qt_points {qt_points}")


def test_strided_input_and_roi():
    big = synth.make_stream(1, 600, 800)[0]
    roi = big[60:540, 80:720]            # non-contiguous rows, like a cv::Mat ROI with step > cols
    assert not roi.flags["C_CONTIGUOUS"]
    gpu = ORBextractor(1000, 1.2, 8, 20, 7)
    assert_same(gpu(roi, None, (0, 1000)), po.OracleExtractor(1000, 1.2, 8, 20, 7).extract(np.ascontiguousarray(roi), (0, 1000)))


def test_empty_and_invalid_inputs():
    gpu = ORBextractor(1000, 1.2, 8, 20, 7)
    mono, kps, desc = gpu(np.zeros((0, 0), np.uint8))
    assert mono == -1 and len(kps) == 0 and desc.shape == (0, 32)        # src/ORBextractor.cc:1090-1091
    with pytest.raises(OrbxError):
        gpu(np.zeros((60, 60), np.uint8))                                  # too small for 8 levels: rejected loudly
    with pytest.raises(OrbxError):
        ORBextractor(0, 1.2, 8, 20, 7)


def test_batch_equals_single_and_is_order_independent():
    frames = synth.make_stream(6)
    gpu = ORBextractor(1000, 1.2, 8, 20, 7)
    ora = po.OracleExtractor(1000, 1.2, 8, 20, 7)
    res = gpu.extract_batch(frames, (0, 1000))
    for t in range(6):
        assert_same(res[t], ora.extract(frames[t], (0, 1000)), f"batch f{t}")
    perm = np.array([4, 2, 5, 0, 3, 1])
    res2 = gpu.extract_batch(frames[perm], (0, 1000))
    for i, t in enumerate(perm):
        assert res2[i][0] == res[t][0] and res2[i][1].tobytes() == res[t][1].tobytes() and np.array_equal(res2[i][2], res[t][2])
    again = gpu.extract_batch(frames, (0, 1000))       # idempotent: persistent buffers carry no state across calls
    for t in range(6):
        assert again[t][1].tobytes() == res[t][1].tobytes() and np.array_equal(again[t][2], res[t][2])


@pytest.mark.parametrize("rows,cols,nf,lap", [(480, 640, 1000, (0, 1000)), (1024, 1024, 2000, (0, 1000)), (376, 1241, 2000, (0, 0)),
                                              (300, 400, 700, (0, 1000)), (480, 752, 1200, (200, 400))])
def test_gaussian_inside_the_descriptor_kernel(rows, cols, nf, lap):
    """k_describe_blur ("desc_fused_blur"): batches under the default blur arithmetic blur the 43 x 48 raw window of every keypoint inside the
    descriptor kernel instead of launching k_blur7 — the same bytes (blur_tile's arithmetic), so the same descriptors: forced on for every
    shape (small levels put many keypoints into the column bands that are staged byte by byte with reflect-101), against the oracle."""
    frames = synth.make_stream(3, rows, cols, 5)
    ora = po.OracleExtractor(nf, 1.2, 8, 20, 7)
    want = [ora.extract(f, lap) for f in frames]
    for mode in (1, 0):
        gpu = ORBextractor(nf, 1.2, 8, 20, 7)
        gpu.set_option("desc_fused_blur", mode)
        res = gpu.extract_batch(frames, lap)
        for t in range(3):
            assert_same(res[t], want[t], f"fused={mode} {cols}x{rows} f{t}")


def test_the_fused_gaussian_is_chosen_where_it_pays_and_never_for_another_arithmetic():
    """Default (-1): by pyramid pixels per keypoint slot (its cost goes with the keypoints, k_blur7's with the pixels) — 1024 x 1024 with 2000 features
    and 640 x 480 with 1000 take the fused kernel (no k_blur7 launch), 640 x 480 with 2500 features the separate one; a CPU-path profile with
    another blur arithmetic always keeps k_blur7 (and its parity)."""
    def blur_launches(gpu, frames):
        gpu.extract_batch(frames, (0, 1000))
        gpu.profile_enable(True)
        gpu.extract_batch(frames, (0, 1000))
        pr = gpu.profile_read()
        gpu.profile_enable(False)
        return [n for k, (ms, n) in pr.items() if k.startswith("k_blur7")][0]
    big = synth.make_stream(2, 1024, 1024, 3)
    small = synth.make_stream(2, 480, 640, 3)
    assert blur_launches(ORBextractor(2000, 1.2, 8, 20, 7), big) == 0
    assert blur_launches(ORBextractor(1000, 1.2, 8, 20, 7), small) == 0
    assert blur_launches(ORBextractor(2500, 1.2, 8, 20, 7), small) > 0
    gpu = ORBextractor(2000, 1.2, 8, 20, 7)
    gpu.set_cpu_profile("opencv-4.4")
    gpu.set_option("desc_fused_blur", 1)
    assert blur_launches(gpu, big) > 0
    res = gpu.extract_batch(big, (0, 1000))
    with po.opencv_variant(*ORBextractor.cpu_profiles()["opencv-4.4"][1]):
        assert_same(res[0], po.OracleExtractor(2000, 1.2, 8, 20, 7).extract(big[0], (0, 1000)), "opencv-4.4, separate blur")
    # the FMA builds of the reference take the fused kernel too (v_pk_fma_f32 taps, the contracted atan)
    gpu = ORBextractor(2000, 1.2, 8, 20, 7)
    gpu.set_cpu_profile("opencv>=4.5.1", 3)
    assert blur_launches(gpu, big) == 0
    res = gpu.extract_batch(big, (0, 1000))
    v = np.zeros(5, np.int32)
    from orb_slam3_modified_amd import _lib
    assert _lib.lib().orbx_cpu_profile_values(b"opencv>=4.5.1", 3, _lib.ptr(v)) == 0
    with po.opencv_variant(*(int(x) for x in v)):
        assert_same(res[1], po.OracleExtractor(2000, 1.2, 8, 20, 7).extract(big[1], (0, 1000)), "default blur, FMA build, fused")


def test_shape_change_and_two_instances():
    a, b = ORBextractor(1000, 1.2, 8, 20, 7), ORBextractor(1000, 1.2, 8, 20, 7)   # stereo: two instances live together
    ora = po.OracleExtractor(1000, 1.2, 8, 20, 7)
    f1, f2 = synth.make_stream(1, 480, 640)[0], synth.make_stream(1, 350, 600, synth.DEFAULT_SEED + 9)[0]
    for img in (f1, f2, f1):
        assert_same(a(img, None, (0, 0)), ora.extract(img, (0, 0)))
        assert_same(b(img, None, (0, 0)), ora.extract(img, (0, 0)))


def test_full_size_batch_properties():
    """BASELINE full size (256 frames/step): size-independent properties + sampled bit-exact frames."""
    B = 256
    base = synth.make_stream(32)
    frames = base[np.arange(B) % 32]
    gpu = ORBextractor(1000, 1.2, 8, 20, 7)
    res = gpu.extract_batch(frames, (0, 1000))
    ora = po.OracleExtractor(1000, 1.2, 8, 20, 7)
    q = gpu.features_per_level()
    for t in range(B):
        mono, kps, desc = res[t]
        assert 990 <= len(kps) <= gpu.capacity and mono == 0
        assert (kps["octave"][::-1][np.argsort(kps["octave"][::-1], kind="stable")] == np.sort(kps["octave"])).all()
        cnt = np.bincount(kps["octave"], minlength=8)
        assert (cnt <= q + 2).all()                                   # at most N+2 per level
        assert (kps["angle"] >= 0).all() and (kps["angle"] < 360.001).all() and (kps["class_id"] == -1).all()
        assert (kps["response"] >= 7).all() and (kps["response"] == np.floor(kps["response"])).all()
        if t >= 32:                                                   # replayed frame -> identical result
            assert kps.tobytes() == res[t - 32][1].tobytes() and np.array_equal(desc, res[t - 32][2])
    for t in (0, 7, 19, 31):
        assert_same(res[t], ora.extract(frames[t], (0, 1000)), f"full-size f{t}")


def test_device_float_paths_wide_sweep():
    """fastAtan2 and the glibc-exact cosf/sinf on far more arguments than frames produce."""
    gpu = ORBextractor(100, 1.2, 1, 20, 7)
    rng = np.random.default_rng(9)
    m01 = rng.integers(-2_000_000, 2_000_000, 400_000).astype(np.float32)
    m10 = rng.integers(-2_000_000, 2_000_000, 400_000).astype(np.float32)
    m01[:1000] = 0; m10[1000:2000] = 0; m01[2000:2100] = m10[2000:2100]
    ang, a, b = gpu.debug_trig(m01, m10)
    L = po.lib()
    import ctypes as C
    for i in list(range(0, 4000)) + list(range(4000, 400_000, 37)):
        ea = po.fast_atan2(float(m01[i]), float(m10[i]))
        assert np.float32(ea).tobytes() == ang[i].tobytes(), (i, m01[i], m10[i], ea, ang[i])
        ca, sb = po.cos_sin_deg(float(ang[i]))
        assert np.float32(ca).tobytes() == a[i].tobytes() and np.float32(sb).tobytes() == b[i].tobytes(), (i, ang[i])
    # dense sweep of angle values straight into cos/sin
    deg = np.linspace(0, 360, 2_000_001, dtype=np.float64).astype(np.float32)
    _, a, b = gpu.debug_trig(deg)
    for i in range(0, len(deg), 997):
        ca, sb = po.cos_sin_deg(float(deg[i]))
        assert np.float32(ca).tobytes() == a[i].tobytes() and np.float32(sb).tobytes() == b[i].tobytes(), deg[i]


def test_device_batch_with_unaligned_strides_and_base():
    """orbx_extract_batch_device on frames whose row stride / base address are not multiples of 4: every kernel's
    byte-staging path (resize, FAST cells, blur, descriptor patches) instead of the aligned dword path."""
    import torch
    from orb_slam3_modified_amd.replay import BlockLayout, unpack_block
    dev = torch.device("cuda", 0)
    frames = synth.make_stream(3, 240, 322)                      # 322-px rows inside a 327-byte pitch, base offset 1
    B, H, W = frames.shape
    pitch = 327
    buf = torch.zeros(B * H * pitch + 8, dtype=torch.uint8, device=dev)
    view = buf[1:1 + B * H * pitch].view(B, H, pitch)
    view[:, :, :W] = torch.from_numpy(frames).to(dev)
    gpu = ORBextractor(600, 1.2, 5, 20, 7)
    lo = BlockLayout(B, gpu.capacity)
    blk = torch.zeros(lo.nbytes, dtype=torch.uint8, device=dev)
    base = blk.data_ptr()
    assert view.data_ptr() % 4 == 1 and pitch % 4 != 0
    gpu.extract_batch_device(view.data_ptr(), B, H, W, pitch, H * pitch, base, base + lo.desc_off, base + lo.counts_off, (0, 1000))
    torch.cuda.synchronize()
    res = unpack_block(blk.cpu().numpy(), lo)
    ora = po.OracleExtractor(600, 1.2, 5, 20, 7)
    for f in range(B):
        assert_same(res[f], ora.extract(frames[f], (0, 1000)), f"unaligned f{f}")


@pytest.mark.parametrize("seed", range(10))
def test_random_shapes_and_parameters(seed):
    """Randomised configurations: odd image sizes (all staging / tile edge cases), scale factors 1.1 .. 1.9, 1 .. 8 levels,
    feature budgets from starved to saturated, arbitrary lapping intervals — always bit-exact against the oracle."""
    rng = np.random.default_rng(1000 + seed)
    sf = float(np.float32(rng.choice([1.1, 1.15, 1.2, 1.25, 1.33, 1.5, 1.7, 1.9])))
    nlev = int(rng.integers(1, 9))
    min_side = int(np.ceil(70 * sf ** (nlev - 1))) + 2           # top level must keep >= 67 px
    lo = max(min_side, 90)
    rows = int(rng.integers(lo, max(700, lo + 200))); cols = int(rng.integers(lo, max(900, lo + 300)))
    if cols > 8 * rows or rows > 2 * cols:
        rows = cols = max(rows, cols) // 2 + min_side
    nf = int(rng.choice([30, 150, 700, 1000, 2500, 6000]))
    ini = int(rng.choice([12, 20, 35])); mn = int(rng.choice([3, 7, ini]))
    lap = tuple(sorted(rng.integers(0, cols + 50, 2).tolist()))
    img = synth.make_stream(1, rows, cols, 4242 + seed)[0]
    try:
        gpu = ORBextractor(nf, sf, nlev, ini, mn)
        res = gpu(img, None, lap)
    except OrbxError as e:
        pytest.skip(f"configuration rejected loudly ({e}); rows={rows} cols={cols} sf={sf} L={nlev}")
    assert_same(res, po.OracleExtractor(nf, sf, nlev, ini, mn).extract(img, lap), f"{cols}x{rows} sf{sf} L{nlev} nf{nf} th{ini}/{mn} lap{lap}")


def test_device_cos_sin_exhaustive_over_all_angles():
    """Every float angle in [0, 360] (1 136 000 000 bit patterns): the device's glibc-exact cosf/sinf of angle * pi/180
    against the host glibc, compared through an order-independent 64-bit digest computed on both sides."""
    gpu = ORBextractor(100, 1.2, 1, 20, 7)
    last = int(np.float32(360.0).view(np.uint32))
    assert last + 1 == 1135869953
    for first, count in ((0, 400_000_000), (400_000_000, 400_000_000), (800_000_000, last + 1 - 800_000_000)):
        assert gpu.debug_trig_hash(first, count) == po.trig_hash(first, count), (first, count)
    assert gpu.debug_trig_hash(5, 1000) != gpu.debug_trig_hash(6, 1000)        # the digest is input-sensitive


def test_device_fast_atan2_on_a_billion_moment_pairs():
    """cv::fastAtan2 on 1e9 pseudo-random (m01, m10) moment pairs: device digest == oracle digest."""
    gpu = ORBextractor(100, 1.2, 1, 20, 7)
    for seed in (1, 0x9E3779B9):
        assert gpu.debug_atan_hash(seed, 500_000_000) == po.atan_hash(seed, 500_000_000), seed
    assert gpu.debug_atan_hash(3, 1000) != gpu.debug_atan_hash(4, 1000)


def test_config4_tumvi_batch_all_frames():
    """BASELINE config 4 shape (1024x1024, 2000 features) as a device batch: every frame bit-exact, both output branches."""
    frames = synth.make_stream(12, 1024, 1024)
    gpu = ORBextractor(2000, 1.2, 8, 20, 7)
    res = gpu.extract_batch(frames, (0, 1000))
    ora = po.OracleExtractor(2000, 1.2, 8, 20, 7)
    for t in range(len(frames)):
        assert_same(res[t], ora.extract(frames[t], (0, 1000)), f"1024^2 f{t}")
        assert 0 < res[t][0] < len(res[t][1])


@pytest.mark.parametrize("channels,rgb,stride_pad", [(3, True, 0), (3, False, 5), (4, True, 0), (4, False, 12)])
def test_color_ingestion_equals_oracle(channels, rgb, stride_pad):
    """cvtColor fused behind the upload (src/Tracking.cc:1572-1585): grey plane and features equal the oracle's."""
    rng = np.random.default_rng(channels * 2 + rgb)
    base = synth.make_stream(1, 480, 640)[0].astype(np.int32)
    planes = [np.clip(base + rng.integers(-40, 41, base.shape) + s, 0, 255).astype(np.uint8) for s in (0, 17, -23)]
    if channels == 4:
        planes.append(rng.integers(0, 256, base.shape).astype(np.uint8))          # alpha: ignored
    wide = np.zeros((480, 640 * channels + stride_pad), np.uint8)
    img = np.lib.stride_tricks.as_strided(wide, (480, 640, channels), (wide.strides[0], channels, 1))
    img[...] = np.stack(planes, -1)
    gray = po.cvt_color_to_gray(img, rgb)
    assert abs(gray.astype(int) - planes[1].astype(int)).max() < 60 and gray.std() > 10
    gpu = ORBextractor(1000, 1.2, 8, 20, 7)
    mono, kps, desc = gpu.extract_color(img, rgb, (0, 1000))
    assert np.array_equal(gpu.pyramid_level(0), gray)
    okps, odesc, omono = po.OracleExtractor(1000, 1.2, 8, 20, 7).extract(gray, (0, 1000))
    assert mono == omono and kps.tobytes() == okps.tobytes() and np.array_equal(desc, odesc)
    # the grey path still works on the same context afterwards (graph path + staging buffers untouched)
    mono2, kps2, desc2 = gpu(gray, None, (0, 1000))
    assert mono2 == omono and kps2.tobytes() == okps.tobytes() and np.array_equal(desc2, odesc)


def test_color_ingestion_formula_extremes():
    assert po.cvt_color_to_gray(np.full((1, 1, 3), 255, np.uint8), True)[0, 0] == 255
    assert po.cvt_color_to_gray(np.zeros((1, 1, 3), np.uint8), True)[0, 0] == 0
    px = np.array([[[255, 0, 0]]], np.uint8)
    assert po.cvt_color_to_gray(px, True)[0, 0] == 76 and po.cvt_color_to_gray(px, False)[0, 0] == 29   # 0.299 / 0.114


def test_two_extractors_from_two_threads_like_stereo():
    """src/Frame.cc:122-125: the left and right extractor run concurrently from two std::threads.  Two contexts, two
    threads (ctypes drops the GIL during the calls), different image shapes so both capture / re-capture their graphs while
    the other one is running; every result must equal the single-threaded one."""
    import threading
    imgs_l = synth.make_stream(6, 480, 752, synth.DEFAULT_SEED + 5)
    imgs_r = synth.make_stream(6, 480, 640, synth.DEFAULT_SEED + 6)
    exl, exr = ORBextractor(1200, 1.2, 8, 20, 7), ORBextractor(1000, 1.2, 8, 20, 7)
    want_l = [exl(f, None, (0, 0)) for f in imgs_l]
    want_r = [exr(f, None, (0, 0)) for f in imgs_r]
    errors = []

    def run(ex, imgs, want, tag):
        try:
            for rep in range(15):
                for i, f in enumerate(imgs):
                    view = f[: 480 - 8 * (rep % 2)]                     # alternate shapes: forces graph re-capture
                    mono, k, d = ex(view, None, (0, 0))
                    if rep % 2 == 0:
                        wm, wk, wd = want[i]
                        if mono != wm or k.tobytes() != wk.tobytes() or not np.array_equal(d, wd):
                            errors.append((tag, rep, i))
        except Exception as e:   # noqa: BLE001
            errors.append((tag, repr(e)))

    ts = [threading.Thread(target=run, args=(exl, imgs_l, want_l, "L")), threading.Thread(target=run, args=(exr, imgs_r, want_r, "R"))]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errors, errors[:5]


@pytest.mark.parametrize("nf,sf,nlev,rows,cols,kind", [(10000, 1.2, 8, 376, 1241, "noise"),      # KITTI: mpIniORBextractor = 5 x 2000
                                                       (6000, 1.33, 3, 327, 454, "synth"),
                                                       (6000, 1.5, 2, 480, 640, "noise"),
                                                       (20000, 1.2, 1, 800, 600, "noise"),       # the fork's own Examples/Monocular/mi.yaml
                                                       (20000, 1.2, 1, 800, 600, "synth"),
                                                       (100000, 1.2, 1, 800, 600, "noise")])     # ... and its 5x initialisation extractor
def test_level_quotas_beyond_the_lds_are_served_from_hbm(nf, sf, nlev, rows, cols, kind):
    """Quadtree node arrays of a level live in one CU's LDS; a quota that does not fit (above ~2100) falls back to HBM node
    arrays — slower, same results (src/ORBextractor.cc:555-779 has no such limit)."""
    rng = np.random.default_rng(nf)
    img = rng.integers(0, 256, (rows, cols)).astype(np.uint8) if kind == "noise" else synth.make_stream(1, rows, cols, 99)[0]
    gpu = ORBextractor(nf, sf, nlev, 20, 7)
    ora = po.OracleExtractor(nf, sf, nlev, 20, 7)
    assert ora.tables()["quota"].max() > 2150
    mono, kps, desc = gpu(img, None, (0, 0))
    okps, odesc, omono = ora.extract(img, (0, 0))
    assert mono == omono and kps.tobytes() == okps.tobytes() and np.array_equal(desc, odesc)
    assert len(kps) > (2500 if kind == "noise" else 500)
    if nf >= 20000:
        return
    # and as a device batch (sub-batch offsets of the HBM node scratch)
    res = gpu.extract_batch(np.stack([img, img[::-1].copy(), img]), (0, 0))
    assert res[0][1].tobytes() == okps.tobytes() and res[2][1].tobytes() == okps.tobytes()
    o2 = ora.extract(img[::-1].copy(), (0, 0))
    assert res[1][1].tobytes() == o2[0].tobytes() and np.array_equal(res[1][2], o2[1])


def test_single_frame_graph_survives_buffer_reallocation():
    """The single-frame path replays a captured graph; a batch call in between reallocates the staging buffers the graph
    points to, a different shape re-sizes the pyramid: the next single-frame calls must re-capture and stay correct."""
    a = synth.make_stream(2, 480, 640, 11)
    b = synth.make_stream(1, 376, 1241, 12)[0]
    gpu = ORBextractor(1000, 1.2, 8, 20, 7)
    ora = po.OracleExtractor(1000, 1.2, 8, 20, 7)
    wa0, wa1, wb = ora.extract(a[0], (0, 0)), ora.extract(a[1], (0, 0)), ora.extract(b, (0, 0))
    for _ in range(2):
        assert_same(gpu(a[0], None, (0, 0)), wa0, "single a0")
        res = gpu.extract_batch(np.stack([a[1], a[0], a[1], a[0], a[1]]), (0, 0))        # grows the staging block
        assert res[0][1].tobytes() == wa1[0].tobytes() and res[3][1].tobytes() == wa0[0].tobytes()
        assert_same(gpu(a[1], None, (0, 0)), wa1, "single a1 after batch")
        assert_same(gpu(b, None, (0, 0)), wb, "other shape")
        assert_same(gpu(a[0], None, (0, 1000)), ora.extract(a[0], (0, 1000)), "other lapping area")
        gpu.set_option("graph", 0)
        assert_same(gpu(a[0], None, (0, 0)), wa0, "graph off")
        gpu.set_option("graph", 1)


@pytest.mark.parametrize("dist", [(-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05),           # EuRoC cam0 (Examples/*/EuRoC.yaml)
                                  (-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05, 0.0123),   # with k3
                                  (0.0, 0.0, 0.0, 0.0), (0.9, -3.0, 0.01, -0.02)])                # none; strong (icdist may turn negative far out)
def test_undistort_keypoints_equals_oracle(dist):
    """Frame::UndistortKeyPoints (src/Frame.cc:747-780) on the GPU — from host arrays and on the resident outputs of a batch
    extraction — against the oracle's restatement of cv::undistortPoints: raw float bits."""
    import torch
    fx, fy, cx, cy = 458.654, 457.296, 367.215, 248.375
    frames = synth.make_stream(3, 480, 752)
    gpu = ORBextractor(1200, 1.2, 8, 20, 7)
    mono, kps, desc = gpu(frames[0], None, (0, 0))
    un = gpu.UndistortKeyPoints(kps, fx, fy, cx, cy, dist)
    want = po.undistort_keypoints(kps, fx, fy, cx, cy, dist)
    assert un.tobytes() == want.tobytes()
    if dist[0] != 0:
        # (with the strong coefficients icdist turns negative away from the centre: those points come back unchanged)
        assert (un["x"] != kps["x"]).mean() > (0.9 if abs(dist[0]) < 0.5 else 0.3)
        assert np.array_equal(un["angle"], kps["angle"]) and np.array_equal(un["octave"], kps["octave"])
    else:
        assert un.tobytes() == kps.tobytes()
    assert len(gpu.UndistortKeyPoints(kps[:0], fx, fy, cx, cy, dist)) == 0
    # device-resident: batch extraction, then undistortion of the resident keypoints; nothing but the result is read back
    dev = torch.device("cuda", 0)
    fr = torch.from_numpy(frames).to(dev)
    cap = gpu.capacity
    d_k = torch.zeros(3 * cap * 28, dtype=torch.uint8, device=dev); d_d = torch.zeros(3 * cap * 32, dtype=torch.uint8, device=dev)
    d_c = torch.zeros(6, dtype=torch.int32, device=dev); d_u = torch.zeros(3 * cap * 28, dtype=torch.uint8, device=dev)
    st = torch.cuda.Stream(device=dev)
    gpu.extract_batch_device(fr.data_ptr(), 3, 480, 752, fr.stride(1), fr.stride(0), d_k.data_ptr(), d_d.data_ptr(), d_c.data_ptr(), (0, 0), st.cuda_stream)
    gpu.undistort_keypoints_device(d_k.data_ptr(), d_c.data_ptr(), 3, fx, fy, cx, cy, dist, d_u.data_ptr(), st.cuda_stream)
    st.synchronize()
    from orb_slam3_modified_amd import KP_DTYPE
    cnt = d_c.cpu().numpy().reshape(3, 2)
    hk = d_k.cpu().numpy().view(KP_DTYPE).reshape(3, cap); hu = d_u.cpu().numpy().view(KP_DTYPE).reshape(3, cap)
    for f in range(3):
        n = cnt[f, 0]
        assert n > 900 and hu[f, :n].tobytes() == po.undistort_keypoints(hk[f, :n], fx, fy, cx, cy, dist).tobytes()


@pytest.mark.parametrize("src,dst", [((480, 752), (350, 600)),      # EuRoC.yaml: Camera.newHeight / newWidth
                                     ((960, 1280), (480, 640)),     # exact 2 x 2: OpenCV's INTER_AREA shortcut
                                     ((480, 640), (600, 800)),      # upscale
                                     ((376, 1241), (300, 990)),     # KITTI-like, ratio 1.2535
                                     ((480, 640), (480, 640))])     # same size: plain operator()
def test_resize_ingestion_equals_oracle(src, dst):
    """cv::resize(im, resizedIm, newImSize) of System::TrackMonocular (src/System.cc:441-446) fused behind the upload."""
    img = synth.make_stream(1, src[0], src[1])[0]
    gpu = ORBextractor(1000, 1.2, 8, 20, 7)
    res = gpu.extract_resized(img, dst, (0, 1000))
    small = po.cv_resize(img, dst[1], dst[0]) if src != dst else img
    assert_same(res, po.OracleExtractor(1000, 1.2, 8, 20, 7).extract(small, (0, 1000)), f"{src}->{dst}")
    assert np.array_equal(gpu.pyramid_level(0), small)
    res2 = gpu.extract_resized(img[:, :src[1] - 8][:, 3:], (dst[0], dst[1] - 16), (0, 1000))   # strided, unaligned source view
    sub = np.ascontiguousarray(img[:, :src[1] - 8][:, 3:])
    small2 = po.cv_resize(sub, dst[1] - 16, dst[0])
    assert_same(res2, po.OracleExtractor(1000, 1.2, 8, 20, 7).extract(small2, (0, 1000)), "strided source")


def test_device_std_sort_restatement_equals_the_host_permutation():
    """The quadtree kernel's workgroup-parallel std::sort (level-synchronous introsort for long segments, one wave in registers for
    segments of <= 65 elements, stable rank inside the final <= 16-element segments) against csrc/gnu_sort.h run on the host —
    which tests/test_support_models.py checks against the host's own std::sort: identical permutation, ties included, on the
    adversarial shapes of that test (few distinct keys, runs, organ pipes, median-of-3 killers that reach the heapsort fallback)."""
    import ctypes as C
    from tests.test_support_models import _model
    L = _model()
    ex = ORBextractor(500, 1.2, 4, 20, 7)
    rng = np.random.default_rng(5)

    def killer(n):
        v = np.zeros(n, np.int64)
        k = n // 2
        for i in range(1, k + 1):
            if i % 2:
                v[i - 1] = i
                if i < n: v[i] = k + i
            if k + i - 1 < n: v[k + i - 1] = 2 * i
        return v

    cases = 0
    for rep in range(260):
        n = int(rng.integers(0, 90)) if rep < 120 else int(rng.integers(0, 400)) if rep < 220 else int(rng.integers(400, 2049))
        kc, kx = int(rng.integers(1, 7)), int(rng.integers(1, 9))
        mode = rep % 6
        i = np.arange(n)
        c = 2 + rng.integers(0, kc, n)
        x = rng.integers(0, kx, n) * 37
        if mode == 1: c = 2 + i * kc // max(n, 1)
        if mode == 2: c = 2 + (n - i) * kc // max(n, 1)
        if mode == 3: c = 2 + np.minimum(i, n - i) % (kc + 1)
        if mode == 4: c, x = 2 + killer(n), np.zeros(n, np.int64)
        key = (c.astype(np.uint64) << np.uint64(13)) | x.astype(np.uint64)
        v = (key << np.uint64(32)) | i.astype(np.uint64)
        want = np.ascontiguousarray(v).copy()
        L.qtm_sort(want.ctypes.data_as(C.c_void_p), C.c_int(n))
        for threads in ((64, 256, 512) if rep % 3 == 0 else (256,)):
            got = ex.debug_gnu_sort(v, threads)
            assert np.array_equal(got, want), (rep, n, mode, threads, int(np.flatnonzero(got != want)[0]))
        cases += 1
    assert cases == 260


@pytest.mark.parametrize("realign", [1, 0])
@pytest.mark.parametrize("rows,cols", [(376, 1241), (370, 1226), (240, 643)])
def test_batches_of_frames_with_odd_row_strides(rows, cols, realign):
    """KITTI's 1241- and 1226-px rows (Examples/*/KITTI00-02.yaml, KITTI04-12.yaml) are not dword-aligned: a batch of such frames is copied
    into an aligned buffer first ("realign", k_realign_rows) — or, with the option off, takes the kernels' byte-wise staging paths.  Same
    keypoints, descriptors and pyramid either way, also for a batch that is a strided view (odd frame stride) of a larger block."""
    nf = 1500
    frames = synth.make_stream(4, rows, cols, 31 + cols)
    ora = po.OracleExtractor(nf, 1.2, 8, 20, 7)
    want = [ora.extract(f, (0, cols)) for f in frames]
    gpu = ORBextractor(nf, 1.2, 8, 20, 7)
    gpu.set_option("realign", realign)
    for rep in range(2):
        for i, r in enumerate(gpu.extract_batch(frames, (0, cols))):
            assert_same(r, want[i], f"{cols}x{rows} realign {realign} rep {rep} frame {i}")
    ora.extract(frames[2], (0, cols))
    for l in (0, 1, 3):
        assert np.array_equal(gpu.pyramid_level(l, frame=2), ora.level(l)), (cols, rows, realign, l)


@pytest.mark.parametrize("opts", [dict(fast_stage_dma=1), dict(fast_stage_dma=0), dict(fast_stage_dma=1, fast_pk=0), dict(fast_stage_dma=0, fast_threads=64)])
def test_fast_tile_staging_variants(opts):
    """How a FAST cell's tile reaches LDS: LDS-DMA loads (global_load_lds, the default on aligned sources: four whole rows per wave-instruction
    on the 64-byte tile pitch, a flat dword stream on the 96-byte pitch of wide cells) or plain loads + ds_write (odd strides) — same keypoints
    and descriptors on batches and single frames, on an image whose row stride is odd, and on a geometry whose cells need the 96-byte pitch.
    (The looped kernel with the next tile in flight, other pitches and the early-out were measured and removed: HISTORY.md.)"""
    ora = po.OracleExtractor(1000, 1.2, 8, 20, 7)
    frames = synth.make_stream(7)
    gpu = ORBextractor(1000, 1.2, 8, 20, 7)
    for k, v in opts.items():
        gpu.set_option(k, v)
    want = [ora.extract(f, (0, 1000)) for f in frames]
    for i, r in enumerate(gpu.extract_batch(frames, (0, 1000))):
        assert_same(r, want[i], f"{opts} batch frame {i}")
    assert_same(gpu(frames[3], None, (0, 1000)), want[3], f"{opts} single")
    # odd row stride: a view into a wider buffer (the aligned-dword staging does not apply)
    wide = np.zeros((480, 643), np.uint8)
    wide[:, 1:641] = frames[2]
    assert_same(gpu(wide[:, 1:641], None, (0, 1000)), want[2], f"{opts} odd stride")
    # a 95-px-wide top level: one column of cells, 75 px wide -> the 96-byte tile pitch for the whole launch
    ora2 = po.OracleExtractor(200, 1.2, 2, 20, 7)
    gpu2 = ORBextractor(200, 1.2, 2, 20, 7)
    for k, v in opts.items():
        gpu2.set_option(k, v)
    big = synth.make_stream(3, 100, 114, 5)
    w2 = [ora2.extract(f, (0, 1000)) for f in big]
    for i, r in enumerate(gpu2.extract_batch(np.stack([big[0], big[1], big[2], big[1], big[0]]), (0, 1000))):
        assert_same(r, w2[(0, 1, 2, 1, 0)[i]], f"{opts} wide cells frame {i}")


@pytest.mark.parametrize("shape,nlevels,sf", [((480, 640), 8, 1.2), ((480, 752), 8, 1.2), ((350, 600), 8, 1.2), ((400, 500), 3, 1.2), ((480, 640), 2, 1.2),
                                              ((600, 800), 5, 1.5), ((1024, 1024), 8, 1.2)])
@pytest.mark.parametrize("opts", [dict(small_fused=1, graph=1), dict(small_fused=1, graph=0), dict(small_fused=0, graph=1), dict(small_fused=0, graph=0),
                                  dict(small_fused=1, graph=1, desc_k=4)])
def test_single_frame_launch_shapes_give_the_same_bytes(shape, nlevels, sf, opts):
    """The latency-bound call (one to four frames) has its own launch plan — the upload as a kernel, groups of pyramid levels per launch
    (k_resize_chain, with the intermediate levels computed redundantly per tile and written on the way), FAST + blur in one launch, the
    assembly as the quadtree's tail, results mirrored into the pinned block, one keypoint per wave — inside a replayed graph or as plain
    launches.  Every combination must return the oracle's bytes, and every pyramid level the same bytes (stereo reads mvImagePyramid):
    level counts that leave a group of one, two or three, widths that are not multiples of 16 (the copy-node upload), other scales."""
    rows, cols = shape
    nf = 2000 if rows * cols > 500000 else 1000
    img = synth.make_stream(2, rows, cols, 1234)[1]
    ora = po.OracleExtractor(nf, sf, nlevels, 20, 7)
    want = ora.extract(img, (0, 1000))
    gpu = ORBextractor(nf, sf, nlevels, 20, 7)
    for k, v in opts.items():
        gpu.set_option(k, v)
    for rep in range(2):          # the second call replays the captured graph
        assert_same(gpu(img, None, (0, 1000)), want, f"{opts} rep {rep}")
        for l in range(nlevels):
            assert np.array_equal(gpu.pyramid_level(l), ora.level(l)), (opts, rep, l)
    # two and four frames per call take the same plan with a second grid dimension; five leave it
    for nb in (2, 4, 5):
        batch = np.stack([img] + [synth.make_stream(1, rows, cols, 77 + i)[0] for i in range(nb - 1)])
        res = gpu.extract_batch(batch, (0, 1000))
        for f in range(nb):
            assert_same(res[f], ora.extract(batch[f], (0, 1000)), f"{opts} batch {nb} frame {f}")


@pytest.mark.parametrize("rows,cols,nf,sf,nlevels,lap", [
    (480, 640, 1000, 1.1, 12, (0, 1000)), (480, 752, 1500, 1.1, 10, (0, 0)), (512, 1024, 2000, 1.08, 16, (0, 1000)), (480, 640, 800, 1.15, 9, (100, 300)),
    (376, 1241, 2000, 1.1, 11, (0, 2000)), (480, 640, 1000, 1.2, 3, (0, 1000)), (480, 640, 1000, 1.3, 5, (0, 1000)), (480, 640, 500, 1.2, 2, (0, 1000))])
@pytest.mark.parametrize("opts", [dict(), dict(chain_long=0), dict(describe_direct=0), dict(chain_first=3, chain_long_tile=32)])
def test_single_frame_level_counts_and_lapping_shortcuts(rows, cols, nf, sf, nlevels, lap, opts):
    """The single-frame plan of round 3: the whole pyramid as one chain launch of up to seven levels (more levels: a second long launch; the
    16-level maximum: three), on 16-px tiles, and no assembly pass when the lapping area holds every keypoint or none ((0, 1000) on a
    narrower image, (0, 0), (0, 2000) on a 1241-wide one) — against the general path ((100, 300); (0, 1000) on a 1024-wide image) and
    against the options that switch each shortcut off.  Keypoints, descriptors, return value and every pyramid level must be the oracle's."""
    img = synth.make_stream(1, rows, cols, 11 + nlevels)[0]
    ora = po.OracleExtractor(nf, sf, nlevels, 20, 7)
    want = ora.extract(img, lap)
    gpu = ORBextractor(nf, sf, nlevels, 20, 7)
    for k, v in opts.items():
        gpu.set_option(k, v)
    for rep in range(2):
        assert_same(gpu(img, None, lap), want, f"{opts} rep {rep}")
        for l in range(1, nlevels):
            assert np.array_equal(gpu.mvImagePyramid[l], ora.level(l)), (opts, rep, l)
