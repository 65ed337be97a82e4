"""The reference's UNMODIFIED src/Frame.cc — the only caller of the drop-in boundary ("must stay untouched") — compiled
where it lies over this repository's include/ORBextractor.h and include/ORBVocabulary.h, against the same file compiled over the
reference's own src/ORBextractor.cc and DBoW2.  tests/support/frame_world.cpp runs all four Frame constructors (rectified stereo with
ComputeStereoMatches on the extractors' mvImagePyramid and both extractions on two threads; RGB-D with UndistortKeyPoints and
ComputeStereoFromRGBD; monocular incl. the 1024-wide case and a 5000-feature extractor; two fisheye cameras with lapping areas, the
kNN-2 matcher and both grids), then ComputeBoW, GetFeaturesInArea, isInFrustum / ProjectPointDistort and the copy constructor, and
prints everything the Frame holds as bit patterns.

  golden   tests/golden/frame_world_ref.txt.gz = output of oracle/_ref/ref_frame_world (reference Frame.cc + reference extractor + DBoW2)
  CPU      oracle/_ref/dropin_frame_world_cpu: the same Frame.cc over the drop-in headers and the oracle-backed stub of the C-ABI
  GPU      oracle/_ref/dropin_frame_world:     the same Frame.cc over the drop-in headers and liborbx.so — the shipped path

  opt-in   oracle/_ref/dropin_frame_world_stereo[_cpu]: src/Frame.cc with integration/Frame_stereo.patch (four lines at the top of
           ComputeStereoMatches, applied to a temporary copy at build time) and -DORBX_DEVICE_STEREO: the association runs on the two extractors'
           device pyramids and no pyramid is mirrored to the host — held to the SAME golden, every constructor

The drop-in executables contain compiled reference code, so they are built by oracle/ref_fragments.mk (from `__graft_entry__.build()`)
in the container that has /root/reference and travel to the GPU box as built files; nothing here reads /root/reference at run time."""
import gzip
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFDIR = os.path.join(ROOT, "oracle", "_ref")
VOC = os.path.join(ROOT, "tests", "golden", "voc_k5_L3.txt")
GOLD = os.path.join(ROOT, "tests", "golden", "frame_world_ref.txt.gz")
HAVE_REF = os.path.isdir("/root/reference")


def _build(target):
    if HAVE_REF:
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "-f", "ref_fragments.mk", "_ref/" + target])
    exe = os.path.join(REFDIR, target)
    if not os.path.exists(exe):
        pytest.skip(f"oracle/_ref/{target} not built (it is compiled from /root/reference/src/Frame.cc by oracle/ref_fragments.mk)")
    return exe


def _run(exe, out):
    subprocess.run([exe, VOC, out], check=True, stdout=subprocess.DEVNULL, timeout=600)   # stdout: the reference's own "Negative depth" chatter
    return open(out).read()


def _first_difference(a, b):
    la, lb = a.splitlines(), b.splitlines()
    frame = "?"
    for i in range(max(len(la), len(lb))):
        x, y = (la[i] if i < len(la) else "<missing>"), (lb[i] if i < len(lb) else "<missing>")
        if i < len(la) and not la[i].startswith(" "):
            frame = la[i].split()[0]
        if x != y:
            j = next((k for k in range(min(len(x), len(y))) if x[k] != y[k]), min(len(x), len(y)))
            return f"first difference in frame {frame}, line {i + 1}, column {j}:\n  expected {x[max(0, j - 60):j + 80]}\n  got      {y[max(0, j - 60):j + 80]}"
    return "identical"


def _golden():
    return gzip.open(GOLD).read().decode()


def test_golden_covers_every_constructor():
    frames = [l.split()[0] for l in _golden().splitlines() if not l.startswith(" ")]
    assert frames == ["stereo_752x480", "stereo_752x480_next", "rgbd_640x480_distorted", "mono_752x480_distorted", "mono_1024x512_5000",
                      "mono_752x480_with_prev", "mono_constant_640x480", "mono_nearly_flat_640x480", "fisheye_pair_512"]
    txt = _golden()
    depth = [int(l.split("with_depth=")[1].split()[0]) for l in txt.splitlines() if "with_depth=" in l]
    assert depth[0] > 500 and depth[1] > 500 and depth[2] > 500 and depth[3] == 0 and depth[-1] > 50     # stereo, RGB-D and fisheye matches exist
    heads = {l.split()[0]: l for l in txt.splitlines() if not l.startswith(" ")}
    assert " N=0 " in heads["mono_constant_640x480"] and "keys=0" in heads["mono_constant_640x480"]       # the early return of the constructor
    assert 0 < int(heads["mono_nearly_flat_640x480"].split("N=")[1].split()[0]) < 200                      # far fewer keypoints than asked for
    both = [l for l in txt.splitlines() if l.startswith("mono_1024x512_5000")][0]
    mono_left = int(both.split("monoLeft=")[1].split()[0])
    assert mono_left == -1        # the monocular constructor resets it after the extraction (:363); the split shows in mvKeys' order instead
    fish = [l for l in txt.splitlines() if l.startswith("fisheye_pair_512")][0]
    assert int(fish.split("Nleft=")[1].split()[0]) > 500 and int(fish.split("monoLeft=")[1].split()[0]) > 0


def test_reference_build_reproduces_golden(tmp_path):
    out = _run(_build("ref_frame_world"), str(tmp_path / "ref.txt"))
    assert out == _golden(), _first_difference(_golden(), out)


def test_reference_frame_cc_over_the_dropin_headers_cpu(tmp_path):
    out = _run(_build("dropin_frame_world_cpu"), str(tmp_path / "cpu.txt"))
    assert out == _golden(), _first_difference(_golden(), out)


def test_patched_frame_cc_device_stereo_cpu(tmp_path):
    """integration/Frame_stereo.patch: the patched file's host logic (argument order, the `mb` it passes, the vectors it sizes) over the stub,
    whose orbx_stereo_matches is the oracle's restatement of src/Frame.cc:811-981 on the oracle's pyramids."""
    out = _run(_build("dropin_frame_world_stereo_cpu"), str(tmp_path / "cpu.txt"))
    assert out == _golden(), _first_difference(_golden(), out)


def test_stereo_patch_is_four_lines_and_applies_to_the_reference(tmp_path):
    patch = os.path.join(ROOT, "integration", "Frame_stereo.patch")
    body = open(patch).read().split("--- a/src/Frame.cc")[1].splitlines()
    added = [l for l in body if l.startswith("+") and not l.startswith("+++")]
    removed = [l for l in body if l.startswith("-") and not l.startswith("---")]
    assert len(added) == 4 and not removed and added[0].startswith("+#ifdef ORBX_DEVICE_STEREO") and added[-1] == "+#endif"
    if not HAVE_REF:
        pytest.skip("needs /root/reference")
    out = tmp_path / "Frame.cc"
    subprocess.check_call(["patch", "-s", "-o", str(out), "/root/reference/src/Frame.cc", patch])
    ref, got = open("/root/reference/src/Frame.cc").read().splitlines(), out.read_text().splitlines()
    assert len(got) == len(ref) + 4
    i = next(k for k, l in enumerate(got) if "ORBX_DEVICE_STEREO" in l)
    assert "void Frame::ComputeStereoMatches()" in got[i - 2] and got[:i] == ref[:i] and got[i + 4:] == ref[i:]


@pytest.mark.parametrize("variant", [8, 34])
def test_patched_frame_cc_other_scenes_cpu(tmp_path, variant):
    ref, cpu = _build("ref_frame_world"), _build("dropin_frame_world_stereo_cpu")
    env = dict(os.environ, FRAME_WORLD_VARIANT=str(variant))
    outs = []
    for exe, name in ((ref, "ref.txt"), (cpu, "cpu.txt")):
        out = str(tmp_path / name)
        subprocess.run([exe, VOC, out], check=True, stdout=subprocess.DEVNULL, timeout=600, env=env)
        outs.append(open(out).read())
    assert outs[0] == outs[1], _first_difference(outs[0], outs[1])


@pytest.mark.parametrize("variant", [3, 8, 21, 34, 55])
def test_other_scenes_sizes_and_feature_counts_cpu(tmp_path, variant):
    """FRAME_WORLD_VARIANT: other scenes, image sizes (640x480 ... 1280x720) and feature counts (1000 ... 2000) through the same driver —
    the reference build against the drop-in build over the oracle-backed stub (tools/fuzz_frame_world.py does the same on the GPU)."""
    ref, cpu = _build("ref_frame_world"), _build("dropin_frame_world_cpu")
    env = dict(os.environ, FRAME_WORLD_VARIANT=str(variant))
    outs = []
    for exe, name in ((ref, "ref.txt"), (cpu, "cpu.txt")):
        out = str(tmp_path / name)
        subprocess.run([exe, VOC, out], check=True, stdout=subprocess.DEVNULL, timeout=600, env=env)
        outs.append(open(out).read())
    assert outs[0] == outs[1], _first_difference(outs[0], outs[1])
    assert outs[0] != _golden()       # it IS another scene


@pytest.mark.gpu
def test_reference_frame_cc_over_the_dropin_headers_gpu(tmp_path):
    exe = os.path.join(REFDIR, "dropin_frame_world")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/dropin_frame_world not built (it is compiled from /root/reference/src/Frame.cc against liborbx.so by oracle/ref_fragments.mk)")
    out = _run(exe, str(tmp_path / "gpu.txt"))
    assert out == _golden(), _first_difference(_golden(), out)


@pytest.mark.gpu
@pytest.mark.parametrize("variant", [5, 13, 89])
def test_other_scenes_sizes_and_feature_counts_gpu(tmp_path, variant):
    """The same variants on the GPU: the reference build (CPU code, prebuilt) against the drop-in build over liborbx.so, both run here."""
    ref, gpu = os.path.join(REFDIR, "ref_frame_world"), os.path.join(REFDIR, "dropin_frame_world")
    if not (os.path.exists(ref) and os.path.exists(gpu)):
        pytest.skip("oracle/_ref/ref_frame_world / dropin_frame_world not built (oracle/ref_fragments.mk compiles them from /root/reference)")
    env = dict(os.environ, FRAME_WORLD_VARIANT=str(variant))
    outs = []
    for exe, name in ((ref, "ref.txt"), (gpu, "gpu.txt")):
        out = str(tmp_path / name)
        subprocess.run([exe, VOC, out], check=True, stdout=subprocess.DEVNULL, timeout=600, env=env)
        outs.append(open(out).read())
    assert outs[0] == outs[1], _first_difference(outs[0], outs[1])


@pytest.mark.gpu
def test_patched_frame_cc_device_stereo_gpu(tmp_path):
    """The opt-in as shipped: the reference's src/Frame.cc + integration/Frame_stereo.patch over liborbx.so, both extractors without a host
    pyramid (-DORBX_DEVICE_STEREO) — every constructor's Frame (stereo, RGB-D, monocular, fisheye pair) byte-identical to the reference's."""
    exe = os.path.join(REFDIR, "dropin_frame_world_stereo")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/dropin_frame_world_stereo not built (oracle/ref_fragments.mk compiles it from /root/reference/src/Frame.cc + the patch)")
    out = _run(exe, str(tmp_path / "gpu.txt"))
    assert out == _golden(), _first_difference(_golden(), out)


@pytest.mark.gpu
@pytest.mark.parametrize("variant", [5, 13, 89, 144])
def test_patched_frame_cc_other_scenes_gpu(tmp_path, variant):
    ref, gpu = os.path.join(REFDIR, "ref_frame_world"), os.path.join(REFDIR, "dropin_frame_world_stereo")
    if not (os.path.exists(ref) and os.path.exists(gpu)):
        pytest.skip("oracle/_ref/ref_frame_world / dropin_frame_world_stereo not built")
    env = dict(os.environ, FRAME_WORLD_VARIANT=str(variant))
    outs = []
    for exe, name in ((ref, "ref.txt"), (gpu, "gpu.txt")):
        out = str(tmp_path / name)
        subprocess.run([exe, VOC, out], check=True, stdout=subprocess.DEVNULL, timeout=600, env=env)
        outs.append(open(out).read())
    assert outs[0] == outs[1], _first_difference(outs[0], outs[1])
