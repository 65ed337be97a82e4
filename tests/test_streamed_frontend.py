"""tools/streamed_frontend.cpp — the per-frame sequence Tracking runs (Frame::Frame -> ExtractORB, ComputeBoW,
SearchByProjection(Cur, Last), SearchLocalPoints -> SearchByProjection(F, points); reference src/Frame.cc:311,418-425,738-745,
src/Tracking.cc:2889,3416) as ONE loop over a synthetic stream, through the three adapters together.  The reference build
(oracle/_ref/ref_streamed_frontend = the reference's own src/ORBextractor.cc + src/ORBmatcher.cc + DBoW2, compiled where they lie)
and the drop-in must produce the same digest of everything the loop computes: keypoints, descriptors, BoW / feature vectors and
both searches' match vectors, frame after frame (each frame's searches consume the previous frames' extractions)."""
import os

import pytest

from tests import world_util as wu

SHAPES = [(480, 640, 1000, 10), (480, 752, 1200, 8)]


def _ref(tmp_path, rows, cols, nfeatures, nframes):
    if not os.path.exists(wu.REF_FRONTEND_EXE):
        pytest.skip("oracle/_ref/ref_streamed_frontend not built (needs /root/reference)")
    raw, voc = wu.frontend_inputs(str(tmp_path), nframes, rows, cols, nfeatures)
    ref = wu.run_frontend(wu.REF_FRONTEND_EXE, raw, rows, cols, nframes, nfeatures, voc, 1)
    assert ref["frames_timed"] == nframes - 5 and ref["matches_last_per_frame"] > 200 and ref["matches_local_per_frame"] > 200, ref
    return raw, voc, ref


@pytest.mark.parametrize("rows,cols,nfeatures,nframes", SHAPES[:1])
def test_streamed_frontend_host_logic_equals_reference(tmp_path, rows, cols, nfeatures, nframes):
    raw, voc, ref = _ref(tmp_path, rows, cols, nfeatures, nframes)
    got = wu.run_frontend(wu.build_frontend("oracle"), raw, rows, cols, nframes, nfeatures, voc, 1)
    assert got["results_digest"] == ref["results_digest"], (got, ref)


@pytest.mark.gpu
@pytest.mark.parametrize("rows,cols,nfeatures,nframes", SHAPES)
def test_streamed_frontend_on_gpu_equals_reference(tmp_path, rows, cols, nfeatures, nframes):
    raw, voc, ref = _ref(tmp_path, rows, cols, nfeatures, nframes)
    got = wu.run_frontend(wu.build_frontend("orbx"), raw, rows, cols, nframes, nfeatures, voc, 2)
    assert got["results_digest"] == ref["results_digest"], (got, ref)
    assert got["features_per_frame"] == ref["features_per_frame"] and got["matches_local_per_frame"] == ref["matches_local_per_frame"]


# ---- the long form: BASELINE config 3 at length (3 682 MH_01 frames at 20 Hz: tools/config3_full.py on the GPU box, profiles/config3_full_r6.txt);
# here: the ring of 8 frames, the forth-and-back walk over a short image set and the pacing, against the reference-compiled build
def test_long_form_host_logic_equals_reference(tmp_path):
    raw, voc, _ = _ref(tmp_path, 480, 640, 1000, 10)
    ref = wu.run_frontend(wu.REF_FRONTEND_EXE, raw, 480, 640, 10, 1000, voc, 1, frames=45)
    got = wu.run_frontend(wu.build_frontend("oracle"), raw, 480, 640, 10, 1000, voc, 1, frames=45)
    assert ref["frames_timed"] == 45 - 8 and ref["stream"]["ring"] == 8 and ref["matches_last_per_frame"] > 200
    assert got["results_digest"] == ref["results_digest"], (got, ref)
    assert set(got["percentiles"]) == {"extract_ms", "bow_ms", "search_last_ms", "search_local_ms", "four_calls_ms", "frame_wall_ms"}


def test_mh01_stamps_fixture_and_pacing(tmp_path):
    stamps = wu.mh01_stamps(str(tmp_path))
    ns = [int(x) for x in open(stamps).read().split()]
    assert len(ns) == 3682 and abs((ns[-1] - ns[0]) / 1e9 - 184.05) < 0.01   # 20 Hz: Examples/Monocular/EuRoC_TimeStamps/MH01.txt
    if not os.path.exists(wu.REF_FRONTEND_EXE):
        pytest.skip("oracle/_ref/ref_streamed_frontend not built (needs /root/reference)")
    raw, voc = wu.frontend_inputs(str(tmp_path), 6, 240, 320, 300)
    free = wu.run_frontend(wu.REF_FRONTEND_EXE, raw, 240, 320, 6, 300, voc, 1, frames=20)
    paced = wu.run_frontend(wu.REF_FRONTEND_EXE, raw, 240, 320, 6, 300, voc, 1, frames=20, stamps=stamps, pace=1)
    assert paced["results_digest"] == free["results_digest"]
    # 19 waits of 50 ms less the tracking time: the paced pass takes the stream's own duration
    assert paced["stream"]["paced"] and 0.90 < paced["stream"]["wall_s"] < 4.0 and paced["stream"]["slept_s"] > 0.2, paced["stream"]      # (loose above: a loaded host)
    assert free["stream"]["wall_s"] < paced["stream"]["wall_s"]


@pytest.mark.gpu
def test_long_form_on_gpu_equals_reference_over_512_frames(tmp_path):
    """512 frames through a ring of 8: more Frames than the 64 resident search targets of a thread context, so the LRU cycles eight times;
    every frame's extraction, BoW vectors and both searches against the reference's own code."""
    if not os.path.exists(wu.REF_FRONTEND_EXE):
        pytest.skip("oracle/_ref/ref_streamed_frontend not built (needs /root/reference)")
    raw, voc = wu.frontend_inputs(str(tmp_path), 48, 480, 640, 1000)
    ref = wu.run_frontend(wu.REF_FRONTEND_EXE, raw, 480, 640, 48, 1000, voc, 1, frames=512)
    got = wu.run_frontend(wu.build_frontend("orbx"), raw, 480, 640, 48, 1000, voc, 1, frames=512)
    assert got["frames_timed"] == 504 and got["results_digest"] == ref["results_digest"], (got, ref)
    assert got["matches_last_per_frame"] == ref["matches_last_per_frame"] and got["wide_retries"] == ref["wide_retries"]
