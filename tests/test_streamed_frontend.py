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
