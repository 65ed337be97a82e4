"""The synthetic stand-in streams have the properties the benchmark relies on (BASELINE.md acceptance property)."""
import numpy as np

from oracle import pyoracle as po
from orb_slam3_modified_amd import synth


def test_deterministic():
    a, b = synth.make_stream(2, 120, 160), synth.make_stream(2, 120, 160)
    assert np.array_equal(a, b) and a.dtype == np.uint8
    assert not np.array_equal(a[0], a[1])
    assert not np.array_equal(synth.make_stream(1, 120, 160, synth.DEFAULT_SEED + 1000)[0], a[0])


def test_s_euroc_640_acceptance():
    fr = synth.make_stream(3)
    ex = po.OracleExtractor()
    q = ex.tables()["quota"]
    for f in fr:
        kps, desc, mono = ex.extract(f, (0, 1000))
        assert 990 <= len(kps) <= 1024 and mono == 0            # 640-wide mono: everything on the lapping side (F6)
        for l in range(4):
            assert len(ex.level_keypoints(l, 0)) >= 3 * q[l]      # sorted-expansion branch runs
        assert ex.sorted_phase_count() >= 4
        c0 = ex.level_keypoints(0, 0)
        assert (c0["response"] < 20).any()                        # minThFAST retry fired in some cell


def test_s_tumvi_1024_has_both_output_branches():
    f = synth.make_stream(1, 1024, 1024)[0]
    ex = po.OracleExtractor(2000)
    kps, desc, mono = ex.extract(f, (0, 1000))
    assert 1980 <= len(kps) <= 2024 and 0 < mono < len(kps)      # F12: keypoints with x > 1000 come first
    assert (kps["x"][:mono] > 1000).all() and (kps["x"][mono:] <= 1000).all()
