"""The oracle's matcher / bag-of-words restatement against the REFERENCE'S OWN DBoW2 code, compiled from
/root/reference by oracle/ref_fragments.mk into oracle/_ref/libref_dbow2.so (travels to the GPU box prebuilt)."""
import numpy as np
import pytest

from oracle import pyoracle as po
from tests.vocab_util import make_vocabulary

pytestmark = pytest.mark.skipif(not po.ref_available(), reason="oracle/_ref/libref_dbow2.so not built")


def test_forb_distance_is_popcount():
    rng = np.random.default_rng(5)
    d = rng.integers(0, 256, (400, 32)).astype(np.uint8)
    for i in range(399):
        assert po.hamming(d[i], d[i + 1]) == po.ref_hamming(d[i], d[i + 1]) == int(np.unpackbits(d[i] ^ d[i + 1]).sum())


@pytest.mark.parametrize("k,L,seed", [(10, 3, 1), (4, 5, 2), (7, 2, 3)])
def test_transform_and_score_match_reference(tmp_path, k, L, seed):
    rng = np.random.default_rng(seed)
    base = rng.integers(0, 256, (40, 32)).astype(np.uint8)
    d = base[rng.integers(0, 40, 2500)] ^ (rng.random((2500, 32)) < 0.08).astype(np.uint8) * rng.integers(0, 256, (2500, 32)).astype(np.uint8)
    p = str(tmp_path / "voc.txt")
    info = make_vocabulary(p, d, k, L, seed)
    ov, rv = po.OracleVocabulary(p), po.RefVocabulary(p)
    assert rv.size() == info["words"]
    bows = []
    for ls in range(0, L + 2):
        (oi, ovals), ofv = ov.transform(d[:700], ls)
        (ri, rvals), rfv = rv.transform(d[:700], ls)
        assert np.array_equal(oi, ri) and ovals.tobytes() == rvals.tobytes()
        # When a word sits ABOVE level L-levelsup the reference never assigns `nid` and reads an uninitialised
        # NodeId (TemplatedVocabulary.h:1150-1158,1226-1252): undefined behaviour, excluded from parity
        # (DESIGN.md "known divergences"); the oracle and orbx report node 0 there.
        if L - ls <= info["min_leaf_depth"]:
            assert ofv == rfv
    for s in range(4):
        bows.append(ov.transform(d[s * 500:(s + 1) * 500], 2)[0])
    for a in bows:
        for b in bows:
            assert po.score_l1(a, b) == rv.score(a, b)
