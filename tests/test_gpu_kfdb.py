"""KeyFrameDatabase on the GPU (BowVectors resident in HBM, full scan) against the oracle's literal restatement with the
reference's inverted file (src/KeyFrameDatabase.cc:39-66 and the opening of the Detect* routines, :100-165, :468-535,
:604-665, :733-790): same keyframe list in the same order, same word counts, same thresholds, bit-identical scores."""
import numpy as np
import pytest

from oracle import pyoracle as po
from orb_slam3_modified_amd import KeyFrameDatabase, ORBextractor

pytestmark = pytest.mark.gpu


def _bow(rng, place, nwords=60000, n=None):
    """A BowVector-like sparse vector: words mostly from the place's pool, L1-normalised like BowVector::normalize."""
    n = n or int(rng.integers(300, 900))
    pool = (place * 1500 + rng.integers(0, 2500, int(n * 0.8))) % nwords
    rest = rng.integers(0, nwords, n - len(pool))
    ids = np.unique(np.concatenate([pool, rest])).astype(np.uint32)
    vals = rng.uniform(0.1, 8.0, len(ids))
    return ids, (vals / vals.sum()).astype(np.float64)


def _same(a, b):
    assert np.array_equal(a["kf"], b["kf"]) and np.array_equal(a["words"], b["words"])
    assert a["max_common"] == b["max_common"] and a["min_common"] == b["min_common"]
    assert a["score"].tobytes() == b["score"].tobytes()


def test_kfdb_queries_equal_inverted_file_oracle():
    rng = np.random.default_rng(5)
    ex = ORBextractor(1000, 1.2, 8, 20, 7)
    db, odb = KeyFrameDatabase(ex), po.OracleKeyFrameDatabase()
    alive = []
    next_id = 0

    def add(k):
        nonlocal next_id
        for _ in range(k):
            b = _bow(rng, int(rng.integers(0, 12)))
            kid = next_id * 7 + 3          # ids are the caller's, not dense
            next_id += 1
            db.add(kid, b); odb.add(kid, b); alive.append(kid)

    def check_queries(nqueries):
        scored = 0
        for _ in range(nqueries):
            q = _bow(rng, int(rng.integers(0, 12)))
            excl = list(rng.choice(alive, size=min(len(alive), int(rng.integers(0, 15))), replace=False)) if alive else []
            floor = int(rng.choice([0, 0, 0, 40]))
            a, b = db.query(q, excl, floor), odb.query(q, excl, floor)
            _same(a, b)
            assert not set(excl) & set(a["kf"].tolist())
            scored += int((a["score"] >= 0).sum())
            assert ((a["score"] >= 0) == (a["words"] > a["min_common"])).all()
        return scored

    assert db.query(_bow(rng, 0))["kf"].size == 0           # empty database
    add(150)
    assert len(db) == 150
    assert check_queries(12) > 20
    for kid in list(rng.choice(alive, 60, replace=False)):  # erase, then add more: insertion order inside the word lists matters
        db.erase(kid); odb.erase(kid); alive.remove(kid)
    db.erase(123456789); odb.erase(123456789)               # absent keyframe: no-op
    assert len(db) == 90
    check_queries(8)
    add(120)
    check_queries(8)
    for kid in list(alive[:170]):                           # > half of the payload dead -> compaction path
        db.erase(kid); odb.erase(kid); alive.remove(kid)
    add(30)
    assert len(db) == len(alive) == 70
    assert check_queries(10) > 10
    # a keyframe queried against a database that contains itself: score 1 (within rounding), listed first for its first word
    kid = alive[5]
    self_bow = None
    db.clear()
    assert len(db) == 0 and db.query(_bow(rng, 1))["kf"].size == 0
    b = _bow(rng, 3)
    db.add(1, b)
    r = db.query(b)
    assert r["kf"].tolist() == [1] and r["words"][0] == len(b[0]) and abs(r["score"][0] - 1.0) < 1e-12


def test_kfdb_rejects_bad_input():
    from orb_slam3_modified_amd import OrbxError
    ex = ORBextractor(1000, 1.2, 8, 20, 7)
    db = KeyFrameDatabase(ex)
    ids = np.array([5, 9, 9], np.uint32); vals = np.ones(3)
    with pytest.raises(OrbxError):
        db.add(1, (ids, vals))                 # not strictly ascending
    db.add(1, (np.array([5, 9], np.uint32), np.ones(2)))
    with pytest.raises(OrbxError):
        db.add(1, (np.array([5, 9], np.uint32), np.ones(2)))   # duplicate id
    with pytest.raises(OrbxError):
        db.query((ids, vals))


def test_kfdb_long_queries_are_served_from_hbm():
    """A BowVector of more than 8192 words (the 20 000-feature frames of the fork's Examples/Monocular/mi.yaml) no longer fits
    the kernels' LDS staging; the query is then searched where it lies.  The keyframe owns an extractor that is otherwise
    unreferenced: the database keeps it alive (KeyFrameDatabase(ORBextractor(...)))."""
    rng = np.random.default_rng(9)
    db, odb = KeyFrameDatabase(ORBextractor(1000, 1.2, 8, 20, 7)), po.OracleKeyFrameDatabase()
    import gc
    gc.collect()
    def long_bow(n):
        ids = np.unique(rng.integers(0, 60000, n + n // 3))[:n].astype(np.uint32)   # dense enough to share thousands of words
        vals = rng.uniform(0.1, 8.0, len(ids))
        return ids, (vals / vals.sum()).astype(np.float64)

    for kid in range(40):
        b = long_bow(int(rng.integers(9000, 14000)))
        db.add(kid, b); odb.add(kid, b)
    for _ in range(4):
        q = long_bow(int(rng.integers(8500, 15000)))
        assert len(q[0]) > 8192
        a, b = db.query(q, [3, 17], 0), odb.query(q, [3, 17], 0)
        _same(a, b)
        assert (a["score"] >= 0).sum() > 0


def test_kfdb_two_phase_entry_points_equal_oracle():
    """orbx_kfdb_sharing / orbx_kfdb_score — the passes the drop-in KeyFrameDatabase class is made of: the full sharing list in the
    inverted file's order (no exclusion) and the L1 score of an arbitrary selection, both against the oracle; interleaved with
    erase / re-add (a re-added keyframe moves to the end of every list)."""
    rng = np.random.default_rng(17)
    ex = ORBextractor(1000, 1.2, 8, 20, 7)
    db, odb = KeyFrameDatabase(ex), po.OracleKeyFrameDatabase()
    bows = {}
    for kid in range(5, 5 + 240):
        bows[kid] = _bow(rng, int(rng.integers(0, 10)))
        db.add(kid, bows[kid]); odb.add(kid, bows[kid])
    for rnd in range(3):
        for _ in range(6):
            q = _bow(rng, int(rng.integers(0, 10)))
            kf, words = db.sharing(q[0])
            want = odb.query(q, [], 0)
            assert np.array_equal(kf, want["kf"]) and np.array_equal(words, want["words"]) and len(kf) > 50
            sel = [int(k) for k in kf[rng.permutation(len(kf))[:40]]]          # any selection, any order
            got = db.score(q, sel)
            ref = np.array([po.score_l1(q, bows[k]) for k in sel])
            assert got.tobytes() == ref.tobytes()
            assert db.score(q, []).size == 0
        gone = [int(k) for k in rng.choice(sorted(bows), 25, replace=False)]
        for k in gone:
            db.erase(k); odb.erase(k)
        back = gone[:10]
        for k in back:
            db.add(k, bows[k]); odb.add(k, bows[k])
        for k in gone[10:]:
            del bows[k]
    with pytest.raises(Exception):
        db.score(_bow(rng, 1), [10 ** 9])          # not in the database: refused, not scored as something else
