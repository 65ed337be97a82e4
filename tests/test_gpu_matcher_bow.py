"""GPU matcher primitives and bag of words against the oracle (which test_ref_fragments.py pins to the reference's
own DBoW2 code): bit-exact integers, bit-identical doubles."""
import numpy as np
import pytest

from oracle import pyoracle as po
from orb_slam3_modified_amd import ORBextractor, ORBmatcher, ORBVocabulary, synth
from tests.vocab_util import make_vocabulary

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def feats():
    gpu = ORBextractor(1000, 1.2, 8, 20, 7)
    fr = synth.make_stream(3)
    out = [gpu(f, None, (0, 1000)) for f in fr]
    return gpu, out


def _grid_candidates(k0, k1, r=20.0, seed=0):
    """CSR candidate lists like Frame::GetFeaturesInArea windows (src/Frame.cc:657-723)."""
    rp, cand = [0], []
    for p in k0:
        m = np.nonzero((np.abs(k1["x"] - p["x"]) < r) & (np.abs(k1["y"] - p["y"]) < r))[0]
        cand.extend(m.tolist()); rp.append(len(cand))
    return np.array(rp, np.int32), np.array(cand, np.int32)


def test_nn_csr_first_and_last_wins(feats):
    gpu, out = feats
    (_, k0, d0), (_, k1, d1) = out[0], out[1]
    m = ORBmatcher(gpu)
    rp, cand = _grid_candidates(k0, k1)
    assert rp[-1] > 5 * len(k0)
    for last in (False, True):
        g = m.nn_csr(d0, d1, rp, cand, last, want_dist=True)
        o = po.nn_csr(d0, d1, rp, cand, last)
        for a, b, name in zip(g, o, ("best_idx", "best_dist", "second_idx", "second_dist", "dist")):
            assert np.array_equal(a, b), (last, name)
    # heavy ties: duplicate descriptors, long lists, empty lists
    rng = np.random.default_rng(1)
    t = np.repeat(d1[:50], 8, axis=0)
    lens = rng.integers(0, 300, 200)
    lens[::7] = 0
    rp = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    cand = rng.integers(0, len(t), rp[-1]).astype(np.int32)
    for last in (False, True):
        g = m.nn_csr(d0[:200], t, rp, cand, last)
        o = po.nn_csr(d0[:200], t, rp, cand, last)
        for a, b in zip(g, o[:4]):
            assert np.array_equal(a, b)
    assert (g[0][::7] == -1).all() and (g[1][::7] == 256).all()


def test_nn_groups_near_candidates_in_list_order(feats):
    """orbx_nn_groups (the node-by-node searches): per query the candidates of its group within max_dist, in list order, against a plain
    numpy statement; empty groups, groups longer than a wave, a pool that is too small (ORBX_E_CAPACITY with the needed size)."""
    from orb_slam3_modified_amd import OrbxError
    gpu, out = feats
    q, t = out[0][2], out[1][2]
    rng = np.random.default_rng(4)
    m = ORBmatcher(gpu)
    sizes = [0, 1, 63, 64, 65, 200, 300, 5]
    group_ptr = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    group_cand = rng.integers(0, len(t), group_ptr[-1]).astype(np.int32)
    q_group = rng.integers(0, len(sizes), len(q)).astype(np.int32)
    pop = np.array([bin(i).count("1") for i in range(256)], np.int32)
    for max_dist in (30, 50, 72, 110, 256):
        want_off, want_cnt, want = [], [], []
        for i in range(len(q)):
            c = group_cand[group_ptr[q_group[i]]:group_ptr[q_group[i] + 1]]
            d = pop[np.bitwise_xor(q[i][None, :], t[c])].sum(1) if len(c) else np.zeros(0, np.int32)
            keep = d <= max_dist
            want_off.append(len(want)); want_cnt.append(int(keep.sum()))
            want += list(zip(c[keep].tolist(), d[keep].tolist()))
        try:
            off, cnt, ent = m.nn_groups(q, q_group, t, group_ptr, group_cand, max_dist, len(want) + 3)
        except OrbxError:
            raise
        assert cnt.tolist() == want_cnt
        for i in range(len(q)):   # the pool is filled in whatever order the waves reserve it: compare per query
            assert [tuple(e) for e in ent[off[i]:off[i] + cnt[i]].tolist()] == want[want_off[i]:want_off[i] + want_cnt[i]], (max_dist, i)
        if len(want) > 10:
            with pytest.raises(OrbxError) as ei:
                m.nn_groups(q, q_group, t, group_ptr, group_cand, max_dist, len(want) - 5)
            assert ei.value.code == -4 or "pool" in str(ei.value)
            # and the next call works again (the pass's counters reset themselves)
            off2, cnt2, _ = m.nn_groups(q, q_group, t, group_ptr, group_cand, max_dist, len(want))
            assert cnt2.tolist() == want_cnt


def test_knn2_allpairs(feats):
    gpu, out = feats
    m = ORBmatcher(gpu)
    d0, d1 = out[0][2], out[2][2]
    gi, gd = m.knn2(d0, d1)
    oi, od = po.knn2(d0, d1)
    assert np.array_equal(gi, oi) and np.array_equal(gd, od)
    gi, gd = m.knn2(d0[:5], d1[:1])      # fewer than 2 train rows
    assert (gi[:, 1] == -1).all() and (gd[:, 1] == 256).all() and (gi[:, 0] == 0).all()
    gi, gd = m.knn2(d0[:300], np.repeat(d1[:3], 100, axis=0))   # ties -> lower train index
    oi, od = po.knn2(d0[:300], np.repeat(d1[:3], 100, axis=0))
    assert np.array_equal(gi, oi) and np.array_equal(gd, od)


@pytest.mark.parametrize("k,L", [(10, 3), (10, 6), (3, 7), (20, 2)])
def test_bow_transform_and_scores(feats, tmp_path, k, L):
    gpu, out = feats
    alld = np.concatenate([o[2] for o in out])
    p = str(tmp_path / "voc.txt")
    info = make_vocabulary(p, alld, k, L, seed=k * 10 + L)
    gv = ORBVocabulary(gpu)
    assert gv.loadFromTextFile(p)
    assert gv.info()["words"] == info["words"] and gv.info()["nodes"] == info["nodes"] + 1
    ov = po.OracleVocabulary(p)
    bows = []
    for ls in (0, 2, 4):
        for o in out:
            (gi, gvals), gfv = gv.transform(o[2], ls)
            (oi, ovals), ofv = ov.transform(o[2], ls)
            assert np.array_equal(gi, oi) and gvals.tobytes() == ovals.tobytes() and gfv == ofv
            bows.append((gi, gvals))
    for a in bows[:3]:
        host = [gv.score(a, b) for b in bows]
        dev = gv.score_batch(a, bows)
        ref = [po.score_l1(a, b) for b in bows]
        assert host == ref and dev.tolist() == ref


def test_bow_descent_inside_the_extraction_graph(tmp_path):
    """Frame::ComputeBoW after Frame::ExtractORB without a device round trip of its own: the first published transform attaches the
    vocabulary to the extractor that produced the rows; from its next extraction on the single-frame graph ends with the tree descent and
    the records are served from the extractor's pinned block.  Same words / weights / nodes as the ordinary call and as the oracle; a
    buffer whose bytes changed, another levelsup, a batch extraction, another extractor or a destroyed vocabulary are not served."""
    frames = synth.make_stream(5)
    ex = ORBextractor(1000, 1.2, 8, 20, 7)
    d0 = ex(frames[0], None, (0, 1000))[2]
    p = str(tmp_path / "voc.txt")
    make_vocabulary(p, d0, 10, 4, seed=5)
    gv = ORBVocabulary(ex)
    assert gv.loadFromTextFile(p)
    ov = po.OracleVocabulary(p)

    def extract(i, e=ex):
        d = np.ascontiguousarray(e(frames[i], None, (0, 1000))[2])
        e.publish_descriptors(d)
        return d

    d = extract(0)
    assert gv.descend_published(d, 4) is None            # not attached yet: this call attaches
    served = 0
    for i in (1, 2, 3, 1):
        d = extract(i)
        got = gv.descend_published(d, 4)
        assert got is not None, i
        w, wt, nd = got
        ow, owt, ond = gv.descend(d, 4)
        assert np.array_equal(w, ow) and wt.tobytes() == owt.tobytes() and np.array_equal(nd, ond)
        (oi, ovals), ofv = ov.transform(d, 4)
        (gi, gvals), gfv = gv.transform(d, 4)
        assert np.array_equal(gi, oi) and gvals.tobytes() == ovals.tobytes() and gfv == ofv
        served += 1
        assert gv.descend_published(d, 4) is not None     # asking twice is fine (the records stay until the next extraction)
    assert served == 4
    # the caller's buffer no longer holds the published bytes
    d = extract(2)
    d[5, 8 + 3] ^= 0xff     # (the digest reads 8 bytes of every row: bytes 8..15 of row 5)
    assert gv.descend_published(d, 4) is None
    # another levelsup: not what the graph computed (and the context is re-attached with the new value)
    d = extract(3)
    assert gv.descend_published(d, 2) is None
    d = extract(3)
    got = gv.descend_published(d, 2)
    assert got is not None and np.array_equal(got[2], gv.descend(d, 2)[2])
    # a batch extraction publishes nothing of the kind
    ex.extract_batch(frames[:2], (0, 1000))
    assert gv.descend_published(d, 2) is None
    # rows of another extractor are found through THAT extractor
    ex2 = ORBextractor(1000, 1.2, 8, 20, 7)
    d2 = extract(4, ex2)
    assert gv.descend_published(d2, 2) is None
    d2 = extract(4, ex2)
    assert gv.descend_published(d2, 2) is not None
    # a vocabulary that goes away detaches itself; extraction keeps working
    del gv
    import gc; gc.collect()
    k = ex(frames[0], None, (0, 1000))
    okps, odesc, omono = po.OracleExtractor(1000, 1.2, 8, 20, 7).extract(frames[0], (0, 1000))
    assert k[0] == omono and k[1].tobytes() == okps.tobytes() and np.array_equal(k[2], odesc)


def test_vocabulary_loader_rejects_garbage(feats, tmp_path):
    gpu, _ = feats
    p = tmp_path / "bad.txt"
    p.write_text("99 1 0 0\n0 1 " + " ".join(["0"] * 32) + " 1.0")
    assert not ORBVocabulary(gpu).loadFromTextFile(str(p))          # k > 20: TemplatedVocabulary.h:1359
    assert not ORBVocabulary(gpu).loadFromTextFile(str(tmp_path / "missing.txt"))
    good = tmp_path / "good.txt"
    good.write_text("2 1 0 0\n0 1 " + " ".join(["0"] * 32) + " 1.5\n0 1 " + " ".join(["255"] * 32) + " 2.5\n\n")  # trailing blank lines
    v = ORBVocabulary(gpu)
    assert v.loadFromTextFile(str(good)) and v.info()["words"] == 2   # no phantom node


def test_knn2_large_train_set_segment_path(feats):
    """nt > 2048 takes the query-in-registers / segmented path (partial top-2 per 64-train segment + merge)."""
    gpu, out = feats
    m = ORBmatcher(gpu)
    rng = np.random.default_rng(4)
    base = np.concatenate([o[2] for o in out])
    t = base[rng.integers(0, len(base), 5000)] ^ (rng.random((5000, 32)) < 0.05).astype(np.uint8) * rng.integers(0, 256, (5000, 32)).astype(np.uint8)
    t[100:140] = t[60:100]                       # exact duplicates: ties must resolve to the lower train index
    q = base[:777]
    gi, gd = m.knn2(q, t)
    oi, od = po.knn2(q, t)
    assert np.array_equal(gi, oi) and np.array_equal(gd, od)


def test_vocabulary_writers_and_binary_cache(feats, tmp_path):
    """saveToTextFile: byte-identical to the REFERENCE's own writer (oracle/_ref, TemplatedVocabulary.h:1428-1449);
    the binary cache round-trips every field exactly (the text format keeps 6 digits of each weight)."""
    gpu, out = feats
    alld = np.concatenate([o[2] for o in out])
    p, p2, p3, pb = (str(tmp_path / n) for n in ("voc.txt", "saved.txt", "ref_saved.txt", "voc.bin"))
    make_vocabulary(p, alld, 6, 4, seed=77)
    gv = ORBVocabulary(gpu)
    assert gv.loadFromTextFile(p)
    gv.saveToTextFile(p2)
    if po.ref_available():
        po.RefVocabulary(p).saveToTextFile(p3)
        assert open(p2, "rb").read() == open(p3, "rb").read()
    # what was written loads again and is a fixed point of load -> save
    g2 = ORBVocabulary(gpu)
    assert g2.loadFromTextFile(p2) and g2.info() == gv.info()
    g2.saveToTextFile(str(tmp_path / "again.txt"))
    assert open(p2, "rb").read() == open(str(tmp_path / "again.txt"), "rb").read()
    # binary cache: exact
    gv.saveBinary(pb)
    g3 = ORBVocabulary(gpu)
    assert g3.loadBinary(pb) and g3.info() == gv.info()
    for o in out[:2]:
        (ai, av), afv = gv.transform(o[2], 2)
        (bi, bv), bfv = g3.transform(o[2], 2)
        assert np.array_equal(ai, bi) and av.tobytes() == bv.tobytes() and afv == bfv
    # corrupt / truncated caches are rejected
    raw = open(pb, "rb").read()
    for bad in (raw[:-5], b"XXXXVOC1" + raw[8:], raw + b"\0"):
        pbad = str(tmp_path / "bad.bin")
        open(pbad, "wb").write(bad)
        assert not ORBVocabulary(gpu).loadBinary(pbad)
    assert not ORBVocabulary(gpu).loadBinary(str(tmp_path / "missing.bin"))


def test_vocabulary_reload_while_an_extractor_descends_it(tmp_path):
    """Round-4 advisor finding: a vocabulary destroyed and loaded again can come back at the same heap address (ORBVocabulary::loadFrom* is
    destroy + load); a context that had it attached must not keep launching the descent over the freed tree, and orbx_voc_destroy must not free
    the tree under an extraction that already took its snapshot.  One thread extracts frame after frame and asks for the published records,
    another reloads the vocabulary the whole time (ctypes calls release the GIL: the threads really run concurrently).  Every served record
    must equal the ordinary descent's of the vocabulary as it is afterwards (the file never changes), nothing may hang or crash."""
    import threading
    import time
    frames = synth.make_stream(4)
    ex = ORBextractor(1000, 1.2, 8, 20, 7)
    d0 = ex(frames[0], None, (0, 1000))[2]
    p = str(tmp_path / "voc.txt")
    make_vocabulary(p, d0, 10, 3, seed=7)
    gv = ORBVocabulary(ORBextractor(1000, 1.2, 8, 20, 7))       # its OWN context, like include/ORBVocabulary.h: a context is not thread-safe
    assert gv.loadFromTextFile(p)
    ref = ORBVocabulary(ORBextractor(1000, 1.2, 8, 20, 7))      # never reloaded: the expected records
    assert ref.loadFromTextFile(p)
    stop = threading.Event()
    errors, served, reloads = [], [0], [0]
    lock = threading.Lock()                                      # the Python mirror's handle swap (not the library) needs it

    def reloader():
        try:
            while not stop.is_set():
                with lock:
                    assert gv.loadFromTextFile(p)                # orbx_voc_destroy (detach everywhere, wait for extractions in flight) + load
                reloads[0] += 1
                time.sleep(0.002)                                # (the extractor thread needs the Python lock now and then)
        except Exception as e:   # noqa: BLE001
            errors.append(("reload", repr(e)))

    def extractor():
        try:
            for it in range(40):
                d = np.ascontiguousarray(ex(frames[it % 4], None, (0, 1000))[2])
                ex.publish_descriptors(d)
                with lock:
                    got = gv.descend_published(d, 4)
                if got is not None:
                    w, wt, nd = got
                    ow, owt, ond = ref.descend(d, 4)
                    if not (np.array_equal(w, ow) and wt.tobytes() == owt.tobytes() and np.array_equal(nd, ond)):
                        errors.append(("records", it))
                    served[0] += 1
        except Exception as e:   # noqa: BLE001
            errors.append(("extract", repr(e)))
        finally:
            stop.set()

    ta, tb = threading.Thread(target=reloader), threading.Thread(target=extractor)
    ta.start(); tb.start()
    tb.join(300); stop.set(); ta.join(60)
    assert not tb.is_alive() and not ta.is_alive(), "a thread hangs"
    assert not errors, errors[:5]
    assert reloads[0] > 3      # the vocabulary really went away and came back while frames were being extracted
