"""Host-side mirror of ORB_SLAM3::ORBVocabulary = DBoW2::TemplatedVocabulary<cv::Mat, FORB>
(reference include/ORBVocabulary.h:27-29): loadFromTextFile, transform (tree descent on the GPU), score."""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Tuple

import numpy as np

from ._lib import OrbxError, check, lib, ptr


class ORBVocabulary:
    def __init__(self, extractor_or_ctx):
        self._L = lib()
        self._owner = extractor_or_ctx   # keeps the extractor (and with it the orbx context) alive for as long as this object uses it
        self._ctx = getattr(extractor_or_ctx, "_ctx", extractor_or_ctx)
        self._voc = C.c_void_p(0)

    def __del__(self):
        try:
            if self._voc and self._voc.value:
                self._L.orbx_voc_destroy(self._voc)
                self._voc = C.c_void_p(0)
        except Exception:
            pass

    def loadFromTextFile(self, filename: str) -> bool:
        """TemplatedVocabulary.h:1338-1424; returns False on a malformed file like the reference."""
        # as include/ORBVocabulary.h (and the reference: loadFromTextFile clears the tree first): the old vocabulary goes away BEFORE the new one
        # is built — the new one may well come back at the same address
        if self._voc and self._voc.value:
            self._L.orbx_voc_destroy(self._voc)
            self._voc = C.c_void_p(0)
        v = C.c_void_p(0)
        rc = self._L.orbx_voc_load_text(self._ctx, filename.encode(), C.byref(v))
        if rc != 0:
            return False
        self._voc = v
        return True

    def saveToTextFile(self, filename: str) -> None:
        """TemplatedVocabulary.h:1428-1449 (byte-identical to the reference's writer)."""
        check(self._L.orbx_voc_save_text(self._voc, filename.encode()), self._ctx)

    def saveBinary(self, filename: str) -> None:
        """Exact binary cache (not a reference format)."""
        check(self._L.orbx_voc_save_binary(self._voc, filename.encode()), self._ctx)

    def loadBinary(self, filename: str) -> bool:
        v = C.c_void_p(0)
        if self._L.orbx_voc_load_binary(self._ctx, filename.encode(), C.byref(v)) != 0:
            return False
        if self._voc and self._voc.value:
            self._L.orbx_voc_destroy(self._voc)
        self._voc = v
        return True

    def info(self):
        k, L, n, w = (C.c_int(0) for _ in range(4))
        check(self._L.orbx_voc_info(self._voc, C.byref(k), C.byref(L), C.byref(n), C.byref(w)), self._ctx)
        return dict(k=k.value, L=L.value, nodes=n.value, words=w.value)

    def descend(self, desc: np.ndarray, levelsup: int = 4):
        d = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
        n = len(d)
        word, node = np.zeros(max(n, 1), np.uint32), np.zeros(max(n, 1), np.uint32)
        weight = np.zeros(max(n, 1), np.float64)
        check(self._L.orbx_bow_transform(self._voc, ptr(d), n, levelsup, ptr(word), ptr(weight), ptr(node)), self._ctx)
        return word[:n], weight[:n], node[:n]

    def descend_published(self, desc: np.ndarray, levelsup: int = 4):
        """orbx_bow_transform_published: `desc` is the very array an ORBextractor's publish_descriptors() named (the rows of its last
        single-frame extraction).  Returns (word, weight, node) when that extraction's own graph already ran the descent with this
        vocabulary, else None — and attaches the vocabulary to that extractor for its next extractions."""
        assert desc.dtype == np.uint8 and desc.flags["C_CONTIGUOUS"]
        n = int(desc.shape[0])
        word, node = np.zeros(max(n, 1), np.uint32), np.zeros(max(n, 1), np.uint32)
        weight = np.zeros(max(n, 1), np.float64)
        rc = check(self._L.orbx_bow_transform_published(self._voc, ptr(desc), n, levelsup, ptr(word), ptr(weight), ptr(node)), self._ctx)
        return (word[:n], weight[:n], node[:n]) if rc == 0 else None

    def transform(self, desc: np.ndarray, levelsup: int = 4
                  ) -> Tuple[Tuple[np.ndarray, np.ndarray], Dict[int, List[int]]]:
        """transform(features, BowVector&, FeatureVector&, levelsup), TemplatedVocabulary.h:1127-1194.
        Returns ((ids ascending, values), {node id: [feature indices]})."""
        word, weight, node = self.descend(desc, levelsup)
        n = len(word)
        ids, vals, nout = np.zeros(max(n, 1), np.uint32), np.zeros(max(n, 1), np.float64), C.c_int(0)
        check(self._L.orbx_bow_finalize(self._voc, ptr(word), ptr(weight), n, ptr(ids), ptr(vals), C.byref(nout)), self._ctx)
        fv: Dict[int, List[int]] = {}
        for i in range(n):
            if weight[i] > 0:
                fv.setdefault(int(node[i]), []).append(i)  # FeatureVector::addFeature, FeatureVector.cpp:30-45
        return (ids[:nout.value].copy(), vals[:nout.value].copy()), dict(sorted(fv.items()))

    def score(self, a, b) -> float:
        """L1Scoring::score (ScoringObject.cpp:23-68) of two (ids, values) vectors."""
        ia, va = np.ascontiguousarray(a[0], np.uint32), np.ascontiguousarray(a[1], np.float64)
        ib, vb = np.ascontiguousarray(b[0], np.uint32), np.ascontiguousarray(b[1], np.float64)
        return float(self._L.orbx_bow_score_l1(ptr(ia), ptr(va), len(ia), ptr(ib), ptr(vb), len(ib)))

    def score_batch(self, q, db: List[Tuple[np.ndarray, np.ndarray]]) -> np.ndarray:
        """One query BowVector against many (KeyFrameDatabase scoring loops), on the GPU."""
        qi, qv = np.ascontiguousarray(q[0], np.uint32), np.ascontiguousarray(q[1], np.float64)
        ptrs = np.zeros(len(db) + 1, np.int32)
        for i, (ids, _) in enumerate(db):
            ptrs[i + 1] = ptrs[i] + len(ids)
        di = np.concatenate([np.asarray(d[0], np.uint32) for d in db]) if db else np.zeros(0, np.uint32)
        dv = np.concatenate([np.asarray(d[1], np.float64) for d in db]) if db else np.zeros(0, np.float64)
        di, dv = np.ascontiguousarray(di), np.ascontiguousarray(dv)
        out = np.zeros(max(len(db), 1), np.float64)
        check(self._L.orbx_bow_score_l1_batch(self._ctx, ptr(qi), ptr(qv), len(qi), ptr(ptrs), ptr(di), ptr(dv), len(db),
                                              ptr(out)), self._ctx)
        return out[:len(db)]


def write_text_vocabulary(path: str, k: int, L: int, parent, is_leaf, desc, weight, scoring: int = 0, weighting: int = 0):
    """saveToTextFile layout (TemplatedVocabulary.h:1428-1449): header + one line per node, NO trailing newline."""
    lines = [f"{k} {L} {scoring} {weighting}"]
    for p, lf, d, w in zip(parent, is_leaf, desc, weight):
        lines.append(f"{int(p)} {int(lf)} " + " ".join(str(int(b)) for b in d) + f" {float(w)!r}")
    with open(path, "w") as f:
        f.write("\n".join(lines))
