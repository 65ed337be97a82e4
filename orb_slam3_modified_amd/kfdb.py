"""Host-side mirror of the data-parallel part of ORB_SLAM3::KeyFrameDatabase (reference src/KeyFrameDatabase.cc): the
keyframes' BowVectors live in HBM; `query` is the first two phases shared by the five Detect* routines."""
from __future__ import annotations

import ctypes as C
from typing import Iterable, Tuple

import numpy as np

from ._lib import check, lib, ptr


class KeyFrameDatabase:
    def __init__(self, extractor_or_ctx):
        self._L = lib()
        self._owner = extractor_or_ctx   # keeps the extractor (and with it the orbx context) alive for as long as this object uses it
        self._ctx = getattr(extractor_or_ctx, "_ctx", extractor_or_ctx)
        self._db = C.c_void_p(0)
        check(self._L.orbx_kfdb_create(self._ctx, C.byref(self._db)), self._ctx)

    def close(self):
        if getattr(self, "_db", None) and self._db.value:
            self._L.orbx_kfdb_destroy(self._db)
            self._db = C.c_void_p(0)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __len__(self) -> int:
        return int(self._L.orbx_kfdb_size(self._db))

    def add(self, kf_id: int, bow: Tuple[np.ndarray, np.ndarray]) -> None:
        """KeyFrameDatabase::add (:39-45); bow = (ascending word ids uint32, values float64) = pKF->mBowVec."""
        ids = np.ascontiguousarray(bow[0], np.uint32); vals = np.ascontiguousarray(bow[1], np.float64)
        assert len(ids) == len(vals)
        check(self._L.orbx_kfdb_add(self._db, int(kf_id), ptr(ids), ptr(vals), len(ids)), self._ctx)

    def erase(self, kf_id: int) -> None:
        """KeyFrameDatabase::erase (:47-66)."""
        check(self._L.orbx_kfdb_erase(self._db, int(kf_id)), self._ctx)

    def clear(self) -> None:
        """KeyFrameDatabase::clear (:68-72)."""
        check(self._L.orbx_kfdb_clear(self._db), self._ctx)

    def query(self, bow: Tuple[np.ndarray, np.ndarray], exclude: Iterable[int] = (), min_words_floor: int = 0):
        """-> dict(kf = ids of the keyframes sharing a word, in the reference's lKFsSharingWords order, words = their common
        word counts, score = mpVoc->score(query, kf) where words > min_common else -1, max_common, min_common)."""
        ids = np.ascontiguousarray(bow[0], np.uint32); vals = np.ascontiguousarray(bow[1], np.float64)
        ex = np.ascontiguousarray(list(exclude), np.int64)
        cap = max(len(self), 1)
        kf = np.zeros(cap, np.int64); words = np.zeros(cap, np.int32); score = np.zeros(cap, np.float64)
        n, mx, mn = C.c_int(0), C.c_int(0), C.c_int(0)
        check(self._L.orbx_kfdb_query(self._db, ptr(ids), ptr(vals), len(ids), ptr(ex) if len(ex) else None, len(ex), int(min_words_floor),
                                      ptr(kf), ptr(words), ptr(score), cap, C.byref(n), C.byref(mx), C.byref(mn)), self._ctx)
        k = n.value
        return dict(kf=kf[:k].copy(), words=words[:k].copy(), score=score[:k].copy(), max_common=mx.value, min_common=mn.value)

    def sharing(self, word_ids) -> Tuple[np.ndarray, np.ndarray]:
        """orbx_kfdb_sharing: (keyframe ids in the reference's list order, common word counts) of EVERY keyframe sharing a word."""
        ids = np.ascontiguousarray(word_ids, np.uint32)
        cap = max(len(self), 1)
        kf = np.zeros(cap, np.int64); words = np.zeros(cap, np.int32)
        n = C.c_int(0)
        check(self._L.orbx_kfdb_sharing(self._db, ptr(ids), len(ids), ptr(kf), ptr(words), cap, C.byref(n)), self._ctx)
        return kf[:n.value].copy(), words[:n.value].copy()

    def score(self, bow: Tuple[np.ndarray, np.ndarray], kf_ids) -> np.ndarray:
        """orbx_kfdb_score: mpVoc->score(query, keyframe) (L1) for the listed keyframes."""
        ids = np.ascontiguousarray(bow[0], np.uint32); vals = np.ascontiguousarray(bow[1], np.float64)
        kf = np.ascontiguousarray(list(kf_ids), np.int64)
        out = np.zeros(max(len(kf), 1), np.float64)
        check(self._L.orbx_kfdb_score(self._db, ptr(ids), ptr(vals), len(ids), ptr(kf) if len(kf) else None, len(kf), ptr(out)), self._ctx)
        return out[:len(kf)].copy()
