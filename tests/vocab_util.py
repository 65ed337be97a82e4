"""Synthetic DBoW2 vocabularies in the reference's text format (TemplatedVocabulary.h:1338-1449).

ORBvoc.txt is a missing blob; tests build a k-ary tree of depth L by hierarchical random
medoids over given descriptors (ids assigned like DBoW2's HKmeansStep: the k children of a node get
consecutive ids, then each child is expanded).  Weights ~ U(0.5, 8), a few leaves get weight 0 ("stopped"
words, TemplatedVocabulary.h:1157)."""
from __future__ import annotations

import numpy as np

from orb_slam3_modified_amd.vocabulary import write_text_vocabulary

_POP = np.array([bin(i).count("1") for i in range(256)], np.int32)


def _hamming_matrix(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    return _POP[a[:, None, :] ^ b[None, :, :]].sum(-1)


def make_vocabulary(path: str, descriptors: np.ndarray, k: int = 10, L: int = 3, seed: int = 0, zero_weight_frac=0.02):
    rng = np.random.default_rng(seed)
    d = np.ascontiguousarray(descriptors, np.uint8).reshape(-1, 32)
    parent, is_leaf, desc, weight, depth = [], [], [], [], []

    def expand(pid: int, idx: np.ndarray, level: int):
        kk = min(k, len(idx))
        if kk < 2:
            return
        centers = d[rng.choice(idx, kk, replace=False)].copy()
        # perturb a few bits so that centres are not exact data points and ties are possible
        flip = rng.random(centers.shape) < 0.03
        centers ^= (flip * (1 << rng.integers(0, 8, centers.shape))).astype(np.uint8)
        first = len(parent) + 1
        for c in range(kk):
            parent.append(pid); is_leaf.append(0); desc.append(centers[c]); weight.append(0.0); depth.append(level + 1)
        assign = _hamming_matrix(d[idx], centers).argmin(1)
        for c in range(kk):
            nid = first + c
            sub = idx[assign == c]
            if level + 1 < L and len(sub) >= 2:
                before = len(parent)
                expand(nid, sub, level + 1)
                if len(parent) == before:
                    is_leaf[nid - 1] = 1
            else:
                is_leaf[nid - 1] = 1
            if is_leaf[nid - 1]:
                weight[nid - 1] = 0.0 if rng.random() < zero_weight_frac else float(rng.uniform(0.5, 8.0))

    expand(0, np.arange(len(d)), 0)
    write_text_vocabulary(path, k, L, parent, is_leaf, desc, weight)
    return dict(nodes=len(parent), words=int(sum(is_leaf)),
                min_leaf_depth=min(dp for dp, lf in zip(depth, is_leaf) if lf))
