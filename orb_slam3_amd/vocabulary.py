"""Bag-of-words transform over the C ABI: the tree descent of DBoW2's ORBVocabulary
(Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1112-1243) that Frame::ComputeBoW /
KeyFrame::ComputeBoW run between extraction and the BoW matchers."""
from __future__ import annotations

import ctypes as C

import numpy as np

from ._lib import check, lib, ptr


class Vocabulary:
    """A k-ary vocabulary tree in flattened form.

    child_off / child_idx: children of node i are child_idx[child_off[i]:child_off[i + 1]] in the order of
    DBoW2's Node::children (node 0 = root); node_desc [n, 32] uint8; word_id [n] (-1 for inner nodes);
    weight [n] float64 (idf of the leaves); depth = the vocabulary's L."""

    def __init__(self, child_off, child_idx, node_desc, word_id, weight, depth: int, device: int = 0):
        self.child_off = np.ascontiguousarray(child_off, np.int32)
        self.child_idx = np.ascontiguousarray(child_idx, np.int32)
        self.node_desc = np.ascontiguousarray(node_desc, np.uint8)
        self.word_id = np.ascontiguousarray(word_id, np.int32)
        self.weight = np.ascontiguousarray(weight, np.float64)
        self.depth = int(depth)
        n = len(self.word_id)
        assert self.node_desc.shape == (n, 32) and len(self.child_off) == n + 1
        h = C.c_void_p()
        check(lib().orbx_vocabulary_create(n, ptr(self.child_off), ptr(self.child_idx), ptr(self.node_desc),
                                           ptr(self.word_id), ptr(self.weight), self.depth, device, C.byref(h)))
        self._h = h

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                lib().orbx_vocabulary_destroy(h)
            except Exception:
                pass

    def transform_features(self, desc, levelsup: int = 4):
        """(word id, weight, feature-vector node id) per descriptor."""
        desc = np.ascontiguousarray(desc, np.uint8)
        n = len(desc)
        word = np.zeros(n, np.int32)
        w = np.zeros(n, np.float64)
        nid = np.zeros(n, np.int32)
        check(lib().orbx_bow_transform(self._h, ptr(desc), n, levelsup, ptr(word), ptr(w), ptr(nid)))
        return word, w, nid

    def transform(self, desc, levelsup: int = 4, l1_normalize: bool = True):
        """transform(features, BowVector, FeatureVector, levelsup): returns ((words, values),
        (nodes, off, idx)); (idx, off) is the CSR the BoW matchers take once restricted to shared nodes."""
        desc = np.ascontiguousarray(desc, np.uint8)
        n = len(desc)
        bw, bv = np.zeros(max(n, 1), np.int32), np.zeros(max(n, 1), np.float64)
        fn, fo, fi = np.zeros(max(n, 1), np.int32), np.zeros(n + 1, np.int32), np.zeros(max(n, 1), np.int32)
        nb, nn = C.c_int(), C.c_int()
        check(lib().orbx_bow_vectors(self._h, ptr(desc), n, levelsup, int(l1_normalize), ptr(bw), ptr(bv), C.byref(nb),
                                     ptr(fn), ptr(fo), ptr(fi), C.byref(nn)))
        return (bw[:nb.value].copy(), bv[:nb.value].copy()), (fn[:nn.value].copy(), fo[:nn.value + 1].copy(),
                                                                fi[:fo[nn.value]].copy())


def shared_nodes(fv1, fv2):
    """Restricts two feature vectors (nodes, off, idx) to the vocabulary nodes both contain, in ascending
    node order (what the reference's lower_bound walk visits): returns ((idx1, off1), (idx2, off2))."""
    n1, o1, i1 = fv1
    n2, o2, i2 = fv2
    common, a, b = np.intersect1d(n1, n2, assume_unique=True, return_indices=True)

    def cut(off, idx, sel):
        parts = [idx[off[s]:off[s + 1]] for s in sel]
        o = np.zeros(len(sel) + 1, np.int32)
        if parts:
            o[1:] = np.cumsum([len(p) for p in parts])
        return (np.concatenate(parts).astype(np.int32) if parts else np.zeros(0, np.int32)), o
    return cut(o1, i1, a), cut(o2, i2, b)
