"""GPU parity of the bag-of-words transform against the oracle's restatement of DBoW2's tree descent."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from orb_slam3_amd import synth


def _gpu():
    import orb_slam3_amd as A
    assert A.lib().orbx_device_count() >= 1, "no HIP device: the GPU tests need an MI355X"
    return A


def _features(voc, n, seed):
    """Descriptors near random leaves (plus some far from everything)."""
    rng = np.random.default_rng(seed)
    child_off, child_idx, nd, word, w, depth = voc
    leaves = np.nonzero(word >= 0)[0]
    d = nd[rng.choice(leaves, n)].copy()
    for i in range(n):
        for b in rng.integers(0, 256, int(rng.integers(0, 40))):
            d[i, b >> 3] ^= np.uint8(1 << (b & 7))
    d[::17] = synth.random_descriptors(len(d[::17]), seed + 1)
    return d


@pytest.mark.parametrize("k,depth,prune,levelsup,seed", [
    (10, 3, 0.0, 1, 1),      # balanced, like ORBvoc (k = 10) but shallower
    (10, 4, 0.0, 4, 2),      # levelsup = depth: the feature-vector node is the root
    (10, 4, 0.0, 2, 3),
    (6, 5, 0.25, 3, 4),      # early leaves
    (70, 2, 0.0, 1, 5),      # more children than lanes
    (2, 8, 0.1, 4, 6),       # deep binary tree, many near-ties
])
def test_bow_transform_matches_oracle(oracle, k, depth, prune, levelsup, seed):
    A = _gpu()
    voc = synth.vocabulary(k, depth, seed, prune)
    desc = _features(voc, 1500, 100 + seed)
    wo, we, no = oracle.bow_transform(*voc[:5], voc[5], desc, levelsup)
    V = A.Vocabulary(*voc)
    wg, weg, ng = V.transform_features(desc, levelsup)
    assert np.array_equal(wg, wo) and np.array_equal(weg, we) and np.array_equal(ng, no)
    assert len(np.unique(wg)) > 20

    # BowVector / FeatureVector as the reference's std::maps hold them
    (bw, bv), (fn, fo, fi) = V.transform(desc, levelsup)
    bow, fv = {}, {}
    for i in range(len(desc)):
        if we[i] > 0:
            bow[int(wo[i])] = bow.get(int(wo[i]), 0.0) + we[i]
            fv.setdefault(int(no[i]), []).append(i)
    words = sorted(bow)
    vals = np.array([bow[x] for x in words])
    norm = 0.0
    for v in vals:
        norm += abs(v)
    assert list(bw) == words and np.array_equal(bv, vals / norm)
    assert list(fn) == sorted(fv)
    for b, node in enumerate(fn):
        assert list(fi[fo[b]:fo[b + 1]]) == fv[int(node)]


def test_bow_feeds_the_bow_matcher(oracle):
    """ComputeBoW on two descriptor sets, then SearchByBoW(KeyFrame, KeyFrame) over the shared nodes."""
    A = _gpu()
    voc = synth.vocabulary(10, 4, 11)
    V = A.Vocabulary(*voc)
    rng = np.random.default_rng(12)
    d1 = _features(voc, 900, 13)
    perm = rng.permutation(900)
    d2 = d1[perm].copy()
    for i in range(len(d2)):
        for b in rng.integers(0, 256, int(rng.integers(0, 12))):
            d2[i, b >> 3] ^= np.uint8(1 << (b & 7))
    _, fv1 = V.transform(d1, 2)
    _, fv2 = V.transform(d2, 2)
    (i1, o1), (i2, o2) = A.shared_nodes(fv1, fv2)
    ang = np.zeros(900, np.float32)
    ones = np.ones(900, np.uint8)
    m12_o, n_o = oracle.search_by_bow_keyframes(d1, ang, ones, i1, o1, d2, ang, ones, i2, o2, 0.9, False)
    m12, n = A.ORBmatcher(0.9, False).SearchByBoWKeyFrames(d1, ang, ones, (i1, o1), d2, ang, ones, (i2, o2))
    assert n == n_o and np.array_equal(m12, m12_o)
    good = m12 >= 0
    assert good.sum() > 300 and (perm[m12[good]] == np.nonzero(good)[0]).mean() > 0.9


def test_distinctive_descriptors(oracle):
    """MapPoint::ComputeDistinctiveDescriptors batched: observation counts 0 .. 384, ties between medians."""
    A = _gpu()
    rng = np.random.default_rng(21)
    counts = [0, 1, 2, 3, 4, 5, 7, 8, 15, 16, 31, 63, 64, 65, 100, 129, 384] + list(rng.integers(2, 30, 400))
    off = np.zeros(len(counts) + 1, np.int32)
    off[1:] = np.cumsum(counts)
    base = synth.random_descriptors(len(counts), 22)
    desc = np.repeat(base, counts, axis=0)
    for i in range(len(desc)):
        for b in rng.integers(0, 256, int(rng.integers(0, 30))):
            desc[i, b >> 3] ^= np.uint8(1 << (b & 7))
    desc[off[10]:off[10] + 8] = desc[off[10]]          # identical observations: all medians tie, the first row wins
    bi_o, bm_o = oracle.distinctive_descriptors(desc, off)
    bi, bm = A.distinctive_descriptors(desc, off)
    assert np.array_equal(bi, bi_o) and np.array_equal(bm, bm_o)
    assert bi[0] == -1 and bi[1] == 0 and bm[1] == 0


def test_bow_and_distinctive_random_shapes(oracle):
    """Vocabulary shapes (branching factor, depth, pruned subtrees), feature counts and levelsup drawn at random;
    random observation-count profiles for the distinctive-descriptor choice."""
    A = _gpu()
    rng = np.random.default_rng(616)
    for case in range(10):
        k = int(rng.choice([2, 3, 5, 9, 10, 17, 64, 65, 100]))
        depth = int(rng.integers(1, 9))
        while k ** depth > 40000:
            depth -= 1
        prune = float(rng.choice([0.0, 0.1, 0.3]))
        levelsup = int(rng.integers(0, depth + 2))
        n = int(rng.integers(1, 3000))
        voc = synth.vocabulary(k, depth, 40 + case, prune)
        desc = _features(voc, n, 140 + case)
        wo, we, no = oracle.bow_transform(*voc[:5], voc[5], desc, levelsup)
        wg, weg, ng = A.Vocabulary(*voc).transform_features(desc, levelsup)
        assert np.array_equal(wg, wo) and np.array_equal(weg, we) and np.array_equal(ng, no), \
            f"case {case}: k={k} depth={depth} prune={prune} levelsup={levelsup} n={n}"
    for case in range(4):
        counts = list(rng.integers(0, [6, 40, 130, 385][case], 300))
        off = np.zeros(len(counts) + 1, np.int32)
        off[1:] = np.cumsum(counts)
        desc = np.repeat(synth.random_descriptors(len(counts), 50 + case), counts, axis=0)
        desc = synth.noisy_copy(rng, desc, [2, 10, 30, 60][case])      # few flips: many equal distances and tied medians
        bi_o, bm_o = oracle.distinctive_descriptors(desc, off)
        bi, bm = A.distinctive_descriptors(desc, off)
        assert np.array_equal(bi, bi_o) and np.array_equal(bm, bm_o), f"distinctive case {case}"
