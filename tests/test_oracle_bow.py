"""Known answers for the oracle's vocabulary descent (DBoW2 transform): first-minimum tie rule, the
feature-vector node level, early leaves."""
import numpy as np


def test_descent_by_hand(oracle):
    # root 0 -> children [1, 2]; 1 -> [3, 4]; 2 -> [5, 6]; depth 2, leaves 3..6 = words 0..3
    child_off = np.array([0, 2, 4, 6, 6, 6, 6, 6], np.int32)
    child_idx = np.array([1, 2, 3, 4, 5, 6], np.int32)
    nd = np.zeros((7, 32), np.uint8)
    nd[2] = 0xFF                      # node 2 is all ones
    nd[4, 0] = 0x0F                   # 4 bits from zero
    nd[5] = 0xFF
    nd[6] = 0xFF; nd[6, 0] = 0xF0
    word = np.array([-1, -1, -1, 0, 1, 2, 3], np.int32)
    w = np.array([0, 0, 0, 1.5, 2.5, 3.5, 4.5])
    q = np.zeros((3, 32), np.uint8)
    q[1] = 0xFF
    q[2] = 0xFF; q[2, 0] = 0xF0
    wo, we, ni = oracle.bow_transform(child_off, child_idx, nd, word, w, 2, q, 1)
    assert list(wo) == [0, 2, 3] and list(we) == [1.5, 3.5, 4.5]
    assert list(ni) == [1, 2, 2]      # node at level depth - levelsup = 1
    wo, we, ni = oracle.bow_transform(child_off, child_idx, nd, word, w, 2, q, 0)
    assert list(ni) == [3, 5, 6]      # levelsup 0: the leaf itself
    wo, we, ni = oracle.bow_transform(child_off, child_idx, nd, word, w, 2, q, 4)
    assert list(ni) == [0, 0, 0]      # above the root: node 0


def test_first_minimum_wins(oracle):
    # three identical children: the first in the child list is taken, whatever its id
    child_off = np.array([0, 3, 3, 3, 3], np.int32)
    child_idx = np.array([3, 1, 2], np.int32)
    nd = np.zeros((4, 32), np.uint8)
    word = np.array([-1, 0, 1, 2], np.int32)
    w = np.array([0, 1.0, 2.0, 3.0])
    wo, we, ni = oracle.bow_transform(child_off, child_idx, nd, word, w, 1, np.zeros((1, 32), np.uint8), 0)
    assert list(wo) == [2] and list(ni) == [3]


def test_distinctive_descriptor_by_hand(oracle):
    # three observations with 0, 4 and 8 bits set: rows of the distance matrix [0 4 8], [4 0 4], [8 4 0];
    # medians (element 1 of the sorted row) 4, 4, 4 -> all tie, row 0 wins; with a fourth far one the middle wins
    d = np.zeros((3, 32), np.uint8)
    d[1, 0] = 0x0F
    d[2, 0] = 0xFF
    bi, bm = oracle.distinctive_descriptors(d, np.array([0, 3], np.int32))
    assert list(bi) == [0] and list(bm) == [4]
    d4 = np.zeros((4, 32), np.uint8)
    d4[1, 0] = 0x0F
    d4[2, 0] = 0xFF
    d4[3, :4] = 0xFF                   # 32 bits
    # rows sorted: [0 4 8 32], [0 4 4 28], [0 4 8 24], [0 24 28 32]; element floor(0.5 * 3) = 1 -> 4, 4, 4, 24
    bi, bm = oracle.distinctive_descriptors(d4, np.array([0, 4], np.int32))
    assert list(bi) == [0] and list(bm) == [4]
    # five observations: element 2 of the sorted rows
    d5 = np.zeros((5, 32), np.uint8)
    for i, nb in enumerate([0, 2, 4, 6, 30]):
        for b in range(nb):
            d5[i, b >> 3] |= np.uint8(1 << (b & 7))
    # distances |a - b| in set bits: rows 1 and 2 both have the sorted row [0 2 2 4 x] -> median 2; row 1 comes first
    bi, bm = oracle.distinctive_descriptors(d5, np.array([0, 5], np.int32))
    assert list(bi) == [1] and list(bm) == [2]
