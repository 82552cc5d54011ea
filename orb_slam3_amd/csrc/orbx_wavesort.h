// libstdc++'s std::sort, executed by ONE WAVE (64 lanes) on an array in LDS, with the same
// resulting permutation as the sequential algorithm (orbx_introsort.h restates that one
// literally; this file reproduces its RESULT with data-parallel steps).
//
// Why the result can be computed in parallel although the algorithm is sequential:
//
//  * __unguarded_partition(first, last, pivot) alternates "scan up to the next element that
//    is not less than the pivot" / "scan down to the next element the pivot is not less than"
//    and swaps the two until the scans meet.  With L_0 < L_1 < ... the positions whose element
//    is not less than the pivot and R_0 > R_1 > ... the positions whose element is not greater
//    (the pivot's own slot just below the range is the last R), the k-th iteration stops at
//    exactly (L_k, R_k) as long as L_k < R_k: swapped slots are never revisited before the scans
//    meet, because the up-scan only ever runs into slots > L_k and the swapped ones among them
//    (R_0..R_k) lie at or beyond R_k, symmetrically for the down-scan.  So K = #{k : L_k < R_k}
//    pairs are swapped (all at once here) and the returned cut is min(L_K, R_{K-1}).
//  * the quicksort phase leaves ranges of at most 16 elements (or heap-sorted ranges); the final
//    insertion sort never moves an element across such a range's boundary (everything to the
//    left is <= it) and insertion sort is stable, so the final pass equals a stable sort of
//    each range on its own.
//
// Keys and payload are as in orbx_introsort.h (ordering on the upper 40 bits).
#ifndef ORBX_WAVESORT_H_
#define ORBX_WAVESORT_H_

#include "orbx_introsort.h"

namespace orbx_sort {

// All 64 lanes of the calling wave must enter together.  v: n elements in LDS.
// stack: 3 * kStackDepth ints (LDS).  posL, posR: n ints each (LDS scratch).
// out: n elements of LDS scratch.  On return v is sorted.
__device__ inline void wave_sort(elem_t* v, int n, int* stack, int* posL, int* posR, elem_t* out) {
    if (n <= 1) return;
    const int lane = threadIdx.x & 63;
    const unsigned long long ltMask = (1ull << lane) - 1ull;
    int* seg_first = stack;
    int* seg_last = stack + kStackDepth;
    int* seg_depth = stack + 2 * kStackDepth;
    int sp = 1;
    if (lane == 0) { seg_first[0] = 0; seg_last[0] = n; seg_depth[0] = 2 * floor_log2(n); }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");

    while (sp > 0) {
        --sp;
        int first = seg_first[sp], last = seg_last[sp], depth = seg_depth[sp];
        bool heap = false;
        while (last - first > 16) {
            if (depth == 0) {
                if (lane == 0) heap_sort(v, first, last);
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                heap = true;
                break;
            }
            --depth;
            const int mid = first + (last - first) / 2;
            // __move_median_to_first(first, first + 1, mid, last - 1): every lane evaluates the
            // same comparisons on the same values, lane 0 performs the swap
            {
                const int a = first + 1, b = mid, c = last - 1;
                const elem_t va = v[a], vb = v[b], vc = v[c];
                int pick;
                if (less(va, vb)) pick = less(vb, vc) ? b : (less(va, vc) ? c : a);
                else pick = less(va, vc) ? a : (less(vb, vc) ? c : b);
                if (lane == 0) swap_at(v, first, pick);
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            }
            // __unguarded_partition(first + 1, last, pivot = first)
            const int lo = first + 1, hi = last;
            const elem_t pv = v[first];
            // position lists live in the slots of the range itself: the slots of finished ranges
            // already hold their range marks
            int* const pL = posL + lo;
            int* const pR = posR + lo;
            int nL = 0, nR = 0;
            for (int base = lo; base < hi; base += 64) {
                const int i = base + lane;
                bool isL = false, isR = false;
                if (i < hi) {
                    const elem_t e = v[i];
                    isL = !less(e, pv);
                    isR = !less(pv, e);
                }
                const unsigned long long bL = __ballot(isL), bR = __ballot(isR);
                if (isL) pL[nL + __popcll(bL & ltMask)] = i;
                if (isR) pR[nR + __popcll(bR & ltMask)] = i;        // ascending; R_k = pR[nR - 1 - k]
                nL += __popcll(bL);
                nR += __popcll(bR);
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            // K = number of k with L_k < R_k (true for a prefix of k); R_k for k >= nR is the pivot slot
            int K = 0;
            const int nPairs = min(nL, nR);
            for (int base = 0; base < nPairs; base += 64) {
                const int k = base + lane;
                const bool ok = (k < nPairs) && (pL[k] < pR[nR - 1 - k]);
                K += __popcll(__ballot(ok));
            }
            for (int base = 0; base < K; base += 64) {
                const int k = base + lane;
                if (k < K) {
                    const int a = pL[k], b = pR[nR - 1 - k];
                    const elem_t ea = v[a], eb = v[b];
                    v[a] = eb; v[b] = ea;
                }
            }
            const int LK = (K < nL) ? pL[K] : 0x7FFFFFFF;
            const int RKm1 = (K > 0) ? pR[nR - K] : 0x7FFFFFFF;
            const int cut = min(LK, RKm1);
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            if (lane == 0) { seg_first[sp] = cut; seg_last[sp] = last; seg_depth[sp] = depth; }
            ++sp;
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            last = cut;
        }
        // [first, last) is final up to a stable sort of itself; a heap-sorted range is already in
        // its final order (posR = -1 marks "leave in place")
        for (int i = first + lane; i < last; i += 64) { posL[i] = first; posR[i] = heap ? -1 : last; }
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");

    // final pass: stable sort of every range = rank inside the range by (key, position)
    for (int base = 0; base < n; base += 64) {
        const int i = base + lane;
        if (i < n) {
            const elem_t e = v[i];
            int rank = i;
            const int s = posL[i], t = posR[i];
            if (t >= 0) {
                rank = s;
                for (int j = s; j < t; ++j) {
                    const elem_t o = v[j];
                    rank += (less(o, e) || (!less(e, o) && j < i)) ? 1 : 0;
                }
            }
            out[rank] = e;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    for (int i = lane; i < n; i += 64) v[i] = out[i];
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
}

}  // namespace orbx_sort
#endif
