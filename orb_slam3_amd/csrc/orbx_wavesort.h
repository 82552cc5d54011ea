// libstdc++'s std::sort, executed by the waves of ONE WORKGROUP on an array in LDS, with the same
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
//    pairs are swapped (all at once here, by one wave) and the returned cut is min(L_K, R_{K-1}).
//  * the two parts of a partitioned range are sorted independently of each other (the depth
//    budget travels with the range), so the order in which pending ranges are taken up does not
//    matter: each round gives every pending range to one wave.
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

// Every thread of the workgroup must enter (whole waves; the call contains workgroup barriers).
// v: n elements in LDS.  segs: 6 * (n / 17 + 1) ints, posL / posR: n ints each, out: n elements (LDS scratch).
// On return v is sorted.
__device__ inline void block_sort(elem_t* v, int n, int* segs, int* posL, int* posR, elem_t* out) {
    __shared__ int s_count[2];      // pending ranges of the current / the next round
    if (n <= 1) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nWaves = blockDim.x >> 6;
    const unsigned long long ltMask = (1ull << lane) - 1ull;
    const int maxSeg = n / 17 + 1;                  // pending ranges are disjoint and longer than 16
    int* const lists[2] = {segs, segs + 3 * maxSeg};   // entries {first, last, depth budget}
    if (threadIdx.x == 0) {
        s_count[0] = 0; s_count[1] = 0;
        if (n > 16) { segs[0] = 0; segs[1] = n; segs[2] = 2 * floor_log2(n); s_count[0] = 1; }
    }
    if (n <= 16)
        for (int i = threadIdx.x; i < n; i += blockDim.x) { posL[i] = 0; posR[i] = n; }
    __syncthreads();

    for (int cur = 0;; cur ^= 1) {
        const int nCur = s_count[cur];
        if (nCur == 0) break;
        const int* in = lists[cur];
        int* nxt = lists[cur ^ 1];
        for (int s = wave; s < nCur; s += nWaves) {
            const int first = in[3 * s], last = in[3 * s + 1];
            int depth = in[3 * s + 2];
            if (depth == 0) {
                // __partial_sort of the whole range = heapsort; already final (posR = -1: leave in place)
                if (lane == 0) heap_sort(v, first, last);
                for (int i = first + lane; i < last; i += 64) { posL[i] = first; posR[i] = -1; }
                continue;
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
            // the position lists live in the scratch slots of the range itself (ranges are disjoint)
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
            // the two parts: still longer than 16 -> next round; otherwise final up to a stable sort of itself
            int slot = 0;
            const int nBig = (cut - first > 16) + (last - cut > 16);
            if (lane == 0 && nBig) slot = atomicAdd(&s_count[cur ^ 1], nBig);
            slot = __builtin_amdgcn_readfirstlane(slot);
#pragma unroll
            for (int part = 0; part < 2; ++part) {
                const int a = part ? cut : first, b = part ? last : cut;
                if (b - a > 16) {
                    if (lane == 0) { nxt[3 * slot] = a; nxt[3 * slot + 1] = b; nxt[3 * slot + 2] = depth; }
                    ++slot;
                } else {
                    for (int i = a + lane; i < b; i += 64) { posL[i] = a; posR[i] = b; }
                }
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) s_count[cur] = 0;      // this round's list becomes the list after next
        __syncthreads();
    }

    // final pass: stable sort of every range = rank inside the range by (key, position)
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
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
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += blockDim.x) v[i] = out[i];
    __syncthreads();
}

}  // namespace orbx_sort
#endif
