// Literal re-statement of libstdc++'s std::sort (introsort: median-of-3 quicksort down to
// partitions of 16, depth limit 2*floor(log2 n) with heapsort fallback, then one final
// insertion sort), usable from host and device code.
//
// Why: the reference sorts the expandable quadtree nodes with std::sort and a comparator
// under which many elements tie (ORBextractor.cc:625-646, :805).  Which of the tied nodes
// is split first depends on the exact permutation the library's algorithm produces, so the
// device must reproduce that algorithm step for step, not merely "a sort".
//
// Elements are 64-bit words; the ordering looks only at the upper 40 bits
// (key = key count << 16 | UL.x), the lower 24 bits are payload (the node's list position)
// that travels with the element.
#ifndef ORBX_INTROSORT_H_
#define ORBX_INTROSORT_H_

#include <stdint.h>

#if defined(__HIPCC__)
#define ORBX_HD __host__ __device__
#else
#define ORBX_HD
#endif

namespace orbx_sort {

typedef unsigned long long elem_t;

constexpr int kPayloadBits = 24;
ORBX_HD inline elem_t make_elem(uint32_t count, uint32_t ulx, uint32_t payload) {
    return ((elem_t)count << 40) | ((elem_t)(ulx & 0xFFFFu) << kPayloadBits) | (payload & 0xFFFFFFu);
}
ORBX_HD inline uint32_t payload_of(elem_t e) { return (uint32_t)(e & 0xFFFFFFu); }
ORBX_HD inline bool less(elem_t a, elem_t b) { return (a >> kPayloadBits) < (b >> kPayloadBits); }
ORBX_HD inline void swap_at(elem_t* v, int i, int j) { elem_t t = v[i]; v[i] = v[j]; v[j] = t; }

ORBX_HD inline void unguarded_linear_insert(elem_t* v, int last) {
    elem_t val = v[last];
    int next = last - 1;
    while (less(val, v[next])) {
        v[last] = v[next];
        last = next;
        --next;
    }
    v[last] = val;
}

ORBX_HD inline void insertion_sort(elem_t* v, int first, int last) {
    if (first == last) return;
    for (int i = first + 1; i != last; ++i) {
        if (less(v[i], v[first])) {
            elem_t val = v[i];
            for (int k = i; k > first; --k) v[k] = v[k - 1];
            v[first] = val;
        } else {
            unguarded_linear_insert(v, i);
        }
    }
}

ORBX_HD inline void push_heap(elem_t* v, int first, int hole, int top, elem_t value) {
    int parent = (hole - 1) / 2;
    while (hole > top && less(v[first + parent], value)) {
        v[first + hole] = v[first + parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    v[first + hole] = value;
}

ORBX_HD inline void adjust_heap(elem_t* v, int first, int hole, int len, elem_t value) {
    const int top = hole;
    int child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (less(v[first + child], v[first + child - 1])) --child;
        v[first + hole] = v[first + child];
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        v[first + hole] = v[first + child - 1];
        hole = child - 1;
    }
    push_heap(v, first, hole, top, value);
}

// std::partial_sort(first, last, last): make_heap + sort_heap.
ORBX_HD inline void heap_sort(elem_t* v, int first, int last) {
    int len = last - first;
    if (len >= 2) {
        int parent = (len - 2) / 2;
        while (true) {
            elem_t value = v[first + parent];
            adjust_heap(v, first, parent, len, value);
            if (parent == 0) break;
            --parent;
        }
    }
    while (last - first > 1) {
        --last;
        elem_t value = v[last];
        v[last] = v[first];
        adjust_heap(v, first, 0, last - first, value);
    }
}

ORBX_HD inline void move_median_to_first(elem_t* v, int result, int a, int b, int c) {
    if (less(v[a], v[b])) {
        if (less(v[b], v[c])) swap_at(v, result, b);
        else if (less(v[a], v[c])) swap_at(v, result, c);
        else swap_at(v, result, a);
    } else if (less(v[a], v[c])) swap_at(v, result, a);
    else if (less(v[b], v[c])) swap_at(v, result, c);
    else swap_at(v, result, b);
}

ORBX_HD inline int unguarded_partition(elem_t* v, int first, int last, int pivot) {
    while (true) {
        while (less(v[first], v[pivot])) ++first;
        --last;
        while (less(v[pivot], v[last])) --last;
        if (!(first < last)) return first;
        swap_at(v, first, last);
        ++first;
    }
}

ORBX_HD inline int floor_log2(int n) {
    int l = 0;
    while (n > 1) { n >>= 1; ++l; }
    return l;
}

// std::sort(v, v + n, less).  __introsort_loop recurses into the right part and iterates
// on the left part; the two parts are disjoint, so the order in which they are finished
// does not change the result and the recursion becomes a stack of pending segments.
// `stack` holds 3 * kStackDepth ints of working storage (LDS on the device).
constexpr int kStackDepth = 72;
ORBX_HD inline void sort(elem_t* v, int n, int* stack) {
    if (n <= 0) return;
    const int kThreshold = 16;
    int* seg_first = stack;
    int* seg_last = stack + kStackDepth;
    int* seg_depth = stack + 2 * kStackDepth;
    int sp = 0;
    seg_first[0] = 0; seg_last[0] = n; seg_depth[0] = 2 * floor_log2(n);
    sp = 1;
    while (sp > 0) {
        --sp;
        int first = seg_first[sp], last = seg_last[sp], depth = seg_depth[sp];
        while (last - first > kThreshold) {
            if (depth == 0) {
                heap_sort(v, first, last);
                break;
            }
            --depth;
            int mid = first + (last - first) / 2;
            move_median_to_first(v, first, first + 1, mid, last - 1);
            int cut = unguarded_partition(v, first + 1, last, first);
            seg_first[sp] = cut; seg_last[sp] = last; seg_depth[sp] = depth;
            ++sp;
            last = cut;
        }
    }
    if (n > kThreshold) {
        insertion_sort(v, 0, kThreshold);
        for (int i = kThreshold; i != n; ++i) unguarded_linear_insert(v, i);
    } else {
        insertion_sort(v, 0, n);
    }
}

}  // namespace orbx_sort
#endif
