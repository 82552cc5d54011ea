"""The device-side restatement of libstdc++'s std::sort must produce the same permutation as
std::sort itself for comparators with many ties (ORBextractor.cc:805 sorts nodes that tie)."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SRC = textwrap.dedent(r'''
    #include "orb_slam3_amd/csrc/orbx_introsort.h"
    #include <algorithm>
    #include <cstdio>
    #include <random>
    #include <vector>
    typedef unsigned long long u64;
    static bool cmp(const u64& a, const u64& b) { return orbx_sort::less(a, b); }

    // Scalar statement of the data-parallel formulation the device sort uses
    // (orbx_wavesort.h): every partition swaps all its (L_k, R_k) pairs at once and returns
    // min(L_K, R_{K-1}); the final pass is a stable sort of each leftover range.
    static void par_sort(std::vector<u64>& v) {
        using namespace orbx_sort;
        const int n = (int)v.size();
        if (n <= 1) return;
        struct Seg { int first, last, depth; };
        std::vector<Seg> st{{0, n, 2 * floor_log2(n)}};
        std::vector<int> leafS(n), leafE(n);
        while (!st.empty()) {
            Seg s = st.back(); st.pop_back();
            int first = s.first, last = s.last, depth = s.depth;
            bool heap = false;
            while (last - first > 16) {
                if (depth == 0) { heap_sort(v.data(), first, last); heap = true; break; }
                --depth;
                int mid = first + (last - first) / 2;
                move_median_to_first(v.data(), first, first + 1, mid, last - 1);
                const u64 pv = v[first];
                std::vector<int> L, R;
                for (int i = first + 1; i < last; ++i) {
                    if (!less(v[i], pv)) L.push_back(i);
                    if (!less(pv, v[i])) R.push_back(i);
                }
                const int nL = (int)L.size(), nR = (int)R.size();
                int K = 0;
                for (int k = 0; k < std::min(nL, nR); ++k) K += (L[k] < R[nR - 1 - k]) ? 1 : 0;
                for (int k = 0; k < K; ++k) std::swap(v[L[k]], v[R[nR - 1 - k]]);
                const int LK = K < nL ? L[K] : 0x7FFFFFFF, RK = K > 0 ? R[nR - K] : 0x7FFFFFFF;
                const int cut = std::min(LK, RK);
                st.push_back({cut, last, depth});
                last = cut;
            }
            for (int i = first; i < last; ++i) { leafS[i] = first; leafE[i] = heap ? -1 : last; }
        }
        std::vector<u64> out(n);
        for (int i = 0; i < n; ++i) {
            int rank = i;
            if (leafE[i] >= 0) {
                rank = leafS[i];
                for (int j = leafS[i]; j < leafE[i]; ++j)
                    rank += (less(v[j], v[i]) || (!less(v[i], v[j]) && j < i)) ? 1 : 0;
            }
            out[rank] = v[i];
        }
        v = out;
    }
    int main() {
        std::mt19937_64 rng(7);
        long bad = 0, cases = 0;
        int stack[3 * orbx_sort::kStackDepth];
        for (int n = 0; n <= 2600; n += (n < 70 ? 1 : 31)) {
            for (int mode = 0; mode < 8; ++mode) {
                std::vector<u64> a(n);
                for (int i = 0; i < n; ++i) {
                    uint32_t cnt, ulx;
                    switch (mode) {
                        case 0: cnt = 2 + rng() % 2; ulx = 16 * (rng() % 4); break;      // heavy ties
                        case 1: cnt = 2 + rng() % 6; ulx = rng() % 40; break;
                        case 2: cnt = 2 + rng() % 50; ulx = rng() % 700; break;
                        case 3: cnt = rng() % 100000; ulx = rng() % 30000; break;         // large counts
                        case 4: cnt = i; ulx = 0; break;                                  // sorted
                        case 5: cnt = n - i; ulx = 0; break;                              // reversed
                        case 6: cnt = (i < n / 2) ? i : (n - i); ulx = 3; break;          // organ pipe
                        default: cnt = (i % 2) ? i : 0; ulx = i % 3; break;
                    }
                    a[i] = orbx_sort::make_elem(cnt, ulx, (uint32_t)i);
                }
                std::vector<u64> b = a, c = a;
                std::sort(a.begin(), a.end(), cmp);
                orbx_sort::sort(b.data(), n, stack);
                par_sort(c);
                ++cases;
                if (a != b) { ++bad; if (bad < 5) printf("MISMATCH n=%d mode=%d\n", n, mode); }
                if (a != c) { ++bad; if (bad < 5) printf("PARALLEL-FORM MISMATCH n=%d mode=%d\n", n, mode); }
            }
        }
        // heapsort fallback (taken when the quicksort depth limit is exhausted)
        for (int n = 2; n < 400; ++n) {
            std::vector<u64> a(n);
            for (int i = 0; i < n; ++i) a[i] = orbx_sort::make_elem(rng() % 5, rng() % 3, i);
            std::vector<u64> b = a;
            std::partial_sort(a.begin(), a.end(), a.end(), cmp);
            orbx_sort::heap_sort(b.data(), 0, n);
            ++cases;
            if (a != b) { ++bad; printf("HEAP MISMATCH n=%d\n", n); }
        }
        printf("cases=%ld bad=%ld\n", cases, bad);
        return bad ? 1 : 0;
    }
''')


def test_introsort_matches_libstdcxx(tmp_path):
    src = tmp_path / "t.cpp"
    src.write_text(SRC)
    exe = tmp_path / "t"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", ROOT, str(src), "-o", str(exe)])
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    sys.stdout.write(out.stdout)
    assert out.returncode == 0, out.stdout
    assert "bad=0" in out.stdout
