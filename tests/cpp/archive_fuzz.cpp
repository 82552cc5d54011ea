// Sanitizer harness for the saved-map record parser (orb_slam3_amd/csrc/orbx_archive.hip, host code that reads
// untrusted file bytes).  Built by tests/test_sanitizers.py with g++ -fsanitize=address,undefined together with the
// parser's source; every image handed to the parser is an exact-size heap copy, so a read past `len` is an ASan
// report.  Feeds: every truncation of a valid text and binary image, then N seeded mutations (byte flips, digit and
// separator substitutions, inserted / deleted runs, spliced headers) of each.  Passes when nothing crashes, no
// sanitizer fires and every successful read is consistent (sizes within the image, payload copies complete).
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "../../include/orbx.h"

void orbx_set_error(const char*, ...) {}      // the library's error-text sink (C++ linkage, orbx_common.h)

static std::vector<uint8_t> make_image(bool text, std::mt19937& rng) {
    std::vector<uint8_t> img;
    auto append = [&](const std::vector<uint8_t>& rec) { img.insert(img.end(), rec.begin(), rec.end()); };
    auto junk = [&](int n) { for (int i = 0; i < n; ++i) img.push_back(text ? (uint8_t)("0123456789 abc\n"[rng() % 15]) : (uint8_t)rng()); };
    junk(40);
    for (int m = 0; m < 3; ++m) {
        const int rows = 1 + (int)(rng() % 40);
        std::vector<uint8_t> d((size_t)rows * 32);
        for (auto& b : d) b = (uint8_t)rng();
        size_t need = 0;
        orbx_archive_write_matrix(d.data(), rows, 32, 0, text, nullptr, 0, &need);
        std::vector<uint8_t> rec(need);
        orbx_archive_write_matrix(d.data(), rows, 32, 0, text, rec.data(), rec.size(), &need);
        append(rec);
        if (text) img.push_back(' ');
        std::vector<orbx_keypoint> k((size_t)rows);
        for (auto& p : k) { p.x = (float)(rng() % 700); p.y = (float)(rng() % 400); p.size = 31.f; p.angle = (float)(rng() % 360); p.response = 20.f; p.octave = (int)(rng() % 8); p.class_id = -1; }
        orbx_archive_write_keypoints(k.data(), rows, text, nullptr, 0, &need);
        rec.assign(need, 0);
        orbx_archive_write_keypoints(k.data(), rows, text, rec.data(), rec.size(), &need);
        append(rec);
        if (text) img.push_back(' ');
        junk(1 + (int)(rng() % 30));
    }
    return img;
}

static long g_calls = 0, g_ok = 0;

// Every entry point on one image held in an exact-size heap block.
static void exercise(const std::vector<uint8_t>& image, bool text, std::mt19937& rng) {
    const size_t len = image.size();
    uint8_t* buf = (uint8_t*)malloc(len ? len : 1);
    if (len) memcpy(buf, image.data(), len);
    // records at a few offsets (the start, and random ones)
    for (int t = 0; t < 4; ++t) {
        const size_t off = t == 0 ? 0 : (len ? rng() % len : 0);
        size_t used = 0; int rows = 0, cols = 0, type = 0, n = 0;
        ++g_calls;
        if (orbx_archive_read_matrix(buf + off, len - off, text, &used, &rows, &cols, &type, nullptr, 0) == ORBX_OK) {
            if (used > len - off) { fprintf(stderr, "read_matrix consumed %zu of %zu\n", used, len - off); abort(); }
            const size_t bytes = (size_t)rows * (size_t)cols;           // at least one byte per element
            if (bytes <= (1u << 24)) {
                std::vector<uint8_t> out(bytes * 8 + 1);
                if (orbx_archive_read_matrix(buf + off, len - off, text, &used, &rows, &cols, &type, out.data(), out.size() - 1) == ORBX_OK) ++g_ok;
            }
        }
        ++g_calls;
        if (orbx_archive_read_keypoints(buf + off, len - off, text, &used, nullptr, 0, &n) == ORBX_OK) {
            if (used > len - off || n < 0) abort();
            if (n <= (1 << 20)) {
                std::vector<orbx_keypoint> k((size_t)n + 1);
                if (orbx_archive_read_keypoints(buf + off, len - off, text, &used, k.data(), n, &n) == ORBX_OK) ++g_ok;
            }
        }
    }
    for (int min_rows : {1, 8}) {
        int found = 0; long long total = 0;
        ++g_calls;
        if (orbx_archive_scan_descriptors(buf, len, text, min_rows, nullptr, nullptr, 0, &found, &total) != ORBX_OK) continue;
        std::vector<size_t> offs((size_t)found + 1);
        std::vector<int> rows((size_t)found + 1);
        if (orbx_archive_scan_descriptors(buf, len, text, min_rows, offs.data(), rows.data(), found, &found, &total) != ORBX_OK) abort();
        long long sum = 0;
        for (int i = 0; i < found; ++i) {
            if (offs[i] > len || rows[i] < min_rows) abort();
            sum += rows[i];
            std::vector<uint8_t> out((size_t)rows[i] * 32);
            ++g_calls;
            if (orbx_archive_copy_descriptors(buf, len, text, offs[i], rows[i], out.data()) != ORBX_OK) { fprintf(stderr, "a scanned matrix does not copy\n"); abort(); }
            ++g_ok;
        }
        if (sum != total) abort();
    }
    // a payload offset / row count the scan never reported
    {
        std::vector<uint8_t> out(64 * 32);
        ++g_calls;
        orbx_archive_copy_descriptors(buf, len, text, len ? rng() % (len + 1) : 0, (int)(rng() % 64), out.data());
    }
    free(buf);
}

int main(int argc, char** argv) {
    const int n_mut = argc > 1 ? atoi(argv[1]) : 10000;
    for (int text = 0; text < 2; ++text) {
        std::mt19937 rng(1234 + text);
        const std::vector<uint8_t> base = make_image(text != 0, rng);
        exercise(base, text != 0, rng);
        for (size_t cut = 0; cut < base.size(); cut += (base.size() > 4000 ? 7 : 1))           // truncations
            exercise(std::vector<uint8_t>(base.begin(), base.begin() + cut), text != 0, rng);
        for (int m = 0; m < n_mut; ++m) {
            std::vector<uint8_t> img = base;
            const int edits = 1 + (int)(rng() % 6);
            for (int e = 0; e < edits && !img.empty(); ++e) {
                const size_t at = rng() % img.size();
                switch (rng() % 6) {
                    case 0: img[at] ^= (uint8_t)(1u << (rng() % 8)); break;
                    case 1: img[at] = text ? (uint8_t)("0123456789 -.e+\n"[rng() % 16]) : (uint8_t)rng(); break;
                    case 2: img.erase(img.begin() + at, img.begin() + std::min(img.size(), at + 1 + rng() % 40)); break;
                    case 3: img.insert(img.begin() + at, 1 + rng() % 40, text ? (uint8_t)('0' + rng() % 10) : (uint8_t)rng()); break;
                    case 4: {                                                          // a header that promises far too much
                        if (text) { const char* h = " 32 2000000000 0 1 "; img.insert(img.begin() + at, h, h + strlen(h)); }
                        else { const int32_t h[3] = {32, (int32_t)(rng() | 0x40000000u) & 0x7FFFFFFF, 0}; img.insert(img.begin() + at, (const uint8_t*)h, (const uint8_t*)h + 12); img.insert(img.begin() + at + 12, (uint8_t)1); }
                        break;
                    }
                    default: img.resize(at); break;
                }
            }
            exercise(img, text != 0, rng);
        }
    }
    printf("archive_fuzz ok: %ld parser calls, %ld successful reads\n", g_calls, g_ok);
    return 0;
}
