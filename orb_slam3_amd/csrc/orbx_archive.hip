// Saved-map records: the two layouts through which the reference writes keypoints and descriptors into its Atlas
// archives (boost::serialization text or binary archive, System.cc:1568-1672):
//   serializeMatrix            include/SerializationUtils.h:76-108   cols, rows, type, continuous, then the bytes
//   serializeVectorKeyPoints   include/SerializationUtils.h:125-169  NumEl, then per keypoint angle, response, size,
//                                                                   pt.x, pt.y, class_id, octave
// used by KeyFrame::serialize (include/KeyFrame.h:118-126: mDistCoef, mvKeys, mvKeysUn, mDescriptors) and
// MapPoint::serialize (mDescriptor).  A boost binary archive stores primitives in native layout (int and float 4 bytes
// little endian, bool 1 byte, a make_array of bytes raw); a text archive stores every primitive -- each byte of a
// make_array of unsigned char included -- as a decimal token separated by spaces.
//
// Host code only.  Neither boost nor a saved map exists in the reference tree, so these layouts follow the code of the
// two templates and boost's documented primitive encoding: parity unpinned.  The rest of an archive (class
// versions, object tracking, the other members) is not interpreted; descriptor matrices are found by their header
// signature instead (32 columns, CV_8UC1, continuous), which is what a large-map matcher needs from a saved map.

#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "orbx_common.h"

namespace {
constexpr int kCv8UC1 = 0;

struct Reader {
    const uint8_t* p;
    size_t len, pos;
    bool text;
    bool ok = true;
    template <class T> T raw() {
        T v{};
        if (pos + sizeof(T) > len) { ok = false; return v; }
        memcpy(&v, p + pos, sizeof(T));
        pos += sizeof(T);
        return v;
    }
    // next whitespace-separated token of a text archive
    bool token(const char** b, size_t* n) {
        while (pos < len && (p[pos] == ' ' || p[pos] == '\n' || p[pos] == '\r' || p[pos] == '\t')) ++pos;
        if (pos >= len) { ok = false; return false; }
        const size_t s = pos;
        while (pos < len && !(p[pos] == ' ' || p[pos] == '\n' || p[pos] == '\r' || p[pos] == '\t')) ++pos;
        *b = reinterpret_cast<const char*>(p + s);
        *n = pos - s;
        return true;
    }
    long long int_token() {
        const char* b; size_t n;
        if (!token(&b, &n) || n == 0 || n > 24) { ok = false; return 0; }
        char tmp[32];
        memcpy(tmp, b, n); tmp[n] = 0;
        char* end = nullptr;
        errno = 0;
        const long long v = strtoll(tmp, &end, 10);
        if (errno || end != tmp + n) ok = false;
        return v;
    }
    float float_token() {
        const char* b; size_t n;
        if (!token(&b, &n) || n == 0 || n > 40) { ok = false; return 0.f; }
        char tmp[48];
        memcpy(tmp, b, n); tmp[n] = 0;
        char* end = nullptr;
        const float v = strtof(tmp, &end);
        if (end != tmp + n) ok = false;
        return v;
    }
    int i32() { return text ? (int)int_token() : raw<int32_t>(); }
    float f32() { return text ? float_token() : raw<float>(); }
    bool b8() { return text ? int_token() != 0 : raw<uint8_t>() != 0; }
    uint8_t u8() {
        if (!text) return raw<uint8_t>();
        const long long v = int_token();
        if (v < 0 || v > 255) ok = false;
        return (uint8_t)v;
    }
};

struct Writer {
    uint8_t* p;
    size_t cap, pos;
    bool text;
    void put(const void* src, size_t n) {
        if (p && pos + n <= cap) memcpy(p + pos, src, n);
        pos += n;                                   // keeps counting past the capacity: the caller learns the size
    }
    void i32(int v) {
        if (!text) { const int32_t x = v; put(&x, 4); return; }
        char tmp[16];
        const int n = snprintf(tmp, sizeof(tmp), " %d", v);
        put(tmp, (size_t)n);
    }
    void f32(float v) {
        if (!text) { put(&v, 4); return; }
        char tmp[32];
        const int n = snprintf(tmp, sizeof(tmp), " %.9e", (double)v);     // 9 significant digits + 1: round-trips binary32
        put(tmp, (size_t)n);
    }
    void b8(bool v) {
        if (!text) { const uint8_t x = v ? 1 : 0; put(&x, 1); return; }
        put(v ? " 1" : " 0", 2);
    }
    void bytes(const uint8_t* src, size_t n) {
        if (!text) { put(src, n); return; }
        char tmp[8];
        for (size_t i = 0; i < n; ++i) {
            const int k = snprintf(tmp, sizeof(tmp), " %u", (unsigned)src[i]);
            put(tmp, (size_t)k);
        }
    }
};

size_t elem_size_of_cv_type(int type) {
    static const int depthBytes[8] = {1, 1, 2, 2, 4, 4, 8, 2};   // CV_8U, 8S, 16U, 16S, 32S, 32F, 64F, 16F
    const int depth = type & 7, channels = (type >> 3) + 1;
    return (size_t)depthBytes[depth] * (size_t)channels;
}
}  // namespace

extern "C" orbx_status orbx_archive_read_matrix(const uint8_t* buf, size_t len, int text, size_t* consumed, int* rows,
                                                int* cols, int* cv_type, uint8_t* out, size_t capacity) {
    ORBX_REQUIRE(buf && consumed && rows && cols && cv_type, "null argument");
    Reader r{buf, len, 0, text != 0};
    const int c = r.i32(), rw = r.i32(), type = r.i32();
    r.b8();                                         // `continuous` only decides how the writer split its make_array calls
    ORBX_REQUIRE(r.ok, "truncated or malformed matrix header");
    ORBX_REQUIRE(c >= 0 && rw >= 0 && type >= 0 && type < 512, "implausible matrix header (%d x %d, type %d)", rw, c, type);
    // the payload has to fit into what is left of the image (a text archive spends at least two characters per byte);
    // checked by division so that a forged header cannot wrap the byte count
    const size_t elem = elem_size_of_cv_type(type), left = (len - r.pos) / (r.text ? 2 : 1);
    ORBX_REQUIRE(c == 0 || rw == 0 || ((size_t)c <= left / elem && (size_t)rw <= left / elem / (size_t)c),
                 "matrix header promises more than the archive holds (%d x %d, type %d)", rw, c, type);
    const size_t bytes = (size_t)c * (size_t)rw * elem;
    *rows = rw; *cols = c; *cv_type = type;
    if (!out) {                                     // size query: skip the payload
        if (!r.text) { ORBX_REQUIRE(r.pos + bytes <= len, "truncated matrix payload"); r.pos += bytes; }
        else for (size_t i = 0; i < bytes && r.ok; ++i) r.u8();
        ORBX_REQUIRE(r.ok, "truncated matrix payload");
        *consumed = r.pos;
        return ORBX_OK;
    }
    if (bytes > capacity) { orbx_set_error("matrix payload is %zu bytes, capacity %zu", bytes, capacity); return ORBX_ERR_CAPACITY; }
    if (!r.text) {
        ORBX_REQUIRE(r.pos + bytes <= len, "truncated matrix payload");
        memcpy(out, buf + r.pos, bytes);
        r.pos += bytes;
    } else {
        for (size_t i = 0; i < bytes && r.ok; ++i) out[i] = r.u8();
        ORBX_REQUIRE(r.ok, "truncated or malformed matrix payload");
    }
    *consumed = r.pos;
    return ORBX_OK;
}

extern "C" orbx_status orbx_archive_write_matrix(const uint8_t* data, int rows, int cols, int cv_type, int text, uint8_t* out,
                                                 size_t capacity, size_t* written) {
    ORBX_REQUIRE(written && rows >= 0 && cols >= 0 && cv_type >= 0 && cv_type < 512, "bad argument");
    const size_t bytes = (size_t)rows * (size_t)cols * elem_size_of_cv_type(cv_type);
    ORBX_REQUIRE(bytes == 0 || data, "null argument");
    Writer w{out, capacity, 0, text != 0};
    w.i32(cols); w.i32(rows); w.i32(cv_type); w.b8(true);
    w.bytes(data, bytes);
    *written = w.pos;
    if (out && w.pos > capacity) { orbx_set_error("record is %zu bytes, capacity %zu", w.pos, capacity); return ORBX_ERR_CAPACITY; }
    return ORBX_OK;
}

extern "C" orbx_status orbx_archive_read_keypoints(const uint8_t* buf, size_t len, int text, size_t* consumed,
                                                   orbx_keypoint* out, int capacity, int* n_out) {
    ORBX_REQUIRE(buf && consumed && n_out, "null argument");
    Reader r{buf, len, 0, text != 0};
    const int n = r.i32();
    ORBX_REQUIRE(r.ok && n >= 0, "truncated or malformed keypoint vector");
    *n_out = n;
    if (out && n > capacity) { orbx_set_error("%d keypoints, capacity %d", n, capacity); return ORBX_ERR_CAPACITY; }
    for (int i = 0; i < n && r.ok; ++i) {
        orbx_keypoint k;
        k.angle = r.f32(); k.response = r.f32(); k.size = r.f32(); k.x = r.f32(); k.y = r.f32();
        k.class_id = r.i32(); k.octave = r.i32();
        if (out) out[i] = k;
    }
    ORBX_REQUIRE(r.ok, "truncated or malformed keypoint vector");
    *consumed = r.pos;
    return ORBX_OK;
}

extern "C" orbx_status orbx_archive_write_keypoints(const orbx_keypoint* kps, int n, int text, uint8_t* out, size_t capacity,
                                                    size_t* written) {
    ORBX_REQUIRE(written && n >= 0 && (n == 0 || kps), "bad argument");
    Writer w{out, capacity, 0, text != 0};
    w.i32(n);
    for (int i = 0; i < n; ++i) {
        w.f32(kps[i].angle); w.f32(kps[i].response); w.f32(kps[i].size); w.f32(kps[i].x); w.f32(kps[i].y);
        w.i32(kps[i].class_id); w.i32(kps[i].octave);
    }
    *written = w.pos;
    if (out && w.pos > capacity) { orbx_set_error("record is %zu bytes, capacity %zu", w.pos, capacity); return ORBX_ERR_CAPACITY; }
    return ORBX_OK;
}

// Descriptor matrices inside an archive image, found by the header serializeMatrix writes for them: cols = 32,
// type = CV_8UC1, continuous = true, min_rows <= rows, and a payload that fits (in a text archive: rows * 32 tokens in
// 0..255).  KeyFrame::mDescriptors records have rows = N, MapPoint::mDescriptor records rows = 1.
extern "C" orbx_status orbx_archive_scan_descriptors(const uint8_t* buf, size_t len, int text, int min_rows, size_t* payload_offset,
                                                     int* rows_out, int capacity, int* n_found, long long* total_rows) {
    ORBX_REQUIRE(buf && n_found && min_rows >= 1, "bad argument");
    int found = 0;
    long long total = 0;
    if (!text) {
        size_t pos = 0;
        while (pos + 13 <= len) {
            int32_t c, rw, type;
            memcpy(&c, buf + pos, 4);
            if (c != 32) { ++pos; continue; }
            memcpy(&rw, buf + pos + 4, 4); memcpy(&type, buf + pos + 8, 4);
            if (type != kCv8UC1 || buf[pos + 12] != 1 || rw < min_rows || (size_t)rw > (len - pos - 13) / 32) { ++pos; continue; }
            if (found < capacity) {
                if (payload_offset) payload_offset[found] = pos + 13;
                if (rows_out) rows_out[found] = rw;
            }
            ++found; total += rw;
            pos += 13 + (size_t)rw * 32;
        }
    } else {
        // token walk: remember the last four tokens; on "32 rows 0 1" verify the payload tokens
        Reader r{buf, len, 0, true};
        long long win[4] = {-1, -1, -1, -1};
        while (true) {
            const char* b; size_t n;
            const size_t before = r.pos;
            if (!r.token(&b, &n)) break;
            long long v = -2;
            if (n <= 10) {
                bool digits = n > 0;
                for (size_t i = 0; i < n; ++i) digits = digits && b[i] >= '0' && b[i] <= '9';
                if (digits) { char tmp[16]; memcpy(tmp, b, n); tmp[n] = 0; v = atoll(tmp); }
            }
            (void)before;
            win[0] = win[1]; win[1] = win[2]; win[2] = win[3]; win[3] = v;
            if (win[0] == 32 && win[1] >= min_rows && win[1] < (1ll << 31) && win[2] == kCv8UC1 && win[3] == 1) {
                Reader t{buf, len, r.pos, true};
                const size_t need = (size_t)win[1] * 32;
                size_t i = 0;
                for (; i < need && t.ok; ++i) t.u8();
                if (t.ok && i == need) {
                    if (found < capacity) {
                        if (payload_offset) payload_offset[found] = r.pos;
                        if (rows_out) rows_out[found] = (int)win[1];
                    }
                    ++found; total += win[1];
                    r.pos = t.pos;
                    win[0] = win[1] = win[2] = win[3] = -1;
                }
            }
        }
    }
    *n_found = found;
    if (total_rows) *total_rows = total;
    if ((payload_offset || rows_out) && found > capacity) { orbx_set_error("%d descriptor matrices, capacity %d", found, capacity); return ORBX_ERR_CAPACITY; }
    return ORBX_OK;
}

// The descriptor rows of one matrix found by the scan, as row-major rows x 32 bytes (a copy for a binary archive, a
// decode of the decimal tokens for a text archive).
extern "C" orbx_status orbx_archive_copy_descriptors(const uint8_t* buf, size_t len, int text, size_t payload_offset, int rows,
                                                     uint8_t* out) {
    ORBX_REQUIRE(buf && out && rows >= 0 && payload_offset <= len, "bad argument");
    const size_t bytes = (size_t)rows * 32;
    if (!text) {
        ORBX_REQUIRE(bytes <= len - payload_offset, "payload beyond the end of the archive");
        memcpy(out, buf + payload_offset, bytes);
        return ORBX_OK;
    }
    Reader r{buf, len, payload_offset, true};
    for (size_t i = 0; i < bytes && r.ok; ++i) out[i] = r.u8();
    ORBX_REQUIRE(r.ok, "malformed descriptor payload");
    return ORBX_OK;
}
