// CPU ORACLE -- TEST INFRASTRUCTURE ONLY.  Never linked or called by the product
// library.  Only tests/, the smoke check and the bench's cpu_baseline leg load it.
//
// PARITY UNPINNED: the reference ships no golden vectors for this path and its
// OpenCV dependency is absent from this machine, so nothing external pins this
// restatement.  The OpenCV routines the reference calls (cv::resize INTER_LINEAR,
// cv::FAST, cv::GaussianBlur 7x7 sigma 2, cv::fastAtan2, cvRound, BFMatcher) are
// restated here from their published OpenCV 4.x generic C++ behaviour.
//
// A plain, dependency-free restatement of the reference's ORB front-end:
//   ORBextractor::ORBextractor        src/ORBextractor.cc:477-536
//   ORBextractor::ComputePyramid      src/ORBextractor.cc:1402-1456
//   ComputeKeyPointsOctTree           src/ORBextractor.cc:887-1066  (incl. the fork's
//                                     per-cell scale round trip + rect erase :989-1025)
//   DistributeOctTree / DivideNode    src/ORBextractor.cc:550-883
//   IC_Angle                          src/ORBextractor.cc:90-137
//   computeOrbDescriptor              src/ORBextractor.cc:150-213
//   ORBextractor::operator()          src/ORBextractor.cc:1271-1399
//   ORBmatcher::DescriptorDistance    src/ORBmatcher.cc:2141-2157
//   Frame::ComputeStereoMatches       src/Frame.cc:962-1133
//   Frame::ComputeStereoFishEyeMatches (BFMatcher knn 2 + ratio)  src/Frame.cc:1286-1338
//   Frame::AssignFeaturesToGrid / GetFeaturesInArea   src/Frame.cc:468-506, :792-858
//   ORBmatcher::SearchByProjection(Frame&, vector<MapPoint*>&...) core  src/ORBmatcher.cc:42-216
//   ORBmatcher::SearchForTriangulation distance core   src/ORBmatcher.cc:1021-1179
//
// Float discipline: every float expression is evaluated as individually rounded
// IEEE-754 binary32 operations (build with -ffp-contract=off, no -ffast-math).

#include <algorithm>
#include <climits>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <list>
#include <utility>
#include <vector>

// camera-model gates of SearchForTriangulation / ComputeStereoFishEyeMatches: the oracle's own binary64 restatement
#include "camera_oracle.h"

namespace {

// ----------------------------------------------------------------------------
// small numeric helpers (OpenCV semantics)
// ----------------------------------------------------------------------------
inline int round_half_even(float v) { return (int)std::nearbyintf(v); }   // cvRound(float)
inline int round_half_even(double v) { return (int)std::nearbyint(v); }   // cvRound(double)
inline int reflect101(int p, int n) {
    // gfedcb|abcdefgh|gfedcba
    if (n == 1) return 0;
    while (p < 0 || p >= n) {
        if (p < 0) p = -p;
        else p = 2 * (n - 1) - p;
    }
    return p;
}

const int kPatch = 31;
const int kHalfPatch = 15;
const int kEdge = 19;

const signed char kPattern[1024] = {
#include "../orb_slam3_amd/csrc/orb_pattern.inc"
};

struct KeyPt {
    float x, y, size, angle, response;
    int octave, class_id;
};
static_assert(sizeof(KeyPt) == 28, "layout must equal cv::KeyPoint");

struct RectI { int x, y, w, h; };

struct Plane {
    int w = 0, h = 0;
    std::vector<uint8_t> px;
    uint8_t at(int y, int x) const { return px[(size_t)y * w + x]; }
};

// ----------------------------------------------------------------------------
// cv::resize(INTER_LINEAR) for 8UC1, fixed point with 11 coefficient bits
// ----------------------------------------------------------------------------
void linear_taps(int ssize, int dsize, std::vector<int>& ofs, std::vector<short>& coef) {
    const double inv_scale = (double)dsize / ssize;
    const double scale = 1. / inv_scale;
    ofs.resize(dsize);
    coef.resize(2 * (size_t)dsize);
    for (int d = 0; d < dsize; ++d) {
        float f = (float)((d + 0.5) * scale - 0.5);
        int s = (int)std::floor(f);
        f -= s;
        if (s < 0) { f = 0; s = 0; }
        if (s >= ssize - 1) { f = 0; s = ssize - 1; }
        ofs[d] = s;
        float c0 = 1.f - f, c1 = f;
        coef[2 * d] = (short)round_half_even(c0 * 2048.f);
        coef[2 * d + 1] = (short)round_half_even(c1 * 2048.f);
    }
}

void resize_linear_u8(const Plane& src, Plane& dst, int dw, int dh) {
    dst.w = dw; dst.h = dh; dst.px.assign((size_t)dw * dh, 0);
    std::vector<int> xo, yo; std::vector<short> xa, ya;
    linear_taps(src.w, dw, xo, xa);
    linear_taps(src.h, dh, yo, ya);
    std::vector<int> r0(dw), r1(dw);
    for (int y = 0; y < dh; ++y) {
        int sy0 = yo[y];
        int sy1 = std::min(sy0 + 1, src.h - 1);
        for (int x = 0; x < dw; ++x) {
            int sx0 = xo[x];
            int sx1 = std::min(sx0 + 1, src.w - 1);
            r0[x] = src.at(sy0, sx0) * xa[2 * x] + src.at(sy0, sx1) * xa[2 * x + 1];
            r1[x] = src.at(sy1, sx0) * xa[2 * x] + src.at(sy1, sx1) * xa[2 * x + 1];
        }
        int b0 = ya[2 * y], b1 = ya[2 * y + 1];
        for (int x = 0; x < dw; ++x) {
            int v = (((b0 * (r0[x] >> 4)) >> 16) + ((b1 * (r1[x] >> 4)) >> 16) + 2) >> 2;
            dst.px[(size_t)y * dw + x] = (uint8_t)v;
        }
    }
}

// ----------------------------------------------------------------------------
// cv::GaussianBlur(7x7, sigma 2, BORDER_REFLECT_101) on 8U: 8.8 fixed point
// ----------------------------------------------------------------------------
const int kGauss[7] = {18, 34, 48, 56, 48, 34, 18};

void gaussian7_u8(const Plane& src, Plane& dst) {
    const int w = src.w, h = src.h;
    dst.w = w; dst.h = h; dst.px.assign((size_t)w * h, 0);
    std::vector<uint32_t> hor((size_t)w * h);
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            uint32_t s = 0;
            for (int k = -3; k <= 3; ++k) s += kGauss[k + 3] * src.at(y, reflect101(x + k, w));
            hor[(size_t)y * w + x] = s;
        }
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            uint32_t s = 0;
            for (int k = -3; k <= 3; ++k) s += kGauss[k + 3] * hor[(size_t)reflect101(y + k, h) * w + x];
            dst.px[(size_t)y * w + x] = (uint8_t)((s + 32768u) >> 16);
        }
}

// ----------------------------------------------------------------------------
// cv::FAST(img, kps, threshold, nonmaxSuppression=true), 9-of-16 segment test
// ----------------------------------------------------------------------------
const int kCircle[16][2] = {{0, 3}, {1, 3}, {2, 2}, {3, 1}, {3, 0}, {3, -1}, {2, -2}, {1, -3},
                            {0, -3}, {-1, -3}, {-2, -2}, {-3, -1}, {-3, 0}, {-3, 1}, {-2, 2}, {-1, 3}};

// Threshold-free arc strength: the largest t' such that 9 contiguous circle pixels
// are all darker than v - t' (or all brighter than v + t'), plus one.  A pixel is a
// corner at threshold t iff strength > t, and its response is then strength - 1.
int fast_strength(const uint8_t* img, int stride, int x, int y) {
    int d[25];
    int v = img[(size_t)y * stride + x];
    for (int k = 0; k < 25; ++k) {
        const int* o = kCircle[k & 15];
        d[k] = v - img[(size_t)(y + o[1]) * stride + (x + o[0])];
    }
    int best = -255;
    for (int k = 0; k < 16; ++k) {
        int mn = d[k], mx = d[k];
        for (int j = 1; j < 9; ++j) { mn = std::min(mn, d[k + j]); mx = std::max(mx, d[k + j]); }
        best = std::max(best, std::max(mn, -mx));
    }
    return best;
}

struct Corner { int x, y, score; };

// img points at the cell's top-left pixel; w,h are the cell image size.
void fast9_nms(const uint8_t* img, int stride, int w, int h, int threshold, std::vector<Corner>& out) {
    out.clear();
    threshold = std::min(std::max(threshold, 0), 255);
    if (w < 7 || h < 7) return;
    std::vector<uint8_t> sc((size_t)w * h, 0);
    for (int y = 3; y < h - 3; ++y)
        for (int x = 3; x < w - 3; ++x) {
            int s = fast_strength(img, stride, x, y);
            if (s > threshold) sc[(size_t)y * w + x] = (uint8_t)(s - 1);
        }
    for (int y = 3; y < h - 3; ++y)
        for (int x = 3; x < w - 3; ++x) {
            int s = sc[(size_t)y * w + x];
            if (s == 0) continue;   // not a corner, or a corner of response 0 that cannot win a strict '>'
            bool keep = true;
            for (int dy = -1; dy <= 1 && keep; ++dy)
                for (int dx = -1; dx <= 1; ++dx) {
                    if (!dx && !dy) continue;
                    if (!(s > sc[(size_t)(y + dy) * w + (x + dx)])) { keep = false; break; }
                }
            if (keep) out.push_back({x, y, s});
        }
}

// ----------------------------------------------------------------------------
// cv::fastAtan2 (degrees, [0,360))
// ----------------------------------------------------------------------------
float fast_atan2_deg(float y, float x) {
    const float k = (float)(180 / 3.14159265358979323846);
    const float p1 = 0.9997878412794807f * k;
    const float p3 = -0.3258083974640975f * k;
    const float p5 = 0.1555786518463281f * k;
    const float p7 = -0.04432655554792128f * k;
    const float eps = (float)2.2204460492503131e-16;
    float ax = std::fabs(x), ay = std::fabs(y), a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + eps);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + eps);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

// ----------------------------------------------------------------------------
// cosf/sinf for angles in [0, 2*pi]: evaluated in binary64 and rounded once.
// (The reference calls the C library's float cos/sin, ORBextractor.cc:157, whose
// last bit is platform dependent; this definition is the canonical one here and is
// cross-checked against the host libm in the tests.)
// ----------------------------------------------------------------------------
void sincos_f32(float xf, float* s_out, float* c_out) {
    const double two_over_pi = 6.36619772367581382433e-01;
    const double pio2_hi = 1.57079632673412561417e+00;  // first 33 bits of pi/2
    const double pio2_lo = 6.07710050650619224932e-11;  // pi/2 - pio2_hi
    double x = (double)xf;
    int k = (int)(x * two_over_pi + 0.5);
    double r = (x - k * pio2_hi) - k * pio2_lo;
    double z = r * r;
    // Taylor series, |r| <= pi/4 (+ a hair): truncation < 1e-17
    double sp = 1.0 / 355687428096000.0;            // 1/17!
    sp = sp * z - 1.0 / 1307674368000.0;            // 1/15!
    sp = sp * z + 1.0 / 6227020800.0;               // 1/13!
    sp = sp * z - 1.0 / 39916800.0;                 // 1/11!
    sp = sp * z + 1.0 / 362880.0;                   // 1/9!
    sp = sp * z - 1.0 / 5040.0;                     // 1/7!
    sp = sp * z + 1.0 / 120.0;                      // 1/5!
    sp = sp * z - 1.0 / 6.0;                        // 1/3!
    double sn = r + (r * z) * sp;
    double cp = 1.0 / 20922789888000.0;             // 1/16!
    cp = cp * z - 1.0 / 87178291200.0;              // 1/14!
    cp = cp * z + 1.0 / 479001600.0;                // 1/12!
    cp = cp * z - 1.0 / 3628800.0;                  // 1/10!
    cp = cp * z + 1.0 / 40320.0;                    // 1/8!
    cp = cp * z - 1.0 / 720.0;                      // 1/6!
    cp = cp * z + 1.0 / 24.0;                       // 1/4!
    double cs = 1.0 - (0.5 * z - (z * z) * cp);
    double s, c;
    switch (k & 3) {
        case 0: s = sn; c = cs; break;
        case 1: s = cs; c = -sn; break;
        case 2: s = -sn; c = -cs; break;
        default: s = -cs; c = sn; break;
    }
    *s_out = (float)s;
    *c_out = (float)c;
}

// ----------------------------------------------------------------------------
// quadtree distribution
// ----------------------------------------------------------------------------
struct Cand {
    float x, y, response;
};

struct QNode {
    int ulx, uly, brx, bry;          // UL = (ulx,uly), UR = (brx,uly), BL = (ulx,bry), BR = (brx,bry)
    std::vector<Cand> keys;
    std::list<QNode>::iterator self;
    bool leaf = false;               // "bNoMore"
};

void split_node(const QNode& p, QNode c[4]) {
    const int hx = (int)std::ceil((float)(p.brx - p.ulx) / 2);
    const int hy = (int)std::ceil((float)(p.bry - p.uly) / 2);
    const int mx = p.ulx + hx, my = p.uly + hy;
    c[0].ulx = p.ulx; c[0].uly = p.uly; c[0].brx = mx;    c[0].bry = my;      // upper-left
    c[1].ulx = mx;    c[1].uly = p.uly; c[1].brx = p.brx; c[1].bry = my;      // upper-right
    c[2].ulx = p.ulx; c[2].uly = my;    c[2].brx = mx;    c[2].bry = p.bry;   // lower-left
    c[3].ulx = mx;    c[3].uly = my;    c[3].brx = p.brx; c[3].bry = p.bry;   // lower-right
    for (const Cand& k : p.keys) {
        int q;
        if (k.x < (float)mx) q = (k.y < (float)my) ? 0 : 2;
        else q = (k.y < (float)my) ? 1 : 3;
        c[q].keys.push_back(k);
    }
    for (int i = 0; i < 4; ++i) c[i].leaf = (c[i].keys.size() == 1);
}

typedef std::pair<int, QNode*> SizedNode;
bool sized_less(const SizedNode& a, const SizedNode& b) {
    if (a.first < b.first) return true;
    if (a.first > b.first) return false;
    return a.second->ulx < b.second->ulx;
}

// Pushes the non-empty children at the FRONT of the list in the order 0,1,2,3 and
// records the ones that can still be split.  Returns how many were recorded.
int emplace_children(std::list<QNode>& nodes, QNode c[4], std::vector<SizedNode>& expandable) {
    int rec = 0;
    for (int i = 0; i < 4; ++i) {
        if (c[i].keys.empty()) continue;
        nodes.push_front(std::move(c[i]));
        nodes.front().self = nodes.begin();
        if (nodes.front().keys.size() > 1) {
            expandable.push_back(SizedNode((int)nodes.front().keys.size(), &nodes.front()));
            ++rec;
        }
    }
    return rec;
}

std::vector<Cand> distribute_quadtree(const std::vector<Cand>& cands, int minX, int maxX, int minY, int maxY, int N) {
    std::vector<Cand> result;
    const int nIni = (int)std::round((float)(maxX - minX) / (maxY - minY));
    if (nIni < 1) return result;
    const float hX = (float)(maxX - minX) / nIni;

    std::list<QNode> nodes;
    std::vector<QNode*> roots(nIni);
    for (int i = 0; i < nIni; ++i) {
        QNode n;
        n.ulx = (int)(hX * (float)i);
        n.brx = (int)(hX * (float)(i + 1));
        n.uly = 0;
        n.bry = maxY - minY;
        nodes.push_back(n);
        roots[i] = &nodes.back();
    }
    for (const Cand& k : cands) {
        size_t r = (size_t)(k.x / hX);
        if (r >= roots.size()) r = roots.size() - 1;   // reference would index out of range
        roots[r]->keys.push_back(k);
    }
    for (auto it = nodes.begin(); it != nodes.end();) {
        if (it->keys.size() == 1) { it->leaf = true; ++it; }
        else if (it->keys.empty()) it = nodes.erase(it);
        else ++it;
    }

    bool done = false;
    std::vector<SizedNode> expandable;
    while (!done) {
        const int prev = (int)nodes.size();
        int nToExpand = 0;
        expandable.clear();
        for (auto it = nodes.begin(); it != nodes.end();) {
            if (it->leaf) { ++it; continue; }
            QNode c[4];
            split_node(*it, c);
            nToExpand += emplace_children(nodes, c, expandable);
            it = nodes.erase(it);
        }
        if ((int)nodes.size() >= N || (int)nodes.size() == prev) {
            done = true;
        } else if ((int)nodes.size() + nToExpand * 3 > N) {
            while (!done) {
                const int prev2 = (int)nodes.size();
                std::vector<SizedNode> todo = expandable;
                expandable.clear();
                std::sort(todo.begin(), todo.end(), sized_less);
                for (int j = (int)todo.size() - 1; j >= 0; --j) {
                    QNode c[4];
                    split_node(*todo[j].second, c);
                    emplace_children(nodes, c, expandable);
                    nodes.erase(todo[j].second->self);
                    if ((int)nodes.size() >= N) break;
                }
                if ((int)nodes.size() >= N || (int)nodes.size() == prev2) done = true;
            }
        }
    }

    result.reserve(nodes.size());
    for (const QNode& n : nodes) {
        const Cand* best = &n.keys[0];
        for (size_t k = 1; k < n.keys.size(); ++k)
            if (n.keys[k].response > best->response) best = &n.keys[k];
        result.push_back(*best);
    }
    return result;
}

// ----------------------------------------------------------------------------
// the extractor
// ----------------------------------------------------------------------------
struct Extractor {
    int nfeatures, nlevels, iniTh, minTh;
    double scaleFactor;
    std::vector<float> sf, inv_sf, sigma2, inv_sigma2;
    std::vector<int> quota;
    int umax[kHalfPatch + 1];
    std::vector<Plane> pyr;                       // borderless level images
    std::vector<std::vector<Cand>> last_cands;    // per level, after FAST + round trip (debug / stage parity)
    std::vector<std::vector<KeyPt>> last_level_kps;

    Extractor(int nf, float scale, int nl, int ini, int mn)
        : nfeatures(nf), nlevels(nl), iniTh(ini), minTh(mn), scaleFactor(scale) {
        sf.resize(nl); inv_sf.resize(nl); sigma2.resize(nl); inv_sigma2.resize(nl);
        sf[0] = 1.f; sigma2[0] = 1.f;
        for (int i = 1; i < nl; ++i) {
            sf[i] = (float)(sf[i - 1] * scaleFactor);
            sigma2[i] = sf[i] * sf[i];
        }
        for (int i = 0; i < nl; ++i) { inv_sf[i] = 1.0f / sf[i]; inv_sigma2[i] = 1.0f / sigma2[i]; }
        quota.resize(nl);
        float factor = (float)(1.0f / scaleFactor);
        float per = nf * (1 - factor) / (1 - (float)std::pow((double)factor, (double)nl));
        int sum = 0;
        for (int l = 0; l < nl - 1; ++l) {
            quota[l] = round_half_even(per);
            sum += quota[l];
            per *= factor;
        }
        quota[nl - 1] = std::max(nf - sum, 0);

        int vmax = (int)std::floor(kHalfPatch * std::sqrt(2.f) / 2 + 1);
        int vmin = (int)std::ceil(kHalfPatch * std::sqrt(2.f) / 2);
        const double hp2 = kHalfPatch * kHalfPatch;
        for (int v = 0; v <= kHalfPatch; ++v) umax[v] = 0;
        for (int v = 0; v <= vmax; ++v) umax[v] = round_half_even(std::sqrt(hp2 - v * v));
        for (int v = kHalfPatch, v0 = 0; v >= vmin; --v) {
            while (umax[v0] == umax[v0 + 1]) ++v0;
            umax[v] = v0;
            ++v0;
        }
        pyr.resize(nl);
    }

    void level_size(int w, int h, int level, int* lw, int* lh) const {
        float s = inv_sf[level];
        *lw = round_half_even((float)w * s);
        *lh = round_half_even((float)h * s);
    }

    void build_pyramid(const uint8_t* img, int w, int h, ptrdiff_t stride) {
        pyr[0].w = w; pyr[0].h = h; pyr[0].px.resize((size_t)w * h);
        for (int y = 0; y < h; ++y) std::memcpy(&pyr[0].px[(size_t)y * w], img + y * stride, w);
        for (int l = 1; l < nlevels; ++l) {
            int lw, lh;
            level_size(w, h, l, &lw, &lh);
            resize_linear_u8(pyr[l - 1], pyr[l], lw, lh);
        }
    }

    // FAST over the 35-px cell grid + the fork's per-cell scale round trip / rect erase.
    void detect_level(int level, const RectI* rects, int nrects, std::vector<Cand>& keys) const {
        keys.clear();
        const Plane& im = pyr[level];
        const int minBX = kEdge - 3, minBY = minBX;
        const int maxBX = im.w - kEdge + 3, maxBY = im.h - kEdge + 3;
        const float width = (float)(maxBX - minBX), height = (float)(maxBY - minBY);
        const float W = 35;
        const int nCols = (int)(width / W), nRows = (int)(height / W);
        if (nCols < 1 || nRows < 1) return;
        const int wCell = (int)std::ceil(width / nCols), hCell = (int)std::ceil(height / nRows);
        const float scale = sf[level];
        const float scale_inv = 1 / scale;
        std::vector<Corner> cell;
        for (int i = 0; i < nRows; ++i) {
            const float iniY = (float)(minBY + i * hCell);
            float maxY = iniY + hCell + 6;
            if (iniY >= maxBY - 3) continue;
            if (maxY > maxBY) maxY = (float)maxBY;
            for (int j = 0; j < nCols; ++j) {
                const float iniX = (float)(minBX + j * wCell);
                float maxX = iniX + wCell + 6;
                if (iniX >= maxBX - 3) continue;
                if (maxX > maxBX) maxX = (float)maxBX;
                const int x0 = (int)iniX, x1 = (int)maxX, y0 = (int)iniY, y1 = (int)maxY;
                const uint8_t* base = &im.px[(size_t)y0 * im.w + x0];
                fast9_nms(base, im.w, x1 - x0, y1 - y0, iniTh, cell);
                if (cell.empty()) fast9_nms(base, im.w, x1 - x0, y1 - y0, minTh, cell);
                for (const Corner& c : cell) {
                    Cand k;
                    k.x = (float)c.x; k.y = (float)c.y; k.response = (float)c.score;
                    k.x += j * wCell;
                    k.y += i * hCell;
                    keys.push_back(k);
                }
                // fork block: to level-0 scale, erase inside dynamic rects, back again
                for (Cand& k : keys) {
                    k.x += minBX; k.y += minBY;
                    k.x *= scale; k.y *= scale;
                }
                for (auto it = keys.begin(); it != keys.end();) {
                    bool hit = false;
                    for (int r = 0; r < nrects; ++r) {
                        int px = round_half_even(it->x), py = round_half_even(it->y);
                        if (rects[r].x <= px && px < rects[r].x + rects[r].w &&
                            rects[r].y <= py && py < rects[r].y + rects[r].h) { hit = true; break; }
                    }
                    if (hit) it = keys.erase(it); else ++it;
                }
                for (Cand& k : keys) {
                    k.x *= scale_inv; k.y *= scale_inv;
                    k.x -= minBX; k.y -= minBY;
                }
            }
        }
    }

    float orientation(const Plane& im, float fx, float fy) const {
        int m01 = 0, m10 = 0;
        const int cx = round_half_even(fx), cy = round_half_even(fy);
        for (int u = -kHalfPatch; u <= kHalfPatch; ++u) m10 += u * im.at(cy, cx + u);
        for (int v = 1; v <= kHalfPatch; ++v) {
            int vs = 0, d = umax[v];
            for (int u = -d; u <= d; ++u) {
                int p = im.at(cy + v, cx + u), m = im.at(cy - v, cx + u);
                vs += p - m;
                m10 += u * (p + m);
            }
            m01 += v * vs;
        }
        return fast_atan2_deg((float)m01, (float)m10);
    }

    void describe(const Plane& blurred, const KeyPt& kp, uint8_t* out) const {
        const float deg2rad = (float)(3.14159265358979323846 / 180.f);
        float angle = kp.angle * deg2rad;
        float a, b;
        sincos_f32(angle, &b, &a);
        const int cx = round_half_even(kp.x), cy = round_half_even(kp.y);
        for (int i = 0; i < 32; ++i) {
            int val = 0;
            for (int j = 0; j < 8; ++j) {
                int t[2];
                for (int e = 0; e < 2; ++e) {
                    int idx = (i * 16 + j * 2 + e) * 2;
                    float px = (float)kPattern[idx], py = (float)kPattern[idx + 1];
                    float ry = px * b + py * a;
                    float rx = px * a - py * b;
                    t[e] = blurred.at(cy + round_half_even(ry), cx + round_half_even(rx));
                }
                val |= (t[0] < t[1]) << j;
            }
            out[i] = (uint8_t)val;
        }
    }

    // Full ORBextractor::operator().  Returns monoIndex (or -1 on an empty image).
    int extract(const uint8_t* img, int w, int h, ptrdiff_t stride, int lap0, int lap1,
                const RectI* rects, int nrects, std::vector<KeyPt>& out_kps, std::vector<uint8_t>& out_desc) {
        out_kps.clear(); out_desc.clear();
        if (!img || w <= 0 || h <= 0) return -1;
        build_pyramid(img, w, h, stride);
        last_cands.assign(nlevels, {});
        last_level_kps.assign(nlevels, {});
        for (int l = 0; l < nlevels; ++l) {
            detect_level(l, rects, nrects, last_cands[l]);
            const Plane& im = pyr[l];
            const int minB = kEdge - 3;
            std::vector<Cand> kept = distribute_quadtree(last_cands[l], minB, im.w - kEdge + 3, minB, im.h - kEdge + 3, quota[l]);
            const int patch = (int)(kPatch * sf[l]);
            for (const Cand& c : kept) {
                KeyPt kp;
                kp.x = c.x + minB; kp.y = c.y + minB;
                kp.size = (float)patch; kp.angle = -1; kp.response = c.response;
                kp.octave = l; kp.class_id = -1;
                last_level_kps[l].push_back(kp);
            }
        }
        for (int l = 0; l < nlevels; ++l)
            for (KeyPt& kp : last_level_kps[l]) kp.angle = orientation(pyr[l], kp.x, kp.y);

        size_t total = 0;
        for (int l = 0; l < nlevels; ++l) total += last_level_kps[l].size();
        out_kps.resize(total);
        out_desc.assign(total * 32, 0);
        int mono = 0, stereo = (int)total - 1;
        for (int l = 0; l < nlevels; ++l) {
            std::vector<KeyPt>& kps = last_level_kps[l];
            if (kps.empty()) continue;
            Plane blurred;
            gaussian7_u8(pyr[l], blurred);
            const float scale = sf[l];
            for (KeyPt kp : kps) {   // by value: the per-level list keeps level coordinates
                uint8_t d[32];
                describe(blurred, kp, d);
                if (l != 0) { kp.x *= scale; kp.y *= scale; }
                int pos;
                if (kp.x >= (float)lap0 && kp.x <= (float)lap1) pos = stereo--;
                else pos = mono++;
                out_kps[pos] = kp;
                std::memcpy(&out_desc[(size_t)pos * 32], d, 32);
            }
        }
        return mono;
    }
};

// ----------------------------------------------------------------------------
// matching
// ----------------------------------------------------------------------------
inline int hamming256(const uint8_t* a, const uint8_t* b) {
    // the bit-trick popcount of ORBmatcher.cc:2141-2157, one 32-bit word at a time
    const uint32_t* pa = (const uint32_t*)a;
    const uint32_t* pb = (const uint32_t*)b;
    int dist = 0;
    for (int i = 0; i < 8; ++i) {
        uint32_t v = pa[i] ^ pb[i];
        v = v - ((v >> 1) & 0x55555555u);
        v = (v & 0x33333333u) + ((v >> 2) & 0x33333333u);
        dist += (int)((((v + (v >> 4)) & 0xF0F0F0Fu) * 0x1010101u) >> 24);
    }
    return dist;
}

inline int hamming256_fast(const uint8_t* a, const uint8_t* b) {
    uint64_t x[4], y[4];
    std::memcpy(x, a, 32); std::memcpy(y, b, 32);
    return __builtin_popcountll(x[0] ^ y[0]) + __builtin_popcountll(x[1] ^ y[1]) +
           __builtin_popcountll(x[2] ^ y[2]) + __builtin_popcountll(x[3] ^ y[3]);
}

struct Match2 { int idx0, dist0, idx1, dist1; };

}  // namespace

// ============================================================================
// C entry points (ctypes)
// ============================================================================
extern "C" {

void* orc_create(int nfeatures, float scale, int nlevels, int ini_th, int min_th) {
    return new Extractor(nfeatures, scale, nlevels, ini_th, min_th);
}
void orc_destroy(void* h) { delete (Extractor*)h; }

void orc_tables(void* h, float* sf, float* inv_sf, float* sigma2, float* inv_sigma2, int* quota, int* umax16) {
    Extractor* e = (Extractor*)h;
    for (int i = 0; i < e->nlevels; ++i) {
        sf[i] = e->sf[i]; inv_sf[i] = e->inv_sf[i]; sigma2[i] = e->sigma2[i];
        inv_sigma2[i] = e->inv_sigma2[i]; quota[i] = e->quota[i];
    }
    for (int i = 0; i < 16; ++i) umax16[i] = e->umax[i];
}

void orc_level_size(void* h, int w, int ht, int level, int* lw, int* lh) {
    ((Extractor*)h)->level_size(w, ht, level, lw, lh);
}

int orc_extract(void* h, const uint8_t* img, int w, int ht, long stride, int lap0, int lap1,
                const int* rects_xywh, int nrects, void* kps_out, uint8_t* desc_out, int cap, int* n_out) {
    Extractor* e = (Extractor*)h;
    std::vector<KeyPt> kps; std::vector<uint8_t> desc;
    int mono = e->extract(img, w, ht, stride, lap0, lap1, (const RectI*)rects_xywh, nrects, kps, desc);
    *n_out = (int)kps.size();
    int n = std::min((int)kps.size(), cap);
    if (n > 0) {
        std::memcpy(kps_out, kps.data(), (size_t)n * sizeof(KeyPt));
        std::memcpy(desc_out, desc.data(), (size_t)n * 32);
    }
    return mono;
}

// Stage accessors, valid after orc_extract.
int orc_pyramid_level(void* h, int level, uint8_t* out, int cap_bytes, int* w, int* ht) {
    Extractor* e = (Extractor*)h;
    const Plane& p = e->pyr[level];
    *w = p.w; *ht = p.h;
    if ((size_t)cap_bytes < p.px.size()) return -1;
    if (out) std::memcpy(out, p.px.data(), p.px.size());
    return 0;
}
int orc_level_candidates(void* h, int level, float* xyr, int cap) {
    Extractor* e = (Extractor*)h;
    const std::vector<Cand>& c = e->last_cands[level];
    int n = std::min((int)c.size(), cap);
    for (int i = 0; i < n; ++i) { xyr[3 * i] = c[i].x; xyr[3 * i + 1] = c[i].y; xyr[3 * i + 2] = c[i].response; }
    return (int)c.size();
}
int orc_level_keypoints(void* h, int level, void* kps, int cap) {
    Extractor* e = (Extractor*)h;
    const std::vector<KeyPt>& k = e->last_level_kps[level];
    int n = std::min((int)k.size(), cap);
    if (n) std::memcpy(kps, k.data(), (size_t)n * sizeof(KeyPt));
    return (int)k.size();
}

// ---- input staging in front of the extractor ------------------------------------------------
// What System::TrackStereo / TrackMonocular and Tracking::GrabImage* do to the camera image before
// ORBextractor sees it (System.cc:355-376, Tracking.cc:1591-1619): cv::remap with the rectification
// maps OR cv::resize to the configured size, on the image as delivered (1, 3 or 4 channels), then
// cv::cvtColor to grey.  The OpenCV arithmetic is restated from the 4.x sources from memory
// (OpenCV is not in this tree): parity unpinned.

// cv::cvtColor(COLOR_{RGB,BGR,RGBA,BGRA}2GRAY), 8U: 15-bit coefficients R 9798, G 19235, B 3735.
void orc_cvt_gray(const uint8_t* src, int w, int h, long stride, int cn, int rgb, uint8_t* dst) {
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            const uint8_t* p = src + (size_t)y * stride + (size_t)x * cn;
            const int r = rgb ? p[0] : p[2], g = p[1], b = rgb ? p[2] : p[0];
            dst[(size_t)y * w + x] = (uint8_t)((r * 9798 + g * 19235 + b * 3735 + (1 << 14)) >> 15);
        }
}

// cv::resize(src, dst, Size(dw, dh)) (INTER_LINEAR) for 8U with cn interleaved channels; dst is packed.
// An exact 2 x 2 decimation is routed to the INTER_AREA fast path by cv::resize: (a + b + c + d + 2) >> 2.
void orc_resize_linear_cn(const uint8_t* src, int sw, int sh, long stride, int cn, uint8_t* dst, int dw, int dh) {
    if (sw == dw && sh == dh) {
        for (int y = 0; y < sh; ++y) std::memcpy(dst + (size_t)y * dw * cn, src + (size_t)y * stride, (size_t)dw * cn);
        return;
    }
    if (sw == 2 * dw && sh == 2 * dh) {
        for (int y = 0; y < dh; ++y)
            for (int x = 0; x < dw; ++x)
                for (int c = 0; c < cn; ++c) {
                    const uint8_t* p = src + (size_t)(2 * y) * stride + (size_t)(2 * x) * cn + c;
                    dst[((size_t)y * dw + x) * cn + c] = (uint8_t)((p[0] + p[cn] + p[stride] + p[stride + cn] + 2) >> 2);
                }
        return;
    }
    std::vector<int> xo, yo; std::vector<short> xa, ya;
    linear_taps(sw, dw, xo, xa);
    linear_taps(sh, dh, yo, ya);
    for (int y = 0; y < dh; ++y) {
        const int sy0 = yo[y], sy1 = std::min(sy0 + 1, sh - 1);
        const int b0 = ya[2 * y], b1 = ya[2 * y + 1];
        for (int x = 0; x < dw; ++x) {
            const int sx0 = xo[x], sx1 = std::min(sx0 + 1, sw - 1);
            for (int c = 0; c < cn; ++c) {
                auto S = [&](int yy, int xx) { return (int)src[(size_t)yy * stride + (size_t)xx * cn + c]; };
                const int r0 = S(sy0, sx0) * xa[2 * x] + S(sy0, sx1) * xa[2 * x + 1];
                const int r1 = S(sy1, sx0) * xa[2 * x] + S(sy1, sx1) * xa[2 * x + 1];
                dst[((size_t)y * dw + x) * cn + c] = (uint8_t)((((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2);
            }
        }
    }
}

// cv::remap(src, dst, map_x, map_y, INTER_LINEAR) with CV_32FC1 maps and the default BORDER_CONSTANT (0):
// coordinates to 5 fractional bits (cvRound(map * 32)), weights from the 32 x 32 table of 15-bit bilinear
// coefficients built the way initInterTab2D builds it (saturation to short and its sum repair included),
// result (sum + 2^14) >> 15.  dst is packed dw x dh x cn.
void orc_remap_linear(const uint8_t* src, int sw, int sh, long stride, int cn, const float* mapx, const float* mapy,
                      int dw, int dh, uint8_t* dst) {
    static short tab[32 * 32][4];
    static bool init = false;
    if (!init) {
        for (int i = 0; i < 32; ++i)
            for (int j = 0; j < 32; ++j) {
                const float vy[2] = {1.f - i / 32.f, i / 32.f}, vx[2] = {1.f - j / 32.f, j / 32.f};
                short* t = tab[i * 32 + j];
                int isum = 0;
                for (int k1 = 0; k1 < 2; ++k1)
                    for (int k2 = 0; k2 < 2; ++k2) {
                        const int v = round_half_even(vy[k1] * vx[k2] * 32768.f);
                        t[k1 * 2 + k2] = (short)std::max(-32768, std::min(32767, v));
                        isum += t[k1 * 2 + k2];
                    }
                if (isum != 32768) {
                    // the repair looks at rows / columns ksize/2 .. ksize/2 + 1 of the kernel; for the 2 x 2 kernel
                    // only its last element lies inside (the others belong to entries not yet written: zero)
                    const int diff = isum - 32768;
                    t[3] = (short)(t[3] - diff);
                }
            }
        init = true;
    }
    const unsigned width1 = (unsigned)std::max(sw - 1, 0), height1 = (unsigned)std::max(sh - 1, 0);
    auto sat16 = [](int v) { return std::max(-32768, std::min(32767, v)); };
    for (int y = 0; y < dh; ++y)
        for (int x = 0; x < dw; ++x) {
            const int fx = round_half_even(mapx[(size_t)y * dw + x] * 32.f);
            const int fy = round_half_even(mapy[(size_t)y * dw + x] * 32.f);
            const int sx = sat16(fx >> 5), sy = sat16(fy >> 5);
            const short* w = tab[(fy & 31) * 32 + (fx & 31)];
            uint8_t* D = dst + ((size_t)y * dw + x) * cn;
            if ((unsigned)sx < width1 && (unsigned)sy < height1) {
                const uint8_t* S = src + (size_t)sy * stride + (size_t)sx * cn;
                for (int c = 0; c < cn; ++c)
                    D[c] = (uint8_t)((S[c] * w[0] + S[c + cn] * w[1] + S[stride + c] * w[2] + S[stride + cn + c] * w[3] + (1 << 14)) >> 15);
            } else if (sx >= sw || sx + 1 < 0 || sy >= sh || sy + 1 < 0) {
                for (int c = 0; c < cn; ++c) D[c] = 0;
            } else {
                const bool x0 = sx >= 0 && sx < sw, x1 = sx + 1 >= 0 && sx + 1 < sw;
                const bool y0 = sy >= 0 && sy < sh, y1 = sy + 1 >= 0 && sy + 1 < sh;
                for (int c = 0; c < cn; ++c) {
                    auto S = [&](int yy, int xx) { return (int)src[(size_t)yy * stride + (size_t)xx * cn + c]; };
                    const int v0 = (x0 && y0) ? S(sy, sx) : 0, v1 = (x1 && y0) ? S(sy, sx + 1) : 0;
                    const int v2 = (x0 && y1) ? S(sy + 1, sx) : 0, v3 = (x1 && y1) ? S(sy + 1, sx + 1) : 0;
                    D[c] = (uint8_t)((v0 * w[0] + v1 * w[1] + v2 * w[2] + v3 * w[3] + (1 << 14)) >> 15);
                }
            }
        }
}

// cv::undistortPoints(pts, pts, K, D, noArray(), K) on CV_32FC2 points, as Frame::UndistortKeyPoints /
// ComputeImageBounds use it (Frame.cc:901, :932); OpenCV 4.x cvUndistortPointsInternal with the default criteria
// (5 iterations, no epsilon test), no tilt terms, restated from memory: parity unpinned.
// K = {fx, fy, cx, cy} as floats, D = k1 k2 p1 p2 [k3 [k4 k5 k6]].
void orc_undistort_points(const float* xy, int n, const float* K, const float* D, int nD, float* out) {
    double k[14] = {0};
    for (int i = 0; i < nD && i < 8; ++i) k[i] = D[i];
    const double fx = K[0], fy = K[1], cx = K[2], cy = K[3];
    const double ifx = 1. / fx, ify = 1. / fy;
    const double RR[3][3] = {{fx, 0, cx}, {0, fy, cy}, {0, 0, 1}};     // P * R with R = I, P = K
    for (int i = 0; i < n; ++i) {
        double x = xy[2 * i], y = xy[2 * i + 1];
        const double u = x, v = y;
        x = (x - cx) * ifx;
        y = (y - cy) * ify;
        const double x0 = x, y0 = y;
        for (int j = 0; j < 5; ++j) {
            const double r2 = x * x + y * y;
            const double icdist = (1 + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2) / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2);
            if (icdist < 0) { x = (u - cx) * ifx; y = (v - cy) * ify; break; }
            const double deltaX = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x) + k[8] * r2 + k[9] * r2 * r2;
            const double deltaY = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y + k[10] * r2 + k[11] * r2 * r2;
            x = (x0 - deltaX) * icdist;
            y = (y0 - deltaY) * icdist;
        }
        const double xx = RR[0][0] * x + RR[0][1] * y + RR[0][2];
        const double yy = RR[1][0] * x + RR[1][1] * y + RR[1][2];
        const double ww = 1. / (RR[2][0] * x + RR[2][1] * y + RR[2][2]);
        out[2 * i] = (float)(xx * ww);
        out[2 * i + 1] = (float)(yy * ww);
    }
}
// Frame::ComputeImageBounds (Frame.cc:916-947): {mnMinX, mnMaxX, mnMinY, mnMaxY}
void orc_image_bounds(int w, int h, const float* K, const float* D, int nD, float* b) {
    if (nD == 0 || D[0] == 0.0f) { b[0] = 0.f; b[1] = (float)w; b[2] = 0.f; b[3] = (float)h; return; }
    const float c[8] = {0.f, 0.f, (float)w, 0.f, 0.f, (float)h, (float)w, (float)h};
    float o[8];
    orc_undistort_points(c, 4, K, D, nD, o);
    b[0] = std::min(o[0], o[4]); b[1] = std::max(o[2], o[6]);
    b[2] = std::min(o[1], o[3]); b[3] = std::max(o[5], o[7]);
}

// Stand-alone stages.
void orc_resize_linear(const uint8_t* src, int sw, int sh, uint8_t* dst, int dw, int dh) {
    Plane s, d;
    s.w = sw; s.h = sh; s.px.assign(src, src + (size_t)sw * sh);
    resize_linear_u8(s, d, dw, dh);
    std::memcpy(dst, d.px.data(), d.px.size());
}
void orc_gaussian7(const uint8_t* src, int w, int h, uint8_t* dst) {
    Plane s, d;
    s.w = w; s.h = h; s.px.assign(src, src + (size_t)w * h);
    gaussian7_u8(s, d);
    std::memcpy(dst, d.px.data(), d.px.size());
}
int orc_fast(const uint8_t* img, int w, int h, long stride, int threshold, int* xys, int cap) {
    std::vector<Corner> c;
    fast9_nms(img, (int)stride, w, h, threshold, c);
    int n = std::min((int)c.size(), cap);
    for (int i = 0; i < n; ++i) { xys[3 * i] = c[i].x; xys[3 * i + 1] = c[i].y; xys[3 * i + 2] = c[i].score; }
    return (int)c.size();
}
int orc_fast_strength(const uint8_t* img, long stride, int x, int y) { return fast_strength(img, (int)stride, x, y); }
float orc_fast_atan2(float y, float x) { return fast_atan2_deg(y, x); }
void orc_sincos(float x, float* s, float* c) { sincos_f32(x, s, c); }
void orc_sincos_many(const float* x, int n, float* s, float* c) {
    for (int i = 0; i < n; ++i) sincos_f32(x[i], s + i, c + i);
}
// Count of floats in [lo,hi) (stepping `stride` representable values at a time) where this
// definition differs from the host C library's sinf/cosf.
long orc_sincos_vs_libm(float lo, float hi, int stride, long* n_tested) {
    long bad = 0, n = 0;
    uint32_t a, b;
    std::memcpy(&a, &lo, 4); std::memcpy(&b, &hi, 4);
    for (uint64_t u = a; u < b; u += (uint32_t)stride) {
        uint32_t uu = (uint32_t)u;
        float x, s, c;
        std::memcpy(&x, &uu, 4);
        sincos_f32(x, &s, &c);
        if (s != sinf(x) || c != cosf(x)) ++bad;
        ++n;
    }
    *n_tested = n;
    return bad;
}
int orc_distribute(const float* xyr, int n, int minX, int maxX, int minY, int maxY, int N, float* out_xyr, int cap) {
    std::vector<Cand> c(n);
    for (int i = 0; i < n; ++i) { c[i].x = xyr[3 * i]; c[i].y = xyr[3 * i + 1]; c[i].response = xyr[3 * i + 2]; }
    std::vector<Cand> r = distribute_quadtree(c, minX, maxX, minY, maxY, N);
    int m = std::min((int)r.size(), cap);
    for (int i = 0; i < m; ++i) { out_xyr[3 * i] = r[i].x; out_xyr[3 * i + 1] = r[i].y; out_xyr[3 * i + 2] = r[i].response; }
    return (int)r.size();
}

// ---- matching -------------------------------------------------------------
int orc_descriptor_distance(const uint8_t* a, const uint8_t* b) { return hamming256(a, b); }

// BFMatcher(NORM_HAMMING).knnMatch(q, t, 2): two smallest distances per query,
// ties resolved towards the lower train index.  out = nq x {idx0,dist0,idx1,dist1};
// missing neighbours are (-1, INT_MAX).  `fast` selects the hardware popcount.
void orc_knn2(const uint8_t* q, int nq, const uint8_t* t, long nt, int* out, int fast) {
    for (int i = 0; i < nq; ++i) {
        int b0 = INT_MAX, b1 = INT_MAX, i0 = -1, i1 = -1;
        const uint8_t* qd = q + (size_t)i * 32;
        for (long j = 0; j < nt; ++j) {
            int d = fast ? hamming256_fast(qd, t + (size_t)j * 32) : hamming256(qd, t + (size_t)j * 32);
            if (d < b0) { b1 = b0; i1 = i0; b0 = d; i0 = (int)j; }
            else if (d < b1) { b1 = d; i1 = (int)j; }
        }
        out[4 * i] = i0; out[4 * i + 1] = b0; out[4 * i + 2] = i1; out[4 * i + 3] = b1;
    }
}

// Frame::ComputeStereoMatches.  Pyramids are borderless level images given as
// per-level pointers / widths / heights (strides == widths).
// Returns the number of left keypoints that keep a match.
int orc_stereo_match(const void* kpsL_, const uint8_t* descL, int nL,
                     const void* kpsR_, const uint8_t* descR, int nR,
                     const uint8_t* const* pyrL, const uint8_t* const* pyrR, const int* lw, const int* lh,
                     const float* sf, const float* inv_sf, int nlevels,
                     float mbf, float mb, float* uRight, float* depth, int* best_dist_out, int* best_idx_out) {
    (void)nlevels; (void)lh;
    const KeyPt* kL = (const KeyPt*)kpsL_;
    const KeyPt* kR = (const KeyPt*)kpsR_;
    const int thOrbDist = (100 + 50) / 2;
    const int nRows = lh[0];
    for (int i = 0; i < nL; ++i) { uRight[i] = -1.f; depth[i] = -1.f; if (best_dist_out) { best_dist_out[i] = -1; best_idx_out[i] = -1; } }
    std::vector<std::vector<size_t>> rows(nRows);
    for (int iR = 0; iR < nR; ++iR) {
        const float y = kR[iR].y;
        const float r = 2.0f * sf[kR[iR].octave];
        const int maxr = (int)std::ceil(y + r);
        const int minr = (int)std::floor(y - r);
        for (int yi = minr; yi <= maxr; ++yi)
            if (yi >= 0 && yi < nRows) rows[yi].push_back(iR);
    }
    const float minZ = mb, minD = 0, maxD = mbf / minZ;
    std::vector<std::pair<int, int>> distIdx;
    for (int iL = 0; iL < nL; ++iL) {
        const KeyPt& kp = kL[iL];
        const int levelL = kp.octave;
        const float vL = kp.y, uL = kp.x;
        size_t row = (size_t)vL;
        if (row >= rows.size()) continue;
        const std::vector<size_t>& cand = rows[row];
        if (cand.empty()) continue;
        const float minU = uL - maxD, maxU = uL - minD;
        if (maxU < 0) continue;
        int bestDist = 100;
        size_t bestIdxR = 0;
        for (size_t iC = 0; iC < cand.size(); ++iC) {
            const size_t iR = cand[iC];
            const KeyPt& kr = kR[iR];
            if (kr.octave < levelL - 1 || kr.octave > levelL + 1) continue;
            const float uR = kr.x;
            if (uR >= minU && uR <= maxU) {
                int d = hamming256(descL + (size_t)iL * 32, descR + iR * 32);
                if (d < bestDist) { bestDist = d; bestIdxR = iR; }
            }
        }
        if (best_dist_out) { best_dist_out[iL] = bestDist; best_idx_out[iL] = (bestDist < 100) ? (int)bestIdxR : -1; }
        if (bestDist < thOrbDist) {
            const float uR0 = kR[bestIdxR].x;
            const float s = inv_sf[kp.octave];
            const float suL = std::round(kp.x * s), svL = std::round(kp.y * s), suR0 = std::round(uR0 * s);
            const int w = 5, L = 5;
            const uint8_t* imL = pyrL[kp.octave];
            const uint8_t* imR = pyrR[kp.octave];
            const int stride = lw[kp.octave];
            int bestSad = INT_MAX, bestInc = 0;
            float dists[2 * L + 1];
            const float iniu = suR0 + L - w, endu = suR0 + L + w + 1;
            if (iniu < 0 || endu >= lw[kp.octave]) continue;
            const int y0 = (int)(svL - w), xl0 = (int)(suL - w);
            for (int inc = -L; inc <= L; ++inc) {
                const int xr0 = (int)(suR0 + inc - w);
                int sad = 0;
                for (int dy = 0; dy < 2 * w + 1; ++dy)
                    for (int dx = 0; dx < 2 * w + 1; ++dx)
                        sad += std::abs((int)imL[(size_t)(y0 + dy) * stride + xl0 + dx] -
                                        (int)imR[(size_t)(y0 + dy) * stride + xr0 + dx]);
                float dist = (float)sad;
                if (dist < (float)bestSad) { bestSad = (int)dist; bestInc = inc; }
                dists[L + inc] = dist;
            }
            if (bestInc == -L || bestInc == L) continue;
            const float d1 = dists[L + bestInc - 1], d2 = dists[L + bestInc], d3 = dists[L + bestInc + 1];
            const float delta = (d1 - d3) / (2.0f * (d1 + d3 - 2.0f * d2));
            if (delta < -1 || delta > 1) continue;
            float bestuR = sf[kp.octave] * ((float)suR0 + (float)bestInc + delta);
            float disparity = uL - bestuR;
            if (disparity >= minD && disparity < maxD) {
                if (disparity <= 0) { disparity = 0.01; bestuR = uL - 0.01; }
                depth[iL] = mbf / disparity;
                uRight[iL] = bestuR;
                distIdx.push_back(std::make_pair(bestSad, iL));
            }
        }
    }
    if (distIdx.empty()) return 0;   // the reference reads element 0 of an empty vector here
    std::sort(distIdx.begin(), distIdx.end());
    const float median = (float)distIdx[distIdx.size() / 2].first;
    const float thDist = 1.5f * 1.4f * median;
    int kept = (int)distIdx.size();
    for (int i = (int)distIdx.size() - 1; i >= 0; --i) {
        if (distIdx[i].first < thDist) break;
        uRight[distIdx[i].second] = -1;
        depth[distIdx[i].second] = -1;
        --kept;
    }
    return kept;
}

// ---- 64x48 keypoint grid and windowed candidate query ---------------------
// Frame::AssignFeaturesToGrid (Frame.cc:468-506) + Frame::GetFeaturesInArea (:792-858),
// mono / rectified case (Nleft == -1).  kps are the (undistorted) keypoints.
struct Grid {
    float minX, minY, invW, invH;
    std::vector<std::vector<int>> cells;   // 64 columns x 48 rows, [col*48 + row]
};
static bool grid_pos(const Grid& g, float x, float y, int* px, int* py) {
    *px = (int)std::round((x - g.minX) * g.invW);
    *py = (int)std::round((y - g.minY) * g.invH);
    return !(*px < 0 || *px >= 64 || *py < 0 || *py >= 48);
}
void* orc_grid_build(const void* kps_, int n, float minX, float maxX, float minY, float maxY) {
    const KeyPt* k = (const KeyPt*)kps_;
    Grid* g = new Grid;
    g->minX = minX; g->minY = minY;
    g->invW = (float)64 / (maxX - minX);
    g->invH = (float)48 / (maxY - minY);
    g->cells.resize(64 * 48);
    for (int i = 0; i < n; ++i) {
        int cx, cy;
        if (grid_pos(*g, k[i].x, k[i].y, &cx, &cy)) g->cells[cx * 48 + cy].push_back(i);
    }
    return g;
}
void orc_grid_destroy(void* g) { delete (Grid*)g; }
// The grid as CSR: cell_start[64*48 + 1] (cell = column * 48 + row), cell_idx[] = keypoint indices in
// insertion order.  Returns the number of keypoints inside the bounds.
int orc_grid_csr(void* g_, int* cell_start, int* cell_idx, int cap) {
    const Grid& g = *(Grid*)g_;
    int n = 0;
    for (int c = 0; c < 64 * 48; ++c) {
        cell_start[c] = n;
        for (int idx : g.cells[c]) { if (n < cap) cell_idx[n] = idx; ++n; }
    }
    cell_start[64 * 48] = n;
    return n;
}
int orc_features_in_area(void* g_, const void* kps_, float x, float y, float r, int minLevel, int maxLevel, int* out, int cap) {
    const Grid& g = *(Grid*)g_;
    const KeyPt* k = (const KeyPt*)kps_;
    int n = 0;
    const float factorX = r, factorY = r;
    const int minCX = std::max(0, (int)std::floor((x - g.minX - factorX) * g.invW));
    if (minCX >= 64) return 0;
    const int maxCX = std::min(63, (int)std::ceil((x - g.minX + factorX) * g.invW));
    if (maxCX < 0) return 0;
    const int minCY = std::max(0, (int)std::floor((y - g.minY - factorY) * g.invH));
    if (minCY >= 48) return 0;
    const int maxCY = std::min(47, (int)std::ceil((y - g.minY + factorY) * g.invH));
    if (maxCY < 0) return 0;
    const bool checkLevels = (minLevel > 0) || (maxLevel >= 0);
    for (int ix = minCX; ix <= maxCX; ++ix)
        for (int iy = minCY; iy <= maxCY; ++iy)
            for (int idx : g.cells[ix * 48 + iy]) {
                const KeyPt& kp = k[idx];
                if (checkLevels) {
                    if (kp.octave < minLevel) continue;
                    if (maxLevel >= 0 && kp.octave > maxLevel) continue;
                }
                const float dx = kp.x - x, dy = kp.y - y;
                if (std::fabs(dx) < factorX && std::fabs(dy) < factorY) {
                    if (n < cap) out[n] = idx;
                    ++n;
                }
            }
    return n;
}

// Core of ORBmatcher::SearchByProjection(Frame&, vector<MapPoint*>&, th, ...) for the
// mono / rectified-stereo case (ORBmatcher.cc:42-147): greedy in query order.
// Per query q: projected (x,y), predicted level, search radius r (already multiplied by
// th and the level scale), descriptor, optional projected right coordinate (xr, used when
// uRight[idx] > 0).  `taken[idx]` marks frame keypoints that already hold a map point
// with observations; accepted matches set it.  match_out[q] = keypoint index or -1.
int orc_search_by_projection(void* grid, const void* kps_, const uint8_t* fdesc, const float* uRight, int nkp,
                             const float* qx, const float* qy, const float* qxr, const int* qlevel,
                             const float* qradius, const uint8_t* qdesc, int nq,
                             uint8_t* taken, float nnratio, int* match_out) {
    const KeyPt* k = (const KeyPt*)kps_;
    std::vector<int> cand(nkp);
    int nmatches = 0;
    for (int q = 0; q < nq; ++q) {
        match_out[q] = -1;
        const int lvl = qlevel[q];
        const float r = qradius[q];
        int nc = orc_features_in_area(grid, kps_, qx[q], qy[q], r, lvl - 1, lvl, cand.data(), nkp);
        if (nc == 0) continue;
        int best = 256, bestLevel = -1, best2 = 256, bestLevel2 = -1, bestIdx = -1;
        for (int c = 0; c < nc; ++c) {
            const int idx = cand[c];
            if (taken[idx]) continue;
            if (uRight && uRight[idx] > 0) {
                const float er = std::fabs(qxr[q] - uRight[idx]);
                if (er > r) continue;
            }
            const int d = hamming256(qdesc + (size_t)q * 32, fdesc + (size_t)idx * 32);
            if (d < best) {
                best2 = best; best = d; bestLevel2 = bestLevel; bestLevel = k[idx].octave; bestIdx = idx;
            } else if (d < best2) {
                bestLevel2 = k[idx].octave; best2 = d;
            }
        }
        if (best <= 100) {
            if (bestLevel == bestLevel2 && best > nnratio * best2) continue;
            if (bestLevel != bestLevel2 || best <= nnratio * best2) {
                match_out[q] = bestIdx;
                taken[bestIdx] = 1;
                ++nmatches;
            }
        }
    }
    return nmatches;
}

// Distance core of SearchForTriangulation (ORBmatcher.cc:1051-1167 without the
// geometric gates): for each idx1 of a bucket, the LAST idx2 of the paired bucket with the
// smallest distance <= 50 (the running test is `dist > best -> skip`, so a later equal
// distance replaces an earlier one).  CSR buckets: off1/off2 have nb+1 entries.
void orc_bucket_best(const uint8_t* d1, const int* idx1, const int* off1,
                     const uint8_t* d2, const int* idx2, const int* off2, int nb,
                     int* best_idx2, int* best_dist) {
    for (int b = 0; b < nb; ++b)
        for (int i = off1[b]; i < off1[b + 1]; ++i) {
            int best = 50, bi = -1;
            for (int j = off2[b]; j < off2[b + 1]; ++j) {
                int d = hamming256(d1 + (size_t)idx1[i] * 32, d2 + (size_t)idx2[j] * 32);
                if (d > 50 || d > best) continue;
                best = d; bi = idx2[j];
            }
            best_idx2[i] = bi; best_dist[i] = best;
        }
}


// ---- rotation-consistency histogram (ORBmatcher.cc:36, :2096-2137) ----------
static const int HISTO_LENGTH = 30;
static void compute_three_maxima(std::vector<int>* histo, const int L, int& ind1, int& ind2, int& ind3) {
    int max1 = 0, max2 = 0, max3 = 0;
    for (int i = 0; i < L; i++) {
        const int s = (int)histo[i].size();
        if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
        else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
        else if (s > max3) { max3 = s; ind3 = i; }
    }
    if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
    else if (max3 < 0.1f * (float)max1) { ind3 = -1; }
}
static int rot_bin(float a1, float a2, float factor) {
    float rot = a1 - a2;
    if (rot < 0.0) rot += 360.0f;
    int bin = (int)std::round(rot * factor);
    if (bin == HISTO_LENGTH) bin = 0;
    return bin;
}

// Best-only projection searches (ORBmatcher.cc:1761-1972, :1974-2094, :476-693, :1409-1524):
// per query the window candidates in GetFeaturesInArea order, skip occupied keypoints, stereo
// gate on uRight > 0, strict-< best, accept best <= max_dist, optional rotation pruning.
// kp_match[idx] = owning query (the reference's mvpMapPoints / vpMatched slot).
int orc_search_window_best(void* grid, const void* kps_, const uint8_t* fdesc, const float* uRight, int nkp,
                           const float* qx, const float* qy, const float* qxr, const float* qradius,
                           const int* qmin, const int* qmax, const uint8_t* qdesc, int nq, float max_dist,
                           uint8_t* taken, const uint8_t* marks, const float* qangle, float factor, int* kp_match) {
    const KeyPt* k = (const KeyPt*)kps_;
    std::vector<int> cand(nkp), rotHist[HISTO_LENGTH];
    for (int i = 0; i < nkp; ++i) kp_match[i] = -1;
    int nmatches = 0;
    for (int q = 0; q < nq; ++q) {
        const float radius = qradius[q];
        const int nc = orc_features_in_area(grid, kps_, qx[q], qy[q], radius, qmin[q], qmax[q], cand.data(), nkp);
        if (nc == 0) continue;
        int bestDist = 256, bestIdx2 = -1;
        for (int c = 0; c < nc; ++c) {
            const int i2 = cand[c];
            if (taken && taken[i2]) continue;
            if (uRight && uRight[i2] > 0) {
                const float er = std::fabs(qxr[q] - uRight[i2]);
                if (er > radius) continue;
            }
            const int dist = hamming256(qdesc + (size_t)q * 32, fdesc + (size_t)i2 * 32);
            if (dist < bestDist) { bestDist = dist; bestIdx2 = i2; }
        }
        if (bestIdx2 >= 0 && bestDist <= max_dist) {
            kp_match[bestIdx2] = q;
            if (taken) taken[bestIdx2] = marks ? marks[q] : 1;
            nmatches++;
            if (qangle) rotHist[rot_bin(qangle[q], k[bestIdx2].angle, factor)].push_back(bestIdx2);
        }
    }
    if (qangle) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        compute_three_maxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++)
            if (i != ind1 && i != ind2 && i != ind3)
                for (size_t j = 0; j < rotHist[i].size(); j++) { kp_match[rotHist[i][j]] = -1; nmatches--; }
    }
    return nmatches;
}

// Fuse(KeyFrame*, const vector<MapPoint*>&, th, bRight), matching part (ORBmatcher.cc:1315-1382).
int orc_fuse_search(void* grid, const void* kps_, const uint8_t* fdesc, const float* uRight, int nkp,
                    const float* qu, const float* qv, const float* qur, const float* qradius, const int* qlevel,
                    const uint8_t* qdesc, int nq, const float* invLevelSigma2, int* match_out) {
    const KeyPt* k = (const KeyPt*)kps_;
    std::vector<int> cand(nkp);
    int nFused = 0;
    for (int q = 0; q < nq; ++q) {
        match_out[q] = -1;
        const int nPredictedLevel = qlevel[q];
        const int nc = orc_features_in_area(grid, kps_, qu[q], qv[q], qradius[q], -1, -1, cand.data(), nkp);
        int bestDist = 256, bestIdx = -1;
        for (int c = 0; c < nc; ++c) {
            const int idx = cand[c];
            const KeyPt& kp = k[idx];
            const int kpLevel = kp.octave;
            if (kpLevel < nPredictedLevel - 1 || kpLevel > nPredictedLevel) continue;
            if (uRight && uRight[idx] >= 0) {
                const float ex = qu[q] - kp.x, ey = qv[q] - kp.y, er = qur[q] - uRight[idx];
                const float e2 = ex * ex + ey * ey + er * er;
                if (e2 * invLevelSigma2[kpLevel] > 7.8) continue;
            } else {
                const float ex = qu[q] - kp.x, ey = qv[q] - kp.y;
                const float e2 = ex * ex + ey * ey;
                if (e2 * invLevelSigma2[kpLevel] > 5.99) continue;
            }
            const int dist = hamming256(qdesc + (size_t)q * 32, fdesc + (size_t)idx * 32);
            if (dist < bestDist) { bestDist = dist; bestIdx = idx; }
        }
        if (bestDist <= 50) { match_out[q] = bestIdx; nFused++; }
    }
    return nFused;
}

// One direction of SearchBySim3 (ORBmatcher.cc:1608-1641): vnMatch[source[q]] = best keypoint.
static void sim3_direction(void* grid, const KeyPt* k, const uint8_t* fdesc, int nkp, const float* qu, const float* qv,
                           const float* qradius, const int* qlevel, const uint8_t* qdesc, const int* source, int nq,
                           std::vector<int>& vnMatch) {
    std::vector<int> cand(nkp);
    for (int q = 0; q < nq; ++q) {
        const int nPredictedLevel = qlevel[q];
        const int nc = orc_features_in_area(grid, k, qu[q], qv[q], qradius[q], -1, -1, cand.data(), nkp);
        if (nc == 0) continue;
        int bestDist = INT_MAX, bestIdx = -1;
        for (int c = 0; c < nc; ++c) {
            const int idx = cand[c];
            if (k[idx].octave < nPredictedLevel - 1 || k[idx].octave > nPredictedLevel) continue;
            const int dist = hamming256(qdesc + (size_t)q * 32, fdesc + (size_t)idx * 32);
            if (dist < bestDist) { bestDist = dist; bestIdx = idx; }
        }
        if (bestDist <= 100) vnMatch[source[q]] = bestIdx;
    }
}
int orc_search_by_sim3(void* grid1, const void* kps1, const uint8_t* desc1, int n1,
                       void* grid2, const void* kps2, const uint8_t* desc2, int n2,
                       const float* u12, const float* v12, const float* r12, const int* l12, const uint8_t* d12, const int* src1, int nq1,
                       const float* u21, const float* v21, const float* r21, const int* l21, const uint8_t* d21, const int* src2, int nq2,
                       int* matches12) {
    std::vector<int> vnMatch1(n1, -1), vnMatch2(n2, -1);
    sim3_direction(grid2, (const KeyPt*)kps2, desc2, n2, u12, v12, r12, l12, d12, src1, nq1, vnMatch1);
    sim3_direction(grid1, (const KeyPt*)kps1, desc1, n1, u21, v21, r21, l21, d21, src2, nq2, vnMatch2);
    int nFound = 0;
    for (int i1 = 0; i1 < n1; i1++) {
        matches12[i1] = -1;
        int idx2 = vnMatch1[i1];
        if (idx2 >= 0) {
            int idx1 = vnMatch2[idx2];
            if (idx1 == i1) { matches12[i1] = idx2; nFound++; }
        }
    }
    return nFound;
}

// SearchForInitialization (ORBmatcher.cc:695-808).  prev = vbPrevMatched as {x, y} pairs.
int orc_search_for_initialization(const void* kps1_, const uint8_t* desc1, int n1, void* grid2, const void* kps2_,
                                  const uint8_t* desc2, int n2, float* prev, int windowSize, float nnratio,
                                  int checkOrientation, int* vnMatches12) {
    const KeyPt* k1 = (const KeyPt*)kps1_;
    const KeyPt* k2 = (const KeyPt*)kps2_;
    int nmatches = 0;
    for (int i = 0; i < n1; ++i) vnMatches12[i] = -1;
    std::vector<int> rotHist[HISTO_LENGTH];
    const float factor = 1.0f / HISTO_LENGTH;
    std::vector<int> vMatchedDistance(n2, INT_MAX), vnMatches21(n2, -1), cand(n2);
    for (int i1 = 0; i1 < n1; i1++) {
        const int level1 = k1[i1].octave;
        if (level1 > 0) continue;
        const int nc = orc_features_in_area(grid2, kps2_, prev[2 * i1], prev[2 * i1 + 1], (float)windowSize, level1, level1, cand.data(), n2);
        if (nc == 0) continue;
        int bestDist = INT_MAX, bestDist2 = INT_MAX, bestIdx2 = -1;
        for (int c = 0; c < nc; ++c) {
            const int i2 = cand[c];
            const int dist = hamming256(desc1 + (size_t)i1 * 32, desc2 + (size_t)i2 * 32);
            if (vMatchedDistance[i2] <= dist) continue;
            if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestIdx2 = i2; }
            else if (dist < bestDist2) bestDist2 = dist;
        }
        if (bestDist <= 50) {
            if (bestDist < (float)bestDist2 * nnratio) {
                if (vnMatches21[bestIdx2] >= 0) { vnMatches12[vnMatches21[bestIdx2]] = -1; nmatches--; }
                vnMatches12[i1] = bestIdx2;
                vnMatches21[bestIdx2] = i1;
                vMatchedDistance[bestIdx2] = bestDist;
                nmatches++;
                if (checkOrientation) rotHist[rot_bin(k1[i1].angle, k2[bestIdx2].angle, factor)].push_back(i1);
            }
        }
    }
    if (checkOrientation) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        compute_three_maxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (size_t j = 0; j < rotHist[i].size(); j++) {
                int idx1 = rotHist[i][j];
                if (vnMatches12[idx1] >= 0) { vnMatches12[idx1] = -1; nmatches--; }
            }
        }
    }
    for (int i1 = 0; i1 < n1; i1++)
        if (vnMatches12[i1] >= 0) { prev[2 * i1] = k2[vnMatches12[i1]].x; prev[2 * i1 + 1] = k2[vnMatches12[i1]].y; }
    return nmatches;
}

// SearchByBoW(KeyFrame*, Frame&, ...) (ORBmatcher.cc:235-474).  The two FeatureVectors arrive as
// CSR over their SHARED nodes (the reference's lower_bound walk visits exactly those).
int orc_search_by_bow(const uint8_t* descKF, const float* angleKF, const uint8_t* validKF, const int* idxKF, const int* offKF,
                      const uint8_t* descF, const float* angleF, int nF, int Nleft, const int* idxF, const int* offF,
                      int nb, float nnratio, int checkOrientation, int* matchF) {
    for (int i = 0; i < nF; ++i) matchF[i] = -1;
    std::vector<int> rotHist[HISTO_LENGTH];
    const float factor = HISTO_LENGTH / 360.0f;
    int nmatches = 0;
    for (int b = 0; b < nb; ++b)
        for (int iKF = offKF[b]; iKF < offKF[b + 1]; iKF++) {
            const int realIdxKF = idxKF[iKF];
            if (!validKF[realIdxKF]) continue;
            const uint8_t* dKF = descKF + (size_t)realIdxKF * 32;
            int bestDist1 = 256, bestIdxF = -1, bestDist2 = 256, bestDist1R = 256, bestIdxFR = -1, bestDist2R = 256;
            for (int iF = offF[b]; iF < offF[b + 1]; iF++) {
                const int realIdxF = idxF[iF];
                if (matchF[realIdxF] >= 0) continue;
                const int dist = hamming256(dKF, descF + (size_t)realIdxF * 32);
                if (Nleft == -1) {
                    if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdxF = realIdxF; }
                    else if (dist < bestDist2) bestDist2 = dist;
                } else {
                    if (realIdxF < Nleft && dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdxF = realIdxF; }
                    else if (realIdxF < Nleft && dist < bestDist2) bestDist2 = dist;
                    if (realIdxF >= Nleft && dist < bestDist1R) { bestDist2R = bestDist1R; bestDist1R = dist; bestIdxFR = realIdxF; }
                    else if (realIdxF >= Nleft && dist < bestDist2R) bestDist2R = dist;
                }
            }
            if (bestDist1 <= 50) {
                if (static_cast<float>(bestDist1) < nnratio * static_cast<float>(bestDist2)) {
                    matchF[bestIdxF] = realIdxKF;
                    if (checkOrientation) rotHist[rot_bin(angleKF[realIdxKF], angleF[bestIdxF], factor)].push_back(bestIdxF);
                    nmatches++;
                }
                if (bestDist1R <= 50) {
                    matchF[bestIdxFR] = realIdxKF;
                    if (checkOrientation) rotHist[rot_bin(angleKF[realIdxKF], angleF[bestIdxFR], factor)].push_back(bestIdxFR);
                    nmatches++;
                }
            }
        }
    if (checkOrientation) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        compute_three_maxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (size_t j = 0; j < rotHist[i].size(); j++) { matchF[rotHist[i][j]] = -1; nmatches--; }
        }
    }
    return nmatches;
}

// SearchByBoW(KeyFrame*, KeyFrame*, ...) (ORBmatcher.cc:824-966).
int orc_search_by_bow_keyframes(const uint8_t* desc1, const float* angle1, const uint8_t* valid1, int n1, const int* idx1, const int* off1,
                                const uint8_t* desc2, const float* angle2, const uint8_t* valid2, int n2, const int* idx2, const int* off2,
                                int nb, float nnratio, int checkOrientation, int* matches12) {
    for (int i = 0; i < n1; ++i) matches12[i] = -1;
    std::vector<bool> vbMatched2(n2, false);
    std::vector<int> rotHist[HISTO_LENGTH];
    const float factor = 1.0f / HISTO_LENGTH;
    int nmatches = 0;
    for (int b = 0; b < nb; ++b)
        for (int i1 = off1[b]; i1 < off1[b + 1]; i1++) {
            const int id1 = idx1[i1];
            if (!valid1[id1]) continue;
            int bestDist1 = 256, bestIdx2 = -1, bestDist2 = 256;
            for (int i2 = off2[b]; i2 < off2[b + 1]; i2++) {
                const int id2 = idx2[i2];
                if (vbMatched2[id2] || !valid2[id2]) continue;
                const int dist = hamming256(desc1 + (size_t)id1 * 32, desc2 + (size_t)id2 * 32);
                if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdx2 = id2; }
                else if (dist < bestDist2) bestDist2 = dist;
            }
            if (bestDist1 < 50) {
                if (static_cast<float>(bestDist1) < nnratio * static_cast<float>(bestDist2)) {
                    matches12[id1] = bestIdx2;
                    vbMatched2[bestIdx2] = true;
                    if (checkOrientation) rotHist[rot_bin(angle1[id1], angle2[bestIdx2], factor)].push_back(id1);
                    nmatches++;
                }
            }
        }
    if (checkOrientation) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        compute_three_maxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (size_t j = 0; j < rotHist[i].size(); j++) { matches12[rotHist[i][j]] = -1; nmatches--; }
        }
    }
    return nmatches;
}

// SearchForTriangulation (ORBmatcher.cc:968-1212), pinhole cameras, no second camera model;
// Pinhole::epipolarConstrain (Pinhole.cpp:118-141) with the fundamental matrix given (row-major).
int orc_search_for_triangulation(const void* kps1_, const uint8_t* desc1, const uint8_t* free1, const float* uRight1, int n1,
                                 const int* idx1, const int* off1, const void* kps2_, const uint8_t* desc2,
                                 const uint8_t* free2, const float* uRight2, int n2, const int* idx2, const int* off2, int nb,
                                 const float* F12, float epx, float epy, const float* scaleFactors2, const float* levelSigma2_2,
                                 int bOnlyStereo, int bCoarse, int checkOrientation, int* vMatches12) {
    const KeyPt* k1 = (const KeyPt*)kps1_;
    const KeyPt* k2 = (const KeyPt*)kps2_;
    (void)n2;
    for (int i = 0; i < n1; ++i) vMatches12[i] = -1;
    std::vector<int> rotHist[HISTO_LENGTH];
    const float factor = 1.0f / HISTO_LENGTH;
    int nmatches = 0;
    for (int b = 0; b < nb; ++b)
        for (int i1 = off1[b]; i1 < off1[b + 1]; i1++) {
            const int id1 = idx1[i1];
            if (!free1[id1]) continue;
            const bool bStereo1 = uRight1 && uRight1[id1] >= 0;
            if (bOnlyStereo && !bStereo1) continue;
            const KeyPt& kp1 = k1[id1];
            int bestDist = 50, bestIdx2 = -1;
            for (int i2 = off2[b]; i2 < off2[b + 1]; i2++) {
                const int id2 = idx2[i2];
                if (!free2[id2]) continue;
                const bool bStereo2 = uRight2 && uRight2[id2] >= 0;
                if (bOnlyStereo && !bStereo2) continue;
                const int dist = hamming256(desc1 + (size_t)id1 * 32, desc2 + (size_t)id2 * 32);
                if (dist > 50 || dist > bestDist) continue;
                const KeyPt& kp2 = k2[id2];
                if (!bStereo1 && !bStereo2) {
                    const float distex = epx - kp2.x, distey = epy - kp2.y;
                    if (distex * distex + distey * distey < 100 * scaleFactors2[kp2.octave]) continue;
                }
                bool ok = bCoarse != 0;
                if (!ok) {
                    const float a = kp1.x * F12[0] + kp1.y * F12[3] + F12[6];
                    const float bb = kp1.x * F12[1] + kp1.y * F12[4] + F12[7];
                    const float c = kp1.x * F12[2] + kp1.y * F12[5] + F12[8];
                    const float num = a * kp2.x + bb * kp2.y + c;
                    const float den = a * a + bb * bb;
                    if (den == 0) ok = false;
                    else { const float dsqr = num * num / den; ok = dsqr < 3.84 * levelSigma2_2[kp2.octave]; }
                }
                if (ok) { bestIdx2 = id2; bestDist = dist; }
            }
            if (bestIdx2 >= 0) {
                vMatches12[id1] = bestIdx2;
                nmatches++;
                if (checkOrientation) rotHist[rot_bin(kp1.angle, k2[bestIdx2].angle, factor)].push_back(id1);
            }
        }
    if (checkOrientation) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        compute_three_maxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (size_t j = 0; j < rotHist[i].size(); j++) { vMatches12[rotHist[i][j]] = -1; nmatches--; }
        }
    }
    return nmatches;
}


// DBoW2::TemplatedVocabulary::transform(feature, word_id, weight, nid, levelsup)
// (Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1200-1243) on a tree given as child lists.
// FORB::distance (FORB.cpp:90-110) is the same bit-hack popcount as DescriptorDistance.
void orc_bow_transform(const int* child_off, const int* child_idx, const uint8_t* node_desc, const int* word_id,
                       const double* weight, int m_L, const uint8_t* desc, int n, int levelsup, int* word_out,
                       double* weight_out, int* nid_out) {
    for (int f = 0; f < n; ++f) {
        const uint8_t* feature = desc + (size_t)f * 32;
        const int nid_level = m_L - levelsup;
        int nid = 0;                       // root when nid_level <= 0 (the reference leaves it unset otherwise)
        int final_id = 0, current_level = 0;
        do {
            ++current_level;
            const int b = child_off[final_id], e = child_off[final_id + 1];
            final_id = child_idx[b];
            double best_d = hamming256(feature, node_desc + (size_t)final_id * 32);
            for (int c = b + 1; c < e; ++c) {
                const int id = child_idx[c];
                const double d = hamming256(feature, node_desc + (size_t)id * 32);
                if (d < best_d) { best_d = d; final_id = id; }
            }
            if (current_level == nid_level) nid = final_id;
        } while (child_off[final_id] != child_off[final_id + 1]);
        word_out[f] = word_id[final_id];
        weight_out[f] = weight[final_id];
        nid_out[f] = nid;
    }
}


// MapPoint::ComputeDistinctiveDescriptors (MapPoint.cc:472-521) for one map point after another.
void orc_distinctive_descriptors(const uint8_t* desc, const int* off, int npoints, int* best_idx, int* best_median) {
    for (int p = 0; p < npoints; ++p) {
        const uint8_t* D = desc + (size_t)off[p] * 32;
        const size_t N = (size_t)(off[p + 1] - off[p]);
        if (N == 0) { best_idx[p] = -1; best_median[p] = INT_MAX; continue; }
        std::vector<std::vector<float>> Distances(N, std::vector<float>(N, 0.f));
        for (size_t i = 0; i < N; i++) {
            Distances[i][i] = 0;
            for (size_t j = i + 1; j < N; j++) {
                int distij = hamming256(D + i * 32, D + j * 32);
                Distances[i][j] = (float)distij;
                Distances[j][i] = (float)distij;
            }
        }
        int BestMedian = INT_MAX, BestIdx = 0;
        for (size_t i = 0; i < N; i++) {
            std::vector<int> vDists(Distances[i].begin(), Distances[i].end());
            std::sort(vDists.begin(), vDists.end());
            int median = vDists[(size_t)(0.5 * (N - 1))];
            if (median < BestMedian) { BestMedian = median; BestIdx = (int)i; }
        }
        best_idx[p] = BestIdx; best_median[p] = BestMedian;
    }
}

// ---- second-camera (Nleft != -1) matcher paths --------------------------------------------------------------
// A fisheye stereo Frame keeps the left keypoints in mvKeys / mGrid and the right ones in mvKeysRight / mGridRight;
// indices >= Nleft in mvpMapPoints / mDescriptors are right-camera features (Frame.cc:1265-1283).  Here every
// camera has its own keypoint array, descriptor array and grid; `taken` / `kp_query` run over the combined index
// (left first, right + nL).

// SearchByProjection(Frame&, const vector<MapPoint*>&, th, ...) with both passes (ORBmatcher.cc:42-216).
// Query q: left projection {lx, ly, lradius, llevel} used when lradius > 0 (mbTrackInView; lradius already holds
// r * th * scale), right projection {rx, ry, rradius, rlevel} used when rradius > 0 (mbTrackInViewR and level != -1;
// r * scale, without th, :151-154).  l2r / r2l = mvLeftToRightMatch / mvRightToLeftMatch.  marks[q] = the map point has
// observations (an assigned keypoint is skipped by later queries only then, :88-90); kp_query[idx] = last query
// assigned to keypoint idx.
int orc_search_by_projection_stereo(void* gridL, const void* kpsL_, const uint8_t* descL, int nL,
                                    void* gridR, const void* kpsR_, const uint8_t* descR, int nR,
                                    const float* lx, const float* ly, const float* lradius, const int* llevel,
                                    const float* rx, const float* ry, const float* rradius, const int* rlevel,
                                    const uint8_t* qdesc, int nq, const int* l2r, const int* r2l,
                                    uint8_t* taken, const uint8_t* marks, float nnratio, int* kp_query) {
    const KeyPt* kL = (const KeyPt*)kpsL_;
    const KeyPt* kR = (const KeyPt*)kpsR_;
    std::vector<int> cand(std::max(nL, nR) + 1);
    for (int i = 0; i < nL + nR; ++i) kp_query[i] = -1;
    int nmatches = 0;
    for (int q = 0; q < nq; ++q) {
        const uint8_t mark = marks ? marks[q] : 1;
        if (lradius[q] > 0) {
            const int lvl = llevel[q];
            const int nc = orc_features_in_area(gridL, kpsL_, lx[q], ly[q], lradius[q], lvl - 1, lvl, cand.data(), nL);
            if (nc > 0) {
                int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
                for (int c = 0; c < nc; ++c) {
                    const int idx = cand[c];
                    if (taken[idx]) continue;
                    const int dist = hamming256(qdesc + (size_t)q * 32, descL + (size_t)idx * 32);
                    if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestLevel2 = bestLevel; bestLevel = kL[idx].octave; bestIdx = idx; }
                    else if (dist < bestDist2) { bestLevel2 = kL[idx].octave; bestDist2 = dist; }
                }
                if (bestDist <= 100) {
                    // :125-126 -- this `continue` leaves the map-point iteration: a failed ratio test in the left
                    // image also skips the right-camera pass (:146-213) of this map point
                    if (bestLevel == bestLevel2 && bestDist > nnratio * bestDist2) continue;
                    if (bestLevel != bestLevel2 || bestDist <= nnratio * bestDist2) {
                        kp_query[bestIdx] = q; taken[bestIdx] = mark;
                        if (l2r && l2r[bestIdx] != -1) { kp_query[l2r[bestIdx] + nL] = q; taken[l2r[bestIdx] + nL] = mark; nmatches++; }
                        nmatches++;
                    }
                }
            }
        }
        if (nR > 0 && rradius[q] > 0) {
            const int lvl = rlevel[q];
            const int nc = orc_features_in_area(gridR, kpsR_, rx[q], ry[q], rradius[q], lvl - 1, lvl, cand.data(), nR);
            if (nc == 0) continue;
            int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
            for (int c = 0; c < nc; ++c) {
                const int idx = cand[c];
                if (taken[idx + nL]) continue;
                const int dist = hamming256(qdesc + (size_t)q * 32, descR + (size_t)idx * 32);
                if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestLevel2 = bestLevel; bestLevel = kR[idx].octave; bestIdx = idx; }
                else if (dist < bestDist2) { bestLevel2 = kR[idx].octave; bestDist2 = dist; }
            }
            if (bestDist <= 100) {
                if (bestLevel == bestLevel2 && bestDist > nnratio * bestDist2) continue;
                if (r2l && r2l[bestIdx] != -1) { kp_query[r2l[bestIdx]] = q; taken[r2l[bestIdx]] = mark; nmatches++; }
                kp_query[bestIdx + nL] = q; taken[bestIdx + nL] = mark;
                nmatches++;
            }
        }
    }
    return nmatches;
}

// SearchByProjection(Frame& Cur, const Frame& Last, th, bMono) with the right-image pass and the shared rotation
// histogram (ORBmatcher.cc:1761-1972).  Query q (a map point of the last frame): left window {lx, ly, radius,
// lmin, lmax}; lradius <= 0 = the point is skipped before either pass (:1795-1803).  An empty left window skips the
// right pass as well (:1820-1821).  Right window {rx, ry, radius, rmin, rmax} on the right camera's grid.
int orc_search_window_best_stereo(void* gridL, const void* kpsL_, const uint8_t* descL, int nL,
                                  void* gridR, const void* kpsR_, const uint8_t* descR, int nR,
                                  const float* lx, const float* ly, const float* lradius, const int* lmin, const int* lmax,
                                  const float* rx, const float* ry, const float* rradius, const int* rmin, const int* rmax,
                                  const uint8_t* qdesc, int nq, float max_dist, uint8_t* taken, const uint8_t* marks,
                                  const float* qangle, float factor, int* kp_query) {
    const KeyPt* kL = (const KeyPt*)kpsL_;
    const KeyPt* kR = (const KeyPt*)kpsR_;
    std::vector<int> cand(std::max(nL, nR) + 1), rotHist[HISTO_LENGTH];
    for (int i = 0; i < nL + nR; ++i) kp_query[i] = -1;
    int nmatches = 0;
    for (int q = 0; q < nq; ++q) {
        if (!(lradius[q] > 0)) continue;
        const uint8_t mark = marks ? marks[q] : 1;
        int nc = orc_features_in_area(gridL, kpsL_, lx[q], ly[q], lradius[q], lmin[q], lmax[q], cand.data(), nL);
        if (nc == 0) continue;
        int bestDist = 256, bestIdx2 = -1;
        for (int c = 0; c < nc; ++c) {
            const int i2 = cand[c];
            if (taken[i2]) continue;
            const int dist = hamming256(qdesc + (size_t)q * 32, descL + (size_t)i2 * 32);
            if (dist < bestDist) { bestDist = dist; bestIdx2 = i2; }
        }
        if (bestIdx2 >= 0 && bestDist <= max_dist) {
            kp_query[bestIdx2] = q; taken[bestIdx2] = mark;
            nmatches++;
            if (qangle) rotHist[rot_bin(qangle[q], kL[bestIdx2].angle, factor)].push_back(bestIdx2);
        }
        if (nR > 0) {
            nc = orc_features_in_area(gridR, kpsR_, rx[q], ry[q], rradius[q], rmin[q], rmax[q], cand.data(), nR);
            bestDist = 256; bestIdx2 = -1;
            for (int c = 0; c < nc; ++c) {
                const int i2 = cand[c];
                if (taken[i2 + nL]) continue;
                const int dist = hamming256(qdesc + (size_t)q * 32, descR + (size_t)i2 * 32);
                if (dist < bestDist) { bestDist = dist; bestIdx2 = i2; }
            }
            if (bestIdx2 >= 0 && bestDist <= max_dist) {
                kp_query[bestIdx2 + nL] = q; taken[bestIdx2 + nL] = mark;
                nmatches++;
                if (qangle) rotHist[rot_bin(qangle[q], kR[bestIdx2].angle, factor)].push_back(bestIdx2 + nL);
            }
        }
    }
    if (qangle) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        compute_three_maxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++)
            if (i != ind1 && i != ind2 && i != ind3)
                for (size_t j = 0; j < rotHist[i].size(); j++) { kp_query[rotHist[i][j]] = -1; nmatches--; }
    }
    return nmatches;
}

// SearchForTriangulation with the per-pair camera selection (ORBmatcher.cc:989-1006, :1099-1139).  Keypoint arrays
// are combined (left first); nLeft1 / nLeft2 = NLeft of each keyframe (-1: one camera, then only pose[0] / cams[0] are
// used and the epipole gate applies).  cams = {cam1 left, cam1 right, cam2 left, cam2 right} as {type, 8 params};
// poses = 4 x {R12 row-major (9), t12 (3)} in the order ll, lr, rl, rr (first letter = keyframe 1's camera).
int orc_search_for_triangulation_cameras(const void* kps1_, const uint8_t* desc1, const uint8_t* free1, const float* uRight1,
                                         int n1, int nLeft1, const int* idx1, const int* off1,
                                         const void* kps2_, const uint8_t* desc2, const uint8_t* free2, const float* uRight2,
                                         int n2, int nLeft2, const int* idx2, const int* off2, int nb,
                                         const int* camTypes, const float* camParams, const float* poses,
                                         float epx, float epy, const float* scaleFactors2, const float* levelSigma2_1,
                                         const float* levelSigma2_2, int bOnlyStereo, int bCoarse, int checkOrientation,
                                         int* vMatches12, double* margin1) {
    const KeyPt* k1 = (const KeyPt*)kps1_;
    const KeyPt* k2 = (const KeyPt*)kps2_;
    (void)n2;
    // margin1[id1] (may be null): the smallest rounding margin of any gate evaluated for keypoint id1 (see
    // camera_oracle.cpp); < 1 means the outcome for id1 may legitimately differ under binary32 rounding.
    camo::Cam cams[4];
    for (int c = 0; c < 4; ++c) cams[c] = camo::make_cam(camTypes[c], camParams + 8 * c);
    double posesD[48];
    for (int i = 0; i < 48; ++i) posesD[i] = poses[i];
    const bool twoCams = nLeft1 != -1 && nLeft2 != -1;
    for (int i = 0; i < n1; ++i) { vMatches12[i] = -1; if (margin1) margin1[i] = 1e300; }
    std::vector<int> rotHist[HISTO_LENGTH];
    const float factor = 1.0f / HISTO_LENGTH;
    int nmatches = 0;
    for (int b = 0; b < nb; ++b)
        for (int i1 = off1[b]; i1 < off1[b + 1]; i1++) {
            const int id1 = idx1[i1];
            if (!free1[id1]) continue;
            const bool bStereo1 = nLeft1 == -1 && uRight1 && uRight1[id1] >= 0;
            if (bOnlyStereo && !bStereo1) continue;
            const KeyPt& kp1 = k1[id1];
            const bool bRight1 = !(nLeft1 == -1 || id1 < nLeft1);
            int bestDist = 50, bestIdx2 = -1;
            for (int i2 = off2[b]; i2 < off2[b + 1]; i2++) {
                const int id2 = idx2[i2];
                if (!free2[id2]) continue;
                const bool bStereo2 = nLeft2 == -1 && uRight2 && uRight2[id2] >= 0;
                if (bOnlyStereo && !bStereo2) continue;
                const int dist = hamming256(desc1 + (size_t)id1 * 32, desc2 + (size_t)id2 * 32);
                if (dist > 50 || dist > bestDist) continue;
                const KeyPt& kp2 = k2[id2];
                const bool bRight2 = !(nLeft2 == -1 || id2 < nLeft2);
                if (!bStereo1 && !bStereo2 && nLeft1 == -1) {
                    const float distex = epx - kp2.x, distey = epy - kp2.y;
                    if (distex * distex + distey * distey < 100 * scaleFactors2[kp2.octave]) continue;
                }
                int sel = 0;
                if (twoCams) sel = (bRight1 ? 2 : 0) + (bRight2 ? 1 : 0);
                const camo::Cam& c1 = cams[twoCams && bRight1 ? 1 : 0];
                const camo::Cam& c2 = cams[2 + (twoCams && bRight2 ? 1 : 0)];
                const double* R12 = posesD + 12 * sel;
                const double* t12 = R12 + 9;
                bool ok = bCoarse != 0;
                if (!ok) {
                    double m = 1e300;
                    ok = camo::epipolar_constrain(c1, c2, kp1.x, kp1.y, kp2.x, kp2.y, R12, t12, levelSigma2_1[kp1.octave],
                                                  levelSigma2_2[kp2.octave], &m);
                    if (margin1 && m < margin1[id1]) margin1[id1] = m;
                }
                if (ok) { bestIdx2 = id2; bestDist = dist; }
            }
            if (bestIdx2 >= 0) {
                vMatches12[id1] = bestIdx2;
                nmatches++;
                if (checkOrientation) rotHist[rot_bin(kp1.angle, k2[bestIdx2].angle, factor)].push_back(id1);
            }
        }
    if (checkOrientation) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        compute_three_maxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (size_t j = 0; j < rotHist[i].size(); j++) { vMatches12[rotHist[i][j]] = -1; nmatches--; }
        }
    }
    return nmatches;
}

// Frame::ComputeStereoFishEyeMatches (Frame.cc:1286-1338): knnMatch(k = 2) of the lapping-area descriptors, ratio 0.7,
// TriangulateMatches, depth > 0.0001.  Returns nMatches.  margin[q] (may be null): rounding margin of left keypoint q's
// TriangulateMatches call (1e300 when none was made), see camera_oracle.cpp.
int orc_stereo_fisheye_matches(const void* kL_, const uint8_t* dL, int nL, int monoL, const void* kR_,
                               const uint8_t* dR, int nR, int monoR, const float* cam1, const float* cam2, const float* Rlr,
                               const float* tlr, const float* sigma2, int* l2r, int* r2l, float* depth, float* p3d,
                               double* margin) {
    const KeyPt* kL = (const KeyPt*)kL_;
    const KeyPt* kR = (const KeyPt*)kR_;
    for (int i = 0; i < nL; ++i) { l2r[i] = -1; depth[i] = -1.f; if (margin) margin[i] = 1e300; }
    for (int i = 0; i < nR; ++i) r2l[i] = -1;
    const camo::Cam c1 = camo::make_cam(camo::kKannalaBrandt8, cam1), c2 = camo::make_cam(camo::kKannalaBrandt8, cam2);
    double R[9], t[3];
    for (int i = 0; i < 9; ++i) R[i] = Rlr[i];
    for (int i = 0; i < 3; ++i) t[i] = tlr[i];
    int nMatches = 0;
    for (int q = monoL; q < nL; ++q) {
        // the two nearest right descriptors, the lower index first among equals (BFMatcher's order)
        int best = INT32_MAX, second = INT32_MAX, bestIdx = -1, secondIdx = -1;
        for (int tr = monoR; tr < nR; ++tr) {
            int d = 0;
            for (int k = 0; k < 32; ++k) d += __builtin_popcount((unsigned)(dL[(size_t)q * 32 + k] ^ dR[(size_t)tr * 32 + k]));
            if (d < best) { second = best; secondIdx = bestIdx; best = d; bestIdx = tr; }
            else if (d < second) { second = d; secondIdx = tr; }
        }
        if (secondIdx < 0) continue;
        if (!((float)best < (float)second * 0.7)) continue;
        double P[3] = {0, 0, 0}, m = 1e300;
        const double z = camo::triangulate_matches(c1, c2, kL[q].x, kL[q].y, kR[bestIdx].x, kR[bestIdx].y, R, t,
                                                   sigma2[kL[q].octave], sigma2[kR[bestIdx].octave], P, &m);
        if (margin) margin[q] = m;
        if (z > 0.0001) {
            l2r[q] = bestIdx; r2l[bestIdx] = q; depth[q] = (float)z;
            p3d[3 * q] = (float)P[0]; p3d[3 * q + 1] = (float)P[1]; p3d[3 * q + 2] = (float)P[2];
            ++nMatches;
        }
    }
    return nMatches;
}

}  // extern "C"
