// Exercises the C++ host layer (include/orbx/*.hpp) end to end: extracts two images, runs
// stereo matching and a kNN search, and dumps the results as raw binary for the Python test
// to compare with the oracle.  usage: adapter_demo left.raw right.raw W H out_prefix
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <vector>

#include "orbx/extractor.hpp"
#include "orbx/matcher.hpp"

static std::vector<uint8_t> slurp(const char* p) {
    std::ifstream f(p, std::ios::binary);
    return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}
template <class T> static void dump(const std::string& p, const std::vector<T>& v) {
    std::ofstream f(p, std::ios::binary);
    f.write((const char*)v.data(), (std::streamsize)(v.size() * sizeof(T)));
}

int main(int argc, char** argv) {
    if (argc < 6) { std::fprintf(stderr, "usage: %s left right W H prefix\n", argv[0]); return 2; }
    try {
        const int W = std::atoi(argv[3]), H = std::atoi(argv[4]);
        std::vector<uint8_t> L = slurp(argv[1]), R = slurp(argv[2]);
        if ((int)L.size() != W * H || (int)R.size() != W * H) { std::fprintf(stderr, "bad image size\n"); return 2; }
        orbx::Extractor exL(1200, 1.2f, 8, 20, 7), exR(1200, 1.2f, 8, 20, 7);
        exL.keepPyramid = true;
        std::vector<orbx_keypoint> kL, kR;
        std::vector<uint8_t> dL, dR;
        std::vector<int> lap = {0, 0};
        const int monoL = exL(L.data(), W, H, W, kL, dL, lap);
        const int monoR = exR(R.data(), W, H, W, kR, dR, lap);
        std::vector<uint8_t> empty_k;
        std::vector<orbx_keypoint> ek; std::vector<uint8_t> ed;
        const int monoEmpty = exL(nullptr, 0, 0, 0, ek, ed, lap);
        std::vector<float> uR, depth;
        const float mbf = 47.90639384423901f, fx = 435.2046959714599f;
        orbx::ComputeStereoMatches(exL, exR, kL, dL, kR, dR, mbf, mbf / fx, uR, depth);
        std::vector<orbx_match2> knn = orbx::KnnMatch2(dL.data(), (int)kL.size(), dR.data(), (int64_t)kR.size());
        // SearchForInitialization between the two images (as ORBmatcher(0.9, true) with a 100 px window)
        orbx::Matcher matcher(0.9f, true);
        orbx::Matcher::View v1{&kL, &dL, nullptr, {0.f, (float)W, 0.f, (float)H}}, v2{&kR, &dR, nullptr, {0.f, (float)W, 0.f, (float)H}};
        std::vector<float> prev(2 * kL.size());
        for (size_t i = 0; i < kL.size(); ++i) { prev[2 * i] = kL[i].x; prev[2 * i + 1] = kL[i].y; }
        std::vector<int> m12;
        const int nInit = matcher.SearchForInitialization(v1, v2, prev, m12, 100);
        std::string pre = argv[5];
        dump(pre + "_m12.bin", m12); dump(pre + "_prev.bin", prev);
        // the right image's features kept on the device; the left keypoints searched in it twice, as
        // Tracking searches one frame several times (SearchByProjection with the frame's own grid)
        orbx::FrameFeatures FR(kR, dR, nullptr, 0.f, (float)W, 0.f, (float)H);
        std::vector<int> gStart, gIdx;
        FR.Grid(gStart, gIdx);
        std::vector<orbx_query> proj(kL.size());
        const std::vector<float> sf = exL.GetScaleFactors();
        for (size_t i = 0; i < kL.size(); ++i)
            proj[i] = orbx_query{kL[i].x, kL[i].y, 0.f, 15.f * sf[kL[i].octave], kL[i].octave - 1, kL[i].octave};
        std::vector<uint8_t> taken(kR.size(), 0);
        std::vector<int> sbp1, sbp2;
        const int nSbp1 = matcher.SearchByProjection(FR, proj, dL, taken, sbp1);
        const int nSbp2 = matcher.SearchByProjection(FR, proj, dL, taken, sbp2);   // second pass skips what the first took
        // input staging through the C++ layer: the left image shifted by (+2.25, -1.5) pixels with cv::remap arithmetic
        std::vector<float> mapX((size_t)W * H), mapY((size_t)W * H);
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) { mapX[(size_t)y * W + x] = x + 2.25f; mapY[(size_t)y * W + x] = y - 1.5f; }
        orbx::InputStager stager(W, H, 1, false, mapX.data(), mapY.data(), W, H);
        dump(pre + "_staged.bin", stager(L.data(), W));
        dump(pre + "_gstart.bin", gStart); dump(pre + "_gidx.bin", gIdx);
        dump(pre + "_sbp1.bin", sbp1); dump(pre + "_sbp2.bin", sbp2);
        std::printf("nSbp1=%d nSbp2=%d\n", nSbp1, nSbp2);
        dump(pre + "_kL.bin", kL); dump(pre + "_dL.bin", dL); dump(pre + "_kR.bin", kR); dump(pre + "_dR.bin", dR);
        dump(pre + "_uR.bin", uR); dump(pre + "_depth.bin", depth); dump(pre + "_knn.bin", knn);
        dump(pre + "_pyr3.bin", exL.mvImagePyramid.at(3).pixels);
        std::printf("monoL=%d monoR=%d monoEmpty=%d nL=%zu nR=%zu levels=%d sf1=%.9g nInit=%d\n", monoL, monoR, monoEmpty,
                    kL.size(), kR.size(), exL.GetLevels(), exL.GetScaleFactors().at(1), nInit);
    } catch (const std::exception& e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
