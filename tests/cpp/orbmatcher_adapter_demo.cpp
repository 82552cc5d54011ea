// Runs the flattening / write-back templates of include/orbx/orbmatcher_adapter.hpp against stand-ins for the
// reference's Frame / KeyFrame / MapPoint (same member names, plain vectors instead of OpenCV types), on features
// extracted from two images, and dumps the resulting map-point slots for the Python test to compare with the CPU
// checker.  usage: orbmatcher_adapter_demo left.raw right.raw W H out_prefix
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <map>
#include <memory>
#include <vector>

#include "orbx/extractor.hpp"
#include "orbx/orbmatcher_adapter.hpp"

typedef std::map<unsigned, std::vector<unsigned>> FeatureVector;      // DBoW2::FeatureVector

struct MapPoint {
    bool mbTrackInView = false, mbTrackInViewR = false;
    float mTrackProjX = 0, mTrackProjY = 0, mTrackProjXR = 0, mTrackProjYR = 0, mTrackDepth = 0;
    float mTrackViewCos = 0, mTrackViewCosR = 0;
    int mnTrackScaleLevel = 0, mnTrackScaleLevelR = -1;
    bool bad = false;
    int nObs = 0;
    int tag = -1;                       // position in the candidate vector (-2: was in the frame before the search)
    std::vector<uint8_t> desc;
    bool isBad() const { return bad; }
    int Observations() const { return nObs; }
    const std::vector<uint8_t>& GetDescriptor() const { return desc; }
};

struct Frame {
    int N = 0, Nleft = -1;
    long mnId = 0;
    std::vector<orbx_keypoint> mvKeys, mvKeysRight, mvKeysUn;
    std::vector<uint8_t> mDescriptors;
    std::vector<float> mvuRight, mvScaleFactors;
    float mnMinX = 0, mnMaxX = 0, mnMinY = 0, mnMaxY = 0;
    std::vector<MapPoint*> mvpMapPoints;
    std::vector<int> mvLeftToRightMatch, mvRightToLeftMatch;
    FeatureVector mFeatVec;
};

struct KeyFrame {
    int N = 0, NLeft = -1;
    long mnId = 0;
    std::vector<orbx_keypoint> mvKeys, mvKeysRight, mvKeysUn;
    std::vector<uint8_t> mDescriptors;
    std::vector<float> mvuRight;
    float mnMinX = 0, mnMaxX = 0, mnMinY = 0, mnMaxY = 0;
    std::vector<MapPoint*> pts;
    FeatureVector mFeatVec;
    std::vector<MapPoint*> GetMapPointMatches() const { return pts; }
};

static std::vector<uint8_t> slurp(const char* p) {
    std::ifstream f(p, std::ios::binary);
    return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}
template <class T> static void dump(const std::string& p, const std::vector<T>& v) {
    std::ofstream f(p, std::ios::binary);
    f.write((const char*)v.data(), (std::streamsize)(v.size() * sizeof(T)));
}
static std::vector<int> tags(const std::vector<MapPoint*>& slots) {
    std::vector<int> t(slots.size(), -1);
    for (size_t i = 0; i < slots.size(); ++i) t[i] = slots[i] ? slots[i]->tag : -1;
    return t;
}
static FeatureVector feature_vector(const std::vector<uint8_t>& desc, int n, unsigned drop) {
    FeatureVector fv;
    for (int i = 0; i < n; ++i) {
        const unsigned node = desc[(size_t)i * 32] & 31u;
        if (node % drop == 0) continue;
        fv[node].push_back((unsigned)i);
    }
    return fv;
}

int main(int argc, char** argv) {
    if (argc < 6) { std::fprintf(stderr, "usage: %s left right W H prefix\n", argv[0]); return 2; }
    try {
        const int W = std::atoi(argv[3]), H = std::atoi(argv[4]);
        std::vector<uint8_t> L = slurp(argv[1]), R = slurp(argv[2]);
        if ((int)L.size() != W * H || (int)R.size() != W * H) { std::fprintf(stderr, "bad image size\n"); return 2; }
        const std::string pre = argv[5];
        orbx::Extractor exL(1200, 1.2f, 8, 20, 7), exR(1200, 1.2f, 8, 20, 7);
        std::vector<orbx_keypoint> kL, kR;
        std::vector<uint8_t> dL, dR;
        std::vector<int> lap = {0, 0};
        exL(L.data(), W, H, W, kL, dL, lap);
        exR(R.data(), W, H, W, kR, dR, lap);
        const int nL = (int)kL.size(), nR = (int)kR.size();
        const std::vector<float> sf = exL.GetScaleFactors();
        std::vector<orbx_match2> knn = orbx::KnnMatch2(dL.data(), nL, dR.data(), nR);

        // the candidates of the local map: one map point per left keypoint
        auto local_map = [&](bool allObserved, bool withRight, std::vector<std::unique_ptr<MapPoint>>& own) {
            std::vector<MapPoint*> v;
            for (int i = 0; i < nL; ++i) {
                own.emplace_back(new MapPoint());
                MapPoint* p = own.back().get();
                p->tag = i;
                p->desc.assign(dL.begin() + (size_t)i * 32, dL.begin() + (size_t)i * 32 + 32);
                p->mbTrackInView = i % 7 != 0;
                p->mTrackProjX = kL[i].x + 0.25f * (float)(i % 5);
                p->mTrackProjY = kL[i].y - 0.5f * (float)(i % 3);
                p->mTrackProjXR = p->mTrackProjX - 5.0f;
                p->mnTrackScaleLevel = kL[i].octave;
                p->mTrackViewCos = (i % 2) ? 0.999f : 0.9f;
                p->mTrackDepth = 10.0f + 5.0f * (float)(i % 13);
                p->bad = i % 11 == 0;
                p->nObs = allObserved ? 1 : (i % 3 != 0 ? 2 : 0);
                if (withRight && i % 2 == 0) {
                    const orbx_keypoint& t = kR[knn[i].idx0];
                    p->mbTrackInViewR = true;
                    p->mTrackProjXR = t.x + 0.5f;
                    p->mTrackProjYR = t.y;
                    p->mnTrackScaleLevelR = (i % 10 == 0) ? -1 : t.octave;
                    p->mTrackViewCosR = (i % 4) ? 0.9f : 0.9995f;
                }
                v.push_back(p);
            }
            return v;
        };
        auto residents = [&](Frame& F, std::vector<std::unique_ptr<MapPoint>>& own) {
            F.mvpMapPoints.assign((size_t)F.N, nullptr);
            for (int j = 0; j < F.N; ++j)
                if (j % 17 <= 1) {
                    own.emplace_back(new MapPoint());
                    own.back()->tag = -2;
                    own.back()->nObs = (j % 17 == 0) ? 1 : 0;
                    F.mvpMapPoints[j] = own.back().get();
                }
        };

        // A: one camera, rectified stereo (mvuRight), every candidate observed, far points dropped, th = 3
        {
            std::vector<std::unique_ptr<MapPoint>> own;
            Frame F;
            F.mnId = 1; F.N = nR; F.mvKeysUn = kR; F.mvKeys = kR; F.mDescriptors = dR; F.mvScaleFactors = sf;
            F.mnMaxX = (float)W; F.mnMaxY = (float)H;
            F.mvuRight.assign(nR, -1.f);
            for (int j = 0; j < nR; j += 3) F.mvuRight[j] = kR[j].x - 5.0f + 3.0f * ((float)(j % 4) - 1.5f);
            residents(F, own);
            std::vector<MapPoint*> cand = local_map(true, false, own);
            auto ff = orbx::adapt::features_of(F, F.mnId, F.Nleft, 0);
            auto again = orbx::adapt::features_of(F, F.mnId, F.Nleft, 0);
            const int n = orbx::adapt::search_by_projection_local_map(F, cand, 3.0f, true, 50.0f, 0.8f, *ff, nullptr);
            dump(pre + "_A.bin", tags(F.mvpMapPoints));
            std::printf("A=%d cacheHit=%d\n", n, ff.get() == again.get() ? 1 : 0);
        }
        // B: one camera without mvuRight, candidates with and without observations, th = 1
        {
            std::vector<std::unique_ptr<MapPoint>> own;
            Frame F;
            F.mnId = 2; F.N = nR; F.mvKeysUn = kR; F.mvKeys = kR; F.mDescriptors = dR; F.mvScaleFactors = sf;
            F.mnMaxX = (float)W; F.mnMaxY = (float)H;
            residents(F, own);
            std::vector<MapPoint*> cand = local_map(false, false, own);
            auto ff = orbx::adapt::features_of(F, F.mnId, F.Nleft, 0);
            const int n = orbx::adapt::search_by_projection_local_map(F, cand, 1.0f, false, 0.f, 0.9f, *ff, nullptr);
            dump(pre + "_B.bin", tags(F.mvpMapPoints));
            std::printf("B=%d\n", n);
        }
        // C: two cameras (left = the left image's features, right = the right image's)
        {
            std::vector<std::unique_ptr<MapPoint>> own;
            Frame F;
            F.mnId = 3; F.Nleft = nL; F.N = nL + nR; F.mvKeys = kL; F.mvKeysRight = kR; F.mvScaleFactors = sf;
            F.mDescriptors = dL;
            F.mDescriptors.insert(F.mDescriptors.end(), dR.begin(), dR.end());
            F.mnMaxX = (float)W; F.mnMaxY = (float)H;
            F.mvLeftToRightMatch.assign(nL, -1);
            F.mvRightToLeftMatch.assign(nR, -1);
            for (int i = 0; i < nL && i < nR; i += 5) { F.mvLeftToRightMatch[i] = i; F.mvRightToLeftMatch[i] = i; }
            residents(F, own);
            std::vector<MapPoint*> cand = local_map(false, true, own);
            auto fl = orbx::adapt::features_of(F, F.mnId, F.Nleft, 0);
            auto fr = orbx::adapt::features_of(F, F.mnId, F.Nleft, 1);
            const int n = orbx::adapt::search_by_projection_local_map(F, cand, 3.0f, false, 0.f, 0.9f, *fl, fr.get());
            dump(pre + "_C.bin", tags(F.mvpMapPoints));
            std::printf("C=%d sizes=%d,%d\n", n, fl->size(), fr->size());
        }
        // D / E: the bag-of-words searches between a keyframe (left image) and a frame / keyframe (right image)
        {
            std::vector<std::unique_ptr<MapPoint>> own;
            KeyFrame K1, K2;
            K1.mnId = 10; K1.N = nL; K1.mvKeysUn = kL; K1.mvKeys = kL; K1.mDescriptors = dL; K1.mnMaxX = (float)W; K1.mnMaxY = (float)H;
            K2.mnId = 11; K2.N = nR; K2.mvKeysUn = kR; K2.mvKeys = kR; K2.mDescriptors = dR; K2.mnMaxX = (float)W; K2.mnMaxY = (float)H;
            K1.mFeatVec = feature_vector(dL, nL, 5);
            K2.mFeatVec = feature_vector(dR, nR, 7);
            auto points = [&](int n, int nullEvery, int badEvery, int base) {
                std::vector<MapPoint*> v((size_t)n, nullptr);
                for (int i = 0; i < n; ++i)
                    if (i % nullEvery != 0) {
                        own.emplace_back(new MapPoint());
                        own.back()->tag = base + i;
                        own.back()->bad = i % badEvery == 0;
                        v[i] = own.back().get();
                    }
                return v;
            };
            K1.pts = points(nL, 4, 9, 0);
            K2.pts = points(nR, 6, 5, 100000);
            Frame F;
            F.mnId = 4; F.N = nR; F.mvKeysUn = kR; F.mvKeys = kR; F.mDescriptors = dR; F.mnMaxX = (float)W; F.mnMaxY = (float)H;
            F.mFeatVec = K2.mFeatVec;
            auto f1 = orbx::adapt::features_of(K1, K1.mnId, K1.NLeft, 0);
            auto f2 = orbx::adapt::features_of(K2, K2.mnId, K2.NLeft, 0);
            auto fF = orbx::adapt::features_of(F, F.mnId, F.Nleft, 0);
            std::vector<MapPoint*> mF, m12;
            const int nD = orbx::adapt::search_by_bow_keyframe_frame(&K1, F, mF, 0.7f, true, *f1, *fF);
            const int nE = orbx::adapt::search_by_bow_keyframes(&K1, &K2, m12, 0.8f, true, *f1, *f2);
            dump(pre + "_D.bin", tags(mF));
            dump(pre + "_E.bin", tags(m12));
            orbx::Matcher::NodeLists a, b;
            orbx::adapt::shared_nodes(K1.mFeatVec, K2.mFeatVec, a, b);
            std::printf("D=%d E=%d nodes=%d\n", nD, nE, a.nodes());
        }
        dump(pre + "_kL.bin", kL); dump(pre + "_dL.bin", dL); dump(pre + "_kR.bin", kR); dump(pre + "_dR.bin", dR);
    } catch (const std::exception& e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
