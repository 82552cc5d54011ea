// C++ host side above the C ABI for the matching half of the path: mirror of the parts of
// ORB_SLAM3::ORBmatcher (include/ORBmatcher.h:34-226) and Frame whose data-parallel core runs
// on the GPU.  Header only; see include/orbx.h for the exact semantics of every call.
#ifndef ORBX_MATCHER_HPP_
#define ORBX_MATCHER_HPP_

#include <cstdint>
#include <vector>

#include "extractor.hpp"

namespace orbx {

class Matcher {
public:
    static constexpr int TH_LOW = 50;         // ORBmatcher.cc:35
    static constexpr int TH_HIGH = 100;       // ORBmatcher.cc:34
    static constexpr int HISTO_LENGTH = 30;   // ORBmatcher.cc:36

    // ORBmatcher::ORBmatcher(float nnratio = 0.6, bool checkOri = true)  (ORBmatcher.cc:38)
    explicit Matcher(float nnratio = 0.6f, bool checkOri = true) : mfNNratio(nnratio), mbCheckOrientation(checkOri) {}

    // ORBmatcher::DescriptorDistance (ORBmatcher.cc:2141) for n descriptor pairs at once.
    static std::vector<int> DescriptorDistance(const uint8_t* a, const uint8_t* b, int n) {
        std::vector<int> out(n);
        check(orbx_descriptor_distance(a, b, n, out.data()));
        return out;
    }

    // ORBmatcher::SearchByProjection(Frame&, const vector<MapPoint*>&, th, ...) (ORBmatcher.cc:42-147),
    // mono / rectified stereo.  `taken` plays the role of "F.mvpMapPoints[idx] has observations".
    int SearchByProjection(const std::vector<orbx_keypoint>& keysUn, const std::vector<uint8_t>& descriptors,
                           const float* uRight, float mnMinX, float mnMaxX, float mnMinY, float mnMaxY,
                           const std::vector<orbx_query>& projections, const std::vector<uint8_t>& mapPointDescriptors,
                           std::vector<uint8_t>& taken, std::vector<int>& matchOut) const {
        int n = 0;
        matchOut.assign(projections.size(), -1);
        check(orbx_search_by_projection(keysUn.data(), descriptors.data(), uRight, (int)keysUn.size(), mnMinX, mnMaxX,
                                        mnMinY, mnMaxY, projections.data(), mapPointDescriptors.data(),
                                        (int)projections.size(), mfNNratio, taken.data(), matchOut.data(), &n));
        return n;
    }

    float mfNNratio;
    bool mbCheckOrientation;
};

// Frame::ComputeStereoMatches (Frame.cc:962-1133): fills mvuRight / mvDepth.
inline void ComputeStereoMatches(Extractor& left, Extractor& right, const std::vector<orbx_keypoint>& mvKeys,
                                 const std::vector<uint8_t>& mDescriptors, const std::vector<orbx_keypoint>& mvKeysRight,
                                 const std::vector<uint8_t>& mDescriptorsRight, float mbf, float mb,
                                 std::vector<float>& mvuRight, std::vector<float>& mvDepth) {
    mvuRight.assign(mvKeys.size(), -1.f);
    mvDepth.assign(mvKeys.size(), -1.f);
    check(orbx_stereo_match(left.handle(), right.handle(), 0, mvKeys.data(), mDescriptors.data(), (int)mvKeys.size(),
                            mvKeysRight.data(), mDescriptorsRight.data(), (int)mvKeysRight.size(), mbf, mb,
                            mvuRight.data(), mvDepth.data()));
}

// Frame::BFmatcher.knnMatch(query, train, 2) (Frame.cc:1309).
inline std::vector<orbx_match2> KnnMatch2(const uint8_t* query, int nq, const uint8_t* train, int64_t nt) {
    std::vector<orbx_match2> out(nq);
    check(orbx_hamming_knn2(query, nq, train, nt, out.data()));
    return out;
}

}  // namespace orbx
#endif
