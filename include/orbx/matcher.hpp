// C++ host side above the C ABI for the matching half of the path: mirror of the parts of
// ORB_SLAM3::ORBmatcher (include/ORBmatcher.h:34-226) and Frame whose data-parallel core runs
// on the GPU.  Header only; see include/orbx.h for the exact semantics of every call.
#ifndef ORBX_MATCHER_HPP_
#define ORBX_MATCHER_HPP_

#include <cstdint>
#include <vector>

#include "extractor.hpp"

namespace orbx {

// What a Frame / KeyFrame keeps for the windowed searches (mvKeysUn, mDescriptors, mvuRight, mGrid), held on
// the device; the grid of Frame::AssignFeaturesToGrid (Frame.cc:468-506) is built there.  Assign once per
// frame, search many times; searches may run from several threads at once.
class FrameFeatures {
public:
    FrameFeatures() { check(orbx_frame_create(&f_)); }
    FrameFeatures(const std::vector<orbx_keypoint>& keysUn, const std::vector<uint8_t>& descriptors, const float* uRight,
                  float mnMinX, float mnMaxX, float mnMinY, float mnMaxY) : FrameFeatures() {
        Assign(keysUn, descriptors, uRight, mnMinX, mnMaxX, mnMinY, mnMaxY);
    }
    ~FrameFeatures() { orbx_frame_destroy(f_); }
    FrameFeatures(const FrameFeatures&) = delete;
    FrameFeatures& operator=(const FrameFeatures&) = delete;

    void Assign(const std::vector<orbx_keypoint>& keysUn, const std::vector<uint8_t>& descriptors, const float* uRight,
                float mnMinX, float mnMaxX, float mnMinY, float mnMaxY) {
        check(orbx_frame_assign(f_, keysUn.data(), descriptors.data(), uRight, (int)keysUn.size(), mnMinX, mnMaxX, mnMinY,
                                mnMaxY));
    }
    // Frame::UndistortKeyPoints + ComputeImageBounds + AssignFeaturesToGrid on the device: `keys` are mvKeys of a
    // pinhole camera with distortion (cam.dist[0] == 0: taken as they are, as the reference does).
    void AssignDistorted(const orbx_camera& cam, int width, int height, const std::vector<orbx_keypoint>& keys,
                         const std::vector<uint8_t>& descriptors, const float* uRight) {
        check(orbx_frame_assign_distorted(f_, &cam, width, height, keys.data(), descriptors.data(), uRight, (int)keys.size()));
    }
    // mvKeysUn and {mnMinX, mnMaxX, mnMinY, mnMaxY} as the frame holds them
    void KeysUn(std::vector<orbx_keypoint>& keysUn, float bounds[4]) const {
        keysUn.resize(size() > 0 ? size() : 1);
        check(orbx_frame_keypoints(f_, keysUn.data(), (int)keysUn.size(), bounds));
        keysUn.resize(size());
    }
    int size() const { return orbx_frame_size(f_); }
    // mGrid as CSR: cellStart[64 * 48 + 1] (cell = column * 48 + row) and the keypoint indices.
    void Grid(std::vector<int>& cellStart, std::vector<int>& cellIdx) const {
        int n = 0;
        cellStart.assign(64 * 48 + 1, 0);
        cellIdx.assign(size() > 0 ? size() : 1, 0);
        check(orbx_frame_grid(f_, cellStart.data(), cellIdx.data(), (int)cellIdx.size(), &n));
        cellIdx.resize(n);
    }
    const orbx_frame* handle() const { return f_; }
    orbx_frame* handle() { return f_; }

private:
    orbx_frame* f_ = nullptr;
};

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

    // The same search on features that are already on the device.
    int SearchByProjection(const FrameFeatures& F, const std::vector<orbx_query>& projections,
                           const std::vector<uint8_t>& mapPointDescriptors, std::vector<uint8_t>& taken,
                           std::vector<int>& matchOut) const {
        int n = 0;
        matchOut.assign(projections.size(), -1);
        check(orbx_frame_search_by_projection(F.handle(), projections.data(), mapPointDescriptors.data(),
                                              (int)projections.size(), mfNNratio, taken.data(), matchOut.data(), &n));
        return n;
    }

    // A frame / keyframe side of a windowed search: undistorted keypoints, descriptors, optional right
    // coordinates and the image bounds of its grid (mnMinX, mnMaxX, mnMinY, mnMaxY).
    struct View {
        const std::vector<orbx_keypoint>* keys;
        const std::vector<uint8_t>* descriptors;
        const float* uRight;
        float bounds[4];
        int size() const { return (int)keys->size(); }
    };

    // Best-only loop shared by SearchByProjection(Frame&, const Frame&, ...) (ORBmatcher.cc:1761),
    // (Frame&, KeyFrame*, sFound, th, ORBdist) (:1974), (KeyFrame*, Sim3, points, matched, th, ratio) (:476, :582)
    // and Fuse(KeyFrame*, Sim3, ...) (:1409).  kpMatch[i] = query that owns keypoint i afterwards (-1 none);
    // queryAngles is used only when this matcher checks orientation.  Returns the reference's return value.
    int SearchWindowBest(const View& v, const std::vector<orbx_query>& queries, const std::vector<uint8_t>& queryDescriptors,
                         float maxDist, std::vector<uint8_t>* taken, const uint8_t* queryMarks, const float* queryAngles,
                         std::vector<int>& matchOut, std::vector<int>& kpMatch, float rotFactor = 1.0f / HISTO_LENGTH) const {
        int n = 0;
        matchOut.assign(queries.size(), -1);
        kpMatch.assign(v.keys->size(), -1);
        check(orbx_search_window_best(v.keys->data(), v.descriptors->data(), v.uRight, v.size(), v.bounds[0], v.bounds[1],
                                      v.bounds[2], v.bounds[3], queries.data(), queryDescriptors.data(), (int)queries.size(),
                                      maxDist, taken ? taken->data() : nullptr, queryMarks,
                                      mbCheckOrientation ? queryAngles : nullptr, rotFactor, matchOut.data(), kpMatch.data(), &n));
        return n;
    }

    int SearchWindowBest(const FrameFeatures& F, const std::vector<orbx_query>& queries,
                         const std::vector<uint8_t>& queryDescriptors, float maxDist, std::vector<uint8_t>* taken,
                         const uint8_t* queryMarks, const float* queryAngles, std::vector<int>& matchOut,
                         std::vector<int>& kpMatch, float rotFactor = 1.0f / HISTO_LENGTH) const {
        int n = 0;
        matchOut.assign(queries.size(), -1);
        kpMatch.assign(F.size(), -1);
        check(orbx_frame_search_window_best(F.handle(), queries.data(), queryDescriptors.data(), (int)queries.size(), maxDist,
                                            taken ? taken->data() : nullptr, queryMarks,
                                            mbCheckOrientation ? queryAngles : nullptr, rotFactor, matchOut.data(),
                                            kpMatch.data(), &n));
        return n;
    }

    int Fuse(const FrameFeatures& F, const std::vector<orbx_query>& queries, const std::vector<uint8_t>& queryDescriptors,
             const std::vector<float>& invLevelSigma2, std::vector<int>& matchOut) const {
        int n = 0;
        matchOut.assign(queries.size(), -1);
        check(orbx_frame_fuse_search(F.handle(), queries.data(), queryDescriptors.data(), (int)queries.size(),
                                     invLevelSigma2.data(), (int)invLevelSigma2.size(), matchOut.data(), &n));
        return n;
    }

    // Matching part of Fuse(KeyFrame*, const vector<MapPoint*>&, th, bRight) (ORBmatcher.cc:1214-1407).
    int Fuse(const View& v, const std::vector<orbx_query>& queries, const std::vector<uint8_t>& queryDescriptors,
             const std::vector<float>& invLevelSigma2, std::vector<int>& matchOut) const {
        int n = 0;
        matchOut.assign(queries.size(), -1);
        check(orbx_fuse_search(v.keys->data(), v.descriptors->data(), v.uRight, v.size(), v.bounds[0], v.bounds[1], v.bounds[2],
                               v.bounds[3], queries.data(), queryDescriptors.data(), (int)queries.size(), invLevelSigma2.data(),
                               (int)invLevelSigma2.size(), matchOut.data(), &n));
        return n;
    }

    // SearchForInitialization (ORBmatcher.cc:695-808).  vbPrevMatched holds {x, y} pairs and is updated.
    int SearchForInitialization(const View& f1, const View& f2, std::vector<float>& vbPrevMatched,
                                std::vector<int>& vnMatches12, int windowSize = 10) const {
        int n = 0;
        vnMatches12.assign(f1.keys->size(), -1);
        check(orbx_search_for_initialization(f1.keys->data(), f1.descriptors->data(), f1.size(), f2.keys->data(),
                                             f2.descriptors->data(), f2.size(), f2.bounds[0], f2.bounds[1], f2.bounds[2],
                                             f2.bounds[3], vbPrevMatched.data(), windowSize, mfNNratio,
                                             mbCheckOrientation ? 1 : 0, vnMatches12.data(), &n));
        return n;
    }

    // A DBoW2::FeatureVector restricted to the nodes two vectors share, as CSR (idx[off[b] .. off[b+1]) = node b).
    struct NodeLists {
        std::vector<int> idx, off;
        int nodes() const { return off.empty() ? 0 : (int)off.size() - 1; }
    };

    // SearchByBoW(KeyFrame*, KeyFrame*, matches12) (ORBmatcher.cc:824-966).
    int SearchByBoW(const std::vector<uint8_t>& desc1, const std::vector<float>& angle1, const std::vector<uint8_t>& valid1,
                    const NodeLists& fv1, const std::vector<uint8_t>& desc2, const std::vector<float>& angle2,
                    const std::vector<uint8_t>& valid2, const NodeLists& fv2, std::vector<int>& matches12) const {
        int n = 0;
        matches12.assign(valid1.size(), -1);
        check(orbx_search_by_bow_keyframes(desc1.data(), angle1.data(), valid1.data(), (int)valid1.size(), fv1.idx.data(),
                                           fv1.off.data(), desc2.data(), angle2.data(), valid2.data(), (int)valid2.size(),
                                           fv2.idx.data(), fv2.off.data(), fv1.nodes(), mfNNratio, mbCheckOrientation ? 1 : 0,
                                           matches12.data(), &n));
        return n;
    }

    // SearchByBoW(KeyFrame*, Frame&, matches) (ORBmatcher.cc:235-474): matchF[i] = keyframe keypoint or -1.
    int SearchByBoW(const std::vector<uint8_t>& descKF, const std::vector<float>& angleKF, const std::vector<uint8_t>& validKF,
                    const NodeLists& fvKF, const std::vector<uint8_t>& descF, const std::vector<float>& angleF, int nLeftF,
                    const NodeLists& fvF, std::vector<int>& matchF) const {
        int n = 0;
        matchF.assign(angleF.size(), -1);
        check(orbx_search_by_bow(descKF.data(), angleKF.data(), validKF.data(), (int)validKF.size(), fvKF.idx.data(),
                                 fvKF.off.data(), descF.data(), angleF.data(), (int)angleF.size(), nLeftF, fvF.idx.data(),
                                 fvF.off.data(), fvKF.nodes(), mfNNratio, mbCheckOrientation ? 1 : 0, matchF.data(), &n));
        return n;
    }

    float mfNNratio;
    bool mbCheckOrientation;
};

// ORBVocabulary::transform as called by Frame::ComputeBoW (Frame.cc:872-879): BowVector as ascending
// (word, value) pairs, FeatureVector as CSR over ascending node ids.
class Vocabulary {
public:
    Vocabulary(int nNodes, const int* childOff, const int* childIdx, const uint8_t* nodeDescriptors, const int* wordId,
               const double* weight, int depth, int device = 0) {
        check(orbx_vocabulary_create(nNodes, childOff, childIdx, nodeDescriptors, wordId, weight, depth, device, &h_));
    }
    ~Vocabulary() { orbx_vocabulary_destroy(h_); }
    Vocabulary(const Vocabulary&) = delete;
    Vocabulary& operator=(const Vocabulary&) = delete;

    void transform(const uint8_t* descriptors, int n, std::vector<int>& bowWord, std::vector<double>& bowValue,
                   std::vector<int>& fvNode, std::vector<int>& fvOff, std::vector<int>& fvIdx, int levelsup = 4) const {
        bowWord.assign(n, 0); bowValue.assign(n, 0.0); fvNode.assign(n, 0); fvOff.assign(n + 1, 0); fvIdx.assign(n, 0);
        int nb = 0, nn = 0;
        check(orbx_bow_vectors(h_, descriptors, n, levelsup, 1, bowWord.data(), bowValue.data(), &nb, fvNode.data(),
                               fvOff.data(), fvIdx.data(), &nn));
        bowWord.resize(nb); bowValue.resize(nb); fvNode.resize(nn); fvOff.resize(nn + 1); fvIdx.resize(fvOff[nn]);
    }
    orbx_vocabulary* handle() const { return h_; }

private:
    orbx_vocabulary* h_ = nullptr;
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

// Frame::ComputeStereoFishEyeMatches (Frame.cc:1286-1338): fills mvLeftToRightMatch / mvRightToLeftMatch / mvDepth /
// mvStereo3Dpoints (x, y, z per left keypoint) and returns nMatches.  camLeft / camRight are KannalaBrandt8 models,
// Rlr (row-major 3x3) / tlr = mRlr / mtlr.
inline int ComputeStereoFishEyeMatches(const std::vector<orbx_keypoint>& mvKeys, const std::vector<uint8_t>& mDescriptors,
                                       int monoLeft, const std::vector<orbx_keypoint>& mvKeysRight,
                                       const std::vector<uint8_t>& mDescriptorsRight, int monoRight,
                                       const orbx_camera_model& camLeft, const orbx_camera_model& camRight, const float* Rlr,
                                       const float* tlr, const std::vector<float>& mvLevelSigma2,
                                       std::vector<int>& mvLeftToRightMatch, std::vector<int>& mvRightToLeftMatch,
                                       std::vector<float>& mvDepth, std::vector<float>& mvStereo3Dpoints) {
    mvLeftToRightMatch.assign(mvKeys.size(), -1);
    mvRightToLeftMatch.assign(mvKeysRight.size(), -1);
    mvDepth.assign(mvKeys.size(), -1.f);
    mvStereo3Dpoints.assign(3 * mvKeys.size(), 0.f);
    int n = 0;
    check(orbx_stereo_fisheye_matches(mvKeys.data(), mDescriptors.data(), (int)mvKeys.size(), monoLeft, mvKeysRight.data(),
                                      mDescriptorsRight.data(), (int)mvKeysRight.size(), monoRight, &camLeft, &camRight, Rlr,
                                      tlr, mvLevelSigma2.data(), (int)mvLevelSigma2.size(), mvLeftToRightMatch.data(),
                                      mvRightToLeftMatch.data(), mvDepth.data(), mvStereo3Dpoints.data(), &n));
    return n;
}

// Frame::BFmatcher.knnMatch(query, train, 2) (Frame.cc:1309).
inline std::vector<orbx_match2> KnnMatch2(const uint8_t* query, int nq, const uint8_t* train, int64_t nt) {
    std::vector<orbx_match2> out(nq);
    check(orbx_hamming_knn2(query, nq, train, nt, out.data()));
    return out;
}

}  // namespace orbx
#endif
