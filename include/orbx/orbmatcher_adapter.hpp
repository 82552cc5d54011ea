// ORB_SLAM3::ORBmatcher over the C ABI (include/orbx.h): the class Frame, Tracking, LocalMapping and LoopClosing
// construct on the stack at each call site (include/ORBmatcher.h:34-226), with the reference's 13 search / fuse
// signatures, implemented as "flatten the object graph -> one orbx_* call -> write the results back".
//
// Two layers:
//  * namespace orbx::adapt -- templates that do the flattening and the write-back.  They name the reference's
//    members (mbTrackInView, mvpMapPoints, mFeatVec, ...) but depend on no OpenCV / Eigen type, so they compile and
//    run against stand-in structs (tests/cpp/adapter_demo.cpp does that on the GPU box).
//  * namespace ORB_SLAM3 (only with -DORBX_WITH_OPENCV, inside the reference's source tree) -- the class itself.  Its
//    methods evaluate the reference's geometric preamble (projection, frustum and distance gates, predicted level)
//    with the reference's own Eigen / Sophus / GeometricCamera objects, hand the survivors to the templates, and apply
//    the map mutations (AddObservation, Replace, ...) in the reference's order.
// Device state: every Frame / KeyFrame searched gets an orbx::FrameFeatures (keypoints, descriptors, mvuRight and the
// grid on the device) from a small cache keyed by the object's address and id; see features_of().
#ifndef ORBX_ORBMATCHER_ADAPTER_HPP_
#define ORBX_ORBMATCHER_ADAPTER_HPP_

#include <climits>
#include <cstring>
#include <list>
#include <memory>
#include <mutex>
#include <set>
#include <utility>
#include <vector>

#include "matcher.hpp"

namespace orbx {
namespace adapt {

// ---- element access ------------------------------------------------------------------------------------------------
// How the templates read the caller's containers: DescriptorRows<D>::row(d, i) = the 32 bytes of descriptor i,
// KeyPointArray<KP>::data(v) = the keypoints as orbx_keypoint (the layout of cv::KeyPoint).  Specialised here for plain
// vectors and, in the OpenCV section, for cv::Mat / cv::KeyPoint.
template <class D> struct DescriptorRows;
template <> struct DescriptorRows<std::vector<uint8_t>> {
    static const uint8_t* row(const std::vector<uint8_t>& d, int i) { return d.data() + (size_t)i * 32; }
};
template <class KP> struct KeyPointArray;
template <> struct KeyPointArray<orbx_keypoint> {
    static const orbx_keypoint* data(const std::vector<orbx_keypoint>& v) { return v.data(); }
};
template <class D> const uint8_t* row_ptr(const D& d, int i) { return DescriptorRows<D>::row(d, i); }
template <class KP> const orbx_keypoint* kp_data(const std::vector<KP>& v) { return KeyPointArray<KP>::data(v); }

// ---- DBoW2::FeatureVector (std::map<NodeId, std::vector<unsigned int>>) -> CSR over the shared nodes ---------------
// The walk of ORBmatcher.cc:259-446 (and :1021-1179) visits the nodes present in both maps, in ascending order.
template <class FeatureVectorT>
void shared_nodes(const FeatureVectorT& fv1, const FeatureVectorT& fv2, Matcher::NodeLists& a, Matcher::NodeLists& b) {
    a.idx.clear(); b.idx.clear();
    a.off.assign(1, 0); b.off.assign(1, 0);
    auto it1 = fv1.begin(), it2 = fv2.begin();
    while (it1 != fv1.end() && it2 != fv2.end()) {
        if (it1->first == it2->first) {
            for (auto v : it1->second) a.idx.push_back((int)v);
            for (auto v : it2->second) b.idx.push_back((int)v);
            a.off.push_back((int)a.idx.size());
            b.off.push_back((int)b.idx.size());
            ++it1; ++it2;
        } else if (it1->first < it2->first) {
            it1 = fv1.lower_bound(it2->first);
        } else {
            it2 = fv2.lower_bound(it1->first);
        }
    }
}

// ---- SearchByProjection(Frame&, const vector<MapPoint*>&, th, bFarPoints, thFarPoints) (ORBmatcher.cc:42-216) ------
// FrameT: Nleft, mvScaleFactors, mvpMapPoints, mvLeftToRightMatch, mvRightToLeftMatch.
// MapPointT: mbTrackInView(R), mTrackProjX/Y, mTrackProjXR/YR, mnTrackScaleLevel(R), mTrackViewCos(R), mTrackDepth,
// isBad(), Observations(), GetDescriptor().  `left` / `right` hold the frame's features on the device (right = null
// for one camera).
inline float radius_by_viewing_cos(float viewCos) { return viewCos > 0.998 ? 2.5f : 4.0f; }   // ORBmatcher.cc:218-224

template <class FrameT, class MapPointT>
int search_by_projection_local_map(FrameT& F, const std::vector<MapPointT*>& vpMapPoints, float th, bool bFarPoints,
                                   float thFarPoints, float nnratio, const FrameFeatures& left, const FrameFeatures* right) {
    const bool bFactor = th != 1.0;
    const bool two = F.Nleft != -1 && right != nullptr;
    std::vector<orbx_query> ql, qr;
    std::vector<uint8_t> qd, marks;
    std::vector<int> src;
    for (size_t iMP = 0; iMP < vpMapPoints.size(); ++iMP) {
        MapPointT* pMP = vpMapPoints[iMP];
        if (!pMP->mbTrackInView && !pMP->mbTrackInViewR) continue;
        if (bFarPoints && pMP->mTrackDepth > thFarPoints) continue;
        if (pMP->isBad()) continue;
        orbx_query a{}, b{};
        if (pMP->mbTrackInView) {
            const int lvl = pMP->mnTrackScaleLevel;
            float r = radius_by_viewing_cos(pMP->mTrackViewCos);
            if (bFactor) r *= th;
            a.x = pMP->mTrackProjX; a.y = pMP->mTrackProjY; a.xr = pMP->mTrackProjXR;
            a.radius = r * F.mvScaleFactors[lvl];
            a.min_level = lvl - 1; a.max_level = lvl;
        }
        if (two && pMP->mbTrackInViewR && pMP->mnTrackScaleLevelR != -1) {
            const int lvl = pMP->mnTrackScaleLevelR;
            b.x = pMP->mTrackProjXR; b.y = pMP->mTrackProjYR;
            b.radius = radius_by_viewing_cos(pMP->mTrackViewCosR) * F.mvScaleFactors[lvl];     // no th here (:151-154)
            b.min_level = lvl - 1; b.max_level = lvl;
        }
        if (!(a.radius > 0) && !(b.radius > 0)) continue;
        ql.push_back(a); qr.push_back(b);
        const uint8_t* d = row_ptr(pMP->GetDescriptor(), 0);
        qd.insert(qd.end(), d, d + 32);
        marks.push_back(pMP->Observations() > 0 ? 1 : 0);
        src.push_back((int)iMP);
    }
    const int nL = left.size(), nR = two ? right->size() : 0;
    std::vector<uint8_t> taken((size_t)nL + nR, 0);
    for (int i = 0; i < nL + nR; ++i)
        taken[i] = (F.mvpMapPoints[i] && F.mvpMapPoints[i]->Observations() > 0) ? 1 : 0;
    std::vector<int> kpQuery((size_t)nL + nR, -1);
    int n = 0;
    // one camera: the rectified-stereo gate (:92-97) is applied on the device from the frame's mvuRight
    check(orbx_frame_search_by_projection_stereo(left.handle(), two ? right->handle() : nullptr, ql.data(), qr.data(),
                                                 qd.data(), (int)ql.size(), nnratio,
                                                 two ? F.mvLeftToRightMatch.data() : nullptr,
                                                 two ? F.mvRightToLeftMatch.data() : nullptr, taken.data(), marks.data(),
                                                 kpQuery.data(), &n));
    for (int i = 0; i < nL + nR; ++i)
        if (kpQuery[i] >= 0) F.mvpMapPoints[i] = vpMapPoints[src[kpQuery[i]]];
    return n;
}

// ---- SearchByBoW(KeyFrame*, Frame&, vector<MapPoint*>&) (ORBmatcher.cc:235-474) -------------------------------------
template <class KeyFrameT, class FrameT, class MapPointT>
int search_by_bow_keyframe_frame(KeyFrameT* pKF, FrameT& F, std::vector<MapPointT*>& vpMapPointMatches, float nnratio,
                                 bool checkOri, const FrameFeatures& kfAll, const FrameFeatures& fAll) {
    const std::vector<MapPointT*> vpMapPointsKF = pKF->GetMapPointMatches();
    vpMapPointMatches.assign((size_t)F.N, static_cast<MapPointT*>(nullptr));
    Matcher::NodeLists a, b;
    shared_nodes(pKF->mFeatVec, F.mFeatVec, a, b);
    std::vector<uint8_t> valid(vpMapPointsKF.size(), 0);
    for (size_t i = 0; i < valid.size(); ++i) valid[i] = (vpMapPointsKF[i] && !vpMapPointsKF[i]->isBad()) ? 1 : 0;
    std::vector<int> matchF((size_t)F.N, -1);
    int n = 0;
    check(orbx_frame_search_by_bow(kfAll.handle(), valid.data(), a.idx.data(), a.off.data(), fAll.handle(), F.Nleft,
                                   b.idx.data(), b.off.data(), a.nodes(), nnratio, checkOri ? 1 : 0, matchF.data(), &n));
    for (int i = 0; i < F.N; ++i)
        if (matchF[i] >= 0) vpMapPointMatches[i] = vpMapPointsKF[matchF[i]];
    return n;
}

// ---- SearchByBoW(KeyFrame*, KeyFrame*, vector<MapPoint*>&) (ORBmatcher.cc:824-966) ----------------------------------
template <class KeyFrameT, class MapPointT>
int search_by_bow_keyframes(KeyFrameT* pKF1, KeyFrameT* pKF2, std::vector<MapPointT*>& vpMatches12, float nnratio,
                            bool checkOri, const FrameFeatures& f1, const FrameFeatures& f2) {
    const std::vector<MapPointT*> vpMapPoints1 = pKF1->GetMapPointMatches();
    const std::vector<MapPointT*> vpMapPoints2 = pKF2->GetMapPointMatches();
    vpMatches12.assign(vpMapPoints1.size(), static_cast<MapPointT*>(nullptr));
    Matcher::NodeLists a, b;
    shared_nodes(pKF1->mFeatVec, pKF2->mFeatVec, a, b);
    // a fisheye keyframe's feature vector also lists right-camera features; :859-862 skips those of keyframe 1 and
    // :880-883 those of keyframe 2 (indices at or beyond mvKeysUn.size() when NLeft != -1)
    const size_t nUn1 = pKF1->NLeft != -1 ? pKF1->mvKeysUn.size() : vpMapPoints1.size();
    const size_t nUn2 = pKF2->NLeft != -1 ? pKF2->mvKeysUn.size() : vpMapPoints2.size();
    std::vector<uint8_t> v1(vpMapPoints1.size(), 0), v2(vpMapPoints2.size(), 0);
    for (size_t i = 0; i < v1.size(); ++i) v1[i] = (i < nUn1 && vpMapPoints1[i] && !vpMapPoints1[i]->isBad()) ? 1 : 0;
    for (size_t i = 0; i < v2.size(); ++i) v2[i] = (i < nUn2 && vpMapPoints2[i] && !vpMapPoints2[i]->isBad()) ? 1 : 0;
    std::vector<int> m12(vpMapPoints1.size(), -1);
    int n = 0;
    check(orbx_frame_search_by_bow_keyframes(f1.handle(), v1.data(), a.idx.data(), a.off.data(), f2.handle(), v2.data(),
                                             b.idx.data(), b.off.data(), a.nodes(), nnratio, checkOri ? 1 : 0, m12.data(), &n));
    for (size_t i = 0; i < m12.size(); ++i)
        if (m12[i] >= 0) vpMatches12[i] = vpMapPoints2[m12[i]];
    return n;
}

// ---- best-only windowed searches: write-back of kp_match / kp_query into a vector of map-point pointers -------------
// queries were built from candidates[src[q]]; owner[i] = query that owns keypoint i afterwards (-1 none).
template <class MapPointT>
void assign_owners(const std::vector<int>& owner, const std::vector<int>& src, const std::vector<MapPointT*>& candidates,
                   std::vector<MapPointT*>& slots) {
    for (size_t i = 0; i < owner.size(); ++i)
        if (owner[i] >= 0) slots[i] = candidates[src[owner[i]]];
}

// Occupancy of a frame's keypoints as the loops test it: a map point with observations (ORBmatcher.cc:1832-1834).
template <class MapPointT>
std::vector<uint8_t> taken_with_observations(const std::vector<MapPointT*>& slots) {
    std::vector<uint8_t> t(slots.size(), 0);
    for (size_t i = 0; i < slots.size(); ++i) t[i] = (slots[i] && slots[i]->Observations() > 0) ? 1 : 0;
    return t;
}
// ... and as the loop-closing searches test it: any map point (ORBmatcher.cc:553, :2037).
template <class MapPointT>
std::vector<uint8_t> taken_non_null(const std::vector<MapPointT*>& slots) {
    std::vector<uint8_t> t(slots.size(), 0);
    for (size_t i = 0; i < slots.size(); ++i) t[i] = slots[i] ? 1 : 0;
    return t;
}

// ---- device features of the objects being searched ---------------------------------------------------------------
// The reference's Frame / KeyFrame have no slot for a device handle, so the adapter keeps the most recently used
// ones here, keyed by (address, id, keypoint count); `build` uploads the object's arrays when it is not present.  Holds
// up to `capacity` entries, the least recently used is dropped.  (An integration that may touch Frame.h / KeyFrame.h
// adds a std::shared_ptr<orbx::FrameFeatures> member instead and skips the cache.)
class FeatureCache {
public:
    explicit FeatureCache(size_t capacity = 96) : cap_(capacity) {}
    template <class Build>
    std::shared_ptr<FrameFeatures> get(const void* object, long id, int n, int part, Build build) {
        std::lock_guard<std::mutex> lock(m_);
        for (auto it = items_.begin(); it != items_.end(); ++it)
            if (it->object == object && it->id == id && it->n == n && it->part == part) {
                items_.splice(items_.begin(), items_, it);
                return items_.front().f;
            }
        Item e{object, id, n, part, std::make_shared<FrameFeatures>()};
        build(*e.f);
        items_.push_front(e);
        if (items_.size() > cap_) items_.pop_back();
        return items_.front().f;
    }
    void forget(const void* object) {
        std::lock_guard<std::mutex> lock(m_);
        items_.remove_if([object](const Item& i) { return i.object == object; });
    }

private:
    struct Item { const void* object; long id; int n; int part; std::shared_ptr<FrameFeatures> f; };
    std::list<Item> items_;
    std::mutex m_;
    size_t cap_;
};
inline FeatureCache& feature_cache() { static FeatureCache c; return c; }

// part: 0 = the camera whose grid the searches walk (one camera: mvKeysUn with the rectified-stereo gate of mvuRight;
// two cameras: the left one), 1 = the right camera, 2 = the combined arrays (left first, right at Nleft + i) for the
// bucketed searches, 3 = one camera without the mvuRight gate (the loop-closing searches read mvuRight nowhere).
template <class FrameT>
std::shared_ptr<FrameFeatures> features_of(const FrameT& F, long id, int nLeft, int part) {
    const int N = F.N;
    return feature_cache().get(&F, id, N, part, [&](FrameFeatures& ff) {
        const uint8_t* D = N > 0 ? row_ptr(F.mDescriptors, 0) : nullptr;
        if (nLeft == -1) {
            const float* ur = (part == 3 || F.mvuRight.empty()) ? nullptr : F.mvuRight.data();
            check(orbx_frame_assign(ff.handle(), kp_data(F.mvKeysUn), D, ur, N, F.mnMinX, F.mnMaxX, F.mnMinY, F.mnMaxY));
        } else if (part == 0 || part == 3) {
            check(orbx_frame_assign(ff.handle(), kp_data(F.mvKeys), D, nullptr, nLeft, F.mnMinX, F.mnMaxX, F.mnMinY, F.mnMaxY));
        } else if (part == 1) {
            check(orbx_frame_assign(ff.handle(), kp_data(F.mvKeysRight), N > nLeft ? row_ptr(F.mDescriptors, nLeft) : nullptr,
                                    nullptr, N - nLeft, F.mnMinX, F.mnMaxX, F.mnMinY, F.mnMaxY));
        } else {
            std::vector<orbx_keypoint> all(kp_data(F.mvKeys), kp_data(F.mvKeys) + nLeft);
            all.insert(all.end(), kp_data(F.mvKeysRight), kp_data(F.mvKeysRight) + (N - nLeft));
            check(orbx_frame_assign(ff.handle(), all.data(), D, nullptr, N, F.mnMinX, F.mnMaxX, F.mnMinY, F.mnMaxY));
        }
    });
}

}  // namespace adapt
}  // namespace orbx

// =====================================================================================================================
#ifdef ORBX_WITH_OPENCV
// Compiled inside the reference's tree in place of src/ORBmatcher.cc (which is dropped from the library's sources):
// include/ORBmatcher.h becomes `#include <orbx/orbmatcher_adapter.hpp>` after the reference's own includes.
#include <opencv2/core/core.hpp>
#include "sophus/sim3.hpp"

#include "Frame.h"
#include "KeyFrame.h"
#include "MapPoint.h"

namespace orbx {
namespace adapt {
template <> struct DescriptorRows<cv::Mat> {
    static const uint8_t* row(const cv::Mat& d, int i) { return d.ptr<uint8_t>(i); }
};
template <> struct KeyPointArray<cv::KeyPoint> {
    static_assert(sizeof(cv::KeyPoint) == sizeof(orbx_keypoint), "cv::KeyPoint layout");
    static const orbx_keypoint* data(const std::vector<cv::KeyPoint>& v) { return reinterpret_cast<const orbx_keypoint*>(v.data()); }
};
inline std::shared_ptr<FrameFeatures> features_of(const ORB_SLAM3::Frame& F, int part = 0) {
    return features_of(F, (long)F.mnId, F.Nleft, part);
}
inline std::shared_ptr<FrameFeatures> features_of(const ORB_SLAM3::KeyFrame& K, int part = 0) {
    return features_of(K, (long)K.mnId, K.NLeft, part);
}
}  // namespace adapt
}  // namespace orbx

namespace ORB_SLAM3 {

class ORBmatcher {
public:
    ORBmatcher(float nnratio = 0.6, bool checkOri = true) : mfNNratio(nnratio), mbCheckOrientation(checkOri) {}

    static const int TH_LOW = 50;
    static const int TH_HIGH = 100;
    static const int HISTO_LENGTH = 30;

    // ORBmatcher.cc:2141-2157 -- one pair: the host popcount is the right tool (the batched form is
    // orbx_descriptor_distance)
    static int DescriptorDistance(const cv::Mat& a, const cv::Mat& b) {
        const uint64_t* pa = a.ptr<uint64_t>();
        const uint64_t* pb = b.ptr<uint64_t>();
        int dist = 0;
        for (int i = 0; i < 4; ++i) dist += __builtin_popcountll(pa[i] ^ pb[i]);
        return dist;
    }

    // :42-216
    int SearchByProjection(Frame& F, const std::vector<MapPoint*>& vpMapPoints, const float th = 3, const bool bFarPoints = false,
                           const float thFarPoints = 50.0f) {
        auto L = orbx::adapt::features_of(F, 0);
        std::shared_ptr<orbx::FrameFeatures> R;
        if (F.Nleft != -1) R = orbx::adapt::features_of(F, 1);
        return orbx::adapt::search_by_projection_local_map(F, vpMapPoints, th, bFarPoints, thFarPoints, mfNNratio, *L, R.get());
    }

    // :1761-1972
    int SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, const float th, const bool bMono) {
        const Sophus::SE3f Tcw = CurrentFrame.GetPose();
        const Eigen::Vector3f twc = Tcw.inverse().translation();
        const Sophus::SE3f Tlw = LastFrame.GetPose();
        const Eigen::Vector3f tlc = Tlw * twc;
        const bool bForward = tlc(2) > CurrentFrame.mb && !bMono;
        const bool bBackward = -tlc(2) > CurrentFrame.mb && !bMono;
        const bool two = CurrentFrame.Nleft != -1;
        std::vector<orbx_query> ql, qr;
        std::vector<uint8_t> qd, marks;
        std::vector<float> angles;
        std::vector<MapPoint*> cand;
        std::vector<int> src;
        for (int i = 0; i < LastFrame.N; i++) {
            MapPoint* pMP = LastFrame.mvpMapPoints[i];
            if (!pMP || LastFrame.mvbOutlier[i]) continue;
            Eigen::Vector3f x3Dw = pMP->GetWorldPos();
            Eigen::Vector3f x3Dc = Tcw * x3Dw;
            const float invzc = 1.0 / x3Dc(2);
            if (invzc < 0) continue;
            Eigen::Vector2f uv = CurrentFrame.mpCamera->project(x3Dc);
            if (uv(0) < CurrentFrame.mnMinX || uv(0) > CurrentFrame.mnMaxX) continue;
            if (uv(1) < CurrentFrame.mnMinY || uv(1) > CurrentFrame.mnMaxY) continue;
            const cv::KeyPoint& kpLF = (LastFrame.Nleft == -1) ? LastFrame.mvKeysUn[i]
                                       : (i < LastFrame.Nleft) ? LastFrame.mvKeys[i]
                                                               : LastFrame.mvKeysRight[i - LastFrame.Nleft];
            const int nLastOctave = (LastFrame.Nleft == -1 || i < LastFrame.Nleft) ? LastFrame.mvKeys[i].octave
                                                                                   : LastFrame.mvKeysRight[i - LastFrame.Nleft].octave;
            const float radius = th * CurrentFrame.mvScaleFactors[nLastOctave];
            orbx_query a{};
            a.x = uv(0); a.y = uv(1); a.radius = radius;
            a.xr = uv(0) - CurrentFrame.mbf * invzc;                       // the rectified-stereo gate (:1836-1842)
            if (bForward) { a.min_level = nLastOctave; a.max_level = -1; }
            else if (bBackward) { a.min_level = 0; a.max_level = nLastOctave; }
            else { a.min_level = nLastOctave - 1; a.max_level = nLastOctave + 1; }
            orbx_query b = a;
            if (two) {
                Eigen::Vector3f x3Dr = CurrentFrame.GetRelativePoseTrl() * x3Dc;
                Eigen::Vector2f uvr = CurrentFrame.mpCamera->project(x3Dr);
                b.x = uvr(0); b.y = uvr(1);
            }
            ql.push_back(a); qr.push_back(b);
            const uint8_t* d = pMP->GetDescriptor().ptr<uint8_t>();
            qd.insert(qd.end(), d, d + 32);
            marks.push_back(pMP->Observations() > 0 ? 1 : 0);
            angles.push_back(kpLF.angle);
            cand.push_back(pMP);
            src.push_back((int)cand.size() - 1);
        }
        auto L = orbx::adapt::features_of(CurrentFrame, 0);
        std::shared_ptr<orbx::FrameFeatures> R;
        if (two) R = orbx::adapt::features_of(CurrentFrame, 1);
        std::vector<uint8_t> taken = orbx::adapt::taken_with_observations(CurrentFrame.mvpMapPoints);
        const std::vector<uint8_t> takenBefore = taken;
        std::vector<int> owner(CurrentFrame.mvpMapPoints.size(), -1);
        int n = 0;
        orbx::check(orbx_frame_search_window_best_stereo(L->handle(), two ? R->handle() : nullptr, ql.data(), qr.data(), qd.data(),
                                                         (int)ql.size(), (float)TH_HIGH, taken.data(), marks.data(),
                                                         mbCheckOrientation ? angles.data() : nullptr, 1.0f / HISTO_LENGTH,
                                                         owner.data(), &n));
        // A slot the loop assigned and the histogram then removed ends as NULL (:1964).  That differs from "never
        // touched" only for a slot that held a map point without observations before the call (Tracking clears the
        // slots first, so normally none does); then a second pass without the histogram tells the two apart.
        bool anyLoose = false;
        for (size_t i = 0; i < taken.size(); ++i) anyLoose = anyLoose || (CurrentFrame.mvpMapPoints[i] && !takenBefore[i]);
        if (mbCheckOrientation && anyLoose) {
            std::vector<int> first(owner.size(), -1);
            std::vector<uint8_t> taken2 = orbx::adapt::taken_with_observations(CurrentFrame.mvpMapPoints);
            int n2 = 0;
            orbx::check(orbx_frame_search_window_best_stereo(L->handle(), two ? R->handle() : nullptr, ql.data(), qr.data(),
                                                             qd.data(), (int)ql.size(), (float)TH_HIGH, taken2.data(),
                                                             marks.data(), nullptr, 0.f, first.data(), &n2));
            for (size_t i = 0; i < first.size(); ++i)
                if (first[i] >= 0 && owner[i] < 0) CurrentFrame.mvpMapPoints[i] = static_cast<MapPoint*>(NULL);
        }
        orbx::adapt::assign_owners(owner, src, cand, CurrentFrame.mvpMapPoints);
        return n;
    }

    // :1974-2094
    int SearchByProjection(Frame& CurrentFrame, KeyFrame* pKF, const std::set<MapPoint*>& sAlreadyFound, const float th,
                           const int ORBdist) {
        const Sophus::SE3f Tcw = CurrentFrame.GetPose();
        Eigen::Vector3f Ow = Tcw.inverse().translation();
        const std::vector<MapPoint*> vpMPs = pKF->GetMapPointMatches();
        std::vector<orbx_query> qs;
        std::vector<uint8_t> qd;
        std::vector<float> angles;
        std::vector<int> src;
        for (size_t i = 0, iend = vpMPs.size(); i < iend; i++) {
            MapPoint* pMP = vpMPs[i];
            if (!pMP || pMP->isBad() || sAlreadyFound.count(pMP)) continue;
            Eigen::Vector3f x3Dw = pMP->GetWorldPos();
            Eigen::Vector3f x3Dc = Tcw * x3Dw;
            const Eigen::Vector2f uv = CurrentFrame.mpCamera->project(x3Dc);
            if (uv(0) < CurrentFrame.mnMinX || uv(0) > CurrentFrame.mnMaxX) continue;
            if (uv(1) < CurrentFrame.mnMinY || uv(1) > CurrentFrame.mnMaxY) continue;
            Eigen::Vector3f PO = x3Dw - Ow;
            float dist3D = PO.norm();
            if (dist3D < pMP->GetMinDistanceInvariance() || dist3D > pMP->GetMaxDistanceInvariance()) continue;
            int nPredictedLevel = pMP->PredictScale(dist3D, &CurrentFrame);
            orbx_query q{};
            q.x = uv(0); q.y = uv(1);
            q.radius = th * CurrentFrame.mvScaleFactors[nPredictedLevel];
            q.min_level = nPredictedLevel - 1; q.max_level = nPredictedLevel + 1;
            qs.push_back(q);
            const uint8_t* d = pMP->GetDescriptor().ptr<uint8_t>();
            qd.insert(qd.end(), d, d + 32);
            angles.push_back(pKF->mvKeysUn[i].angle);
            src.push_back((int)i);
        }
        // this variant reads mvuRight nowhere: search the features without the stereo gate
        auto Fh = orbx::adapt::features_of(CurrentFrame, CurrentFrame.Nleft == -1 ? 3 : 0);
        std::vector<uint8_t> taken = orbx::adapt::taken_non_null(CurrentFrame.mvpMapPoints);
        taken.resize(Fh->size());
        std::vector<int> out, owner;
        orbx::Matcher m(mfNNratio, mbCheckOrientation);
        const int n = m.SearchWindowBest(*Fh, qs, qd, (float)ORBdist, &taken, nullptr, angles.data(), out, owner);
        // only empty slots can be assigned here (:2037), so what the histogram removes is NULL as it was before
        for (size_t i = 0; i < owner.size(); ++i)
            if (owner[i] >= 0) CurrentFrame.mvpMapPoints[i] = vpMPs[src[owner[i]]];
        return n;
    }

    // :476-580
    int SearchByProjection(KeyFrame* pKF, Sophus::Sim3<float>& Scw, const std::vector<MapPoint*>& vpPoints,
                           std::vector<MapPoint*>& vpMatched, int th, float ratioHamming = 1.0) {
        std::vector<int> src;
        std::vector<orbx_query> qs;
        std::vector<uint8_t> qd;
        Sim3Queries(pKF, Scw, vpPoints, std::set<MapPoint*>(vpMatched.begin(), vpMatched.end()), (float)th, qs, qd, src);
        auto Kh = orbx::adapt::features_of(*pKF, pKF->NLeft == -1 ? 3 : 0);
        std::vector<uint8_t> taken = orbx::adapt::taken_non_null(vpMatched);
        taken.resize(Kh->size());
        std::vector<int> out, owner;
        orbx::Matcher m(mfNNratio, false);
        const int n = m.SearchWindowBest(*Kh, qs, qd, TH_LOW * ratioHamming, &taken, nullptr, nullptr, out, owner);
        for (size_t i = 0; i < owner.size(); ++i)
            if (owner[i] >= 0) vpMatched[i] = vpPoints[src[owner[i]]];
        return n;
    }

    // :582-693
    int SearchByProjection(KeyFrame* pKF, Sophus::Sim3<float>& Scw, const std::vector<MapPoint*>& vpPoints,
                           const std::vector<KeyFrame*>& vpPointsKFs, std::vector<MapPoint*>& vpMatched,
                           std::vector<KeyFrame*>& vpMatchedKF, int th, float ratioHamming = 1.0) {
        std::vector<int> src;
        std::vector<orbx_query> qs;
        std::vector<uint8_t> qd;
        // this overload projects with fx, fy, cx, cy itself (:621-626) instead of calling the camera model
        Sim3Queries(pKF, Scw, vpPoints, std::set<MapPoint*>(vpMatched.begin(), vpMatched.end()), (float)th, qs, qd, src, true);
        auto Kh = orbx::adapt::features_of(*pKF, pKF->NLeft == -1 ? 3 : 0);
        std::vector<uint8_t> taken = orbx::adapt::taken_non_null(vpMatched);
        taken.resize(Kh->size());
        std::vector<int> out, owner;
        orbx::Matcher m(mfNNratio, false);
        const int n = m.SearchWindowBest(*Kh, qs, qd, TH_LOW * ratioHamming, &taken, nullptr, nullptr, out, owner);
        for (size_t i = 0; i < owner.size(); ++i)
            if (owner[i] >= 0) { vpMatched[i] = vpPoints[src[owner[i]]]; vpMatchedKF[i] = vpPointsKFs[src[owner[i]]]; }
        return n;
    }

    // :235-474
    int SearchByBoW(KeyFrame* pKF, Frame& F, std::vector<MapPoint*>& vpMapPointMatches) {
        auto K = orbx::adapt::features_of(*pKF, pKF->NLeft == -1 ? 0 : 2);
        auto Fh = orbx::adapt::features_of(F, F.Nleft == -1 ? 0 : 2);
        return orbx::adapt::search_by_bow_keyframe_frame(pKF, F, vpMapPointMatches, mfNNratio, mbCheckOrientation, *K, *Fh);
    }
    // :824-966
    int SearchByBoW(KeyFrame* pKF1, KeyFrame* pKF2, std::vector<MapPoint*>& vpMatches12) {
        auto A = orbx::adapt::features_of(*pKF1, pKF1->NLeft == -1 ? 0 : 2);
        auto B = orbx::adapt::features_of(*pKF2, pKF2->NLeft == -1 ? 0 : 2);
        return orbx::adapt::search_by_bow_keyframes(pKF1, pKF2, vpMatches12, mfNNratio, mbCheckOrientation, *A, *B);
    }

    // :695-808
    int SearchForInitialization(Frame& F1, Frame& F2, std::vector<cv::Point2f>& vbPrevMatched, std::vector<int>& vnMatches12,
                                int windowSize = 10) {
        static_assert(sizeof(cv::Point2f) == 2 * sizeof(float), "cv::Point2f layout");
        std::vector<uint8_t> d1(F1.mDescriptors.ptr<uint8_t>(0), F1.mDescriptors.ptr<uint8_t>(0) + (size_t)F1.N * 32);
        std::vector<uint8_t> d2(F2.mDescriptors.ptr<uint8_t>(0), F2.mDescriptors.ptr<uint8_t>(0) + (size_t)F2.N * 32);
        vnMatches12.assign(F1.mvKeysUn.size(), -1);
        int n = 0;
        orbx::check(orbx_search_for_initialization(orbx::adapt::kp_data(F1.mvKeysUn), d1.data(), F1.N,
                                                   orbx::adapt::kp_data(F2.mvKeysUn), d2.data(), F2.N, F2.mnMinX, F2.mnMaxX,
                                                   F2.mnMinY, F2.mnMaxY, reinterpret_cast<float*>(vbPrevMatched.data()), windowSize,
                                                   mfNNratio, mbCheckOrientation ? 1 : 0, vnMatches12.data(), &n));
        return n;
    }

    // :968-1212 -- the camera-model gate is the reference's own epipolarConstrain, called back per surviving candidate
    int SearchForTriangulation(KeyFrame* pKF1, KeyFrame* pKF2, std::vector<std::pair<size_t, size_t>>& vMatchedPairs,
                               const bool bOnlyStereo, const bool bCoarse = false) {
        Sophus::SE3f T1w = pKF1->GetPose();
        Sophus::SE3f T2w = pKF2->GetPose();
        Sophus::SE3f Tw2 = pKF2->GetPoseInverse();
        Eigen::Vector3f Cw = pKF1->GetCameraCenter();
        Eigen::Vector3f C2 = T2w * Cw;
        Eigen::Vector2f ep = pKF2->mpCamera->project(C2);
        struct Gate {
            KeyFrame *k1, *k2;
            bool two;
            Sophus::SE3f T[2][2];
            static int call(void* u, int idx1, int idx2) {
                Gate* g = static_cast<Gate*>(u);
                const bool r1 = !(g->k1->NLeft == -1 || idx1 < g->k1->NLeft), r2 = !(g->k2->NLeft == -1 || idx2 < g->k2->NLeft);
                const cv::KeyPoint& kp1 = (g->k1->NLeft == -1) ? g->k1->mvKeysUn[idx1] : !r1 ? g->k1->mvKeys[idx1] : g->k1->mvKeysRight[idx1 - g->k1->NLeft];
                const cv::KeyPoint& kp2 = (g->k2->NLeft == -1) ? g->k2->mvKeysUn[idx2] : !r2 ? g->k2->mvKeys[idx2] : g->k2->mvKeysRight[idx2 - g->k2->NLeft];
                GeometricCamera* c1 = (g->two && r1) ? g->k1->mpCamera2 : g->k1->mpCamera;
                GeometricCamera* c2 = (g->two && r2) ? g->k2->mpCamera2 : g->k2->mpCamera;
                const Sophus::SE3f& T12 = g->T[g->two && r1][g->two && r2];
                return c1->epipolarConstrain(c2, kp1, kp2, T12.rotationMatrix(), T12.translation(),
                                             g->k1->mvLevelSigma2[kp1.octave], g->k2->mvLevelSigma2[kp2.octave]) ? 1 : 0;
            }
        } gate;
        gate.k1 = pKF1; gate.k2 = pKF2;
        gate.two = pKF1->mpCamera2 && pKF2->mpCamera2;
        if (!pKF1->mpCamera2 && !pKF2->mpCamera2) {
            gate.T[0][0] = T1w * Tw2;
        } else {
            Sophus::SE3f Tr1w = pKF1->GetRightPose();
            Sophus::SE3f Twr2 = pKF2->GetRightPoseInverse();
            gate.T[0][0] = T1w * Tw2; gate.T[0][1] = T1w * Twr2; gate.T[1][0] = Tr1w * Tw2; gate.T[1][1] = Tr1w * Twr2;
        }
        orbx_triangulation_geometry geom;
        std::memset(&geom, 0, sizeof(geom));
        geom.n_left1 = pKF1->mpCamera2 ? pKF1->NLeft : -1;
        geom.n_left2 = pKF2->mpCamera2 ? pKF2->NLeft : -1;
        geom.epipole_x = ep(0); geom.epipole_y = ep(1);
        geom.scale_factors2 = pKF2->mvScaleFactors.data();
        geom.level_sigma2_1 = pKF1->mvLevelSigma2.data();
        geom.level_sigma2_2 = pKF2->mvLevelSigma2.data();
        geom.n_levels = (int)pKF2->mvScaleFactors.size();
        geom.gate = &Gate::call;
        geom.gate_user = &gate;
        orbx::Matcher::NodeLists a, b;
        orbx::adapt::shared_nodes(pKF1->mFeatVec, pKF2->mFeatVec, a, b);
        std::vector<uint8_t> free1((size_t)pKF1->N, 0), free2((size_t)pKF2->N, 0);
        for (int i = 0; i < pKF1->N; ++i) free1[i] = pKF1->GetMapPoint(i) ? 0 : 1;
        for (int i = 0; i < pKF2->N; ++i) free2[i] = pKF2->GetMapPoint(i) ? 0 : 1;
        auto A = orbx::adapt::features_of(*pKF1, pKF1->NLeft == -1 ? 0 : 2);
        auto B = orbx::adapt::features_of(*pKF2, pKF2->NLeft == -1 ? 0 : 2);
        std::vector<int> m12((size_t)pKF1->N, -1);
        int n = 0;
        orbx::check(orbx_frame_search_for_triangulation(A->handle(), free1.data(), a.idx.data(), a.off.data(), B->handle(),
                                                        free2.data(), b.idx.data(), b.off.data(), a.nodes(), &geom,
                                                        bOnlyStereo ? 1 : 0, bCoarse ? 1 : 0, mbCheckOrientation ? 1 : 0,
                                                        m12.data(), &n));
        vMatchedPairs.clear();
        vMatchedPairs.reserve(n);
        for (size_t i = 0; i < m12.size(); ++i)
            if (m12[i] >= 0) vMatchedPairs.push_back(std::make_pair(i, (size_t)m12[i]));
        return n;
    }

    // :1526-1743
    int SearchBySim3(KeyFrame* pKF1, KeyFrame* pKF2, std::vector<MapPoint*>& vpMatches12, const Sophus::Sim3f& S12, const float th) {
        Sophus::SE3f T1w = pKF1->GetPose();
        Sophus::SE3f T2w = pKF2->GetPose();
        Sophus::Sim3f S21 = S12.inverse();
        const std::vector<MapPoint*> vpMapPoints1 = pKF1->GetMapPointMatches();
        const std::vector<MapPoint*> vpMapPoints2 = pKF2->GetMapPointMatches();
        const int N1 = vpMapPoints1.size(), N2 = vpMapPoints2.size();
        std::vector<bool> done1(N1, false), done2(N2, false);
        for (int i = 0; i < N1; i++) {
            MapPoint* pMP = vpMatches12[i];
            if (pMP) {
                done1[i] = true;
                int idx2 = std::get<0>(pMP->GetIndexInKeyFrame(pKF2));
                if (idx2 >= 0 && idx2 < N2) done2[idx2] = true;
            }
        }
        auto direction = [&](KeyFrame* from, KeyFrame* into, const std::vector<MapPoint*>& pts, const std::vector<bool>& done,
                             const Sophus::SE3f& Tfw, const Sophus::Sim3f& S, std::vector<orbx_query>& qs, std::vector<uint8_t>& qd,
                             std::vector<int>& source) {
            (void)from;
            for (int i = 0; i < (int)pts.size(); i++) {
                MapPoint* pMP = pts[i];
                if (!pMP || done[i] || pMP->isBad()) continue;
                Eigen::Vector3f p3Dc = S * (Tfw * pMP->GetWorldPos());
                if (p3Dc(2) < 0.0) continue;
                const float invz = 1.0 / p3Dc(2);
                const float u = into->fx * (p3Dc(0) * invz) + into->cx, v = into->fy * (p3Dc(1) * invz) + into->cy;
                if (!into->IsInImage(u, v)) continue;
                const float dist3D = p3Dc.norm();
                if (dist3D < pMP->GetMinDistanceInvariance() || dist3D > pMP->GetMaxDistanceInvariance()) continue;
                const int lvl = pMP->PredictScale(dist3D, into);
                orbx_query q{};
                q.x = u; q.y = v; q.radius = th * into->mvScaleFactors[lvl]; q.min_level = lvl - 1; q.max_level = lvl;
                qs.push_back(q);
                const uint8_t* d = pMP->GetDescriptor().ptr<uint8_t>();
                qd.insert(qd.end(), d, d + 32);
                source.push_back(i);
            }
        };
        std::vector<orbx_query> q12, q21;
        std::vector<uint8_t> d12, d21;
        std::vector<int> s1, s2;
        direction(pKF1, pKF2, vpMapPoints1, done1, T1w, S21, q12, d12, s1);
        direction(pKF2, pKF1, vpMapPoints2, done2, T2w, S12, q21, d21, s2);
        const float b1[4] = {(float)pKF1->mnMinX, (float)pKF1->mnMaxX, (float)pKF1->mnMinY, (float)pKF1->mnMaxY};
        const float b2[4] = {(float)pKF2->mnMinX, (float)pKF2->mnMaxX, (float)pKF2->mnMinY, (float)pKF2->mnMaxY};
        std::vector<int> m12(N1, -1);
        int n = 0;
        orbx::check(orbx_search_by_sim3(orbx::adapt::kp_data(pKF1->mvKeysUn), pKF1->mDescriptors.ptr<uint8_t>(0), N1, b1,
                                        orbx::adapt::kp_data(pKF2->mvKeysUn), pKF2->mDescriptors.ptr<uint8_t>(0), N2, b2, q12.data(),
                                        d12.data(), s1.data(), (int)q12.size(), q21.data(), d21.data(), s2.data(), (int)q21.size(),
                                        m12.data(), &n));
        for (int i1 = 0; i1 < N1; ++i1)
            if (m12[i1] >= 0) vpMatches12[i1] = vpMapPoints2[m12[i1]];
        return n;
    }

    // :1214-1407
    int Fuse(KeyFrame* pKF, const std::vector<MapPoint*>& vpMapPoints, const float th = 3.0, const bool bRight = false) {
        Sophus::SE3f Tcw = bRight ? pKF->GetRightPose() : pKF->GetPose();
        Eigen::Vector3f Ow = bRight ? pKF->GetRightCameraCenter() : pKF->GetCameraCenter();
        GeometricCamera* pCamera = bRight ? pKF->mpCamera2 : pKF->mpCamera;
        std::vector<orbx_query> qs;
        std::vector<uint8_t> qd;
        std::vector<int> src;
        for (int i = 0; i < (int)vpMapPoints.size(); i++) {
            MapPoint* pMP = vpMapPoints[i];
            if (!pMP || pMP->isBad() || pMP->IsInKeyFrame(pKF)) continue;
            Eigen::Vector3f p3Dw = pMP->GetWorldPos();
            Eigen::Vector3f p3Dc = Tcw * p3Dw;
            if (p3Dc(2) < 0.0f) continue;
            const float invz = 1 / p3Dc(2);
            const Eigen::Vector2f uv = pCamera->project(p3Dc);
            if (!pKF->IsInImage(uv(0), uv(1))) continue;
            Eigen::Vector3f PO = p3Dw - Ow;
            const float dist3D = PO.norm();
            if (dist3D < pMP->GetMinDistanceInvariance() || dist3D > pMP->GetMaxDistanceInvariance()) continue;
            if (PO.dot(pMP->GetNormal()) < 0.5 * dist3D) continue;
            const int lvl = pMP->PredictScale(dist3D, pKF);
            orbx_query q{};
            q.x = uv(0); q.y = uv(1); q.xr = uv(0) - pKF->mbf * invz;
            q.radius = th * pKF->mvScaleFactors[lvl]; q.min_level = lvl - 1; q.max_level = lvl;
            qs.push_back(q);
            const uint8_t* d = pMP->GetDescriptor().ptr<uint8_t>();
            qd.insert(qd.end(), d, d + 32);
            src.push_back(i);
        }
        auto K = orbx::adapt::features_of(*pKF, bRight ? 1 : 0);
        std::vector<int> out;
        orbx::Matcher(mfNNratio, mbCheckOrientation).Fuse(*K, qs, qd, pKF->mvInvLevelSigma2, out);
        // the map is changed in the reference's order; a point an earlier replacement made bad or put into the keyframe
        // is skipped as the reference's loop would skip it (:1255-1264)
        int nFused = 0;
        for (size_t q = 0; q < out.size(); ++q) {
            if (out[q] < 0) continue;
            MapPoint* pMP = vpMapPoints[src[q]];
            if (pMP->isBad() || pMP->IsInKeyFrame(pKF)) continue;
            const int bestIdx = out[q] + (bRight ? pKF->NLeft : 0);
            MapPoint* pMPinKF = pKF->GetMapPoint(bestIdx);
            if (pMPinKF) {
                if (!pMPinKF->isBad()) {
                    if (pMPinKF->Observations() > pMP->Observations()) pMP->Replace(pMPinKF);
                    else pMPinKF->Replace(pMP);
                }
            } else {
                pMP->AddObservation(pKF, bestIdx);
                pKF->AddMapPoint(pMP, bestIdx);
            }
            nFused++;
        }
        return nFused;
    }

    // :1409-1524
    int Fuse(KeyFrame* pKF, Sophus::Sim3f& Scw, const std::vector<MapPoint*>& vpPoints, float th, std::vector<MapPoint*>& vpReplacePoint) {
        std::vector<int> src;
        std::vector<orbx_query> qs;
        std::vector<uint8_t> qd;
        Sim3Queries(pKF, Scw, vpPoints, pKF->GetMapPoints(), th, qs, qd, src);
        auto Kh = orbx::adapt::features_of(*pKF, pKF->NLeft == -1 ? 3 : 0);
        std::vector<int> out, owner;
        orbx::Matcher(mfNNratio, false).SearchWindowBest(*Kh, qs, qd, (float)TH_LOW, nullptr, nullptr, nullptr, out, owner);
        int nFused = 0;
        for (size_t q = 0; q < out.size(); ++q) {
            // SearchWindowBest without `taken` reports every query's own best keypoint
            if (out[q] < 0) continue;
            MapPoint* pMP = vpPoints[src[q]];
            MapPoint* pMPinKF = pKF->GetMapPoint(out[q]);
            if (pMPinKF) {
                if (!pMPinKF->isBad()) vpReplacePoint[src[q]] = pMPinKF;
            } else {
                pMP->AddObservation(pKF, out[q]);
                pKF->AddMapPoint(pMP, out[q]);
            }
            nFused++;
        }
        return nFused;
    }

protected:
    // The geometric preamble shared by the Sim3 projection searches and Fuse (:485-538, :591-650, :1418-1473).
    // inlinePinhole: the projection written out as in :621-626 (u = fx * (x * invz) + cx), which rounds differently
    // from Pinhole::project's fx * x / z + cx and ignores a fisheye model.
    void Sim3Queries(KeyFrame* pKF, Sophus::Sim3f& Scw, const std::vector<MapPoint*>& vpPoints, const std::set<MapPoint*>& already,
                     float th, std::vector<orbx_query>& qs, std::vector<uint8_t>& qd, std::vector<int>& src,
                     bool inlinePinhole = false) {
        Sophus::SE3f Tcw = Sophus::SE3f(Scw.rotationMatrix(), Scw.translation() / Scw.scale());
        Eigen::Vector3f Ow = Tcw.inverse().translation();
        for (int iMP = 0, iend = vpPoints.size(); iMP < iend; iMP++) {
            MapPoint* pMP = vpPoints[iMP];
            if (pMP->isBad() || already.count(pMP)) continue;
            Eigen::Vector3f p3Dw = pMP->GetWorldPos();
            Eigen::Vector3f p3Dc = Tcw * p3Dw;
            if (p3Dc(2) < 0.0) continue;
            Eigen::Vector2f uv;
            if (inlinePinhole) {
                const float invz = 1 / p3Dc(2);
                const float x = p3Dc(0) * invz, y = p3Dc(1) * invz;
                uv(0) = pKF->fx * x + pKF->cx;
                uv(1) = pKF->fy * y + pKF->cy;
            } else {
                uv = pKF->mpCamera->project(p3Dc);
            }
            if (!pKF->IsInImage(uv(0), uv(1))) continue;
            Eigen::Vector3f PO = p3Dw - Ow;
            const float dist = PO.norm();
            if (dist < pMP->GetMinDistanceInvariance() || dist > pMP->GetMaxDistanceInvariance()) continue;
            if (PO.dot(pMP->GetNormal()) < 0.5 * dist) continue;
            const int lvl = pMP->PredictScale(dist, pKF);
            orbx_query q{};
            q.x = uv(0); q.y = uv(1); q.radius = th * pKF->mvScaleFactors[lvl]; q.min_level = lvl - 1; q.max_level = lvl;
            qs.push_back(q);
            const uint8_t* d = pMP->GetDescriptor().ptr<uint8_t>();
            qd.insert(qd.end(), d, d + 32);
            src.push_back(iMP);
        }
    }

    float mfNNratio;
    bool mbCheckOrientation;
};

}  // namespace ORB_SLAM3
#endif  // ORBX_WITH_OPENCV
#endif
