// Declarations only, written for type-checking include/orbx/orbmatcher_adapter.hpp with -DORBX_WITH_OPENCV on a
// machine that has neither OpenCV, Eigen, Sophus nor the reference's tree: the names, members and signatures the
// adapter touches (include/Frame.h, KeyFrame.h, MapPoint.h, CameraModels/GeometricCamera.h of the reference), with no
// bodies.  Nothing here is linked or run; the check is `g++ -fsyntax-only`.
#ifndef ORBX_TYPECHECK_REFERENCE_DECLS_HPP_
#define ORBX_TYPECHECK_REFERENCE_DECLS_HPP_
#include <cstddef>
#include <cstdint>
#include <map>
#include <set>
#include <tuple>
#include <vector>

#define CV_8U 0
#define CV_8UC1 0
#define CV_Assert(expr) ((void)(expr))
namespace cv {
struct Point2f { float x, y; };
struct KeyPoint { Point2f pt; float size, angle, response; int octave, class_id; };
struct Rect2i { int x, y, width, height; };
struct Mat {
    Mat();
    Mat(int rows, int cols, int type, void* data);
    template <class T> T* ptr(int row = 0);
    template <class T> const T* ptr(int row = 0) const;
    int type() const;
    void create(int rows, int cols, int type);
    unsigned char* data;
    int rows, cols;
    size_t step;
};
struct _InputArray {
    _InputArray(const Mat& m);
    bool empty() const;
    Mat getMat() const;
};
struct _OutputArray : _InputArray {
    _OutputArray(Mat& m);
    void release() const;
    void create(int rows, int cols, int type) const;
};
typedef const _InputArray& InputArray;
typedef const _OutputArray& OutputArray;
}  // namespace cv

namespace Eigen {
template <int N> struct Vec {
    float& operator()(int i);
    float operator()(int i) const;
    float norm() const;
    float dot(const Vec& o) const;
    Vec operator-(const Vec& o) const;
    Vec operator/(float s) const;
};
typedef Vec<3> Vector3f;
typedef Vec<2> Vector2f;
struct Matrix3f {};
}  // namespace Eigen

namespace Sophus {
template <class T> struct SE3 {
    SE3();
    SE3(const Eigen::Matrix3f& R, const Eigen::Vector3f& t);
    SE3 inverse() const;
    Eigen::Vector3f translation() const;
    Eigen::Matrix3f rotationMatrix() const;
    Eigen::Vector3f operator*(const Eigen::Vector3f& p) const;
    SE3 operator*(const SE3& o) const;
};
typedef SE3<float> SE3f;
template <class T> struct Sim3 {
    Sim3 inverse() const;
    Eigen::Vector3f translation() const;
    Eigen::Matrix3f rotationMatrix() const;
    T scale() const;
    Eigen::Vector3f operator*(const Eigen::Vector3f& p) const;
};
typedef Sim3<float> Sim3f;
}  // namespace Sophus

namespace DBoW2 { typedef std::map<unsigned int, std::vector<unsigned int>> FeatureVector; }

namespace ORB_SLAM3 {
class Frame;
class KeyFrame;
class GeometricCamera {
public:
    virtual Eigen::Vector2f project(const Eigen::Vector3f& v3D) = 0;
    virtual bool epipolarConstrain(GeometricCamera* pCamera2, const cv::KeyPoint& kp1, const cv::KeyPoint& kp2,
                                   const Eigen::Matrix3f& R12, const Eigen::Vector3f& t12, const float sigmaLevel,
                                   const float unc) = 0;
};
class MapPoint {
public:
    Eigen::Vector3f GetWorldPos();
    Eigen::Vector3f GetNormal();
    int Observations();
    void AddObservation(KeyFrame* pKF, int idx);
    std::tuple<int, int> GetIndexInKeyFrame(KeyFrame* pKF);
    bool IsInKeyFrame(KeyFrame* pKF);
    bool isBad();
    void Replace(MapPoint* pMP);
    cv::Mat GetDescriptor();
    float GetMinDistanceInvariance();
    float GetMaxDistanceInvariance();
    int PredictScale(const float& currentDist, KeyFrame* pKF);
    int PredictScale(const float& currentDist, Frame* pF);
    bool mbTrackInView, mbTrackInViewR;
    float mTrackProjX, mTrackProjY, mTrackDepth, mTrackDepthR, mTrackProjXR, mTrackProjYR;
    int mnTrackScaleLevel, mnTrackScaleLevelR;
    float mTrackViewCos, mTrackViewCosR;
};
class Frame {
public:
    Sophus::SE3<float> GetPose() const;
    Sophus::SE3f GetRelativePoseTrl();
    long unsigned int mnId;
    int N, Nleft, Nright;
    float mbf, mb;
    std::vector<cv::KeyPoint> mvKeys, mvKeysRight, mvKeysUn;
    std::vector<float> mvuRight, mvDepth;
    cv::Mat mDescriptors, mDescriptorsRight;
    DBoW2::FeatureVector mFeatVec;
    std::vector<MapPoint*> mvpMapPoints;
    std::vector<bool> mvbOutlier;
    std::vector<float> mvScaleFactors, mvLevelSigma2, mvInvLevelSigma2;
    static float mnMinX, mnMaxX, mnMinY, mnMaxY;
    GeometricCamera *mpCamera, *mpCamera2;
    std::vector<int> mvLeftToRightMatch, mvRightToLeftMatch;
};
class KeyFrame {
public:
    Sophus::SE3f GetPose();
    Sophus::SE3f GetPoseInverse();
    Eigen::Vector3f GetCameraCenter();
    Sophus::SE3<float> GetRightPose();
    Sophus::SE3<float> GetRightPoseInverse();
    Eigen::Vector3f GetRightCameraCenter();
    void AddMapPoint(MapPoint* pMP, const size_t& idx);
    std::set<MapPoint*> GetMapPoints();
    std::vector<MapPoint*> GetMapPointMatches();
    MapPoint* GetMapPoint(const size_t& idx);
    bool IsInImage(const float& x, const float& y) const;
    long unsigned int mnId;
    const float fx, fy, cx, cy, invfx, invfy, mbf, mb, mThDepth;
    const int N;
    const std::vector<cv::KeyPoint> mvKeys, mvKeysUn, mvKeysRight;
    const std::vector<float> mvuRight, mvDepth;
    const cv::Mat mDescriptors;
    DBoW2::FeatureVector mFeatVec;
    const std::vector<float> mvScaleFactors, mvLevelSigma2, mvInvLevelSigma2;
    const int mnMinX, mnMinY, mnMaxX, mnMaxY;
    GeometricCamera *mpCamera, *mpCamera2;
    const int NLeft, NRight;
};
}  // namespace ORB_SLAM3
#endif
