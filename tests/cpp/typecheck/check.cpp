// Type-checks the ORB_SLAM3::ORBextractor / ORBmatcher adapters and instantiates each of the matcher's 13 search / fuse
// members with the reference's signatures (include/ORBmatcher.h:38-181, include/ORBextractor.h:52-87).
#define ORBX_WITH_OPENCV 1
#include "orbx/orbmatcher_adapter.hpp"

using namespace ORB_SLAM3;
typedef std::vector<MapPoint*> Points;

int (ORBmatcher::*const p01)(Frame&, const Points&, const float, const bool, const float) = &ORBmatcher::SearchByProjection;
int (ORBmatcher::*const p02)(Frame&, const Frame&, const float, const bool) = &ORBmatcher::SearchByProjection;
int (ORBmatcher::*const p03)(Frame&, KeyFrame*, const std::set<MapPoint*>&, const float, const int) = &ORBmatcher::SearchByProjection;
int (ORBmatcher::*const p04)(KeyFrame*, Sophus::Sim3<float>&, const Points&, Points&, int, float) = &ORBmatcher::SearchByProjection;
int (ORBmatcher::*const p05)(KeyFrame*, Sophus::Sim3<float>&, const Points&, const std::vector<KeyFrame*>&, Points&,
                             std::vector<KeyFrame*>&, int, float) = &ORBmatcher::SearchByProjection;
int (ORBmatcher::*const p06)(KeyFrame*, Frame&, Points&) = &ORBmatcher::SearchByBoW;
int (ORBmatcher::*const p07)(KeyFrame*, KeyFrame*, Points&) = &ORBmatcher::SearchByBoW;
int (ORBmatcher::*const p08)(Frame&, Frame&, std::vector<cv::Point2f>&, std::vector<int>&, int) = &ORBmatcher::SearchForInitialization;
int (ORBmatcher::*const p09)(KeyFrame*, KeyFrame*, std::vector<std::pair<size_t, size_t>>&, const bool, const bool) =
    &ORBmatcher::SearchForTriangulation;
int (ORBmatcher::*const p10)(KeyFrame*, KeyFrame*, Points&, const Sophus::Sim3f&, const float) = &ORBmatcher::SearchBySim3;
int (ORBmatcher::*const p11)(KeyFrame*, const Points&, const float, const bool) = &ORBmatcher::Fuse;
int (ORBmatcher::*const p12)(KeyFrame*, Sophus::Sim3f&, const Points&, float, Points&) = &ORBmatcher::Fuse;
int (*const p13)(const cv::Mat&, const cv::Mat&) = &ORBmatcher::DescriptorDistance;
int (ORBextractor::*const e01)(cv::InputArray, cv::InputArray, std::vector<cv::KeyPoint>&, cv::OutputArray, std::vector<int>&) =
    &ORBextractor::operator();
