// C++ host side above the C ABI (include/orbx.h): RAII mirror of ORB_SLAM3::ORBextractor
// (include/ORBextractor.h:42-121) with the same constructor arguments, getters, public
// members (mvImagePyramid, mvDynamicArea) and operator() contract.  Header only.
//
// orbx::Extractor works on plain buffers and is what this repository compiles and tests.
// When OpenCV is available, define ORBX_WITH_OPENCV before including this header to get
// ORB_SLAM3::ORBextractor with the reference's exact cv-typed signature on top of it; that
// class is a drop-in for src/ORBextractor.cc (Frame, Tracking keep calling it unchanged).
#ifndef ORBX_EXTRACTOR_HPP_
#define ORBX_EXTRACTOR_HPP_

#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "../orbx.h"

namespace orbx {

struct Error : std::runtime_error {
    int status;
    Error(int s, const char* m) : std::runtime_error(std::string("orbx: ") + m), status(s) {}
};
inline void check(orbx_status s) {
    if (s != ORBX_OK) throw Error(s, orbx_last_error());
}

// Borderless host copy of one pyramid level (ORBextractor::mvImagePyramid element).
struct Level {
    int width = 0, height = 0;
    std::vector<uint8_t> pixels;   // width * height, tightly packed
};

class Extractor {
public:
    // ORBextractor::ORBextractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST)
    // (ORBextractor.cc:477).  The device buffers are sized lazily from the first image.
    Extractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST, int device = -1)
        : nfeatures_(nfeatures), scaleFactor_(scaleFactor), nlevels_(nlevels), iniTh_(iniThFAST), minTh_(minThFAST),
          device_(device) {
        // the tables are host arithmetic on the constructor arguments: available before any image or device exists
        sf_.resize(nlevels_); isf_.resize(nlevels_); s2_.resize(nlevels_); is2_.resize(nlevels_);
        check(orbx_scale_tables(nfeatures_, scaleFactor_, nlevels_, sf_.data(), isf_.data(), s2_.data(), is2_.data(), nullptr));
    }
    ~Extractor() { if (h_) orbx_extractor_destroy(h_); }
    Extractor(const Extractor&) = delete;
    Extractor& operator=(const Extractor&) = delete;

    int GetLevels() const { return nlevels_; }
    float GetScaleFactor() const { return scaleFactor_; }
    std::vector<float> GetScaleFactors() const { return sf_; }
    std::vector<float> GetInverseScaleFactors() const { return isf_; }
    std::vector<float> GetScaleSigmaSquares() const { return s2_; }
    std::vector<float> GetInverseScaleSigmaSquares() const { return is2_; }

    // operator(): returns monoIndex, or -1 for an empty image (ORBextractor.cc:1271-1399).
    // keypoints / descriptors are resized to the number of keypoints (descriptors: n x 32).
    int operator()(const uint8_t* image, int width, int height, ptrdiff_t stride, std::vector<orbx_keypoint>& keypoints,
                   std::vector<uint8_t>& descriptors, const std::vector<int>& vLappingArea) {
        if (!image || width <= 0 || height <= 0) { keypoints.clear(); descriptors.clear(); return -1; }
        ensure(width, height);
        int cap = orbx_extractor_max_keypoints(h_);
        int n = 0, mono = 0;
        for (;;) {
            keypoints.resize(cap);
            descriptors.resize((size_t)cap * 32);
            const orbx_status s = orbx_extract(h_, image, width, height, stride, vLappingArea.at(0), vLappingArea.at(1),
                                               mvDynamicArea.empty() ? nullptr : mvDynamicArea.data(),
                                               (int)mvDynamicArea.size(), keypoints.data(), descriptors.data(), cap, &n, &mono);
            // the maximum follows the image size (see orbx_extractor_max_keypoints): another size may need larger arrays
            const int now = orbx_extractor_max_keypoints(h_);
            if (s == ORBX_ERR_CAPACITY && now > cap) { cap = now; continue; }
            check(s);
            break;
        }
        keypoints.resize(n);
        descriptors.resize((size_t)n * 32);
        if (keepPyramid) fetch_pyramid();
        return mono;
    }

    // The same call with the output storage supplied by the caller: keypointStorage(capacity) / descriptorStorage(capacity)
    // return buffers of `capacity` keypoints / capacity x 32 bytes; *n = keypoints written.  Returns monoIndex.
    template <class KpAlloc, class DescAlloc>
    int extract_into(const uint8_t* image, int width, int height, ptrdiff_t stride, const std::vector<int>& vLappingArea,
                     KpAlloc keypointStorage, DescAlloc descriptorStorage, int* n) {
        *n = 0;
        if (!image || width <= 0 || height <= 0) return -1;
        ensure(width, height);
        int cap = orbx_extractor_max_keypoints(h_);
        int mono = 0;
        for (;;) {
            const orbx_status s = orbx_extract(h_, image, width, height, stride, vLappingArea.at(0), vLappingArea.at(1),
                                               mvDynamicArea.empty() ? nullptr : mvDynamicArea.data(),
                                               (int)mvDynamicArea.size(), keypointStorage(cap), descriptorStorage(cap), cap, n, &mono);
            const int now = orbx_extractor_max_keypoints(h_);
            if (s == ORBX_ERR_CAPACITY && now > cap) { cap = now; continue; }     // another image size: larger storage
            check(s);
            break;
        }
        return mono;
    }

    orbx_extractor* handle() { return h_; }

    // mvImagePyramid of the last call, copied from the device (done by operator() itself when keepPyramid is set)
    void fetch_pyramid() {
        mvImagePyramid.resize(nlevels_);
        for (int l = 0; l < nlevels_; ++l) {
            Level& L = mvImagePyramid[l];
            check(orbx_pyramid_level(h_, 0, l, nullptr, 0, &L.width, &L.height));
            L.pixels.resize((size_t)L.width * L.height);
            check(orbx_pyramid_level(h_, 0, l, L.pixels.data(), L.pixels.size(), &L.width, &L.height));
        }
    }

    // public members of the reference class
    std::vector<Level> mvImagePyramid;        // filled after each call when keepPyramid is set
    std::vector<orbx_rect> mvDynamicArea;     // ORBextractor.h:95-96
    bool keepPyramid = false;                  // Frame::ComputeStereoMatches reads the pyramid only in the
                                               // CPU path; orbx_stereo_match uses the device copy directly

private:
    void ensure(int w, int h) {
        if (h_ && w <= maxW_ && h <= maxH_) return;
        if (h_) { orbx_extractor_destroy(h_); h_ = nullptr; }
        maxW_ = w > maxW_ ? w : maxW_;
        maxH_ = h > maxH_ ? h : maxH_;
        orbx_params p{nfeatures_, scaleFactor_, nlevels_, iniTh_, minTh_, maxW_, maxH_, 1, device_};
        check(orbx_extractor_create(&p, &h_));
    }
    int nfeatures_; float scaleFactor_; int nlevels_, iniTh_, minTh_, device_;
    int maxW_ = 0, maxH_ = 0;
    orbx_extractor* h_ = nullptr;
    std::vector<float> sf_, isf_, s2_, is2_;
};

// Input staging in front of the extractor: cv::remap with the rectification maps or cv::resize to the configured
// size (System.cc:355-376), then the grey conversion of Tracking::GrabImage* (Tracking.cc:1591-1619), in one
// kernel.  One instance per camera; the maps are the CV_32FC1 pair of cv::initUndistortRectifyMap.
class InputStager {
public:
    // rectification: mapX / mapY have dstWidth x dstHeight entries
    InputStager(int srcWidth, int srcHeight, int channels, bool rgb, const float* mapX, const float* mapY, int dstWidth,
                int dstHeight, int device = 0) {
        create(srcWidth, srcHeight, channels, rgb, ORBX_STAGE_REMAP, mapX, mapY, dstWidth, dstHeight, device);
    }
    // resize to dstWidth x dstHeight
    InputStager(int srcWidth, int srcHeight, int channels, bool rgb, int dstWidth, int dstHeight, int device = 0) {
        create(srcWidth, srcHeight, channels, rgb, ORBX_STAGE_RESIZE, nullptr, nullptr, dstWidth, dstHeight, device);
    }
    // grey conversion only
    InputStager(int srcWidth, int srcHeight, int channels, bool rgb, int device = 0) {
        create(srcWidth, srcHeight, channels, rgb, ORBX_STAGE_NONE, nullptr, nullptr, srcWidth, srcHeight, device);
    }
    ~InputStager() { orbx_stager_destroy(h_); }
    InputStager(const InputStager&) = delete;
    InputStager& operator=(const InputStager&) = delete;

    int width() const { return w_; }
    int height() const { return h2_; }
    // one host image (stride in bytes) -> the packed grey image ORBextractor receives
    std::vector<uint8_t> operator()(const uint8_t* image, ptrdiff_t stride) const {
        std::vector<uint8_t> out((size_t)w_ * h2_);
        check(orbx_stage(h_, image, stride, out.data(), w_));
        return out;
    }
    orbx_stager* handle() const { return h_; }

private:
    void create(int sw, int sh, int cn, bool rgb, int mode, const float* mx, const float* my, int dw, int dh, int device) {
        orbx_stage_params p{};
        p.src_width = sw; p.src_height = sh; p.channels = cn; p.rgb_order = rgb ? 1 : 0; p.mode = mode;
        p.dst_width = dw; p.dst_height = dh; p.map_x = mx; p.map_y = my; p.device = device;
        check(orbx_stager_create(&p, &h_));
        check(orbx_stager_output_size(h_, &w_, &h2_));
    }
    orbx_stager* h_ = nullptr;
    int w_ = 0, h2_ = 0;
};

}  // namespace orbx

#ifdef ORBX_WITH_OPENCV
#include <opencv2/core/core.hpp>
namespace ORB_SLAM3 {
// Same public surface as the reference class; link with -lorbx instead of compiling src/ORBextractor.cc.
class ORBextractor {
public:
    enum { HARRIS_SCORE = 0, FAST_SCORE = 1 };
    ORBextractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST)
        : impl_(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST) {}
    int operator()(cv::InputArray _image, cv::InputArray /*_mask*/, std::vector<cv::KeyPoint>& _keypoints,
                   cv::OutputArray _descriptors, std::vector<int>& vLappingArea) {
        if (_image.empty()) return -1;
        cv::Mat image = _image.getMat();
        CV_Assert(image.type() == CV_8UC1);
        static_assert(sizeof(cv::KeyPoint) == sizeof(orbx_keypoint), "cv::KeyPoint layout");
        impl_.mvDynamicArea.clear();
        for (const cv::Rect2i& r : mvDynamicArea) impl_.mvDynamicArea.push_back(orbx_rect{r.x, r.y, r.width, r.height});
        // the library writes straight into the caller's containers: keypoints sized to the handle's capacity and cut
        // to the count afterwards, descriptors into a capacity-row matrix of which the first n rows are returned
        int n = 0;
        const int mono = impl_.extract_into(image.data, image.cols, image.rows, (ptrdiff_t)image.step, vLappingArea,
                                            [&](int cap) { _keypoints.resize(cap); return reinterpret_cast<orbx_keypoint*>(_keypoints.data()); },
                                            [&](int cap) { desc_.create(cap, 32, CV_8U); return desc_.data; }, &n);
        _keypoints.resize(n);
        if (n == 0) _descriptors.release();
        else { _descriptors.create(n, 32, CV_8U); std::memcpy(_descriptors.getMat().data, desc_.data, (size_t)n * 32); }
        // Frame::ComputeStereoMatches of the reference indexes mvImagePyramid[kpL.octave] (Frame.cc:970, :1060-1075), so
        // the drop-in class fills it by default; an integration that routes stereo matching to orbx_stereo_match
        // (which reads the device copy) sets keepPyramid = false and saves the eight downloads per frame
        if (keepPyramid) {
            impl_.fetch_pyramid();
            mvImagePyramid.resize(impl_.mvImagePyramid.size());
            for (size_t l = 0; l < impl_.mvImagePyramid.size(); ++l) {
                orbx::Level& L = impl_.mvImagePyramid[l];
                mvImagePyramid[l] = cv::Mat(L.height, L.width, CV_8UC1, L.pixels.data());   // header over the host copy
            }
        }
        return mono;
    }
    bool keepPyramid = true;
    int GetLevels() { return impl_.GetLevels(); }
    float GetScaleFactor() { return impl_.GetScaleFactor(); }
    std::vector<float> GetScaleFactors() { return impl_.GetScaleFactors(); }
    std::vector<float> GetInverseScaleFactors() { return impl_.GetInverseScaleFactors(); }
    std::vector<float> GetScaleSigmaSquares() { return impl_.GetScaleSigmaSquares(); }
    std::vector<float> GetInverseScaleSigmaSquares() { return impl_.GetInverseScaleSigmaSquares(); }
    std::vector<cv::Mat> mvImagePyramid;
    std::vector<cv::Rect2i> mvDynamicArea;
    orbx::Extractor& impl() { return impl_; }
private:
    orbx::Extractor impl_;
    cv::Mat desc_;     // capacity rows, reused from call to call
};
}  // namespace ORB_SLAM3
#endif  // ORBX_WITH_OPENCV
#endif
