// Feature.h -- drop-in for SSVIO's include/Feature.h:32-103: class ORBextractor with the same
// constructor, Extract signature, public mvImagePyramid and getters, implemented on the
// gfx950 library through the C ABI (include/ssvio_gpu.h).  ExtractorNode (include/Feature.h:14-29) is
// an implementation detail of the reference's quadtree and has no counterpart here.
#pragma once
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>
#include "cv_compat.h"
#include "../ssvio_gpu.h"

class ORBextractor {
public:
    typedef std::shared_ptr<ORBextractor> Ptr;
    enum { HARRIS_SCORE = 0, FAST_SCORE = 1 };

    // include/Feature.h:38 -- throws std::runtime_error if no GPU/library (there is no CPU fallback)
    ORBextractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST, int device = 0)
        : nfeatures(nfeatures), scaleFactor(scaleFactor), nlevels(nlevels), iniThFAST(iniThFAST), minThFAST(minThFAST)
    {
        ssvio_orb_params p = {nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST};
        if (ssvio_orb_create(&p, device, &h_) != SSVIO_OK) throw std::runtime_error(std::string("ORBextractor: ") + ssvio_last_error());
        mvScaleFactor.resize(nlevels); mvInvScaleFactor.resize(nlevels); mvLevelSigma2.resize(nlevels); mvInvLevelSigma2.resize(nlevels);
        mnFeaturesPerLevel.resize(nlevels);
        ssvio_orb_get_tables(h_, mvScaleFactor.data(), mvInvScaleFactor.data(), mvLevelSigma2.data(), mvInvLevelSigma2.data(), mnFeaturesPerLevel.data());
        mvImagePyramid.resize(nlevels);
    }
    ~ORBextractor() { ssvio_orb_destroy(h_); }
    ORBextractor(const ORBextractor&) = delete;
    ORBextractor& operator=(const ORBextractor&) = delete;

    // include/Feature.h:42-43, src/Feature.cpp:504-568.  `mask` is ignored exactly as in the reference.
    // Errors (image too small for the pyramid, HIP failure) leave the outputs empty; lastStatus() tells why.
    void Extract(cv::InputArray image, cv::InputArray /*mask*/, std::vector<cv::KeyPoint>& keypoints, cv::OutputArray descriptors)
    {
        status_ = SSVIO_OK;
        cv::Mat img = image.getMat();
        if (img.empty()) return;                                   // :507-508 silent return
        // assert(image.type() == CV_8UC1), :511 (active in the reference's Release build: it aborts).  The adapter does not throw
        // across the reference's exception-free interface: outputs empty, lastStatus() = SSVIO_ERR_INVALID.
        if (img.type() != CV_8UC1) { status_ = SSVIO_ERR_INVALID; keypoints.clear(); descriptors.release(); return; }
        if (keepPyramid != exporting_) { ssvio_orb_set_export_pyramid(h_, keepPyramid ? 1 : 0); exporting_ = keepPyramid; }
        const int cap = ssvio_orb_max_keypoints(h_);
        kp_.resize(cap); desc_.resize((size_t)cap * 32);
        int n = 0;
        status_ = ssvio_orb_extract(h_, img.ptr(0), img.cols, img.rows, (int)img.step, kp_.data(), desc_.data(), cap, &n);
        keypoints.clear();
        if (status_ == SSVIO_OK && keepPyramid) fetchPyramid();                     // ComputePyramid runs before anything can return (:514)
        if (status_ != SSVIO_OK || n == 0) { descriptors.release(); return; }     // :526-527
        descriptors.create(n, 32, CV_8U);                                         // :530
        cv::Mat d = descriptors.getMat();
        keypoints.resize(n);
        static_assert(sizeof(cv::KeyPoint) == sizeof(ssvio_keypoint), "cv::KeyPoint layout");
        std::memcpy((void*)keypoints.data(), kp_.data(), (size_t)n * sizeof(ssvio_keypoint));
        for (int i = 0; i < n; ++i) std::memcpy(d.ptr(i), desc_.data() + (size_t)i * 32, 32);
    }

    // public member of the reference (include/Feature.h:45), filled by every Extract like there (src/Feature.cpp:514, :800-824):
    // level l is the ROI at (19, 19) of a buffer with the 19-px BORDER_REFLECT_101 rim.  Nothing outside Feature.cpp reads it in
    // the reference; here a kernel stores the levels into pinned host memory beside the FAST / quadtree stages (a few microseconds
    // per call at VGA): a caller that never looks at it may set keepPyramid = false (interface table).
    std::vector<cv::Mat> mvImagePyramid;
    bool keepPyramid = true;

    int inline GetLevels() { return nlevels; }
    float inline GetScaleFactor() { return (float)scaleFactor; }
    std::vector<float> inline GetScaleFactors() { return mvScaleFactor; }
    std::vector<float> inline GetInvScaleFactor() { return mvInvScaleFactor; }
    std::vector<float> inline GetLevelSigmaSquares() { return mvLevelSigma2; }
    std::vector<float> inline GetInvLevelSigmaSquares() { return mvInvLevelSigma2; }

    int lastStatus() const { return status_; }
    ssvio_orb* handle() { return h_; }

protected:
    void fetchPyramid()
    {
        for (int l = 0; l < nlevels; ++l) {
            const uint8_t* base = nullptr; int stride = 0, w = 0, h = 0;
            if (ssvio_orb_exported_pyramid_level(h_, l, &base, &stride, &w, &h) != SSVIO_OK) return;
            // the reference's entry is the ROI at (19,19) of the padded buffer (src/Feature.cpp:806-808); the buffer is the handle's
            // pinned host memory, rewritten by the next Extract exactly as the reference rewrites its own
            mvImagePyramid[l] = cv::Mat(h, w, CV_8UC1, const_cast<uint8_t*>(base) + (size_t)19 * stride + 19, (size_t)stride);
        }
    }
    int nfeatures; double scaleFactor; int nlevels; int iniThFAST; int minThFAST;
    std::vector<int> mnFeaturesPerLevel;
    std::vector<float> mvScaleFactor, mvInvScaleFactor, mvLevelSigma2, mvInvLevelSigma2;
    ssvio_orb* h_ = nullptr;
    int status_ = SSVIO_OK;
    std::vector<ssvio_keypoint> kp_; std::vector<uint8_t> desc_;
    bool exporting_ = false;
};
