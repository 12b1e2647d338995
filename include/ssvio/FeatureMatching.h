// FeatureMatching.h -- drop-in for SSVIO's include/FeatureMatching.h:13-54 on the gfx950 library.
// The constructor keeps its shape (vocabulary path + config); the config is anything with
// GetParam<T>(key) like SSVIO's Config (include/Config.h:26-30).  MatchByDBoW needs DBoW3 and its
// vocabulary (un-vendored, blob missing from the reference tree): it is declared and returns false
// without touching `matches`' contents beyond what the reference's early return does.
#pragma once
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>
#include "cv_compat.h"
#include "../ssvio_gpu.h"

class FeatureMatching {
public:
    typedef std::shared_ptr<FeatureMatching> Ptr;

    template <class ConfigPtr>
    FeatureMatching(const char* voc_dataset, ConfigPtr config, int device = 0)
        : voc_(voc_dataset ? voc_dataset : "")
    {
        mTh_ = config->template GetParam<int>("Threshold");          // src/FeatureMatching.cpp:10
        mNNRatio_ = config->template GetParam<float>("NNRatio");      // :11
        init(device);
    }
    FeatureMatching(int threshold, float nnRatio, int device = 0) : mTh_(threshold), mNNRatio_(nnRatio) { init(device); }
    ~FeatureMatching() { ssvio_matcher_destroy(m_); }
    FeatureMatching(const FeatureMatching&) = delete;
    FeatureMatching& operator=(const FeatureMatching&) = delete;

    // include/FeatureMatching.h:24, src/FeatureMatching.cpp:20-55
    bool MatchByBruteForce(cv::Mat& desp1, cv::Mat& desp2, std::vector<cv::DMatch>& matches)
    { return run(ssvio_match_bruteforce, desp1, desp2, matches); }
    // brute-force form of the 2-NN + Threshold/NNRatio loop, src/FeatureMatching.cpp:85-112
    bool MatchByKnnRatio(cv::Mat& desp1, cv::Mat& desp2, std::vector<cv::DMatch>& matches)
    { return run(ssvio_match_knn2_ratio, desp1, desp2, matches); }
    bool MatchByCrossCheck(cv::Mat& desp1, cv::Mat& desp2, std::vector<cv::DMatch>& matches)
    { return run(ssvio_match_crosscheck, desp1, desp2, matches); }
    // include/FeatureMatching.h:27 -- out of scope (DBoW3); mirrors the early return of :61-63
    bool MatchByDBoW(cv::Mat& desp1, cv::Mat& desp2, std::vector<cv::DMatch>& matches)
    {
        if (!(desp1.rows > 0 && desp2.rows > 0)) return false;
        matches.clear();
        status_ = SSVIO_ERR_INVALID;
        return false;
    }
    // include/FeatureMatching.h:30-32, src/FeatureMatching.cpp:128-189; K, R 3x3 and t 3x1 are CV_64F like the reference's
    bool MatchByProject(std::vector<cv::Point3d>& vp3d1, cv::Mat& desp1, std::vector<cv::Point2d>& vp2d2, cv::Mat& desp2,
                        double& radian, cv::Mat& K, cv::Mat& R, cv::Mat& t, std::vector<cv::DMatch>& matches)
    {
        matches.clear();                                              // :132
        status_ = SSVIO_OK;
        if (vp3d1.empty()) return false;
        if ((int)vp3d1.size() != desp1.rows || (int)vp2d2.size() != desp2.rows || desp1.cols != 32 || (desp2.rows > 0 && desp2.cols != 32)) {
            status_ = SSVIO_ERR_INVALID; return false;
        }
        static_assert(sizeof(cv::Point3d) == 24 && sizeof(cv::Point2d) == 16, "cv::Point3d / Point2d layout");
        double k[9], r[9], tr[3];
        for (int i = 0; i < 3; ++i) {
            for (int j = 0; j < 3; ++j) { k[3 * i + j] = K.at<double>(i, j); r[3 * i + j] = R.at<double>(i, j); }
            tr[i] = t.at<double>(i, 0);
        }
        matches.resize(vp3d1.size());
        int n = 0;
        status_ = ssvio_match_project(m_, &vp3d1[0].x, desp1.ptr(0), desp1.rows, (int)desp1.step,
                                      vp2d2.empty() ? nullptr : &vp2d2[0].x, desp2.rows > 0 ? desp2.ptr(0) : nullptr, desp2.rows, (int)desp2.step,
                                      radian, k, r, tr, (ssvio_dmatch*)matches.data(), (int)matches.size(), &n);
        matches.resize(status_ == SSVIO_OK ? n : 0);
        return matches.size() > 0;                                    // :188
    }
    int lastStatus() const { return status_; }
    ssvio_matcher* handle() { return m_; }

private:
    typedef int (*MatchFn)(ssvio_matcher*, const uint8_t*, int, int, const uint8_t*, int, int, ssvio_dmatch*, int, int*);
    void init(int device)
    {
        if (ssvio_matcher_create(mTh_, mNNRatio_, device, &m_) != SSVIO_OK) throw std::runtime_error(std::string("FeatureMatching: ") + ssvio_last_error());
    }
    bool run(MatchFn fn, cv::Mat& d1, cv::Mat& d2, std::vector<cv::DMatch>& matches)
    {
        matches.clear();                                              // :22
        status_ = SSVIO_OK;
        if (!(d1.rows > 0 && d2.rows > 0)) return false;              // :25-27
        if (d1.cols != 32 || d2.cols != 32) { status_ = SSVIO_ERR_INVALID; return false; }
        static_assert(sizeof(cv::DMatch) == sizeof(ssvio_dmatch), "cv::DMatch layout");
        matches.resize(d1.rows);
        int n = 0;
        status_ = fn(m_, d1.ptr(0), d1.rows, (int)d1.step, d2.ptr(0), d2.rows, (int)d2.step, (ssvio_dmatch*)matches.data(), d1.rows, &n);
        matches.resize(status_ == SSVIO_OK ? n : 0);
        return matches.size() > 0;                                    // :54
    }
    std::string voc_;
    int mTh_; float mNNRatio_;
    ssvio_matcher* m_ = nullptr;
    int status_ = SSVIO_OK;
};
