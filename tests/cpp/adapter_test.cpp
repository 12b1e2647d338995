// Exercises the C++ adapters exactly as SSVIO's Odometry / Frame do (src/Odometry.cpp:13-14,
// src/Frame.cpp:3-9): make_shared<ORBextractor>(...), Extract(gray, Mat(), kps, desc),
// MatchByBruteForce(desc1, desc2, matches).  Reads a raw 8-bit frame pair, writes results as raw
// bytes so the python test can compare them with the oracle.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <map>
#include <string>
#include "ssvio/Feature.h"
#include "ssvio/FeatureMatching.h"
#include "ssvio/FrameOps.h"

struct Config {                                   // stands in for SSVIO's Config (include/Config.h:26-30)
    typedef std::shared_ptr<Config> Ptr;
    std::map<std::string, double> kv;
    template <typename T> T GetParam(const std::string& k) { return (T)kv.at(k); }
};

static bool dump(const char* path, const void* p, size_t n) { FILE* f = fopen(path, "wb"); if (!f) return false; fwrite(p, 1, n, f); fclose(f); return true; }

int main(int argc, char** argv)
{
    if (argc < 6) { fprintf(stderr, "usage: %s frames.raw width height nfeatures outprefix [voc.txt depth.raw]\n", argv[0]); return 2; }
    const int w = atoi(argv[2]), h = atoi(argv[3]), nf = atoi(argv[4]);
    std::string out = argv[5];
    std::vector<unsigned char> raw((size_t)2 * w * h);
    FILE* f = fopen(argv[1], "rb");
    if (!f || fread(raw.data(), 1, raw.size(), f) != raw.size()) { fprintf(stderr, "cannot read frames\n"); return 2; }
    fclose(f);
    try {
        ORBextractor::Ptr orb = std::make_shared<ORBextractor>(nf, 1.2f, 8, 20, 7);
        Config::Ptr cfg = std::make_shared<Config>(); cfg->kv["Threshold"] = 50; cfg->kv["NNRatio"] = 0.6f;
        FeatureMatching::Ptr fm = std::make_shared<FeatureMatching>("../vocabulary/orbvoc.dbow3", cfg);
        cv::Mat a(h, w, CV_8UC1, raw.data(), (size_t)w), b(h, w, CV_8UC1, raw.data() + (size_t)w * h, (size_t)w);
        std::vector<cv::KeyPoint> ka, kb; cv::Mat da, db;
        orb->Extract(a, cv::Mat(), ka, da);                  // mvImagePyramid is filled by default, like the reference's (checked at the end)
        orb->Extract(b, cv::Mat(), kb, db);
        {   // a non-8UC1 image (the reference asserts, src/Feature.cpp:511): no exception, outputs empty, the status says why
            cv::Mat c3(h / 3, w, CV_8UC3, raw.data(), (size_t)w * 3);
            std::vector<cv::KeyPoint> kx(5); cv::Mat dx(4, 32, CV_8U);
            orb->Extract(c3, cv::Mat(), kx, dx);
            if (orb->lastStatus() != SSVIO_ERR_INVALID || !kx.empty() || !dx.empty()) { fprintf(stderr, "wrong-type contract broken\n"); return 1; }
            orb->Extract(a, cv::Mat(), kx, dx);
            if (orb->lastStatus() != SSVIO_OK || kx.size() != ka.size()) { fprintf(stderr, "extractor unusable after a wrong-type call\n"); return 1; }
        }
        std::vector<cv::DMatch> m;
        const bool ok = fm->MatchByBruteForce(db, da, m);
        std::vector<cv::DMatch> m2; fm->MatchByKnnRatio(db, da, m2);
        cv::Mat e; std::vector<cv::DMatch> m3; m3.resize(3);
        const bool okEmpty = fm->MatchByBruteForce(e, da, m3);
        if (okEmpty || !m3.empty()) { fprintf(stderr, "empty-input contract broken\n"); return 1; }
        // MatchByProject the way Odometry would call it: frame A's keypoints back-projected at made-up depths, frame B's
        // keypoints as 2-D candidates (include/FeatureMatching.h:30-32)
        std::vector<cv::Point3d> p3; std::vector<cv::Point2d> p2;
        const double fx = 260.0, fy = 260.0, cx = w / 2.0, cy = h / 2.0;
        for (size_t i = 0; i < ka.size(); ++i) {
            const double d = 1.0 + 0.5 * (double)(i % 7);
            p3.push_back(cv::Point3d((ka[i].pt.x - cx) / fx * d, (ka[i].pt.y - cy) / fy * d, d));
        }
        for (size_t i = 0; i < kb.size(); ++i) p2.push_back(cv::Point2d(kb[i].pt.x, kb[i].pt.y));
        cv::Mat K(3, 3, CV_64F), R(3, 3, CV_64F), t(3, 1, CV_64F);
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { K.at<double>(i, j) = 0; R.at<double>(i, j) = i == j ? 1.0 : 0.0; }
        K.at<double>(0, 0) = fx; K.at<double>(1, 1) = fy; K.at<double>(0, 2) = cx; K.at<double>(1, 2) = cy; K.at<double>(2, 2) = 1.0;
        t.at<double>(0, 0) = 0.02; t.at<double>(1, 0) = -0.01; t.at<double>(2, 0) = 0.0;
        double radian = 12.0;
        std::vector<cv::DMatch> m4;
        fm->MatchByProject(p3, da, p2, db, radian, K, R, t, m4);
        if (fm->lastStatus() != SSVIO_OK) { fprintf(stderr, "MatchByProject: %s\n", ssvio_last_error()); return 1; }
        if (argc >= 8) {
            // the live path of Odometry (src/Odometry.cpp:145-176, :205): MatchByDBoW on a vocabulary read from a text file,
            // then the correspondence lists, then the match chaining of Map
            FeatureMatching::Ptr fd = std::make_shared<FeatureMatching>(argv[6], cfg);
            if (!fd->HasVocabulary()) { fprintf(stderr, "vocabulary not loaded\n"); return 1; }
            std::vector<cv::DMatch> bow; bow.resize(2);
            const bool okb = fd->MatchByDBoW(db, da, bow);
            if (fd->lastStatus() != SSVIO_OK || okb != !bow.empty()) { fprintf(stderr, "MatchByDBoW: %s\n", ssvio_last_error()); return 1; }
            std::vector<cv::DMatch> untouched(2); cv::Mat none;
            if (fd->MatchByDBoW(none, da, untouched) || untouched.size() != 2) { fprintf(stderr, "MatchByDBoW early return must not clear\n"); return 1; }
            std::vector<uint16_t> dep((size_t)2 * w * h);
            FILE* g = fopen(argv[7], "rb");
            if (!g || fread(dep.data(), 2, dep.size(), g) != dep.size()) { fprintf(stderr, "cannot read depth\n"); return 2; }
            fclose(g);
            cv::Mat depA(h, w, CV_16UC1, dep.data(), (size_t)w * 2), depB(h, w, CV_16UC1, dep.data() + (size_t)w * h, (size_t)w * 2);
            const ssvio_camera cam = ssvio::MakeCamera(160.5912f, 120.4792f, 253.0589f, 254.1649f, 10.0f);
            ssvio::Correspondences c;
            if (!ssvio::AssembleCorrespondences(*fd, bow, ka, depA, kb, depB, cam, c)) { fprintf(stderr, "assemble: %s\n", ssvio_last_error()); return 1; }
            std::vector<cv::DMatch> last = m;                         // stands in for last_match_vec
            if (!ssvio::TrackMapPoints(*fd, last, c.final_good_match)) { fprintf(stderr, "track: %s\n", ssvio_last_error()); return 1; }
            std::vector<int> inm;
            if (!ssvio::InMatchVec(*fd, m, (int)kb.size(), inm)) { fprintf(stderr, "inmatch: %s\n", ssvio_last_error()); return 1; }
            // Map::ManageMapPoints (src/Map.cpp:125-149): `last` after TrackMapPoints, the observations of the previous key frame
            // stood in for by (1000 + j, last[j].trainIdx) pairs, world_trans = a fixed rigid motion (Eigen::Matrix4f::data() order)
            std::vector<std::pair<int, int>> observed;
            for (size_t j = 0; j < m.size(); j += 2) observed.push_back(std::make_pair(1000 + (int)j, m[j].queryIdx));
            const float wt[16] = {0.f, 1.f, 0.f, 0.f,  -1.f, 0.f, 0.f, 0.f,  0.f, 0.f, 1.f, 0.f,  0.25f, -0.5f, 1.5f, 1.f};
            std::vector<int> mnum, mpid; std::vector<ssvio::Point3f> npt;
            if (!ssvio::ManageMapPoints(*fd, kb, depB, cam, wt, last, observed, mnum, mpid, npt)) { fprintf(stderr, "manage: %s\n", ssvio_last_error()); return 1; }
            dump((out + ".mnum").c_str(), mnum.data(), mnum.size() * 4);
            dump((out + ".mpid").c_str(), mpid.data(), mpid.size() * 4);
            dump((out + ".npt").c_str(), npt.data(), npt.size() * 12);
            std::vector<unsigned> fvflat;                             // mFeatVec1_ flattened: node, count, indices...
            for (auto& e : fd->mFeatVec1_) { fvflat.push_back(e.first); fvflat.push_back((unsigned)e.second.size()); fvflat.insert(fvflat.end(), e.second.begin(), e.second.end()); }
            std::vector<double> bvflat;
            for (auto& e : fd->mBowVec1_) { bvflat.push_back((double)e.first); bvflat.push_back(e.second); }
            dump((out + ".bow").c_str(), bow.data(), bow.size() * sizeof(cv::DMatch));
            dump((out + ".fv1").c_str(), fvflat.data(), fvflat.size() * 4);
            dump((out + ".bv1").c_str(), bvflat.data(), bvflat.size() * 8);
            dump((out + ".l3").c_str(), c.last_frame_3d_point_set.data(), c.last_frame_3d_point_set.size() * 12);
            dump((out + ".c2").c_str(), c.cur_frame_2d_point_set.data(), c.cur_frame_2d_point_set.size() * 8);
            dump((out + ".fin").c_str(), c.final_good_match.data(), c.final_good_match.size() * sizeof(cv::DMatch));
            dump((out + ".trk").c_str(), last.data(), last.size() * sizeof(cv::DMatch));
            dump((out + ".inm").c_str(), inm.data(), inm.size() * 4);
            // OpenCV-compat cross-check, and Map::MapPointCloudFusion over two key frames (colour = the gray frame in all three channels)
            std::vector<cv::DMatch> m5;
            fm->MatchByCrossCheckCV(db, da, m5);
            if (fm->lastStatus() != SSVIO_OK) { fprintf(stderr, "MatchByCrossCheckCV: %s\n", ssvio_last_error()); return 1; }
            dump((out + ".m5").c_str(), m5.data(), m5.size() * sizeof(cv::DMatch));
            std::vector<uint8_t> bgrA((size_t)3 * w * h), bgrB((size_t)3 * w * h);
            for (size_t i = 0; i < (size_t)w * h; ++i) for (int ch = 0; ch < 3; ++ch) { bgrA[3 * i + ch] = a.data[i]; bgrB[3 * i + ch] = b.data[i]; }
            cv::Mat cA(h, w, CV_8UC3, bgrA.data(), (size_t)3 * w), cB(h, w, CV_8UC3, bgrB.data(), (size_t)3 * w);
            const float ident[16] = {1.f, 0.f, 0.f, 0.f,  0.f, 1.f, 0.f, 0.f,  0.f, 0.f, 1.f, 0.f,  0.f, 0.f, 0.f, 1.f};
            std::vector<ssvio::PointXYZRGB> map;
            if (!ssvio::MapClear(*fd) || !ssvio::MapFuseFrame(*fd, depA, cA, cam, 10.0f, ident, 0.05f) ||
                !ssvio::MapFuseFrame(*fd, depB, cB, cam, 10.0f, wt, 0.05f) || !ssvio::MapFetch(*fd, map)) { fprintf(stderr, "map fusion: %s\n", ssvio_last_error()); return 1; }
            dump((out + ".map").c_str(), map.data(), map.size() * sizeof(ssvio::PointXYZRGB));
            if (argc >= 9) {
                // the same vocabulary as DBoW3's uncompressed binary stream (what Vocabulary::load tries first): same matches
                FeatureMatching::Ptr fb = std::make_shared<FeatureMatching>(argv[8], cfg);
                std::vector<cv::DMatch> bow2;
                if (!fb->HasVocabulary()) { fprintf(stderr, "binary vocabulary not loaded\n"); return 1; }
                fb->MatchByDBoW(db, da, bow2);
                if (bow2.size() != bow.size() || (bow.size() && memcmp(bow2.data(), bow.data(), bow.size() * sizeof(cv::DMatch)) != 0)) {
                    fprintf(stderr, "binary and text vocabulary disagree\n"); return 1;
                }
                FeatureMatching::Ptr fbad = std::make_shared<FeatureMatching>(argv[1], cfg);      // the raw frames: not a vocabulary at all
                if (fbad->HasVocabulary()) { fprintf(stderr, "garbage accepted as a vocabulary\n"); return 1; }
                for (int a = 9; a < argc; ++a) {                      // corrupt binary streams: refused, never guessed at
                    FeatureMatching::Ptr fc = std::make_shared<FeatureMatching>(argv[a], cfg);
                    if (fc->HasVocabulary()) { fprintf(stderr, "corrupt vocabulary %s accepted\n", argv[a]); return 1; }
                }
                printf("binary_voc=same\n");
            }
            if (getenv("SSVIO_ADAPTER_TIME")) {
                // per-call cost of the unchanged C++ interfaces (pageable cv::Mat / std::vector in and out), 50 calls each after 5 warm-ups
                auto timeit = [](const char* name, const std::function<void()>& fn) {
                    for (int i = 0; i < 5; ++i) fn();
                    const auto t0 = std::chrono::steady_clock::now();
                    for (int i = 0; i < 50; ++i) fn();
                    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 50;
                    printf("time %-28s %8.1f us per call\n", name, us);
                };
                std::vector<cv::KeyPoint> kt; cv::Mat dt; std::vector<cv::DMatch> mt; ssvio::Correspondences ct;
                timeit("ORBextractor::Extract", [&] { orb->Extract(b, cv::Mat(), kt, dt); });         // with mvImagePyramid (the default)
                orb->keepPyramid = false;
                timeit("Extract, keepPyramid=false", [&] { orb->Extract(b, cv::Mat(), kt, dt); });
                orb->keepPyramid = true;
                timeit("MatchByBruteForce", [&] { fm->MatchByBruteForce(db, da, mt); });
                timeit("MatchByDBoW", [&] { fd->MatchByDBoW(db, da, mt); });
                timeit("MatchByProject", [&] { fm->MatchByProject(p3, da, p2, db, radian, K, R, t, mt); });
                timeit("AssembleCorrespondences", [&] { ssvio::AssembleCorrespondences(*fd, bow, ka, depA, kb, depB, cam, ct); });
                timeit("TrackMapPoints", [&] { std::vector<cv::DMatch> l2 = m; ssvio::TrackMapPoints(*fd, l2, c.final_good_match); });
                timeit("ManageMapPoints", [&] { ssvio::ManageMapPoints(*fd, kb, depB, cam, wt, last, observed, mnum, mpid, npt); });
            }
            printf("bow=%zu pre=%zu final=%zu track=%zu\n", bow.size(), c.pre_good_match.size(), c.final_good_match.size(), last.size());
        }
        dump((out + ".p3").c_str(), p3.data(), p3.size() * sizeof(cv::Point3d));
        dump((out + ".m4").c_str(), m4.data(), m4.size() * sizeof(cv::DMatch));
        dump((out + ".ka").c_str(), ka.data(), ka.size() * sizeof(cv::KeyPoint));
        dump((out + ".kb").c_str(), kb.data(), kb.size() * sizeof(cv::KeyPoint));
        dump((out + ".da").c_str(), da.data, (size_t)da.rows * 32);
        dump((out + ".db").c_str(), db.data, (size_t)db.rows * 32);
        dump((out + ".m").c_str(), m.data(), m.size() * sizeof(cv::DMatch));
        dump((out + ".m2").c_str(), m2.data(), m2.size() * sizeof(cv::DMatch));
        printf("ok=%d ka=%zu kb=%zu m=%zu m2=%zu m4=%zu levels=%d pyr0=%dx%d\n", (int)ok, ka.size(), kb.size(), m.size(), m2.size(), m4.size(),
               orb->GetLevels(), orb->mvImagePyramid[0].cols, orb->mvImagePyramid[0].rows);
    } catch (const std::exception& ex) {
        fprintf(stderr, "exception: %s\n", ex.what());
        return 3;
    }
    return 0;
}
