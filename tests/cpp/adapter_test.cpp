// Exercises the C++ adapters exactly as SSVIO's Odometry / Frame do (src/Odometry.cpp:13-14,
// src/Frame.cpp:3-9): make_shared<ORBextractor>(...), Extract(gray, Mat(), kps, desc),
// MatchByBruteForce(desc1, desc2, matches).  Reads a raw 8-bit frame pair, writes results as raw
// bytes so the python test can compare them with the oracle.
#include <cstdio>
#include <cstdlib>
#include <map>
#include <string>
#include "ssvio/Feature.h"
#include "ssvio/FeatureMatching.h"

struct Config {                                   // stands in for SSVIO's Config (include/Config.h:26-30)
    typedef std::shared_ptr<Config> Ptr;
    std::map<std::string, double> kv;
    template <typename T> T GetParam(const std::string& k) { return (T)kv.at(k); }
};

static bool dump(const char* path, const void* p, size_t n) { FILE* f = fopen(path, "wb"); if (!f) return false; fwrite(p, 1, n, f); fclose(f); return true; }

int main(int argc, char** argv)
{
    if (argc < 6) { fprintf(stderr, "usage: %s frames.raw width height nfeatures outprefix\n", argv[0]); return 2; }
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
        orb->keepPyramid = true;
        orb->Extract(a, cv::Mat(), ka, da);
        orb->Extract(b, cv::Mat(), kb, db);
        std::vector<cv::DMatch> m;
        const bool ok = fm->MatchByBruteForce(db, da, m);
        std::vector<cv::DMatch> m2; fm->MatchByKnnRatio(db, da, m2);
        cv::Mat e; std::vector<cv::DMatch> m3; m3.resize(3);
        const bool okEmpty = fm->MatchByBruteForce(e, da, m3);
        if (okEmpty || !m3.empty()) { fprintf(stderr, "empty-input contract broken\n"); return 1; }
        dump((out + ".ka").c_str(), ka.data(), ka.size() * sizeof(cv::KeyPoint));
        dump((out + ".kb").c_str(), kb.data(), kb.size() * sizeof(cv::KeyPoint));
        dump((out + ".da").c_str(), da.data, (size_t)da.rows * 32);
        dump((out + ".db").c_str(), db.data, (size_t)db.rows * 32);
        dump((out + ".m").c_str(), m.data(), m.size() * sizeof(cv::DMatch));
        dump((out + ".m2").c_str(), m2.data(), m2.size() * sizeof(cv::DMatch));
        printf("ok=%d ka=%zu kb=%zu m=%zu m2=%zu levels=%d pyr0=%dx%d\n", (int)ok, ka.size(), kb.size(), m.size(), m2.size(),
               orb->GetLevels(), orb->mvImagePyramid[0].cols, orb->mvImagePyramid[0].rows);
    } catch (const std::exception& ex) {
        fprintf(stderr, "exception: %s\n", ex.what());
        return 3;
    }
    return 0;
}
