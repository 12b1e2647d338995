// FeatureMatching.h -- drop-in for SSVIO's include/FeatureMatching.h:13-54 on the gfx950 library.
// The constructor keeps its shape (vocabulary path + config); the config is anything with
// GetParam<T>(key) like SSVIO's Config (include/Config.h:26-30).  The vocabulary (DBoW3::Vocabulary in the
// reference, src/FeatureMatching.cpp:8-9) is read from voc_dataset the way DBoW3's Vocabulary::load tries it:
// the uncompressed binary stream when the file starts with DBoW3's signature (toStream / fromStream layout, restated
// from DBoW3's published source -- no .dbow3 file exists here to check it against), else the text layout (DBoW3
// load_fromtxt / ORBvoc.txt: "k L scoring weighting", then "parent is_leaf d0..d31 weight" per node); or it is handed
// over as arrays with SetVocabulary.  The QuickLZ-compressed stream and the YAML container are not read.  Without
// a vocabulary MatchByDBoW returns false (lastStatus() tells why).
#pragma once
#include <algorithm>
#include <cmath>
#include <fstream>
#include <map>
#include <memory>
#include <sstream>
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
        LoadVocabulary(voc_.c_str());                                 // :8-9 (a missing / unreadable file leaves no vocabulary)
    }
    FeatureMatching(int threshold, float nnRatio, int device = 0) : mTh_(threshold), mNNRatio_(nnRatio) { init(device); }
    ~FeatureMatching() { ssvio_matcher_destroy(m_); ssvio_vocab_destroy(v_); }
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
    // cv::BFMatcher(NORM_HAMMING, true).match(desp1, desp2) the way OpenCV 3.x filters it ("OpenCV-compat")
    bool MatchByCrossCheckCV(cv::Mat& desp1, cv::Mat& desp2, std::vector<cv::DMatch>& matches)
    { return run(ssvio_match_crosscheck_cv, desp1, desp2, matches); }
    // DBoW3's typedefs: BowVector = map<WordId, WordValue>, FeatureVector = map<NodeId, vector<unsigned>>
    typedef std::map<unsigned int, double> BowVector;
    typedef std::map<unsigned int, std::vector<unsigned int> > FeatureVector;

    // the vocabulary as a node table (include/ssvio_gpu.h, ssvio_vocab_create): node 0 = root, parent[n] < n
    bool SetVocabulary(int k, int L, const std::vector<int32_t>& parent, const std::vector<uint8_t>& desc,
                       const std::vector<double>& weight, const std::vector<int32_t>& word, int scoring = 0, int weighting = 0)
    {
        ssvio_vocab_destroy(v_); v_ = nullptr;
        const int n = (int)parent.size();
        if (desc.size() != (size_t)n * 32 || weight.size() != (size_t)n || word.size() != (size_t)n) { status_ = SSVIO_ERR_INVALID; return false; }
        status_ = ssvio_vocab_create(device_, k, L, n, parent.data(), desc.data(), weight.data(), word.data(), scoring, weighting, &v_);
        scoring_ = scoring; weighting_ = weighting;
        return status_ == SSVIO_OK;
    }
    bool LoadVocabularyTxt(const char* path)
    {
        std::ifstream f(path);
        if (!f) return false;
        int k = 0, L = 0, scoring = 0, weighting = 0;
        std::string line;
        if (!std::getline(f, line)) return false;
        { std::istringstream ls(line); if (!(ls >> k >> L >> scoring >> weighting)) return false; }
        if (k < 1 || k > 65536 || L < 0 || L > 64 || scoring < 0 || scoring > 5 || weighting < 0 || weighting > 3) return false;
        std::vector<int32_t> parent(1, -1), word(1, -1); std::vector<uint8_t> desc(32, 0); std::vector<double> weight(1, 0.0);
        int nwords = 0;
        while (std::getline(f, line)) {
            std::istringstream ls(line);
            int pid, leaf;
            if (!(ls >> pid >> leaf)) continue;
            for (int i = 0; i < 32; ++i) { int b; if (!(ls >> b)) return false; desc.push_back((uint8_t)b); }
            double w; if (!(ls >> w)) return false;
            parent.push_back(pid); weight.push_back(w); word.push_back(leaf > 0 ? nwords++ : -1);
        }
        return SetVocabulary(k, L, parent, desc, weight, word, scoring, weighting);
    }
    // DBoW3's uncompressed binary stream: u64 88877711233 | u8 compressed | u32 nnodes | i32 k, L, scoring, weighting |
    // nnodes-1 x {u32 id, u32 parent, f64 weight, 3 x i32 {cols, rows, type}, 32 bytes} | u32 nwords | nwords x {u32 word, u32 node}
    bool LoadVocabularyBinary(const char* path)
    {
        std::ifstream f(path, std::ios::binary);
        if (!f) return false;
        uint64_t sig = 0; uint8_t compressed = 1; uint32_t n = 0; int32_t hd[4] = {0, 0, 0, 0};
        f.read((char*)&sig, 8); f.read((char*)&compressed, 1); f.read((char*)&n, 4); f.read((char*)hd, 16);
        if (!f || sig != 88877711233ull || compressed != 0 || n < 1) return false;
        f.seekg(0, std::ios::end);
        const uint64_t fileBytes = (uint64_t)f.tellg();
        if (fileBytes < 29 + (uint64_t)(n - 1) * 60 + 4) return false;  // the node records must fit the file before anything is sized by n
        f.seekg(29, std::ios::beg);
        std::vector<int32_t> parent(n, -2), word(n, -1); std::vector<uint8_t> desc((size_t)n * 32, 0); std::vector<double> weight(n, 0.0);
        std::vector<uint32_t> lastChild(n, 0); std::vector<uint8_t> inner(n, 0);
        parent[0] = -1;
        for (uint32_t i = 1; i < n; ++i) {
            uint32_t id = 0, pid = 0; double w = 0; int32_t h3[3] = {0, 0, 0};
            f.read((char*)&id, 4); f.read((char*)&pid, 4); f.read((char*)&w, 8); f.read((char*)h3, 12);
            if (!f || id < 1 || id >= n || pid >= id || parent[id] != -2) return false;        // a tree numbered parent-first, every id once
            if (h3[0] < 0 || h3[1] < 0 || h3[2] < 0 || h3[0] + h3[1] + h3[2] != 33 || h3[0] * h3[1] + h3[1] * h3[2] + h3[0] * h3[2] != 32) return false;   // {32 cols, 1 row, CV_8U = 0}, any order
            if (id <= lastChild[pid]) return false;                    // children are tried in file order: must be ascending id
            lastChild[pid] = id; inner[pid] = 1;
            parent[id] = (int32_t)pid; weight[id] = w;
            f.read((char*)&desc[(size_t)id * 32], 32);
            if (!f) return false;
        }
        uint32_t nw = 0; f.read((char*)&nw, 4);
        if (!f || nw > n) return false;
        uint32_t leaves = 0;
        for (uint32_t i = (n > 1 ? 1 : 0); i < n; ++i) leaves += inner[i] ? 0 : 1;
        if (nw != leaves) return false;
        std::vector<uint8_t> used(nw, 0);
        for (uint32_t i = 0; i < nw; ++i) {
            uint32_t wid = 0, nid = 0; f.read((char*)&wid, 4); f.read((char*)&nid, 4);
            if (!f || wid >= nw || nid >= n || inner[nid] || used[wid] || word[nid] != -1) return false;
            used[wid] = 1; word[nid] = (int32_t)wid;
        }
        return SetVocabulary(hd[0], hd[1], parent, desc, weight, word, hd[2], hd[3]);
    }
    // DBoW3 Vocabulary::load: the binary stream first, else the text layout
    bool LoadVocabulary(const char* path)
    {
        uint64_t sig = 0;
        { std::ifstream f(path, std::ios::binary); if (!f) return false; f.read((char*)&sig, 8); if (!f) sig = 0; }
        return sig == 88877711233ull ? LoadVocabularyBinary(path) : LoadVocabularyTxt(path);
    }
    bool HasVocabulary() const { return v_ != nullptr; }

    // src/FeatureMatching.cpp:192-208: mVoc_->transform(allDesp, bowVec, featVec, 4)
    void ComputeBoWVector(cv::Mat& desp, BowVector& bowVec, FeatureVector& featVec)
    {
        bowVec.clear(); featVec.clear();                              // :195-196
        if (!v_ || desp.rows <= 0 || desp.cols != 32) { status_ = v_ ? SSVIO_OK : SSVIO_ERR_INVALID; return; }
        std::vector<int32_t> word(desp.rows), node(desp.rows); std::vector<double> weight(desp.rows);
        status_ = ssvio_bow_transform(m_, v_, desp.ptr(0), desp.rows, (int)desp.step, 4, word.data(), node.data(), weight.data());
        if (status_ == SSVIO_OK) fill_vectors(word, node, weight, bowVec, featVec);
    }

    // include/FeatureMatching.h:27, src/FeatureMatching.cpp:58-125
    bool MatchByDBoW(cv::Mat& desp1, cv::Mat& desp2, std::vector<cv::DMatch>& matches)
    {
        if (!(desp1.rows > 0 && desp2.rows > 0)) return false;        // :61-63, before the clear
        matches.clear();                                              // :65
        if (!v_ || desp1.cols != 32 || desp2.cols != 32) { status_ = SSVIO_ERR_INVALID; return false; }
        matches.resize(desp1.rows);
        int n = 0;
        status_ = ssvio_match_dbow(m_, v_, desp1.ptr(0), desp1.rows, (int)desp1.step, desp2.ptr(0), desp2.rows, (int)desp2.step, 4,
                                   (ssvio_dmatch*)matches.data(), desp1.rows, &n);
        matches.resize(status_ == SSVIO_OK ? n : 0);
        if (status_ == SSVIO_OK) {                                    // the members :68-69 fill
            for (int side = 0; side < 2; ++side) {
                const int rows = side ? desp2.rows : desp1.rows; int got = 0;
                std::vector<int32_t> word(rows), node(rows); std::vector<double> weight(rows);
                if (ssvio_bow_last_transform(m_, side, word.data(), node.data(), weight.data(), rows, &got) != SSVIO_OK) break;
                fill_vectors(word, node, weight, side ? mBowVec2_ : mBowVec1_, side ? mFeatVec2_ : mFeatVec1_);
            }
        }
        return matches.size() > 0;                                    // :124
    }
    BowVector mBowVec1_, mBowVec2_;                                   // include/FeatureMatching.h:46-49 (private there)
    FeatureVector mFeatVec1_, mFeatVec2_;
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
    // the two maps of Vocabulary::transform(features, v, fv, levelsup) from the per-feature results
    void fill_vectors(const std::vector<int32_t>& word, const std::vector<int32_t>& node, const std::vector<double>& weight,
                      BowVector& bv, FeatureVector& fv)
    {
        bv.clear(); fv.clear();
        const bool tf = weighting_ == 0 || weighting_ == 1, must = scoring_ != 5;
        // Both maps are built from sorted runs with end hints (linear), not by a tree search per feature.  A stable sort keeps the
        // features of one word / node in ascending index, so a word's weights are summed in the order addWeight would see them.
        std::vector<uint32_t> order;
        order.reserve(word.size());
        for (size_t i = 0; i < word.size(); ++i) if (weight[i] > 0) order.push_back((uint32_t)i);     // stopped words enter neither map
        std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return (unsigned)word[a] < (unsigned)word[b]; });
        for (size_t p = 0; p < order.size();) {
            const unsigned w = (unsigned)word[order[p]];
            double v = weight[order[p]];                              // addIfNotExist keeps the first weight, addWeight sums
            size_t q = p + 1;
            for (; q < order.size() && (unsigned)word[order[q]] == w; ++q) if (tf) v += weight[order[q]];
            bv.emplace_hint(bv.end(), w, v);
            p = q;
        }
        std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return (unsigned)node[a] < (unsigned)node[b]; });
        for (size_t p = 0; p < order.size();) {
            const unsigned nd = (unsigned)node[order[p]];
            size_t q = p;
            while (q < order.size() && (unsigned)node[order[q]] == nd) ++q;
            std::vector<unsigned int>& list = fv.emplace_hint(fv.end(), nd, std::vector<unsigned int>())->second;
            list.assign(order.begin() + p, order.begin() + q);
            std::sort(list.begin(), list.end());                       // ascending feature index (the second sort started from word order)
            p = q;
        }
        if (tf && !bv.empty() && !must) { const double nd = (double)bv.size(); for (auto& e : bv) e.second /= nd; }
        if (must) {
            double norm = 0.0;
            if (scoring_ == 1) { for (auto& e : bv) norm += e.second * e.second; norm = std::sqrt(norm); }
            else for (auto& e : bv) norm += std::fabs(e.second);
            if (norm > 0.0) for (auto& e : bv) e.second /= norm;
        }
    }
    void init(int device)
    {
        device_ = device;
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
    ssvio_vocab* v_ = nullptr;
    int device_ = 0, scoring_ = 0, weighting_ = 0;
    int status_ = SSVIO_OK;
};
