// cv_compat.h -- the few cv:: types the adapters touch.
// With -DSSVIO_HAVE_OPENCV the real OpenCV headers are used and SSVIO's other sources compile
// unchanged against the adapters.  Without OpenCV (this build image has none) a minimal set of
// layout-compatible PODs (KeyPoint 28 B, DMatch 16 B, Point2d/3d, a ref-counted 8-bit / 16U / 64F Mat) keeps the adapters
// compilable and testable.  This is NOT an OpenCV replacement and is not used to build the reference.
#pragma once
#ifdef SSVIO_HAVE_OPENCV
#include <opencv2/core/core.hpp>
#include <opencv2/features2d/features2d.hpp>
#else
#include <cstdint>
#include <cstring>
#include <memory>
#include <vector>
namespace cv {
enum { CV_8UC1_ = 0, CV_8UC3_ = 16 };
#ifndef CV_8U
#define CV_8U 0
#define CV_8UC1 0
#define CV_8UC3 16
#define CV_64F 6
#define CV_16U 2
#define CV_16UC1 2
#endif
struct Point2f { float x, y; Point2f(float x_ = 0, float y_ = 0) : x(x_), y(y_) {} };
struct Point2d { double x, y; Point2d(double x_ = 0, double y_ = 0) : x(x_), y(y_) {} };
struct Point3d { double x, y, z; Point3d(double x_ = 0, double y_ = 0, double z_ = 0) : x(x_), y(y_), z(z_) {} };
struct KeyPoint {
    Point2f pt; float size, angle, response; int octave, class_id;
    KeyPoint() : pt(0, 0), size(0), angle(-1), response(0), octave(0), class_id(-1) {}
};
struct DMatch {
    int queryIdx, trainIdx, imgIdx; float distance;
    DMatch() : queryIdx(-1), trainIdx(-1), imgIdx(-1), distance(3.402823466e+38f) {}
};
static_assert(sizeof(KeyPoint) == 28 && sizeof(DMatch) == 16, "layout must equal OpenCV's");
class Mat {
public:
    int rows = 0, cols = 0; size_t step = 0; unsigned char* data = nullptr;
    Mat() {}
    Mat(int r, int c, int type) { create(r, c, type); }
    Mat(int r, int c, int type, void* ext, size_t step_) : rows(r), cols(c), step(step_), data((unsigned char*)ext), type_(type) {}
    void create(int r, int c, int type) {
        if (r == rows && c == cols && type == type_ && data && buf_) return;
        rows = r; cols = c; type_ = type; step = (size_t)c * elemSize();
        buf_.reset(new unsigned char[(size_t)(r > 0 ? r : 0) * step + 1], std::default_delete<unsigned char[]>());
        data = buf_.get();
    }
    void release() { rows = cols = 0; step = 0; data = nullptr; buf_.reset(); }
    bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
    int type() const { return type_; }
    int channels() const { return type_ == CV_8UC3 ? 3 : 1; }
    size_t elemSize() const { return type_ == CV_64F ? 8 : type_ == CV_16U ? 2 : (size_t)channels(); }
    bool isContinuous() const { return step == (size_t)cols * elemSize(); }
    template <class T> T& at(int r, int c = 0) { return *(T*)(data + (size_t)r * step + (size_t)c * sizeof(T)); }
    unsigned char* ptr(int r = 0) { return data + (size_t)r * step; }
    const unsigned char* ptr(int r = 0) const { return data + (size_t)r * step; }
    Mat getMat() const { return *this; }
private:
    int type_ = CV_8UC1;
    std::shared_ptr<unsigned char> buf_;
};
typedef const Mat& InputArray;
typedef Mat& OutputArray;
}  // namespace cv
#endif
