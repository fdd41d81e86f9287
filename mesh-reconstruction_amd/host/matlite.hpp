// matlite.hpp -- the subset of cv::Mat the hot path's interface needs, for builds without OpenCV.
//
// recon.hpp of the reference is written against cv::Mat (recon.hpp:17-25).  OpenCV is not in this image and
// cannot be assumed on the GPU box, so the host mirror is written against this small dense, ref-counted
// array with the same member names and the same copy semantics (copying a Mat shares pixels, like
// cv::Mat; clone() copies).  A maintainer linking the original recon.cpp uses host/render_hip_cv.cpp instead, the same
// calls written against the real cv::Mat.
#pragma once

#include <cassert>
#include <cstdint>
#include <cstring>
#include <memory>
#include <vector>

namespace mvs {

enum MatType { U8C1 = 0, U8C3 = 1, F32C1 = 2, F32C2 = 3, F32C4 = 4, S32C1 = 5, F32C3 = 6 };

inline int type_channels(int t)
{
    switch (t) {
    case U8C3: case F32C3: return 3;
    case F32C2: return 2;
    case F32C4: return 4;
    default: return 1;
    }
}
inline int type_elem1(int t) { return (t == U8C1 || t == U8C3) ? 1 : 4; }

class Mat {
public:
    int rows = 0, cols = 0;
    Mat() {}
    Mat(int r, int c, int type) { create(r, c, type); }
    void create(int r, int c, int type)
    {
        rows = r;
        cols = c;
        type_ = type;
        buf_ = std::make_shared<std::vector<uint8_t>>((size_t)r * c * elemSize());
        data = buf_->data();
    }
    static Mat zeros(int r, int c, int type)
    {
        Mat m(r, c, type);
        if (m.data) std::memset(m.data, 0, m.total() * m.elemSize());
        return m;
    }
    static Mat eye4()
    {
        Mat m = zeros(4, 4, F32C1);
        for (int i = 0; i < 4; i++) m.at<float>(i, i) = 1.f;
        return m;
    }
    int type() const { return type_; }
    int channels() const { return type_channels(type_); }
    size_t elemSize() const { return (size_t)type_channels(type_) * type_elem1(type_); }
    size_t total() const { return (size_t)rows * cols; }
    bool empty() const { return !data || rows == 0 || cols == 0; }
    bool isContinuous() const { return true; }
    Mat clone() const
    {
        Mat m(rows, cols, type_);
        if (data) std::memcpy(m.data, data, total() * elemSize());
        return m;
    }
    template <class T> T *ptr(int r = 0) { return reinterpret_cast<T *>(data + (size_t)r * cols * elemSize()); }
    template <class T> const T *ptr(int r = 0) const { return reinterpret_cast<const T *>(data + (size_t)r * cols * elemSize()); }
    template <class T> T &at(int r, int c = 0) { return ptr<T>(r)[c]; }
    template <class T> const T &at(int r, int c = 0) const { return ptr<T>(r)[c]; }
    // a new header on a copy of rows [r0, r1)
    Mat rowRange(int r0, int r1) const
    {
        Mat m(r1 - r0, cols, type_);
        if (r1 > r0) std::memcpy(m.data, data + (size_t)r0 * cols * elemSize(), (size_t)(r1 - r0) * cols * elemSize());
        return m;
    }
    void push_back(const Mat &o)
    {
        if (o.empty()) return;
        if (empty()) {
            *this = o.clone();
            return;
        }
        assert(o.cols == cols && o.type_ == type_);
        Mat m(rows + o.rows, cols, type_);
        std::memcpy(m.data, data, total() * elemSize());
        std::memcpy(m.data + total() * elemSize(), o.data, o.total() * o.elemSize());
        *this = m;
    }
    uint8_t *data = nullptr;

private:
    int type_ = U8C1;
    std::shared_ptr<std::vector<uint8_t>> buf_;
};

struct Size {
    int width = 0, height = 0;
    Size() {}
    Size(int w, int h) : width(w), height(h) {}
};

// 4x4 * 4x4 and 4x4 * 4x1 products in f32 (cv::Mat operator* on CV_32F accumulates in double via gemm;
// the heuristics are tolerance-level policy code, f64 accumulation is used here as well)
inline Mat matmul(const Mat &a, const Mat &b)
{
    assert(a.cols == b.rows);
    Mat c(a.rows, b.cols, F32C1);
    for (int i = 0; i < a.rows; i++)
        for (int j = 0; j < b.cols; j++) {
            double s = 0;
            for (int k = 0; k < a.cols; k++) s += (double)a.at<float>(i, k) * (double)b.at<float>(k, j);
            c.at<float>(i, j) = (float)s;
        }
    return c;
}

}  // namespace mvs
