// DECLARATIONS ONLY -- not OpenCV.  The few cv:: names the reference's recon.hpp (recon.hpp:7,17-25,117) and
// mesh-reconstruction_amd/host/render_hip_cv.cpp use, so that the seam file can be COMPILED (never linked or run) in an image
// without OpenCV: a boundary check that the file matches the reference interface, not parity evidence (tests/test_host_cpu.py).
#pragma once
#include <cassert>  // (OpenCV's own core.hpp brings it in: render_glx.cpp relies on that)
#include <cstddef>

#define CV_8U 0
#define CV_32S 4
#define CV_32F 5
#define CV_MAKETYPE(depth, cn) ((depth) + (((cn)-1) << 3))
#define CV_8UC1 CV_MAKETYPE(CV_8U, 1)
#define CV_8UC3 CV_MAKETYPE(CV_8U, 3)
#define CV_32SC1 CV_MAKETYPE(CV_32S, 1)
#define CV_32FC1 CV_MAKETYPE(CV_32F, 1)
#define CV_32FC4 CV_MAKETYPE(CV_32F, 4)

namespace cv {
struct Size {
    int width, height;
    Size();
    Size(int w, int h);
};
class Mat {
public:
    Mat();
    Mat(int rows, int cols, int type);
    Mat(int rows, int cols, int type, void *data);
    Mat(const Mat &);
    ~Mat();
    Mat &operator=(const Mat &);
    Mat clone() const;
    Mat rowRange(int a, int b) const;
    Mat colRange(int a, int b) const;
    bool isContinuous() const;
    bool empty() const;
    int type() const;
    int depth() const;
    size_t elemSize() const;
    int channels() const;
    size_t total() const;
    template <class T> T *ptr(int row = 0);
    template <class T> const T *ptr(int row = 0) const;
    template <class T> T &at(int r, int c);
    template <class T> const T &at(int r, int c) const;
    int rows, cols;
    unsigned char *data;
};
void flip(const Mat &src, Mat &dst, int flipCode);
// (cv::MatExpr in OpenCV; declarations enough for `2 * m - 1`, render_glx.cpp:395)
Mat operator*(double a, const Mat &m);
Mat operator-(const Mat &m, double b);
}  // namespace cv
