/*
 * triangulate_oracle.c -- CPU restatement of triangulatePixels() / triangulatePixel() / goodSample() /
 * sampleImage<T>() / imageGradient() (util.cpp:44-329, 435-479): the consumer of depth + flows (recon.cpp:114).
 * TEST INFRASTRUCTURE ONLY (see mvs_oracle.h).  PARITY UNPINNED (OpenCV's gemm / invert / Sobel / PCA / eigen are
 * not in the reference tree; their arithmetic is restated as documented below).
 *
 * Followed literally, quirks included (SURVEY.md Appendix A-2, A-14 ... A-17):
 *   - pixel -> NDC without the half pixel: x = (col - W/2) 2/W, y = (H/2 - row) 2/H          util.cpp:185-188
 *   - the flow's y component is ADDED to an upward-pointing y: y + fly*scaleY                  util.cpp:209
 *   - sampleImage<T> multiplies the LEFT texel by frac(x) and the RIGHT one by 1 - frac(x) (weights swapped), rows
 *     likewise; its integer fast paths are unreachable                                        util.cpp:446-458
 *   - the CV_32FC2 Sobel gradient is read as cv::Point (two int32), interpolated in INTEGER arithmetic
 *     (each product rounded to int), and the int bits are stored back into a float matrix      util.cpp:213-217
 *   - Newton on the main-camera NDC depth: <= 50 steps, stop at |dz| < 1e-7; the Jacobian omits the
 *     -p dw/dz term                                                                           util.cpp:86-126
 *   - pdf = 0.159 * prod det(icov_i) * exp(exponent / 2)                                     util.cpp:141
 *   - a pixel is dropped if any side camera sees its measured point at NDC z < -1            util.cpp:229-233
 *   - normals: PCA of the valid points in the 21x21 pixel neighbourhood, smallest eigenvector, flipped by the sign of
 *     sum_i 1/(n . (c_i - point)); `float dot` is uninitialised in the reference (util.cpp:302) -- 0 here;
 *     fewer than 3 neighbours: sum_i v_i/|v_i|^2 with v_i = c_i - point.xyz (NOT dehomogenised, util.cpp:319);
 *     result scaled by pdf^(1/n_side) / |n|                                                   util.cpp:278-279, 324
 * Arithmetic where OpenCV is silent: every cv::Mat product accumulates in double and is stored as f32 (one rounding
 * per materialised Mat); cv::Mat::inv() of the 4x4 camera is the double cofactor inverse rounded to f32; 2x2 inverse
 * and determinant in double; PCA mean/covariance/Jacobi in double.
 */
#include "mvs_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ---- small f32 matrix helpers with double accumulation -------------------------------------- */
static void mat44_mul(const float *a, const float *b, float *c)
{
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            double s = 0;
            for (int k = 0; k < 4; k++) s += (double)a[4 * i + k] * (double)b[4 * k + j];
            c[4 * i + j] = (float)s;
        }
}
static void mat44_vec(const float *a, const float *v, float *o)
{
    for (int i = 0; i < 4; i++) {
        double s = 0;
        for (int k = 0; k < 4; k++) s += (double)a[4 * i + k] * (double)v[k];
        o[i] = (float)s;
    }
}

void orc_inv44_f32(const float *m, float *out)
{
    double a[16], c[16];
    for (int i = 0; i < 16; i++) a[i] = m[i];
    c[0] = a[5] * a[10] * a[15] - a[5] * a[11] * a[14] - a[9] * a[6] * a[15] + a[9] * a[7] * a[14] + a[13] * a[6] * a[11] - a[13] * a[7] * a[10];
    c[4] = -a[4] * a[10] * a[15] + a[4] * a[11] * a[14] + a[8] * a[6] * a[15] - a[8] * a[7] * a[14] - a[12] * a[6] * a[11] + a[12] * a[7] * a[10];
    c[8] = a[4] * a[9] * a[15] - a[4] * a[11] * a[13] - a[8] * a[5] * a[15] + a[8] * a[7] * a[13] + a[12] * a[5] * a[11] - a[12] * a[7] * a[9];
    c[12] = -a[4] * a[9] * a[14] + a[4] * a[10] * a[13] + a[8] * a[5] * a[14] - a[8] * a[6] * a[13] - a[12] * a[5] * a[10] + a[12] * a[6] * a[9];
    c[1] = -a[1] * a[10] * a[15] + a[1] * a[11] * a[14] + a[9] * a[2] * a[15] - a[9] * a[3] * a[14] - a[13] * a[2] * a[11] + a[13] * a[3] * a[10];
    c[5] = a[0] * a[10] * a[15] - a[0] * a[11] * a[14] - a[8] * a[2] * a[15] + a[8] * a[3] * a[14] + a[12] * a[2] * a[11] - a[12] * a[3] * a[10];
    c[9] = -a[0] * a[9] * a[15] + a[0] * a[11] * a[13] + a[8] * a[1] * a[15] - a[8] * a[3] * a[13] - a[12] * a[1] * a[11] + a[12] * a[3] * a[9];
    c[13] = a[0] * a[9] * a[14] - a[0] * a[10] * a[13] - a[8] * a[1] * a[14] + a[8] * a[2] * a[13] + a[12] * a[1] * a[10] - a[12] * a[2] * a[9];
    c[2] = a[1] * a[6] * a[15] - a[1] * a[7] * a[14] - a[5] * a[2] * a[15] + a[5] * a[3] * a[14] + a[13] * a[2] * a[7] - a[13] * a[3] * a[6];
    c[6] = -a[0] * a[6] * a[15] + a[0] * a[7] * a[14] + a[4] * a[2] * a[15] - a[4] * a[3] * a[14] - a[12] * a[2] * a[7] + a[12] * a[3] * a[6];
    c[10] = a[0] * a[5] * a[15] - a[0] * a[7] * a[13] - a[4] * a[1] * a[15] + a[4] * a[3] * a[13] + a[12] * a[1] * a[7] - a[12] * a[3] * a[5];
    c[14] = -a[0] * a[5] * a[14] + a[0] * a[6] * a[13] + a[4] * a[1] * a[14] - a[4] * a[2] * a[13] - a[12] * a[1] * a[6] + a[12] * a[2] * a[5];
    c[3] = -a[1] * a[6] * a[11] + a[1] * a[7] * a[10] + a[5] * a[2] * a[11] - a[5] * a[3] * a[10] - a[9] * a[2] * a[7] + a[9] * a[3] * a[6];
    c[7] = a[0] * a[6] * a[11] - a[0] * a[7] * a[10] - a[4] * a[2] * a[11] + a[4] * a[3] * a[10] + a[8] * a[2] * a[7] - a[8] * a[3] * a[6];
    c[11] = -a[0] * a[5] * a[11] + a[0] * a[7] * a[9] + a[4] * a[1] * a[11] - a[4] * a[3] * a[9] - a[8] * a[1] * a[7] + a[8] * a[3] * a[5];
    c[15] = a[0] * a[5] * a[10] - a[0] * a[6] * a[9] - a[4] * a[1] * a[10] + a[4] * a[2] * a[9] + a[8] * a[1] * a[6] - a[8] * a[2] * a[5];
    const double det = a[0] * c[0] + a[1] * c[4] + a[2] * c[8] + a[3] * c[12];
    const double rd = 1.0 / det;
    for (int i = 0; i < 16; i++) out[i] = (float)(c[i] * rd);
}

/* util.cpp:33-41, as in host/render_hip.cpp: null vector of rows 0,1,3 */
static void camera_center(const float *cam, float *c3)
{
    const int rows[3] = {0, 1, 3};
    double p[3][4];
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 4; c++) p[r][c] = cam[4 * rows[r] + c];
#define DET3(c0, c1, c2)                                                                                          \
    (p[0][c0] * (p[1][c1] * p[2][c2] - p[1][c2] * p[2][c1]) - p[0][c1] * (p[1][c0] * p[2][c2] - p[1][c2] * p[2][c0]) + \
     p[0][c2] * (p[1][c0] * p[2][c1] - p[1][c1] * p[2][c0]))
    const double c[4] = {DET3(1, 2, 3), -DET3(0, 2, 3), DET3(0, 1, 3), -DET3(0, 1, 2)};
#undef DET3
    const double n = sqrt(c[0] * c[0] + c[1] * c[1] + c[2] * c[2] + c[3] * c[3]);
    float t[4];
    for (int i = 0; i < 4; i++) t[i] = (float)(n > 0 ? c[i] / n : c[i]);
    for (int i = 0; i < 3; i++) c3[i] = t[i] / t[3]; /* util.cpp:272 */
}

/* imageGradient (util.cpp:465-479): Sobel 3x3, CV_32F, BORDER_REFLECT_101 */
static inline int refl(int p, int n)
{
    if (n == 1) return 0;
    while (p < 0 || p >= n) {
        if (p < 0) p = -p;
        if (p >= n) p = 2 * n - 2 - p;
    }
    return p;
}
void orc_image_gradient(const float *img, int W, int H, float *grad2)
{
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            float v[3][3];
            for (int j = 0; j < 3; j++)
                for (int i = 0; i < 3; i++) v[j][i] = img[(size_t)refl(y + j - 1, H) * W + refl(x + i - 1, W)];
            /* separable: smooth [1 2 1] across, difference [-1 0 1] along; row filter first, then column filter */
            float rdx[3], rsm[3];
            for (int j = 0; j < 3; j++) {
                rdx[j] = v[j][2] - v[j][0];
                rsm[j] = v[j][0] + v[j][1] * 2.0f + v[j][2];
            }
            grad2[((size_t)y * W + x) * 2] = rdx[0] + rdx[1] * 2.0f + rdx[2];
            grad2[((size_t)y * W + x) * 2 + 1] = rsm[2] - rsm[0];
        }
}

/* util.cpp:44-53 */
static int good_sample(const float *depth, int W, int H, float x, float y)
{
    const int ix = (int)x, iy = (int)y;
    if (ix <= 0 || ix >= W - 1 || iy <= 0 || iy >= H - 1) return 0;
    return depth[(size_t)iy * W + ix] != ORC_BACKGROUND_DEPTH && depth[(size_t)iy * W + ix + 1] != ORC_BACKGROUND_DEPTH &&
           depth[(size_t)(iy + 1) * W + ix] != ORC_BACKGROUND_DEPTH && depth[(size_t)(iy + 1) * W + ix + 1] != ORC_BACKGROUND_DEPTH;
}

/* util.cpp:438-461 with T = float (only called where goodSample held, so never out of range) */
static float sample_f32(const float *img, int W, float x, float y)
{
    const float lw = fmodf(x, 1), rw = 1 - lw, tw = fmodf(y, 1), bw = 1 - tw;
    const int ix = (int)x, iy = (int)y;
    const float *p = img + (size_t)iy * W + ix;
    return (p[0] * lw + p[1] * rw) * tw + (p[W] * lw + p[W + 1] * rw) * bw;
}

/* util.cpp:438-461 with T = cv::Point on a CV_32FC2 matrix: the float bits are int32, each product is rounded to
 * int (cv::Point_<int> * float = saturate_cast<int>(v * w)), sums are int; the result's bits are stored as float */
static int clampi_round(float v)
{
    if (!(v > -2147483648.0f)) return (int)0x80000000;
    if (!(v < 2147483648.0f)) return 0x7fffffff;
    return (int)lrintf(v);
}
static void sample_point_punned(const float *grad2, int W, int H, float x, float y, float out[2])
{
    const float lw = fmodf(x, 1), rw = 1 - lw, tw = fmodf(y, 1), bw = 1 - tw;
    int ix = (int)x, iy = (int)y;
    /* the fallback call samples at integer (col,row) of the pixel itself, possibly on the last row/column:
     * clamp the +1 taps (the reference would read past the row there) */
    const int ix1 = ix + 1 < W ? ix + 1 : ix, iy1 = iy + 1 < H ? iy + 1 : iy;
    for (int c = 0; c < 2; c++) {
        int32_t b00, b01, b10, b11;
        memcpy(&b00, &grad2[((size_t)iy * W + ix) * 2 + c], 4);
        memcpy(&b01, &grad2[((size_t)iy * W + ix1) * 2 + c], 4);
        memcpy(&b10, &grad2[((size_t)iy1 * W + ix) * 2 + c], 4);
        memcpy(&b11, &grad2[((size_t)iy1 * W + ix1) * 2 + c], 4);
        /* (P00*lw + P01*rw)*tw + (P10*lw + P11*rw)*bw with every Point*float rounded to int */
        const int top = (int)((uint32_t)clampi_round((float)b00 * lw) + (uint32_t)clampi_round((float)b01 * rw));
        const int bot = (int)((uint32_t)clampi_round((float)b10 * lw) + (uint32_t)clampi_round((float)b11 * rw));
        const int r = (int)((uint32_t)clampi_round((float)top * tw) + (uint32_t)clampi_round((float)bot * bw));
        memcpy(&out[c], &r, 4);
    }
}

/* symmetric 3x3 eigen decomposition by cyclic Jacobi (double); returns the eigenvector of the smallest eigenvalue */
static void smallest_eigvec3(double a[3][3], double v[3])
{
    double V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int sweep = 0; sweep < 32; sweep++) {
        const double off = fabs(a[0][1]) + fabs(a[0][2]) + fabs(a[1][2]);
        if (off < 1e-300) break;
        for (int p = 0; p < 2; p++)
            for (int q = p + 1; q < 3; q++) {
                if (a[p][q] == 0.0) continue;
                const double theta = (a[q][q] - a[p][p]) / (2.0 * a[p][q]);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < 3; k++) {
                    const double akp = a[k][p], akq = a[k][q];
                    a[k][p] = c * akp - s * akq;
                    a[k][q] = s * akp + c * akq;
                }
                for (int k = 0; k < 3; k++) {
                    const double apk = a[p][k], aqk = a[q][k];
                    a[p][k] = c * apk - s * aqk;
                    a[q][k] = s * apk + c * aqk;
                }
                for (int k = 0; k < 3; k++) {
                    const double vkp = V[k][p], vkq = V[k][q];
                    V[k][p] = c * vkp - s * vkq;
                    V[k][q] = s * vkp + c * vkq;
                }
            }
    }
    int m = 0;
    if (a[1][1] < a[m][m]) m = 1;
    if (a[2][2] < a[m][m]) m = 2;
    for (int k = 0; k < 3; k++) v[k] = V[k][m];
}

#define MAXCAM 64

typedef struct {
    float CM[16];   /* camera * mainCameraInv                          util.cpp:97,208 */
    float B[6];     /* camera.rows(0,2).cols(0,3) * mainCameraInv.rows(0,3).cols(0,3)   util.cpp:219 */
    float projDeriv[2]; /* camera.rows(0,2) * mainCameraInv.col(2)     util.cpp:84 */
    float projW[4];     /* camera.row(3) * mainCameraInv               util.cpp:85,87 */
} CamPre;

/*
 * Dense form: for every pixel writes valid (0/1), point (4 floats), pdf and, in the second stage, the scaled normal.
 * orc_triangulate_pixels compacts in scan order to the reference's N x 7 layout (x, y, z, w, nx, ny, nz).
 */
int orc_triangulate_pixels(const float *const *flows /* V x (H*W*4) */, const float main_cam[16], const float *side_cams,
                           int V, const float *depth, int W, int H, float *out_points7 /* up to H*W*7 */)
{
    if (V > MAXCAM) return -1;
    const size_t P = (size_t)W * H;
    float Minv[16];
    orc_inv44_f32(main_cam, Minv);
    CamPre *pre = (CamPre *)malloc(sizeof(CamPre) * (size_t)(V > 0 ? V : 1));
    for (int i = 0; i < V; i++) {
        const float *C = side_cams + 16 * i;
        mat44_mul(C, Minv, pre[i].CM);
        for (int r = 0; r < 2; r++)
            for (int c = 0; c < 3; c++) {
                double s = 0;
                for (int k = 0; k < 3; k++) s += (double)C[4 * r + k] * (double)Minv[4 * k + c];
                pre[i].B[3 * r + c] = (float)s;
            }
        for (int r = 0; r < 2; r++) {
            double s = 0;
            for (int k = 0; k < 4; k++) s += (double)C[4 * r + k] * (double)Minv[4 * k + 2];
            pre[i].projDeriv[r] = (float)s;
        }
        for (int c = 0; c < 4; c++) {
            double s = 0;
            for (int k = 0; k < 4; k++) s += (double)C[12 + k] * (double)Minv[4 * k + c];
            pre[i].projW[c] = (float)s;
        }
    }
    float *grad = (float *)malloc(sizeof(float) * P * 2);
    orc_image_gradient(depth, W, H, grad);
    uint8_t *valid = (uint8_t *)calloc(P, 1);
    float *pts = (float *)malloc(sizeof(float) * P * 4), *pdfs = (float *)malloc(sizeof(float) * P);

    const float centerX = W / 2.0f, centerY = H / 2.0f, scaleX = 2.0f / W, scaleY = 2.0f / H;
    for (int row = 0; row < H; row++)
        for (int col = 0; col < W; col++) {
            const size_t pix = (size_t)row * W + col;
            const float d0 = depth[pix];
            if (d0 == ORC_BACKGROUND_DEPTH) continue;
            const float x = (col - centerX) * scaleX, y = (centerY - row) * scaleY;
            float meas[MAXCAM][2], icov[MAXCAM][4];
            int okay = 1;
            for (int i = 0; i < V && okay; i++) {
                const float *fl = flows[i] + pix * 4;
                const float flx = fl[0], fly = fl[1], variance = fl[2];
                const int gs = good_sample(depth, W, H, col + flx, row + fly);
                const float z = gs ? sample_f32(depth, W, col + flx, row + fly) : d0;
                const float v4[4] = {x + flx * scaleX, y + fly * scaleY, z, 1.0f};
                float mp[4];
                mat44_vec(pre[i].CM, v4, mp);
                float g[2];
                if (gs)
                    sample_point_punned(grad, W, H, col + flx, row + fly, g);
                else
                    sample_point_punned(grad, W, H, (float)col, (float)row, g);
                /* A = B * D, D = [[1,0],[0,1],[gx,gy]]; A /= mp.w */
                float A[4];
                for (int r = 0; r < 2; r++)
                    for (int c = 0; c < 2; c++) {
                        const double s = (double)pre[i].B[3 * r + 0] * (c == 0 ? 1.0 : 0.0) + (double)pre[i].B[3 * r + 1] * (c == 1 ? 1.0 : 0.0) +
                                         (double)pre[i].B[3 * r + 2] * (double)g[c];
                        A[2 * r + c] = (float)s / mp[3];
                    }
                /* icovar = (A A^T)^-1 / variance */
                float S[4];
                for (int r = 0; r < 2; r++)
                    for (int c = 0; c < 2; c++) S[2 * r + c] = (float)((double)A[2 * r] * A[2 * c] + (double)A[2 * r + 1] * A[2 * c + 1]);
                const double det = (double)S[0] * S[3] - (double)S[1] * S[2];
                float inv[4];
                if (det != 0.0) {
                    const double rd = 1.0 / det;
                    inv[0] = (float)(S[3] * rd);
                    inv[1] = (float)(-S[1] * rd);
                    inv[2] = (float)(-S[2] * rd);
                    inv[3] = (float)(S[0] * rd);
                } else {
                    inv[0] = inv[1] = inv[2] = inv[3] = 0.f; /* cv::Mat::inv() of a singular matrix is all zeros */
                }
                for (int k = 0; k < 4; k++) icov[i][k] = inv[k] / variance;
                const float mz = mp[2] / mp[3];
                if (mz < -1) {
                    okay = 0;
                    break;
                }
                meas[i][0] = mp[0] / mp[3];
                meas[i][1] = mp[1] / mp[3];
            }
            if (!okay) continue;
            /* triangulatePixel, util.cpp:62-164 */
            float k[4] = {x, y, d0, 1.0f};
            float pdf = 1.0f;
            for (int iter = 0;; iter++) {
                double firstDz = 0, secondDz = 0;
                float diff[MAXCAM][2];
                for (int i = 0; i < V; i++) {
                    float ep[4];
                    mat44_vec(pre[i].CM, k, ep);
                    const float px = ep[0] / ep[3], py = ep[1] / ep[3];
                    double w = 0;
                    for (int c = 0; c < 4; c++) w += (double)pre[i].projW[c] * (double)k[c];
                    const float pw = (float)w;
                    const float dpx = pre[i].projDeriv[0] / pw, dpy = pre[i].projDeriv[1] / pw;
                    diff[i][0] = px - meas[i][0];
                    diff[i][1] = py - meas[i][1];
                    const float t0 = (float)((double)icov[i][0] * dpx + (double)icov[i][1] * dpy);
                    const float t1 = (float)((double)icov[i][2] * dpx + (double)icov[i][3] * dpy);
                    firstDz += (double)diff[i][0] * t0 + (double)diff[i][1] * t1;
                    secondDz += (double)dpx * t0 + (double)dpy * t1;
                }
                const double delta_z = -firstDz / secondDz, eps = 1e-7;
                if (iter >= 50 || (delta_z < eps && delta_z > -eps)) {
                    double exponent = 0, product_ivar = 1;
                    for (int i = 0; i < V; i++) {
                        const float t0 = (float)((double)icov[i][0] * diff[i][0] + (double)icov[i][1] * diff[i][1]);
                        const float t1 = (float)((double)icov[i][2] * diff[i][0] + (double)icov[i][3] * diff[i][1]);
                        exponent -= (double)diff[i][0] * t0 + (double)diff[i][1] * t1;
                        product_ivar *= (double)icov[i][0] * icov[i][3] - (double)icov[i][1] * icov[i][2];
                    }
                    pdf = (float)(0.159 * product_ivar * exp(0.5 * exponent));
                    break;
                }
                k[2] = (float)((double)k[2] + delta_z);
            }
            mat44_vec(Minv, k, pts + pix * 4);
            pdfs[pix] = pdf;
            valid[pix] = 1;
        }

    /* normals, util.cpp:262-325 */
    float centers[MAXCAM + 1][3];
    camera_center(main_cam, centers[0]);
    for (int i = 0; i < V; i++) camera_center(side_cams + 16 * i, centers[i + 1]);
    const int radius = 10;
    int count = 0;
    for (int row = 0; row < H; row++)
        for (int col = 0; col < W; col++) {
            const size_t pix = (size_t)row * W + col;
            if (!valid[pix]) continue;
            float pdf = pdfs[pix];
            if (V > 1) pdf = (float)pow((double)pdf, 1.0 / V);
            int n = 0;
            double mean[3] = {0, 0, 0};
            for (int ny = row - radius; ny <= row + radius; ny++) {
                if (ny < 0 || ny >= H) continue;
                for (int nx = col - radius; nx <= col + radius; nx++) {
                    if (nx < 0 || nx >= W || !valid[(size_t)ny * W + nx]) continue;
                    const float *q = pts + ((size_t)ny * W + nx) * 4;
                    for (int c = 0; c < 3; c++) mean[c] += (double)(q[c] / q[3]);
                    n++;
                }
            }
            float normal[3];
            const float *pp = pts + pix * 4;
            if (n >= 3) {
                for (int c = 0; c < 3; c++) mean[c] /= n;
                double cov[3][3] = {{0}};
                for (int ny = row - radius; ny <= row + radius; ny++) {
                    if (ny < 0 || ny >= H) continue;
                    for (int nx = col - radius; nx <= col + radius; nx++) {
                        if (nx < 0 || nx >= W || !valid[(size_t)ny * W + nx]) continue;
                        const float *q = pts + ((size_t)ny * W + nx) * 4;
                        double d[3];
                        for (int c = 0; c < 3; c++) d[c] = (double)(q[c] / q[3]) - mean[c];
                        for (int a = 0; a < 3; a++)
                            for (int b = a; b < 3; b++) cov[a][b] += d[a] * d[b];
                    }
                }
                for (int a = 0; a < 3; a++)
                    for (int b = a; b < 3; b++) {
                        cov[a][b] /= n;
                        cov[b][a] = cov[a][b];
                    }
                double ev[3];
                smallest_eigvec3(cov, ev);
                for (int c = 0; c < 3; c++) normal[c] = (float)ev[c];
                float dot = 0.f; /* uninitialised in the reference (util.cpp:302) */
                for (int i = 0; i <= V; i++) {
                    double s = 0;
                    for (int c = 0; c < 3; c++) s += (double)normal[c] * (double)(centers[i][c] - pp[c] / pp[3]);
                    dot += (float)(1.0 / s);
                }
                if (dot < 0)
                    for (int c = 0; c < 3; c++) normal[c] = -normal[c];
            } else {
                normal[0] = normal[1] = normal[2] = 0.f;
                for (int i = 0; i <= V; i++) {
                    float vec[3];
                    double vv = 0;
                    for (int c = 0; c < 3; c++) {
                        vec[c] = centers[i][c] - pp[c]; /* not dehomogenised: util.cpp:319 */
                        vv += (double)vec[c] * vec[c];
                    }
                    for (int c = 0; c < 3; c++) normal[c] += (float)(vec[c] / vv);
                }
            }
            const double nn = sqrt((double)normal[0] * normal[0] + (double)normal[1] * normal[1] + (double)normal[2] * normal[2]);
            float *o = out_points7 + (size_t)count * 7;
            memcpy(o, pp, sizeof(float) * 4);
            for (int c = 0; c < 3; c++) o[4 + c] = (float)((double)normal[c] * pdf / nn);
            count++;
        }
    free(pdfs);
    free(pts);
    free(valid);
    free(grad);
    free(pre);
    return count;
}
