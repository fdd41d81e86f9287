/*
 * flow_oracle.c -- CPU restatement of calculateFlow() (flow.cpp:19-42): dense optical flow + variance channel.
 * TEST INFRASTRUCTURE ONLY (see mvs_oracle.h).  PARITY UNPINNED.
 *
 * The reference delegates the arithmetic to OpenCV: cv::FarnebackOpticalFlow (video module, flow.cpp:24-26) or
 * cv::optflow::createVariationalFlowRefinement (opencv_contrib, flow.cpp:29).  Neither is in /root/reference nor in
 * this image and no version is pinned (Makefile:10-13), so both are restated from their published algorithms as
 * OpenCV 3.x structures them:
 *
 * Farneback ("Two-Frame Motion Estimation Based on Polynomial Expansion", SCIA 2003), OpenCV layout:
 *   levels: k = L..0 with scale 0.8^k (L = 10 unless a level would be < 32 px); per level both frames are converted
 *   to f32, Gaussian-blurred (sigma = (1/scale - 1)/2, ksize = max(round(5 sigma)|1, 3), REFLECT_101; sigma = 0 with
 *   ksize 3 is the fixed [1 2 1]/4 kernel), bilinearly resized to round(W scale) x round(H scale), expanded into
 *   the 5 polynomial coefficients per pixel (Gaussian applicability of half-width poly_n, sigma poly_sigma; replicate
 *   border); the flow of the coarser level is bilinearly resized and multiplied by 1/0.8; then `iterations` rounds of
 *   [box-blur the 5 normal-equation terms over winsize, solve the 2x2 system, rebuild the terms from the new flow].
 *   The box blur is a separable double-precision sum over 2*(winsize/2)+1 replicated taps scaled by 1/winsize^2,
 *   summed top-to-bottom then left-to-right (OpenCV keeps running sums; the order here is fixed so the HIP kernels
 *   can reproduce it bit for bit).
 * Parameters are the reference's: levels 10, pyr_scale 0.8, winsize (H+W)/100, iterations 7, poly_n 5 or 7,
 * poly_sigma (H+W)/1000, flags 0 (box window, zero initial flow).
 *
 * Threads: the row loops below carry `#pragma omp parallel for`.  Every output value is computed by ONE thread from
 * inputs no thread of the same loop writes, with the operations in the order written -- the results do not depend on
 * the thread count (tests/test_flow_cpu.py checks 1 thread against many); it only makes 1080p / 4K parity runs minutes
 * instead of tens of minutes.
 *
 * Variational refinement (Brox et al., ECCV 2004, as organised in OpenCV's VariationalRefinement): see
 * orc_variational_refine below.  The reference passes an UNINITIALISED flow matrix into calc() (flow.cpp:31-32);
 * the restatement starts from zero flow.
 */
#include "mvs_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

static inline int refl101(int p, int n)
{
    if (n == 1) return 0;
    while (p < 0 || p >= n) {
        if (p < 0) p = -p;
        if (p >= n) p = 2 * n - 2 - p;
    }
    return p;
}

/* cv::getGaussianKernel(n, sigma, CV_32F) */
void orc_gaussian_kernel(int n, double sigma, float *k)
{
    if (n == 3 && sigma <= 0) {
        k[0] = 0.25f;
        k[1] = 0.5f;
        k[2] = 0.25f;
        return;
    }
    if (n == 5 && sigma <= 0) {
        const float t[5] = {0.0625f, 0.25f, 0.375f, 0.25f, 0.0625f};
        memcpy(k, t, sizeof(t));
        return;
    }
    if (n == 7 && sigma <= 0) {
        const float t[7] = {0.03125f, 0.109375f, 0.21875f, 0.28125f, 0.21875f, 0.109375f, 0.03125f};
        memcpy(k, t, sizeof(t));
        return;
    }
    const double sigmaX = sigma > 0 ? sigma : ((n - 1) * 0.5 - 1) * 0.3 + 0.8;
    const double scale2X = -0.5 / (sigmaX * sigmaX);
    double sum = 0;
    for (int i = 0; i < n; i++) {
        const double x = i - (n - 1) * 0.5;
        const double t = exp(scale2X * x * x);
        k[i] = (float)t;
        sum += k[i];
    }
    sum = 1. / sum;
    for (int i = 0; i < n; i++) k[i] = (float)(k[i] * sum);
}

/* separable symmetric filter, REFLECT_101: s = k[c]*S[0] + sum_j k[c+j]*(S[+j] + S[-j]) */
static void gaussian_blur(const float *src, int w, int h, int ksize, double sigma, float *dst)
{
    float *k = (float *)malloc(sizeof(float) * (size_t)ksize);
    orc_gaussian_kernel(ksize, sigma, k);
    const int c = ksize / 2;
    float *tmp = (float *)malloc(sizeof(float) * (size_t)w * h);
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const float *s = src + (size_t)y * w;
            float acc = k[c] * s[x];
            for (int j = 1; j <= c; j++) acc += k[c + j] * (s[refl101(x + j, w)] + s[refl101(x - j, w)]);
            tmp[(size_t)y * w + x] = acc;
        }
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            float acc = k[c] * tmp[(size_t)y * w + x];
            for (int j = 1; j <= c; j++)
                acc += k[c + j] * (tmp[(size_t)refl101(y + j, h) * w + x] + tmp[(size_t)refl101(y - j, h) * w + x]);
            dst[(size_t)y * w + x] = acc;
        }
    free(tmp);
    free(k);
}

/* cv::resize INTER_LINEAR for f32 with cn interleaved channels */
static void linear_coeff(int d, int dsize, int ssize, int *ofs, float *a0, float *a1)
{
    const double scale = 1. / ((double)dsize / ssize);
    float f = (float)((d + 0.5) * scale - 0.5);
    int s = (int)floorf(f);
    f -= s;
    if (s < 0) {
        f = 0;
        s = 0;
    }
    if (s >= ssize - 1) {
        f = 0;
        s = ssize - 1;
    }
    *ofs = s;
    *a0 = 1.f - f;
    *a1 = f;
}

static void resize_linear(const float *src, int sw, int sh, int cn, float *dst, int dw, int dh)
{
#pragma omp parallel for schedule(static)
    for (int y = 0; y < dh; y++) {
        int sy;
        float b0, b1;
        linear_coeff(y, dh, sh, &sy, &b0, &b1);
        const int sy1 = sy + 1 < sh ? sy + 1 : sy;
        for (int x = 0; x < dw; x++) {
            int sx;
            float a0, a1;
            linear_coeff(x, dw, sw, &sx, &a0, &a1);
            const int sx1 = sx + 1 < sw ? sx + 1 : sx;
            for (int c = 0; c < cn; c++) {
                const float r0 = src[((size_t)sy * sw + sx) * cn + c] * a0 + src[((size_t)sy * sw + sx1) * cn + c] * a1;
                const float r1 = src[((size_t)sy1 * sw + sx) * cn + c] * a0 + src[((size_t)sy1 * sw + sx1) * cn + c] * a1;
                dst[((size_t)y * dw + x) * cn + c] = r0 * b0 + r1 * b1;
            }
        }
    }
}

/* FarnebackPrepareGaussian: applicability g, x g, x^2 g and the four needed entries of inverse(G) */
void orc_farneback_gaussian(int n, double sigma, float *g, float *xg, float *xxg, double ig[4])
{
    if (sigma < 1.1920929e-07) sigma = n * 0.3;
    double s = 0.;
    for (int x = -n; x <= n; x++) {
        g[x + n] = (float)exp(-x * x / (2 * sigma * sigma));
        s += g[x + n];
    }
    s = 1. / s;
    for (int x = -n; x <= n; x++) {
        g[x + n] = (float)(g[x + n] * s);
        xg[x + n] = (float)(x * g[x + n]);
        xxg[x + n] = (float)(x * x * g[x + n]);
    }
    double G[6][6];
    memset(G, 0, sizeof(G));
    for (int y = -n; y <= n; y++)
        for (int x = -n; x <= n; x++) {
            G[0][0] += g[y + n] * g[x + n];
            G[1][1] += g[y + n] * g[x + n] * x * x;
            G[3][3] += g[y + n] * g[x + n] * x * x * x * x;
            G[5][5] += g[y + n] * g[x + n] * x * x * y * y;
        }
    G[2][2] = G[0][3] = G[0][4] = G[3][0] = G[4][0] = G[1][1];
    G[4][4] = G[3][3];
    G[3][4] = G[4][3] = G[5][5];
    /* inverse by Gauss-Jordan with partial pivoting (G is SPD; OpenCV uses Cholesky) */
    double A[6][12];
    for (int i = 0; i < 6; i++)
        for (int j = 0; j < 6; j++) {
            A[i][j] = G[i][j];
            A[i][j + 6] = i == j;
        }
    for (int c = 0; c < 6; c++) {
        int p = c;
        for (int r = c + 1; r < 6; r++)
            if (fabs(A[r][c]) > fabs(A[p][c])) p = r;
        if (p != c)
            for (int j = 0; j < 12; j++) {
                const double t = A[c][j];
                A[c][j] = A[p][j];
                A[p][j] = t;
            }
        const double d = 1. / A[c][c];
        for (int j = 0; j < 12; j++) A[c][j] *= d;
        for (int r = 0; r < 6; r++)
            if (r != c) {
                const double f = A[r][c];
                if (f != 0)
                    for (int j = 0; j < 12; j++) A[r][j] -= f * A[c][j];
            }
    }
    ig[0] = A[1][7];  /* ig11 */
    ig[1] = A[0][9];  /* ig03 */
    ig[2] = A[3][9];  /* ig33 */
    ig[3] = A[5][11]; /* ig55 */
}

/* FarnebackPolyExp: src w x h f32 -> dst w x h x 5 */
static void poly_exp(const float *src, int w, int h, int n, double sigma, float *dst)
{
    float *g = (float *)malloc(sizeof(float) * (size_t)(2 * n + 1) * 3), *xg = g + 2 * n + 1, *xxg = xg + 2 * n + 1;
    double ig[4];
    orc_farneback_gaussian(n, sigma, g, xg, xxg, ig);
    const double ig11 = ig[0], ig03 = ig[1], ig33 = ig[2], ig55 = ig[3];
#pragma omp parallel
    {
    float *rowbuf = (float *)malloc(sizeof(float) * (size_t)(w + 2 * n) * 3); /* one per thread */
    float *row = rowbuf + n * 3;
#pragma omp for schedule(static)
    for (int y = 0; y < h; y++) {
        const float *s0 = src + (size_t)y * w;
        for (int x = 0; x < w; x++) {
            row[x * 3] = s0[x] * g[n];
            row[x * 3 + 1] = row[x * 3 + 2] = 0.f;
        }
        for (int k = 1; k <= n; k++) {
            const float g0 = g[n + k], g1 = xg[n + k], g2 = xxg[n + k];
            const float *a = src + (size_t)(y - k > 0 ? y - k : 0) * w;
            const float *b = src + (size_t)(y + k < h - 1 ? y + k : h - 1) * w;
            for (int x = 0; x < w; x++) {
                const float p = a[x] + b[x];
                const float t0 = row[x * 3] + g0 * p;
                const float t1 = row[x * 3 + 1] + g1 * (b[x] - a[x]);
                const float t2 = row[x * 3 + 2] + g2 * p;
                row[x * 3] = t0;
                row[x * 3 + 1] = t1;
                row[x * 3 + 2] = t2;
            }
        }
        for (int x = 0; x < n * 3; x++) { /* replicate */
            row[-1 - x] = row[2 - x];
            row[w * 3 + x] = row[w * 3 + x - 3];
        }
        float *d = dst + (size_t)y * w * 5;
        for (int x = 0; x < w; x++) {
            float g0 = g[n];
            double b1 = row[x * 3] * g0, b2 = 0, b3 = row[x * 3 + 1] * g0, b4 = 0, b5 = row[x * 3 + 2] * g0, b6 = 0;
            for (int k = 1; k <= n; k++) {
                const double tg = row[(x + k) * 3] + row[(x - k) * 3];
                g0 = g[n + k];
                b1 += tg * g0;
                b4 += tg * xxg[n + k];
                b2 += (row[(x + k) * 3] - row[(x - k) * 3]) * xg[n + k];
                b3 += (row[(x + k) * 3 + 1] + row[(x - k) * 3 + 1]) * g0;
                b6 += (row[(x + k) * 3 + 1] - row[(x - k) * 3 + 1]) * xg[n + k];
                b5 += (row[(x + k) * 3 + 2] + row[(x - k) * 3 + 2]) * g0;
            }
            d[x * 5 + 1] = (float)(b2 * ig11);
            d[x * 5] = (float)(b3 * ig11);
            d[x * 5 + 3] = (float)(b1 * ig03 + b4 * ig33);
            d[x * 5 + 2] = (float)(b1 * ig03 + b5 * ig33);
            d[x * 5 + 4] = (float)(b6 * ig55);
        }
    }
    free(rowbuf);
    }
    free(g);
}

/* exposed for the independent cross-checks in tests/test_pinning_cpu.py */
void orc_poly_exp(const float *src, int w, int h, int n, double sigma, float *dst5) { poly_exp(src, w, h, n, sigma, dst5); }
void orc_gaussian_blur(const float *src, int w, int h, int ksize, double sigma, float *dst) { gaussian_blur(src, w, h, ksize, sigma, dst); }
void orc_resize_linear(const float *src, int sw, int sh, int cn, float *dst, int dw, int dh) { resize_linear(src, sw, sh, cn, dst, dw, dh); }

/* FarnebackUpdateMatrices */
static void update_matrices(const float *R0, const float *R1, const float *flow, int w, int h, float *M)
{
    static const float border[5] = {0.14f, 0.14f, 0.4472f, 0.4472f, 0.4472f};
    const size_t step1 = (size_t)w * 5;
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const float *r0 = R0 + ((size_t)y * w + x) * 5;
            const float dx = flow[((size_t)y * w + x) * 2], dy = flow[((size_t)y * w + x) * 2 + 1];
            float fx = x + dx, fy = y + dy;
            const int x1 = (int)floorf(fx), y1 = (int)floorf(fy);
            float r2, r3, r4, r5, r6;
            fx -= x1;
            fy -= y1;
            if ((unsigned)x1 < (unsigned)(w - 1) && (unsigned)y1 < (unsigned)(h - 1)) {
                const float *p = R1 + (size_t)y1 * step1 + (size_t)x1 * 5;
                const float a00 = (1.f - fx) * (1.f - fy), a01 = fx * (1.f - fy), a10 = (1.f - fx) * fy, a11 = fx * fy;
                r2 = a00 * p[0] + a01 * p[5] + a10 * p[step1] + a11 * p[step1 + 5];
                r3 = a00 * p[1] + a01 * p[6] + a10 * p[step1 + 1] + a11 * p[step1 + 6];
                r4 = a00 * p[2] + a01 * p[7] + a10 * p[step1 + 2] + a11 * p[step1 + 7];
                r5 = a00 * p[3] + a01 * p[8] + a10 * p[step1 + 3] + a11 * p[step1 + 8];
                r6 = a00 * p[4] + a01 * p[9] + a10 * p[step1 + 4] + a11 * p[step1 + 9];
                r4 = (r0[2] + r4) * 0.5f;
                r5 = (r0[3] + r5) * 0.5f;
                r6 = (r0[4] + r6) * 0.25f;
            } else {
                r2 = r3 = 0.f;
                r4 = r0[2];
                r5 = r0[3];
                r6 = r0[4] * 0.5f;
            }
            r2 = (r0[0] - r2) * 0.5f;
            r3 = (r0[1] - r3) * 0.5f;
            r2 += r4 * dy + r6 * dx;
            r3 += r6 * dy + r5 * dx;
            if ((unsigned)(x - 5) >= (unsigned)(w - 10) || (unsigned)(y - 5) >= (unsigned)(h - 10)) {
                const float scale = (x < 5 ? border[x] : 1.f) * (x >= w - 5 ? border[w - x - 1] : 1.f) *
                                    (y < 5 ? border[y] : 1.f) * (y >= h - 5 ? border[h - y - 1] : 1.f);
                r2 *= scale;
                r3 *= scale;
                r4 *= scale;
                r5 *= scale;
                r6 *= scale;
            }
            float *m = M + ((size_t)y * w + x) * 5;
            m[0] = r4 * r4 + r6 * r6;
            m[1] = (r4 + r5) * r6;
            m[2] = r5 * r5 + r6 * r6;
            m[3] = r4 * r2 + r6 * r3;
            m[4] = r6 * r2 + r5 * r3;
        }
}

/* FarnebackUpdateFlow_Blur: box-blur M (2m+1 replicated taps per axis, scale 1/block^2), solve for the flow */
static void update_flow_blur(const float *M, int w, int h, int block, float *flow, double *vs /* w*h*5 scratch */)
{
    const int m = block / 2;
    const double scale = 1. / ((double)block * block);
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
            for (int c = 0; c < 5; c++) {
                double s = 0;
                for (int d = -m; d <= m; d++) s += M[((size_t)clampi(y + d, 0, h - 1) * w + x) * 5 + c];
                vs[((size_t)y * w + x) * 5 + c] = s;
            }
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            double t[5];
            for (int c = 0; c < 5; c++) {
                double s = 0;
                for (int d = -m; d <= m; d++) s += vs[((size_t)y * w + clampi(x + d, 0, w - 1)) * 5 + c];
                t[c] = s * scale;
            }
            const double idet = 1. / (t[0] * t[2] - t[1] * t[1] + 1e-3);
            flow[((size_t)y * w + x) * 2] = (float)((t[0] * t[4] - t[1] * t[3]) * idet);
            flow[((size_t)y * w + x) * 2 + 1] = (float)((t[2] * t[3] - t[1] * t[4]) * idet);
        }
}

/* number of pyramid levels beyond level 0 and their geometry (OpenCV: stop when a level would be < 32 px) */
int orc_farneback_levels(int W, int H, int levels, double pyr_scale, int *lw, int *lh, double *lscale)
{
    int k;
    double scale = 1;
    for (k = 0; k < levels; k++) {
        scale *= pyr_scale;
        if (W * scale < 32 || H * scale < 32) break;
    }
    levels = k;
    for (k = 0; k <= levels; k++) {
        scale = 1;
        for (int i = 0; i < k; i++) scale *= pyr_scale;
        lscale[k] = scale;
        lw[k] = (int)lrint(W * scale);
        lh[k] = (int)lrint(H * scale);
    }
    return levels;
}

void orc_farneback(const uint8_t *prev, const uint8_t *next, int W, int H, int levels, double pyr_scale, int winsize,
                   int iterations, int poly_n, double poly_sigma, float *flow_out /* H*W*2 */)
{
    int lw[64], lh[64];
    double ls[64];
    levels = orc_farneback_levels(W, H, levels, pyr_scale, lw, lh, ls);
    const size_t P = (size_t)W * H;
    float *f0 = (float *)malloc(sizeof(float) * P), *f1 = (float *)malloc(sizeof(float) * P);
    for (size_t i = 0; i < P; i++) {
        f0[i] = (float)prev[i];
        f1[i] = (float)next[i];
    }
    float *blur = (float *)malloc(sizeof(float) * P), *I = (float *)malloc(sizeof(float) * P);
    float *R[2] = {(float *)malloc(sizeof(float) * P * 5), (float *)malloc(sizeof(float) * P * 5)};
    float *M = (float *)malloc(sizeof(float) * P * 5);
    double *vs = (double *)malloc(sizeof(double) * P * 5);
    float *flow = NULL, *prevflow = NULL;
    int pw = 0, ph = 0;
    for (int k = levels; k >= 0; k--) {
        const double scale = ls[k];
        const double sigma = (1. / scale - 1) * 0.5;
        int smooth_sz = (int)lrint(sigma * 5) | 1;
        if (smooth_sz < 3) smooth_sz = 3;
        const int w = lw[k], h = lh[k];
        flow = (float *)malloc(sizeof(float) * (size_t)w * h * 2);
        if (!prevflow) {
            memset(flow, 0, sizeof(float) * (size_t)w * h * 2); /* flags == 0: no initial flow */
        } else {
            resize_linear(prevflow, pw, ph, 2, flow, w, h);
            const float mul = (float)(1. / pyr_scale);
            for (size_t i = 0; i < (size_t)w * h * 2; i++) flow[i] *= mul;
            free(prevflow);
        }
        for (int i = 0; i < 2; i++) {
            gaussian_blur(i == 0 ? f0 : f1, W, H, smooth_sz, sigma, blur);
            resize_linear(blur, W, H, 1, I, w, h);
            poly_exp(I, w, h, poly_n, poly_sigma, R[i]);
        }
        update_matrices(R[0], R[1], flow, w, h, M);
        for (int it = 0; it < iterations; it++) {
            update_flow_blur(M, w, h, winsize, flow, vs);
            if (it < iterations - 1) update_matrices(R[0], R[1], flow, w, h, M);
        }
        prevflow = flow;
        pw = w;
        ph = h;
    }
    memcpy(flow_out, flow, sizeof(float) * P * 2);
    free(flow);
    free(vs);
    free(M);
    free(R[0]);
    free(R[1]);
    free(I);
    free(blur);
    free(f0);
    free(f1);
}

/* ------------------------------------------------------------------------------------------------------------------
 * Variational refinement (flow.cpp:29, the reference's default: useFarneback = false, configuration.cpp:26).
 *
 * OpenCV's VariationalRefinement with its defaults: fixedPointIterations 5, sorIterations 5, omega 1.6, alpha 20,
 * delta 5, gamma 10, zeta 0.1, epsilon 0.001.  One call refines the flow W = (u, v) it is given:
 *   warp      I1 (f32) sampled at (x + u, y + v): bilinear at 1/32-pixel quantised positions (cv::remap), BORDER_REPLICATE
 *   images    A = (I0 + warped)/2,  Iz = warped - I0
 *   gradients central differences without the 1/2 (Sobel ksize 1, BORDER_REPLICATE):
 *             Ix, Iy of A;  Ixx, Ixy of Ix;  Iyy of Iy;  Ixz, Iyz of Iz
 *   repeat fixedPointIterations times, with increment dW = 0 initially:
 *     data term     robust (Charbonnier) brightness + gradient constancy, normalised by gradient magnitudes
 *     smoothness    diffusivity w = (alpha/2)/sqrt(|grad(u+du)|^2 + |grad(v+dv)|^2 + eps^2) from forward differences,
 *                   one weight per pixel used for its right and down edges
 *     sorIterations red-black SOR sweeps on the 2x2-block linear system (red = (x + y) even first)
 *   W += dW
 * Sums are accumulated in the fixed order written below so the HIP kernels reproduce them bit for bit.
 * ------------------------------------------------------------------------------------------------------------------ */

static void warp_linear_q5(const float *img, int W, int H, const float *u, const float *v, float *out)
{
#pragma omp parallel for schedule(static)
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            const size_t p = (size_t)y * W + x;
            const float mx = (float)x + u[p], my = (float)y + v[p];
            const int qx = (int)lrintf(mx * 32.0f), qy = (int)lrintf(my * 32.0f);
            const int sx = qx >> 5, sy = qy >> 5;
            const float fx = (float)(qx & 31) * (1.0f / 32), fy = (float)(qy & 31) * (1.0f / 32);
            const float w0 = (1.f - fy) * (1.f - fx), w1 = (1.f - fy) * fx, w2 = fy * (1.f - fx), w3 = fy * fx;
            const int x0 = clampi(sx, 0, W - 1), x1 = clampi(sx + 1, 0, W - 1), y0 = clampi(sy, 0, H - 1), y1 = clampi(sy + 1, 0, H - 1);
            out[p] = img[(size_t)y0 * W + x0] * w0 + img[(size_t)y0 * W + x1] * w1 + img[(size_t)y1 * W + x0] * w2 +
                     img[(size_t)y1 * W + x1] * w3;
        }
}

static void ddx(const float *a, int W, int H, float *o)
{
#pragma omp parallel for schedule(static)
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) o[(size_t)y * W + x] = a[(size_t)y * W + clampi(x + 1, 0, W - 1)] - a[(size_t)y * W + clampi(x - 1, 0, W - 1)];
}
static void ddy(const float *a, int W, int H, float *o)
{
#pragma omp parallel for schedule(static)
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) o[(size_t)y * W + x] = a[(size_t)clampi(y + 1, 0, H - 1) * W + x] - a[(size_t)clampi(y - 1, 0, H - 1) * W + x];
}

void orc_variational_refine(const uint8_t *I0u, const uint8_t *I1u, int W, int H, float *flow /* H*W*2 in/out */)
{
    const int fixedPointIterations = 5, sorIterations = 5;
    const float alpha = 20.f, delta = 5.f, gamma = 10.f, omega = 1.6f, zeta = 0.1f, epsilon = 0.001f;
    const float zeta2 = zeta * zeta, eps2 = epsilon * epsilon, alpha2 = alpha / 2, gamma2 = gamma / 2, delta2 = delta / 2;
    const size_t P = (size_t)W * H;
    float *buf = (float *)malloc(sizeof(float) * P * 24);
    float *I0 = buf, *I1 = I0 + P, *Wu = I1 + P, *Wv = Wu + P, *du = Wv + P, *dv = du + P, *warped = dv + P, *A = warped + P,
          *Iz = A + P, *Ix = Iz + P, *Iy = Ix + P, *Ixx = Iy + P, *Ixy = Ixx + P, *Iyy = Ixy + P, *Ixz = Iyy + P, *Iyz = Ixz + P,
          *a11 = Iyz + P, *a12 = a11 + P, *a22 = a12 + P, *b1 = a22 + P, *b2 = b1 + P, *wgt = b2 + P;
    for (size_t i = 0; i < P; i++) {
        I0[i] = (float)I0u[i];
        I1[i] = (float)I1u[i];
        Wu[i] = flow[2 * i];
        Wv[i] = flow[2 * i + 1];
        du[i] = dv[i] = 0.f;
    }
    warp_linear_q5(I1, W, H, Wu, Wv, warped);
    for (size_t i = 0; i < P; i++) {
        A[i] = (I0[i] + warped[i]) * 0.5f;
        Iz[i] = warped[i] - I0[i];
    }
    ddx(A, W, H, Ix);
    ddy(A, W, H, Iy);
    ddx(Iz, W, H, Ixz);
    ddy(Iz, W, H, Iyz);
    ddx(Ix, W, H, Ixx);
    ddy(Ix, W, H, Ixy);
    ddy(Iy, W, H, Iyy);

    for (int fp = 0; fp < fixedPointIterations; fp++) {
        /* data term */
#pragma omp parallel for schedule(static)
        for (size_t p = 0; p < P; p++) {
            float derivNorm = Ix[p] * Ix[p] + Iy[p] * Iy[p] + zeta2;
            const float Ik1z = Iz[p] + Ix[p] * du[p] + Iy[p] * dv[p];
            float weight = (delta2 / sqrtf(Ik1z * Ik1z / derivNorm + eps2)) / derivNorm;
            float A11 = weight * (Ix[p] * Ix[p]) + zeta2;
            float A12 = weight * (Ix[p] * Iy[p]);
            float A22 = weight * (Iy[p] * Iy[p]) + zeta2;
            float B1 = -weight * (Iz[p] * Ix[p]);
            float B2 = -weight * (Iz[p] * Iy[p]);
            derivNorm = Ixx[p] * Ixx[p] + Ixy[p] * Ixy[p] + zeta2;
            const float derivNorm2 = Iyy[p] * Iyy[p] + Ixy[p] * Ixy[p] + zeta2;
            const float Ik1zx = Ixz[p] + Ixx[p] * du[p] + Ixy[p] * dv[p];
            const float Ik1zy = Iyz[p] + Ixy[p] * du[p] + Iyy[p] * dv[p];
            weight = gamma2 / sqrtf(Ik1zx * Ik1zx / derivNorm + Ik1zy * Ik1zy / derivNorm2 + eps2);
            A11 += weight * (Ixx[p] * Ixx[p] / derivNorm + Ixy[p] * Ixy[p] / derivNorm2);
            A12 += weight * (Ixx[p] * Ixy[p] / derivNorm + Ixy[p] * Iyy[p] / derivNorm2);
            A22 += weight * (Ixy[p] * Ixy[p] / derivNorm + Iyy[p] * Iyy[p] / derivNorm2);
            B1 += -weight * (Ixx[p] * Ixz[p] / derivNorm + Ixy[p] * Iyz[p] / derivNorm2);
            B2 += -weight * (Ixy[p] * Ixz[p] / derivNorm + Iyy[p] * Iyz[p] / derivNorm2);
            a11[p] = A11;
            a12[p] = A12;
            a22[p] = A22;
            b1[p] = B1;
            b2[p] = B2;
        }
        /* diffusivity from the current total flow */
#pragma omp parallel for schedule(static)
        for (int y = 0; y < H; y++)
            for (int x = 0; x < W; x++) {
                const size_t p = (size_t)y * W + x;
                const float cu = Wu[p] + du[p], cv = Wv[p] + dv[p];
                const float ux = x + 1 < W ? (Wu[p + 1] + du[p + 1]) - cu : 0.f, vx = x + 1 < W ? (Wv[p + 1] + dv[p + 1]) - cv : 0.f;
                const float uy = y + 1 < H ? (Wu[p + W] + du[p + W]) - cu : 0.f, vy = y + 1 < H ? (Wv[p + W] + dv[p + W]) - cv : 0.f;
                wgt[p] = alpha2 / sqrtf(ux * ux + vx * vx + uy * uy + vy * vy + eps2);
            }
        /* smoothness contributions, gathered per pixel in the order left edge, right edge, upper edge, lower edge */
#pragma omp parallel for schedule(static)
        for (int y = 0; y < H; y++)
            for (int x = 0; x < W; x++) {
                const size_t p = (size_t)y * W + x;
                float A11 = a11[p], A22 = a22[p], B1 = b1[p], B2 = b2[p];
                if (x > 0) {
                    const float w = wgt[p - 1];
                    B1 -= w * (Wu[p] - Wu[p - 1]);
                    B2 -= w * (Wv[p] - Wv[p - 1]);
                    A11 += w;
                    A22 += w;
                }
                if (x + 1 < W) {
                    const float w = wgt[p];
                    B1 += w * (Wu[p + 1] - Wu[p]);
                    B2 += w * (Wv[p + 1] - Wv[p]);
                    A11 += w;
                    A22 += w;
                }
                if (y > 0) {
                    const float w = wgt[p - W];
                    B1 -= w * (Wu[p] - Wu[p - W]);
                    B2 -= w * (Wv[p] - Wv[p - W]);
                    A11 += w;
                    A22 += w;
                }
                if (y + 1 < H) {
                    const float w = wgt[p];
                    B1 += w * (Wu[p + W] - Wu[p]);
                    B2 += w * (Wv[p + W] - Wv[p]);
                    A11 += w;
                    A22 += w;
                }
                a11[p] = A11;
                a22[p] = A22;
                b1[p] = B1;
                b2[p] = B2;
            }
        /* red-black SOR: a half-sweep reads the other colour's du, dv (and the pixel's own), so its rows are independent */
        for (int it = 0; it < sorIterations; it++)
            for (int colour = 0; colour < 2; colour++)
#pragma omp parallel for schedule(static)
                for (int y = 0; y < H; y++)
                    for (int x = (y + colour) & 1; x < W; x += 2) {
                        const size_t p = (size_t)y * W + x;
                        float sU = 0.f, sV = 0.f;
                        if (x > 0) {
                            sU += wgt[p - 1] * du[p - 1];
                            sV += wgt[p - 1] * dv[p - 1];
                        }
                        if (x + 1 < W) {
                            sU += wgt[p] * du[p + 1];
                            sV += wgt[p] * dv[p + 1];
                        }
                        if (y > 0) {
                            sU += wgt[p - W] * du[p - W];
                            sV += wgt[p - W] * dv[p - W];
                        }
                        if (y + 1 < H) {
                            sU += wgt[p] * du[p + W];
                            sV += wgt[p] * dv[p + W];
                        }
                        du[p] += omega * ((sU + b1[p] - dv[p] * a12[p]) / a11[p] - du[p]);
                        dv[p] += omega * ((sV + b2[p] - du[p] * a12[p]) / a22[p] - dv[p]);
                    }
    }
    for (size_t i = 0; i < P; i++) {
        flow[2 * i] = Wu[i] + du[i];
        flow[2 * i + 1] = Wv[i] + dv[i];
    }
    free(buf);
}

/* flow.cpp:19-42 */
void orc_calculate_flow(const uint8_t *prev, const uint8_t *next, int W, int H, int use_farneback, float *out4)
{
    const size_t P = (size_t)W * H;
    float *flow = (float *)calloc(P * 2, sizeof(float)); /* flow.cpp:31 leaves it uninitialised; zero here (A-11) */
    if (use_farneback) {
        const double poly_sigma = (H + W) / 1000.0; /* flow.cpp:24 */
        const int winsize = (H + W) / 100, poly_n = poly_sigma < 1.5 ? 5 : 7;
        orc_farneback(prev, next, W, H, 10, 0.8, winsize, 7, poly_n, poly_sigma, flow);
    } else {
        orc_variational_refine(prev, next, W, H, flow);
    }
    uint8_t *remapped = (uint8_t *)malloc(P);
    float *var = (float *)malloc(sizeof(float) * P);
    orc_flow_remap(flow, 2, next, W, H, remapped); /* flow.cpp:34 */
    orc_compare_u8(prev, remapped, W, H, var);
    for (size_t i = 0; i < P; i++) { /* flow.cpp:37-40: (u, v, variance, 0) */
        out4[4 * i] = flow[2 * i];
        out4[4 * i + 1] = flow[2 * i + 1];
        out4[4 * i + 2] = var[i];
        out4[4 * i + 3] = 0.f;
    }
    free(var);
    free(remapped);
    free(flow);
}
