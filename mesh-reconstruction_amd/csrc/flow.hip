// flow.hip -- calculateFlow() of the reference (flow.cpp:19-42) on gfx950: dense optical flow (u, v), the
// per-pixel "variance" channel compare(prev, flowRemap(flow, next)) and the CV_32FC4 packing.
//
// The reference calls OpenCV: cv::FarnebackOpticalFlow::create(10, 0.8, false, (H+W)/100, 7, 5|7, (H+W)/1000, 0)
// (flow.cpp:24-26) or cv::optflow::createVariationalFlowRefinement() (flow.cpp:29, the default path).  OpenCV is
// neither in the reference tree nor in this image, so both algorithms are restated from their publications in
// OpenCV's organisation; the operation order is the one documented in oracle/flow_oracle.c, which these kernels
// reproduce bit for bit (f32 ops one rounding each under -ffp-contract=off, f64 where OpenCV accumulates in double).
//
// Farneback, per pyramid level (coarse to fine): gauss (rows, columns) -> resize_linear -> polyexp_vert/polyexp_horiz,
// both frames per launch (blockIdx.z), then update_matrices and `iterations` x farneback_iteration_fused (vertical box
// sums in LDS + horizontal sums + 2x2 solve + the next update matrices; the three separate kernels remain as the A/B form).
// Everything stays in one HBM arena per context; the reference's per-level Mat allocations and the CPU
// round trip of every intermediate disappear.
//
// Variational refinement: warp_q5 -> avg_diff -> central differences -> 5 x var_fixed_point_fused (data term,
// diffusivity, smoothness gather and the 5 red-black SOR sweeps of one fixed-point iteration in ONE launch by temporal
// blocking in LDS; the 13 separate kernels remain as the A/B form) -> add increment.
#include "mvs_internal.hpp"

#include <atomic>

#include <cmath>
#include <cstdlib>

namespace mvs {

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
__device__ __forceinline__ int refl101(int p, int n)
{
    if (n == 1) return 0;
    while (p < 0 || p >= n) {
        if (p < 0) p = -p;
        if (p >= n) p = 2 * n - 2 - p;
    }
    return p;
}

struct Taps {
    float k[64];
};

#define PIX2D                                             \
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);   \
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);    \
    if (x >= w || y >= h) return;

__global__ __launch_bounds__(256) void zero_f32_kernel(float *__restrict__ d, size_t n)  // (test hook MVS_FLOW_GRAPH=2: a kernel in place of the captured hipMemsetAsync nodes)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) d[i] = 0.0f;
}

__global__ __launch_bounds__(256) void u8_to_f32_kernel(const uint8_t *__restrict__ s, float *__restrict__ d, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) d[i] = (float)s[i];
}

__global__ __launch_bounds__(256) void u8_to_f32_pair_kernel(const uint8_t *__restrict__ s0, const uint8_t *__restrict__ s1, float *__restrict__ d0, float *__restrict__ d1, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        d0[i] = (float)s0[i];
        d1[i] = (float)s1[i];
    }
}

// symmetric separable filter, REFLECT_101: acc = k[c]*S[0]; acc += k[c+j]*(S[+j] + S[-j])
template <bool COLS>
__global__ __launch_bounds__(256) void gauss_kernel(const float *__restrict__ src, int w, int h, Taps t, int ksize,
                                                    float *__restrict__ dst, ptrdiff_t src_z = 0, ptrdiff_t dst_z = 0)
{
    PIX2D
    src += (ptrdiff_t)blockIdx.z * src_z;  // blockIdx.z: the second frame of a pair, processed by the same launch
    dst += (ptrdiff_t)blockIdx.z * dst_z;
    const int c = ksize / 2;
    float acc = t.k[c] * src[(size_t)y * w + x];
    for (int j = 1; j <= c; j++) {
        float a, b;
        if (COLS) {
            a = src[(size_t)refl101(y + j, h) * w + x];
            b = src[(size_t)refl101(y - j, h) * w + x];
        } else {
            a = src[(size_t)y * w + refl101(x + j, w)];
            b = src[(size_t)y * w + refl101(x - j, w)];
        }
        acc += t.k[c + j] * (a + b);
    }
    dst[(size_t)y * w + x] = acc;
}

// Both passes of the separable filter in ONE launch (round 5): a workgroup stages its 64 x 16 output tile's (64 + 2c) x (16 + 2c) input
// neighbourhood in LDS (reflected coordinates resolved while staging), runs the row filter over the 16 + 2c rows into a second LDS array and
// the column filter from there -- the expressions of gauss_kernel<false> and <true> on the same values in the same order (bit-identical),
// without the intermediate image's round trip through HBM: cv::GaussianBlur of BOTH full-resolution frames is paid once per pyramid
// level (fastPyramids false), 22 passes over 16.6 MB per 1080p flow.
// (body: src / dst already point at the image; kc[j] = tap c + j of the symmetric kernel, j = 0 .. c)
__device__ __forceinline__ void gauss_fused_body(const float *__restrict__ src, int w, int h, const float *__restrict__ kc, int c, float *__restrict__ dst,
                                                 int X0, int Y0, float *__restrict__ g_lds)
{
    const int IW = 64 + 2 * c, IH = 16 + 2 * c;
    float *in = g_lds, *tmp = g_lds + IW * IH;
    for (int i = threadIdx.x; i < IW * IH; i += 256) {
        const int r = i / IW, q = i - r * IW;
        in[i] = src[(size_t)refl101(Y0 - c + r, h) * w + refl101(X0 - c + q, w)];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * IH; i += 256) {  // rows: gauss_kernel<false> at (refl(Y0 - c + r), X0 + x)
        const int r = i >> 6, x = i & 63;
        const float *p = in + r * IW + x + c;
        float acc = kc[0] * p[0];
        for (int j = 1; j <= c; j++) acc += kc[j] * (p[j] + p[-j]);
        tmp[i] = acc;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 16; i += 256) {  // columns: gauss_kernel<true> at (Y0 + r, X0 + x)
        const int r = i >> 6, x = i & 63, gy = Y0 + r, gx = X0 + x;
        if (gx >= w || gy >= h) continue;
        const float *p = tmp + (r + c) * 64 + x;
        float acc = kc[0] * p[0];
        for (int j = 1; j <= c; j++) acc += kc[j] * (p[64 * j] + p[-64 * j]);
        dst[(size_t)gy * w + gx] = acc;
    }
}

__global__ __launch_bounds__(256) void gauss_fused_kernel(const float *__restrict__ src, int w, int h, Taps t, int ksize, float *__restrict__ dst,
                                                          ptrdiff_t src_z, ptrdiff_t dst_z)
{
    extern __shared__ float g_lds[];
    const int c = ksize / 2;
    gauss_fused_body(src + (ptrdiff_t)blockIdx.z * src_z, w, h, t.k + c, c, dst + (ptrdiff_t)blockIdx.z * dst_z, blockIdx.x * 64, blockIdx.y * 16, g_lds);
}

// The pyramid preparation of ALL levels in three launches (round 6).  cv::FarnebackOpticalFlow blurs both full-resolution frames once per
// level (fastPyramids false), resizes and expands them -- 3 launches per level, 33 per flow, none of which depends on the flow chain; at
// 640 x 480 they were a quarter of a call's time, each on the critical path (and a second stream does not hide them: a cross-stream event
// wait costs ~25 us on this ROCm, measured -- DESIGN.md section 6).  Here blockIdx.z = level * images + image and the level's constants
// come from the kernel arguments; the bodies are the per-level kernels' own.  Needs every level's blur / I / R at once (FbLevels::*_off,
// in floats from the bases): used for frames up to FB_BATCH_PREP_MAX_PIXELS, where launches -- not bytes -- are what a flow costs.
struct FbLevelDesc {
    int w, h, c, pad;              // level size, half width of the blur kernel
    size_t blur_off, i_off, r_off; // per level; images n * W*H (blur), n * w*h (I), n * 5 w*h (R) apart inside it
    float kc[32];                  // taps c .. 2c of the symmetric blur kernel
};
struct FbLevels {
    int n, nimg;
    FbLevelDesc lv[11];
};
constexpr size_t FB_BATCH_PREP_MAX_PIXELS = 1280 * 720;

__global__ __launch_bounds__(256) void gauss_fused_levels_kernel(const float *__restrict__ F, int W, int H, FbLevels L, float *__restrict__ blur)
{
    extern __shared__ float g_lds[];
    const int lev = blockIdx.z / L.nimg, img = blockIdx.z - lev * L.nimg;
    const FbLevelDesc &d = L.lv[lev];
    const size_t P = (size_t)W * H;
    gauss_fused_body(F + img * P, W, H, d.kc, d.c, blur + d.blur_off + img * P, blockIdx.x * 64, blockIdx.y * 16, g_lds);
}

__device__ __forceinline__ void linear_coeff(int d, int dsize, int ssize, int &ofs, float &a0, float &a1)
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
    ofs = s;
    a0 = 1.f - f;
    a1 = f;
}

// cv::resize INTER_LINEAR (f32, CN interleaved channels), optionally followed by `*= mul` (flow *= 1/pyrScale)
template <int CN>
__global__ __launch_bounds__(256) void resize_linear_kernel(const float *__restrict__ src, int sw, int sh,
                                                            float *__restrict__ dst, int w, int h, float mul, int domul,
                                                            ptrdiff_t src_z = 0, ptrdiff_t dst_z = 0)
{
    PIX2D
    src += (ptrdiff_t)blockIdx.z * src_z;
    dst += (ptrdiff_t)blockIdx.z * dst_z;
    int sx, sy;
    float a0, a1, b0, b1;
    linear_coeff(x, w, sw, sx, a0, a1);
    linear_coeff(y, h, sh, sy, b0, b1);
    const int sx1 = sx + 1 < sw ? sx + 1 : sx, sy1 = sy + 1 < sh ? sy + 1 : sy;
#pragma unroll
    for (int c = 0; c < CN; c++) {
        const float r0 = src[((size_t)sy * sw + sx) * CN + c] * a0 + src[((size_t)sy * sw + sx1) * CN + c] * a1;
        const float r1 = src[((size_t)sy1 * sw + sx) * CN + c] * a0 + src[((size_t)sy1 * sw + sx1) * CN + c] * a1;
        float v = r0 * b0 + r1 * b1;
        if (domul) v *= mul;
        dst[((size_t)y * w + x) * CN + c] = v;
    }
}

__global__ __launch_bounds__(256) void resize_levels_kernel(const float *__restrict__ blur, int W, int H, FbLevels L, float *__restrict__ I)
{
    const int lev = blockIdx.z / L.nimg, img = blockIdx.z - lev * L.nimg;
    const FbLevelDesc &d = L.lv[lev];
    const int w = d.w, h = d.h;
    PIX2D
    const float *src = blur + d.blur_off + (size_t)img * W * H;
    float *dst = I + d.i_off + (size_t)img * w * h;
    int sx, sy;
    float a0, a1, b0, b1;
    linear_coeff(x, w, W, sx, a0, a1);
    linear_coeff(y, h, H, sy, b0, b1);
    const int sx1 = sx + 1 < W ? sx + 1 : sx, sy1 = sy + 1 < H ? sy + 1 : sy;
    const float r0 = src[(size_t)sy * W + sx] * a0 + src[(size_t)sy * W + sx1] * a1;   // resize_linear_kernel<1>, no `*= mul`
    const float r1 = src[(size_t)sy1 * W + sx] * a0 + src[(size_t)sy1 * W + sx1] * a1;
    dst[(size_t)y * w + x] = r0 * b0 + r1 * b1;
}

struct PolyTaps {
    float g[16], xg[16], xxg[16];  // index k = 0..n (symmetric / antisymmetric halves)
    double ig11, ig03, ig33, ig55;
    int n;
};

// FarnebackPolyExp, vertical pass: row[x] = (sum g I, sum y g I, sum y^2 g I) over the column, replicate border
__global__ __launch_bounds__(256) void polyexp_vert(const float *__restrict__ src, int w, int h, PolyTaps t,
                                                    float *__restrict__ row3, ptrdiff_t src_z = 0, ptrdiff_t dst_z = 0)
{
    PIX2D
    src += (ptrdiff_t)blockIdx.z * src_z;
    row3 += (ptrdiff_t)blockIdx.z * dst_z;
    float t0 = src[(size_t)y * w + x] * t.g[0], t1 = 0.f, t2 = 0.f;
    for (int k = 1; k <= t.n; k++) {
        const float a = src[(size_t)(y - k > 0 ? y - k : 0) * w + x];
        const float b = src[(size_t)(y + k < h - 1 ? y + k : h - 1) * w + x];
        const float p = a + b;
        t0 = t0 + t.g[k] * p;
        t1 = t1 + t.xg[k] * (b - a);
        t2 = t2 + t.xxg[k] * p;
    }
    float *r = row3 + ((size_t)y * w + x) * 3;
    r[0] = t0;
    r[1] = t1;
    r[2] = t2;
}

// horizontal pass + projection onto the polynomial basis: 5 coefficients per pixel (y, x, y^2, x^2, xy)
__global__ __launch_bounds__(256) void polyexp_horiz(const float *__restrict__ row3, int w, int h, PolyTaps t,
                                                     float *__restrict__ dst5, ptrdiff_t src_z = 0, ptrdiff_t dst_z = 0)
{
    PIX2D
    row3 += (ptrdiff_t)blockIdx.z * src_z;
    dst5 += (ptrdiff_t)blockIdx.z * dst_z;
    const float *base = row3 + (size_t)y * w * 3;
    const float *c = base + (size_t)x * 3;
    float g0 = t.g[0];
    double b1 = c[0] * g0, b2 = 0, b3 = c[1] * g0, b4 = 0, b5 = c[2] * g0, b6 = 0;
    for (int k = 1; k <= t.n; k++) {
        const float *p = base + (size_t)clampi(x + k, 0, w - 1) * 3;
        const float *m = base + (size_t)clampi(x - k, 0, w - 1) * 3;
        const double tg = p[0] + m[0];
        g0 = t.g[k];
        b1 += tg * g0;
        b4 += tg * t.xxg[k];
        b2 += (p[0] - m[0]) * t.xg[k];
        b3 += (p[1] + m[1]) * g0;
        b6 += (p[1] - m[1]) * t.xg[k];
        b5 += (p[2] + m[2]) * g0;
    }
    float *d = dst5 + ((size_t)y * w + x) * 5;
    d[1] = (float)(b2 * t.ig11);
    d[0] = (float)(b3 * t.ig11);
    d[3] = (float)(b1 * t.ig03 + b4 * t.ig33);
    d[2] = (float)(b1 * t.ig03 + b5 * t.ig33);
    d[4] = (float)(b6 * t.ig55);
}

// polyexp_vert + polyexp_horiz in one launch (round 5): the tile's (16 + 2n) x (64 + 2n) neighbourhood staged in LDS with the replicate
// border resolved while staging, the vertical sums of its 16 rows x (64 + 2n) columns into a second LDS array, the horizontal pass from
// there -- the same expressions on the same values (bit-identical), no round trip of the 12-byte-per-pixel intermediate through HBM.
__device__ __forceinline__ void polyexp_fused_body(const float *__restrict__ src, int w, int h, const PolyTaps &t, float *__restrict__ dst5, int X0, int Y0,
                                                   float *__restrict__ p_lds)
{
    const int n = t.n;
    const int IW = 64 + 2 * n, IH = 16 + 2 * n;
    float *in = p_lds, *row3 = p_lds + IW * IH;  // row3: [16][IW][3]
    for (int i = threadIdx.x; i < IW * IH; i += 256) {
        const int r = i / IW, q = i - r * IW;
        in[i] = src[(size_t)clampi(Y0 - n + r, 0, h - 1) * w + clampi(X0 - n + q, 0, w - 1)];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 16 * IW; i += 256) {  // polyexp_vert at (Y0 + r, clamp(X0 - n + q)): rows y -+ k are tile rows r + n -+ k
        const int r = i / IW, q = i - r * IW;
        const float *c = in + (r + n) * IW + q;
        float t0 = c[0] * t.g[0], t1 = 0.f, t2 = 0.f;
        for (int k = 1; k <= n; k++) {
            const float a = c[-k * IW], b = c[k * IW];
            const float p = a + b;
            t0 = t0 + t.g[k] * p;
            t1 = t1 + t.xg[k] * (b - a);
            t2 = t2 + t.xxg[k] * p;
        }
        float *o = row3 + (size_t)i * 3;
        o[0] = t0;
        o[1] = t1;
        o[2] = t2;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 16 * 64; i += 256) {  // polyexp_horiz at (Y0 + r, X0 + x): columns x +- k are tile columns x + n +- k
        const int r = i >> 6, x = i & 63, gy = Y0 + r, gx = X0 + x;
        if (gx >= w || gy >= h) continue;
        const float *c = row3 + ((size_t)r * IW + x + n) * 3;
        float g0 = t.g[0];
        double b1 = c[0] * g0, b2 = 0, b3 = c[1] * g0, b4 = 0, b5 = c[2] * g0, b6 = 0;
        for (int k = 1; k <= n; k++) {
            const float *p = c + 3 * k, *m = c - 3 * k;
            const double tg = p[0] + m[0];
            g0 = t.g[k];
            b1 += tg * g0;
            b4 += tg * t.xxg[k];
            b2 += (p[0] - m[0]) * t.xg[k];
            b3 += (p[1] + m[1]) * g0;
            b6 += (p[1] - m[1]) * t.xg[k];
            b5 += (p[2] + m[2]) * g0;
        }
        float *d = dst5 + ((size_t)gy * w + gx) * 5;
        d[1] = (float)(b2 * t.ig11);
        d[0] = (float)(b3 * t.ig11);
        d[3] = (float)(b1 * t.ig03 + b4 * t.ig33);
        d[2] = (float)(b1 * t.ig03 + b5 * t.ig33);
        d[4] = (float)(b6 * t.ig55);
    }
}

__global__ __launch_bounds__(256) void polyexp_fused_kernel(const float *__restrict__ src, int w, int h, PolyTaps t, float *__restrict__ dst5,
                                                            ptrdiff_t src_z, ptrdiff_t dst_z)
{
    extern __shared__ float p_lds[];
    polyexp_fused_body(src + (ptrdiff_t)blockIdx.z * src_z, w, h, t, dst5 + (ptrdiff_t)blockIdx.z * dst_z, blockIdx.x * 64, blockIdx.y * 16, p_lds);
}

__global__ __launch_bounds__(256) void polyexp_levels_kernel(const float *__restrict__ I, FbLevels L, PolyTaps t, float *__restrict__ R)
{
    extern __shared__ float p_lds[];
    const int lev = blockIdx.z / L.nimg, img = blockIdx.z - lev * L.nimg;
    const FbLevelDesc &d = L.lv[lev];
    const int X0 = blockIdx.x * 64, Y0 = blockIdx.y * 16;
    if (X0 >= d.w || Y0 >= d.h) return;   // (the grid is the finest level's)
    const size_t pk = (size_t)d.w * d.h;
    polyexp_fused_body(I + d.i_off + img * pk, d.w, d.h, t, R + d.r_off + img * 5 * pk, X0, Y0, p_lds);
}

// FarnebackUpdateMatrices at one pixel: the five products of the displaced polynomial coefficients, from the pixel's own flow
__device__ __forceinline__ void update_matrix_at(const float *__restrict__ R0, const float *__restrict__ R1, float dx, float dy,
                                                 int x, int y, int w, int h, float *__restrict__ m)
{
    const float border[5] = {0.14f, 0.14f, 0.4472f, 0.4472f, 0.4472f};
    const size_t step1 = (size_t)w * 5;
    const float *r0 = R0 + ((size_t)y * w + x) * 5;
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
    m[0] = r4 * r4 + r6 * r6;
    m[1] = (r4 + r5) * r6;
    m[2] = r5 * r5 + r6 * r6;
    m[3] = r4 * r2 + r6 * r3;
    m[4] = r6 * r2 + r5 * r3;
}

// update_matrix_at without a branch, into registers: the 2 x 2 footprint of R1 is loaded whatever the flow says (from a clamped address
// when it falls outside) and the two cases are selected afterwards -- the same operations on the same values, so the same bits; a thread
// that owns several pixels can have all their gathers in flight at once.
__device__ __forceinline__ void update_matrix_vals(const float *__restrict__ R0, const float *__restrict__ R1, float dx, float dy,
                                                   int x, int y, int w, int h, float m[5])
{
    const float border[5] = {0.14f, 0.14f, 0.4472f, 0.4472f, 0.4472f};
    const size_t step1 = (size_t)w * 5;
    const float *r0 = R0 + ((size_t)y * w + x) * 5;
    const float r00 = r0[0], r01 = r0[1], r02 = r0[2], r03 = r0[3], r04 = r0[4];
    float fx = x + dx, fy = y + dy;
    const int x1 = (int)floorf(fx), y1 = (int)floorf(fy);
    fx -= x1;
    fy -= y1;
    const bool inb = (unsigned)x1 < (unsigned)(w - 1) && (unsigned)y1 < (unsigned)(h - 1);
    const float *p = R1 + (size_t)(inb ? y1 : 0) * step1 + (size_t)(inb ? x1 : 0) * 5;
    const float a00 = (1.f - fx) * (1.f - fy), a01 = fx * (1.f - fy), a10 = (1.f - fx) * fy, a11 = fx * fy;
    float r2 = a00 * p[0] + a01 * p[5] + a10 * p[step1] + a11 * p[step1 + 5];
    float r3 = a00 * p[1] + a01 * p[6] + a10 * p[step1 + 1] + a11 * p[step1 + 6];
    float r4 = a00 * p[2] + a01 * p[7] + a10 * p[step1 + 2] + a11 * p[step1 + 7];
    float r5 = a00 * p[3] + a01 * p[8] + a10 * p[step1 + 3] + a11 * p[step1 + 8];
    float r6 = a00 * p[4] + a01 * p[9] + a10 * p[step1 + 4] + a11 * p[step1 + 9];
    r4 = inb ? (r02 + r4) * 0.5f : r02;
    r5 = inb ? (r03 + r5) * 0.5f : r03;
    r6 = inb ? (r04 + r6) * 0.25f : r04 * 0.5f;
    r2 = inb ? r2 : 0.f;
    r3 = inb ? r3 : 0.f;
    r2 = (r00 - r2) * 0.5f;
    r3 = (r01 - r3) * 0.5f;
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
    m[0] = r4 * r4 + r6 * r6;
    m[1] = (r4 + r5) * r6;
    m[2] = r5 * r5 + r6 * r6;
    m[3] = r4 * r2 + r6 * r3;
    m[4] = r6 * r2 + r5 * r3;
}

// (blockIdx.z = flow of a batch: R1, flow and M advance by their strides, R0 -- the common first frame -- does not)
__global__ __launch_bounds__(256) void update_matrices_kernel(const float *__restrict__ R0, const float *__restrict__ R1,
                                                              const float *__restrict__ flow, int w, int h,
                                                              float *__restrict__ M, ptrdiff_t r1_z = 0, ptrdiff_t flow_z = 0, ptrdiff_t m_z = 0)
{
    PIX2D
    R1 += r1_z * blockIdx.z;
    flow += flow_z * blockIdx.z;
    M += m_z * blockIdx.z;
    update_matrix_at(R0, R1, flow[((size_t)y * w + x) * 2], flow[((size_t)y * w + x) * 2 + 1], x, y, w, h,
                     M + ((size_t)y * w + x) * 5);
}

// The first two launches of a pyramid level in one (round 6): the flow carried down from the coarser level -- resize_linear_kernel<2> with
// its `*= 1 / pyrScale`, or zeros at the coarsest level (flags == 0: no initial flow) -- and FarnebackUpdateMatrices on it.  The matrices
// need the pixel's OWN flow only, so the value goes from the register into both: the same operations on the same values, two launches
// of ~5 us fewer on each of the 11 levels' critical path (and no memset node for a caller's graph to trip over).
__global__ __launch_bounds__(256) void upsample_update_kernel(const float *__restrict__ prevflow, int pw, int ph, float mul, const float *__restrict__ R0,
                                                              const float *__restrict__ R1, float *__restrict__ flow, int w, int h, float *__restrict__ M,
                                                              ptrdiff_t r1_z, ptrdiff_t flow_z, ptrdiff_t m_z)
{
    PIX2D
    R1 += r1_z * blockIdx.z;
    flow += flow_z * blockIdx.z;
    M += m_z * blockIdx.z;
    float v[2] = {0.f, 0.f};
    if (prevflow) {
        prevflow += flow_z * blockIdx.z;
        int sx, sy;
        float a0, a1, b0, b1;
        linear_coeff(x, w, pw, sx, a0, a1);
        linear_coeff(y, h, ph, sy, b0, b1);
        const int sx1 = sx + 1 < pw ? sx + 1 : sx, sy1 = sy + 1 < ph ? sy + 1 : sy;
#pragma unroll
        for (int c = 0; c < 2; c++) {
            const float r0 = prevflow[((size_t)sy * pw + sx) * 2 + c] * a0 + prevflow[((size_t)sy * pw + sx1) * 2 + c] * a1;
            const float r1 = prevflow[((size_t)sy1 * pw + sx) * 2 + c] * a0 + prevflow[((size_t)sy1 * pw + sx1) * 2 + c] * a1;
            v[c] = (r0 * b0 + r1 * b1) * mul;
        }
    }
    flow[((size_t)y * w + x) * 2] = v[0];
    flow[((size_t)y * w + x) * 2 + 1] = v[1];
    update_matrix_at(R0, R1, v[0], v[1], x, y, w, h, M + ((size_t)y * w + x) * 5);
}

__global__ __launch_bounds__(256) void box_vert_kernel(const float *__restrict__ M, int w, int h, int m,
                                                       double *__restrict__ vs)
{
    PIX2D
    double s[5] = {0, 0, 0, 0, 0};
    for (int d = -m; d <= m; d++) {
        const float *p = M + ((size_t)clampi(y + d, 0, h - 1) * w + x) * 5;
#pragma unroll
        for (int c = 0; c < 5; c++) s[c] += p[c];
    }
    double *o = vs + ((size_t)y * w + x) * 5;
#pragma unroll
    for (int c = 0; c < 5; c++) o[c] = s[c];
}

__global__ __launch_bounds__(256) void box_horiz_solve_kernel(const double *__restrict__ vs, int w, int h, int m,
                                                              double scale, float *__restrict__ flow)
{
    PIX2D
    double t[5] = {0, 0, 0, 0, 0};
    for (int d = -m; d <= m; d++) {
        const double *p = vs + ((size_t)y * w + clampi(x + d, 0, w - 1)) * 5;
#pragma unroll
        for (int c = 0; c < 5; c++) t[c] += p[c];
    }
#pragma unroll
    for (int c = 0; c < 5; c++) t[c] = t[c] * scale;
    const double idet = 1. / (t[0] * t[2] - t[1] * t[1] + 1e-3);
    flow[((size_t)y * w + x) * 2] = (float)((t[0] * t[4] - t[1] * t[3]) * idet);
    flow[((size_t)y * w + x) * 2 + 1] = (float)((t[2] * t[3] - t[1] * t[4]) * idet);
}

// One Farneback iteration in ONE launch (box_vert + box_horiz_solve + update_matrices).  A workgroup owns 64 x 4 pixels:
// it first forms the vertical box sums of M for its 64 + 2m columns (the columns box_horiz will address, clamped like
// it does) into LDS, then every pixel sums its 2m+1 neighbours horizontally, solves for the flow and -- because
// FarnebackUpdateMatrices needs only the pixel's OWN new flow -- writes the next M right away.  M is read with a halo by
// the neighbouring workgroups, so it ping-pongs between two buffers.  Same summation order as the separate kernels.
constexpr int FB_MAXM = 40;  // window radius the LDS buffer is sized for (winsize <= 81; 4K uses 60 -> m = 30)

__global__ __launch_bounds__(256) void farneback_iteration_fused(const float *__restrict__ M_in, const float *__restrict__ R0,
                                                                 const float *__restrict__ R1, int w, int h, int m, double scale,
                                                                 float *__restrict__ flow, float *__restrict__ M_out, ptrdiff_t m_z = 0, ptrdiff_t r1_z = 0,
                                                                 ptrdiff_t flow_z = 0)
{
    __shared__ double vsum[4 * (64 + 2 * FB_MAXM) * 5];
    M_in += m_z * blockIdx.z;   // blockIdx.z = flow of a batch (mvs_process_frame: every side view of a main frame in one launch)
    R1 += r1_z * blockIdx.z;
    flow += flow_z * blockIdx.z;
    if (M_out) M_out += m_z * blockIdx.z;
    const int X0 = blockIdx.x * 64, Y0 = blockIdx.y * 4, cols = 64 + 2 * m;
    for (int i = threadIdx.x; i < 4 * cols; i += 256) {
        const int r = i / cols, cx = i - r * cols, gy = Y0 + r;
        if (gy >= h) continue;
        const int gx = clampi(X0 - m + cx, 0, w - 1);
        double s[5] = {0, 0, 0, 0, 0};
        for (int d = -m; d <= m; d++) {
            const float *p = M_in + ((size_t)clampi(gy + d, 0, h - 1) * w + gx) * 5;
#pragma unroll
            for (int c = 0; c < 5; c++) s[c] += p[c];
        }
#pragma unroll
        for (int c = 0; c < 5; c++) vsum[(size_t)i * 5 + c] = s[c];
    }
    __syncthreads();
    const int lx = threadIdx.x & 63, r = threadIdx.x >> 6, x = X0 + lx, y = Y0 + r;
    if (x >= w || y >= h) return;
    double t[5] = {0, 0, 0, 0, 0};
    for (int d = -m; d <= m; d++) {
        // box_horiz addresses column clampi(x + d, 0, w - 1); LDS column k holds clampi(X0 - m + k, 0, w - 1)
        const double *p = vsum + ((size_t)r * cols + (lx + d + m)) * 5;
#pragma unroll
        for (int c = 0; c < 5; c++) t[c] += p[c];
    }
#pragma unroll
    for (int c = 0; c < 5; c++) t[c] = t[c] * scale;
    const double idet = 1. / (t[0] * t[2] - t[1] * t[1] + 1e-3);
    const float dx = (float)((t[0] * t[4] - t[1] * t[3]) * idet), dy = (float)((t[2] * t[3] - t[1] * t[4]) * idet);
    flow[((size_t)y * w + x) * 2] = dx;
    flow[((size_t)y * w + x) * 2 + 1] = dy;
    if (M_out) update_matrix_at(R0, R1, dx, dy, x, y, w, h, M_out + ((size_t)y * w + x) * 5);
}

// The same iteration with every value read ONCE per thread instead of once per sum (round 5).  The arithmetic is that of the kernels
// above, operation for operation: a vertical sum is 0 + M[y-m] + ... + M[y+m] in f64 in that order, a horizontal sum 0 + vs[x-m] + ...
// + vs[x+m] -- but a thread owns PY consecutive rows of one column (PX consecutive pixels of one row) and walks the 2m + PY (2m + PX)
// values they need once, adding each value to every running sum whose window holds it: output p takes the values j = p .. p + 2m, in
// ascending j, which IS its own fixed order.  What changes is the traffic: 2m + PY loads for PY vertical sums instead of PY (2m + 1), and
// (2m + PX) / PX LDS reads per horizontal sum instead of 2m + 1 (the kernel above was bound by exactly those: 910 bytes of loads per
// pixel and iteration at 1080p).  A workgroup owns 64 x TY pixels, TY = 2 PY = 4 PX; the vertical sums of its 64 + 2m columns live in LDS
// as vs[row][channel][column], the column index padded by one slot every PX columns so that the lanes of a wavefront -- PX columns apart
// -- hit different banks (stride PX + 1 doubles, odd).  Needs 2m >= PY - 1 (head and tail of the walk are unrolled); smaller windows,
// and the test hook MVS_FB_DIRECT_BOX, take the kernel above.
template <int N>
struct FbAcc {
    double s[N][5];
};

template <int PY, int PX, int THREADS>
__global__ __launch_bounds__(THREADS) void farneback_iteration_tiled(const float *__restrict__ M_in, const float *__restrict__ R0,
                                                                 const float *__restrict__ R1, int w, int h, int m, double scale,
                                                                 float *__restrict__ flow, float *__restrict__ M_out, ptrdiff_t m_z, ptrdiff_t r1_z,
                                                                 ptrdiff_t flow_z)
{
    constexpr int TY = PX * 4, NQ = TY / PY;  // rows of the tile (64 / PX threads per row, 256 threads in the horizontal phase), row groups of the vertical phase
    extern __shared__ double fb_vs[];  // [TY][5][colsP]
    M_in += m_z * blockIdx.z;
    R1 += r1_z * blockIdx.z;
    flow += flow_z * blockIdx.z;
    if (M_out) M_out += m_z * blockIdx.z;
    const int X0 = blockIdx.x * 64, Y0 = blockIdx.y * TY, cols = 64 + 2 * m, colsP = cols + cols / PX + 1;
    // ---- vertical sums: one (column, group of PY rows) per thread ----
    for (int item = threadIdx.x; item < NQ * cols; item += THREADS) {
        const int q = item / cols, k = item - q * cols;
        const int yb = Y0 + q * PY;
        if (yb >= h) continue;
        const int gx = clampi(X0 - m + k, 0, w - 1);
        const float *col = M_in + (size_t)gx * 5;
        FbAcc<PY> a;
#pragma unroll
        for (int p = 0; p < PY; p++)
#pragma unroll
            for (int c = 0; c < 5; c++) a.s[p][c] = 0.;
        auto row = [&](int j) { return col + (size_t)clampi(yb - m + j, 0, h - 1) * w * 5; };
#pragma unroll
        for (int j = 0; j < PY - 1; j++) {  // head: value j belongs to the windows of outputs 0 .. j
            const float *v = row(j);
            const float v0 = v[0], v1 = v[1], v2 = v[2], v3 = v[3], v4 = v[4];
#pragma unroll
            for (int p = 0; p <= j; p++) {
                a.s[p][0] += v0;
                a.s[p][1] += v1;
                a.s[p][2] += v2;
                a.s[p][3] += v3;
                a.s[p][4] += v4;
            }
        }
        // every output's window holds the values j = PY - 1 .. 2m: walked in batches of FB_U rows, the loads of batch b + 1 issued before
        // batch b is added (a workgroup's time is a chain of memory latencies, not arithmetic: the first form, one row per trip, ran
        // no faster than the kernel it replaced)
        {
            constexpr int U = 8;
            float cur[U][5], nxt[U][5];
            int j = PY - 1;
            const int end = 2 * m + 1;  // one past the last steady value
            auto fetch = [&](float (&dst)[U][5], int j0) {
#pragma unroll
                for (int u = 0; u < U; u++) {
                    const float *v = row(min(j0 + u, end - 1));  // (rows past the end are loaded again and not added)
#pragma unroll
                    for (int c = 0; c < 5; c++) dst[u][c] = v[c];
                }
            };
            fetch(cur, j);
            for (; j < end; j += U) {
                if (j + U < end) fetch(nxt, j + U);
#pragma unroll
                for (int u = 0; u < U; u++) {
                    if (j + u < end) {
#pragma unroll
                        for (int p = 0; p < PY; p++)
#pragma unroll
                            for (int c = 0; c < 5; c++) a.s[p][c] += cur[u][c];
                    }
                }
#pragma unroll
                for (int u = 0; u < U; u++)
#pragma unroll
                    for (int c = 0; c < 5; c++) cur[u][c] = nxt[u][c];
            }
        }
#pragma unroll
        for (int t = 0; t < PY - 1; t++) {  // tail: value 2m + 1 + t belongs to outputs t + 1 .. PY - 1
            const float *v = row(2 * m + 1 + t);
            const float v0 = v[0], v1 = v[1], v2 = v[2], v3 = v[3], v4 = v[4];
#pragma unroll
            for (int p = t + 1; p < PY; p++) {
                a.s[p][0] += v0;
                a.s[p][1] += v1;
                a.s[p][2] += v2;
                a.s[p][3] += v3;
                a.s[p][4] += v4;
            }
        }
        const int kp = k + k / PX;
#pragma unroll
        for (int p = 0; p < PY; p++)
#pragma unroll
            for (int c = 0; c < 5; c++) fb_vs[((size_t)(q * PY + p) * 5 + c) * colsP + kp] = a.s[p][c];
    }
    __syncthreads();
    // ---- horizontal sums, solve, next M: PX consecutive pixels of one row per thread ----
    constexpr int GPR = 64 / PX;  // threads per row
    if (threadIdx.x >= 256) return;  // (a 512-thread workgroup: twice the wavefronts for the vertical walk, the latency-bound half)
    const int r = threadIdx.x / GPR, g = threadIdx.x - r * GPR, lx0 = g * PX, y = Y0 + r;
    if (y >= h || X0 + lx0 >= w) return;
    FbAcc<PX> t;
#pragma unroll
    for (int p = 0; p < PX; p++)
#pragma unroll
        for (int c = 0; c < 5; c++) t.s[p][c] = 0.;
    // box_horiz addresses column clampi(x + d, 0, w - 1); LDS column k holds clampi(X0 - m + k, 0, w - 1): value j of this thread's walk is
    // column lx0 + j (padded index lx0 + j + g + j / PX)
    const double *base = fb_vs + (size_t)r * 5 * colsP + lx0 + g;
    auto val = [&](int j, int c) { return base[(size_t)c * colsP + j + j / PX]; };
#pragma unroll
    for (int j = 0; j < PX - 1; j++) {
#pragma unroll
        for (int c = 0; c < 5; c++) {
            const double v = val(j, c);
#pragma unroll
            for (int p = 0; p <= j; p++) t.s[p][c] += v;
        }
    }
#pragma unroll 4
    for (int j = PX - 1; j <= 2 * m; j++) {
#pragma unroll
        for (int c = 0; c < 5; c++) {
            const double v = val(j, c);
#pragma unroll
            for (int p = 0; p < PX; p++) t.s[p][c] += v;
        }
    }
#pragma unroll
    for (int u = 0; u < PX - 1; u++) {
#pragma unroll
        for (int c = 0; c < 5; c++) {
            const double v = val(2 * m + 1 + u, c);
#pragma unroll
            for (int p = u + 1; p < PX; p++) t.s[p][c] += v;
        }
    }
    float dxs[PX], dys[PX];
#pragma unroll
    for (int p = 0; p < PX; p++) {
        double tt[5];
#pragma unroll
        for (int c = 0; c < 5; c++) tt[c] = t.s[p][c] * scale;
        const double idet = 1. / (tt[0] * tt[2] - tt[1] * tt[1] + 1e-3);
        dxs[p] = (float)((tt[0] * tt[4] - tt[1] * tt[3]) * idet);
        dys[p] = (float)((tt[2] * tt[3] - tt[1] * tt[4]) * idet);
    }
#pragma unroll
    for (int p = 0; p < PX; p++) {
        const int x = X0 + lx0 + p;
        if (x < w) {
            flow[((size_t)y * w + x) * 2] = dxs[p];
            flow[((size_t)y * w + x) * 2 + 1] = dys[p];
        }
    }
    if (!M_out) return;
    // the next M of the thread's PX pixels: all gathers issued before any is used (pixels past the right edge compute on the last column
    // and store nothing)
    float mv[PX][5];
#pragma unroll
    for (int p = 0; p < PX; p++) update_matrix_vals(R0, R1, dxs[p], dys[p], min(X0 + lx0 + p, w - 1), y, w, h, mv[p]);
#pragma unroll
    for (int p = 0; p < PX; p++) {
        const int x = X0 + lx0 + p;
        if (x < w) {
            float *o = M_out + ((size_t)y * w + x) * 5;
#pragma unroll
            for (int c = 0; c < 5; c++) o[c] = mv[p][c];
        }
    }
}

// ---- variational refinement -------------------------------------------------------------------------------------

__global__ __launch_bounds__(256) void split_flow_kernel(const float *__restrict__ flow, float *__restrict__ u,
                                                         float *__restrict__ v, float *__restrict__ du,
                                                         float *__restrict__ dv, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u[i] = flow[2 * i];
    v[i] = flow[2 * i + 1];
    du[i] = 0.f;
    dv[i] = 0.f;
}

__global__ __launch_bounds__(256) void warp_q5_kernel(const float *__restrict__ img, int w, int h,
                                                      const float *__restrict__ u, const float *__restrict__ v,
                                                      const float *__restrict__ I0, float *__restrict__ A,
                                                      float *__restrict__ Iz)
{
    PIX2D
    const size_t p = (size_t)y * w + x;
    const float mx = (float)x + u[p], my = (float)y + v[p];
    const int qx = __float2int_rn(mx * 32.0f), qy = __float2int_rn(my * 32.0f);
    const int sx = qx >> 5, sy = qy >> 5;
    const float fx = (float)(qx & 31) * (1.0f / 32), fy = (float)(qy & 31) * (1.0f / 32);
    const float w0 = (1.f - fy) * (1.f - fx), w1 = (1.f - fy) * fx, w2 = fy * (1.f - fx), w3 = fy * fx;
    const int x0 = clampi(sx, 0, w - 1), x1 = clampi(sx + 1, 0, w - 1), y0 = clampi(sy, 0, h - 1), y1 = clampi(sy + 1, 0, h - 1);
    const float warped = img[(size_t)y0 * w + x0] * w0 + img[(size_t)y0 * w + x1] * w1 + img[(size_t)y1 * w + x0] * w2 +
                         img[(size_t)y1 * w + x1] * w3;
    A[p] = (I0[p] + warped) * 0.5f;
    Iz[p] = warped - I0[p];
}

template <bool DY>
__global__ __launch_bounds__(256) void diff_kernel(const float *__restrict__ a, int w, int h, float *__restrict__ o)
{
    PIX2D
    if (DY)
        o[(size_t)y * w + x] = a[(size_t)clampi(y + 1, 0, h - 1) * w + x] - a[(size_t)clampi(y - 1, 0, h - 1) * w + x];
    else
        o[(size_t)y * w + x] = a[(size_t)y * w + clampi(x + 1, 0, w - 1)] - a[(size_t)y * w + clampi(x - 1, 0, w - 1)];
}

struct VarBufs {
    float *Wu, *Wv, *du, *dv, *Iz, *Ix, *Iy, *Ixx, *Ixy, *Iyy, *Ixz, *Iyz, *a11, *a12, *a22, *b1, *b2, *wgt;
};

// All seven derivative images of the refinement in ONE launch (round 6; seven diff_kernel launches before -- a 640 x 480 refinement is launches, not
// bytes, and mvs_process_frame runs four of them per main frame): Ix, Iy of A; Ixz, Iyz of Iz; Ixx, Ixy of Ix; Iyy of Iy.  A second derivative needs the
// first one at the pixel's neighbours; each of those is ONE subtraction of two values of A, so it is formed again here from A -- the same operation on the
// same operands as diff_kernel's stored value: the same bits -- instead of waiting for a kernel that stores it.
__global__ __launch_bounds__(256) void var_derivatives_kernel(const float *__restrict__ A, const float *__restrict__ Iz, int w, int h, VarBufs B)
{
    PIX2D
    auto cx = [&](int v) { return clampi(v, 0, w - 1); };
    auto cy = [&](int v) { return clampi(v, 0, h - 1); };
    auto ddx = [&](const float *__restrict__ a, int yy, int xx) { return a[(size_t)yy * w + cx(xx + 1)] - a[(size_t)yy * w + cx(xx - 1)]; };   // diff_kernel<false> at (yy, xx)
    auto ddy = [&](const float *__restrict__ a, int yy, int xx) { return a[(size_t)cy(yy + 1) * w + xx] - a[(size_t)cy(yy - 1) * w + xx]; };   // diff_kernel<true> at (yy, xx)
    const size_t p = (size_t)y * w + x;
    B.Ix[p] = ddx(A, y, x);
    B.Iy[p] = ddy(A, y, x);
    B.Ixz[p] = ddx(Iz, y, x);
    B.Iyz[p] = ddy(Iz, y, x);
    B.Ixx[p] = ddx(A, y, cx(x + 1)) - ddx(A, y, cx(x - 1));   // diff_kernel<false>(Ix)
    B.Ixy[p] = ddx(A, cy(y + 1), x) - ddx(A, cy(y - 1), x);   // diff_kernel<true>(Ix)
    B.Iyy[p] = ddy(A, cy(y + 1), x) - ddy(A, cy(y - 1), x);   // diff_kernel<true>(Iy)
}

__global__ __launch_bounds__(256) void var_data_term(VarBufs B, int w, int h)
{
    PIX2D
    const size_t p = (size_t)y * w + x;
    const float zeta2 = 0.1f * 0.1f, eps2 = 0.001f * 0.001f, gamma2 = 10.f / 2, delta2 = 5.f / 2;
    const float Ix = B.Ix[p], Iy = B.Iy[p], Iz = B.Iz[p], Ixx = B.Ixx[p], Ixy = B.Ixy[p], Iyy = B.Iyy[p], Ixz = B.Ixz[p],
                Iyz = B.Iyz[p], du = B.du[p], dv = B.dv[p];
    float derivNorm = Ix * Ix + Iy * Iy + zeta2;
    const float Ik1z = Iz + Ix * du + Iy * dv;
    float weight = (delta2 / sqrtf(Ik1z * Ik1z / derivNorm + eps2)) / derivNorm;
    float A11 = weight * (Ix * Ix) + zeta2;
    float A12 = weight * (Ix * Iy);
    float A22 = weight * (Iy * Iy) + zeta2;
    float B1 = -weight * (Iz * Ix);
    float B2 = -weight * (Iz * Iy);
    derivNorm = Ixx * Ixx + Ixy * Ixy + zeta2;
    const float derivNorm2 = Iyy * Iyy + Ixy * Ixy + zeta2;
    const float Ik1zx = Ixz + Ixx * du + Ixy * dv;
    const float Ik1zy = Iyz + Ixy * du + Iyy * dv;
    weight = gamma2 / sqrtf(Ik1zx * Ik1zx / derivNorm + Ik1zy * Ik1zy / derivNorm2 + eps2);
    A11 += weight * (Ixx * Ixx / derivNorm + Ixy * Ixy / derivNorm2);
    A12 += weight * (Ixx * Ixy / derivNorm + Ixy * Iyy / derivNorm2);
    A22 += weight * (Ixy * Ixy / derivNorm + Iyy * Iyy / derivNorm2);
    B1 += -weight * (Ixx * Ixz / derivNorm + Ixy * Iyz / derivNorm2);
    B2 += -weight * (Ixy * Ixz / derivNorm + Iyy * Iyz / derivNorm2);
    B.a11[p] = A11;
    B.a12[p] = A12;
    B.a22[p] = A22;
    B.b1[p] = B1;
    B.b2[p] = B2;
}

__global__ __launch_bounds__(256) void var_diffusivity(VarBufs B, int w, int h)
{
    PIX2D
    const size_t p = (size_t)y * w + x;
    const float eps2 = 0.001f * 0.001f, alpha2 = 20.f / 2;
    const float cu = B.Wu[p] + B.du[p], cv = B.Wv[p] + B.dv[p];
    const float ux = x + 1 < w ? (B.Wu[p + 1] + B.du[p + 1]) - cu : 0.f, vx = x + 1 < w ? (B.Wv[p + 1] + B.dv[p + 1]) - cv : 0.f;
    const float uy = y + 1 < h ? (B.Wu[p + w] + B.du[p + w]) - cu : 0.f, vy = y + 1 < h ? (B.Wv[p + w] + B.dv[p + w]) - cv : 0.f;
    B.wgt[p] = alpha2 / sqrtf(ux * ux + vx * vx + uy * uy + vy * vy + eps2);
}

__global__ __launch_bounds__(256) void var_smooth_gather(VarBufs B, int w, int h)
{
    PIX2D
    const size_t p = (size_t)y * w + x;
    float A11 = B.a11[p], A22 = B.a22[p], B1 = B.b1[p], B2 = B.b2[p];
    if (x > 0) {
        const float wt = B.wgt[p - 1];
        B1 -= wt * (B.Wu[p] - B.Wu[p - 1]);
        B2 -= wt * (B.Wv[p] - B.Wv[p - 1]);
        A11 += wt;
        A22 += wt;
    }
    if (x + 1 < w) {
        const float wt = B.wgt[p];
        B1 += wt * (B.Wu[p + 1] - B.Wu[p]);
        B2 += wt * (B.Wv[p + 1] - B.Wv[p]);
        A11 += wt;
        A22 += wt;
    }
    if (y > 0) {
        const float wt = B.wgt[p - w];
        B1 -= wt * (B.Wu[p] - B.Wu[p - w]);
        B2 -= wt * (B.Wv[p] - B.Wv[p - w]);
        A11 += wt;
        A22 += wt;
    }
    if (y + 1 < h) {
        const float wt = B.wgt[p];
        B1 += wt * (B.Wu[p + w] - B.Wu[p]);
        B2 += wt * (B.Wv[p + w] - B.Wv[p]);
        A11 += wt;
        A22 += wt;
    }
    B.a11[p] = A11;
    B.a22[p] = A22;
    B.b1[p] = B1;
    B.b2[p] = B2;
}

// one colour of a red-black SOR sweep: thread per pixel of that colour
__global__ __launch_bounds__(256) void var_sor_pass(VarBufs B, int w, int h, int colour)
{
    const int xi = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int x = 2 * xi + ((y + colour) & 1);
    if (x >= w || y >= h) return;
    const size_t p = (size_t)y * w + x;
    const float omega = 1.6f;
    float sU = 0.f, sV = 0.f;
    if (x > 0) {
        sU += B.wgt[p - 1] * B.du[p - 1];
        sV += B.wgt[p - 1] * B.dv[p - 1];
    }
    if (x + 1 < w) {
        sU += B.wgt[p] * B.du[p + 1];
        sV += B.wgt[p] * B.dv[p + 1];
    }
    if (y > 0) {
        sU += B.wgt[p - w] * B.du[p - w];
        sV += B.wgt[p - w] * B.dv[p - w];
    }
    if (y + 1 < h) {
        sU += B.wgt[p] * B.du[p + w];
        sV += B.wgt[p] * B.dv[p + w];
    }
    float du = B.du[p], dv = B.dv[p];
    du += omega * ((sU + B.b1[p] - dv * B.a12[p]) / B.a11[p] - du);
    dv += omega * ((sV + B.b2[p] - du * B.a12[p]) / B.a22[p] - dv);
    B.du[p] = du;
    B.dv[p] = dv;
}

// One fixed-point iteration of the refinement in ONE launch: data term, diffusivity, smoothness gather and the 5
// red-black SOR sweeps (10 half-steps), which the kernels above run as 13 launches of a few microseconds each.
// A 1024-thread workgroup owns a VT_W x VT_H core and works on the core plus a VT_HALO-cell halo held in LDS
// (temporal blocking): a half-step only reads the four neighbours, so after half-step s every cell at least s+1 cells
// inside the region edge holds exactly the value the global sweep would have produced -- with a halo of 10 the core
// is exact after all 10 half-steps (image borders do not shrink the valid set: there is no neighbour to be wrong).
// Every cell is evaluated with the expressions of the kernels above in the same order, so the result is bit-identical
// (tests/test_flow_gpu.py compares both forms).  du/dv ping-pong between two buffers because neighbouring workgroups
// read each other's halo at the start.
//
// Thread mapping (round 6, second form): a thread owns SLOTS, a slot = two horizontally adjacent cells (2j, 2j + 1) of a region
// row -- one red, one black -- and keeps both cells' coefficients in registers under their COLOUR, so in a half-step every
// lane of a wavefront updates a cell (the first form gave a thread single cells in raster order: half the lanes of every
// wavefront sat out each half-step, and the 4368 cells of its 84 x 52 region made a ragged fifth pass that the barrier
// made everybody wait for).  84 x 48 cells = 2016 slots: two per thread, no ragged pass.  In LDS the two colours live in
// separate planes -- [colour][row][x >> 1] -- so consecutive lanes read consecutive words; the neighbours of a cell are all
// in the other plane: left / right at j - 1 + e and j + e (e = parity of row + colour), up / down at j of the rows above / below.
constexpr int VT_W = 64, VT_H = 28, VT_HALO = 10, VT_RW = VT_W + 2 * VT_HALO, VT_RH = VT_H + 2 * VT_HALO,
              VT_CELLS = VT_RW * VT_RH, VT_HW = VT_RW / 2, VT_SLOTS = VT_CELLS / 2, VT_THREADS = 1024,
              VT_PER = (VT_SLOTS + VT_THREADS - 1) / VT_THREADS, VT_LOADS = (VT_CELLS + VT_THREADS - 1) / VT_THREADS,
              VT_PLANE = VT_SLOTS + 16,   // (+ 16 words: the two planes' banks interleave when raster-ordered lanes fill them)
              VT_LDS = 2 * VT_PLANE;
static_assert(VT_W % 2 == 0 && VT_H % 2 == 0 && VT_HALO % 2 == 0, "a region's origin must be an even cell: a cell's colour is the parity of its region coordinates");

__device__ __forceinline__ int vt_idx(int ry, int rx) { return ((rx + ry) & 1) * VT_PLANE + ry * VT_HW + (rx >> 1); }

__global__ __launch_bounds__(VT_THREADS) void var_fixed_point_fused(VarBufs B, const float *__restrict__ du_in,
                                                                      const float *__restrict__ dv_in,
                                                                      float *__restrict__ du_out,
                                                                      float *__restrict__ dv_out, int w, int h, int sor_iters)
{
    __shared__ float s_du[VT_LDS], s_dv[VT_LDS], s_wu[VT_LDS], s_wv[VT_LDS], s_wgt[VT_LDS];
    const int X0 = blockIdx.x * VT_W - VT_HALO, Y0 = blockIdx.y * VT_H - VT_HALO;   // both even
    const float omega = 1.6f;

    // per (slot, colour): the cell's coefficients, the half-steps it can take part in (-1: outside the image) and which neighbours the image has
    float a11[VT_PER][2], a12[VT_PER][2], a22[VT_PER][2], b1[VT_PER][2], b2[VT_PER][2];
    int lim[VT_PER][2];
    unsigned nb[VT_PER][2];   // bit 0..3: has left / right / up / down neighbour in the image
    int own[VT_PER][2];       // the cell's word in its plane's arrays

    // phase 0: region -> LDS (raster order: coalesced reads)
#pragma unroll
    for (int k = 0; k < VT_LOADS; k++) {
        const int c = threadIdx.x + k * VT_THREADS;
        if (c < VT_CELLS) {
            const int ry = c / VT_RW, rx = c - ry * VT_RW, x = X0 + rx, y = Y0 + ry, i = vt_idx(ry, rx);
            const bool in = x >= 0 && x < w && y >= 0 && y < h;
            const size_t p = in ? (size_t)y * w + x : 0;
            s_du[i] = in ? du_in[p] : 0.f;
            s_dv[i] = in ? dv_in[p] : 0.f;
            s_wu[i] = in ? B.Wu[p] : 0.f;
            s_wv[i] = in ? B.Wv[p] : 0.f;
        }
    }
    __syncthreads();

    // phase 1: data term (pointwise) and diffusivity (right / down neighbours)
#pragma unroll
    for (int k = 0; k < VT_PER; k++) {
        const int q = threadIdx.x + k * VT_THREADS;
        const int ry = q / VT_HW, j = q - ry * VT_HW;
#pragma unroll
        for (int col = 0; col < 2; col++) {
            lim[k][col] = -1;
            nb[k][col] = 0;
            own[k][col] = 0;
            a11[k][col] = a12[k][col] = a22[k][col] = 1.f;
            b1[k][col] = b2[k][col] = 0.f;
            if (q >= VT_SLOTS) continue;
            const int rx = 2 * j + ((ry + col) & 1), x = X0 + rx, y = Y0 + ry, c = vt_idx(ry, rx);
            own[k][col] = c;
            if (!(x >= 0 && x < w && y >= 0 && y < h)) {
                s_wgt[c] = 0.f;
                continue;
            }
            const size_t p = (size_t)y * w + x;
            // distance to each region side that is a cut through the image (a side at/after the image border is no cut)
            const int dl = X0 > 0 ? rx : 1 << 20, dr = X0 + VT_RW < w ? VT_RW - 1 - rx : 1 << 20;
            const int du_ = Y0 > 0 ? ry : 1 << 20, dd = Y0 + VT_RH < h ? VT_RH - 1 - ry : 1 << 20;
            lim[k][col] = min(min(dl, dr), min(du_, dd));
            nb[k][col] = (x > 0 ? 1u : 0u) | (x + 1 < w ? 2u : 0u) | (y > 0 ? 4u : 0u) | (y + 1 < h ? 8u : 0u);
            {
                const float zeta2 = 0.1f * 0.1f, eps2 = 0.001f * 0.001f, gamma2 = 10.f / 2, delta2 = 5.f / 2;
                const float Ix = B.Ix[p], Iy = B.Iy[p], Iz = B.Iz[p], Ixx = B.Ixx[p], Ixy = B.Ixy[p], Iyy = B.Iyy[p],
                            Ixz = B.Ixz[p], Iyz = B.Iyz[p], du = s_du[c], dv = s_dv[c];
                float derivNorm = Ix * Ix + Iy * Iy + zeta2;
                const float Ik1z = Iz + Ix * du + Iy * dv;
                float weight = (delta2 / sqrtf(Ik1z * Ik1z / derivNorm + eps2)) / derivNorm;
                float A11 = weight * (Ix * Ix) + zeta2;
                float A12 = weight * (Ix * Iy);
                float A22 = weight * (Iy * Iy) + zeta2;
                float B1 = -weight * (Iz * Ix);
                float B2 = -weight * (Iz * Iy);
                derivNorm = Ixx * Ixx + Ixy * Ixy + zeta2;
                const float derivNorm2 = Iyy * Iyy + Ixy * Ixy + zeta2;
                const float Ik1zx = Ixz + Ixx * du + Ixy * dv;
                const float Ik1zy = Iyz + Ixy * du + Iyy * dv;
                weight = gamma2 / sqrtf(Ik1zx * Ik1zx / derivNorm + Ik1zy * Ik1zy / derivNorm2 + eps2);
                A11 += weight * (Ixx * Ixx / derivNorm + Ixy * Ixy / derivNorm2);
                A12 += weight * (Ixx * Ixy / derivNorm + Ixy * Iyy / derivNorm2);
                A22 += weight * (Ixy * Ixy / derivNorm + Iyy * Iyy / derivNorm2);
                B1 += -weight * (Ixx * Ixz / derivNorm + Ixy * Iyz / derivNorm2);
                B2 += -weight * (Ixy * Ixz / derivNorm + Iyy * Iyz / derivNorm2);
                a11[k][col] = A11;
                a12[k][col] = A12;
                a22[k][col] = A22;
                b1[k][col] = B1;
                b2[k][col] = B2;
            }
            {
                const float eps2 = 0.001f * 0.001f, alpha2 = 20.f / 2;
                const int cr = rx + 1 < VT_RW ? vt_idx(ry, rx + 1) : c, cd = ry + 1 < VT_RH ? vt_idx(ry + 1, rx) : c;  // clamped: such cells have lim 0
                const float cu = s_wu[c] + s_du[c], cv = s_wv[c] + s_dv[c];
                const float ux = (nb[k][col] & 2u) ? (s_wu[cr] + s_du[cr]) - cu : 0.f, vx = (nb[k][col] & 2u) ? (s_wv[cr] + s_dv[cr]) - cv : 0.f;
                const float uy = (nb[k][col] & 8u) ? (s_wu[cd] + s_du[cd]) - cu : 0.f, vy = (nb[k][col] & 8u) ? (s_wv[cd] + s_dv[cd]) - cv : 0.f;
                s_wgt[c] = alpha2 / sqrtf(ux * ux + vx * vx + uy * uy + vy * vy + eps2);
            }
        }
    }
    __syncthreads();

    // phase 2: smoothness gather -> final coefficients; neighbour weights stay in registers for the sweeps
    float wl[VT_PER][2], wr[VT_PER][2], wt_[VT_PER][2];  // wgt[p-1], wgt[p] (right and down), wgt[p-w]
#pragma unroll
    for (int k = 0; k < VT_PER; k++) {
        const int q = threadIdx.x + k * VT_THREADS;
        const int ry = q / VT_HW, j = q - ry * VT_HW;
#pragma unroll
        for (int col = 0; col < 2; col++) {
            wl[k][col] = wr[k][col] = wt_[k][col] = 0.f;
            if (q >= VT_SLOTS || lim[k][col] < 0) continue;
            const int rx = 2 * j + ((ry + col) & 1), c = own[k][col];
            const int cl = rx > 0 ? vt_idx(ry, rx - 1) : c, cr = rx + 1 < VT_RW ? vt_idx(ry, rx + 1) : c, cu = ry > 0 ? vt_idx(ry - 1, rx) : c,
                      cd = ry + 1 < VT_RH ? vt_idx(ry + 1, rx) : c;
            float A11 = a11[k][col], A22 = a22[k][col], B1 = b1[k][col], B2 = b2[k][col];
            const unsigned n = nb[k][col];
            wl[k][col] = s_wgt[cl];
            wr[k][col] = s_wgt[c];
            wt_[k][col] = s_wgt[cu];
            if (n & 1u) {
                const float wt = wl[k][col];
                B1 -= wt * (s_wu[c] - s_wu[cl]);
                B2 -= wt * (s_wv[c] - s_wv[cl]);
                A11 += wt;
                A22 += wt;
            }
            if (n & 2u) {
                const float wt = wr[k][col];
                B1 += wt * (s_wu[cr] - s_wu[c]);
                B2 += wt * (s_wv[cr] - s_wv[c]);
                A11 += wt;
                A22 += wt;
            }
            if (n & 4u) {
                const float wt = wt_[k][col];
                B1 -= wt * (s_wu[c] - s_wu[cu]);
                B2 -= wt * (s_wv[c] - s_wv[cu]);
                A11 += wt;
                A22 += wt;
            }
            if (n & 8u) {
                const float wt = wr[k][col];
                B1 += wt * (s_wu[cd] - s_wu[c]);
                B2 += wt * (s_wv[cd] - s_wv[c]);
                A11 += wt;
                A22 += wt;
            }
            a11[k][col] = A11;
            a22[k][col] = A22;
            b1[k][col] = B1;
            b2[k][col] = B2;
        }
    }
    // (no barrier needed: the sweeps below read s_du / s_dv only, which nobody has written since phase 0)

    // phase 3: red-black SOR, colour 0 then colour 1 per iteration.  A cell that takes part (lim > s >= 0) is at least one cell inside the region, so
    // its neighbours' words exist; they are in the other colour's plane: row * VT_HW + j is the cell's word within a plane
    for (int it = 0; it < sor_iters; it++) {
#pragma unroll
        for (int col = 0; col < 2; col++) {
            const int s = 2 * it + col;
#pragma unroll
            for (int k = 0; k < VT_PER; k++) {
                if (lim[k][col] <= s) continue;  // lim >= s+1: all inputs still exact
                const int c = own[k][col], o = c + (1 - 2 * col) * VT_PLANE;   // the same word in the other plane
                const int q = threadIdx.x + k * VT_THREADS, e = ((q / VT_HW) + col) & 1;
                const unsigned n = nb[k][col];
                float sU = 0.f, sV = 0.f;
                if (n & 1u) {
                    sU += wl[k][col] * s_du[o - 1 + e];
                    sV += wl[k][col] * s_dv[o - 1 + e];
                }
                if (n & 2u) {
                    sU += wr[k][col] * s_du[o + e];
                    sV += wr[k][col] * s_dv[o + e];
                }
                if (n & 4u) {
                    sU += wt_[k][col] * s_du[o - VT_HW];
                    sV += wt_[k][col] * s_dv[o - VT_HW];
                }
                if (n & 8u) {
                    sU += wr[k][col] * s_du[o + VT_HW];
                    sV += wr[k][col] * s_dv[o + VT_HW];
                }
                float du = s_du[c], dv = s_dv[c];
                du += omega * ((sU + b1[k][col] - dv * a12[k][col]) / a11[k][col] - du);
                dv += omega * ((sV + b2[k][col] - du * a12[k][col]) / a22[k][col] - dv);
                s_du[c] = du;
                s_dv[c] = dv;
            }
            __syncthreads();
        }
    }

    // phase 4: core -> global
#pragma unroll
    for (int k = 0; k < VT_PER; k++) {
        const int q = threadIdx.x + k * VT_THREADS;
        const int ry = q / VT_HW, j = q - ry * VT_HW;
#pragma unroll
        for (int col = 0; col < 2; col++) {
            if (q >= VT_SLOTS || lim[k][col] < 0) continue;
            const int rx = 2 * j + ((ry + col) & 1);
            if (rx < VT_HALO || rx >= VT_HALO + VT_W || ry < VT_HALO || ry >= VT_HALO + VT_H) continue;
            const size_t p = (size_t)(Y0 + ry) * w + (X0 + rx);
            du_out[p] = s_du[own[k][col]];
            dv_out[p] = s_dv[own[k][col]];
        }
    }
}

__global__ __launch_bounds__(256) void var_finish(VarBufs B, float *__restrict__ flow, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    flow[2 * i] = B.Wu[i] + B.du[i];
    flow[2 * i + 1] = B.Wv[i] + B.dv[i];
}

__global__ __launch_bounds__(256) void pack_flow4(const float *__restrict__ flow2, const float *__restrict__ var,
                                                  float *__restrict__ out4, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    flow2 += 2 * n * blockIdx.y;  // blockIdx.y = flow of a batch (tightly packed)
    var += n * blockIdx.y;
    out4 += 4 * n * blockIdx.y;
    out4[4 * i] = flow2[2 * i];
    out4[4 * i + 1] = flow2[2 * i + 1];
    out4[4 * i + 2] = var[i];
    out4[4 * i + 3] = 0.f;  // mixChannels {-1, 3}, flow.cpp:39
}

// ---- host orchestration ------------------------------------------------------------------------------------------

static void gaussian_taps(int n, double sigma, float *k)  // cv::getGaussianKernel(n, sigma, CV_32F)
{
    if (sigma <= 0 && n == 3) {
        k[0] = 0.25f; k[1] = 0.5f; k[2] = 0.25f;
        return;
    }
    if (sigma <= 0 && n == 5) {
        const float t[5] = {0.0625f, 0.25f, 0.375f, 0.25f, 0.0625f};
        memcpy(k, t, sizeof(t));
        return;
    }
    if (sigma <= 0 && n == 7) {
        const float t[7] = {0.03125f, 0.109375f, 0.21875f, 0.28125f, 0.21875f, 0.109375f, 0.03125f};
        memcpy(k, t, sizeof(t));
        return;
    }
    const double sigmaX = sigma > 0 ? sigma : ((n - 1) * 0.5 - 1) * 0.3 + 0.8;
    const double scale2X = -0.5 / (sigmaX * sigmaX);
    double sum = 0;
    for (int i = 0; i < n; i++) {
        const double x = i - (n - 1) * 0.5;
        k[i] = (float)std::exp(scale2X * x * x);
        sum += k[i];
    }
    sum = 1. / sum;
    for (int i = 0; i < n; i++) k[i] = (float)(k[i] * sum);
}

static void farneback_taps(int n, double sigma, PolyTaps &t)  // FarnebackPrepareGaussian
{
    std::vector<float> g(2 * n + 1), xg(2 * n + 1), xxg(2 * n + 1);
    if (sigma < 1.1920929e-07) sigma = n * 0.3;
    double s = 0.;
    for (int x = -n; x <= n; x++) {
        g[x + n] = (float)std::exp(-x * x / (2 * sigma * sigma));
        s += g[x + n];
    }
    s = 1. / s;
    for (int x = -n; x <= n; x++) {
        g[x + n] = (float)(g[x + n] * s);
        xg[x + n] = (float)(x * g[x + n]);
        xxg[x + n] = (float)(x * x * g[x + n]);
    }
    double G[6][6] = {};
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
    double A[6][12];
    for (int i = 0; i < 6; i++)
        for (int j = 0; j < 6; j++) {
            A[i][j] = G[i][j];
            A[i][j + 6] = i == j;
        }
    for (int c = 0; c < 6; c++) {  // Gauss-Jordan, partial pivoting (same elimination order as the oracle)
        int p = c;
        for (int r = c + 1; r < 6; r++)
            if (std::fabs(A[r][c]) > std::fabs(A[p][c])) p = r;
        if (p != c)
            for (int j = 0; j < 12; j++) std::swap(A[c][j], A[p][j]);
        const double d = 1. / A[c][c];
        for (int j = 0; j < 12; j++) A[c][j] *= d;
        for (int r = 0; r < 6; r++)
            if (r != c) {
                const double f = A[r][c];
                if (f != 0)
                    for (int j = 0; j < 12; j++) A[r][j] -= f * A[c][j];
            }
    }
    t.n = n;
    t.ig11 = A[1][7];
    t.ig03 = A[0][9];
    t.ig33 = A[3][9];
    t.ig55 = A[5][11];
    for (int k = 0; k <= n; k++) {
        t.g[k] = g[n + k];
        t.xg[k] = xg[n + k];
        t.xxg[k] = xxg[n + k];
    }
}

static dim3 g2(int w, int h) { return dim3(div_up(w, 64), div_up(h, 4)); }
static unsigned g1(size_t n) { return (unsigned)((n + 255) / 256); }

// one fused Farneback iteration over B flows (blockIdx.z): the tiled kernel where the window allows it, tall tiles where the image is large
// enough to fill the chip with them
static int launch_fb_iteration(mvs_ctx *ctx, const float *M_in, const float *R0, const float *R1, int w, int h, int m, double scale, float *flow, float *M_out, int B,
                               ptrdiff_t m_z, ptrdiff_t r1_z, ptrdiff_t flow_z)
{
    hipStream_t st = ctx->stream;
    if (m < 4 || ctx->hooks.fb_direct_box) {
        dim3 g = g2(w, h);
        g.z = (unsigned)B;
        farneback_iteration_fused<<<g, 256, 0, st>>>(M_in, R0, R1, w, h, m, scale, flow, M_out, m_z, r1_z, flow_z);
        return MVS_OK;
    }
    const int cols = 64 + 2 * m;
    const bool tall = (size_t)div_up(w, 64) * div_up(h, 16) * B >= 2 * (size_t)ctx->num_cus;
    static std::atomic<unsigned> attr_set{0};  // per process: the LDS ceiling is a property of the kernel, set once each (bit = variant)
    auto go = [&](auto kernel, int ty, int px, int threads, unsigned bit) -> int {
        const size_t lds = (size_t)ty * 5 * (cols + cols / px + 1) * sizeof(double);
        if (!(attr_set.load() & bit)) {
            MVS_HIP(ctx, hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            attr_set.fetch_or(bit);
        }
        kernel<<<dim3(div_up(w, 64), div_up(h, ty), B), threads, lds, st>>>(M_in, R0, R1, w, h, m, scale, flow, M_out, m_z, r1_z, flow_z);
        return MVS_OK;
    };
    // 64 x 16 tiles, 8 rows per thread of the vertical walk; (test hook MVS_FB_VARIANT=2: 4 rows per thread on 512 threads -- twice the
    // wavefronts for the walk, measured SLOWER: 3.51 vs 3.16 ms per 1080p flow, its extra loads cost more than its occupancy buys)
    if (tall && ctx->hooks.fb_variant == 2) return go(farneback_iteration_tiled<4, 4, 512>, 16, 4, 512, 4u);
    if (tall) return go(farneback_iteration_tiled<8, 4, 256>, 16, 4, 256, 1u);
    return go(farneback_iteration_tiled<4, 2, 256>, 8, 2, 256, 2u);   // 64 x 8 tiles
}

// The buffers of one Farneback chain over B flows that share their first frame (B = 1: calculateFlow itself).  Images: 0 = the previous frame,
// 1 + i = next frame i, P floats apart in F and blur, w*h apart in a level's I, 5 w*h apart in its R.
struct FbBufs {
    const float *F;
    float *tmp, *row3;            // the unfused A/B form only (MVS_FB_UNFUSED)
    float *blur, *I, *R;          // one level's worth each (n P, n P, 5 n P) -- or, prepared for all levels at once, `levels_floats` of them
    size_t blur_cap, i_cap, r_cap; // floats available behind the three pointers
    float *M, *M2;                // B x 5 P each (ping-pong of the fused iteration)
    double *vs;                   // 5 P doubles, B == 1, unfused form only
    float *flowA, *flowB;         // B x 2 P each
};

// floats of blur / I / R when every level is prepared at once: (levels + 1) n P, n sum(w_k h_k), 5 n sum(w_k h_k) with sum < 2.8 P
static void fb_all_levels_floats(size_t P, size_t n, size_t &blur, size_t &I, size_t &R)
{
    blur = 11 * n * P;
    I = 3 * n * P;
    R = 15 * n * P;
}

// cv::FarnebackOpticalFlow::calc, flags 0, for B flows against one previous frame (farneback_device: B = 1; mvs_process_frame's batched pass:
// every side view of a main frame, blockIdx.z = flow).  flow_out: B x W*H*2.  Per pyramid level, coarse to fine:
//   preparation  GaussianBlur of every full-resolution frame, resize to the level, polynomial expansion -- depends on the frames alone:
//                for frames up to FB_BATCH_PREP_MAX_PIXELS all levels are prepared up front in THREE launches (round 6; 33 before);
//   chain        the flow carried down + the first matrices (one launch; two before), then `iterations` fused iterations -- each needs
//                the whole previous one.
// The same kernels' bodies on the same values either way (MVS_FB_SERIAL_PREP=1 keeps the per-level preparation: A/B, tests).
static int farneback_run(mvs_ctx *ctx, const FbBufs &b, int B, float *flow_out, int levels, double pyr_scale, int winsize, int iterations, int poly_n,
                         double poly_sigma)
{
    const int W = ctx->W, H = ctx->H;
    const size_t P = (size_t)W * H;
    const ptrdiff_t sP = (ptrdiff_t)P;
    if (poly_n > 15) return fail(ctx, MVS_EINVAL, "farneback: poly_n %d too large", poly_n);
    const int m = winsize / 2;
    // MVS_FB_UNFUSED=1 keeps the three-launch form of an iteration (A/B timing, single flows); windows beyond the LDS buffer use it too
    const bool unfused = B == 1 && (ctx->hooks.fb_unfused || m > FB_MAXM);
    if (B != 1 && m > FB_MAXM) return fail(ctx, MVS_EINVAL, "farneback (batched): window %d beyond the fused iteration's buffer", winsize);
    int lw[64], lh[64];
    double ls[64];
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
            ls[k] = scale;
            lw[k] = (int)std::lrint(W * scale);
            lh[k] = (int)std::lrint(H * scale);
        }
    }
    PolyTaps pt;
    farneback_taps(poly_n, poly_sigma, pt);
    hipStream_t st = ctx->stream;
    const unsigned nimg = (unsigned)B + 1;
    auto blur_size = [&](int k, double &sigma) {
        sigma = (1. / ls[k] - 1) * 0.5;
        int smooth_sz = (int)std::lrint(sigma * 5) | 1;
        return smooth_sz < 3 ? 3 : smooth_sz;
    };
    for (int k = 0; k <= levels; k++) {
        double sigma;
        if (blur_size(k, sigma) > 63) return fail(ctx, MVS_EINVAL, "farneback: smoothing kernel %d too large", blur_size(k, sigma));
    }
    // ---- preparation of every level at once ----
    FbLevels L;
    L.n = levels + 1;
    L.nimg = (int)nimg;
    bool all_levels = !ctx->hooks.fb_unfused && !ctx->hooks.fb_serial_prep && P <= FB_BATCH_PREP_MAX_PIXELS && levels + 1 <= 11;
    if (all_levels) {
        size_t bo = 0, io = 0, ro = 0;
        int cmax = 0;
        for (int k = 0; k <= levels; k++) {
            double sigma;
            const int smooth_sz = blur_size(k, sigma);
            Taps taps;
            gaussian_taps(smooth_sz, sigma, taps.k);
            FbLevelDesc &d = L.lv[k];
            d.w = lw[k];
            d.h = lh[k];
            d.c = smooth_sz / 2;
            d.pad = 0;
            d.blur_off = bo;
            d.i_off = io;
            d.r_off = ro;
            for (int j = 0; j <= d.c; j++) d.kc[j] = taps.k[d.c + j];
            for (int j = d.c + 1; j < 32; j++) d.kc[j] = 0.f;
            cmax = std::max(cmax, d.c);
            bo += (size_t)nimg * P;
            io += (size_t)nimg * d.w * d.h;
            ro += (size_t)nimg * 5 * d.w * d.h;
        }
        all_levels = bo <= b.blur_cap && io <= b.i_cap && ro <= b.r_cap;
        if (all_levels) {
            const unsigned z = (unsigned)(levels + 1) * nimg;
            gauss_fused_levels_kernel<<<dim3(div_up(W, 64), div_up(H, 16), z), 256, sizeof(float) * (size_t)(16 + 2 * cmax) * (128 + 2 * cmax), st>>>(b.F, W, H, L, b.blur);
            dim3 gL = g2(W, H);
            gL.z = z;
            resize_levels_kernel<<<gL, 256, 0, st>>>(b.blur, W, H, L, b.I);
            polyexp_levels_kernel<<<dim3(div_up(W, 64), div_up(H, 16), z), 256, sizeof(float) * ((size_t)(16 + 2 * pt.n) * (64 + 2 * pt.n) + 3 * 16 * (size_t)(64 + 2 * pt.n)), st>>>(b.I, L, pt, b.R);
            MVS_HIP(ctx, hipGetLastError());
        }
    }
    auto prepare = [&](int k) {   // one level, into the first level's worth of blur / I / R (image strides P / P / 5 P)
        double sigma;
        const int smooth_sz = blur_size(k, sigma);
        const int w = lw[k], h = lh[k];
        Taps taps;
        gaussian_taps(smooth_sz, sigma, taps.k);
        dim3 gF = g2(W, H), gL = g2(w, h);
        gF.z = gL.z = nimg;
        if (ctx->hooks.fb_unfused) {
            gauss_kernel<false><<<gF, 256, 0, st>>>(b.F, W, H, taps, smooth_sz, b.tmp, sP, sP);
            gauss_kernel<true><<<gF, 256, 0, st>>>(b.tmp, W, H, taps, smooth_sz, b.blur, sP, sP);
        } else {
            const int c = smooth_sz / 2;
            gauss_fused_kernel<<<dim3(div_up(W, 64), div_up(H, 16), nimg), 256, sizeof(float) * (size_t)(16 + 2 * c) * (128 + 2 * c), st>>>(b.F, W, H, taps, smooth_sz, b.blur, sP, sP);
        }
        resize_linear_kernel<1><<<gL, 256, 0, st>>>(b.blur, W, H, b.I, w, h, 1.f, 0, sP, sP);
        if (ctx->hooks.fb_unfused) {
            polyexp_vert<<<gL, 256, 0, st>>>(b.I, w, h, pt, b.row3, sP, 3 * sP);
            polyexp_horiz<<<gL, 256, 0, st>>>(b.row3, w, h, pt, b.R, 3 * sP, 5 * sP);
        } else {
            polyexp_fused_kernel<<<dim3(div_up(w, 64), div_up(h, 16), nimg), 256, sizeof(float) * ((size_t)(16 + 2 * pt.n) * (64 + 2 * pt.n) + 3 * 16 * (size_t)(64 + 2 * pt.n)), st>>>(b.I, w, h, pt, b.R, sP, 5 * sP);
        }
    };
    int r;
    float *flow = nullptr, *prevflow = nullptr;
    int pw = 0, ph = 0;
    const double bscale = 1. / ((double)winsize * winsize);
    for (int k = levels; k >= 0; k--) {
        const int w = lw[k], h = lh[k];
        if (!all_levels) prepare(k);
        // R of image i of this level: 5 w h apart behind the level's offset when all levels were prepared, 5 P apart otherwise
        const ptrdiff_t r_img = all_levels ? 5 * (ptrdiff_t)w * h : 5 * sP;
        const float *R0 = b.R + (all_levels ? L.lv[k].r_off : 0), *R1 = R0 + r_img;
        flow = k == 0 ? flow_out : (prevflow == b.flowA ? b.flowB : b.flowA);
        dim3 gB = g2(w, h);
        gB.z = (unsigned)B;
        upsample_update_kernel<<<gB, 256, 0, st>>>(prevflow, pw, ph, (float)(1. / pyr_scale), R0, R1, flow, w, h, b.M, r_img, 2 * sP, 5 * sP);
        if (unfused) {
            for (int it = 0; it < iterations; it++) {
                box_vert_kernel<<<g2(w, h), 256, 0, st>>>(b.M, w, h, m, b.vs);
                box_horiz_solve_kernel<<<g2(w, h), 256, 0, st>>>(b.vs, w, h, m, bscale, flow);
                if (it < iterations - 1) update_matrices_kernel<<<g2(w, h), 256, 0, st>>>(R0, R1, flow, w, h, b.M);
            }
        } else {
            float *M_cur = b.M, *M_nxt = b.M2;
            for (int it = 0; it < iterations; it++) {
                if ((r = launch_fb_iteration(ctx, M_cur, R0, R1, w, h, m, bscale, flow, it < iterations - 1 ? M_nxt : nullptr, B, 5 * sP, r_img, 2 * sP))) return r;
                std::swap(M_cur, M_nxt);
            }
        }
        MVS_HIP(ctx, hipGetLastError());
        prevflow = flow;
        pw = w;
        ph = h;
    }
    return MVS_OK;
}

// calculateFlow's Farneback on the context's arena.  f0 / f1: f32 frames (W*H each, f1 = f0 + W*H); flow_out: W*H*2.  arena: fb_work_floats(P) floats.
static size_t fb_work_floats(size_t P)
{
    size_t blur = 2 * P, I = 2 * P, R = 10 * P;
    if (P <= FB_BATCH_PREP_MAX_PIXELS) fb_all_levels_floats(P, 2, blur, I, R);
    return 10 * P + 2 * P + 6 * P + blur + I + R + 5 * P + 2 * P + 2 * P;   // vs, tmp, row3, blur, I, R, M, flowA, flowB
}

static int farneback_device(mvs_ctx *ctx, const float *f0, const float *f1, float *flow_out, float *arena, int levels,
                            double pyr_scale, int winsize, int iterations, int poly_n, double poly_sigma)
{
    const size_t P = (size_t)ctx->W * ctx->H;
    if (f1 != f0 + P) return fail(ctx, MVS_EINVAL, "farneback: the two frames must be adjacent in memory");
    FbBufs b;
    b.F = f0;
    b.vs = (double *)arena;  // 5 P doubles first: keeps them 8-byte aligned for any P
    float *p = arena + 10 * P;
    auto take = [&](size_t count) {
        float *q = p;
        p += count;
        return q;
    };
    b.tmp = take(2 * P);
    b.row3 = take(6 * P);
    b.blur_cap = 2 * P;
    b.i_cap = 2 * P;
    b.r_cap = 10 * P;
    if (P <= FB_BATCH_PREP_MAX_PIXELS) fb_all_levels_floats(P, 2, b.blur_cap, b.i_cap, b.r_cap);
    b.blur = take(b.blur_cap);
    b.I = take(b.i_cap);
    b.R = take(b.r_cap);
    b.M = take(5 * P);
    b.M2 = (float *)b.vs;  // the fused form keeps the vertical sums on chip: vs is free
    b.flowA = take(2 * P);
    b.flowB = take(2 * P);
    if ((size_t)(p - arena) > fb_work_floats(P)) return fail(ctx, MVS_ESTATE, "farneback: arena layout exceeds its %zu floats", fb_work_floats(P));
    return farneback_run(ctx, b, 1, flow_out, levels, pyr_scale, winsize, iterations, poly_n, poly_sigma);
}

// VariationalRefinement::calc with OpenCV's defaults; flow (W*H*2) is refined in place.  arena: >= 22*P floats.
static int variational_device(mvs_ctx *ctx, const float *I0, const float *I1, float *flow, float *arena)
{
    const int w = ctx->W, h = ctx->H;
    const size_t P = (size_t)w * h;
    hipStream_t st = ctx->stream;
    VarBufs B;
    float *A;
    float **slots[] = {&B.Wu, &B.Wv, &B.du, &B.dv, &A, &B.Iz, &B.Ix, &B.Iy, &B.Ixx, &B.Ixy, &B.Iyy, &B.Ixz, &B.Iyz,
                       &B.a11, &B.a12, &B.a22, &B.b1, &B.b2, &B.wgt};
    for (size_t i = 0; i < sizeof(slots) / sizeof(slots[0]); i++) *slots[i] = arena + i * P;
    split_flow_kernel<<<g1(P), 256, 0, st>>>(flow, B.Wu, B.Wv, B.du, B.dv, P);
    warp_q5_kernel<<<g2(w, h), 256, 0, st>>>(I1, w, h, B.Wu, B.Wv, I0, A, B.Iz);
    if (ctx->hooks.var_unfused) {   // (the A/B form keeps the seven separate launches too)
        diff_kernel<false><<<g2(w, h), 256, 0, st>>>(A, w, h, B.Ix);
        diff_kernel<true><<<g2(w, h), 256, 0, st>>>(A, w, h, B.Iy);
        diff_kernel<false><<<g2(w, h), 256, 0, st>>>(B.Iz, w, h, B.Ixz);
        diff_kernel<true><<<g2(w, h), 256, 0, st>>>(B.Iz, w, h, B.Iyz);
        diff_kernel<false><<<g2(w, h), 256, 0, st>>>(B.Ix, w, h, B.Ixx);
        diff_kernel<true><<<g2(w, h), 256, 0, st>>>(B.Ix, w, h, B.Ixy);
        diff_kernel<true><<<g2(w, h), 256, 0, st>>>(B.Iy, w, h, B.Iyy);
    } else {
        var_derivatives_kernel<<<g2(w, h), 256, 0, st>>>(A, B.Iz, w, h, B);
    }
    const dim3 half(div_up((w + 1) / 2, 64), div_up(h, 4));
    // MVS_VAR_UNFUSED=1 keeps the 13-launch form of a fixed-point iteration (A/B timing and the cross-check test)
    const bool unfused = ctx->hooks.var_unfused;
    const int sor_iters = 5;
    if (2 * sor_iters > VT_HALO) return fail(ctx, MVS_EINVAL, "variational: %d SOR sweeps need a halo of %d", sor_iters, 2 * sor_iters);
    float *du_cur = B.du, *dv_cur = B.dv, *du_nxt = B.a11, *dv_nxt = B.a22;  // the fused form keeps coefficients on chip
    for (int fp = 0; fp < 5; fp++) {
        if (unfused) {
            var_data_term<<<g2(w, h), 256, 0, st>>>(B, w, h);
            var_diffusivity<<<g2(w, h), 256, 0, st>>>(B, w, h);
            var_smooth_gather<<<g2(w, h), 256, 0, st>>>(B, w, h);
            for (int it = 0; it < sor_iters; it++) {
                var_sor_pass<<<half, 256, 0, st>>>(B, w, h, 0);
                var_sor_pass<<<half, 256, 0, st>>>(B, w, h, 1);
            }
        } else {
            var_fixed_point_fused<<<dim3(div_up(w, VT_W), div_up(h, VT_H)), VT_THREADS, 0, st>>>(B, du_cur, dv_cur, du_nxt, dv_nxt, w, h, sor_iters);
            std::swap(du_cur, du_nxt);
            std::swap(dv_cur, dv_nxt);
        }
    }
    B.du = du_cur;
    B.dv = dv_cur;
    var_finish<<<g1(P), 256, 0, st>>>(B, flow, P);
    MVS_HIP(ctx, hipGetLastError());
    return MVS_OK;
}

struct FlowBufs {
    float *arena, *f0, *f1, *flow2, *var, *out4;
    uint8_t *p8, *n8, *r8;
};

static int flow_prepare(mvs_ctx *ctx, FlowBufs &b, int use_farneback, bool flow_only = false)
{
    const size_t P = (size_t)ctx->W * ctx->H;
    // arena: work (fb_work_floats(P) floats, at least the variational path's 22 P; starts with 5P doubles) first, then f0, f1, flow2, var, out4; u8: prev, next, remapped
    // (sized for the algorithm that runs: the variational default needs 22 P, and mvs_process_frame keeps one arena per flow lane)
    const size_t work = use_farneback ? std::max(fb_work_floats(P), (size_t)42 * P) : (size_t)42 * P;
    int rc;
    if ((rc = ensure(ctx, ctx->flow_arena, sizeof(float) * (P * (2 + 2 + 1 + 4) + work) + 3 * P + 256))) return rc;
    b.arena = (float *)ctx->flow_arena.ptr;
    b.f0 = b.arena + work;
    b.f1 = b.f0 + P;
    b.flow2 = b.f1 + P;
    b.var = b.flow2 + 2 * P;
    b.out4 = b.var + P;
    b.p8 = (uint8_t *)(b.out4 + 4 * P);
    b.n8 = b.p8 + P;
    b.r8 = b.n8 + P;
    if (flow_only) return MVS_OK;  // (no variance channel: neither the bicubic table nor compare()'s pyramids -- a lane's shadow context never has them)
    if ((rc = ensure_cubic_table(ctx))) return rc;
    return compare_prepare(ctx);
}

// flow.cpp:19-42 on the buffers of `b`: inputs b.p8 / b.n8 (u8), output b.out4.  Everything between is a fixed
// sequence of kernels on fixed buffers.
// flow_only: stop after the flow (b.flow2): the variance channel and the packing are the caller's (flow_variance_batch_device)
static int flow_run(mvs_ctx *ctx, const FlowBufs &b, int use_farneback, bool flow_only = false)
{
    const int W = ctx->W, H = ctx->H;
    const size_t P = (size_t)W * H;
    hipStream_t st = ctx->stream;
    auto enqueue = [&]() -> int {
        u8_to_f32_pair_kernel<<<g1(P), 256, 0, st>>>(b.p8, b.n8, b.f0, b.f1, P);
        int r;
        if (use_farneback) {
            const double poly_sigma = (H + W) / 1000.0;  // flow.cpp:24-25
            const int winsize = (H + W) / 100, poly_n = poly_sigma < 1.5 ? 5 : 7;
            if ((r = farneback_device(ctx, b.f0, b.f1, b.flow2, b.arena, 10, 0.8, winsize, 7, poly_n, poly_sigma))) return r;
        } else {
            if (ctx->hooks.flow_graph_kernel_memset)
                zero_f32_kernel<<<g1(2 * P), 256, 0, st>>>(b.flow2, 2 * P);
            else
                MVS_HIP(ctx, hipMemsetAsync(b.flow2, 0, sizeof(float) * 2 * P, st));  // flow.cpp:31 (uninitialised there), A-11
            if ((r = variational_device(ctx, b.f0, b.f1, b.flow2, b.arena))) return r;
        }
        if (flow_only) {
            MVS_HIP(ctx, hipGetLastError());
            return MVS_OK;
        }
        if ((r = remap_device(ctx, b.flow2, 2, b.n8, b.r8))) return r;  // flow.cpp:34
        if ((r = compare_device(ctx, b.p8, b.r8, b.var))) return r;
        pack_flow4<<<g1(P), 256, 0, st>>>(b.flow2, b.var, b.out4, P);
        MVS_HIP(ctx, hipGetLastError());
        return MVS_OK;
    };
    // Eager launches.  Rounds 2-3 replayed this sequence as a hipGraph; round 4 found that a graph instantiated before some first-time
    // event elsewhere in the process (the first mvs_poisson_surface call: rocFFT's run-time kernels, new code objects) replays with WRONG
    // results afterwards -- silently, deterministically within the process, differently from process to process --
    // and that on this ROCm the eager launches are no slower (mvs_process_frame 1.69 ms against 1.95 with the graphs).
    ProfileScope ps(ctx, MVS_K_FLOW);
    if (ctx->hooks.flow_graph) {  // test hook: the graph replay of rounds 2-3, kept for the reproducer of what round 4 found (tools/graph_repro.py)
        const int gi = use_farneback ? 1 : 0;
        if (ctx->flow_graph[gi] && ctx->flow_graph_arena[gi] != ctx->flow_arena.ptr) {
            (void)hipGraphExecDestroy(ctx->flow_graph[gi]);
            ctx->flow_graph[gi] = nullptr;
        }
        if (!ctx->flow_graph[gi]) {
            hipGraph_t graph = nullptr;
            MVS_HIP(ctx, hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
            const int r = enqueue();
            const hipError_t e = hipStreamEndCapture(st, &graph);
            if (r != MVS_OK || e != hipSuccess || !graph || hipGraphInstantiate(&ctx->flow_graph[gi], graph, nullptr, nullptr, 0) != hipSuccess) {
                ctx->flow_graph[gi] = nullptr;
                if (graph) (void)hipGraphDestroy(graph);
                return fail(ctx, MVS_EHIP, "MVS_FLOW_GRAPH: capture / instantiation failed");
            }
            (void)hipGraphDestroy(graph);
            ctx->flow_graph_arena[gi] = ctx->flow_arena.ptr;
        }
        MVS_HIP(ctx, hipGraphLaunch(ctx->flow_graph[gi], st));
        return MVS_OK;
    }
    return enqueue();
}

// ---- calculateFlow (Farneback) of SEVERAL next-frames against one previous frame in one pass --------------------------------
// recon.cpp:76-112 computes calculateFlow(main frame, mixed_i) for every side view i of a main frame: same sizes, same parameters,
// the same first frame.  As separate chains (one per side view, even on concurrent streams) a flow is ~150 launches of a few
// microseconds each (11 pyramid levels x 7 iterations: 47 % of the time in 9 us launches, profiles/r02); here every launch covers
// all B flows (blockIdx.z), so the launch count per main frame falls B-fold and the small pyramid levels fill more of the chip,
// and everything that depends on the first frame alone (its blur, its polynomial expansion R0) is computed once, not B times.
// Per-pixel arithmetic and summation orders are those of farneback_device: results are bit-identical.
struct FlowBatchBufs {
    FbBufs fb;
    float *F, *flow2, *var;
    uint8_t *r8;
    size_t floats;
};

static FlowBatchBufs flow_batch_layout(float *arena, size_t P, int B)
{
    FlowBatchBufs b;
    const size_t n = (size_t)B + 1;  // image 0 = the common previous frame, 1 + i = next frame i
    float *p = arena;
    auto take = [&](size_t count) {
        float *r = p;
        p += count;
        return r;
    };
    b.F = take(n * P);
    b.fb.F = b.F;
    b.fb.vs = nullptr;
    b.fb.tmp = take(n * P);
    b.fb.row3 = take(n * 3 * P);
    b.fb.blur_cap = n * P;
    b.fb.i_cap = n * P;
    b.fb.r_cap = n * 5 * P;
    if (P <= FB_BATCH_PREP_MAX_PIXELS) fb_all_levels_floats(P, n, b.fb.blur_cap, b.fb.i_cap, b.fb.r_cap);
    b.fb.blur = take(b.fb.blur_cap);
    b.fb.I = take(b.fb.i_cap);
    b.fb.R = take(b.fb.r_cap);
    b.fb.M = take((size_t)B * 5 * P);
    b.fb.M2 = take((size_t)B * 5 * P);
    b.fb.flowA = take((size_t)B * 2 * P);
    b.fb.flowB = take((size_t)B * 2 * P);
    b.flow2 = take((size_t)B * 2 * P);
    b.var = take((size_t)B * P);
    b.r8 = (uint8_t *)take(((size_t)B * P + 3) / 4);
    b.floats = (size_t)(p - arena);
    return b;
}

static int farneback_batch_enqueue(mvs_ctx *ctx, const uint8_t *prev8, const uint8_t *next8 /* B frames, P apart */, int B, float *out4 /* B x 4P */, const FlowBatchBufs &b)
{
    const int W = ctx->W, H = ctx->H;
    const size_t P = (size_t)W * H;
    hipStream_t st = ctx->stream;
    const double poly_sigma = (H + W) / 1000.0;  // flow.cpp:24-25
    const int winsize = (H + W) / 100, poly_n = poly_sigma < 1.5 ? 5 : 7;
    u8_to_f32_kernel<<<g1(P), 256, 0, st>>>(prev8, b.F, P);
    u8_to_f32_kernel<<<g1(P * B), 256, 0, st>>>(next8, b.F + P, P * B);
    int r;
    if ((r = farneback_run(ctx, b.fb, B, b.flow2, 10, 0.8, winsize, 7, poly_n, poly_sigma))) return r;
    // variance channel of every flow: compare(prev, flowRemap(flow_i, next_i)) (flow.cpp:34), then the packing (37-41) -- each launch covers all B
    // flows (round 6: twelve launches instead of twelve per side view)
    if ((r = remap_batch_device(ctx, b.flow2, 2, (ptrdiff_t)(2 * P), next8, B, b.r8))) return r;
    if ((r = compare_batch_device(ctx, prev8, b.r8, B, b.var))) return r;
    pack_flow4<<<dim3(g1(P), (unsigned)B), 256, 0, st>>>(b.flow2, b.var, out4, P);
    MVS_HIP(ctx, hipGetLastError());
    return MVS_OK;
}

// calculateFlow(prev, next_i, useFarneback = true) for i < B on device buffers (next8: B frames P bytes apart, out4: B x H*W*4 f32);
// in-stream
int flow_farneback_batch_device(mvs_ctx *ctx, const uint8_t *prev_dev, const uint8_t *next_dev, int B, float *out4_dev)
{
    const size_t P = (size_t)ctx->W * ctx->H;
    if (B < 1 || B > 64) return fail(ctx, MVS_EINVAL, "flow batch of %d", B);
    int rc;
    const FlowBatchBufs probe = flow_batch_layout(nullptr, P, B);
    if ((rc = ensure(ctx, ctx->flow_batch_arena, sizeof(float) * probe.floats + 256))) return rc;
    if ((rc = ensure_cubic_table(ctx))) return rc;
    if ((rc = compare_prepare(ctx, B))) return rc;
    const FlowBatchBufs b = flow_batch_layout((float *)ctx->flow_batch_arena.ptr, P, B);
    ProfileScope ps(ctx, MVS_K_FLOW);  // (eager launches: see flow_run)
    return farneback_batch_enqueue(ctx, prev_dev, next_dev, B, out4_dev, b);
}

// calculateFlow on device buffers (prev/next: H*W u8, out4: H*W*4 f32), all in-stream
int flow_device(mvs_ctx *ctx, const uint8_t *prev_dev, const uint8_t *next_dev, int use_farneback, float *out4_dev)
{
    const size_t P = (size_t)ctx->W * ctx->H;
    FlowBufs b;
    int rc = flow_prepare(ctx, b, use_farneback);
    if (rc) return rc;
    // the caller's device buffers ARE the inputs and the output (round 6: three device-to-device copies per flow fewer -- each a launch of its own
    // with ~10 us of host latency, per side view of every main frame); the arena's own p8 / n8 / out4 serve mvs_flow's host buffers
    (void)P;
    b.p8 = const_cast<uint8_t *>(prev_dev);
    b.n8 = const_cast<uint8_t *>(next_dev);
    b.out4 = out4_dev;
    return flow_run(ctx, b, use_farneback);
}

// the dense flow alone (W*H*2 f32 into flow2_dev), without the variance channel: mvs_process_frame computes the variance channels of all its
// side views in one batched pass afterwards (flow_variance_batch_device) instead of twelve launches per side view on the flow's own lane
int flow_only_device(mvs_ctx *ctx, const uint8_t *prev_dev, const uint8_t *next_dev, int use_farneback, float *flow2_dev)
{
    FlowBufs b;
    int rc = flow_prepare(ctx, b, use_farneback, true);
    if (rc) return rc;
    b.p8 = const_cast<uint8_t *>(prev_dev);
    b.n8 = const_cast<uint8_t *>(next_dev);
    b.flow2 = flow2_dev;
    return flow_run(ctx, b, use_farneback, true);
}

// flow.cpp:34-41 for B flows against one previous frame: variance_i = compare(prev, flowRemap(flow_i, next_i)), out4_i = (u, v, variance, 0).
// flow2: B x 2P f32, next8 / r8 (scratch): B x P u8, var (scratch): B x P f32, out4: B x 4P f32.  Twelve launches whatever B.
int flow_variance_batch_device(mvs_ctx *ctx, const uint8_t *prev8, const uint8_t *next8, const float *flow2, int B, uint8_t *r8, float *var, float *out4)
{
    const size_t P = (size_t)ctx->W * ctx->H;
    int r;
    if ((r = compare_prepare(ctx, B))) return r;
    if ((r = remap_batch_device(ctx, flow2, 2, (ptrdiff_t)(2 * P), next8, B, r8))) return r;
    if ((r = compare_batch_device(ctx, prev8, r8, B, var))) return r;
    pack_flow4<<<dim3(g1(P), (unsigned)B), 256, 0, ctx->stream>>>(flow2, var, out4, P);
    MVS_HIP(ctx, hipGetLastError());
    return MVS_OK;
}

}  // namespace mvs

using namespace mvs;

extern "C" {

// test hook (not in mvs.h; tools/graph_repro.py): the first `count` floats of calculateFlow's work arena, as the last mvs_flow left them
int mvs_test_flow_arena(mvs_ctx *ctx, float *out, size_t count)
{
    if (!ctx || !out) return MVS_EINVAL;
    if (!ctx->flow_arena.ptr || count * sizeof(float) > ctx->flow_arena.bytes) return fail(ctx, MVS_ESTATE, "mvs_test_flow_arena: the arena holds %zu bytes", ctx->flow_arena.bytes);
    MVS_HIP(ctx, hipSetDevice(ctx->device));
    MVS_HIP(ctx, hipMemcpyAsync(out, ctx->flow_arena.ptr, count * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    MVS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return MVS_OK;
}

int mvs_flow(mvs_ctx *ctx, const uint8_t *prev_hw, const uint8_t *next_hw, int use_farneback, float *out_hw4)
{
    if (!ctx || !prev_hw || !next_hw || !out_hw4) return fail(ctx, MVS_EINVAL, "mvs_flow: null argument");
    MVS_HIP(ctx, hipSetDevice(ctx->device));
    const size_t P = (size_t)ctx->W * ctx->H;
    FlowBufs b;
    int rc = flow_prepare(ctx, b, use_farneback);
    if (rc) return rc;
    MVS_HIP(ctx, hipMemcpyAsync(b.p8, prev_hw, P, hipMemcpyHostToDevice, ctx->stream));
    MVS_HIP(ctx, hipMemcpyAsync(b.n8, next_hw, P, hipMemcpyHostToDevice, ctx->stream));
    if ((rc = flow_run(ctx, b, use_farneback))) return rc;
    MVS_HIP(ctx, hipMemcpyAsync(out_hw4, b.out4, sizeof(float) * 4 * P, hipMemcpyDeviceToHost, ctx->stream));
    MVS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return MVS_OK;
}

}  // extern "C"
