// triangulate.hip -- triangulatePixels() of the reference (util.cpp:167-329, with triangulatePixel 62-164, goodSample
// 44-53, sampleImage<T> 438-461, imageGradient 465-479) on gfx950: the consumer of depth + flows (recon.cpp:114).
//
// The reference loops over pixels on one CPU thread, allocating cv::Mat temporaries per pixel per Newton step
// (util.cpp:190-243) and running a PCA per pixel (299).  Here every pixel is a thread:
//   sobel_kernel       imageGradient(depth)
//   tri_points_kernel  measured points + inverse covariances per side camera, the <= 50-step Newton solve, the pdf
//   tri_normals_kernel 21x21-neighbourhood PCA (3x3 Jacobi in f64), orientation vote, pdf scaling
//   compact_count / compact_scan / compact_scatter   the valid pixels in pixelId order (util.cpp:172,247-248), packed on the
//                      device so that only the n x 7 result crosses PCIe
// Quirks are reproduced as catalogued in oracle/triangulate_oracle.c (swapped bilinear weights, y + fly, the
// type-punned gradient, the un-dehomogenised fallback normal).  Arithmetic mirrors the oracle statement by statement
// (f32 storage, f64 accumulation of every matrix product); exp()/pow() come from the device math library, so pdf and
// the normal scale carry a last-ulp tolerance (tests/test_triangulate_gpu.py).
#include "mvs_internal.hpp"

#include <cmath>

namespace mvs {

constexpr int TRI_MAXCAM = 32;

struct CamPre {
    float CM[16];
    float B[6];
    float projDeriv[2];
    float projW[4];
    float center[3];
    float pad;
};

__device__ __forceinline__ int refl(int p, int n)
{
    if (n == 1) return 0;
    while (p < 0 || p >= n) {
        if (p < 0) p = -p;
        if (p >= n) p = 2 * n - 2 - p;
    }
    return p;
}

__global__ __launch_bounds__(256) void sobel_kernel(const float *__restrict__ img, int W, int H, float *__restrict__ grad2)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    float rdx[3], rsm[3];
#pragma unroll
    for (int j = 0; j < 3; j++) {
        const float *r = img + (size_t)refl(y + j - 1, H) * W;
        const float v0 = r[refl(x - 1, W)], v1 = r[x], v2 = r[refl(x + 1, W)];
        rdx[j] = v2 - v0;
        rsm[j] = v0 + v1 * 2.0f + v2;
    }
    grad2[((size_t)y * W + x) * 2] = rdx[0] + rdx[1] * 2.0f + rdx[2];
    grad2[((size_t)y * W + x) * 2 + 1] = rsm[2] - rsm[0];
}

__device__ __forceinline__ void mat44_vec(const float *a, const float *v, float *o)
{
#pragma unroll
    for (int i = 0; i < 4; i++) {
        double s = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) s += (double)a[4 * i + k] * (double)v[k];
        o[i] = (float)s;
    }
}

__device__ __forceinline__ bool good_sample(const float *__restrict__ depth, int W, int H, float x, float y)
{
    const int ix = (int)x, iy = (int)y;
    if (ix <= 0 || ix >= W - 1 || iy <= 0 || iy >= H - 1) return false;
    const float *p = depth + (size_t)iy * W + ix;
    return p[0] != MVS_BACKGROUND_DEPTH && p[1] != MVS_BACKGROUND_DEPTH && p[W] != MVS_BACKGROUND_DEPTH &&
           p[W + 1] != MVS_BACKGROUND_DEPTH;
}

__device__ __forceinline__ float sample_f32(const float *__restrict__ img, int W, float x, float y)
{
    const float lw = fmodf(x, 1.0f), rw = 1 - lw, tw = fmodf(y, 1.0f), bw = 1 - tw;
    const float *p = img + (size_t)(int)y * W + (int)x;
    return (p[0] * lw + p[1] * rw) * tw + (p[W] * lw + p[W + 1] * rw) * bw;
}

__device__ __forceinline__ int round_sat(float v)
{
    if (!(v > -2147483648.0f)) return (int)0x80000000;
    if (!(v < 2147483648.0f)) return 0x7fffffff;
    return __float2int_rn(v);
}

__device__ __forceinline__ void sample_point_punned(const float *__restrict__ grad2, int W, int H, float x, float y,
                                                    float out[2])
{
    const float lw = fmodf(x, 1.0f), rw = 1 - lw, tw = fmodf(y, 1.0f), bw = 1 - tw;
    const int ix = (int)x, iy = (int)y;
    const int ix1 = ix + 1 < W ? ix + 1 : ix, iy1 = iy + 1 < H ? iy + 1 : iy;
#pragma unroll
    for (int c = 0; c < 2; c++) {
        const int b00 = __float_as_int(grad2[((size_t)iy * W + ix) * 2 + c]);
        const int b01 = __float_as_int(grad2[((size_t)iy * W + ix1) * 2 + c]);
        const int b10 = __float_as_int(grad2[((size_t)iy1 * W + ix) * 2 + c]);
        const int b11 = __float_as_int(grad2[((size_t)iy1 * W + ix1) * 2 + c]);
        const int top = (int)((uint32_t)round_sat((float)b00 * lw) + (uint32_t)round_sat((float)b01 * rw));
        const int bot = (int)((uint32_t)round_sat((float)b10 * lw) + (uint32_t)round_sat((float)b11 * rw));
        const int r = (int)((uint32_t)round_sat((float)top * tw) + (uint32_t)round_sat((float)bot * bw));
        out[c] = __int_as_float(r);
    }
}

// VMAX: compile-time capacity of the per-view arrays.  With VMAX = 32 (TRI_MAXCAM) they are indexed by a runtime loop and live
// in scratch memory (784 B per lane), re-read on each of up to 50 Newton iterations; the reference uses 2-8 side views per main
// frame, so the kernel is also instantiated for 4 and 8 with fully unrolled, predicated view loops (arrays in registers).
// Same statements per view in the same order: bit-identical.
//
// PHASE: the Newton solve stops when |delta z| < 1e-7 or after 50 steps.  Most pixels converge in a handful of steps, but a
// wavefront iterates until its slowest lane is done, so a few non-converging pixels kept whole waves at the cap (275 us at
// 640x480 where ~5 steps per pixel would take ~45).  PHASE 0 runs every pixel for at most TRI_SPLIT steps and queues the
// unconverged ones (pixel, current z); PHASE 1 finishes the queue densely packed, one queued pixel per thread, recomputing the
// per-view set-up (a pure function of the inputs) and continuing from step TRI_SPLIT.  The per-pixel step sequence is
// unchanged: bit-identical.
constexpr int TRI_SPLIT = 6;  // 4: 2.01 ms per frame at 640x480 (too many pixels queued), 6: 1.89, 9: 1.90
struct TriQueued {
    int pix;
    float z;
};

template <int VMAX, int PHASE>
__global__ __launch_bounds__(128) void tri_points_kernel(const float *const *__restrict__ flows, const CamPre *__restrict__ pre,
                                                         int V, const float *__restrict__ Minv, const float *__restrict__ depth,
                                                         const float *__restrict__ grad, int W, int H,
                                                         uint8_t *__restrict__ valid, float *__restrict__ pts,
                                                         float *__restrict__ xyz3, float *__restrict__ pdfs,
                                                         int *__restrict__ queue_count, TriQueued *__restrict__ queue)
{
    int col, row;
    float z_start = 0.f;
    if (PHASE == 0) {
        col = blockIdx.x * 64 + (threadIdx.x & 63);
        row = blockIdx.y * 2 + (threadIdx.x >> 6);
        if (col >= W || row >= H) return;
    } else {
        const int q = (blockIdx.y * gridDim.x + blockIdx.x) * 128 + threadIdx.x;
        if (q >= *queue_count) return;
        const int qp = queue[q].pix;
        z_start = queue[q].z;
        row = qp / W;
        col = qp - row * W;
    }
    const size_t pix = (size_t)row * W + col;
    if (PHASE == 0) {
        valid[pix] = 0;
        ((float4 *)xyz3)[pix] = make_float4(0.f, 0.f, 0.f, 0.f);  // (x, y, z, valid) per pixel, one aligned 16-byte record
    }
    const float d0 = depth[pix];
    if (d0 == MVS_BACKGROUND_DEPTH) return;
    const float centerX = W / 2.0f, centerY = H / 2.0f, scaleX = 2.0f / W, scaleY = 2.0f / H;
    const float x = (col - centerX) * scaleX, y = (centerY - row) * scaleY;
    float meas[VMAX][2], icov[VMAX][4];
#pragma unroll VMAX <= 8 ? VMAX : 1
    for (int i = 0; i < VMAX; i++) {
        if (i >= V) break;
        const float *fl = flows[i] + pix * 4;
        const float flx = fl[0], fly = fl[1], variance = fl[2];
        const bool gs = good_sample(depth, W, H, col + flx, row + fly);
        const float z = gs ? sample_f32(depth, W, col + flx, row + fly) : d0;
        const float v4[4] = {x + flx * scaleX, y + fly * scaleY, z, 1.0f};
        float mp[4];
        mat44_vec(pre[i].CM, v4, mp);
        float g[2];
        if (gs)
            sample_point_punned(grad, W, H, col + flx, row + fly, g);
        else
            sample_point_punned(grad, W, H, (float)col, (float)row, g);
        float A[4];
#pragma unroll
        for (int r = 0; r < 2; r++)
#pragma unroll
            for (int c = 0; c < 2; c++) {
                const double s = (double)pre[i].B[3 * r + 0] * (c == 0 ? 1.0 : 0.0) + (double)pre[i].B[3 * r + 1] * (c == 1 ? 1.0 : 0.0) +
                                 (double)pre[i].B[3 * r + 2] * (double)g[c];
                A[2 * r + c] = (float)s / mp[3];
            }
        float S[4];
#pragma unroll
        for (int r = 0; r < 2; r++)
#pragma unroll
            for (int c = 0; c < 2; c++) S[2 * r + c] = (float)((double)A[2 * r] * A[2 * c] + (double)A[2 * r + 1] * A[2 * c + 1]);
        const double det = (double)S[0] * S[3] - (double)S[1] * S[2];
        float inv[4] = {0.f, 0.f, 0.f, 0.f};
        if (det != 0.0) {
            const double rd = 1.0 / det;
            inv[0] = (float)(S[3] * rd);
            inv[1] = (float)(-S[1] * rd);
            inv[2] = (float)(-S[2] * rd);
            inv[3] = (float)(S[0] * rd);
        }
#pragma unroll
        for (int k = 0; k < 4; k++) icov[i][k] = inv[k] / variance;
        const float mz = mp[2] / mp[3];
        if (mz < -1) return;  // util.cpp:229-233
        meas[i][0] = mp[0] / mp[3];
        meas[i][1] = mp[1] / mp[3];
    }
    float k[4] = {x, y, PHASE == 0 ? d0 : z_start, 1.0f};
    float pdf = 1.0f;
    for (int iter = PHASE == 0 ? 0 : TRI_SPLIT;; iter++) {
        if (PHASE == 0 && iter == TRI_SPLIT) {  // not converged yet: hand over to the densely packed second pass
            const int slot = atomicAdd(queue_count, 1);
            queue[slot].pix = (int)pix;
            queue[slot].z = k[2];
            return;
        }
        double firstDz = 0, secondDz = 0;
#pragma unroll VMAX <= 8 ? VMAX : 1
        for (int i = 0; i < VMAX; i++) {
            if (i >= V) break;
            float ep[4];
            mat44_vec(pre[i].CM, k, ep);
            const float px = ep[0] / ep[3], py = ep[1] / ep[3];
            double w = 0;
#pragma unroll
            for (int c = 0; c < 4; c++) w += (double)pre[i].projW[c] * (double)k[c];
            const float pw = (float)w;
            const float dpx = pre[i].projDeriv[0] / pw, dpy = pre[i].projDeriv[1] / pw;
            const float dfx = px - meas[i][0], dfy = py - meas[i][1];
            const float t0 = (float)((double)icov[i][0] * dpx + (double)icov[i][1] * dpy);
            const float t1 = (float)((double)icov[i][2] * dpx + (double)icov[i][3] * dpy);
            firstDz += (double)dfx * t0 + (double)dfy * t1;
            secondDz += (double)dpx * t0 + (double)dpy * t1;
        }
        const double delta_z = -firstDz / secondDz, eps = 1e-7;
        if (iter >= 50 || (delta_z < eps && delta_z > -eps)) {
            double exponent = 0, product_ivar = 1;
#pragma unroll VMAX <= 8 ? VMAX : 1
            for (int i = 0; i < VMAX; i++) {
                if (i >= V) break;
                float ep[4];
                mat44_vec(pre[i].CM, k, ep);
                const float dfx = ep[0] / ep[3] - meas[i][0], dfy = ep[1] / ep[3] - meas[i][1];
                const float t0 = (float)((double)icov[i][0] * dfx + (double)icov[i][1] * dfy);
                const float t1 = (float)((double)icov[i][2] * dfx + (double)icov[i][3] * dfy);
                exponent -= (double)dfx * t0 + (double)dfy * t1;
                product_ivar *= (double)icov[i][0] * icov[i][3] - (double)icov[i][1] * icov[i][2];
            }
            pdf = (float)(0.159 * product_ivar * exp(0.5 * exponent));
            break;
        }
        k[2] = (float)((double)k[2] + delta_z);
    }
    float out[4];
    mat44_vec(Minv, k, out);
#pragma unroll
    for (int c = 0; c < 4; c++) pts[pix * 4 + c] = out[c];
    // the normals pass visits every point up to 2 x 441 times: dehomogenise once (same f32 quotients as util.cpp:291)
    ((float4 *)xyz3)[pix] = make_float4(out[0] / out[3], out[1] / out[3], out[2] / out[3], 1.0f);
    pdfs[pix] = pdf;
    valid[pix] = 1;
}

__device__ void smallest_eigvec3(double a[3][3], double v[3])
{
    double Vm[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
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
                    const double vkp = Vm[k][p], vkq = Vm[k][q];
                    Vm[k][p] = c * vkp - s * vkq;
                    Vm[k][q] = s * vkp + c * vkq;
                }
            }
    }
    int m = 0;
    if (a[1][1] < a[m][m]) m = 1;
    if (a[2][2] < a[m][m]) m = 2;
    for (int k = 0; k < 3; k++) v[k] = Vm[k][m];
}

constexpr int TN_W = 64, TN_H = 4, TN_R = 10, TN_RW = TN_W + 2 * TN_R, TN_RH = TN_H + 2 * TN_R;

__global__ __launch_bounds__(TN_W * TN_H) void tri_normals_kernel(const uint8_t *__restrict__ valid, const float *__restrict__ pts,
                                                          const float *__restrict__ xyz3, const float *__restrict__ pdfs, const CamPre *__restrict__ pre,
                                                          const float *__restrict__ main_center, int V, int W, int H,
                                                          float *__restrict__ normals)
{
    // The 64 x 4-pixel tile and its 10-pixel apron are staged once as (x, y, z, valid) records: every pixel reads its 441
    // neighbours twice, which through L1 is 882 16-byte loads per pixel against 8 staged records per pixel here.  Cells
    // outside the image are staged as invalid, so the window walk needs no bounds logic at all.
    __shared__ float4 cell[TN_RW * TN_RH];
    const int X0 = blockIdx.x * TN_W - TN_R, Y0 = blockIdx.y * TN_H - TN_R;
    for (int c = threadIdx.x; c < TN_RW * TN_RH; c += TN_W * TN_H) {
        const int ry = c / TN_RW, rx = c - ry * TN_RW, gx = X0 + rx, gy = Y0 + ry;
        cell[c] = (gx >= 0 && gx < W && gy >= 0 && gy < H) ? ((const float4 *)xyz3)[(size_t)gy * W + gx] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
    const int lx = threadIdx.x & 63, ly = threadIdx.x >> 6;
    const int col = blockIdx.x * TN_W + lx;
    const int row = blockIdx.y * TN_H + ly;
    if (col >= W || row >= H) return;
    const size_t pix = (size_t)row * W + col;
    if (!valid[pix]) return;
    const int radius = TN_R;
    float pdf = pdfs[pix];
    if (V > 1) pdf = (float)pow((double)pdf, 1.0 / V);
    // The window is walked row by row; a row's 21 neighbours are read unconditionally and only then accumulated, in the
    // reference's order, with selects: an invalid neighbour adds +0.0, which leaves a double accumulator unchanged, so the sums
    // are bit-identical to a branchy walk.  History at 640x480: branchy global loads 501 us; branch-free 485; aligned float4
    // records 361; staged in LDS (this form) about the same at 640x480 and 1.6 % better per frame at 1080p.  f64 ops issue at
    // the f32 rate on gfx950 (profiles/r01/valu_issue_microbench.txt): what bounds the kernel is its instruction count.
    constexpr int WIN = 21;
    // Round 6: an invalid cell is the record (0, 0, 0, 0) -- tri_points_kernel writes exactly that for a pixel without a point, the staging loop for a
    // cell outside the image -- and a valid one carries w = 1.0f.  So the first pass adds the coordinates as they are (an invalid neighbour adds +0.0,
    // which is what the select added) and counts with w itself (sums of 0.0f / 1.0f up to 441 are exact in f32); the second pass multiplies the
    // difference by (double)w instead of selecting it per coordinate: a finite difference times 0.0 is a zero of either sign, every product with it
    // again, and a double accumulator that started at +0.0 is unchanged by adding either zero.  Same sums, bit for bit; 30 vector instructions per
    // neighbour instead of 37.
    float nf = 0.f;
    double mean[3] = {0, 0, 0};
    for (int dy = 0; dy <= 2 * radius; dy++) {
        if (row - radius + dy < 0 || row - radius + dy >= H) continue;  // rows outside the image hold only invalid cells
        const float4 *rowp = cell + (ly + dy) * TN_RW + lx;
        float qx[WIN], qy[WIN], qz[WIN], qw[WIN];
#pragma unroll
        for (int k = 0; k < WIN; k++) {
            const float4 q = rowp[k];
            qx[k] = q.x;
            qy[k] = q.y;
            qz[k] = q.z;
            qw[k] = q.w;
        }
#pragma unroll
        for (int k = 0; k < WIN; k++) {
            mean[0] += (double)qx[k];
            mean[1] += (double)qy[k];
            mean[2] += (double)qz[k];
            nf += qw[k];
        }
    }
    const int n = (int)nf;
    float normal[3];
    const float *pp = pts + pix * 4;
    if (n >= 3) {
        for (int c = 0; c < 3; c++) mean[c] /= n;
        double cov[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
        for (int dy = 0; dy <= 2 * radius; dy++) {
            if (row - radius + dy < 0 || row - radius + dy >= H) continue;
            const float4 *rowp = cell + (ly + dy) * TN_RW + lx;
            float qx[WIN], qy[WIN], qz[WIN], qw[WIN];
#pragma unroll
            for (int k = 0; k < WIN; k++) {
                const float4 q = rowp[k];
                qx[k] = q.x;
                qy[k] = q.y;
                qz[k] = q.z;
                qw[k] = q.w;
            }
#pragma unroll
            for (int k = 0; k < WIN; k++) {
                // an invalid neighbour (w = 0) contributes d = +-0: every product is a zero and the accumulators are unchanged
                const double wk = (double)qw[k];
                const double d0 = ((double)qx[k] - mean[0]) * wk, d1 = ((double)qy[k] - mean[1]) * wk, d2 = ((double)qz[k] - mean[2]) * wk;
                cov[0][0] += d0 * d0;
                cov[0][1] += d0 * d1;
                cov[0][2] += d0 * d2;
                cov[1][1] += d1 * d1;
                cov[1][2] += d1 * d2;
                cov[2][2] += d2 * d2;
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
        float dot = 0.f;  // uninitialised in the reference (util.cpp:302)
        for (int i = 0; i <= V; i++) {
            const float *ctr = i == 0 ? main_center : pre[i - 1].center;
            double s = 0;
            for (int c = 0; c < 3; c++) s += (double)normal[c] * (double)(ctr[c] - pp[c] / pp[3]);
            dot += (float)(1.0 / s);
        }
        if (dot < 0)
            for (int c = 0; c < 3; c++) normal[c] = -normal[c];
    } else {
        normal[0] = normal[1] = normal[2] = 0.f;
        for (int i = 0; i <= V; i++) {
            const float *ctr = i == 0 ? main_center : pre[i - 1].center;
            float vec[3];
            double vv = 0;
            for (int c = 0; c < 3; c++) {
                vec[c] = ctr[c] - pp[c];  // not dehomogenised: util.cpp:319
                vv += (double)vec[c] * vec[c];
            }
            for (int c = 0; c < 3; c++) normal[c] += (float)(vec[c] / vv);
        }
    }
    const double nn = sqrt((double)normal[0] * normal[0] + (double)normal[1] * normal[1] + (double)normal[2] * normal[2]);
    for (int c = 0; c < 3; c++) normals[pix * 3 + c] = (float)((double)normal[c] * pdf / nn);
}

static void host_center(const float *cam, float *c3)
{
    const int rows[3] = {0, 1, 3};
    double p[3][4];
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 4; c++) p[r][c] = cam[4 * rows[r] + c];
    auto det3 = [&](int c0, int c1, int c2) {
        return p[0][c0] * (p[1][c1] * p[2][c2] - p[1][c2] * p[2][c1]) - p[0][c1] * (p[1][c0] * p[2][c2] - p[1][c2] * p[2][c0]) +
               p[0][c2] * (p[1][c0] * p[2][c1] - p[1][c1] * p[2][c0]);
    };
    const double c[4] = {det3(1, 2, 3), -det3(0, 2, 3), det3(0, 1, 3), -det3(0, 1, 2)};
    const double n = std::sqrt(c[0] * c[0] + c[1] * c[1] + c[2] * c[2] + c[3] * c[3]);
    float t[4];
    for (int i = 0; i < 4; i++) t[i] = (float)(n > 0 ? c[i] / n : c[i]);
    for (int i = 0; i < 3; i++) c3[i] = t[i] / t[3];
}

// triangulatePixels; `flows_hw4` and `depth_hw` are host pointers, or device pointers when on_device (pipeline.hip)

// ---- ordered compaction of the valid pixels (the reference's pixelId order, util.cpp:172,247-248) on the device -------------
constexpr int CP_CHUNK = 2048;  // pixels per workgroup: 256 threads x 8 consecutive pixels

__global__ __launch_bounds__(256) void compact_count(const uint8_t *__restrict__ valid, size_t P, int *__restrict__ block_counts)
{
    __shared__ int wave_sums[4];
    const size_t base = (size_t)blockIdx.x * CP_CHUNK + (size_t)threadIdx.x * 8;
    int c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++)
        if (base + i < P && valid[base + i]) c++;
    for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o, 64);
    if ((threadIdx.x & 63) == 0) wave_sums[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) block_counts[blockIdx.x] = wave_sums[0] + wave_sums[1] + wave_sums[2] + wave_sums[3];
}

// exclusive scan of the block counts by one workgroup; block_counts[nb] receives the total
__global__ __launch_bounds__(256) void compact_scan(int *__restrict__ block_counts, int nb)
{
    __shared__ int part[256];
    const int per = (nb + 255) / 256, first = threadIdx.x * per, last = min(nb, first + per);
    int s = 0;
    for (int i = first; i < last; i++) s += block_counts[i];
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        int run = 0;
        for (int i = 0; i < 256; i++) {
            const int v = part[i];
            part[i] = run;
            run += v;
        }
        block_counts[nb] = run;
    }
    __syncthreads();
    int run = part[threadIdx.x];
    for (int i = first; i < last; i++) {
        const int v = block_counts[i];
        block_counts[i] = run;
        run += v;
    }
}

__global__ __launch_bounds__(256) void compact_scatter(const uint8_t *__restrict__ valid, const float *__restrict__ pts,
                                                       const float *__restrict__ nrm, size_t P,
                                                       const int *__restrict__ block_offsets, float *__restrict__ out7)
{
    __shared__ int thread_off[256];
    const size_t base = (size_t)blockIdx.x * CP_CHUNK + (size_t)threadIdx.x * 8;
    unsigned mask = 0;
#pragma unroll
    for (int i = 0; i < 8; i++)
        if (base + i < P && valid[base + i]) mask |= 1u << i;
    thread_off[threadIdx.x] = __popc(mask);
    __syncthreads();
    if (threadIdx.x < 64) {  // one wavefront scans the 256 per-thread counts, 4 each
        int v[4], s = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            v[i] = thread_off[threadIdx.x * 4 + i];
            s += v[i];
        }
        int incl = s;
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(incl, o, 64);
            if ((int)threadIdx.x >= o) incl += t;
        }
        int run = incl - s;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            thread_off[threadIdx.x * 4 + i] = run;
            run += v[i];
        }
    }
    __syncthreads();
    size_t n = (size_t)block_offsets[blockIdx.x] + (size_t)thread_off[threadIdx.x];
#pragma unroll
    for (int i = 0; i < 8; i++)
        if (mask & (1u << i)) {
            const size_t p = base + i;
            float *o = out7 + n * 7;
            o[0] = pts[4 * p];
            o[1] = pts[4 * p + 1];
            o[2] = pts[4 * p + 2];
            o[3] = pts[4 * p + 3];
            o[4] = nrm[3 * p];
            o[5] = nrm[3 * p + 1];
            o[6] = nrm[3 * p + 2];
            n++;
        }
}

int triangulate_impl(mvs_ctx *ctx, int nviews, const float *const *flows_hw4, bool on_device, const float main_cam[16],
                     const float *side_cams, const float *depth_hw, float *out_points7, int *out_count)
{
    if (!ctx || !main_cam || !depth_hw || !out_points7 || !out_count || nviews < 0 || (nviews > 0 && (!flows_hw4 || !side_cams)))
        return fail(ctx, MVS_EINVAL, "mvs_triangulate: bad arguments");
    if (nviews > TRI_MAXCAM) return fail(ctx, MVS_EINVAL, "mvs_triangulate: at most %d side views", TRI_MAXCAM);
    MVS_HIP(ctx, hipSetDevice(ctx->device));
    const int W = ctx->W, H = ctx->H, V = nviews;
    const size_t P = (size_t)W * H;
    // host precomputation: mainCameraInv (double cofactor inverse -> f32) and the per-camera products
    double Md[16], Mi[16];
    for (int i = 0; i < 16; i++) Md[i] = main_cam[i];
    invert4(Md, Mi);
    float Minv[16];
    for (int i = 0; i < 16; i++) Minv[i] = (float)Mi[i];
    std::vector<CamPre> pre((size_t)(V > 0 ? V : 1));
    for (int i = 0; i < V; i++) {
        const float *C = side_cams + 16 * i;
        CamPre &q = pre[i];
        for (int r = 0; r < 4; r++)
            for (int c = 0; c < 4; c++) {
                double s = 0;
                for (int k = 0; k < 4; k++) s += (double)C[4 * r + k] * (double)Minv[4 * k + c];
                q.CM[4 * r + c] = (float)s;
            }
        for (int r = 0; r < 2; r++)
            for (int c = 0; c < 3; c++) {
                double s = 0;
                for (int k = 0; k < 3; k++) s += (double)C[4 * r + k] * (double)Minv[4 * k + c];
                q.B[3 * r + c] = (float)s;
            }
        for (int r = 0; r < 2; r++) {
            double s = 0;
            for (int k = 0; k < 4; k++) s += (double)C[4 * r + k] * (double)Minv[4 * k + 2];
            q.projDeriv[r] = (float)s;
        }
        for (int c = 0; c < 4; c++) {
            double s = 0;
            for (int k = 0; k < 4; k++) s += (double)C[12 + k] * (double)Minv[4 * k + c];
            q.projW[c] = (float)s;
        }
        host_center(C, q.center);
        q.pad = 0.f;
    }
    float main_center[3];
    host_center(main_cam, main_center);

    // arena: flows (V*4P) | xyz+valid 4P | depth P | grad 2P | pts 4P | pdf P | normals 3P | packed 7P | block counts | valid P bytes | tables
    const size_t flow_floats = on_device ? 0 : (size_t)V * 4 * P;
    const int nb = (int)((P + CP_CHUNK - 1) / CP_CHUNK);
    const size_t floats = flow_floats + P + 2 * P + 4 * P + P + 3 * P + 4 * P + 7 * P + 2 * P + 4 + (size_t)nb + 2;
    const size_t tables = sizeof(CamPre) * pre.size() + sizeof(float) * (16 + 4) + sizeof(float *) * (size_t)(V > 0 ? V : 1);
    int rc = ensure(ctx, ctx->flow_arena, floats * sizeof(float) + P + tables + 256);
    if (rc) return rc;
    // xyz+valid first: its float4 view needs 16-byte alignment, and flow_floats (V * 4P) is a multiple of 4 floats for any P
    float *d_flows = (float *)ctx->flow_arena.ptr, *d_xyz = d_flows + flow_floats, *d_depth = d_xyz + 4 * P, *d_grad = d_depth + P,
          *d_pts = d_grad + 2 * P, *d_pdf = d_pts + 4 * P, *d_nrm = d_pdf + P, *d_packed = d_nrm + 3 * P;
    TriQueued *d_queue = (TriQueued *)(d_packed + 7 * P);  // unconverged pixels of the first Newton pass
    int *d_qcount = (int *)(d_queue + P);
    int *d_counts = d_qcount + 4;
    uint8_t *d_valid = (uint8_t *)(d_counts + nb + 2);
    uintptr_t t = ((uintptr_t)(d_valid + P) + 63) & ~(uintptr_t)63;
    CamPre *d_pre = (CamPre *)t;
    float *d_minv = (float *)(d_pre + pre.size());
    float *d_mc = d_minv + 16;
    const float **d_ptrs = (const float **)(d_mc + 4);
    hipStream_t st = ctx->stream;
    std::vector<const float *> ptrs((size_t)(V > 0 ? V : 1), nullptr);
    for (int i = 0; i < V; i++) {
        if (!flows_hw4[i]) return fail(ctx, MVS_EINVAL, "mvs_triangulate: flows[%d] is null", i);
        if (on_device) {
            ptrs[i] = flows_hw4[i];
        } else {
            MVS_HIP(ctx, hipMemcpyAsync(d_flows + (size_t)i * 4 * P, flows_hw4[i], sizeof(float) * 4 * P, hipMemcpyHostToDevice, st));
            ptrs[i] = d_flows + (size_t)i * 4 * P;
        }
    }
    // the depth map: a device buffer of the caller is read in place (mvs_process_frame: one copy launch fewer), a host buffer goes up
    if (on_device)
        d_depth = const_cast<float *>(depth_hw);
    else
        MVS_HIP(ctx, hipMemcpyAsync(d_depth, depth_hw, sizeof(float) * P, hipMemcpyHostToDevice, st));
    // the small tables are adjacent in the arena (camera records | inverse main matrix | main centre | flow pointers): ONE upload instead of four
    {
        std::vector<unsigned char> host_tables(tables, 0);
        unsigned char *hp = host_tables.data();
        memcpy(hp, pre.data(), sizeof(CamPre) * pre.size());
        hp += sizeof(CamPre) * pre.size();
        memcpy(hp, Minv, sizeof(Minv));
        hp += sizeof(float) * 16;
        memcpy(hp, main_center, sizeof(main_center));
        hp += sizeof(float) * 4;
        memcpy(hp, ptrs.data(), sizeof(float *) * ptrs.size());
        MVS_HIP(ctx, hipMemcpyAsync(d_pre, host_tables.data(), tables, hipMemcpyHostToDevice, st));  // (pageable source: staged before the call returns)
    }
    sobel_kernel<<<dim3(div_up(W, 64), div_up(H, 4)), 256, 0, st>>>(d_depth, W, H, d_grad);
    {
        const dim3 grid(div_up(W, 64), div_up(H, 2));
        MVS_HIP(ctx, hipMemsetAsync(d_qcount, 0, sizeof(int), st));
#define MVS_TRI_POINTS(VM, PH) tri_points_kernel<VM, PH><<<grid, 128, 0, st>>>(d_ptrs, d_pre, V, d_minv, d_depth, d_grad, W, H, d_valid, d_pts, d_xyz, d_pdf, d_qcount, d_queue)
        for (int phase = 0; phase < 2; phase++) {  // phase 1: same grid size, threads beyond the queue length exit at once
            if (V <= 4) {
                if (phase == 0) MVS_TRI_POINTS(4, 0); else MVS_TRI_POINTS(4, 1);
            } else if (V <= 8) {
                if (phase == 0) MVS_TRI_POINTS(8, 0); else MVS_TRI_POINTS(8, 1);
            } else {
                if (phase == 0) MVS_TRI_POINTS(TRI_MAXCAM, 0); else MVS_TRI_POINTS(TRI_MAXCAM, 1);
            }
        }
#undef MVS_TRI_POINTS
    }
    tri_normals_kernel<<<dim3(div_up(W, TN_W), div_up(H, TN_H)), TN_W * TN_H, 0, st>>>(d_valid, d_pts, d_xyz, d_pdf, d_pre, d_mc, V, W, H, d_nrm);
    // compaction in pixel scan order (the reference's pixelId, util.cpp:172,247-248) on the device: only the packed
    // rows cross PCIe, straight into the caller's buffer
    compact_count<<<nb, 256, 0, st>>>(d_valid, P, d_counts);
    compact_scan<<<1, 256, 0, st>>>(d_counts, nb);
    compact_scatter<<<nb, 256, 0, st>>>(d_valid, d_pts, d_nrm, P, d_counts, d_packed);
    MVS_HIP(ctx, hipGetLastError());
    int n = 0;
    MVS_HIP(ctx, hipMemcpyAsync(&n, d_counts + nb, sizeof(int), hipMemcpyDeviceToHost, st));
    MVS_HIP(ctx, hipStreamSynchronize(st));
    if (n > 0) {
        MVS_HIP(ctx, hipMemcpyAsync(out_points7, d_packed, sizeof(float) * 7 * (size_t)n, hipMemcpyDeviceToHost, st));
        MVS_HIP(ctx, hipStreamSynchronize(st));
    }
    *out_count = n;
    return MVS_OK;
}

}  // namespace mvs

using namespace mvs;

extern "C" {

int mvs_triangulate(mvs_ctx *ctx, int nviews, const float *const *flows_hw4, const float main_cam[16], const float *side_cams,
                    const float *depth_hw, float *out_points7, int *out_count)
{
    return triangulate_impl(ctx, nviews, flows_hw4, false, main_cam, side_cams, depth_hw, out_points7, out_count);
}

}  // extern "C"
