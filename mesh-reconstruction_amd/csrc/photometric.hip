// photometric.hip -- compare() and flowRemap() of the reference (util.cpp:332-361, 390-403) on gfx950.
//
// The reference calls OpenCV for the arithmetic (pyrDown / pyrUp / absdiff / remap, CV_INTER_CUBIC);
// these kernels restate those routines (OpenCV 3.x behaviour for CV_32F pyramids and CV_8U remap) with
// the operation order documented in oracle/photometric_oracle.c so both sides agree bit for bit.
// All levels stay in HBM; the reference's Mat temporaries (one allocation per level per call) become
// one arena carved per context.
//
//   pyr_down_kernel    5x5 binomial, REFLECT_101, one thread per coarse pixel (25 taps, L2-resident)
//   pyr_up_add_kernel  zero-insert x2 + [1 4 6 4 1]/8, fused with the `diffPyramid[i] += upscaled` of
//                      util.cpp:357 so the up-sampled level is never materialised
//   absdiff_kernel     |a - b| (util.cpp:343), with the u8 -> f32 conversion of util.cpp:337-338 fused at level 0
//   pyr_down_pair_absdiff   one level of compare(): pyrDown of both images + their absolute difference in one launch
//   pyramid_tail       every level of <= 4096 cells, down and back up, in a single workgroup
//   remap_cubic_kernel 1/32-pixel Q15 bicubic, BORDER_CONSTANT 0 (util.cpp:401)
#include "mvs_internal.hpp"

#include <cmath>

namespace mvs {

__device__ __forceinline__ int reflect101(int p, int n)
{
    if (n == 1) return 0;
    while (p < 0 || p >= n) {
        if (p < 0) p = -p;
        if (p >= n) p = 2 * n - 2 - p;
    }
    return p;
}

// cv::pyrDown value at coarse pixel (x, y) of a w x h source
__device__ __forceinline__ float pyr_down_at(const float *__restrict__ src, int w, int h, int x, int y)
{
    int xs[5];
#pragma unroll
    for (int k = 0; k < 5; k++) xs[k] = reflect101(2 * x + k - 2, w);
    float r[5];
#pragma unroll
    for (int k = 0; k < 5; k++) {
        const float *s = src + (size_t)reflect101(2 * y + k - 2, h) * w;
        r[k] = s[xs[2]] * 6.0f + (s[xs[1]] + s[xs[3]]) * 4.0f + s[xs[0]] + s[xs[4]];
    }
    return (r[2] * 6.0f + (r[1] + r[3]) * 4.0f + r[0] + r[4]) * (1.0f / 256.0f);
}

__global__ __launch_bounds__(256) void pyr_down_kernel(const float *__restrict__ src, int w, int h, float *__restrict__ dst,
                                                       int dw, int dh)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= dw || y >= dh) return;
    dst[(size_t)y * dw + x] = pyr_down_at(src, w, h, x, y);
}

// one level of compare(): pyrDown of both images and their absolute difference (util.cpp:343-350) in one launch
__global__ __launch_bounds__(256) void pyr_down_pair_absdiff(const float *__restrict__ a, const float *__restrict__ b, int w, int h,
                                                             float *__restrict__ da, float *__restrict__ db,
                                                             float *__restrict__ dd, int dw, int dh, ptrdiff_t z_stride = 0)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= dw || y >= dh) return;
    const ptrdiff_t zo = z_stride * blockIdx.z;  // blockIdx.z = image pair of a batch (every pointer lives in that pair's arena)
    a += zo;
    b += zo;
    da += zo;
    db += zo;
    dd += zo;
    const float va = pyr_down_at(a, w, h, x, y), vb = pyr_down_at(b, w, h, x, y);
    const size_t o = (size_t)y * dw + x;
    da[o] = va;
    db[o] = vb;
    dd[o] = fabsf(va - vb);
}

// horizontal pyrUp value at fine column x of coarse row s (sw entries)
__device__ __forceinline__ float up_row(const float *__restrict__ s, int sw, int x)
{
    const int k = x >> 1;
    if (sw == 1) return s[0] * 8.0f;
    if (x & 1) return k < sw - 1 ? (s[k] + s[k + 1]) * 4.0f : s[sw - 1] * 8.0f;
    if (k == 0) return s[0] * 6.0f + s[1] * 2.0f;
    if (k == sw - 1) return s[sw - 2] + s[sw - 1] * 7.0f;
    return s[k - 1] + s[k] * 6.0f + s[k + 1];
}

// cv::pyrUp value at fine pixel (x, y) of a sw x sh source
__device__ __forceinline__ float pyr_up_at(const float *__restrict__ src, int sw, int sh, int x, int y)
{
    const int j = y >> 1;
    const int jm = j > 0 ? j - 1 : (sh > 1 ? 1 : 0);
    const int jp = j < sh - 1 ? j + 1 : sh - 1;
    const float r1 = up_row(src + (size_t)j * sw, sw, x);
    const float r2 = up_row(src + (size_t)jp * sw, sw, x);
    if (y & 1) return (r1 + r2) * 4.0f * (1.0f / 64.0f);
    const float r0 = up_row(src + (size_t)jm * sw, sw, x);
    return (r0 + r1 * 6.0f + r2) * (1.0f / 64.0f);
}

// acc[y][x] += pyrUp(src)[y][x]
__global__ __launch_bounds__(256) void pyr_up_add_kernel(const float *__restrict__ src, int sw, int sh,
                                                         float *__restrict__ acc, int dw, int dh, ptrdiff_t src_z = 0, ptrdiff_t acc_z = 0)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= dw || y >= dh) return;
    src += src_z * blockIdx.z;
    acc += acc_z * blockIdx.z;
    acc[(size_t)y * dw + x] += pyr_up_at(src, sw, sh, x, y);
}

// The small end of compare()'s pyramid in ONE workgroup: from level `first` (already built, <= 4096 cells) down to the
// coarsest level and back up to `first`, i.e. the launches of levels first+1 .. nlev-1 and the pyrUp-adds that end
// in level `first`.  Levels are tiny (40x30 and below at 640x480), so a launch each is pure overhead.  Steps are
// separated by workgroup barriers; every value is computed by the same device functions as the per-level kernels.
struct PyrTail {
    int first, nlev;
    int w[24], h[24];
    unsigned off[24];  // element offset of each level inside the A / B / D arenas
};

__global__ __launch_bounds__(1024) void pyramid_tail(float *__restrict__ A, float *__restrict__ B, float *__restrict__ D, PyrTail t, ptrdiff_t z_stride = 0)
{
    A += z_stride * blockIdx.x;  // one workgroup per image pair of a batch
    B += z_stride * blockIdx.x;
    D += z_stride * blockIdx.x;
    for (int i = t.first + 1; i < t.nlev; i++) {
        const int n = t.w[i] * t.h[i];
        for (int c = threadIdx.x; c < n; c += blockDim.x) {
            const int y = c / t.w[i], x = c - y * t.w[i];
            const float va = pyr_down_at(A + t.off[i - 1], t.w[i - 1], t.h[i - 1], x, y);
            const float vb = pyr_down_at(B + t.off[i - 1], t.w[i - 1], t.h[i - 1], x, y);
            A[t.off[i] + c] = va;
            B[t.off[i] + c] = vb;
            D[t.off[i] + c] = fabsf(va - vb);
        }
        __syncthreads();
    }
    for (int i = t.nlev - 2; i >= t.first; i--) {
        const int n = t.w[i] * t.h[i];
        for (int c = threadIdx.x; c < n; c += blockDim.x) {
            const int y = c / t.w[i], x = c - y * t.w[i];
            D[t.off[i] + c] += pyr_up_at(D + t.off[i + 1], t.w[i + 1], t.h[i + 1], x, y);
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void u8_to_f32_pair_absdiff(const uint8_t *__restrict__ a8, const uint8_t *__restrict__ b8,
                                                              float *__restrict__ a, float *__restrict__ b,
                                                              float *__restrict__ d, size_t n, ptrdiff_t b8_z = 0, ptrdiff_t arena_z = 0, ptrdiff_t d_z = 0)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    b8 += b8_z * blockIdx.y;  // blockIdx.y = image pair of a batch: the first image is common, the second and the outputs advance
    a += arena_z * blockIdx.y;
    b += arena_z * blockIdx.y;
    d += d_z * blockIdx.y;
    const float fa = (float)a8[i], fb = (float)b8[i];
    a[i] = fa;
    b[i] = fb;
    d[i] = fabsf(fa - fb);
}

__global__ __launch_bounds__(256) void absdiff_kernel(const float *__restrict__ a, const float *__restrict__ b,
                                                      float *__restrict__ d, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    d[i] = fabsf(a[i] - b[i]);
}

__global__ __launch_bounds__(256) void remap_cubic_kernel(const float *__restrict__ flow, int stride,
                                                          const uint8_t *__restrict__ img, int W, int H,
                                                          const short *__restrict__ itab, uint8_t *__restrict__ out, ptrdiff_t flow_z = 0, ptrdiff_t img_z = 0)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    flow += flow_z * blockIdx.z;  // blockIdx.z = flow of a batch (image and output W*H bytes apart)
    img += img_z * blockIdx.z;
    out += img_z * blockIdx.z;
    const float *f = flow + ((size_t)y * W + x) * stride;
    const float mx = f[0] + (float)x, my = f[1] + (float)y;
    const int qx = __float2int_rn(mx * 32.0f), qy = __float2int_rn(my * 32.0f);
    int sx = (qx >> 5) - 1, sy = (qy >> 5) - 1;
    const int fx = qx & 31, fy = qy & 31;
    sx = max(-32767, min(32767, sx));
    sy = max(-32767, min(32767, sy));
    const short *wt = itab + (size_t)(fy * 32 + fx) * 16;
    int sum = 0;
#pragma unroll
    for (int k1 = 0; k1 < 4; k1++) {
        const int yy = sy + k1;
        if (yy < 0 || yy >= H) continue;
#pragma unroll
        for (int k2 = 0; k2 < 4; k2++) {
            const int xx = sx + k2;
            if (xx < 0 || xx >= W) continue;
            sum += (int)img[(size_t)yy * W + xx] * (int)wt[k1 * 4 + k2];
        }
    }
    const int v = (sum + (1 << 14)) >> 15;
    out[(size_t)y * W + x] = (uint8_t)max(0, min(255, v));
}

// host: the Q15 bicubic table of OpenCV's initInterTab2D(INTER_CUBIC, fixpt) (a = -0.75)
static void build_cubic_table(short *itab)
{
    float t1[32][4];
    for (int i = 0; i < 32; i++) {
        const float x = (float)i * (1.0f / 32), A = -0.75f;
        float *c = t1[i];
        c[0] = ((A * (x + 1) - 5 * A) * (x + 1) + 8 * A) * (x + 1) - 4 * A;
        c[1] = ((A + 2) * x - (A + 3)) * x * x + 1;
        c[2] = ((A + 2) * (1 - x) - (A + 3)) * (1 - x) * (1 - x) + 1;
        c[3] = 1.f - c[0] - c[1] - c[2];
    }
    for (int i = 0; i < 32; i++)
        for (int j = 0; j < 32; j++) {
            short *it = itab + (size_t)(i * 32 + j) * 16;
            int isum = 0;
            for (int k1 = 0; k1 < 4; k1++)
                for (int k2 = 0; k2 < 4; k2++) {
                    long r = lrintf(t1[i][k1] * t1[j][k2] * 32768.0f);
                    r = r > 32767 ? 32767 : (r < -32768 ? -32768 : r);
                    it[k1 * 4 + k2] = (short)r;
                    isum += (int)r;
                }
            if (isum != 32768) {
                const int diff = isum - 32768;
                int Mk = 2 * 4 + 2, mk = 2 * 4 + 2;
                for (int k1 = 2; k1 < 4; k1++)
                    for (int k2 = 2; k2 < 4; k2++) {
                        const int k = k1 * 4 + k2;
                        if (it[k] < it[mk])
                            mk = k;
                        else if (it[k] > it[Mk])
                            Mk = k;
                    }
                if (diff < 0)
                    it[Mk] = (short)(it[Mk] - diff);
                else
                    it[mk] = (short)(it[mk] - diff);
            }
        }
}

static dim3 grid2d(int w, int h) { return dim3(div_up(w, 64), div_up(h, 4)); }

static size_t compare_total(int W, int H)
{
    int size = H < W ? H : W, w = W, h = H;
    size_t total = 0;
    for (;;) {
        total += (size_t)w * h;
        if (size <= 2) break;
        w = (w + 1) / 2;
        h = (h + 1) / 2;
        size /= 2;
    }
    return total;
}

int compare_prepare(mvs_ctx *ctx, int pairs) { return ensure(ctx, ctx->r_tmp1, sizeof(float) * compare_total(ctx->W, ctx->H) * 3 * (size_t)(pairs < 1 ? 1 : pairs)); }

// compare() on device buffers for B image pairs that share their first image: prev8 (W*H u8), next8 (B x W*H u8) -> out (B x W*H f32).
// Every launch covers all pairs (blockIdx.z / .y = pair; each pair has its own a / b / diff pyramids): the pyramids of a 640 x 480 pair
// are ten launches of ~5 us, and mvs_process_frame's batched Farneback pass used to pay them once per side view (round 6).  All
// in-stream, no sync; B = 1 is compare() itself.
int compare_batch_device(mvs_ctx *ctx, const uint8_t *prev8, const uint8_t *next8, int B, float *out)
{
    const int W = ctx->W, H = ctx->H;
    // level geometry (util.cpp:335,341-351)
    int lw[32], lh[32], nlev = 0;
    {
        int size = H < W ? H : W, w = W, h = H;
        for (;;) {
            lw[nlev] = w;
            lh[nlev] = h;
            nlev++;
            if (size <= 2) break;
            w = (w + 1) / 2;
            h = (h + 1) / 2;
            size /= 2;
        }
    }
    size_t total = 0;
    for (int i = 0; i < nlev; i++) total += (size_t)lw[i] * lh[i];
    // arena per pair: a-pyramid, b-pyramid, diff-pyramid (level 0 of diff is `out`)
    int rc = ensure(ctx, ctx->r_tmp1, sizeof(float) * total * 3 * (size_t)B);
    if (rc) return rc;
    const ptrdiff_t az = (ptrdiff_t)(3 * total);
    float *A = (float *)ctx->r_tmp1.ptr, *Bp = A + total, *D = Bp + total;
    std::vector<size_t> off(nlev);
    size_t o = 0;
    for (int i = 0; i < nlev; i++) {
        off[i] = o;
        o += (size_t)lw[i] * lh[i];
    }
    const size_t P = (size_t)W * H;
    u8_to_f32_pair_absdiff<<<dim3((unsigned)((P + 255) / 256), (unsigned)B), 256, 0, ctx->stream>>>(prev8, next8, A, Bp, out, P, (ptrdiff_t)P, az, (ptrdiff_t)P);
    MVS_HIP(ctx, hipGetLastError());
    // levels of at most 4096 cells (never level 0, whose difference lives in `out`) are finished by one workgroup
    int tail = nlev;
    for (int i = 1; i < nlev; i++)
        if ((size_t)lw[i] * lh[i] <= 4096 && nlev <= 24) {
            tail = i;
            break;
        }
    auto grid = [&](int w, int h) {
        dim3 g = grid2d(w, h);
        g.z = (unsigned)B;
        return g;
    };
    for (int i = 1; i < nlev && i <= tail; i++) {
        pyr_down_pair_absdiff<<<grid(lw[i], lh[i]), 256, 0, ctx->stream>>>(A + off[i - 1], Bp + off[i - 1], lw[i - 1], lh[i - 1],
                                                                           A + off[i], Bp + off[i], D + off[i], lw[i], lh[i], az);
        MVS_HIP(ctx, hipGetLastError());
    }
    int up_from = nlev - 2;
    if (tail < nlev - 1) {
        PyrTail t;
        t.first = tail;
        t.nlev = nlev;
        for (int i = 0; i < nlev; i++) {
            t.w[i] = lw[i];
            t.h[i] = lh[i];
            t.off[i] = (unsigned)off[i];
        }
        pyramid_tail<<<(unsigned)B, 1024, 0, ctx->stream>>>(A, Bp, D, t, az);
        MVS_HIP(ctx, hipGetLastError());
        up_from = tail - 1;
    }
    for (int i = up_from; i >= 0; i--) {
        float *dst = i == 0 ? out : D + off[i];
        pyr_up_add_kernel<<<grid(lw[i], lh[i]), 256, 0, ctx->stream>>>(D + off[i + 1], lw[i + 1], lh[i + 1], dst, lw[i], lh[i], az, i == 0 ? (ptrdiff_t)P : az);
        MVS_HIP(ctx, hipGetLastError());
    }
    return MVS_OK;
}

int compare_device(mvs_ctx *ctx, const uint8_t *prev8, const uint8_t *next8, float *out) { return compare_batch_device(ctx, prev8, next8, 1, out); }

int ensure_cubic_table(mvs_ctx *ctx)
{
    if (ctx->cubic_tab.ptr) return MVS_OK;
    int rc = ensure(ctx, ctx->cubic_tab, sizeof(short) * 1024 * 16);
    if (rc) return rc;
    std::vector<short> tab(1024 * 16);
    build_cubic_table(tab.data());
    MVS_HIP(ctx, hipMemcpyAsync(ctx->cubic_tab.ptr, tab.data(), sizeof(short) * tab.size(), hipMemcpyHostToDevice, ctx->stream));
    MVS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return MVS_OK;
}

// flowRemap on device buffers
// flowRemap for B (flow, image) pairs in one launch: flows `flow_z` floats apart, images and outputs W*H bytes apart (B = 1: flowRemap itself)
int remap_batch_device(mvs_ctx *ctx, const float *flow, int stride, ptrdiff_t flow_z, const uint8_t *img, int B, uint8_t *out)
{
    int rc = ensure_cubic_table(ctx);
    if (rc) return rc;
    dim3 g = grid2d(ctx->W, ctx->H);
    g.z = (unsigned)B;
    remap_cubic_kernel<<<g, 256, 0, ctx->stream>>>(flow, stride, img, ctx->W, ctx->H, (const short *)ctx->cubic_tab.ptr, out, flow_z, (ptrdiff_t)ctx->W * ctx->H);
    MVS_HIP(ctx, hipGetLastError());
    return MVS_OK;
}

int remap_device(mvs_ctx *ctx, const float *flow, int stride, const uint8_t *img, uint8_t *out) { return remap_batch_device(ctx, flow, stride, 0, img, 1, out); }

}  // namespace mvs

using namespace mvs;

// cv::resize(..., Size(dw, dh)) INTER_LINEAR on u8 (configuration.cpp:233; the fixed-point arithmetic is stated in
// oracle/photometric_oracle.c: orc_resize_linear_u8): one thread per destination pixel, all channels
__global__ __launch_bounds__(256) void resize_linear_u8_kernel(const uint8_t *__restrict__ src, int sw, int sh, int channels, uint8_t *__restrict__ dst, int dw, int dh,
                                                               double sx, double sy)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= dw || y >= dh) return;
    auto axis = [](int d, double scale, int ssize, int &s, int &w0, int &w1) {
        float f = (float)((d + 0.5) * scale - 0.5);
        s = (int)floorf(f);
        f -= (float)s;
        if (s < 0) {
            s = 0;
            f = 0.f;
        }
        if (s >= ssize - 1) {
            s = ssize - 1;
            f = 0.f;
        }
        w0 = (short)__float2int_rn((1.f - f) * 2048.f);
        w1 = (short)__float2int_rn(f * 2048.f);
    };
    int x0, a0, a1, y0, b0, b1;
    axis(x, sx, sw, x0, a0, a1);
    axis(y, sy, sh, y0, b0, b1);
    const int x1 = min(x0 + 1, sw - 1), y1 = min(y0 + 1, sh - 1);
    for (int c = 0; c < channels; c++) {
        const int S0 = src[((size_t)y0 * sw + x0) * channels + c] * a0 + src[((size_t)y0 * sw + x1) * channels + c] * a1;
        const int S1 = src[((size_t)y1 * sw + x0) * channels + c] * a0 + src[((size_t)y1 * sw + x1) * channels + c] * a1;
        dst[((size_t)y * dw + x) * channels + c] = (uint8_t)((((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2);
    }
}

extern "C" {

int mvs_resize_u8(mvs_ctx *ctx, const uint8_t *src, int sw, int sh, int channels, uint8_t *dst, int dw, int dh)
{
    if (!ctx || !src || !dst) return fail(ctx, MVS_EINVAL, "mvs_resize_u8: null argument");
    if (sw < 1 || sh < 1 || dw < 1 || dh < 1 || (channels != 1 && channels != 3) || (size_t)sw * sh > ((size_t)1 << 30) || (size_t)dw * dh > ((size_t)1 << 30))
        return fail(ctx, MVS_EINVAL, "mvs_resize_u8: bad sizes (%d x %d x %d -> %d x %d)", sw, sh, channels, dw, dh);
    MVS_HIP(ctx, hipSetDevice(ctx->device));
    const size_t ns = (size_t)sw * sh * channels, nd = (size_t)dw * dh * channels;
    int rc;
    if ((rc = ensure(ctx, ctx->upload, ns + nd + 256))) return rc;
    uint8_t *s = (uint8_t *)ctx->upload.ptr, *d = s + ((ns + 255) & ~(size_t)255);
    MVS_HIP(ctx, hipMemcpyAsync(s, src, ns, hipMemcpyHostToDevice, ctx->stream));
    resize_linear_u8_kernel<<<dim3(div_up(dw, 64), div_up(dh, 4)), 256, 0, ctx->stream>>>(s, sw, sh, channels, d, dw, dh, (double)sw / dw, (double)sh / dh);
    MVS_HIP(ctx, hipGetLastError());
    MVS_HIP(ctx, hipMemcpyAsync(dst, d, nd, hipMemcpyDeviceToHost, ctx->stream));
    MVS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return MVS_OK;
}

int mvs_compare(mvs_ctx *ctx, const uint8_t *prev_hw, const uint8_t *next_hw, float *out_hw)
{
    if (!ctx || !prev_hw || !next_hw || !out_hw) return fail(ctx, MVS_EINVAL, "mvs_compare: null argument");
    MVS_HIP(ctx, hipSetDevice(ctx->device));
    const size_t P = (size_t)ctx->W * ctx->H;
    int rc;
    if ((rc = ensure(ctx, ctx->upload, 2 * P))) return rc;
    if ((rc = ensure(ctx, ctx->r_zbuf, P * sizeof(float)))) return rc;
    uint8_t *a = (uint8_t *)ctx->upload.ptr, *b = a + P;
    MVS_HIP(ctx, hipMemcpyAsync(a, prev_hw, P, hipMemcpyHostToDevice, ctx->stream));
    MVS_HIP(ctx, hipMemcpyAsync(b, next_hw, P, hipMemcpyHostToDevice, ctx->stream));
    if ((rc = compare_device(ctx, a, b, (float *)ctx->r_zbuf.ptr))) return rc;
    MVS_HIP(ctx, hipMemcpyAsync(out_hw, ctx->r_zbuf.ptr, P * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    MVS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return MVS_OK;
}

int mvs_flow_remap(mvs_ctx *ctx, const float *flow, int flow_stride, const uint8_t *image_hw, uint8_t *out_hw)
{
    if (!ctx || !flow || !image_hw || !out_hw) return fail(ctx, MVS_EINVAL, "mvs_flow_remap: null argument");
    if (flow_stride < 2 || flow_stride > 4) return fail(ctx, MVS_EINVAL, "mvs_flow_remap: flow_stride %d not in 2..4", flow_stride);
    MVS_HIP(ctx, hipSetDevice(ctx->device));
    const size_t P = (size_t)ctx->W * ctx->H;
    int rc;
    if ((rc = ensure(ctx, ctx->upload, 2 * P))) return rc;
    if ((rc = ensure(ctx, ctx->r_tmp0, P * sizeof(float) * flow_stride))) return rc;
    uint8_t *img = (uint8_t *)ctx->upload.ptr, *out = img + P;
    MVS_HIP(ctx, hipMemcpyAsync(img, image_hw, P, hipMemcpyHostToDevice, ctx->stream));
    MVS_HIP(ctx, hipMemcpyAsync(ctx->r_tmp0.ptr, flow, P * sizeof(float) * flow_stride, hipMemcpyHostToDevice, ctx->stream));
    if ((rc = remap_device(ctx, (const float *)ctx->r_tmp0.ptr, flow_stride, img, out))) return rc;
    MVS_HIP(ctx, hipMemcpyAsync(out_hw, out, P, hipMemcpyDeviceToHost, ctx->stream));
    MVS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return MVS_OK;
}

}  // extern "C"
