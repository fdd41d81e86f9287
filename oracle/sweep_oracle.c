/*
 * sweep_oracle.c -- CPU restatement of the plane-sweep cost volume + per-pixel depth selection.
 * TEST INFRASTRUCTURE ONLY (see mvs_oracle.h).  PARITY UNPINNED (no reference goldens exist).
 *
 * What it follows in the reference:
 *   shader.frag:13-15   s = sideMVP * vec4(pos, 1)                      -> orc_sweep_sample (s = Q * ndc)
 *   shader.frag:17,22   uv = s.xy / (2 s.w) - 0.5 with GL_REPEAT        -> padded coordinates, wrap padding
 *   shader.frag:19      strict in-frame test |s.x/s.w| < 1, |s.y/s.w| < 1 -> 0.5 < c < size + 0.5
 *   render_glx.cpp:65-88 GL_LINEAR magnification of a GL_RED u8 texture  -> bilinear, level 0 only
 *   render_glx.cpp:356-359 RGB8 read-back (warped intensity quantised to u8) -> Iq = trunc(bilinear + 0.5)
 *   render_glx.cpp:369-397 depth map contract: NDC z, empty = 1.0       -> depth = z[best] or 1.0
 * The D-plane sweep itself is the generalisation described in SURVEY.md section 0.2: the shader's warp
 * evaluated at pos_d = main^-1 * (x_ndc, y_ndc, z_d, 1) for D planes instead of at the mesh position.
 *
 * Arithmetic contract (shared with the HIP kernels): every f32 operation below is a
 * single correctly-rounded IEEE operation (fmaf, *, -, 1/x); compile with -ffp-contract=off.
 */
#include "mvs_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* 4x4 inverse by cofactor expansion, double.  Order of operations is part of the contract. */
static void inv4(const double m[16], double out[16])
{
    double inv[16];
    inv[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] + m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
    inv[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] - m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
    inv[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] + m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
    inv[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] - m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
    inv[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] - m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
    inv[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] + m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
    inv[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] - m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
    inv[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] + m[8] * m[2] * m[13] + m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
    inv[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] + m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
    inv[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] - m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
    inv[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] + m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
    inv[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] - m[4] * m[2] * m[13] - m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
    inv[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] - m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
    inv[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] + m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
    inv[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] - m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
    inv[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] + m[4] * m[2] * m[9] + m[8] * m[1] * m[6] - m[8] * m[2] * m[5];
    double det = m[0] * inv[0] + m[1] * inv[4] + m[2] * inv[8] + m[3] * inv[12];
    double rdet = 1.0 / det;
    for (int i = 0; i < 16; i++) out[i] = inv[i] * rdet;
}

void orc_view_matrix(const float main_cam[16], const float side_cam[16], int W, int H, float Q[12])
{
    double M[16], Mi[16], C[16], T[16];
    for (int i = 0; i < 16; i++) {
        M[i] = main_cam[i];
        C[i] = side_cam[i];
    }
    inv4(M, Mi);
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) {
            double s = 0.0;
            for (int k = 0; k < 4; k++) s += C[4 * r + k] * Mi[4 * k + c];
            T[4 * r + c] = s;
        }
    /* padded-pixel scaling: cx_p = (ndc_x/2 + 1/2) * W - 1/2 + 1, cy_p = (1/2 - ndc_y/2) * H - 1/2 + 1
     * (texel centres + vertical flip: SURVEY Appendix A-7; +1 = wrap padding) */
    const double hw = 0.5 * W, hh = 0.5 * H;
    for (int c = 0; c < 4; c++) {
        Q[0 + c] = (float)(hw * T[0 + c] + (hw + 0.5) * T[12 + c]);
        Q[4 + c] = (float)(-hh * T[4 + c] + (hh + 0.5) * T[12 + c]);
        Q[8 + c] = (float)(T[12 + c]);
    }
}

void orc_pad_image(const uint8_t *img, int W, int H, uint8_t *pad, int pitch)
{
    for (int r = 0; r < H + 2; r++) {
        int sr = (r - 1 + H) % H;
        uint8_t *d = pad + (size_t)r * pitch;
        const uint8_t *s = img + (size_t)sr * W;
        d[0] = s[W - 1];
        memcpy(d + 1, s, W);
        d[W + 1] = s[0];
        for (int c = W + 2; c < pitch; c++) d[c] = 0;
    }
}

void orc_plane_table(int D, float z_lo, float z_hi, float *z)
{
    for (int d = 0; d < D; d++)
        z[d] = (float)((double)z_lo + ((double)z_hi - (double)z_lo) * ((double)d + 0.5) / (double)D);
}

float orc_pixel_xn(int col, int W)
{
    const float invW = 1.0f / (float)W;
    return fmaf((float)(2 * col + 1), invW, -1.0f);
}

float orc_pixel_yn(int row, int H)
{
    const float invH = 1.0f / (float)H;
    return fmaf(-(float)(2 * row + 1), invH, 1.0f);
}

/* per-pixel, per-view affine-in-z form: s(z) = A + z*B */
static inline void view_affine(const float Q[12], float xn, float yn, float A[3], float B[3])
{
    for (int r = 0; r < 3; r++) {
        A[r] = fmaf(Q[4 * r + 0], xn, fmaf(Q[4 * r + 1], yn, Q[4 * r + 3]));
        B[r] = Q[4 * r + 2];
    }
}

static inline int sample_affine(const float A[3], const float B[3], float z, const uint8_t *pad, int pitch,
                                int W, int H, int *Iq)
{
    const float sx = fmaf(z, B[0], A[0]);
    const float sy = fmaf(z, B[1], A[1]);
    const float sw = fmaf(z, B[2], A[2]);
    if (!(sw > 0.0f)) return 0;
    const float r = 1.0f / sw;
    const float cx = sx * r;
    const float cy = sy * r;
    /* strict in-frame test of shader.frag:19 in padded pixel units */
    if (!(cx > 0.5f && cx < (float)W + 0.5f && cy > 0.5f && cy < (float)H + 0.5f)) return 0;
    const int ix = (int)cx, iy = (int)cy;
    const float ax = cx - (float)ix, ay = cy - (float)iy; /* exact: cx, cy > 0 */
    const uint8_t *p = pad + (size_t)iy * pitch + ix;
    const float t00 = (float)p[0], t01 = (float)p[1], t10 = (float)p[pitch], t11 = (float)p[pitch + 1];
    /* bilinear in polynomial form; the tap differences are exact small integers */
    const float dxt = t01 - t00;
    const float dy = t10 - t00;
    const float dxy = (t11 - t10) - dxt;
    /* the +0.5 of the u8 rounding rides on the (exact) t00 term: Iq = trunc(bilinear + 0.5) */
    const float a = fmaf(ax, dxt, t00 + 0.5f);
    const float b = fmaf(ax, dxy, dy);
    const float res = fmaf(ay, b, a);
    *Iq = (int)res;
    return 1;
}

int orc_sweep_sample(const float Q[12], float xn, float yn, float z, const uint8_t *pad, int pitch,
                     int W, int H, int *Iq)
{
    float A[3], B[3];
    view_affine(Q, xn, yn, A, B);
    return sample_affine(A, B, z, pad, pitch, W, H, Iq);
}

void orc_argmin(const uint32_t *volume, int W, int H, int D, const float *z,
                float *depth, float *best_cost, int32_t *best_idx)
{
    const size_t P = (size_t)W * H;
    for (size_t p = 0; p < P; p++) {
        uint32_t bs = 0, bc = 0;
        int bi = -1;
        for (int d = 0; d < D; d++) {
            const uint32_t cell = volume[(size_t)d * P + p];
            const uint32_t s = cell & 0xffffu, c = cell >> 16;
            if (c == 0) continue;
            /* s/c < bs/bc  <=>  s*bc < bs*c  (exact in 64-bit; strict: ties keep the lowest d) */
            if (bi < 0 || (uint64_t)s * bc < (uint64_t)bs * c) {
                bs = s;
                bc = c;
                bi = d;
            }
        }
        depth[p] = bi >= 0 ? z[bi] : ORC_BACKGROUND_DEPTH;
        if (best_cost) best_cost[p] = bi >= 0 ? (float)bs / (float)bc : INFINITY;
        if (best_idx) best_idx[p] = bi;
    }
}

void orc_sweep(const float main_cam[16], const uint8_t *main_img, int W, int H,
               int V, const float *side_cams, const uint8_t *const *side_imgs,
               int D, float z_lo, float z_hi,
               uint32_t *volume, float *depth, float *best_cost, int32_t *best_idx, int nthreads)
{
    const int pitch = W + 2;
    const size_t P = (size_t)W * H;
    float *Q = (float *)malloc(sizeof(float) * 12 * (size_t)(V > 0 ? V : 1));
    float *z = (float *)malloc(sizeof(float) * (size_t)D);
    uint8_t **pads = (uint8_t **)malloc(sizeof(uint8_t *) * (size_t)(V > 0 ? V : 1));
    for (int v = 0; v < V; v++) {
        orc_view_matrix(main_cam, side_cams + 16 * v, W, H, Q + 12 * v);
        pads[v] = (uint8_t *)malloc((size_t)pitch * (H + 2));
        orc_pad_image(side_imgs[v], W, H, pads[v], pitch);
    }
    orc_plane_table(D, z_lo, z_hi, z);
    if (nthreads < 1) nthreads = 1;

#ifdef _OPENMP
#pragma omp parallel for num_threads(nthreads) schedule(dynamic, 4)
#endif
    for (int row = 0; row < H; row++) {
        uint32_t *acc = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)D);
        const float yn = orc_pixel_yn(row, H);
        for (int col = 0; col < W; col++) {
            const float xn = orc_pixel_xn(col, W);
            const int Im = main_img[(size_t)row * W + col];
            memset(acc, 0, sizeof(uint32_t) * (size_t)D);
            for (int v = 0; v < V; v++) {
                float A[3], B[3];
                view_affine(Q + 12 * v, xn, yn, A, B);
                for (int d = 0; d < D; d++) {
                    int Iq;
                    if (sample_affine(A, B, z[d], pads[v], pitch, W, H, &Iq)) {
                        const int diff = Iq > Im ? Iq - Im : Im - Iq;
                        acc[d] += (1u << 16) + (uint32_t)diff;
                    }
                }
            }
            const size_t p = (size_t)row * W + col;
            uint32_t bs = 0, bc = 0;
            int bi = -1;
            for (int d = 0; d < D; d++) {
                const uint32_t s = acc[d] & 0xffffu, c = acc[d] >> 16;
                if (volume) volume[(size_t)d * P + p] = acc[d];
                if (c == 0) continue;
                if (bi < 0 || (uint64_t)s * bc < (uint64_t)bs * c) {
                    bs = s;
                    bc = c;
                    bi = d;
                }
            }
            depth[p] = bi >= 0 ? z[bi] : ORC_BACKGROUND_DEPTH;
            if (best_cost) best_cost[p] = bi >= 0 ? (float)bs / (float)bc : INFINITY;
            if (best_idx) best_idx[p] = bi;
        }
        free(acc);
    }
    for (int v = 0; v < V; v++) free(pads[v]);
    free(pads);
    free(z);
    free(Q);
}

/* The parity hook of SURVEY.md section 0.2: the sweep's sampler evaluated at ONE plane per pixel, z = depth[p] (e.g.
 * Render::depth of the main camera).  out_hw2 = (warped u8, mask 255/0) per pixel; background pixels (depth == 1.0)
 * and out-of-frame samples get (0, 0).  Equals Render::projected without its shadow test. */
void orc_warp_by_depth(const float main_cam[16], const float *depth, const float side_cam[16], const uint8_t *frame,
                       int W, int H, uint8_t *out_hw2)
{
    const int pitch = W + 2;
    float Q[12];
    orc_view_matrix(main_cam, side_cam, W, H, Q);
    uint8_t *pad = (uint8_t *)malloc((size_t)pitch * (H + 2));
    orc_pad_image(frame, W, H, pad, pitch);
    for (int row = 0; row < H; row++)
        for (int col = 0; col < W; col++) {
            const size_t p = (size_t)row * W + col;
            int Iq = 0, ok = 0;
            if (depth[p] != ORC_BACKGROUND_DEPTH)
                ok = orc_sweep_sample(Q, orc_pixel_xn(col, W), orc_pixel_yn(row, H), depth[p], pad, pitch, W, H, &Iq);
            out_hw2[2 * p] = ok ? (uint8_t)Iq : 0;
            out_hw2[2 * p + 1] = ok ? 255 : 0;
        }
    free(pad);
}


/* ==================================================================================================================
 * Sampler "fixed" (arithmetic contract v2): the same warp, with the texture unit modelled the way
 * fixed-function samplers and OpenCV's own remap (INTER_BITS = 5, the fixed-point path the reference's flowRemap runs
 * through cv::remap, util.cpp:401) work: the sampling position is quantised to 1/32 texel and the four bilinear weights
 * come from a 32 x 32 table of 8-bit integers that sum to 255.  The warped intensity is kept at the table's precision
 * (dot = sum of weight * texel, in [0, 65025] = 255 * grey level) instead of being re-quantised to u8 per view; the cost
 * cell is  count << 24 | sum |dot - 255 * I_main|  (integer, order independent, all-reducible; <= 255 views).
 *   s, r          as in the exact sampler above (s = fma(z, B, A), r = RN(1 / s.w))
 *   u             = RNE(s.xy * (256 r) + 4)      one rounding of the real product: 1/256 texel units, +4 = half of 1/32
 *   in frame      <=> s.w > 0  and  132 < ux < 256 W + 132  and  132 < uy < 256 H + 132
 *   i, k          = u >> 8, (u >> 3) & 31          texel and 1/32 sub-texel position
 *   dot           = w[ky][kx] . (t00, t01, t10, t11)
 * ================================================================================================================== */
void orc_fx_weight_table(uint8_t *lut /* 32*32*4 */)
{
    for (int b = 0; b < 32; b++)
        for (int a = 0; a < 32; a++) {
            const int p[4] = {(32 - a) * (32 - b), a * (32 - b), (32 - a) * b, a * b}; /* sum 1024 */
            int w[4], sum = 0, big = 0;
            for (int k = 0; k < 4; k++) {
                w[k] = (255 * p[k] + 512) >> 10;
                sum += w[k];
                if (p[k] > p[big]) big = k; /* first of the largest */
            }
            w[big] += 255 - sum;
            for (int k = 0; k < 4; k++) lut[((b * 32 + a) << 2) + k] = (uint8_t)w[k];
        }
}

#define ORC_FX_MAGIC 12582912.0f /* 1.5 * 2^23: floats in [2^23, 2^24) have ulp 1 */

static inline int sample_affine_fx(const float A[3], const float B[3], float z, const uint8_t *pad, int pitch, int W, int H,
                                   const uint8_t *lut, int *dot)
{
    const float sx = fmaf(z, B[0], A[0]);
    const float sy = fmaf(z, B[1], A[1]);
    const float sw = fmaf(z, B[2], A[2]);
    if (!(sw > 0.0f)) return 0;
    const float r = 1.0f / sw;
    const float r256 = r * 256.0f; /* exact */
    /* RNE(product + 4) by the magic-number addition: one rounding, to an integer (the sum lies in [2^23, 2^24)) */
    const float tx = fmaf(sx, r256, ORC_FX_MAGIC + 4.0f);
    const float ty = fmaf(sy, r256, ORC_FX_MAGIC + 4.0f);
    if (!(tx > ORC_FX_MAGIC + 132.0f && tx < ORC_FX_MAGIC + 132.0f + 256.0f * (float)W && ty > ORC_FX_MAGIC + 132.0f &&
          ty < ORC_FX_MAGIC + 132.0f + 256.0f * (float)H))
        return 0;
    const int ux = (int)(tx - ORC_FX_MAGIC), uy = (int)(ty - ORC_FX_MAGIC); /* exact */
    const int ix = ux >> 8, iy = uy >> 8, kx = (ux >> 3) & 31, ky = (uy >> 3) & 31;
    const uint8_t *p = pad + (size_t)iy * pitch + ix;
    const uint8_t *w = lut + ((ky * 32 + kx) << 2);
    *dot = w[0] * p[0] + w[1] * p[1] + w[2] * p[pitch] + w[3] * p[pitch + 1];
    return 1;
}

int orc_sweep_sample_fx(const float Q[12], float xn, float yn, float z, const uint8_t *pad, int pitch, int W, int H, int *dot)
{
    static uint8_t lut[4096];
    static int have = 0;
    if (!have) {
        orc_fx_weight_table(lut);
        have = 1;
    }
    float A[3], B[3];
    view_affine(Q, xn, yn, A, B);
    return sample_affine_fx(A, B, z, pad, pitch, W, H, lut, dot);
}

void orc_argmin_fx(const uint32_t *volume, int W, int H, int D, const float *z, float *depth, float *best_cost, int32_t *best_idx)
{
    const size_t P = (size_t)W * H;
    for (size_t p = 0; p < P; p++) {
        uint32_t bs = 0, bc = 0;
        int bi = -1;
        for (int d = 0; d < D; d++) {
            const uint32_t cell = volume[(size_t)d * P + p];
            const uint32_t s = cell & 0xffffffu, c = cell >> 24;
            if (c == 0) continue;
            if (bi < 0 || (uint64_t)s * bc < (uint64_t)bs * c) {
                bs = s;
                bc = c;
                bi = d;
            }
        }
        depth[p] = bi >= 0 ? z[bi] : ORC_BACKGROUND_DEPTH;
        if (best_cost) best_cost[p] = bi >= 0 ? (float)bs / (float)(255u * bc) : INFINITY;
        if (best_idx) best_idx[p] = bi;
    }
}

void orc_sweep_fx(const float main_cam[16], const uint8_t *main_img, int W, int H, int V, const float *side_cams,
                  const uint8_t *const *side_imgs, int D, float z_lo, float z_hi, uint32_t *volume, float *depth, float *best_cost,
                  int32_t *best_idx, int nthreads)
{
    const int pitch = W + 2;
    const size_t P = (size_t)W * H;
    float *Q = (float *)malloc(sizeof(float) * 12 * (size_t)(V > 0 ? V : 1));
    float *z = (float *)malloc(sizeof(float) * (size_t)D);
    uint8_t **pads = (uint8_t **)malloc(sizeof(uint8_t *) * (size_t)(V > 0 ? V : 1));
    uint8_t lut[4096];
    orc_fx_weight_table(lut);
    for (int v = 0; v < V; v++) {
        orc_view_matrix(main_cam, side_cams + 16 * v, W, H, Q + 12 * v);
        pads[v] = (uint8_t *)malloc((size_t)pitch * (H + 2));
        orc_pad_image(side_imgs[v], W, H, pads[v], pitch);
    }
    orc_plane_table(D, z_lo, z_hi, z);
    if (nthreads < 1) nthreads = 1;
#ifdef _OPENMP
#pragma omp parallel for num_threads(nthreads) schedule(dynamic, 4)
#endif
    for (int row = 0; row < H; row++) {
        uint32_t *acc = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)D);
        const float yn = orc_pixel_yn(row, H);
        for (int col = 0; col < W; col++) {
            const float xn = orc_pixel_xn(col, W);
            const int Im255 = 255 * (int)main_img[(size_t)row * W + col];
            memset(acc, 0, sizeof(uint32_t) * (size_t)D);
            for (int v = 0; v < V; v++) {
                float A[3], B[3];
                view_affine(Q + 12 * v, xn, yn, A, B);
                for (int d = 0; d < D; d++) {
                    int dot;
                    if (sample_affine_fx(A, B, z[d], pads[v], pitch, W, H, lut, &dot)) {
                        const int diff = dot > Im255 ? dot - Im255 : Im255 - dot;
                        acc[d] += (1u << 24) + (uint32_t)diff;
                    }
                }
            }
            const size_t p = (size_t)row * W + col;
            uint32_t bs = 0, bc = 0;
            int bi = -1;
            for (int d = 0; d < D; d++) {
                const uint32_t s = acc[d] & 0xffffffu, c = acc[d] >> 24;
                if (volume) volume[(size_t)d * P + p] = acc[d];
                if (c == 0) continue;
                if (bi < 0 || (uint64_t)s * bc < (uint64_t)bs * c) {
                    bs = s;
                    bc = c;
                    bi = d;
                }
            }
            depth[p] = bi >= 0 ? z[bi] : ORC_BACKGROUND_DEPTH;
            if (best_cost) best_cost[p] = bi >= 0 ? (float)bs / (float)(255u * bc) : INFINITY;
            if (best_idx) best_idx[p] = bi;
        }
        free(acc);
    }
    for (int v = 0; v < V; v++) free(pads[v]);
    free(pads);
    free(z);
    free(Q);
}

/* the fixed sampler at one plane per pixel: out_hw2 = (round(dot / 255), mask) */
void orc_warp_by_depth_fx(const float main_cam[16], const float *depth, const float side_cam[16], const uint8_t *frame, int W, int H,
                          uint8_t *out_hw2)
{
    const int pitch = W + 2;
    float Q[12];
    orc_view_matrix(main_cam, side_cam, W, H, Q);
    uint8_t *pad = (uint8_t *)malloc((size_t)pitch * (H + 2));
    orc_pad_image(frame, W, H, pad, pitch);
    for (int row = 0; row < H; row++)
        for (int col = 0; col < W; col++) {
            const size_t p = (size_t)row * W + col;
            int dot = 0, ok = 0;
            if (depth[p] != ORC_BACKGROUND_DEPTH)
                ok = orc_sweep_sample_fx(Q, orc_pixel_xn(col, W), orc_pixel_yn(row, H), depth[p], pad, pitch, W, H, &dot);
            out_hw2[2 * p] = ok ? (uint8_t)((dot + 127) / 255) : 0;
            out_hw2[2 * p + 1] = ok ? 255 : 0;
        }
    free(pad);
}

/* util.cpp:366-387 */
void orc_mix_background(const uint8_t *img_hw3, const uint8_t *bg_hw, float *depth_hw, uint8_t *out_hw,
                        int W, int H)
{
    const size_t P = (size_t)W * H;
    for (size_t i = 0; i < P; i++) {
        if (depth_hw[i] == ORC_BACKGROUND_DEPTH || !img_hw3[3 * i + 1]) {
            out_hw[i] = bg_hw[i];
            depth_hw[i] = ORC_BACKGROUND_DEPTH;
        } else {
            out_hw[i] = img_hw3[3 * i];
        }
    }
}


/* ---- sub-plane refinement of the selected depth (SURVEY.md section 7.2 K6 "optional sub-plane parabola refinement") --------------
 * The plane sweep quantises depth to the plane step; two samplers (or two runs with slightly different costs) that pick neighbouring
 * planes of nearly equal cost then differ by a whole step.  The usual remedy: fit a parabola through the mean costs of the selected
 * plane and its two neighbours and move to its vertex.  Defined here in f32, one rounding per operation (the HIP kernel
 * refine_depth mirrors it bit for bit):
 *   c_k = (float)sum_k / (float)count_k          (fixed sampler: / (255 count_k): the same scale for all three, it cancels)
 *   den = (c_- - 2 c_0) + c_+ ;  refined only if 0 < i < D - 1, both neighbours have a view in frame, and den > 0
 *   t   = 0.5 (c_- - c_+) / den, clamped to [-0.5, 0.5]
 *   z   = t >= 0 ? fma(t, z_{i+1} - z_i, z_i) : fma(-t, z_{i-1} - z_i, z_i)      (planes need not be equidistant)
 * cs = 16 (exact sampler's cells) or 24 (fixed sampler's). */
void orc_refine_depth(const uint32_t *volume, int W, int H, int D, const float *z, const int32_t *idx, int cs, float *depth)
{
    const size_t P = (size_t)W * H;
    const uint32_t M = (1u << cs) - 1u;
    for (size_t p = 0; p < P; p++) {
        const int i = idx[p];
        if (i < 0) {
            depth[p] = 1.0f; /* backgroundDepth */
            continue;
        }
        float zr = z[i];
        if (i > 0 && i < D - 1) {
            const uint32_t a = volume[(size_t)(i - 1) * P + p], b = volume[(size_t)i * P + p], c = volume[(size_t)(i + 1) * P + p];
            if ((a >> cs) != 0u && (c >> cs) != 0u && (b >> cs) != 0u) {
                const float ca = (float)(a & M) / (float)(a >> cs), cb = (float)(b & M) / (float)(b >> cs), cc = (float)(c & M) / (float)(c >> cs);
                const float den = (ca - 2.0f * cb) + cc;
                if (den > 0.0f) {
                    float t = (0.5f * (ca - cc)) / den;
                    t = t < -0.5f ? -0.5f : (t > 0.5f ? 0.5f : t);
                    zr = t >= 0.0f ? fmaf(t, z[i + 1] - z[i], z[i]) : fmaf(-t, z[i - 1] - z[i], z[i]);
                }
            }
        }
        depth[p] = zr;
    }
}
