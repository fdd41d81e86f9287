/*
 * photometric_oracle.c -- CPU restatement of compare() and flowRemap() (util.cpp:332-361, 390-403).
 * TEST INFRASTRUCTURE ONLY (see mvs_oracle.h).  PARITY UNPINNED.
 *
 * The arithmetic of both functions lives in OpenCV (imgproc: pyrDown, pyrUp, absdiff, remap), which is
 * not in /root/reference and not installed here; the reference pins no version (Makefile:10-13 links
 * whatever `opencv_imgproc` is installed; the API mix implies OpenCV 3.x).  This file restates the
 * published algorithms of those routines as OpenCV 3.x implements them for CV_32F / CV_8U data:
 *
 *   pyrDown  5x5 binomial [1 4 6 4 1]/16 per axis, BORDER_REFLECT_101, dst = ((w+1)/2, (h+1)/2);
 *            row pass  r[x] = s[2x]*6 + (s[2x-1] + s[2x+1])*4 + s[2x-2] + s[2x+2]
 *            col pass  d    = (r2*6 + (r1 + r3)*4 + r0 + r4) * (1/256)
 *   pyrUp    zero-insertion x2 then [1 4 6 4 1]/8 per axis, to an explicit dsize in {2s, 2s-1};
 *            even: s[x-1] + s[x]*6 + s[x+1], odd: (s[x] + s[x+1])*4; left border reflects (s[-1] = s[1]),
 *            right border replicates (s[w] = s[w-1]); result * (1/64)
 *   remap    CV_INTER_CUBIC on CV_8U with float maps: positions quantised to 1/32 pixel
 *            (cvRound(map*32)), 4x4 taps with the a = -0.75 cubic, weights in Q15 fixed point forced
 *            to sum to 32768, BORDER_CONSTANT 0, result (sum + 2^14) >> 15 saturated to u8.
 * compare() itself (the level loop, `size /= 2`, the coarse-to-fine accumulation) is followed literally.
 */
#include "mvs_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

static inline int reflect101(int p, int n)
{
    if (n == 1) return 0;
    while (p < 0 || p >= n) {
        if (p < 0) p = -p;
        if (p >= n) p = 2 * n - 2 - p;
    }
    return p;
}

/* src (w x h) -> dst ((w+1)/2 x (h+1)/2) */
static void pyr_down(const float *src, int w, int h, float *dst)
{
    const int dw = (w + 1) / 2, dh = (h + 1) / 2;
    float *rows = (float *)malloc(sizeof(float) * (size_t)dw * 5);
    for (int y = 0; y < dh; y++) {
        for (int k = 0; k < 5; k++) {
            const int sy = reflect101(2 * y + k - 2, h);
            const float *s = src + (size_t)sy * w;
            float *r = rows + (size_t)k * dw;
            for (int x = 0; x < dw; x++) {
                const float s0 = s[reflect101(2 * x - 2, w)], s1 = s[reflect101(2 * x - 1, w)], s2 = s[reflect101(2 * x, w)],
                            s3 = s[reflect101(2 * x + 1, w)], s4 = s[reflect101(2 * x + 2, w)];
                r[x] = s2 * 6.0f + (s1 + s3) * 4.0f + s0 + s4;
            }
        }
        float *d = dst + (size_t)y * dw;
        const float *r0 = rows, *r1 = rows + dw, *r2 = rows + 2 * dw, *r3 = rows + 3 * dw, *r4 = rows + 4 * dw;
        for (int x = 0; x < dw; x++) d[x] = (r2[x] * 6.0f + (r1[x] + r3[x]) * 4.0f + r0[x] + r4[x]) * (1.0f / 256.0f);
    }
    free(rows);
}

/* horizontal pyrUp pass of one source row into a row of 2*sw entries (the caller uses the first dw) */
static void pyr_up_row(const float *s, int sw, float *r)
{
    if (sw == 1) {
        r[0] = s[0] * 8.0f;
        r[1] = s[0] * 8.0f;
        return;
    }
    r[0] = s[0] * 6.0f + s[1] * 2.0f;
    r[1] = (s[0] + s[1]) * 4.0f;
    for (int x = 1; x < sw - 1; x++) {
        r[2 * x] = s[x - 1] + s[x] * 6.0f + s[x + 1];
        r[2 * x + 1] = (s[x] + s[x + 1]) * 4.0f;
    }
    r[2 * (sw - 1)] = s[sw - 2] + s[sw - 1] * 7.0f;
    r[2 * (sw - 1) + 1] = s[sw - 1] * 8.0f;
}

/* src (sw x sh) -> dst (dw x dh), dw in {2sw, 2sw-1}, dh in {2sh, 2sh-1} */
static void pyr_up(const float *src, int sw, int sh, float *dst, int dw, int dh)
{
    float *rows = (float *)malloc(sizeof(float) * (size_t)sw * 2 * 3);
    float *r0 = rows, *r1 = rows + 2 * sw, *r2 = rows + 4 * sw;
    for (int y = 0; y < sh; y++) {
        /* source rows y-1, y, y+1 with the vertical border rule of the row pass: top reflects, bottom replicates */
        const int ym = y > 0 ? y - 1 : (sh > 1 ? 1 : 0);
        const int yp = y < sh - 1 ? y + 1 : sh - 1;
        pyr_up_row(src + (size_t)ym * sw, sw, r0);
        pyr_up_row(src + (size_t)y * sw, sw, r1);
        pyr_up_row(src + (size_t)yp * sw, sw, r2);
        float *d0 = dst + (size_t)(2 * y) * dw;
        const int y1 = 2 * y + 1;
        for (int x = 0; x < dw; x++) {
            const float t1 = (r1[x] + r2[x]) * 4.0f * (1.0f / 64.0f);
            const float t0 = (r0[x] + r1[x] * 6.0f + r2[x]) * (1.0f / 64.0f);
            if (y1 < dh) dst[(size_t)y1 * dw + x] = t1;
            d0[x] = t0;
        }
    }
    free(rows);
}

/* exposed for the independent cross-checks in tests/test_pinning_cpu.py */
void orc_pyr_down(const float *src, int w, int h, float *dst) { pyr_down(src, w, h, dst); }
void orc_pyr_up(const float *src, int sw, int sh, float *dst, int dw, int dh) { pyr_up(src, sw, sh, dst, dw, dh); }

void orc_compare_f32(const float *prev, const float *next, int W, int H, float *out)
{
    /* util.cpp:334-351 */
    int size = H < W ? H : W;
    int nlev = 0;
    float *diff[32];
    int lw[32], lh[32];
    float *a = (float *)malloc(sizeof(float) * (size_t)W * H), *b = (float *)malloc(sizeof(float) * (size_t)W * H);
    memcpy(a, prev, sizeof(float) * (size_t)W * H);
    memcpy(b, next, sizeof(float) * (size_t)W * H);
    int w = W, h = H;
    for (;;) {
        float *d = (float *)malloc(sizeof(float) * (size_t)w * h);
        for (size_t i = 0; i < (size_t)w * h; i++) d[i] = fabsf(a[i] - b[i]);
        diff[nlev] = d;
        lw[nlev] = w;
        lh[nlev] = h;
        nlev++;
        if (size <= 2) break;
        const int nw = (w + 1) / 2, nh = (h + 1) / 2;
        float *a2 = (float *)malloc(sizeof(float) * (size_t)nw * nh), *b2 = (float *)malloc(sizeof(float) * (size_t)nw * nh);
        pyr_down(a, w, h, a2);
        pyr_down(b, w, h, b2);
        free(a);
        free(b);
        a = a2;
        b = b2;
        w = nw;
        h = nh;
        size /= 2;
    }
    free(a);
    free(b);
    /* util.cpp:354-358 */
    for (int i = nlev - 2; i >= 0; i--) {
        float *up = (float *)malloc(sizeof(float) * (size_t)lw[i] * lh[i]);
        pyr_up(diff[i + 1], lw[i + 1], lh[i + 1], up, lw[i], lh[i]);
        for (size_t k = 0; k < (size_t)lw[i] * lh[i]; k++) diff[i][k] += up[k];
        free(up);
    }
    memcpy(out, diff[0], sizeof(float) * (size_t)W * H);
    for (int i = 0; i < nlev; i++) free(diff[i]);
}

void orc_compare_u8(const uint8_t *prev, const uint8_t *next, int W, int H, float *out)
{
    const size_t P = (size_t)W * H;
    float *a = (float *)malloc(sizeof(float) * P), *b = (float *)malloc(sizeof(float) * P);
    for (size_t i = 0; i < P; i++) { /* convertTo(CV_32FC1), util.cpp:337-338 */
        a[i] = (float)prev[i];
        b[i] = (float)next[i];
    }
    orc_compare_f32(a, b, W, H, out);
    free(a);
    free(b);
}

/* ---- remap, CV_INTER_CUBIC, 8-bit ----------------------------------------------------------- */

#define TAB 32
#define COEF_SCALE 32768

static void cubic_coeffs(float x, float *c)
{
    const float A = -0.75f;
    c[0] = ((A * (x + 1) - 5 * A) * (x + 1) + 8 * A) * (x + 1) - 4 * A;
    c[1] = ((A + 2) * x - (A + 3)) * x * x + 1;
    c[2] = ((A + 2) * (1 - x) - (A + 3)) * (1 - x) * (1 - x) + 1;
    c[3] = 1.f - c[0] - c[1] - c[2];
}

static short g_itab[TAB * TAB][16];
static int g_itab_ready = 0;

static short sat_short(float v)
{
    long r = lrintf(v);
    if (r > 32767) r = 32767;
    if (r < -32768) r = -32768;
    return (short)r;
}

void orc_remap_cubic_table(short *out /* 1024*16, nullable */)
{
    if (!g_itab_ready) {
        float t1[TAB][4];
        for (int i = 0; i < TAB; i++) cubic_coeffs((float)i * (1.0f / TAB), t1[i]);
        for (int i = 0; i < TAB; i++)
            for (int j = 0; j < TAB; j++) {
                short *it = g_itab[i * TAB + j];
                int isum = 0;
                for (int k1 = 0; k1 < 4; k1++)
                    for (int k2 = 0; k2 < 4; k2++) {
                        const float v = t1[i][k1] * t1[j][k2];
                        it[k1 * 4 + k2] = sat_short(v * COEF_SCALE);
                        isum += it[k1 * 4 + k2];
                    }
                if (isum != COEF_SCALE) {
                    /* push the rounding residue into the largest (or smallest) tap of the 2x2 block at
                     * taps (2..3, 2..3), as OpenCV's initInterTab2D does */
                    const int diff = isum - COEF_SCALE;
                    int Mk1 = 2, Mk2 = 2, mk1 = 2, mk2 = 2;
                    for (int k1 = 2; k1 < 4; k1++)
                        for (int k2 = 2; k2 < 4; k2++) {
                            if (it[k1 * 4 + k2] < it[mk1 * 4 + mk2]) {
                                mk1 = k1;
                                mk2 = k2;
                            } else if (it[k1 * 4 + k2] > it[Mk1 * 4 + Mk2]) {
                                Mk1 = k1;
                                Mk2 = k2;
                            }
                        }
                    if (diff < 0)
                        it[Mk1 * 4 + Mk2] = (short)(it[Mk1 * 4 + Mk2] - diff);
                    else
                        it[mk1 * 4 + mk2] = (short)(it[mk1 * 4 + mk2] - diff);
                }
            }
        g_itab_ready = 1;
    }
    if (out) memcpy(out, g_itab, sizeof(g_itab));
}

void orc_flow_remap(const float *flow, int stride, const uint8_t *image, int W, int H, uint8_t *out)
{
    orc_remap_cubic_table(NULL);
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            const float *f = flow + ((size_t)y * W + x) * stride;
            /* util.cpp:395-398: absolute map = flow + (x, y), in f32 */
            const float mx = f[0] + (float)x, my = f[1] + (float)y;
            /* convertMaps: cvRound(m * 32) (round half to even), then split into integer and 5-bit fraction */
            const int qx = (int)lrintf(mx * (float)TAB), qy = (int)lrintf(my * (float)TAB);
            int sx = (qx >> 5) - 1, sy = (qy >> 5) - 1;
            /* coordinates are carried as short in OpenCV's fixed-point maps */
            const int fx = qx & (TAB - 1), fy = qy & (TAB - 1);
            if (sx < -32768 + 1) sx = -32768 + 1;
            if (sx > 32767) sx = 32767;
            if (sy < -32768 + 1) sy = -32768 + 1;
            if (sy > 32767) sy = 32767;
            const short *wt = g_itab[fy * TAB + fx];
            int sum = 0;
            for (int k1 = 0; k1 < 4; k1++) {
                const int yy = sy + k1;
                if (yy < 0 || yy >= H) continue; /* BORDER_CONSTANT, value 0 */
                for (int k2 = 0; k2 < 4; k2++) {
                    const int xx = sx + k2;
                    if (xx < 0 || xx >= W) continue;
                    sum += (int)image[(size_t)yy * W + xx] * wt[k1 * 4 + k2];
                }
            }
            int v = (sum + (1 << 14)) >> 15;
            out[(size_t)y * W + x] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
        }
}


/* ---- cv::resize(src, dst, Size(w, h)) with INTER_LINEAR on 8-bit images (configuration.cpp:233) ------------------------------------
 * The reference calls cv::resize(frame, frames[fi], cv::Size(width, height), CV_INTER_AREA): the fourth POSITIONAL argument of
 * cv::resize is fx, not the interpolation, so CV_INTER_AREA (= 3) lands in a scale factor that is ignored because dsize is given,
 * and the interpolation stays at its default, INTER_LINEAR.  OpenCV's 8-bit path is fixed point (imgproc/resize.cpp, the classic
 * HResizeLinear + VResizeLinear<uchar, int, short, FixedPtCast<int, uchar, 22>> pair): source coordinate (d + 0.5) scale - 0.5,
 * clamped at both ends; weights cvRound((1 - f) 2048), cvRound(f 2048) as shorts; horizontal pass in int; vertical pass
 * (((b0 (S0 >> 4)) >> 16) + ((b1 (S1 >> 4)) >> 16) + 2) >> 2.  OpenCV is not in the tree (version unpinned): restated from the
 * published source; tests/test_pinning_cpu.py checks it against the plain float formula. */
static void resize_axis(int dsize, int ssize, int *idx, short *w0, short *w1)
{
    const double scale = (double)ssize / dsize;
    for (int d = 0; d < dsize; d++) {
        float f = (float)((d + 0.5) * scale - 0.5);
        int s = (int)floorf(f);
        f -= (float)s;
        if (s < 0) {
            s = 0;
            f = 0.f;
        }
        if (s >= ssize - 1) {
            s = ssize - 1;
            f = 0.f;
        }
        idx[d] = s;
        /* saturate_cast<short>(x) = cvRound: round half to even, like lrintf in the default rounding mode */
        w0[d] = (short)lrintf((1.f - f) * 2048.f);
        w1[d] = (short)lrintf(f * 2048.f);
    }
}

void orc_resize_linear_u8(const uint8_t *src, int sw, int sh, int channels, uint8_t *dst, int dw, int dh)
{
    int *xi = (int *)malloc(sizeof(int) * dw), *yi = (int *)malloc(sizeof(int) * dh);
    short *xa = (short *)malloc(sizeof(short) * 2 * dw), *ya = (short *)malloc(sizeof(short) * 2 * dh);
    resize_axis(dw, sw, xi, xa, xa + dw);
    resize_axis(dh, sh, yi, ya, ya + dh);
    for (int y = 0; y < dh; y++) {
        const int y0 = yi[y], y1 = y0 + 1 < sh ? y0 + 1 : sh - 1;
        const int b0 = ya[y], b1 = ya[dh + y];
        for (int x = 0; x < dw; x++) {
            const int x0 = xi[x], x1 = x0 + 1 < sw ? x0 + 1 : sw - 1;
            const int a0 = xa[x], a1 = xa[dw + x];
            for (int c = 0; c < channels; c++) {
                const int S0 = src[((size_t)y0 * sw + x0) * channels + c] * a0 + src[((size_t)y0 * sw + x1) * channels + c] * a1;
                const int S1 = src[((size_t)y1 * sw + x0) * channels + c] * a0 + src[((size_t)y1 * sw + x1) * channels + c] * a1;
                dst[((size_t)y * dw + x) * channels + c] = (uint8_t)((((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2);
            }
        }
    }
    free(xi);
    free(yi);
    free(xa);
    free(ya);
}
