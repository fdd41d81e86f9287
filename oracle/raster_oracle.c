/*
 * raster_oracle.c -- CPU restatement of the reference's off-screen renderer.
 * TEST INFRASTRUCTURE ONLY (see mvs_oracle.h).  PARITY UNPINNED: the reference's renderer is the
 * OpenGL driver (render_glx.cpp + shader.vert/frag); there is no GL here, so this follows the OpenGL
 * 3.0 rasterisation rules the reference relies on.
 *
 *   orc_load_mesh        render_glx.cpp:230-258  dehomogenise (242-244) + triangle soup
 *   orc_raster_window_z  render_glx.cpp:276-286 / 370-391: glClear(depth = 1), GL_LESS (196-197), glDrawArrays
 *   orc_depth            render_glx.cpp:369-397  flip + `2*result - 1` (392-395)
 *   orc_shadow_dilate    render_glx.cpp:287-314  literal transcription of the in-place loop, quirks included
 *   orc_projected        render_glx.cpp:261-367  + shader.vert:9-13 + shader.frag:11-25
 *
 * Rasterisation contract (shared with raster.hip; DESIGN.md "raster arithmetic"):
 *   - vertex shader: clip = M * (v,1), each row fma(M0,x, fma(M1,y, fma(M2,z, M3)))        (shader.vert:11)
 *   - coverage by 2-D homogeneous edge functions E_i(p) = a_i px + b_i py + c_i built from the cofactors
 *     of [x_i y_i w_i]: no clipping is needed for triangles that cross w = 0 (the face cameras of
 *     heuristic.cpp:193-247 sit ON the mesh with near = 0.001);
 *   - a pixel centre is inside iff all three sign-normalised E_i > 0, or == 0 on an edge with
 *     (a > 0 || (a == 0 && b > 0)) -- one consistent tie rule, as the GL spec demands;
 *   - z_ndc = plane(px, py) (linear in screen space), fragment kept iff -1 <= z_ndc <= 1 (clip volume),
 *     window z = fma(0.5, z_ndc, 0.5), depth test GL_LESS against 1.0; equal z keeps the earlier face;
 *   - varyings perspective-correct: pos = sum(E_i P_i) / sum(E_i).
 * Depth-buffer quantisation (typically 24-bit fixed point) is NOT modelled: window z stays f32.
 */
#include "mvs_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

void orc_load_mesh(const float *verts4, int nverts, const int32_t *faces3, int nfaces, float *out_soup)
{
    (void)nverts;
    for (int f = 0; f < nfaces; f++)
        for (int j = 0; j < 3; j++) {
            const float *p = verts4 + 4 * (size_t)faces3[3 * f + j];
            out_soup[9 * f + 3 * j + 0] = p[0] / p[3];
            out_soup[9 * f + 3 * j + 1] = p[1] / p[3];
            out_soup[9 * f + 3 * j + 2] = p[2] / p[3];
        }
}

typedef struct {
    float a[3], b[3], c[3]; /* sign-normalised edge functions */
    float za, zb, zc;       /* z_ndc plane */
    int valid;
    int x0, x1, y0, y1; /* conservative pixel bounding box (inclusive) */
} TriSetup;

static inline float xform(const float *m, float x, float y, float z)
{
    return fmaf(m[0], x, fmaf(m[1], y, fmaf(m[2], z, m[3])));
}

/* a*b - c*d with one fused step: fma(a, b, -(c*d)) */
static inline float dop(float a, float b, float c, float d)
{
    const float t = c * d;
    return fmaf(a, b, -t);
}

static void tri_setup(const float *v /* 9 floats */, const float *cam, int W, int H, TriSetup *t)
{
    float x[3], y[3], z[3], w[3];
    for (int i = 0; i < 3; i++) {
        const float px = v[3 * i], py = v[3 * i + 1], pz = v[3 * i + 2];
        x[i] = xform(cam + 0, px, py, pz);
        y[i] = xform(cam + 4, px, py, pz);
        z[i] = xform(cam + 8, px, py, pz);
        w[i] = xform(cam + 12, px, py, pz);
    }
    float a[3], b[3], c[3];
    a[0] = dop(y[1], w[2], w[1], y[2]);
    b[0] = dop(w[1], x[2], x[1], w[2]);
    c[0] = dop(x[1], y[2], y[1], x[2]);
    a[1] = dop(w[0], y[2], y[0], w[2]);
    b[1] = dop(x[0], w[2], w[0], x[2]);
    c[1] = dop(y[0], x[2], x[0], y[2]);
    a[2] = dop(y[0], w[1], w[0], y[1]);
    b[2] = dop(w[0], x[1], x[0], w[1]);
    c[2] = dop(x[0], y[1], y[0], x[1]);
    float det = fmaf(x[0], a[0], fmaf(y[0], b[0], w[0] * c[0]));
    t->valid = (det != 0.0f) && (det == det) && isfinite(det);
    if (!t->valid) return;
    if (det < 0.0f) {
        det = -det;
        for (int i = 0; i < 3; i++) {
            a[i] = -a[i];
            b[i] = -b[i];
            c[i] = -c[i];
        }
    }
    const float rdet = 1.0f / det;
    for (int i = 0; i < 3; i++) {
        t->a[i] = a[i];
        t->b[i] = b[i];
        t->c[i] = c[i];
    }
    t->za = fmaf(a[0], z[0], fmaf(a[1], z[1], a[2] * z[2])) * rdet;
    t->zb = fmaf(b[0], z[0], fmaf(b[1], z[1], b[2] * z[2])) * rdet;
    t->zc = fmaf(c[0], z[0], fmaf(c[1], z[1], c[2] * z[2])) * rdet;
    /* bounding box: exact only when every vertex is in front of the camera */
    t->x0 = 0;
    t->y0 = 0;
    t->x1 = W - 1;
    t->y1 = H - 1;
    if (w[0] > 0.0f && w[1] > 0.0f && w[2] > 0.0f) {
        float xmin = 1e30f, xmax = -1e30f, ymin = 1e30f, ymax = -1e30f;
        for (int i = 0; i < 3; i++) {
            const float nx = x[i] / w[i], ny = y[i] / w[i];
            xmin = fminf(xmin, nx);
            xmax = fmaxf(xmax, nx);
            ymin = fminf(ymin, ny);
            ymax = fmaxf(ymax, ny);
        }
        /* pixel col has xn = (2 col + 1)/W - 1  ->  col = ((xn + 1) W - 1)/2; one pixel of slack */
        const float cx0 = ((xmin + 1.0f) * (float)W - 1.0f) * 0.5f - 1.0f;
        const float cx1 = ((xmax + 1.0f) * (float)W - 1.0f) * 0.5f + 1.0f;
        const float ry0 = ((1.0f - ymax) * (float)H - 1.0f) * 0.5f - 1.0f;
        const float ry1 = ((1.0f - ymin) * (float)H - 1.0f) * 0.5f + 1.0f;
        if (cx0 > 0.0f) t->x0 = cx0 < (float)W ? (int)cx0 : W;
        if (cx1 < (float)(W - 1)) t->x1 = cx1 >= 0.0f ? (int)cx1 : -1;
        if (ry0 > 0.0f) t->y0 = ry0 < (float)H ? (int)ry0 : H;
        if (ry1 < (float)(H - 1)) t->y1 = ry1 >= 0.0f ? (int)ry1 : -1;
    }
}

static inline int edge_inside(float e, float a, float b)
{
    return e > 0.0f || (e == 0.0f && (a > 0.0f || (a == 0.0f && b > 0.0f)));
}

/* returns 1 and the window z if pixel (xn, yn) is covered by t and inside the clip volume */
static inline int tri_fragment(const TriSetup *t, float xn, float yn, float *zwin, float e[3])
{
    for (int i = 0; i < 3; i++) {
        e[i] = fmaf(t->a[i], xn, fmaf(t->b[i], yn, t->c[i]));
        if (!edge_inside(e[i], t->a[i], t->b[i])) return 0;
    }
    const float zn = fmaf(t->za, xn, fmaf(t->zb, yn, t->zc));
    if (!(zn >= -1.0f && zn <= 1.0f)) return 0;
    *zwin = fmaf(0.5f, zn, 0.5f);
    return 1;
}

/* z-buffer render; zwin_td top-down, id_td (nullable) = index of the visible face or -1 */
static void raster_td(const float *soup, int nfaces, const float cam[16], int W, int H, float *zwin_td, int32_t *id_td)
{
    const size_t P = (size_t)W * H;
    for (size_t i = 0; i < P; i++) zwin_td[i] = 1.0f; /* glClear(GL_DEPTH_BUFFER_BIT), default clear depth */
    if (id_td)
        for (size_t i = 0; i < P; i++) id_td[i] = -1;
    for (int f = 0; f < nfaces; f++) {
        TriSetup t;
        tri_setup(soup + 9 * (size_t)f, cam, W, H, &t);
        if (!t.valid) continue;
        for (int row = t.y0; row <= t.y1; row++) {
            const float yn = orc_pixel_yn(row, H);
            for (int col = t.x0; col <= t.x1; col++) {
                const float xn = orc_pixel_xn(col, W);
                float zw, e[3];
                if (!tri_fragment(&t, xn, yn, &zw, e)) continue;
                const size_t p = (size_t)row * W + col;
                if (zw < zwin_td[p]) { /* GL_LESS */
                    zwin_td[p] = zw;
                    if (id_td) id_td[p] = f;
                }
            }
        }
    }
}

void orc_raster_window_z(const float *soup, int nfaces, const float cam[16], int W, int H, float *zwin_gl)
{
    float *td = (float *)malloc(sizeof(float) * (size_t)W * H);
    raster_td(soup, nfaces, cam, W, H, td, NULL);
    for (int r = 0; r < H; r++) memcpy(zwin_gl + (size_t)r * W, td + (size_t)(H - 1 - r) * W, sizeof(float) * (size_t)W);
    free(td);
}

void orc_depth(const float *soup, int nfaces, const float cam[16], int W, int H, float *depth_td)
{
    raster_td(soup, nfaces, cam, W, H, depth_td, NULL);
    const size_t P = (size_t)W * H;
    for (size_t i = 0; i < P; i++) depth_td[i] = fmaf(2.0f, depth_td[i], -1.0f); /* render_glx.cpp:395 */
}

/* render_glx.cpp:287-314, transcribed statement by statement (shadow is in GL orientation: row 0 = bottom) */
void orc_shadow_dilate(float *shadow, int W, int H)
{
    float *prevRowHF = (float *)malloc(sizeof(float) * (size_t)W);
    float *curRow = shadow, *prevRow;
    for (int j = 1; j < W - 1; j++) {
        prevRowHF[j] = curRow[j];
        if (curRow[j - 1] < prevRowHF[j]) prevRowHF[j] = curRow[j - 1];
        if (curRow[j + 1] < prevRowHF[j]) prevRowHF[j] = curRow[j + 1];
        curRow[j] = prevRowHF[j];
    }
    for (int i = 1; i < H; i++) {
        prevRow = curRow;
        curRow = shadow + (size_t)i * W;
        float prevVal = curRow[0];
        for (int j = 1; j < W - 1; j++) {
            float val = curRow[j];
            if (prevVal > curRow[j]) curRow[j] = prevVal;
            if (curRow[j + 1] > curRow[j]) curRow[j] = curRow[j + 1];
            float valHF = curRow[j];
            if (prevRowHF[j] > curRow[j]) curRow[j] = prevRowHF[j];
            if (curRow[j] > prevRow[j]) prevRow[j] = curRow[j];
            prevRowHF[j] = valHF;
            prevVal = val;
        }
    }
    free(prevRowHF);
}

/* ---- mip chain of the frame texture (render_glx.cpp:83-85: GL_LINEAR_MIPMAP_LINEAR + glGenerateMipmap) ----------------------
 * The contract this project states for what the reference leaves to the OpenGL driver:
 *   levels    level l has max(1, w >> 1) x max(1, h >> 1) texels of level l - 1's w x h; texel (i, j) = (a + b + c + d + 2) >> 2 over the
 *             2 x 2 block at (2 i, 2 j) of level l - 1 (indices clamped to its last row / column), kept as u8 like a GL_R8 level
 *   footprint fine derivatives on the pixel's 2 x 2 quad with the pixel's OWN face: partner pixels col ^ 1 and row ^ 1,
 *             rho = sqrt(max(dudx^2 + dvdx^2, dudy^2 + dvdy^2)) in level-0 texels (GL 3.0 eq. 3.21, isotropic: the reference's
 *             request for maximal anisotropy, render_glx.cpp:79-80, is implementation-defined and not modelled)
 *   lod       rho <= 1: magnification, level 0 GL_LINEAR (the level-0 path unchanged).  Otherwise l0 = floor(log2 rho) from the
 *             exponent and the fraction f = rho 2^-l0 - 1 (the piecewise-linear log2 of texture hardware; exact in f32), both levels
 *             sampled bilinearly with GL_REPEAT, result fma(f, s1 - s0, s0); past the last level: that level alone
 * Every operation is spelled out so that the HIP kernel (raster.hip: project_texture) reproduces it bit for bit. */
#define ORC_MAX_MIPS 16
typedef struct {
    int levels;                      /* levels above 0 */
    int w[ORC_MAX_MIPS], h[ORC_MAX_MIPS], pitch[ORC_MAX_MIPS];
    uint8_t *pad[ORC_MAX_MIPS];      /* wrap-padded level images, [0] = the caller's level-0 padded frame */
} MipChain;

static void mip_build(const uint8_t *pad0, int W, int H, int pitch0, MipChain *mc)
{
    mc->levels = 0;
    mc->w[0] = W;
    mc->h[0] = H;
    mc->pitch[0] = pitch0;
    mc->pad[0] = (uint8_t *)pad0;
    int l = 0;
    while ((mc->w[l] > 1 || mc->h[l] > 1) && l + 1 < ORC_MAX_MIPS) {
        const int pw = mc->w[l], ph = mc->h[l], w = pw > 1 ? pw >> 1 : 1, h = ph > 1 ? ph >> 1 : 1;
        uint8_t *tight = (uint8_t *)malloc((size_t)w * h);
        const uint8_t *src = mc->pad[l];
        const int sp = mc->pitch[l];
        for (int j = 0; j < h; j++)
            for (int i = 0; i < w; i++) {
                const int x0 = 2 * i < pw ? 2 * i : pw - 1, x1 = 2 * i + 1 < pw ? 2 * i + 1 : pw - 1;
                const int y0 = 2 * j < ph ? 2 * j : ph - 1, y1 = 2 * j + 1 < ph ? 2 * j + 1 : ph - 1;
                const int sum = src[(size_t)(y0 + 1) * sp + x0 + 1] + src[(size_t)(y0 + 1) * sp + x1 + 1] + src[(size_t)(y1 + 1) * sp + x0 + 1] +
                                src[(size_t)(y1 + 1) * sp + x1 + 1];
                tight[(size_t)j * w + i] = (uint8_t)((sum + 2) >> 2);
            }
        l++;
        mc->w[l] = w;
        mc->h[l] = h;
        mc->pitch[l] = w + 2;
        mc->pad[l] = (uint8_t *)malloc((size_t)(w + 2) * (h + 2));
        orc_pad_image(tight, w, h, mc->pad[l], w + 2);
        free(tight);
    }
    mc->levels = l;
}

static void mip_free(MipChain *mc)
{
    for (int l = 1; l <= mc->levels; l++) free(mc->pad[l]);
}

/* GL_LINEAR with GL_REPEAT at level l, texture coordinates (u, 1 - v) in [0, 1): the level-0 arithmetic of SURVEY A-7 */
static float mip_bilinear(const MipChain *mc, int l, float u, float vv)
{
    const float cx = fmaf(u, (float)mc->w[l], 0.5f);
    const float cy = fmaf(1.0f - vv, (float)mc->h[l], 0.5f);
    const int ix = (int)cx, iy = (int)cy;
    const float ax = cx - (float)ix, ay = cy - (float)iy;
    const uint8_t *q = mc->pad[l] + (size_t)iy * mc->pitch[l] + ix;
    const int pitch = mc->pitch[l];
    const float t00 = (float)q[0], t01 = (float)q[1], t10 = (float)q[pitch], t11 = (float)q[pitch + 1];
    const float dxt = t01 - t00, dy = t10 - t00, dxy = (t11 - t10) - dxt;
    return fmaf(ay, fmaf(ax, dxy, dy), fmaf(ax, dxt, t00));
}

/* texture coordinates of face (t, v) at NDC pixel centre (xn, yn), the face extrapolated past its edges (what a GPU's helper
 * invocations do for the finite differences) */
static void face_uv(const TriSetup *t, const float *v, const float *projector, float xn, float yn, float *u, float *vv)
{
    float e[3];
    for (int i = 0; i < 3; i++) e[i] = fmaf(t->a[i], xn, fmaf(t->b[i], yn, t->c[i]));
    const float esum = (e[0] + e[1]) + e[2];
    float pos[3];
    for (int k = 0; k < 3; k++) pos[k] = fmaf(e[0], v[k], fmaf(e[1], v[3 + k], e[2] * v[6 + k])) / esum;
    const float sx = xform(projector + 0, pos[0], pos[1], pos[2]);
    const float sy = xform(projector + 4, pos[0], pos[1], pos[2]);
    const float sw = xform(projector + 12, pos[0], pos[1], pos[2]);
    *u = fmaf(0.5f, sx / sw, 0.5f);
    *vv = fmaf(0.5f, sy / sw, 0.5f);
}

void orc_projected(const float *soup, int nfaces, const float cam[16], const uint8_t *frame,
                   const float projector[16], int W, int H, uint8_t *out_hw3)
{
    orc_projected_filter(soup, nfaces, cam, frame, projector, W, H, 1, out_hw3);
}

/* mipmap != 0: the frame texture is sampled with the mip chain above (the reference's request); 0: level 0 only */
void orc_projected_filter(const float *soup, int nfaces, const float cam[16], const uint8_t *frame,
                          const float projector[16], int W, int H, int mipmap, uint8_t *out_hw3)
{
    const size_t P = (size_t)W * H;
    float *shadow = (float *)malloc(sizeof(float) * P);
    float *zmain = (float *)malloc(sizeof(float) * P);
    int32_t *id = (int32_t *)malloc(sizeof(int32_t) * P);
    const int pitch = W + 2;
    uint8_t *pad = (uint8_t *)malloc((size_t)pitch * (H + 2));

    /* pass 1: shadow map from the projector (both MVPs = projector, render_glx.cpp:265-266), read back in GL
     * orientation (286), dilated on the CPU (287-314), re-uploaded NEAREST/REPEAT (317-325) */
    orc_raster_window_z(soup, nfaces, projector, W, H, shadow);
    orc_shadow_dilate(shadow, W, H);
    /* frame texture: GL_RED u8, REPEAT, bilinear at level 0 (65-88) */
    orc_pad_image(frame, W, H, pad, pitch);
    MipChain mc;
    mip_build(pad, W, H, pitch, &mc);
    /* pass 2: main camera, colour + depth cleared (336) */
    raster_td(soup, nfaces, cam, W, H, zmain, id);

    const float fW = (float)W, fH = (float)H;
    for (int row = 0; row < H; row++) {
        const float yn = orc_pixel_yn(row, H);
        for (int col = 0; col < W; col++) {
            const size_t p = (size_t)row * W + col;
            uint8_t *o = out_hw3 + 3 * p;
            o[0] = o[1] = o[2] = 0; /* glClearColor(0,0,0,0), render_glx.cpp:195 */
            if (id[p] < 0) continue;
            const float xn = orc_pixel_xn(col, W);
            TriSetup t;
            const float *v = soup + 9 * (size_t)id[p];
            tri_setup(v, cam, W, H, &t);
            float zw, e[3];
            if (!tri_fragment(&t, xn, yn, &zw, e)) continue; /* cannot happen: id came from the same test */
            /* perspective-correct varying `pos` (shader.vert:12) */
            const float esum = (e[0] + e[1]) + e[2];
            float pos[3];
            for (int k = 0; k < 3; k++) pos[k] = fmaf(e[0], v[k], fmaf(e[1], v[3 + k], e[2] * v[6 + k])) / esum;
            /* shader.frag:13-15 */
            const float sx = xform(projector + 0, pos[0], pos[1], pos[2]);
            const float sy = xform(projector + 4, pos[0], pos[1], pos[2]);
            const float sz = xform(projector + 8, pos[0], pos[1], pos[2]);
            const float sw = xform(projector + 12, pos[0], pos[1], pos[2]);
            const float nx = sx / sw, ny = sy / sw, nz = sz / sw;
            /* shader.frag:19 */
            const int inframe = nx > -1.0f && nx < 1.0f && ny > -1.0f && ny < 1.0f;
            if (!inframe) continue;
            /* uv = s.xy/(2 s.w) - 0.5 == ndc/2 + 0.5 under GL_REPEAT (shader.frag:17,22) */
            const float u = fmaf(0.5f, nx, 0.5f), vv = fmaf(0.5f, ny, 0.5f);
            /* shadow texture: NEAREST, REPEAT, GL orientation (render_glx.cpp:319-325) */
            int si = (int)floorf(u * fW), sj = (int)floorf(vv * fH);
            si = ((si % W) + W) % W;
            sj = ((sj % H) + H) % H;
            const float shadowDepth = fmaf(2.0f, shadow[(size_t)sj * W + si], -1.0f);
            const int visible = shadowDepth + 0.01f > nz; /* shader.frag:18 */
            if (!visible) continue;
            /* frame texture: bilinear at padded coordinates (SURVEY A-7): col_f + 1, row_f + 1 */
            const float cx = fmaf(u, fW, 0.5f);
            const float cy = fmaf(1.0f - vv, fH, 0.5f);
            const int ix = (int)cx, iy = (int)cy;
            const float ax = cx - (float)ix, ay = cy - (float)iy;
            const uint8_t *q = pad + (size_t)iy * pitch + ix;
            const float t00 = (float)q[0], t01 = (float)q[1], t10 = (float)q[pitch], t11 = (float)q[pitch + 1];
            const float dxt = t01 - t00, dy = t10 - t00, dxy = (t11 - t10) - dxt;
            float res = fmaf(ay, fmaf(ax, dxy, dy), fmaf(ax, dxt, t00));
            if (mipmap && mc.levels > 0) {
                /* footprint of the pixel in level-0 texels: finite differences on its 2 x 2 quad, same face */
                float ux, vx, uy, vy;
                face_uv(&t, v, projector, orc_pixel_xn(col ^ 1, W), yn, &ux, &vx);
                face_uv(&t, v, projector, xn, orc_pixel_yn(row ^ 1, H), &uy, &vy);
                const float dudx = (ux - u) * fW, dvdx = (vx - vv) * fH, dudy = (uy - u) * fW, dvdy = (vy - vv) * fH;
                const float rx = dudx * dudx + dvdx * dvdx, ry = dudy * dudy + dvdy * dvdy;
                const float rho = sqrtf(rx > ry ? rx : ry);
                if (rho > 1.0f && rho < 3.0e38f) {
                    uint32_t bits;
                    memcpy(&bits, &rho, 4);
                    int l0 = (int)((bits >> 23) & 0xffu) - 127;
                    /* rho 2^-l0 - 1: the significand's fraction, exact */
                    uint32_t mb = (bits & 0x007fffffu) | 0x3f800000u;
                    float f;
                    memcpy(&f, &mb, 4);
                    f = f - 1.0f;
                    if (l0 >= mc.levels) {
                        res = mip_bilinear(&mc, mc.levels, u, vv);
                    } else {
                        const float s0 = l0 == 0 ? res : mip_bilinear(&mc, l0, u, vv), s1 = mip_bilinear(&mc, l0 + 1, u, vv);
                        res = fmaf(f, s1 - s0, s0);
                    }
                }
            }
            o[0] = (uint8_t)(int)(res + 0.5f); /* RGB8 framebuffer write */
            o[1] = o[2] = 255;                 /* shader.frag:24 */
        }
    }
    mip_free(&mc);
    free(pad);
    free(id);
    free(zmain);
    free(shadow);
}
