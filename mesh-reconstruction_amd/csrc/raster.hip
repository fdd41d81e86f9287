// raster.hip -- the reference's off-screen renderer as HIP kernels for gfx950.
//
// Replaces (one kernel per GL draw / transfer of SURVEY.md section 2.1):
//   Render::loadMesh   render_glx.cpp:230-258   -> mvs_load_mesh   (dehomogenised triangle soup in HBM)
//   Render::depth      render_glx.cpp:369-397   -> tri_setup + raster_tiles
//   Render::projected  render_glx.cpp:261-367   -> tri_setup + raster_tiles (shadow pass, GL orientation)
//                                                  -> row0_prefix_min + shadow_dilate (the CPU loop 287-314)
//                                                  -> tri_setup + raster_tiles (main pass) -> project_texture
//                                                  (shader.vert:9-13, shader.frag:11-25)
//   mixBackground      util.cpp:366-387         -> mix_background
// No readbacks between the passes: the reference's two glReadPixels + two uploads per (main, side)
// pair (render_glx.cpp:286,325,359,75) become HBM-resident buffers.
//
// Arithmetic contract: identical to oracle/raster_oracle.c (homogeneous edge functions, one tie rule,
// f32 window z, perspective-correct `pos`); built with -ffp-contract=off so results are bit-exact.
//
// Rasterisation strategy: no atomics.  One 256-thread workgroup owns a 16x16 pixel tile; triangles
// are culled against the tile 256 at a time (one bounding box per lane, survivors compacted through
// LDS), then every thread walks the survivor list for its own pixel with wave-uniform (SGPR) triangle
// records.  Visibility is resolved in registers in submission order, exactly like GL_LESS.
#include "mvs_internal.hpp"

namespace mvs {

struct TriRec {          // 64 bytes
    float a[3], b[3], c[3];
    float za, zb, zc;
    int x0y0;            // packed int16 bbox (inclusive); x0 > x1 marks an invalid triangle
    int x1y1;
    int pad0, pad1;
};

// Face binning for large meshes.  raster_tiles lets every 16x16 tile scan every face -- O(tiles x faces): fine for the proxy
// meshes of the first iteration (a depth map of 32 k faces: 0.16 ms) but 2.0 ms at 0.5 M faces and 4.8 ms at 1 M, the size of
// the Poisson surfaces the reference renders 200 + N_main times per later iteration.  With one bin per 16x16 raster tile, a
// face whose bounds touch at most BIN_MAXCOVER bins is listed in those bins, the larger ones in one shared list, and a tile
// scans its own list plus the shared one.  (Bins of 64x64 pixels were tried first: 80 counters at 640x480 took ~13 k atomic
// increments each for a 1 M-face mesh and the contention ate the gain -- 4.8 -> 2.7 ms; per-tile bins spread them over 1200.)  Lists are filled through atomics, so their order is arbitrary -- harmless, because
// visibility ties are resolved by face id, not by arrival order.
constexpr int BIN = 16, BIN_MAXCOVER = 4, BIN_MIN_FACES = 16384;  // BIN == RT: one bin per raster tile
struct BinState {
    int *count, *off, *fill, *list, *large, *large_count;  // count == nullptr: binning off
    int bx, by;
};

__device__ __forceinline__ bool bin_range(int p0, int p1, const BinState &b, int &bx0, int &by0, int &bx1, int &by1)
{
    const int x0 = (short)(p0 & 0xffff), y0 = p0 >> 16, x1 = (short)(p1 & 0xffff), y1 = p1 >> 16;
    if (x0 > x1 || y0 > y1) return false;  // culled or empty
    bx0 = max(0, x0 / BIN);
    by0 = max(0, y0 / BIN);
    bx1 = min(b.bx - 1, x1 / BIN);
    by1 = min(b.by - 1, y1 / BIN);
    return bx0 <= bx1 && by0 <= by1;
}

__device__ __forceinline__ float xform(const float *__restrict__ m, float x, float y, float z)
{
    return __builtin_fmaf(m[0], x, __builtin_fmaf(m[1], y, __builtin_fmaf(m[2], z, m[3])));
}

__device__ __forceinline__ float dop(float a, float b, float c, float d)
{
    const float t = c * d;
    return __builtin_fmaf(a, b, -t);
}

struct CamArg {
    float m[16];
};

__global__ __launch_bounds__(256) void tri_setup(const float *__restrict__ soup, int nfaces, CamArg cam, int W, int H,
                                                 TriRec *__restrict__ out, BinState bins)
{
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= nfaces) return;
    const float *v = soup + 9 * (size_t)f;
    float x[3], y[3], z[3], w[3];
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const float px = v[3 * i], py = v[3 * i + 1], pz = v[3 * i + 2];
        x[i] = xform(cam.m + 0, px, py, pz);
        y[i] = xform(cam.m + 4, px, py, pz);
        z[i] = xform(cam.m + 8, px, py, pz);
        w[i] = xform(cam.m + 12, px, py, pz);
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
    float det = __builtin_fmaf(x[0], a[0], __builtin_fmaf(y[0], b[0], w[0] * c[0]));
    TriRec t;
    const bool valid = (det != 0.0f) && (det == det) && !__builtin_isinf(det);
    if (det < 0.0f) {
        det = -det;
#pragma unroll
        for (int i = 0; i < 3; i++) {
            a[i] = -a[i];
            b[i] = -b[i];
            c[i] = -c[i];
        }
    }
    const float rdet = 1.0f / det;
#pragma unroll
    for (int i = 0; i < 3; i++) {
        t.a[i] = a[i];
        t.b[i] = b[i];
        t.c[i] = c[i];
    }
    t.za = __builtin_fmaf(a[0], z[0], __builtin_fmaf(a[1], z[1], a[2] * z[2])) * rdet;
    t.zb = __builtin_fmaf(b[0], z[0], __builtin_fmaf(b[1], z[1], b[2] * z[2])) * rdet;
    t.zc = __builtin_fmaf(c[0], z[0], __builtin_fmaf(c[1], z[1], c[2] * z[2])) * rdet;
    int x0 = 0, y0 = 0, x1 = W - 1, y1 = H - 1;
    bool culled = false;
    float xmin = 1e30f, xmax = -1e30f, ymin = 1e30f, ymax = -1e30f;
    const bool in_front = w[0] > 0.0f && w[1] > 0.0f && w[2] > 0.0f;
    if (in_front) {
#pragma unroll
        for (int i = 0; i < 3; i++) {
            const float nx = x[i] / w[i], ny = y[i] / w[i];
            xmin = fminf(xmin, nx);
            xmax = fmaxf(xmax, nx);
            ymin = fminf(ymin, ny);
            ymax = fmaxf(ymax, ny);
        }
    } else {
        // A vertex at or behind the camera plane: the projected vertices say nothing about the footprint.  tri_fragment only
        // accepts pixels whose point on the face has w > 0 and |z| <= w (all edge functions positive after the sign
        // normalisation above implies w > 0), and pixel centres lie inside (-1, 1)^2, so every fragment comes from the part
        // of the face inside the view frustum.  Clip the triangle against a frustum enlarged by 1 % (Sutherland-Hodgman, six
        // planes, at most nine vertices) and bound what is left; nothing left -> the face cannot produce a fragment.
        // The cameras Heuristic::chooseCameras puts ON the mesh (near = 0.001, heuristic.cpp:193-247) have half the mesh
        // behind them and a band of faces across w = 0; with whole-screen boxes for those, a depth map took ~0.8 ms.
        const float slack = 1.01f;
        float P[10][4], Q[10][4];
        int n = 3;
        for (int i = 0; i < 3; i++) {
            P[i][0] = x[i];
            P[i][1] = y[i];
            P[i][2] = z[i];
            P[i][3] = w[i];
        }
        for (int plane = 0; plane < 6 && n > 0; plane++) {
            const int axis = plane >> 1;
            const float sgn = (plane & 1) ? -1.0f : 1.0f;
            int m = 0;
            for (int i = 0; i < n; i++) {
                const float *cur = P[i], *nxt = P[i + 1 < n ? i + 1 : 0];
                const float fc = slack * cur[3] + sgn * cur[axis], fn = slack * nxt[3] + sgn * nxt[axis];
                if (fc >= 0.0f) {
                    for (int k = 0; k < 4; k++) Q[m][k] = cur[k];
                    m++;
                }
                if ((fc >= 0.0f) != (fn >= 0.0f)) {
                    const float tt = fc / (fc - fn);
                    for (int k = 0; k < 4; k++) Q[m][k] = cur[k] + tt * (nxt[k] - cur[k]);
                    m++;
                }
            }
            n = m;
            for (int i = 0; i < n; i++)
                for (int k = 0; k < 4; k++) P[i][k] = Q[i][k];
        }
        if (n < 3) {
            culled = true;
        } else {
            for (int i = 0; i < n; i++) {
                const float ww = fmaxf(P[i][3], 1e-30f);
                const float nx = P[i][0] / ww, ny = P[i][1] / ww;
                xmin = fminf(xmin, nx);
                xmax = fmaxf(xmax, nx);
                ymin = fminf(ymin, ny);
                ymax = fmaxf(ymax, ny);
            }
        }
    }
    if (!culled) {
        const float cx0 = ((xmin + 1.0f) * (float)W - 1.0f) * 0.5f - 1.0f;
        const float cx1 = ((xmax + 1.0f) * (float)W - 1.0f) * 0.5f + 1.0f;
        const float ry0 = ((1.0f - ymax) * (float)H - 1.0f) * 0.5f - 1.0f;
        const float ry1 = ((1.0f - ymin) * (float)H - 1.0f) * 0.5f + 1.0f;
        if (cx0 > 0.0f) x0 = cx0 < (float)W ? (int)cx0 : W;
        if (cx1 < (float)(W - 1)) x1 = cx1 >= 0.0f ? (int)cx1 : -1;
        if (ry0 > 0.0f) y0 = ry0 < (float)H ? (int)ry0 : H;
        if (ry1 < (float)(H - 1)) y1 = ry1 >= 0.0f ? (int)ry1 : -1;
    }
    if (!valid || culled) {
        x0 = 1;
        x1 = 0;
    }
    t.x0y0 = (x0 & 0xffff) | (y0 << 16);
    t.x1y1 = (x1 & 0xffff) | (y1 << 16);
    t.pad0 = t.pad1 = 0;
    out[f] = t;
    if (bins.count) {  // first binning pass: how many faces per bin, and the shared list of the large ones
        int bx0, by0, bx1, by1;
        if (bin_range(t.x0y0, t.x1y1, bins, bx0, by0, bx1, by1)) {
            if ((bx1 - bx0 + 1) * (by1 - by0 + 1) <= BIN_MAXCOVER) {
                for (int y = by0; y <= by1; y++)
                    for (int x = bx0; x <= bx1; x++) atomicAdd(&bins.count[y * bins.bx + x], 1);
            } else {
                bins.large[atomicAdd(bins.large_count, 1)] = f;
            }
        }
    }
}

// second pass: exclusive offsets of the bins (one workgroup; at most a few thousand bins)
__global__ __launch_bounds__(256) void bin_scan(BinState b)
{
    __shared__ int part[256];
    const int nbins = b.bx * b.by, per = (nbins + 255) / 256, first = threadIdx.x * per, last = min(nbins, first + per);
    int s = 0;
    for (int i = first; i < last; i++) s += b.count[i];
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        int run = 0;
        for (int i = 0; i < 256; i++) {
            const int v = part[i];
            part[i] = run;
            run += v;
        }
    }
    __syncthreads();
    int run = part[threadIdx.x];
    for (int i = first; i < last; i++) {
        b.off[i] = run;
        run += b.count[i];
        b.fill[i] = 0;
    }
}

// third pass: the per-bin lists
__global__ __launch_bounds__(256) void bin_fill(const TriRec *__restrict__ tris, int nfaces, BinState b)
{
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= nfaces) return;
    int bx0, by0, bx1, by1;
    if (!bin_range(tris[f].x0y0, tris[f].x1y1, b, bx0, by0, bx1, by1)) return;
    if ((bx1 - bx0 + 1) * (by1 - by0 + 1) > BIN_MAXCOVER) return;
    for (int y = by0; y <= by1; y++)
        for (int x = bx0; x <= bx1; x++) {
            const int bin = y * b.bx + x;
            b.list[b.off[bin] + atomicAdd(&b.fill[bin], 1)] = f;
        }
}

__device__ __forceinline__ bool edge_inside(float e, float a, float b)
{
    return e > 0.0f || (e == 0.0f && (a > 0.0f || (a == 0.0f && b > 0.0f)));
}

__device__ __forceinline__ bool tri_fragment(const TriRec &t, float xn, float yn, float &zwin, float e[3])
{
    bool in = true;
#pragma unroll
    for (int i = 0; i < 3; i++) {
        e[i] = __builtin_fmaf(t.a[i], xn, __builtin_fmaf(t.b[i], yn, t.c[i]));
        in = in && edge_inside(e[i], t.a[i], t.b[i]);
    }
    const float zn = __builtin_fmaf(t.za, xn, __builtin_fmaf(t.zb, yn, t.zc));
    in = in && (zn >= -1.0f && zn <= 1.0f);
    zwin = __builtin_fmaf(0.5f, zn, 0.5f);
    return in;
}

constexpr int RT = 16;  // raster tile edge

// MODE 0: write window z top-down.  MODE 1: write window z in GL orientation (row 0 = bottom).
// MODE 2: write NDC depth 2z-1 top-down (Render::depth).  ids (nullable): visible face per pixel, top-down.
template <int MODE>
__global__ __launch_bounds__(256) void raster_tiles(const TriRec *__restrict__ tris, int nfaces, int W, int H,
                                                    float invW, float invH, float *__restrict__ zout,
                                                    int *__restrict__ ids, BinState bins)
{
    __shared__ int list[256];
    __shared__ int count;
    const int tx0 = blockIdx.x * RT, ty0 = blockIdx.y * RT;
    const int col = tx0 + (threadIdx.x & (RT - 1));
    const int row = ty0 + (threadIdx.x >> 4);
    const float xn = __builtin_fmaf((float)(2 * col + 1), invW, -1.0f);
    const float yn = __builtin_fmaf(-(float)(2 * row + 1), invH, 1.0f);
    const int tx1 = min(tx0 + RT, W) - 1, ty1 = min(ty0 + RT, H) - 1;
    // NDC coordinates of the tile's corner pixel centres, computed exactly as xn / yn above
    const float cxn0 = __builtin_fmaf((float)(2 * tx0 + 1), invW, -1.0f), cxn1 = __builtin_fmaf((float)(2 * tx1 + 1), invW, -1.0f);
    const float cyn0 = __builtin_fmaf(-(float)(2 * ty0 + 1), invH, 1.0f), cyn1 = __builtin_fmaf(-(float)(2 * ty1 + 1), invH, 1.0f);
    float best = 1.0f;  // glClear(GL_DEPTH_BUFFER_BIT)
    int best_id = -1;
    // candidates: every face, or (binned) this tile's bin list followed by the shared list of large faces
    int cand_a = nfaces, cand_total = nfaces, list_off = 0;
    if (bins.count) {
        const int bin = (ty0 / BIN) * bins.bx + (tx0 / BIN);
        list_off = bins.off[bin];
        cand_a = bins.count[bin];
        cand_total = cand_a + *bins.large_count;
    }
    for (int base = 0; base < cand_total; base += 256) {
        if (threadIdx.x == 0) count = 0;
        __syncthreads();
        const int ci = base + threadIdx.x;
        int f = ci;
        if (bins.count && ci < cand_total) f = ci < cand_a ? bins.list[list_off + ci] : bins.large[ci - cand_a];
        if (ci < cand_total) {
            const int p0 = tris[f].x0y0, p1 = tris[f].x1y1;
            const int bx0 = (short)(p0 & 0xffff), by0 = p0 >> 16, bx1 = (short)(p1 & 0xffff), by1 = p1 >> 16;
            if (bx0 <= tx1 && bx1 >= tx0 && by0 <= ty1 && by1 >= ty0) {
                // tile-level reject: each edge function -- as tri_fragment computes it, fma(a, xn, fma(b, yn, c)) -- is
                // monotone in xn for fixed yn and in yn for fixed xn (rounding is monotone), so its maximum over the
                // tile's pixel centres is attained at a corner pixel; a negative maximum means no pixel can pass that
                // edge.  The same argument bounds zn.  Long slivers seen from a camera on the mesh (chooseCameras) have
                // whole-screen boxes but touch few tiles: a depth map from such a camera took 539 us; chooseCameras' 200 of them now take
                // 17.6 ms in total, host policy code included.
                const TriRec t = tris[f];
                bool keep = true;
#pragma unroll
                for (int i = 0; i < 3 && keep; i++) {
                    const float e00 = __builtin_fmaf(t.a[i], cxn0, __builtin_fmaf(t.b[i], cyn0, t.c[i]));
                    const float e10 = __builtin_fmaf(t.a[i], cxn1, __builtin_fmaf(t.b[i], cyn0, t.c[i]));
                    const float e01 = __builtin_fmaf(t.a[i], cxn0, __builtin_fmaf(t.b[i], cyn1, t.c[i]));
                    const float e11 = __builtin_fmaf(t.a[i], cxn1, __builtin_fmaf(t.b[i], cyn1, t.c[i]));
                    keep = !(fmaxf(fmaxf(e00, e10), fmaxf(e01, e11)) < 0.0f);  // NaN keeps the face
                }
                if (keep) {
                    const float z00 = __builtin_fmaf(t.za, cxn0, __builtin_fmaf(t.zb, cyn0, t.zc));
                    const float z10 = __builtin_fmaf(t.za, cxn1, __builtin_fmaf(t.zb, cyn0, t.zc));
                    const float z01 = __builtin_fmaf(t.za, cxn0, __builtin_fmaf(t.zb, cyn1, t.zc));
                    const float z11 = __builtin_fmaf(t.za, cxn1, __builtin_fmaf(t.zb, cyn1, t.zc));
                    keep = !(fmaxf(fmaxf(z00, z10), fmaxf(z01, z11)) < -1.0f) && !(fminf(fminf(z00, z10), fminf(z01, z11)) > 1.0f);
                }
                if (keep) list[atomicAdd(&count, 1)] = f;
            }
        }
        __syncthreads();
        const int n = count;
        // LDS compaction is unordered; GL_LESS keeps the EARLIER face on equal z, so resolve ties by id
        for (int k = 0; k < n; k++) {
            const int fi = __builtin_amdgcn_readfirstlane(list[k]);
            const TriRec t = tris[fi];
            float zw, e[3];
            if (tri_fragment(t, xn, yn, zw, e)) {
                if (zw < best || (zw == best && best_id >= 0 && fi < best_id)) {
                    best = zw;
                    best_id = fi;
                }
            }
        }
        __syncthreads();
    }
    if (col < W && row < H) {
        if (MODE == 1)
            zout[(size_t)(H - 1 - row) * W + col] = best;
        else if (MODE == 2)
            zout[(size_t)row * W + col] = __builtin_fmaf(2.0f, best, -1.0f);
        else
            zout[(size_t)row * W + col] = best;
        if (ids) ids[(size_t)row * W + col] = best_id;
    }
}

// row 0 (GL orientation) of the shadow filter is a running MIN: HF[0][j] = min(a[0][0..j+1])
// (render_glx.cpp:292-297 reads the already-filtered left neighbour).  One workgroup, blocked scan.
__global__ __launch_bounds__(256) void row0_prefix_min(const float *__restrict__ a, int W, float *__restrict__ hf0)
{
    __shared__ float part[2][256];
    const int per = (W + 255) / 256;
    const int s = threadIdx.x * per, e = min(s + per, W);
    float m = 3.0e38f;
    for (int j = s; j < e; j++) m = fminf(m, a[j]);
    // inclusive prefix minima of the 256 block minima by doubling (min is exact and associative: any order gives the same values; the first
    // form had thread t walk t partials one after the other -- 14 us of one workgroup on the critical path of every side view)
    int cur = 0;
    part[0][threadIdx.x] = m;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {
        const float v = threadIdx.x >= (unsigned)d ? fminf(part[cur][threadIdx.x], part[cur][threadIdx.x - d]) : part[cur][threadIdx.x];
        part[cur ^ 1][threadIdx.x] = v;
        cur ^= 1;
        __syncthreads();
    }
    float run = threadIdx.x > 0 ? part[cur][threadIdx.x - 1] : 3.0e38f;  // exclusive prefix of block minima
    for (int j = s; j < e; j++) {
        // hf0[j] = min(a[0..j+1]) for 1 <= j <= W-2
        run = fminf(run, a[j]);
        if (j >= 1 && j <= W - 2) hf0[j] = fminf(run, a[j + 1]);
    }
}

// closed form of the in-place loop render_glx.cpp:298-312 (validated against its literal transcription
// in oracle/raster_oracle.c by tests/test_raster_cpu.py)
__global__ __launch_bounds__(256) void shadow_dilate(const float *__restrict__ a, const float *__restrict__ hf0, int W,
                                                     int H, float *__restrict__ out)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y;
    if (j >= W) return;
    const size_t p = (size_t)i * W + j;
    if (j == 0 || j == W - 1 || W < 3) {
        out[p] = a[p];
        return;
    }
    float m = -3.0e38f;
#pragma unroll
    for (int dr = -1; dr <= 1; dr++) {
        const int r = i + dr;
        if (r < 0 || r >= H) continue;
        float hf;
        if (r == 0) {
            hf = hf0[j];
        } else {
            const float *q = a + (size_t)r * W + j;
            hf = fmaxf(fmaxf(q[-1], q[0]), q[1]);
        }
        m = fmaxf(m, hf);
    }
    out[p] = m;
}

// ---- mip chain of the frame texture (render_glx.cpp:83-85: GL_LINEAR_MIPMAP_LINEAR + glGenerateMipmap) ---------------------------
// The contract is stated in oracle/raster_oracle.c (mip_build / mip_bilinear / orc_projected_filter) and DESIGN.md section 5: u8 levels
// by 2 x 2 box with (sum + 2) >> 2, fine derivatives on the pixel's 2 x 2 quad with its own face, rho from the larger of the two
// footprint axes, level = exponent of rho, blend fraction = rho 2^-level - 1, GL_REPEAT per level.
constexpr int MAX_MIPS = 14;
struct MipArgs {
    int levels;                                       // levels above 0 (0: level 0 only)
    int w[MAX_MIPS + 1], h[MAX_MIPS + 1], pitch[MAX_MIPS + 1];
    unsigned off[MAX_MIPS + 1];                       // byte offset of the wrap-padded level image in `mips` (level 0: the padded frame itself)
};

// wrap-padded level l from wrap-padded level l - 1 (one thread per padded texel)
__global__ __launch_bounds__(256) void mip_reduce(const uint8_t *__restrict__ src, int pw, int ph, int sp, uint8_t *__restrict__ dst, int w, int h, int dp,
                                                  size_t src_z = 0, size_t dst_z = 0)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x, r = blockIdx.y;
    if (c >= w + 2 || r >= h + 2) return;
    src += src_z * blockIdx.z;  // blockIdx.z = frame of a batch (mvs_process_frame prepares the textures of all its side views at once)
    dst += dst_z * blockIdx.z;
    int j = r - 1, i = c - 1;
    j = j < 0 ? h - 1 : (j >= h ? 0 : j);
    i = i < 0 ? w - 1 : (i >= w ? 0 : i);
    const int x0 = min(2 * i, pw - 1), x1 = min(2 * i + 1, pw - 1), y0 = min(2 * j, ph - 1), y1 = min(2 * j + 1, ph - 1);
    const int sum = src[(size_t)(y0 + 1) * sp + x0 + 1] + src[(size_t)(y0 + 1) * sp + x1 + 1] + src[(size_t)(y1 + 1) * sp + x0 + 1] + src[(size_t)(y1 + 1) * sp + x1 + 1];
    dst[(size_t)r * dp + c] = (uint8_t)((sum + 2) >> 2);
}

// the small levels (<= 64 x 64 texels and everything above them) in one workgroup: one launch instead of one per level
__global__ __launch_bounds__(256) void mip_tail(uint8_t *__restrict__ mips, const uint8_t *__restrict__ level0, MipArgs m, int first, size_t mips_z = 0, size_t level0_z = 0)
{
    mips += mips_z * blockIdx.x;  // one workgroup per frame of a batch
    level0 += level0_z * blockIdx.x;
    for (int l = first; l <= m.levels; l++) {
        const uint8_t *src = l - 1 == 0 ? level0 : mips + m.off[l - 1];
        uint8_t *dst = mips + m.off[l];
        const int pw = m.w[l - 1], ph = m.h[l - 1], sp = m.pitch[l - 1], w = m.w[l], h = m.h[l], dp = m.pitch[l];
        for (int t = threadIdx.x; t < (w + 2) * (h + 2); t += blockDim.x) {
            const int r = t / (w + 2), c = t % (w + 2);
            int j = r - 1, i = c - 1;
            j = j < 0 ? h - 1 : (j >= h ? 0 : j);
            i = i < 0 ? w - 1 : (i >= w ? 0 : i);
            const int x0 = min(2 * i, pw - 1), x1 = min(2 * i + 1, pw - 1), y0 = min(2 * j, ph - 1), y1 = min(2 * j + 1, ph - 1);
            const int sum = src[(size_t)(y0 + 1) * sp + x0 + 1] + src[(size_t)(y0 + 1) * sp + x1 + 1] + src[(size_t)(y1 + 1) * sp + x0 + 1] + src[(size_t)(y1 + 1) * sp + x1 + 1];
            dst[(size_t)r * dp + c] = (uint8_t)((sum + 2) >> 2);
        }
        __threadfence_block();
        __syncthreads();
    }
}

__device__ __forceinline__ float mip_bilinear(const uint8_t *__restrict__ img, int w, int h, int pitch, float u, float vv)
{
    const float cx = __builtin_fmaf(u, (float)w, 0.5f);
    const float cy = __builtin_fmaf(1.0f - vv, (float)h, 0.5f);
    const int ix = (int)cx, iy = (int)cy;
    const float ax = cx - (float)ix, ay = cy - (float)iy;
    const uint8_t *q = img + (size_t)iy * pitch + ix;
    const float t00 = (float)q[0], t01 = (float)q[1], t10 = (float)q[pitch], t11 = (float)q[pitch + 1];
    const float dxt = t01 - t00, dy = t10 - t00, dxy = (t11 - t10) - dxt;
    return __builtin_fmaf(ay, __builtin_fmaf(ax, dxy, dy), __builtin_fmaf(ax, dxt, t00));
}

// texture coordinates of a face at an NDC pixel centre, the face extrapolated past its edges (a GPU's helper invocations)
__device__ __forceinline__ void face_uv(const TriRec &t, const float *__restrict__ v, const CamArg &prj, float xn, float yn, float &u, float &vv)
{
    float e[3];
#pragma unroll
    for (int i = 0; i < 3; i++) e[i] = __builtin_fmaf(t.a[i], xn, __builtin_fmaf(t.b[i], yn, t.c[i]));
    const float esum = (e[0] + e[1]) + e[2];
    float pos[3];
#pragma unroll
    for (int k = 0; k < 3; k++) pos[k] = __builtin_fmaf(e[0], v[k], __builtin_fmaf(e[1], v[3 + k], e[2] * v[6 + k])) / esum;
    const float sx = xform(prj.m + 0, pos[0], pos[1], pos[2]);
    const float sy = xform(prj.m + 4, pos[0], pos[1], pos[2]);
    const float sw = xform(prj.m + 12, pos[0], pos[1], pos[2]);
    u = __builtin_fmaf(0.5f, sx / sw, 0.5f);
    vv = __builtin_fmaf(0.5f, sy / sw, 0.5f);
}

// shader.vert:9-13 + shader.frag:11-25 for the visible face of every main-view pixel
__global__ __launch_bounds__(256) void project_texture(const float *__restrict__ soup, const TriRec *__restrict__ tris,
                                                       const int *__restrict__ ids, const float *__restrict__ shadow_gl,
                                                       const uint8_t *__restrict__ pad, int pitch, CamArg prj, int W,
                                                       int H, float invW, float invH, uint8_t *__restrict__ out3,
                                                       const uint8_t *__restrict__ mips, MipArgs mip, const uint8_t *__restrict__ mix_bg = nullptr,
                                                       float *__restrict__ mix_depth = nullptr, uint8_t *__restrict__ mix_out = nullptr)
{
    const int col = blockIdx.x * 64 + (threadIdx.x & 63);
    const int row = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (col >= W || row >= H) return;
    const size_t p = (size_t)row * W + col;
    uint8_t r = 0, g = 0;
    const int id = ids[p];
    if (id >= 0) {
        const float xn = __builtin_fmaf((float)(2 * col + 1), invW, -1.0f);
        const float yn = __builtin_fmaf(-(float)(2 * row + 1), invH, 1.0f);
        const TriRec t = tris[id];
        const float *v = soup + 9 * (size_t)id;
        float zw, e[3];
        if (tri_fragment(t, xn, yn, zw, e)) {
            const float esum = (e[0] + e[1]) + e[2];
            float pos[3];
#pragma unroll
            for (int k = 0; k < 3; k++)
                pos[k] = __builtin_fmaf(e[0], v[k], __builtin_fmaf(e[1], v[3 + k], e[2] * v[6 + k])) / esum;
            const float sx = xform(prj.m + 0, pos[0], pos[1], pos[2]);
            const float sy = xform(prj.m + 4, pos[0], pos[1], pos[2]);
            const float sz = xform(prj.m + 8, pos[0], pos[1], pos[2]);
            const float sw = xform(prj.m + 12, pos[0], pos[1], pos[2]);
            const float nx = sx / sw, ny = sy / sw, nz = sz / sw;
            const bool inframe = nx > -1.0f && nx < 1.0f && ny > -1.0f && ny < 1.0f;
            if (inframe) {
                const float fW = (float)W, fH = (float)H;
                const float u = __builtin_fmaf(0.5f, nx, 0.5f), vv = __builtin_fmaf(0.5f, ny, 0.5f);
                int si = (int)floorf(u * fW), sj = (int)floorf(vv * fH);
                si = ((si % W) + W) % W;
                sj = ((sj % H) + H) % H;
                const float shadowDepth = __builtin_fmaf(2.0f, shadow_gl[(size_t)sj * W + si], -1.0f);
                if (shadowDepth + 0.01f > nz) {
                    float res = mip_bilinear(pad, W, H, pitch, u, vv);
                    if (mip.levels > 0) {
                        // the pixel's footprint in level-0 texels: finite differences on its 2 x 2 quad, same face
                        float ux, vx, uy, vy;
                        face_uv(t, v, prj, __builtin_fmaf((float)(2 * (col ^ 1) + 1), invW, -1.0f), yn, ux, vx);
                        face_uv(t, v, prj, xn, __builtin_fmaf(-(float)(2 * (row ^ 1) + 1), invH, 1.0f), uy, vy);
                        const float dudx = (ux - u) * fW, dvdx = (vx - vv) * fH, dudy = (uy - u) * fW, dvdy = (vy - vv) * fH;
                        const float rx = dudx * dudx + dvdx * dvdx, ry = dudy * dudy + dvdy * dvdy;
                        const float rho = sqrtf(rx > ry ? rx : ry);
                        if (rho > 1.0f && rho < 3.0e38f) {
                            const uint32_t bits = __builtin_bit_cast(uint32_t, rho);
                            const int l0 = (int)((bits >> 23) & 0xffu) - 127;
                            const float f = __builtin_bit_cast(float, (bits & 0x007fffffu) | 0x3f800000u) - 1.0f;  // rho 2^-l0 - 1, exact
                            if (l0 >= mip.levels) {
                                res = mip_bilinear(mips + mip.off[mip.levels], mip.w[mip.levels], mip.h[mip.levels], mip.pitch[mip.levels], u, vv);
                            } else {
                                const float s0 = l0 == 0 ? res : mip_bilinear(mips + mip.off[l0], mip.w[l0], mip.h[l0], mip.pitch[l0], u, vv);
                                const float s1 = mip_bilinear(mips + mip.off[l0 + 1], mip.w[l0 + 1], mip.h[l0 + 1], mip.pitch[l0 + 1], u, vv);
                                res = __builtin_fmaf(f, s1 - s0, s0);
                            }
                        }
                    }
                    r = (uint8_t)(int)(res + 0.5f);
                    g = 255;
                }
            }
        }
    }
    if (mix_out) {
        // mixBackground (util.cpp:366-387) on the fragment just shaded: mvs_process_frame needs the mixed image and the masked depth, never the RGB8
        // frame itself -- the same selects on the same values as the mix_background kernel, one launch and 3 P bytes of traffic fewer per side view
        const bool masked = mix_depth[p] == MVS_BACKGROUND_DEPTH || g == 0;
        mix_out[p] = masked ? mix_bg[p] : r;
        if (masked) mix_depth[p] = MVS_BACKGROUND_DEPTH;
        return;
    }
    out3[3 * p + 0] = r;
    out3[3 * p + 1] = g;
    out3[3 * p + 2] = g;
}

// util.cpp:366-387 (depth is updated in place)
__global__ __launch_bounds__(256) void mix_background(const uint8_t *__restrict__ img3, const uint8_t *__restrict__ bg,
                                                      float *__restrict__ depth, uint8_t *__restrict__ out, size_t P)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    const bool masked = depth[i] == MVS_BACKGROUND_DEPTH || img3[3 * i + 1] == 0;
    out[i] = masked ? bg[i] : img3[3 * i];
    if (masked) depth[i] = MVS_BACKGROUND_DEPTH;
}

// pad_wrap_kernel (context.hip) for N frames per launch: the frames wherever they are (a pointer each: side by side in mvs_process_frame's buffer, or
// slots of the frame store), padded copies `pad_z` bytes apart (blockIdx.z = frame)
struct FramePtrs {
    const uint8_t *p[32];
};

__global__ __launch_bounds__(256) void pad_wrap_frames(FramePtrs frames, uint8_t *__restrict__ pad, int W, int H, int pitch, size_t pad_z)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    const int r = blockIdx.y;
    if (c >= pitch) return;
    const uint8_t *__restrict__ img = frames.p[blockIdx.z];
    pad += pad_z * blockIdx.z;
    uint8_t v = 0;
    if (c < W + 2) {
        int sr = r - 1;
        sr = sr < 0 ? H - 1 : (sr >= H ? 0 : sr);
        int sc = c - 1;
        sc = sc < 0 ? W - 1 : (sc >= W ? 0 : sc);
        v = img[(size_t)sr * W + sc];
    }
    pad[(size_t)r * pitch + c] = v;
}

// (tris_buf: where this camera's triangle records go -- the context's scratch by default; mvs_process_frame keeps the main camera's in a buffer of
// their own for the whole frame)
static int run_raster(mvs_ctx *ctx, const float cam[16], int mode, float *zout, int *ids, DevBuf *tris_buf = nullptr)
{
    CamArg c;
    memcpy(c.m, cam, sizeof(c.m));
    const int W = ctx->W, H = ctx->H;
    DevBuf &tb = tris_buf ? *tris_buf : ctx->r_tmp2;
    int rc = ensure(ctx, tb, sizeof(TriRec) * (size_t)(ctx->nfaces > 0 ? ctx->nfaces : 1));
    if (rc) return rc;
    BinState bins = {};
    const int force_bins = ctx->hooks.raster_bins;  // test hook: 1 always, 0 never
    const int nbx = div_up(W, BIN), nby = div_up(H, BIN);
    if (ctx->nfaces > 0 && nbx * nby <= 65536 && (force_bins >= 0 ? force_bins == 1 : ctx->nfaces >= BIN_MIN_FACES)) {
        const size_t nbins = (size_t)nbx * nby, F = (size_t)ctx->nfaces;
        if ((rc = ensure(ctx, ctx->raster_bins, sizeof(int) * (3 * nbins + 1 + BIN_MAXCOVER * F + F)))) return rc;
        bins.count = (int *)ctx->raster_bins.ptr;
        bins.large_count = bins.count + nbins;  // adjacent to the counts: one memset clears both
        bins.off = bins.large_count + 1;
        bins.fill = bins.off + nbins;
        bins.list = bins.fill + nbins;
        bins.large = bins.list + BIN_MAXCOVER * F;
        bins.bx = nbx;
        bins.by = nby;
        MVS_HIP(ctx, hipMemsetAsync(bins.count, 0, sizeof(int) * (nbins + 1), ctx->stream));
    }
    if (ctx->nfaces > 0) {
        tri_setup<<<div_up(ctx->nfaces, 256), 256, 0, ctx->stream>>>((const float *)ctx->soup.ptr, ctx->nfaces, c, W, H,
                                                                      (TriRec *)tb.ptr, bins);
        if (bins.count) {
            bin_scan<<<1, 256, 0, ctx->stream>>>(bins);
            bin_fill<<<div_up(ctx->nfaces, 256), 256, 0, ctx->stream>>>((const TriRec *)tb.ptr, ctx->nfaces, bins);
        }
        MVS_HIP(ctx, hipGetLastError());
    }
    dim3 grid(div_up(W, RT), div_up(H, RT));
    const float invW = 1.0f / (float)W, invH = 1.0f / (float)H;
    const TriRec *tris = (const TriRec *)tb.ptr;
    ProfileScope ps(ctx, MVS_K_RASTER);
    if (mode == 0)
        raster_tiles<0><<<grid, 256, 0, ctx->stream>>>(tris, ctx->nfaces, W, H, invW, invH, zout, ids, bins);
    else if (mode == 1)
        raster_tiles<1><<<grid, 256, 0, ctx->stream>>>(tris, ctx->nfaces, W, H, invW, invH, zout, ids, bins);
    else
        raster_tiles<2><<<grid, 256, 0, ctx->stream>>>(tris, ctx->nfaces, W, H, invW, invH, zout, ids, bins);
    MVS_HIP(ctx, hipGetLastError());
    return MVS_OK;
}

// ---- device-buffer forms (also used by pipeline.hip) -----------------------------------------------------------------

__global__ __launch_bounds__(256) void gather_pixels(const float *__restrict__ map, int W, int n, const int32_t *__restrict__ rows,
                                                     const int32_t *__restrict__ cols, float *__restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = map[(size_t)rows[i] * W + cols[i]];
}

int depth_device(mvs_ctx *ctx, const float cam[16], float *out_dev)
{
    if (!ctx->soup.ptr) return fail(ctx, MVS_ESTATE, "no mesh loaded (mvs_load_mesh)");
    return run_raster(ctx, cam, 2, out_dev, nullptr);
}

// Render::projected on device buffers: frame_dev = H*W u8 (tight), out3_dev = H*W*3 u8
// Render::projected in two halves.  The MAIN pass (the mesh rasterised from the main camera: window z + face ids, the main camera's triangle
// records) does not depend on the side view: mvs_process_frame runs it once per main frame (projected_main_pass) and the per-view half
// (projected_side_pass: frame padding + mip chain, shadow map from the projector with its dilation, the fragment program) once per side view --
// the same kernels on the same inputs as the one-call form below, 4 x 30 us of identical rasterisation fewer per main frame with 4 side views (round 6).
int projected_main_pass(mvs_ctx *ctx, const float cam[16])
{
    if (!ctx->soup.ptr) return fail(ctx, MVS_ESTATE, "no mesh loaded (mvs_load_mesh)");
    const size_t P = (size_t)ctx->W * ctx->H;
    int rc;
    if ((rc = ensure(ctx, ctx->r_zbuf, P * sizeof(float)))) return rc;       // main pass window z
    if ((rc = ensure(ctx, ctx->r_tmp0, P * sizeof(int)))) return rc;         // ids
    return run_raster(ctx, cam, 0, (float *)ctx->r_zbuf.ptr, (int *)ctx->r_tmp0.ptr, &ctx->r_tris_main);
}

// The frame texture of Render::projected for `nframes` side frames at once (frames W*H bytes apart): the wrap-padded copy and its mip chain, the
// same kernels' arithmetic per frame, every launch covering all frames (round 6: five launches per main frame instead of five per side view).
// Frame i's padded copy is at r_frame + i * tex_frame_bytes, its mips at r_mips + i * tex_mips_bytes; `mip` describes one frame's chain.
static int projected_textures(mvs_ctx *ctx, const FramePtrs &frames, int nframes, MipArgs &mip)
{
    const int W = ctx->W, H = ctx->H;
    const int pitch = ((W + 2 + 63) / 64) * 64;
    const size_t frame_bytes = ((size_t)pitch * (H + 2) + 64 + 63) & ~(size_t)63;
    int rc;
    if (nframes > 32) return fail(ctx, MVS_EINVAL, "projected: at most 32 frame textures per pass");
    if ((rc = ensure(ctx, ctx->r_frame, frame_bytes * (size_t)nframes))) return rc;
    pad_wrap_frames<<<dim3(div_up(pitch, 256), H + 2, (unsigned)nframes), 256, 0, ctx->stream>>>(frames, (uint8_t *)ctx->r_frame.ptr, W, H, pitch, frame_bytes);
    MVS_HIP(ctx, hipGetLastError());
    memset(&mip, 0, sizeof(mip));
    mip.w[0] = W;
    mip.h[0] = H;
    mip.pitch[0] = pitch;
    ctx->tex_frame_bytes = frame_bytes;
    ctx->tex_mips_bytes = 0;
    // the frame texture's mip chain (what the reference asks GL for; mvs_set_texture_filter(MVS_FILTER_LEVEL0) switches it off)
    if (ctx->texture_filter == MVS_FILTER_MIPMAP) {
        size_t off = 0;
        int l = 0;
        while ((mip.w[l] > 1 || mip.h[l] > 1) && l < MAX_MIPS) {
            l++;
            mip.w[l] = mip.w[l - 1] > 1 ? mip.w[l - 1] >> 1 : 1;
            mip.h[l] = mip.h[l - 1] > 1 ? mip.h[l - 1] >> 1 : 1;
            mip.pitch[l] = mip.w[l] + 2;
            mip.off[l] = (unsigned)off;
            off += ((size_t)mip.pitch[l] * (mip.h[l] + 2) + 63) & ~(size_t)63;
        }
        mip.levels = l;
        const size_t mips_bytes = (off + 64 + 63) & ~(size_t)63;
        ctx->tex_mips_bytes = mips_bytes;
        if ((rc = ensure(ctx, ctx->r_mips, mips_bytes * (size_t)nframes))) return rc;
        uint8_t *mips = (uint8_t *)ctx->r_mips.ptr;
        int first_tail = mip.levels + 1;
        for (int k = 1; k <= mip.levels; k++) {
            if (mip.w[k] <= 64 && mip.h[k] <= 64) {
                first_tail = k;
                break;
            }
            const uint8_t *src = k == 1 ? (const uint8_t *)ctx->r_frame.ptr : mips + mip.off[k - 1];
            mip_reduce<<<dim3(div_up(mip.w[k] + 2, 256), mip.h[k] + 2, (unsigned)nframes), 256, 0, ctx->stream>>>(src, mip.w[k - 1], mip.h[k - 1], mip.pitch[k - 1], mips + mip.off[k],
                                                                                                          mip.w[k], mip.h[k], mip.pitch[k], k == 1 ? frame_bytes : mips_bytes, mips_bytes);
        }
        if (first_tail <= mip.levels) mip_tail<<<(unsigned)nframes, 256, 0, ctx->stream>>>(mips, (const uint8_t *)ctx->r_frame.ptr, mip, first_tail, mips_bytes, frame_bytes);
        MVS_HIP(ctx, hipGetLastError());
    }
    static_assert(sizeof(MipArgs) <= sizeof(ctx->tex_mip), "mvs_ctx::tex_mip holds a MipArgs");
    memcpy(ctx->tex_mip, &mip, sizeof(mip));
    return MVS_OK;
}

// mvs_process_frame: the textures of all side frames up front (they depend on the frames alone)
int projected_prepare_views(mvs_ctx *ctx, const uint8_t *const *frames_dev, int nframes)
{
    MipArgs mip;
    FramePtrs fp;
    memset(&fp, 0, sizeof(fp));
    for (int i = 0; i < nframes && i < 32; i++) fp.p[i] = frames_dev[i];
    const int rc = projected_textures(ctx, fp, nframes, mip);
    ctx->tex_prepared = rc == MVS_OK ? nframes : 0;
    return rc;
}

// prepared_view >= 0: the texture of that frame was made by projected_prepare_views (frame_dev is not read); -1: made here.
// mix_out != nullptr: mixBackground(projected, mix_bg, mix_depth) is what comes out (into mix_out; mix_depth updated), out3_dev is not written.
int projected_side_pass(mvs_ctx *ctx, const uint8_t *frame_dev, const float projector[16], uint8_t *out3_dev, int prepared_view, const uint8_t *mix_bg, float *mix_depth,
                        uint8_t *mix_out)
{
    if (!ctx->soup.ptr) return fail(ctx, MVS_ESTATE, "no mesh loaded (mvs_load_mesh)");
    const int W = ctx->W, H = ctx->H;
    const size_t P = (size_t)W * H;
    const int pitch = ((W + 2 + 63) / 64) * 64;
    int rc;
    if ((rc = ensure(ctx, ctx->r_shadow, 2 * P * sizeof(float) + sizeof(float) * (size_t)W))) return rc;  // raw, dilated, hf0
    float *sh_raw = (float *)ctx->r_shadow.ptr, *sh_dil = sh_raw + P, *hf0 = sh_dil + P;
    MipArgs mip;
    if (prepared_view >= 0) {
        if (prepared_view >= ctx->tex_prepared) return fail(ctx, MVS_ESTATE, "projected: texture %d was not prepared", prepared_view);
        memcpy(&mip, ctx->tex_mip, sizeof(mip));
    } else {
        FramePtrs fp;
        memset(&fp, 0, sizeof(fp));
        fp.p[0] = frame_dev;
        if ((rc = projected_textures(ctx, fp, 1, mip))) return rc;
        ctx->tex_prepared = 0;
        prepared_view = 0;
    }
    const uint8_t *tex_frame = (const uint8_t *)ctx->r_frame.ptr + ctx->tex_frame_bytes * (size_t)prepared_view;
    const uint8_t *tex_mips = ctx->r_mips.ptr ? (const uint8_t *)ctx->r_mips.ptr + ctx->tex_mips_bytes * (size_t)prepared_view : nullptr;
    // pass 1: shadow map from the projector, GL orientation, then the dilation quirk
    if ((rc = run_raster(ctx, projector, 1, sh_raw, nullptr))) return rc;
    row0_prefix_min<<<1, 256, 0, ctx->stream>>>(sh_raw, W, hf0);
    MVS_HIP(ctx, hipGetLastError());
    shadow_dilate<<<dim3(div_up(W, 256), H), 256, 0, ctx->stream>>>(sh_raw, hf0, W, H, sh_dil);
    MVS_HIP(ctx, hipGetLastError());
    // pass 2 (projected_main_pass: main camera) has run; the fragment program on the visible faces
    CamArg prj;
    memcpy(prj.m, projector, sizeof(prj.m));
    ProfileScope ps(ctx, MVS_K_PROJECT);
    project_texture<<<dim3(div_up(W, 64), div_up(H, 4)), 256, 0, ctx->stream>>>(
        (const float *)ctx->soup.ptr, (const TriRec *)ctx->r_tris_main.ptr, (const int *)ctx->r_tmp0.ptr, sh_dil,
        tex_frame, pitch, prj, W, H, 1.0f / (float)W, 1.0f / (float)H, out3_dev, tex_mips, mip, mix_bg, mix_depth, mix_out);
    MVS_HIP(ctx, hipGetLastError());
    return MVS_OK;
}

int projected_device(mvs_ctx *ctx, const float cam[16], const uint8_t *frame_dev, const float projector[16], uint8_t *out3_dev)
{
    int rc = projected_main_pass(ctx, cam);
    return rc ? rc : projected_side_pass(ctx, frame_dev, projector, out3_dev, -1, nullptr, nullptr, nullptr);
}

int mix_background_device(mvs_ctx *ctx, const uint8_t *img3_dev, const uint8_t *bg_dev, float *depth_dev, uint8_t *out_dev)
{
    const size_t P = (size_t)ctx->W * ctx->H;
    mix_background<<<(unsigned)((P + 255) / 256), 256, 0, ctx->stream>>>(img3_dev, bg_dev, depth_dev, out_dev, P);
    MVS_HIP(ctx, hipGetLastError());
    return MVS_OK;
}

}  // namespace mvs

using namespace mvs;

extern "C" {

int mvs_load_mesh(mvs_ctx *ctx, const float *verts4, int nverts, const int32_t *faces3, int nfaces)
{
    if (!ctx) return MVS_EINVAL;
    if (nverts < 0 || nfaces < 0 || (nfaces > 0 && (!verts4 || !faces3)))
        return fail(ctx, MVS_EINVAL, "mvs_load_mesh: bad arguments");
    MVS_HIP(ctx, hipSetDevice(ctx->device));
    std::vector<float> soup((size_t)nfaces * 9);
    for (int f = 0; f < nfaces; f++)
        for (int j = 0; j < 3; j++) {
            const int vi = faces3[3 * f + j];
            if (vi < 0 || vi >= nverts)
                return fail(ctx, MVS_EINVAL, "mvs_load_mesh: face %d references vertex %d of %d", f, vi, nverts);
            const float *p = verts4 + 4 * (size_t)vi;
            soup[9 * (size_t)f + 3 * j + 0] = p[0] / p[3];  // render_glx.cpp:242-244
            soup[9 * (size_t)f + 3 * j + 1] = p[1] / p[3];
            soup[9 * (size_t)f + 3 * j + 2] = p[2] / p[3];
        }
    int rc = ensure(ctx, ctx->soup, sizeof(float) * 9 * (size_t)(nfaces > 0 ? nfaces : 1));
    if (rc) return rc;
    if (nfaces > 0) {
        MVS_HIP(ctx, hipMemcpyAsync(ctx->soup.ptr, soup.data(), sizeof(float) * soup.size(), hipMemcpyHostToDevice,
                                    ctx->stream));
        MVS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    ctx->nfaces = nfaces;
    return MVS_OK;
}

int mvs_depth(mvs_ctx *ctx, const float cam[16], float *out_hw)
{
    if (!ctx || !cam || !out_hw) return fail(ctx, MVS_EINVAL, "mvs_depth: null argument");
    if (!ctx->soup.ptr) return fail(ctx, MVS_ESTATE, "mvs_depth: no mesh loaded (mvs_load_mesh)");
    MVS_HIP(ctx, hipSetDevice(ctx->device));
    const size_t P = (size_t)ctx->W * ctx->H;
    int rc = ensure(ctx, ctx->r_zbuf, P * sizeof(float));
    if (rc) return rc;
    if ((rc = run_raster(ctx, cam, 2, (float *)ctx->r_zbuf.ptr, nullptr))) return rc;
    MVS_HIP(ctx, hipMemcpyAsync(out_hw, ctx->r_zbuf.ptr, P * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    MVS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return MVS_OK;
}

int mvs_depth_probe(mvs_ctx *ctx, const float cam[16], int n, const int32_t *rows, const int32_t *cols, float *out)
{
    if (!ctx || !cam || n < 0 || (n > 0 && (!rows || !cols || !out))) return fail(ctx, MVS_EINVAL, "mvs_depth_probe: bad argument");
    if (!ctx->soup.ptr) return fail(ctx, MVS_ESTATE, "mvs_depth_probe: no mesh loaded (mvs_load_mesh)");
    MVS_HIP(ctx, hipSetDevice(ctx->device));
    for (int i = 0; i < n; i++)
        if (rows[i] < 0 || rows[i] >= ctx->H || cols[i] < 0 || cols[i] >= ctx->W)
            return fail(ctx, MVS_EINVAL, "mvs_depth_probe: pixel %d (%d, %d) outside the %d x %d map", i, rows[i], cols[i], ctx->H, ctx->W);
    const size_t P = (size_t)ctx->W * ctx->H;
    int rc = ensure(ctx, ctx->r_zbuf, P * sizeof(float));
    if (rc) return rc;
    if ((rc = ensure(ctx, ctx->probe_buf, (size_t)(n > 0 ? n : 1) * 3 * sizeof(int32_t)))) return rc;
    if ((rc = run_raster(ctx, cam, 2, (float *)ctx->r_zbuf.ptr, nullptr))) return rc;
    if (n == 0) return MVS_OK;
    int32_t *d_rows = (int32_t *)ctx->probe_buf.ptr, *d_cols = d_rows + n;
    float *d_out = (float *)(d_cols + n);
    MVS_HIP(ctx, hipMemcpyAsync(d_rows, rows, sizeof(int32_t) * n, hipMemcpyHostToDevice, ctx->stream));
    MVS_HIP(ctx, hipMemcpyAsync(d_cols, cols, sizeof(int32_t) * n, hipMemcpyHostToDevice, ctx->stream));
    gather_pixels<<<(unsigned)((n + 255) / 256), 256, 0, ctx->stream>>>((const float *)ctx->r_zbuf.ptr, ctx->W, n, d_rows, d_cols, d_out);
    MVS_HIP(ctx, hipGetLastError());
    MVS_HIP(ctx, hipMemcpyAsync(out, d_out, sizeof(float) * n, hipMemcpyDeviceToHost, ctx->stream));
    MVS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return MVS_OK;
}

int mvs_projected(mvs_ctx *ctx, const float cam[16], const uint8_t *frame_hw, const float projector[16],
                  uint8_t *out_hw3)
{
    if (!ctx || !cam || !frame_hw || !projector || !out_hw3) return fail(ctx, MVS_EINVAL, "mvs_projected: null argument");
    if (!ctx->soup.ptr) return fail(ctx, MVS_ESTATE, "mvs_projected: no mesh loaded (mvs_load_mesh)");
    MVS_HIP(ctx, hipSetDevice(ctx->device));
    const size_t P = (size_t)ctx->W * ctx->H;
    int rc;
    if ((rc = ensure(ctx, ctx->upload, P))) return rc;
    if ((rc = ensure(ctx, ctx->r_out3, 3 * P))) return rc;
    MVS_HIP(ctx, hipMemcpyAsync(ctx->upload.ptr, frame_hw, P, hipMemcpyHostToDevice, ctx->stream));
    if ((rc = projected_device(ctx, cam, (const uint8_t *)ctx->upload.ptr, projector, (uint8_t *)ctx->r_out3.ptr))) return rc;
    MVS_HIP(ctx, hipMemcpyAsync(out_hw3, ctx->r_out3.ptr, 3 * P, hipMemcpyDeviceToHost, ctx->stream));
    MVS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return MVS_OK;
}

int mvs_mix_background(mvs_ctx *ctx, const uint8_t *img_hw3, const uint8_t *bg_hw, float *depth_hw_inout,
                       uint8_t *out_hw)
{
    if (!ctx || !img_hw3 || !bg_hw || !depth_hw_inout || !out_hw)
        return fail(ctx, MVS_EINVAL, "mvs_mix_background: null argument");
    MVS_HIP(ctx, hipSetDevice(ctx->device));
    const size_t P = (size_t)ctx->W * ctx->H;
    int rc;
    if ((rc = ensure(ctx, ctx->r_out3, 3 * P))) return rc;
    if ((rc = ensure(ctx, ctx->upload, P))) return rc;
    if ((rc = ensure(ctx, ctx->r_zbuf, P * sizeof(float)))) return rc;
    if ((rc = ensure(ctx, ctx->r_tmp1, P))) return rc;
    MVS_HIP(ctx, hipMemcpyAsync(ctx->r_out3.ptr, img_hw3, 3 * P, hipMemcpyHostToDevice, ctx->stream));
    MVS_HIP(ctx, hipMemcpyAsync(ctx->upload.ptr, bg_hw, P, hipMemcpyHostToDevice, ctx->stream));
    MVS_HIP(ctx, hipMemcpyAsync(ctx->r_zbuf.ptr, depth_hw_inout, P * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    mix_background<<<(unsigned)((P + 255) / 256), 256, 0, ctx->stream>>>(
        (const uint8_t *)ctx->r_out3.ptr, (const uint8_t *)ctx->upload.ptr, (float *)ctx->r_zbuf.ptr,
        (uint8_t *)ctx->r_tmp1.ptr, P);
    MVS_HIP(ctx, hipGetLastError());
    MVS_HIP(ctx, hipMemcpyAsync(out_hw, ctx->r_tmp1.ptr, P, hipMemcpyDeviceToHost, ctx->stream));
    MVS_HIP(ctx, hipMemcpyAsync(depth_hw_inout, ctx->r_zbuf.ptr, P * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    MVS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return MVS_OK;
}

}  // extern "C"
