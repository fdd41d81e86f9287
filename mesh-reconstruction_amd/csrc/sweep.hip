// sweep.hip -- plane-sweep photometric cost volume + per-pixel depth selection for gfx950 (MI355X).
//
// What it replaces: the reference evaluates ONE depth hypothesis per pixel per (main, side) pair by
// rasterising the proxy mesh and running shader.frag:11-25 (projective texture fetch + in-frame test)
// followed by an RGB8 read-back (render_glx.cpp:261-367).  This file evaluates the same fragment
// program at D synthetic positions pos_d = main^-1 * (x_ndc, y_ndc, z_d, 1) per pixel and V side
// views, accumulates |I_main - I_warp| and picks the best plane.
//
// Arithmetic contract (identical in oracle/sweep_oracle.c; DESIGN.md "sweep arithmetic"):
//   s      = fma(z, B, A)            A = fma(Q0, xn, fma(Q1, yn, Q3)), B = Q2  (per row of Q)
//   r      = RN(1 / s.w)             v_rcp_f32 + one FMA Newton step
//   c      = s.xy * r                padded-pixel coordinates, in frame iff 0.5 < c < size + 0.5
//   i, a   = trunc(c), c - i         exact (c > 0)
//   res    = fma(ay, fma(ax, dxy, dy), fma(ax, dxt, t00 + 0.5))   (t00 + 0.5 is exact)
//   Iq     = (int)res                the u8 the RGB8 framebuffer would hold (render_glx.cpp:359)
//   cell  += (1 << 16) + |Iq - Im|   integer: exact, order independent, all-reducible
// Build with -ffp-contract=off: every f32 op above is exactly one rounding.
//
// Kernels:
//   plan_regions      per (tile, plane chunk, view): bounding box of the warped tile -> LDS region
//   sweep_tiled       LDS-staged side-image tiles; 64x8-pixel tiles x 32 planes or 64x16 x 16 planes per chunk
//   sweep_generic     no tiling, global gathers; any geometry; also the in-kernel fallback
//   argmin_volume     depth selection over the packed volume (after an optional cross-rank reduction)
#include "sweep_shared.hpp"

#include <cstdlib>

namespace mvs {

// t00h = t00 + 0.5 (exact): the rounding bias of the u8 conversion rides on the first term
__device__ __forceinline__ int bilerp_u8(float ax, float ay, float t00h, float dxt, float dy, float dxy)
{
    const float a = __builtin_fmaf(ax, dxt, t00h);
    const float b = __builtin_fmaf(ax, dxy, dy);
    const float res = __builtin_fmaf(ay, b, a);
    return (int)res;
}

// one sample with every check, taps gathered from the padded image in global memory
__device__ __forceinline__ uint32_t sample_global(const Affine &A, float bx, float by, float bw, float z,
                                                  const uint8_t *__restrict__ pad, int pitch, float Wp, float Hp,
                                                  int Im)
{
    const float sx = __builtin_fmaf(z, bx, A.ax);
    const float sy = __builtin_fmaf(z, by, A.ay);
    const float sw = __builtin_fmaf(z, bw, A.aw);
    if (!(sw > 0.0f)) return 0u;
    const float r = rcp_rn(sw);
    const float cx = sx * r, cy = sy * r;
    if (!(cx > 0.5f && cx < Wp && cy > 0.5f && cy < Hp)) return 0u;
    const int ix = (int)cx, iy = (int)cy;
    const float fx = cx - (float)ix, fy = cy - (float)iy;
    const uint8_t *p = pad + (size_t)iy * pitch + ix;
    const float t00 = (float)p[0], t01 = (float)p[1], t10 = (float)p[pitch], t11 = (float)p[pitch + 1];
    const float dxt = t01 - t00, dy = t10 - t00, dxy = (t11 - t10) - dxt;
    const int Iq = bilerp_u8(fx, fy, t00 + 0.5f, dxt, dy, dxy);
    return 65536u + (uint32_t)__builtin_abs(Iq - Im);
}

// ------------------------------------------------------------------------------------------------------
// generic kernel: one pixel per thread, global gathers
// ------------------------------------------------------------------------------------------------------
template <bool WRITE_VOLUME, bool FUSED>
__global__ __launch_bounds__(256) void sweep_generic(SweepParams p)
{
    const int col = blockIdx.x * 64 + (threadIdx.x & 63);
    const int row = p.row_begin + blockIdx.y * 4 + (threadIdx.x >> 6);
    if (col >= p.W || row >= p.row_end) return;
    const size_t P = (size_t)p.W * p.H;
    const size_t pix = (size_t)row * p.W + col;
    const float xn = __builtin_fmaf((float)(2 * col + 1), p.invW, -1.0f);
    const float yn = __builtin_fmaf(-(float)(2 * row + 1), p.invH, 1.0f);
    const int Im = p.main_img[pix];
    uint32_t bs = 0, bc = 0;
    int bi = -1;
    for (int d0 = p.plane_begin; d0 < p.plane_end; d0 += PCG) {
        uint32_t acc[PCG];
#pragma unroll
        for (int k = 0; k < PCG; k++) acc[k] = 0u;
        for (int v = p.v0; v < p.v0 + p.vcount; v++) {
            const float *q = p.Q + 12 * v;
            const Affine A = view_affine(q, xn, yn);
            const float bx = q[2], by = q[6], bw = q[10];
            const uint8_t *pad = p.pads + p.pad_slab * v;
#pragma unroll
            for (int k = 0; k < PCG; k++) {
                if (d0 + k < p.plane_end)
                    acc[k] += sample_global(A, bx, by, bw, p.z[d0 + k], pad, p.pitch, p.Wp, p.Hp, Im);
            }
        }
#pragma unroll
        for (int k = 0; k < PCG; k++) {
            if (d0 + k < p.plane_end) {
                if (WRITE_VOLUME) p.volume[(size_t)(d0 + k) * P + pix] = acc[k];
                if (FUSED) argmin_update(acc[k], d0 + k, bs, bc, bi);
            }
        }
    }
    if (FUSED) store_best(p, pix, bs, bc, bi);
}

// ------------------------------------------------------------------------------------------------------
// region planner: where does tile t land in side view v over the planes of chunk c?
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void plan_regions(SweepParams p, uint2 *__restrict__ plan)
{
    const int total = p.tiles_x * p.tiles_y * p.nchunks * p.V;
    const bool live = (int)(blockIdx.x * blockDim.x + threadIdx.x) < total;  // (no early return: the counters are reduced per wavefront)
    const int tid = min((int)(blockIdx.x * blockDim.x + threadIdx.x), total - 1);
    const int v = tid % p.V;
    const int rest = tid / p.V;
    const int chunk = rest % p.nchunks;
    const int tile = rest / p.nchunks;
    const int tx = tile % p.tiles_x, ty = tile / p.tiles_x;
    const int c0 = tx * TILE_W, c1 = min(c0 + TILE_W, p.W) - 1;
    const int r0 = ty * p.tile_h, r1 = min(r0 + p.tile_h, p.H) - 1;
    int too_large = 0;
    const int d0 = chunk * p.pc, d1 = min(d0 + p.pc, p.D) - 1;
    const float *q = p.Q + 12 * v;
    const float bx = q[2], by = q[6], bw = q[10];
    float xmin = 3.0e38f, xmax = -3.0e38f, ymin = 3.0e38f, ymax = -3.0e38f;
    bool behind = false;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const int col = (k & 1) ? c1 : c0;
        const int row = (k & 2) ? r1 : r0;
        const float z = p.z[(k & 4) ? d1 : d0];
        const float xn = __builtin_fmaf((float)(2 * col + 1), p.invW, -1.0f);
        const float yn = __builtin_fmaf(-(float)(2 * row + 1), p.invH, 1.0f);
        const Affine A = view_affine(q, xn, yn);
        const float sx = __builtin_fmaf(z, bx, A.ax);
        const float sy = __builtin_fmaf(z, by, A.ay);
        const float sw = __builtin_fmaf(z, bw, A.aw);
        if (!(sw > 0.0f)) behind = true;
        const float r = rcp_rn(sw);
        const float cx = sx * r, cy = sy * r;
        xmin = fminf(xmin, cx);
        xmax = fmaxf(xmax, cx);
        ymin = fminf(ymin, cy);
        ymax = fmaxf(ymax, cy);
    }
    unsigned mode;
    int x0 = 0, y0 = 0, rw = 0, rh = 0, pitch = 0;
    const float m = PLAN_MARGIN;
    if (behind || !(xmin == xmin) || !(ymin == ymin) || !(xmax < 1.0e9f) || !(ymax < 1.0e9f) || !(xmin > -1.0e9f) ||
        !(ymin > -1.0e9f)) {
        mode = R_GENERIC;
    } else if (xmax < 0.5f - m || xmin > p.Wp + m || ymax < 0.5f - m || ymin > p.Hp + m) {
        mode = R_SKIP;  // the whole warped box is out of frame: convex hull argument, DESIGN.md
    } else {
        x0 = max(0, (int)floorf(xmin - m)) & ~3;
        const int x1 = min(p.W, (int)floorf(xmax + m));
        y0 = max(0, (int)floorf(ymin - m));
        const int y1 = min(p.H, (int)floorf(ymax + m));
        rw = ((x1 - x0 + 1) + 3) & ~3;
        rh = y1 - y0 + 1;
        pitch = (rw + 31) & ~31;
        if (rw > MAX_RW || rw <= 0 || rh <= 0 || pitch * rh > LDS_QUADS) {
            mode = R_GENERIC;
            too_large = 1;
        } else if (xmin > 0.5f + m && xmax < p.Wp - m && ymin > 0.5f + m && ymax < p.Hp - m)
            mode = R_FAST;
        else
            mode = R_BORDER;
    }
    uint2 d;
    d.x = (unsigned)x0 | ((unsigned)y0 << 16);
    d.y = (unsigned)rw | ((unsigned)rh << 8) | (mode << 16) | ((unsigned)(pitch >> 5) << 24);
    if (live) plan[tid] = d;
    const int n_large = wave_sum_i32(live ? too_large : 0), n_regions = wave_sum_i32(live && mode != R_SKIP ? 1 : 0);  // one atomic per wavefront, not per thread
    if ((threadIdx.x & 63) == 0) {
        if (n_large) atomicAdd(p.plan_stats, n_large);
        if (n_regions) atomicAdd(p.plan_stats + 1, n_regions);
    }
}

// ------------------------------------------------------------------------------------------------------
// tiled kernel
// ------------------------------------------------------------------------------------------------------
// LDS region format: one 8-byte quad per texel position (y, x) of the padded side image:
//   { t00 + 0.5, t01 - t00, t10 - t00, (t11 - t10) - (t01 - t00) } as four f16 (all exactly representable),
// so the bilinear fetch is one ds_read_b64 and three v_fma_mix_f32.

// Stage the region [x0, x0+rw) x [y0, y0+rh) of a view into LDS from its precomputed f16 quad image (context.hip:
// quad16_image_views_kernel): global->LDS copies, 16 bytes = 2 quads per lane, one or two instructions per region row (rw <= 192
// quads), rows dealt to the four wavefronts; no VALU work and no registers.  The caller's __syncthreads() awaits the copies (it
// drains vmcnt).  (Round 1 built the quads from the padded bytes with permutes and packed-f16 subtractions for every (tile, chunk,
// view): 1.1 VALU instructions per sample; 2.04 -> 1.95 ms at c3.)
__device__ __forceinline__ void stage_region_q16(const uint2 *__restrict__ q16, int pitch, int x0, int y0, int rw, int rh, int rp,
                                                 uint2 *__restrict__ lds)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int pairs = rw >> 1;  // lanes a row needs
    const uint2 *src = q16 + (uint32_t)(y0 * pitch + x0) + 2 * lane;
    for (int blk = 0; blk * 64 < pairs; blk++) {
        if (lane + blk * 64 < pairs) {
            for (int ry = wave; ry < rh; ry += 4)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + (size_t)ry * pitch + blk * 128),
                                                 (__attribute__((address_space(3))) void *)(lds + ry * rp + blk * 128), 16, 0, 0);
        }
    }
}

// acc + |a - b| in one VALU op (v_sad_u32); hipcc lowers __usad() to a 4-instruction max/min/sub/add
__device__ __forceinline__ uint32_t sad_u32(uint32_t a, uint32_t b, uint32_t acc)
{
    uint32_t d;
    asm("v_sad_u32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(acc));
    return d;
}

// Branch-free LDS sample.  FAST: the planner proved every sample of this (tile, chunk, view) in
// frame, so no test is evaluated.  Otherwise the test of shader.frag:19 is evaluated per sample and
// the texel address is clamped into the staged region before the (then discarded) fetch.
struct RegionView {
    const char *lds_bytes;
    int rp8;   // LDS row pitch in bytes
    int org8;  // byte offset of padded texel (0,0) relative to the region origin (subtracted)
    int xlo, xhi, ylo, yhi;
};

template <bool FAST>
__device__ __forceinline__ uint32_t sample_lds(const Affine &A, float bx, float by, float bw, float z,
                                               const RegionView &rv, float Wp, float Hp, int Im, uint32_t acc)
{
    const float sx = __builtin_fmaf(z, bx, A.ax);
    const float sy = __builtin_fmaf(z, by, A.ay);
    const float sw = __builtin_fmaf(z, bw, A.aw);
    const float r = rcp_rn(sw);
    const float cx = sx * r, cy = sy * r;
    // c - trunc(c) for c > 0 (v_fract_f32 is exact there); garbage-in/garbage-out for masked samples
    const float fx = __builtin_amdgcn_fractf(cx), fy = __builtin_amdgcn_fractf(cy);
    int ix = (int)cx, iy = (int)cy;
    if (!FAST) {
        ix = min(max(ix, rv.xlo), rv.xhi);
        iy = min(max(iy, rv.ylo), rv.yhi);
    }
    const int off = (ix << 3) + (__mul24(iy, rv.rp8) - rv.org8);
    const half4_t h = *(const half4_t *)(rv.lds_bytes + off);
    const int Iq = bilerp_u8(fx, fy, (float)h[0], (float)h[1], (float)h[2], (float)h[3]);
    if (FAST) return sad_u32((uint32_t)Iq, (uint32_t)Im, acc);
    const bool ok = (sw > 0.0f) && (cx > 0.5f) && (cx < Wp) && (cy > 0.5f) && (cy < Hp);
    return ok ? sad_u32((uint32_t)Iq, (uint32_t)Im, acc + 65536u) : acc;
}

typedef float f32x2 __attribute__((ext_vector_type(2)));

// Two planes of one (pixel, view) at once on the FAST path.  The cost model (tools/valu_microbench,
// profiles/r01) prices a wave64 v_fma_f32 at 4 cycles and a v_pk_fma_f32 at 4.7 for two lanes, so the
// projection s = fma(z, B, A), the Newton step of the reciprocal, c = s * r and the last bilinear FMA are
// issued as packed pairs; the per-sample remainder (v_rcp, fract/cvt, address, ds_read_b64, 2 v_fma_mix,
// cvt, v_sad_u32) stays scalar.  Element-wise identical arithmetic to sample_lds<true>.
// WCONST: the view's w row does not depend on the plane (bw == 0: the side camera's centre lies in the main
// camera's focal plane -- pure sideways translation, the classic fronto-parallel sweep).  Then s.w = fma(z, 0, A.aw)
// = A.aw exactly for every plane, and r = RN(1/s.w) is the same number for all of them: it is computed once per
// (pixel, view) and passed in as r_const.  Same values bit for bit, one v_rcp_f32 + three FMAs less per sample.
template <bool WCONST>
__device__ __forceinline__ void sample_lds_pair(const Affine &A, float bx, float by, float bw, float r_const, float z0,
                                                float z1, const RegionView &rv, int negorg8, uint32_t Im,
                                                uint32_t &acc0, uint32_t &acc1)
{
    const f32x2 z = {z0, z1};
    const f32x2 sx = __builtin_elementwise_fma(z, (f32x2)(bx), (f32x2)(A.ax));
    const f32x2 sy = __builtin_elementwise_fma(z, (f32x2)(by), (f32x2)(A.ay));
    f32x2 r;
    if (WCONST) {
        r = (f32x2)(r_const);
    } else {
        const f32x2 sw = __builtin_elementwise_fma(z, (f32x2)(bw), (f32x2)(A.aw));
        const f32x2 r0 = {__builtin_amdgcn_rcpf(sw.x), __builtin_amdgcn_rcpf(sw.y)};
        const f32x2 e = __builtin_elementwise_fma(-sw, r0, (f32x2)(1.0f));
        r = __builtin_elementwise_fma(e, r0, r0);
    }
    const f32x2 cx = sx * r, cy = sy * r;
    f32x2 fy, a, b;
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const float fx = __builtin_amdgcn_fractf(cx[i]);
        fy[i] = __builtin_amdgcn_fractf(cy[i]);
        const int ix = (int)cx[i], iy = (int)cy[i];
        // byte offset of the quad: iy * pitch_bytes - origin + ix * 8, as v_mad_i32_i24 + v_lshl_add_u32
        int row_off, off;
        asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(row_off) : "v"(iy), "s"(rv.rp8), "v"(negorg8));
        asm("v_lshl_add_u32 %0, %1, 3, %2" : "=v"(off) : "v"(ix), "v"(row_off));
        const half4_t h = *(const half4_t *)(rv.lds_bytes + off);
        a[i] = __builtin_fmaf(fx, (float)h[1], (float)h[0]);
        b[i] = __builtin_fmaf(fx, (float)h[3], (float)h[2]);
    }
    const f32x2 res = __builtin_elementwise_fma(fy, b, a);
    acc0 = sad_u32((uint32_t)(int)res.x, Im, acc0);
    acc1 = sad_u32((uint32_t)(int)res.y, Im, acc1);
}

// All PC planes of one (pixel, view) on the FAST path, software-pipelined by hand: the LDS reads of plane pair k+1 are
// issued before the arithmetic of pair k, so each ds_read_b64 has ~100 cycles of this wave's own VALU work between
// issue and use instead of an immediate s_waitcnt (the compiler's schedule: profiles/r01/README.md).  The reads are
// inline asm; the wait is tied to the loaded registers ("+v") so no consumer can be scheduled above it.
// Element-wise the same arithmetic as sample_lds_pair.
typedef unsigned long long quad_bits_t;

template <int PC, bool WCONST>
__device__ __forceinline__ void sample_chunk_pipelined(const Affine &A, float bx, float by, float bw, float r_const,
                                                       const float (&zc)[PC], int rp8, int lds_minus_org8, uint32_t Im,
                                                       uint32_t (&acc)[PC])
{
    float fx[2];
    f32x2 fy;
    int off[2];
    quad_bits_t q0, q1;
    auto address_stage = [&](int k) {
        const f32x2 z = {zc[k], zc[k + 1]};
        const f32x2 sx = __builtin_elementwise_fma(z, (f32x2)(bx), (f32x2)(A.ax));
        const f32x2 sy = __builtin_elementwise_fma(z, (f32x2)(by), (f32x2)(A.ay));
        f32x2 r;
        if (WCONST) {
            r = (f32x2)(r_const);
        } else {
            const f32x2 sw = __builtin_elementwise_fma(z, (f32x2)(bw), (f32x2)(A.aw));
            const f32x2 r0 = {__builtin_amdgcn_rcpf(sw.x), __builtin_amdgcn_rcpf(sw.y)};
            const f32x2 e = __builtin_elementwise_fma(-sw, r0, (f32x2)(1.0f));
            r = __builtin_elementwise_fma(e, r0, r0);
        }
        const f32x2 cx = sx * r, cy = sy * r;
#pragma unroll
        for (int i = 0; i < 2; i++) {
            fx[i] = __builtin_amdgcn_fractf(cx[i]);
            fy[i] = __builtin_amdgcn_fractf(cy[i]);
            const int ix = (int)cx[i], iy = (int)cy[i];
            int row_off;
            asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(row_off) : "v"(iy), "s"(rp8), "v"(lds_minus_org8));
            asm("v_lshl_add_u32 %0, %1, 3, %2" : "=v"(off[i]) : "v"(ix), "v"(row_off));
        }
    };
    address_stage(0);
    asm volatile("ds_read_b64 %0, %1" : "=v"(q0) : "v"(off[0]));
    asm volatile("ds_read_b64 %0, %1" : "=v"(q1) : "v"(off[1]));
#pragma unroll
    for (int k = 0; k < PC; k += 2) {
        const float fx0 = fx[0], fx1 = fx[1];
        const f32x2 fyk = fy;
        if (k + 2 < PC) address_stage(k + 2);  // VALU work of the next pair while this pair's reads are in flight
        // the next pair's offsets are operands too: the scheduler may not sink the address stage below the wait
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(q0), "+v"(q1), "+v"(off[0]), "+v"(off[1]));
        const half4_t h0 = __builtin_bit_cast(half4_t, q0), h1 = __builtin_bit_cast(half4_t, q1);
        if (k + 2 < PC) {
            asm volatile("ds_read_b64 %0, %1" : "=v"(q0) : "v"(off[0]));
            asm volatile("ds_read_b64 %0, %1" : "=v"(q1) : "v"(off[1]));
        }
        const f32x2 a = {__builtin_fmaf(fx0, (float)h0[1], (float)h0[0]), __builtin_fmaf(fx1, (float)h1[1], (float)h1[0])};
        const f32x2 b = {__builtin_fmaf(fx0, (float)h0[3], (float)h0[2]), __builtin_fmaf(fx1, (float)h1[3], (float)h1[2])};
        const f32x2 res = __builtin_elementwise_fma(fyk, b, a);
        acc[k] = sad_u32((uint32_t)(int)res.x, Im, acc[k]);
        acc[k + 1] = sad_u32((uint32_t)(int)res.y, Im, acc[k + 1]);
    }
}

// 3 workgroups per CU (<= 168 VGPRs): 2 per CU measured 9 % slower (not enough waves to cover the two barriers per
// staged region).  The fused variants keep the running best plane of each pixel in LDS (8 KiB, touched once per
// plane chunk) instead of 8-12 more registers, which would spill under the cap.
template <int NPX, int PC, bool WRITE_VOLUME, bool FUSED>
__global__ __launch_bounds__(256, 3) void sweep_tiled(SweepParams p)
{
    constexpr int TILE_H = 4 * NPX;  // 4 wavefronts, NPX rows each
    __shared__ __attribute__((aligned(16))) uint2 lds[LDS_QUADS];
    __shared__ uint2 best_state[FUSED ? 256 * NPX : 1];  // (packed best cell, best index) per (pixel j, thread)

    const int band_tile = (p.debug & 2) ? ((int)blockIdx.x < p.tiles_x * p.tyn ? (int)blockIdx.x : -1)
                                        : grouped_tile(blockIdx.x, p.tiles_x, p.tyn);
    if (band_tile < 0) return;
    const int tx = band_tile % p.tiles_x, ty = band_tile / p.tiles_x + p.ty0;
    const int tile = ty * p.tiles_x + tx;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = tx * TILE_W + lane;
    const int row0 = ty * TILE_H + wave * NPX;
    const bool col_ok = col < p.W;
    const size_t P = (size_t)p.W * p.H;
    const float xn = __builtin_fmaf((float)(2 * col + 1), p.invW, -1.0f);

    float yn[NPX];
    int Im[NPX];
    bool ok[NPX];
#pragma unroll
    for (int j = 0; j < NPX; j++) {
        const int row = row0 + j;
        ok[j] = col_ok && row < p.H;
        yn[j] = __builtin_fmaf(-(float)(2 * row + 1), p.invH, 1.0f);
        Im[j] = ok[j] ? (int)p.main_img[(size_t)row * p.W + col] : 0;
        if (FUSED) best_state[j * 256 + threadIdx.x] = make_uint2(1u, 0xffffffffu);  // own slot only: no barrier needed
    }

    // plane split: a launch with too few tiles to fill the chip (a row band, a small frame) gives each workgroup only
    // cps of the tile's plane chunks; planes are independent, the partial bests are merged by combine_best
    const int chunk_first = p.chunk0 + (int)blockIdx.y * p.cps;
    const int chunk_last = min(p.chunk1, chunk_first + p.cps);
    for (int chunk = chunk_first; chunk < chunk_last; chunk++) {
        const int d0 = chunk * PC;
        // plane constants of this chunk live in SGPRs; planes past D (last chunk) are evaluated on a
        // clamped z and never stored, so the sample loops carry no per-plane control flow
        float zc[PC];
#pragma unroll
        for (int k = 0; k < PC; k++) zc[k] = uniform_f(p.z[min(d0 + k, p.D - 1)]);

        uint32_t acc[NPX][PC];
#pragma unroll
        for (int j = 0; j < NPX; j++)
#pragma unroll
            for (int k = 0; k < PC; k++) acc[j][k] = 0u;
        uint32_t fast_views = 0u;
        const uint2 *plan = p.plan + ((size_t)tile * p.nchunks + chunk) * p.V;

        for (int v = p.v0; v < p.v0 + p.vcount; v++) {
            const uint2 desc = plan[v];
            const unsigned mode = (unsigned)__builtin_amdgcn_readfirstlane((int)((desc.y >> 16) & 3u));
            if (mode == R_SKIP) continue;
            float q[12];
#pragma unroll
            for (int i = 0; i < 12; i++) q[i] = uniform_f(p.Q[12 * v + i]);
            const float bx = q[2], by = q[6], bw = q[10];
            const uint8_t *pad = p.pads + p.pad_slab * v;
            if (mode == R_GENERIC) {
#pragma unroll
                for (int j = 0; j < NPX; j++) {
                    if (ok[j]) {
                        const Affine A = view_affine(q, xn, yn[j]);
#pragma unroll
                        for (int k = 0; k < PC; k++)
                            acc[j][k] += sample_global(A, bx, by, bw, zc[k], pad, p.pitch, p.Wp, p.Hp, Im[j]);
                    }
                }
                continue;
            }
            const int x0 = __builtin_amdgcn_readfirstlane((int)(desc.x & 0xffffu));
            const int y0 = __builtin_amdgcn_readfirstlane((int)(desc.x >> 16));
            const int rw = __builtin_amdgcn_readfirstlane((int)(desc.y & 0xffu));
            const int rh = __builtin_amdgcn_readfirstlane((int)((desc.y >> 8) & 0xffu));
            const int rp = __builtin_amdgcn_readfirstlane((int)(desc.y >> 24) << 5);
            __syncthreads();  // all reads of the previous region are done
            if (!(p.debug & 1)) stage_region_q16(p.quads16 + p.pad_slab * v, p.pitch, x0, y0, rw, rh, rp, lds);
            __syncthreads();
            RegionView rv;
            rv.lds_bytes = (const char *)lds;
            rv.rp8 = rp * 8;
            rv.org8 = (y0 * rp + x0) * 8;
            rv.xlo = x0;
            rv.xhi = x0 + rw - 1;
            rv.ylo = y0;
            rv.yhi = y0 + rh - 1;
            if (mode == R_FAST) {
                fast_views += 65536u;
                // the low 32 bits of a generic pointer into LDS are the LDS byte address
                int lds_minus_org8 = (int)(uint32_t)(uintptr_t)lds - rv.org8;
                asm volatile("" : "+v"(lds_minus_org8));  // keep it in a VGPR so mul24 + add fuses into v_mad_i32_i24
                if (bw == 0.0f && !(p.debug & 4)) {  // wave-uniform: plane-independent w (see sample_lds_pair)
#pragma unroll
                    for (int j = 0; j < NPX; j++) {
                        if (ok[j]) {
                            const Affine A = view_affine(q, xn, yn[j]);
                            const float r_const = rcp_rn(__builtin_fmaf(zc[0], bw, A.aw));
                            sample_chunk_pipelined<PC, true>(A, bx, by, bw, r_const, zc, rv.rp8, lds_minus_org8, (uint32_t)Im[j], acc[j]);
                        }
                    }
                } else if (NPX == 2 && !(p.debug & 16)) {
                    // general cameras in the 2 x 32 shape (no spills): the hand-pipelined reads pay here too,
                    // 2.54 -> 2.39 ms at c3 (tools/ab_general.py; debug bit 4 of the high byte forces the loop below)
#pragma unroll
                    for (int j = 0; j < NPX; j++) {
                        if (ok[j]) {
                            const Affine A = view_affine(q, xn, yn[j]);
                            sample_chunk_pipelined<PC, false>(A, bx, by, bw, 0.0f, zc, rv.rp8, lds_minus_org8, (uint32_t)Im[j], acc[j]);
                        }
                    }
                } else {
                    // general cameras in the 4 x 16 shape: three more live values per pair (sw, r0, e) on top of its spills; the
                    // hand-pipelined form measured 2 % slower there (2.635 vs 2.58 ms at c3): compiler-scheduled pair loop
                    int negorg8 = -rv.org8;
                    asm volatile("" : "+v"(negorg8));
#pragma unroll
                    for (int j = 0; j < NPX; j++) {
                        if (ok[j]) {
                            const Affine A = view_affine(q, xn, yn[j]);
#pragma unroll
                            for (int k = 0; k < PC; k += 2)
                                sample_lds_pair<false>(A, bx, by, bw, 0.0f, zc[k], zc[k + 1], rv, negorg8, (uint32_t)Im[j], acc[j][k], acc[j][k + 1]);
                        }
                    }
                }
            } else {
#pragma unroll
                for (int j = 0; j < NPX; j++) {
                    if (ok[j]) {
                        const Affine A = view_affine(q, xn, yn[j]);
#pragma unroll
                        for (int k = 0; k < PC; k++)
                            acc[j][k] = sample_lds<false>(A, bx, by, bw, zc[k], rv, p.Wp, p.Hp, Im[j], acc[j][k]);
                    }
                }
            }
        }

        // chunk epilogue: as in sweep_fx_tiled (uniform plane base + 32-bit lane offset; running best started at (sum 1, count 0),
        // so one cross-multiplied comparison per plane decides, empty cells included)
        uint32_t *const vol_chunk = WRITE_VOLUME ? p.volume + (size_t)d0 * P : nullptr;
        const bool whole = d0 + PC <= p.D;
#pragma unroll
        for (int j = 0; j < NPX; j++) {
            if (ok[j]) {
                const uint32_t pix = (uint32_t)((row0 + j) * p.W + col);
                uint32_t best = 1u;
                int bi = -1;
                if (FUSED) {
                    const uint2 st = best_state[j * 256 + threadIdx.x];
                    best = st.x;
                    bi = (int)st.y;
                }
                if (whole) {
#pragma unroll
                    for (int k = 0; k < PC; k++) {
                        const uint32_t cell = acc[j][k] + fast_views;
                        if (WRITE_VOLUME) (vol_chunk + (size_t)k * P)[pix] = cell;
                        if (FUSED) {
                            const bool better = umul24u(cell & 0xffffu, best >> 16) < umul24u(best & 0xffffu, cell >> 16);
                            best = better ? cell : best;
                            bi = better ? d0 + k : bi;
                        }
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < PC; k++) {
                        if (d0 + k < p.D) {
                            const uint32_t cell = acc[j][k] + fast_views;
                            if (WRITE_VOLUME) (vol_chunk + (size_t)k * P)[pix] = cell;
                            if (FUSED) {
                                const bool better = umul24u(cell & 0xffffu, best >> 16) < umul24u(best & 0xffffu, cell >> 16);
                                best = better ? cell : best;
                                bi = better ? d0 + k : bi;
                            }
                        }
                    }
                }
                if (FUSED) best_state[j * 256 + threadIdx.x] = make_uint2(best, (uint32_t)bi);
            }
        }
    }
    if (FUSED) {
#pragma unroll
        for (int j = 0; j < NPX; j++)
            if (ok[j]) {
                uint2 st = best_state[j * 256 + threadIdx.x];
                if ((int)st.y < 0) st.x = 0u;  // no plane had a view in frame: the empty cell
                const size_t pix = (size_t)(row0 + j) * p.W + col;
                if (p.part)
                    p.part[(size_t)blockIdx.y * P + pix] = st;
                else
                    store_best(p, pix, st.x & 0xffffu, st.x >> 16, (int)st.y);
            }
    }
}

// Sub-plane refinement of the selected depth: parabola through the mean costs of the selected plane and its
// two neighbours, f32 with one rounding per operation -- oracle/sweep_oracle.c: orc_refine_depth states the arithmetic.
template <int CS>
__global__ __launch_bounds__(256) void refine_depth(const uint32_t *__restrict__ vol, size_t P, int D, const float *__restrict__ z, const int *__restrict__ index,
                                                    float *__restrict__ depth)
{
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    const int i = index[p];
    if (i < 0) {
        depth[p] = MVS_BACKGROUND_DEPTH;
        return;
    }
    constexpr uint32_t M = (1u << CS) - 1u;
    float zr = z[i];
    if (i > 0 && i < D - 1) {
        const uint32_t a = vol[(size_t)(i - 1) * P + p], b = vol[(size_t)i * P + p], c = vol[(size_t)(i + 1) * P + p];
        if ((a >> CS) != 0u && (c >> CS) != 0u && (b >> CS) != 0u) {
            const float ca = (float)(a & M) / (float)(a >> CS), cb = (float)(b & M) / (float)(b >> CS), cc = (float)(c & M) / (float)(c >> CS);
            const float den = (ca - 2.0f * cb) + cc;
            if (den > 0.0f) {
                float t = (0.5f * (ca - cc)) / den;
                t = t < -0.5f ? -0.5f : (t > 0.5f ? 0.5f : t);
                zr = t >= 0.0f ? __builtin_fmaf(t, z[i + 1] - z[i], z[i]) : __builtin_fmaf(-t, z[i - 1] - z[i], z[i]);
            }
        }
    }
    depth[p] = zr;
}

// merge of the partial bests of a plane-split launch: splits are in ascending plane order and a later split wins only
// if strictly better, so ties still go to the lowest plane
template <int CS>
__global__ __launch_bounds__(256) void combine_best(SweepParams p, int nsplit, size_t pix_first, size_t pix_count)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= pix_count) return;
    const size_t pix = pix_first + i, P = (size_t)p.W * p.H;
    uint32_t best = 0u;
    int bi = -1;
    for (int s = 0; s < nsplit; s++) {
        const uint2 st = p.part[(size_t)s * P + pix];
        argmin_update_packed<CS>(st.x, (int)st.y, best, bi);
    }
    store_best<CS>(p, pix, best & ((1u << CS) - 1u), best >> CS, bi);
}

// ------------------------------------------------------------------------------------------------------
// depth selection over the packed volume
// ------------------------------------------------------------------------------------------------------
// Layout [D][P] keeps a pixel on a lane, so the reduction over planes runs in registers; each thread
// streams 16-byte loads of 4 consecutive pixels per plane (P % 4 == 0) or single cells otherwise.
// part != nullptr: `vol` holds planes [d_first, d_first + D) only (the slice a rank owns after a reduce-scatter) and the result
// is the partial selection (best packed cell, best ABSOLUTE plane index) per pixel, to be merged by combine_best
template <int VEC, int CS>
__global__ __launch_bounds__(256) void argmin_volume(const uint32_t *__restrict__ vol, size_t P, int D,
                                                     const float *__restrict__ z, float *__restrict__ depth,
                                                     float *__restrict__ cost, int *__restrict__ index,
                                                     uint2 *__restrict__ part = nullptr, int d_first = 0)
{
    const size_t base = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * VEC;
    if (base >= P) return;
    uint32_t bs[VEC], bc[VEC];
    int bi[VEC];
#pragma unroll
    for (int i = 0; i < VEC; i++) {
        bs[i] = 0u;
        bc[i] = 0u;
        bi[i] = -1;
    }
    int d = 0;
    constexpr int UNR = 8;  // planes in flight per thread: 8 x 16 B, non-temporal (the stream is read once): 4.75 -> 5.64 TB/s at c3
    for (; d + UNR <= D; d += UNR) {
        uint32_t c[UNR][VEC];
#pragma unroll
        for (int u = 0; u < UNR; u++) {
            if (VEC == 4) {
                typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                const u32x4 t = __builtin_nontemporal_load((const u32x4 *)(vol + (size_t)(d + u) * P + base));
                c[u][0] = t.x;
                c[u][1 % VEC] = t.y;
                c[u][2 % VEC] = t.z;
                c[u][3 % VEC] = t.w;
            } else {
                c[u][0] = vol[(size_t)(d + u) * P + base];
            }
        }
#pragma unroll
        for (int u = 0; u < UNR; u++)
#pragma unroll
            for (int i = 0; i < VEC; i++) argmin_update<CS>(c[u][i], d + u, bs[i], bc[i], bi[i]);
    }
    for (; d < D; d++) {
#pragma unroll
        for (int i = 0; i < VEC; i++) argmin_update<CS>(vol[(size_t)d * P + base + i], d, bs[i], bc[i], bi[i]);
    }
    if (part) {
#pragma unroll
        for (int i = 0; i < VEC; i++)
            part[base + i] = bi[i] >= 0 ? make_uint2((bc[i] << CS) | bs[i], (uint32_t)(bi[i] + d_first)) : make_uint2(0u, 0xffffffffu);
        return;
    }
#pragma unroll
    for (int i = 0; i < VEC; i++) {
        depth[base + i] = bi[i] >= 0 ? z[bi[i]] : MVS_BACKGROUND_DEPTH;
        cost[base + i] = bi[i] >= 0 ? cell_cost<CS>(bs[i], bc[i]) : __builtin_inff();
        index[base + i] = bi[i];
    }
}

// the sweep's sampler at one plane per pixel, z = depth[p] (parity hook)
__global__ __launch_bounds__(256) void warp_by_depth_kernel(const float *__restrict__ depth, const float *__restrict__ Q,
                                                            const uint8_t *__restrict__ pad, int pitch, int W, int H,
                                                            float invW, float invH, uint8_t *__restrict__ out2)
{
    const int col = blockIdx.x * 64 + (threadIdx.x & 63);
    const int row = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (col >= W || row >= H) return;
    const size_t pix = (size_t)row * W + col;
    const float z = depth[pix];
    uint32_t cell = 0u;
    if (z != MVS_BACKGROUND_DEPTH) {
        const float xn = __builtin_fmaf((float)(2 * col + 1), invW, -1.0f);
        const float yn = __builtin_fmaf(-(float)(2 * row + 1), invH, 1.0f);
        const Affine A = view_affine(Q, xn, yn);
        cell = sample_global(A, Q[2], Q[6], Q[10], z, pad, pitch, (float)W + 0.5f, (float)H + 0.5f, 0);  // |Iq - 0| = Iq
    }
    out2[2 * pix] = (uint8_t)(cell & 0xffu);
    out2[2 * pix + 1] = cell ? 255 : 0;
}

template <int CS>
__global__ void unpack_volume(const uint32_t *__restrict__ vol, float *__restrict__ out, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t cell = vol[i];
    const uint32_t s = cell & ((1u << CS) - 1u), c = cell >> CS;
    out[i] = c ? cell_cost<CS>(s, c) : __builtin_inff();
}

// exhaustive self-check helper for the reciprocal (tests): out[0] counts w where rcp_rn(w) != 1.0f/w
__global__ void rcp_check_kernel(uint32_t exp_bits, unsigned long long *out)
{
    const uint32_t m = blockIdx.x * blockDim.x + threadIdx.x;  // 2^23 mantissas
    if (m >= (1u << 23)) return;
    const float w = __builtin_bit_cast(float, (exp_bits << 23) | m);
    const float a = rcp_rn(w);
    const float b = 1.0f / w;  // hipcc default: correctly rounded IEEE division
    if (__builtin_bit_cast(uint32_t, a) != __builtin_bit_cast(uint32_t, b)) {
        atomicAdd(out, 1ull);
        if (m == 0x7fffffu) atomicAdd(out + 1, 1ull);
    }
}

// sweep_fx.hip: the fixed-point sampler (contract v2)
int sweep_fx_plan_general(mvs_ctx *ctx);
int sweep_fx_launch(mvs_ctx *ctx, SweepParams &p, bool vol, bool fused, bool generic, unsigned flags);
int sweep_rect_launch(mvs_ctx *ctx, SweepParams &p, bool vol, bool fused, unsigned flags);
int sweep_xrect_plan(mvs_ctx *ctx);  // sweep_xrect.hip: the exact sampler on rectified views
int sweep_xrect_launch(mvs_ctx *ctx, SweepParams &p, bool vol, bool fused, unsigned flags);
int warp_by_depth_fx_launch(mvs_ctx *ctx, const float *depth_dev, const float *q_dev, const uint8_t *pad_dev, int pitch, uint8_t *out2_dev);

// defined in context.hip
__global__ void pad_wrap_kernel(const uint8_t *__restrict__ img, uint8_t *__restrict__ pad, int W, int H, int pitch);

}  // namespace mvs

using namespace mvs;

extern "C" {

int mvs_sweep_use_volume(mvs_ctx *ctx, void *device_ptr, size_t bytes)
{
    if (!ctx) return MVS_EINVAL;
    MVS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (!device_ptr) {
        ctx->volume_external = false;
        ctx->volume = (uint32_t *)ctx->volume_own.ptr;
        ctx->volume_bytes = ctx->volume_own.bytes;
        return MVS_OK;
    }
    ctx->volume_external = true;
    ctx->volume = (uint32_t *)device_ptr;
    ctx->volume_bytes = bytes;
    return MVS_OK;
}

static int ensure_outputs(mvs_ctx *ctx, bool need_volume)
{
    const size_t P = (size_t)ctx->W * ctx->H;
    int rc;
    if ((rc = ensure(ctx, ctx->depth, P * sizeof(float)))) return rc;
    if ((rc = ensure(ctx, ctx->cost, P * sizeof(float)))) return rc;
    if ((rc = ensure(ctx, ctx->index, P * sizeof(int)))) return rc;
    if (need_volume) {
        const size_t need = P * (size_t)ctx->D * sizeof(uint32_t);
        if (ctx->volume_external) {
            if (ctx->volume_bytes < need)
                return fail(ctx, MVS_EINVAL, "caller volume is %zu bytes, need %zu", ctx->volume_bytes, need);
        } else {
            if ((rc = ensure(ctx, ctx->volume_own, need))) return rc;
            ctx->volume = (uint32_t *)ctx->volume_own.ptr;
            ctx->volume_bytes = ctx->volume_own.bytes;
        }
    }
    return MVS_OK;
}

static int sweep_run_impl(mvs_ctx *ctx, int view_first, int view_count, int plane_first, int plane_count, int row_first,
                          int row_count, unsigned flags);

int mvs_sweep_run(mvs_ctx *ctx, int view_first, int view_count, unsigned flags)
{
    if (!ctx) return MVS_EINVAL;
    return sweep_run_impl(ctx, view_first, view_count, 0, ctx->D, 0, ctx->H, flags);
}

int mvs_sweep_run_rows(mvs_ctx *ctx, int view_first, int view_count, int row_first, int row_count, unsigned flags)
{
    if (!ctx) return MVS_EINVAL;
    const int gran = mvs_sweep_row_granularity_of(ctx);
    if (row_first < 0 || row_count < 0 || row_first + row_count > ctx->H || (row_first % gran) != 0 ||
        ((row_first + row_count) % gran != 0 && row_first + row_count != ctx->H))
        return fail(ctx, MVS_EINVAL, "mvs_sweep_run_rows: row range [%d,%d) must lie in 0..%d and start/end on multiples of %d",
                    row_first, row_first + row_count, ctx->H, gran);
    return sweep_run_impl(ctx, view_first, view_count, 0, ctx->D, row_first, row_count, flags);
}

int mvs_sweep_row_granularity(void) { return ROW_GRAN; }

int mvs_sweep_set_plan_cache(mvs_ctx *ctx, int enable)
{
    if (!ctx) return MVS_EINVAL;
    ctx->plan_cache = enable != 0;
    return MVS_OK;
}

// the fixed sampler's tiles are 8 rows tall whatever the plan; the exact sampler's 8 or 16 (hence 16)
int mvs_sweep_row_granularity_of(const mvs_ctx *ctx) { return (ctx && ctx->sampler == MVS_SAMPLER_FIXED) ? 8 : ROW_GRAN; }

int mvs_sweep_set_sampler(mvs_ctx *ctx, int sampler)
{
    if (!ctx) return MVS_EINVAL;
    if (sampler != MVS_SAMPLER_FIXED && sampler != MVS_SAMPLER_EXACT_F32) return fail(ctx, MVS_EINVAL, "mvs_sweep_set_sampler: unknown sampler %d", sampler);
    if (sampler != ctx->sampler) {
        MVS_HIP(ctx, hipStreamSynchronize(ctx->stream));  // queued kernels read the plan of the old sampler
        ctx->sampler = sampler;
        ctx->plan_valid = false;
        ctx->snap_valid = false;  // (the exact sampler's planner reuses the plan buffer)
    }
    return MVS_OK;
}

int mvs_sweep_sampler(const mvs_ctx *ctx) { return ctx ? ctx->sampler : MVS_EINVAL; }

int mvs_sweep_plan_shape(const mvs_ctx *ctx)
{
    if (!ctx || !ctx->plan_valid) return 0;
    if (ctx->plan_shape == 3) return ctx->rect_ok ? 4 : 3;
    return ctx->exact_last_shape;  // exact sampler: what served the last run (1 / 2: sweep_tiled's thread shapes, 5: sweep_exact_rect)
}

int mvs_sweep_run_planes(mvs_ctx *ctx, int view_first, int view_count, int plane_first, int plane_count, unsigned flags)
{
    if (!ctx) return MVS_EINVAL;
    if (flags & MVS_SWEEP_FUSED_ARGMIN)
        return fail(ctx, MVS_EINVAL, "mvs_sweep_run_planes: depth selection needs all planes; use MVS_SWEEP_VOLUME + mvs_sweep_argmin");
    if (plane_first < 0 || plane_count < 0 || plane_first + plane_count > ctx->D || (plane_first % mvs_sweep_plane_granularity()) != 0 ||
        ((plane_first + plane_count) % mvs_sweep_plane_granularity() != 0 && plane_first + plane_count != ctx->D))
        return fail(ctx, MVS_EINVAL, "mvs_sweep_run_planes: plane range [%d,%d) must lie in 0..%d and start/end on multiples of %d",
                    plane_first, plane_first + plane_count, ctx->D, mvs_sweep_plane_granularity());
    return sweep_run_impl(ctx, view_first, view_count, plane_first, plane_count, 0, ctx->H, flags);
}

int mvs_sweep_plane_granularity(void) { return PLANE_GRAN; }

static int sweep_run_impl(mvs_ctx *ctx, int view_first, int view_count, int plane_first, int plane_count, int row_first,
                          int row_count, unsigned flags)
{
    if (!ctx) return MVS_EINVAL;
    if (!ctx->have_main || !ctx->have_views || !ctx->have_planes)
        return fail(ctx, MVS_ESTATE, "mvs_sweep_run: set main view, side views and planes first");
    if (view_first < 0 || view_count < 0 || view_first + view_count > ctx->V)
        return fail(ctx, MVS_EINVAL, "mvs_sweep_run: view range [%d,%d) outside 0..%d", view_first,
                    view_first + view_count, ctx->V);
    // Only the documented bits of `flags` reach the kernels.  Bits 8-23 carry timing-experiment switches (debug bits, forced plane-split
    // count: tools/exp_*.py, tests) and are honoured only when the process sets MVS_DEBUG_FLAGS=1: a caller's stray high bits must not
    // change tile order, look-ahead or split counts silently.
    if (!ctx->hooks.debug_flags) flags &= (MVS_SWEEP_VOLUME | MVS_SWEEP_FUSED_ARGMIN | MVS_SWEEP_FORCE_GENERIC | MVS_SWEEP_NO_RECT);
    const bool vol = flags & MVS_SWEEP_VOLUME, fused = flags & MVS_SWEEP_FUSED_ARGMIN;
    if (!vol && !fused) return fail(ctx, MVS_EINVAL, "mvs_sweep_run: flags select neither volume nor fused argmin");
    MVS_HIP(ctx, hipSetDevice(ctx->device));
    int rc = ensure_outputs(ctx, vol);
    if (rc) return rc;

    const bool generic = (flags & MVS_SWEEP_FORCE_GENERIC) != 0;
    const int debug = (int)((flags >> 8) & 0xff);  // undocumented timing-experiment bits (bit 3: never use the 2 x 32 shape)
    if (row_count <= 0 || plane_count <= 0) return MVS_OK;  // empty band or empty plane group: nothing to compute

    if (ctx->sampler == MVS_SAMPLER_FIXED) {
        if (ctx->V > 255) return fail(ctx, MVS_EINVAL, "mvs_sweep_run: the fixed sampler's cells hold at most 255 views (have %d)", ctx->V);
        // 1/256-texel coordinates live in the low 22 mantissa bits of a float in [2^23, 2^24): 256 * size + 132 must stay below 2^22
        if (ctx->W > 16383 || ctx->H > 16383)
            return fail(ctx, MVS_EINVAL, "mvs_sweep_run: the fixed sampler addresses images of up to 16383 x 16383 (have %d x %d); use MVS_SAMPLER_EXACT_F32",
                        ctx->W, ctx->H);
        if (ctx->plan_valid && ctx->plan_shape != 3) ctx->plan_valid = false;
        if (!generic && !ctx->plan_valid && ctx->V > 0) {
            ProfileScope ps(ctx, MVS_K_PLAN);
            if ((rc = sweep_fx_plan(ctx))) return rc;
            ctx->plan_shape = 3;
            ctx->plan_valid = true;
        }
        if (generic && (rc = ensure_pads(ctx))) return rc;  // the un-tiled kernel gathers single texels of the padded frames
        // (a run over zero views writes empty cells: the general kernel's job -- the rectified one iterates over regions)
        const bool rect = ctx->rect_ok && !generic && ctx->V > 0 && view_count > 0 && !(flags & MVS_SWEEP_NO_RECT);
        if (!rect && !generic && ctx->V > 0 && !ctx->fx_general_planned) {  // left out while the rectified kernel served the plan; before
            ProfileScope pp(ctx, MVS_K_PLAN);                               // fill_params: it may (re)allocate the plan
            if ((rc = sweep_fx_plan_general(ctx))) return rc;
        }
        SweepParams p;
        fill_params(ctx, p, view_first, view_count, 8, 16);
        p.debug = debug;
        p.chunk0 = plane_first / 16;
        p.chunk1 = div_up(plane_first + plane_count, 16);
        p.ty0 = row_first / 8;
        p.tyn = div_up(row_first + row_count, 8) - p.ty0;
        p.row_begin = row_first;
        p.row_end = min(ctx->H, row_first + row_count);
        p.plane_begin = plane_first;
        p.plane_end = min(ctx->D, plane_first + plane_count);
        ProfileScope ps(ctx, MVS_K_SWEEP);
        const int nsplit = rect ? sweep_rect_launch(ctx, p, vol, fused, flags) : sweep_fx_launch(ctx, p, vol, fused, generic, flags);
        if (nsplit < 0) return nsplit;
        if (p.part) {
            const size_t first = (size_t)p.row_begin * ctx->W;
            const size_t count = (size_t)(p.row_end - p.row_begin) * ctx->W;
            combine_best<CS_FIXED><<<(unsigned)((count + 255) / 256), 256, 0, ctx->stream>>>(p, nsplit, first, count);
            MVS_HIP(ctx, hipGetLastError());
        }
        return MVS_OK;
    }
    if (ctx->plan_valid && ctx->plan_shape == 3) ctx->plan_valid = false;  // the plan in memory belongs to the fixed sampler
    if (!generic && (rc = ensure_quads16(ctx))) return rc;

    // Rectified views (the ring of SURVEY 8d) are served by sweep_exact_rect (sweep_xrect.hip); anything else, MVS_SWEEP_NO_RECT and the
    // forced 4 x 16 shape by sweep_tiled, whose region plan is made when it is first needed.
    // Thread shape of the tiled kernel (see the constants at the top): 2 pixels x 32 planes unless the planner finds
    // that more than 2 % of the regions a 32-plane chunk touches do not fit the LDS staging buffer
    const bool force_tall = (debug & 8) != 0;
    if (!generic && !ctx->plan_valid && ctx->V > 0) {
        ProfileScope ps(ctx, MVS_K_PLAN);
        if ((rc = sweep_xrect_plan(ctx))) return rc;
        ctx->exact_tiled_planned = false;
        ctx->plan_shape = 1;
        ctx->plan_valid = true;
    }
    const bool xrect = ctx->xrect_ok && ctx->plan_valid && !generic && ctx->V > 0 && view_count > 0 && !force_tall && !(flags & MVS_SWEEP_NO_RECT);
    if (xrect) {
        SweepParams p;
        fill_params(ctx, p, view_first, view_count, 8, 16);
        p.debug = debug;
        p.chunk0 = plane_first / 16;
        p.chunk1 = div_up(plane_first + plane_count, 16);
        p.ty0 = row_first / 8;
        p.tyn = div_up(row_first + row_count, 8) - p.ty0;
        p.row_begin = row_first;
        p.row_end = min(ctx->H, row_first + row_count);
        ProfileScope ps(ctx, MVS_K_SWEEP);
        const int nsplit = sweep_xrect_launch(ctx, p, vol, fused, flags);
        if (nsplit < 0) return nsplit;
        if (p.part) {
            const size_t first = (size_t)p.row_begin * ctx->W;
            const size_t count = (size_t)(p.row_end - p.row_begin) * ctx->W;
            combine_best<CS_EXACT><<<(unsigned)((count + 255) / 256), 256, 0, ctx->stream>>>(p, nsplit, first, count);
            MVS_HIP(ctx, hipGetLastError());
        }
        ctx->exact_last_shape = 5;
        return MVS_OK;
    }
    if ((rc = ensure_pads(ctx))) return rc;  // sweep_tiled's generic regions and the un-tiled kernel gather single texels of the padded frames
    if (ctx->exact_tiled_planned && ctx->plan_forced != force_tall) ctx->exact_tiled_planned = false;
    if (!generic && !ctx->exact_tiled_planned && ctx->V > 0) {
        ctx->snap_valid = false;  // ctx->plan is about to hold the exact sampler's regions
        if ((rc = ensure(ctx, ctx->plan_stats, 64))) return rc;
        int *stats = (int *)ctx->plan_stats.ptr;
        ProfileScope ps(ctx, MVS_K_PLAN);
        for (int shape = force_tall ? 2 : 1; shape <= 2; shape++) {
            SweepParams q;
            fill_params(ctx, q, 0, ctx->V, shape == 1 ? 8 : 16, shape == 1 ? 32 : 16);
            const size_t n = (size_t)q.tiles_x * q.tiles_y * q.nchunks * q.V;
            if ((rc = ensure(ctx, ctx->plan, n * sizeof(uint2)))) return rc;
            q.plan = (const uint2 *)ctx->plan.ptr;
            q.plan_stats = stats;
            MVS_HIP(ctx, hipMemsetAsync(stats, 0, 2 * sizeof(int), ctx->stream));
            plan_regions<<<(unsigned)((n + 255) / 256), 256, 0, ctx->stream>>>(q, (uint2 *)ctx->plan.ptr);
            MVS_HIP(ctx, hipGetLastError());
            ctx->plan_shape = shape;
            if (shape == 2) break;
            int h[2] = {0, 0};
            MVS_HIP(ctx, hipMemcpyAsync(h, stats, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
            MVS_HIP(ctx, hipStreamSynchronize(ctx->stream));
            if ((long long)h[0] * 50 <= (long long)h[1]) break;  // at most 2 % oversize: keep 2 x 32
        }
        ctx->exact_tiled_planned = true;
        ctx->plan_forced = force_tall;
    }
    ctx->exact_last_shape = ctx->plan_shape;
    const int shape = (generic || ctx->V == 0) ? 2 : ctx->plan_shape;
    const int tile_h = shape == 1 ? 8 : 16, pc = shape == 1 ? 32 : 16;

    SweepParams p;
    fill_params(ctx, p, view_first, view_count, tile_h, pc);
    p.debug = debug;
    p.chunk0 = plane_first / pc;
    p.chunk1 = div_up(plane_first + plane_count, pc);
    p.ty0 = row_first / tile_h;
    p.tyn = div_up(row_first + row_count, tile_h) - p.ty0;
    p.row_begin = row_first;
    p.row_end = min(ctx->H, row_first + row_count);
    p.plane_begin = plane_first;
    p.plane_end = min(ctx->D, plane_first + plane_count);
    {
        ProfileScope ps(ctx, MVS_K_SWEEP);
        if (generic || ctx->V == 0) {
            dim3 grid(div_up(ctx->W, 64), div_up(p.row_end - p.row_begin, 4));
            if (vol && fused)
                sweep_generic<true, true><<<grid, 256, 0, ctx->stream>>>(p);
            else if (vol)
                sweep_generic<true, false><<<grid, 256, 0, ctx->stream>>>(p);
            else
                sweep_generic<false, true><<<grid, 256, 0, ctx->stream>>>(p);
        } else {
            // padded group grid: 8 tiles per group, 8 groups per 64-id super block
            const int groups = div_up(p.tiles_x, 2) * div_up(p.tyn, 4);
            // plane split: aim at ~32 workgroups per CU (64 in round 1).  Finer work units smooth the tail of the launch and keep small
            // frames / row bands from leaving CUs idle: c3 2.50 -> 2.31 ms, c2 0.414 -> 0.333 ms, c1 0.080 -> 0.055 ms,
            // a 1/8 row band of c3 0.86 -> 0.4 ms; flat at c4 (8100 tiles) -- profiles/r01/exp_split.json
            const int nch = p.chunk1 - p.chunk0, tiles = p.tiles_x * p.tyn;
            int want = (int)((flags >> 16) & 0xffu);  // undocumented: forced split count for timing experiments
            if (!want) want = div_up(32 * ctx->num_cus, tiles);  // round 2: 32 instead of 64 workgroups per CU (c3: 2 splits 1.92 ms, 4 splits 1.94)
            p.cps = div_up(nch, max(1, min(want, nch)));
            const int nsplit = div_up(nch, p.cps);
            if (fused && nsplit > 1) {
                if ((rc = ensure(ctx, ctx->best_parts, (size_t)nsplit * ctx->W * ctx->H * sizeof(uint2)))) return rc;
                p.part = (uint2 *)ctx->best_parts.ptr;
            }
            const dim3 grid((unsigned)(div_up(groups, 8) * 64), (unsigned)nsplit);
            if (shape == 1) {
                if (vol && fused)
                    sweep_tiled<2, 32, true, true><<<grid, 256, 0, ctx->stream>>>(p);
                else if (vol)
                    sweep_tiled<2, 32, true, false><<<grid, 256, 0, ctx->stream>>>(p);
                else
                    sweep_tiled<2, 32, false, true><<<grid, 256, 0, ctx->stream>>>(p);
            } else {
                if (vol && fused)
                    sweep_tiled<4, 16, true, true><<<grid, 256, 0, ctx->stream>>>(p);
                else if (vol)
                    sweep_tiled<4, 16, true, false><<<grid, 256, 0, ctx->stream>>>(p);
                else
                    sweep_tiled<4, 16, false, true><<<grid, 256, 0, ctx->stream>>>(p);
            }
            if (p.part) {
                const size_t first = (size_t)p.row_begin * ctx->W;
                const size_t count = (size_t)(p.row_end - p.row_begin) * ctx->W;
                combine_best<CS_EXACT><<<(unsigned)((count + 255) / 256), 256, 0, ctx->stream>>>(p, nsplit, first, count);
            }
        }
        MVS_HIP(ctx, hipGetLastError());
    }
    return MVS_OK;
}

// depth selection over `vol` with the cell layout of the context's sampler; 16-byte loads when the pixel count and the pointer allow
static void launch_argmin(mvs_ctx *ctx, const uint32_t *vol, size_t P, int D, const float *z, float *depth, float *cost, int *index, uint2 *part,
                          int d_first)
{
    const bool wide = P % 4 == 0 && ((uintptr_t)vol % 16) == 0;
    const unsigned blocks = (unsigned)(((wide ? P / 4 : P) + 255) / 256);
    const bool fx = ctx->sampler == MVS_SAMPLER_FIXED;
    if (wide && fx)
        argmin_volume<4, CS_FIXED><<<blocks, 256, 0, ctx->stream>>>(vol, P, D, z, depth, cost, index, part, d_first);
    else if (wide)
        argmin_volume<4, CS_EXACT><<<blocks, 256, 0, ctx->stream>>>(vol, P, D, z, depth, cost, index, part, d_first);
    else if (fx)
        argmin_volume<1, CS_FIXED><<<blocks, 256, 0, ctx->stream>>>(vol, P, D, z, depth, cost, index, part, d_first);
    else
        argmin_volume<1, CS_EXACT><<<blocks, 256, 0, ctx->stream>>>(vol, P, D, z, depth, cost, index, part, d_first);
}

int mvs_sweep_argmin(mvs_ctx *ctx)
{
    if (!ctx) return MVS_EINVAL;
    if (!ctx->have_planes || !ctx->volume)
        return fail(ctx, MVS_ESTATE, "mvs_sweep_argmin: no cost volume (run mvs_sweep_run with MVS_SWEEP_VOLUME)");
    MVS_HIP(ctx, hipSetDevice(ctx->device));
    int rc = ensure_outputs(ctx, true);
    if (rc) return rc;
    const size_t P = (size_t)ctx->W * ctx->H;
    ProfileScope ps(ctx, MVS_K_ARGMIN);
    launch_argmin(ctx, ctx->volume, P, ctx->D, (const float *)ctx->ztab.ptr, (float *)ctx->depth.ptr, (float *)ctx->cost.ptr, (int *)ctx->index.ptr,
                  nullptr, 0);
    MVS_HIP(ctx, hipGetLastError());
    return MVS_OK;
}

int mvs_sweep_refine_depth(mvs_ctx *ctx)
{
    if (!ctx) return MVS_EINVAL;
    if (!ctx->have_planes || !ctx->volume || !ctx->index.ptr)
        return fail(ctx, MVS_ESTATE, "mvs_sweep_refine_depth: needs the packed volume and a depth selection (MVS_SWEEP_VOLUME | MVS_SWEEP_FUSED_ARGMIN, or mvs_sweep_argmin)");
    MVS_HIP(ctx, hipSetDevice(ctx->device));
    const size_t P = (size_t)ctx->W * ctx->H;
    if (ctx->sampler == MVS_SAMPLER_FIXED)
        refine_depth<CS_FIXED><<<(unsigned)((P + 255) / 256), 256, 0, ctx->stream>>>(ctx->volume, P, ctx->D, (const float *)ctx->ztab.ptr, (const int *)ctx->index.ptr, (float *)ctx->depth.ptr);
    else
        refine_depth<CS_EXACT><<<(unsigned)((P + 255) / 256), 256, 0, ctx->stream>>>(ctx->volume, P, ctx->D, (const float *)ctx->ztab.ptr, (const int *)ctx->index.ptr, (float *)ctx->depth.ptr);
    MVS_HIP(ctx, hipGetLastError());
    return MVS_OK;
}

int mvs_sweep_argmin_partial(mvs_ctx *ctx, const void *volume_slice_dev, int plane_first, int plane_count, void *partial_out_dev)
{
    if (!ctx || !volume_slice_dev || !partial_out_dev) return fail(ctx, MVS_EINVAL, "mvs_sweep_argmin_partial: null argument");
    if (!ctx->have_planes) return fail(ctx, MVS_ESTATE, "mvs_sweep_argmin_partial: set the planes first");
    if (plane_first < 0 || plane_count <= 0 || plane_first + plane_count > ctx->D)
        return fail(ctx, MVS_EINVAL, "mvs_sweep_argmin_partial: planes [%d,%d) outside 0..%d", plane_first, plane_first + plane_count, ctx->D);
    MVS_HIP(ctx, hipSetDevice(ctx->device));
    const size_t P = (size_t)ctx->W * ctx->H;
    ProfileScope ps(ctx, MVS_K_ARGMIN);
    launch_argmin(ctx, (const uint32_t *)volume_slice_dev, P, plane_count, nullptr, nullptr, nullptr, nullptr, (uint2 *)partial_out_dev, plane_first);
    MVS_HIP(ctx, hipGetLastError());
    return MVS_OK;
}

int mvs_sweep_combine_partials(mvs_ctx *ctx, const void *partials_dev, int nparts)
{
    if (!ctx || !partials_dev || nparts <= 0) return fail(ctx, MVS_EINVAL, "mvs_sweep_combine_partials: bad argument");
    if (!ctx->have_planes) return fail(ctx, MVS_ESTATE, "mvs_sweep_combine_partials: set the planes first");
    MVS_HIP(ctx, hipSetDevice(ctx->device));
    int rc = ensure_outputs(ctx, false);
    if (rc) return rc;
    SweepParams p;
    fill_params(ctx, p, 0, 0, 16, 16);
    p.part = (uint2 *)partials_dev;  // read-only here
    const size_t P = (size_t)ctx->W * ctx->H;
    ProfileScope ps(ctx, MVS_K_ARGMIN);
    if (ctx->sampler == MVS_SAMPLER_FIXED)
        combine_best<CS_FIXED><<<(unsigned)((P + 255) / 256), 256, 0, ctx->stream>>>(p, nparts, 0, P);
    else
        combine_best<CS_EXACT><<<(unsigned)((P + 255) / 256), 256, 0, ctx->stream>>>(p, nparts, 0, P);
    MVS_HIP(ctx, hipGetLastError());
    return MVS_OK;
}

void *mvs_sweep_volume_device(mvs_ctx *ctx, size_t *bytes)
{
    if (!ctx || !ctx->have_planes) return nullptr;
    if (ensure_outputs(ctx, true) != MVS_OK) return nullptr;
    if (bytes) *bytes = (size_t)ctx->W * ctx->H * (size_t)ctx->D * sizeof(uint32_t);
    return ctx->volume;
}

void *mvs_sweep_depth_device(mvs_ctx *ctx) { return ctx ? ctx->depth.ptr : nullptr; }
void *mvs_sweep_cost_device(mvs_ctx *ctx) { return ctx ? ctx->cost.ptr : nullptr; }
void *mvs_sweep_index_device(mvs_ctx *ctx) { return ctx ? ctx->index.ptr : nullptr; }

int mvs_sweep_fetch(mvs_ctx *ctx, float *depth_hw, float *cost_hw, int32_t *index_hw, uint32_t *packed_volume_dhw)
{
    if (!ctx) return MVS_EINVAL;
    MVS_HIP(ctx, hipSetDevice(ctx->device));
    const size_t P = (size_t)ctx->W * ctx->H;
    if ((depth_hw || cost_hw || index_hw) && !ctx->depth.ptr)
        return fail(ctx, MVS_ESTATE, "mvs_sweep_fetch: no results yet");
    if (depth_hw) MVS_HIP(ctx, hipMemcpyAsync(depth_hw, ctx->depth.ptr, P * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (cost_hw) MVS_HIP(ctx, hipMemcpyAsync(cost_hw, ctx->cost.ptr, P * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (index_hw) MVS_HIP(ctx, hipMemcpyAsync(index_hw, ctx->index.ptr, P * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (packed_volume_dhw) {
        if (!ctx->volume) return fail(ctx, MVS_ESTATE, "mvs_sweep_fetch: no volume");
        MVS_HIP(ctx, hipMemcpyAsync(packed_volume_dhw, ctx->volume, P * ctx->D * 4, hipMemcpyDeviceToHost, ctx->stream));
    }
    MVS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return MVS_OK;
}

// ---- the one-call mvs_sweep as a pipeline of row bands ---------------------------------------------------------------------------
// RenderHIP users reach mvs_sweep first, and there the side frames' trip over PCIe is most of the call (c3: 33 MB up, 0.6 ms of sweep, 8 MB
// down).  The fixed sampler's sweep runs on row bands (mvs_sweep_run_rows) and a band of main rows samples a band of side rows, so the call
// is cut into bands: the side rows band b needs go up on the copy stream, their quad rows are built and the band is swept on the context's
// stream while the rows of band b + 1 cross the bus, and the band's depths come back as soon as it is done.  WHICH side rows a band needs
// is bounded on the host: the sample row sy / sw is a ratio of two functions affine in (xn, yn, z), so over the box (all columns) x (the
// band's rows) x (all planes) with sw > 0 at its eight corners it takes its extremes at the corners; +- 1 row covers the f32 / 1/256-pixel
// rounding of the kernels (a row that is copied into a region but not sampled may hold anything).  A corner behind a side camera, the exact
// sampler, a volume request or a small image take the unbanded path.  Results are the same bits either way (tests/test_sweep_gpu.py).
struct BandPipeline : PlanHook {
    static constexpr int kMaxBands = 8;
    mvs_ctx *c = nullptr;
    const uint8_t *const *frames = nullptr;
    int nb = 0;
    int row0[kMaxBands + 1] = {};            // main rows of band b: [row0[b], row0[b + 1])
    int qa[kMaxBands] = {}, qb[kMaxBands] = {};  // quad rows band b may sample: [qa, qb)
    int up_lo = 0, up_hi = 0;                // raw rows [up_lo, up_hi) of every view are queued into the staging block ...
    bool wrap_top = false, wrap_bot = false; // ... and raw row H - 1 / raw row 0 on their own (the wrap rows of quad rows 0 and H)
    int q_lo = 0, q_hi = 0;                  // quad rows [q_lo, q_hi) are queued for building
    int staged = 0;                          // bands whose rows have been queued on the copy stream

    hipEvent_t up(int b) const { return c->band_events[2 * b]; }
    hipEvent_t swept(int b) const { return c->band_events[2 * b + 1]; }

    // false: some view may see part of the sweep from behind (sw <= 0): no bound, no pipeline
    bool plan_rows()
    {
        const int W = c->W, H = c->H, D = c->D;
        const double zc[2] = {c->z_host[0], c->z_host[D - 1]};
        const double xc[2] = {1.0 / W - 1.0, (2.0 * W - 1.0) / W - 1.0};
        for (int b = 0; b < nb; b++) {
            const double yc[2] = {1.0 - (2.0 * row0[b] + 1.0) / H, 1.0 - (2.0 * row0[b + 1] - 1.0) / H};
            int lo = H + 2, hi = -1;
            for (int v = 0; v < c->V; v++) {
                const float *q = c->q_host.data() + 12 * v;
                double cmin = 1e300, cmax = -1e300;
                for (int k = 0; k < 8; k++) {
                    const double x = xc[k & 1], y = yc[(k >> 1) & 1], z = zc[k >> 2];
                    const double sy = q[4] * x + q[5] * y + q[6] * z + q[7], sw = q[8] * x + q[9] * y + q[10] * z + q[11];
                    // The kernels form sy, sw and sy / sw in f32 (FMAs + a correctly rounded reciprocal): each of the two sums carries a few ulps
                    // of its LARGEST term, so when sw is the small difference of large terms -- a side camera whose centre lies just outside
                    // the sweep box -- the f32 row can differ from this double bound by more than the fixed margin.  Such a view takes the
                    // unbanded path: sw must be well conditioned at every corner, and what rounding is left goes into `slack`.
                    const double wmag = fabs(q[8] * x) + fabs(q[9] * y) + fabs(q[10] * z) + fabs(q[11]);
                    const double ymag = fabs(q[4] * x) + fabs(q[5] * y) + fabs(q[6] * z) + fabs(q[7]);
                    if (!(sw > 1e-30) || !(sw >= 1e-3 * wmag)) return false;
                    const double cy = sy / sw;
                    if (!(cy == cy)) return false;
                    // |d(sy / sw)| <= (|d sy| + |cy| |d sw|) / sw with |d s| <= 4 ulp_f32 x the sum's magnitude
                    const double slack = 4.0 * 5.97e-8 * (ymag + fabs(cy) * wmag) / sw;
                    cmin = cy - slack < cmin ? cy - slack : cmin;
                    cmax = cy + slack > cmax ? cy + slack : cmax;
                }
                if (cmax < 0.25 || cmin > H + 0.75) continue;  // (in the frame: 0.5 < cy < H + 0.5) this view sees nothing of the band
                const int a = (int)floor(cmin < 0.0 ? 0.0 : cmin) - 1, e = (int)floor(cmax > H + 1.0 ? H + 1.0 : cmax) + 2;
                lo = a < lo ? a : lo;
                hi = e > hi ? e : hi;
            }
            qa[b] = lo < 0 ? 0 : lo;
            qb[b] = hi > H + 1 ? H + 1 : hi;  // quad row H + 1 is never sampled
            if (qb[b] < qa[b]) qa[b] = qb[b] = 0;
        }
        return true;
    }

    int upload(int r0, int r1)
    {
        if (wrap_bot && r0 == 0) r0 = 1;  // already there: not written a second time under the eyes of a running band
        if (wrap_top && r1 == c->H) r1 = c->H - 1;
        return sweep_upload_rows_impl(c, frames, r0, r1, c->copy_stream);
    }

    // copy stream: the raw rows quad rows [a, e) read and the staging block does not hold yet; then the "rows of band b are up" event
    int stage(int b, int a, int e)
    {
        const int H = c->H;
        int rc;
        if (e > a) {
            const bool have_q = q_hi > q_lo, have_r = up_hi > up_lo;
            const int na = have_q && q_lo < a ? q_lo : a, ne = have_q && q_hi > e ? q_hi : e;  // hull of the quad rows once this band is built
            const int r0 = na > 0 ? na - 1 : 0, r1 = ne < H ? ne : H;                           // quad row y reads raw rows y - 1 and y (wrapped)
            const int h0 = have_r && up_lo < r0 ? up_lo : r0, h1 = have_r && up_hi > r1 ? up_hi : r1;
            if (na == 0 && h1 != H && !wrap_top) {  // quad row 0 reads raw row H - 1
                if ((rc = sweep_upload_rows_impl(c, frames, H - 1, H, c->copy_stream))) return rc;
                wrap_top = true;
            }
            if (ne > H && h0 != 0 && !wrap_bot) {   // quad row H reads raw row 0
                if ((rc = sweep_upload_rows_impl(c, frames, 0, 1, c->copy_stream))) return rc;
                wrap_bot = true;
            }
            if (!have_r) {
                if ((rc = upload(h0, h1))) return rc;
            } else {
                if (h0 < up_lo && (rc = upload(h0, up_lo))) return rc;
                if (h1 > up_hi && (rc = upload(up_hi, h1))) return rc;
            }
            up_lo = h0;
            up_hi = h1;
        }
        MVS_HIP(c, hipEventRecord(up(b), c->copy_stream));
        return MVS_OK;
    }

    // context's stream: quad rows [a, e) that are not built yet, once their raw rows have landed
    int build(int b, int a, int e)
    {
        int rc;
        MVS_HIP(c, hipStreamWaitEvent(c->stream, up(b), 0));
        if (e <= a) return MVS_OK;
        if (q_hi <= q_lo) {
            if ((rc = sweep_build_quads_impl(c, a, e))) return rc;
            q_lo = a;
            q_hi = e;
            return MVS_OK;
        }
        if (a < q_lo) {
            if ((rc = sweep_build_quads_impl(c, a, q_lo))) return rc;
            q_lo = a;
        }
        if (e > q_hi) {
            if ((rc = sweep_build_quads_impl(c, q_hi, e))) return rc;
            q_hi = e;
        }
        return MVS_OK;
    }

    // the planner's hook: band 0's rows cross the bus while the host waits for the planner's counters
    int run() override
    {
        staged = 1;
        return stage(0, qa[0], qb[0]);
    }
};

static int sweep_banded(mvs_ctx *ctx, BandPipeline &bp, float *depth_hw, float *cost_hw)
{
    const int W = ctx->W, H = ctx->H;
    int rc;
    if (!bp.staged && (rc = bp.run())) return rc;  // (the planner had nothing to wait for: plan cache hit)
    for (int b = 0; b < bp.nb; b++) {
        if (b > 0 && (rc = bp.stage(b, bp.qa[b], bp.qb[b]))) return rc;
        if ((rc = bp.build(b, bp.qa[b], bp.qb[b]))) return rc;
        if ((rc = mvs_sweep_run_rows(ctx, 0, ctx->V, bp.row0[b], bp.row0[b + 1] - bp.row0[b], MVS_SWEEP_FUSED_ARGMIN))) return rc;
        MVS_HIP(ctx, hipEventRecord(bp.swept(b), ctx->stream));
        // the band before this one comes home now: its sweep has had a band's upload of time, and the copy stream is idle until the next stage
        const int done = b - 1;
        if (done >= 0) {
            const size_t off = (size_t)bp.row0[done] * W, n = (size_t)(bp.row0[done + 1] - bp.row0[done]) * W;
            MVS_HIP(ctx, hipStreamWaitEvent(ctx->copy_stream, bp.swept(done), 0));
            MVS_HIP(ctx, hipMemcpyAsync(depth_hw + off, (const float *)ctx->depth.ptr + off, n * 4, hipMemcpyDeviceToHost, ctx->copy_stream));
            if (cost_hw) MVS_HIP(ctx, hipMemcpyAsync(cost_hw + off, (const float *)ctx->cost.ptr + off, n * 4, hipMemcpyDeviceToHost, ctx->copy_stream));
        }
    }
    // whatever no band asked for: the context's views are complete when the call returns (mvs_sweep_run on them must work)
    if ((rc = bp.stage(bp.nb, 0, H + 2))) return rc;
    if ((rc = bp.build(bp.nb, 0, H + 2))) return rc;
    {
        const int done = bp.nb - 1;
        const size_t off = (size_t)bp.row0[done] * W, n = (size_t)(bp.row0[done + 1] - bp.row0[done]) * W;
        MVS_HIP(ctx, hipStreamWaitEvent(ctx->copy_stream, bp.swept(done), 0));
        MVS_HIP(ctx, hipMemcpyAsync(depth_hw + off, (const float *)ctx->depth.ptr + off, n * 4, hipMemcpyDeviceToHost, ctx->copy_stream));
        if (cost_hw) MVS_HIP(ctx, hipMemcpyAsync(cost_hw + off, (const float *)ctx->cost.ptr + off, n * 4, hipMemcpyDeviceToHost, ctx->copy_stream));
    }
    MVS_HIP(ctx, hipStreamSynchronize(ctx->copy_stream));
    MVS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return MVS_OK;
}

// test hook (not in mvs.h; tests/test_sweep_gpu.py): the number of row bands the last mvs_sweep of this context went through (0: unbanded)
int mvs_test_onecall_bands(const mvs_ctx *ctx) { return ctx ? ctx->onecall_bands_last : MVS_EINVAL; }

int mvs_sweep(mvs_ctx *ctx, const float main_cam[16], const uint8_t *main_hw, int nviews, const float *side_cams,
              const uint8_t *const *side_frames, int nplanes, float z_lo, float z_hi, float *depth_hw, float *cost_hw,
              float *volume_dhw)
{
    if (!ctx) return MVS_EINVAL;
    if (!depth_hw) return fail(ctx, MVS_EINVAL, "mvs_sweep: depth_hw is null");
    // whatever a later step would reject is rejected here, before the first copy of a caller buffer is queued
    if (nviews < 0 || nviews > 256 || (nviews > 0 && (!side_cams || !side_frames))) return fail(ctx, MVS_EINVAL, "mvs_sweep: bad arguments (nviews=%d, must be 0..256)", nviews);
    if (ctx->sampler == MVS_SAMPLER_FIXED && nviews > 255) return fail(ctx, MVS_EINVAL, "mvs_sweep: the fixed sampler's cells hold at most 255 views (have %d)", nviews);
    if (nplanes < 1 || nplanes > 4096) return fail(ctx, MVS_EINVAL, "mvs_sweep: nplanes=%d out of range 1..4096", nplanes);
    for (int v = 0; v < nviews; v++)
        if (!side_frames[v]) return fail(ctx, MVS_EINVAL, "mvs_sweep: side_frames[%d] is null", v);
    int rc;
    // Everything below is queued on the stream without intermediate waits (the q / z host tables live in the context; pageable
    // uploads return once staged); mvs_sweep_fetch at the end is the one synchronisation of the call.  An early return must not
    // leave copies of the caller's frames in flight (with mvs_host_alloc buffers they are truly asynchronous) nor the context
    // claiming views whose frames never arrived: the guard joins the stream and drops the half-set inputs.
    struct JoinOnError {
        mvs_ctx *c;
        bool armed = true;
        hipStream_t copy = nullptr;  // the band pipeline's second stream
        ~JoinOnError()
        {
            if (!armed) return;
            if (copy) (void)hipStreamSynchronize(copy);
            (void)hipStreamSynchronize(c->stream);
            c->have_views = false;
            c->plan_valid = false;
            c->quads16_valid = false;
        }
    } join{ctx};
    ctx->onecall_bands_last = 0;
    if ((rc = sweep_set_main_impl(ctx, main_cam, main_hw, false))) return rc;
    if ((rc = sweep_set_views_impl(ctx, nviews, side_cams, side_frames, false, true))) return rc;  // tables only: the frames follow below
    if ((rc = sweep_set_planes_impl(ctx, nplanes, z_lo, z_hi, false))) return rc;
    // the region plan needs the cameras and the planes, not the frames: queued first, it runs while the host stages the uploads
    // (and the uploads are queued BEFORE the host waits for the rectified planner's counters: they run during that round trip)
    struct Upload : PlanHook {
        mvs_ctx *c;
        const uint8_t *const *frames;
        int run() override { return sweep_upload_frames_impl(c, frames); }
    } upload;
    upload.c = ctx;
    upload.frames = side_frames;
    if (ctx->sampler == MVS_SAMPLER_FIXED && nviews > 0 && nviews <= 255 && ctx->W <= 16383 && ctx->H <= 16383) {
        // the band pipeline (above): when nothing but depths and costs is asked for and every band's side rows can be bounded
        BandPipeline bands;
        // Two bands by default: every band costs one more copy per view (~9 us of submission each, page-locked or not), which eats what a
        // third band's overlap would win (c3, 16 views, ms per call: 1 band 1.90, 2: 1.67, 3: 1.98, 4: 2.04, 8: 2.90; profiles/r05/onecall_bands.json)
        const int want = ctx->hooks.onecall_bands > 0 ? ctx->hooks.onecall_bands : 2;
        bands.nb = want > BandPipeline::kMaxBands ? BandPipeline::kMaxBands : want;
        if (bands.nb > ctx->H / 64) bands.nb = ctx->H / 64;  // a band is at least 64 rows
        // (small view sets: the overlap would not pay for the extra copies -- one per view and band at ~9 us each; 16 MB of side frames = 0.4 ms of upload)
        if (ctx->hooks.onecall_bands <= 0 && (ctx->H < 256 || (size_t)nviews * (size_t)ctx->W * (size_t)ctx->H < ((size_t)16 << 20))) bands.nb = 0;
        if (volume_dhw) bands.nb = 0;
        if (bands.nb >= 2) {
            bands.c = ctx;
            bands.frames = side_frames;
            const int per = div_up(div_up(ctx->H, 8), bands.nb) * 8;  // mvs_sweep_run_rows: bands start on multiples of 8 rows
            for (int b = 0; b <= bands.nb; b++) bands.row0[b] = b * per < ctx->H ? b * per : ctx->H;
            // two bands: the second one's sweep and download are the tail nothing overlaps, so it is the smaller one (c3: 1.64 ms at 50 %, 1.60 at 58 %, 1.65 at 72 %)
            if (bands.nb == 2) bands.row0[1] = (int)((long long)ctx->H * (ctx->hooks.onecall_first_permille > 0 ? ctx->hooks.onecall_first_permille : 580) / 8000) * 8;
            if (bands.row0[bands.nb - 1] >= ctx->H || !bands.plan_rows()) bands.nb = 0;
        }
        if (bands.nb >= 2) {
            if (!ctx->copy_stream) MVS_HIP(ctx, hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
            while ((int)ctx->band_events.size() < 2 * (BandPipeline::kMaxBands + 1)) {
                hipEvent_t e = nullptr;
                MVS_HIP(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
                ctx->band_events.push_back(e);
            }
            join.copy = ctx->copy_stream;
            // the staging block may still be read by what an earlier call queued on the context's stream
            MVS_HIP(ctx, hipEventRecord(bands.swept(bands.nb), ctx->stream));
            MVS_HIP(ctx, hipStreamWaitEvent(ctx->copy_stream, bands.swept(bands.nb), 0));
            {
                ProfileScope ps(ctx, MVS_K_PLAN);
                if ((rc = sweep_fx_plan(ctx, &bands))) return rc;
                ctx->plan_shape = 3;
                ctx->plan_valid = true;
            }
            if ((rc = sweep_banded(ctx, bands, depth_hw, cost_hw))) return rc;
            ctx->onecall_bands_last = bands.nb;
            join.armed = false;
            return MVS_OK;
        }
        ProfileScope ps(ctx, MVS_K_PLAN);
        if ((rc = sweep_fx_plan(ctx, &upload))) return rc;
        ctx->plan_shape = 3;
        ctx->plan_valid = true;
    } else if ((rc = upload.run())) {
        return rc;
    }
    const unsigned flags = MVS_SWEEP_FUSED_ARGMIN | (volume_dhw ? MVS_SWEEP_VOLUME : 0u);
    if ((rc = mvs_sweep_run(ctx, 0, nviews, flags))) return rc;
    if ((rc = mvs_sweep_fetch(ctx, depth_hw, cost_hw, nullptr, nullptr))) return rc;
    if (volume_dhw) {
        const size_t n = (size_t)ctx->W * ctx->H * (size_t)nplanes;
        if ((rc = ensure(ctx, ctx->r_tmp0, n * sizeof(float)))) return rc;
        if (ctx->sampler == MVS_SAMPLER_FIXED)
            unpack_volume<CS_FIXED><<<(unsigned)((n + 255) / 256), 256, 0, ctx->stream>>>(ctx->volume, (float *)ctx->r_tmp0.ptr, n);
        else
            unpack_volume<CS_EXACT><<<(unsigned)((n + 255) / 256), 256, 0, ctx->stream>>>(ctx->volume, (float *)ctx->r_tmp0.ptr, n);
        MVS_HIP(ctx, hipGetLastError());
        MVS_HIP(ctx, hipMemcpyAsync(volume_dhw, ctx->r_tmp0.ptr, n * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
        MVS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    join.armed = false;
    return MVS_OK;
}

// One main view whose frames are already slots of the frame store (mvs_frame_store / mvs_frame_upload): the loop of recon.cpp:65-117
// with the sequence's frames cached on the device.  No frame crosses PCIe and none is copied or re-prepared -- the quad images were
// built at upload; the call pays the view matrices, the region plan, the sweep with depth selection and the depth download.
int mvs_sweep_handles(mvs_ctx *ctx, int main_slot, const float main_cam[16], int nside, const int *side_slots, const float *side_cams, int nplanes, float z_lo,
                      float z_hi, float *depth_hw, float *cost_hw)
{
    if (!ctx) return MVS_EINVAL;
    if (!main_cam || !depth_hw || nside < 0 || nside > 255 || (nside > 0 && (!side_slots || !side_cams)))
        return fail(ctx, MVS_EINVAL, "mvs_sweep_handles: bad arguments (nside=%d: 0..255 side views)", nside);
    if (nplanes < 1 || nplanes > 4096) return fail(ctx, MVS_EINVAL, "mvs_sweep_handles: nplanes=%d out of range 1..4096", nplanes);
    if (ctx->sampler != MVS_SAMPLER_FIXED) return fail(ctx, MVS_EINVAL, "mvs_sweep_handles: implemented for MVS_SAMPLER_FIXED (the library default)");
    if (ctx->W > 16383 || ctx->H > 16383) return fail(ctx, MVS_EINVAL, "mvs_sweep_handles: images of up to 16383 x 16383");
    if (main_slot < 0 || main_slot >= ctx->store_cap || !ctx->store_have[main_slot]) return fail(ctx, MVS_EINVAL, "mvs_sweep_handles: main slot %d holds no frame", main_slot);
    for (int v = 0; v < nside; v++)
        if (side_slots[v] < 0 || side_slots[v] >= ctx->store_cap || !ctx->store_have[side_slots[v]])
            return fail(ctx, MVS_EINVAL, "mvs_sweep_handles: side view %d: slot %d holds no frame", v, side_slots[v]);
    MVS_HIP(ctx, hipSetDevice(ctx->device));
    const int W = ctx->W, H = ctx->H;
    int rc;
    // the context's sweep inputs become references into the store (whatever fails below leaves it without views)
    ctx->have_views = false;
    ctx->plan_valid = false;
    ctx->quads16_valid = false;
    ctx->pads_valid = false;
    struct DropOnError {
        mvs_ctx *c;
        bool armed = true;
        ~DropOnError()
        {
            if (!armed) return;
            (void)hipStreamSynchronize(c->stream);
            c->have_views = false;
            c->views_in_store = false;
            c->plan_valid = false;
        }
    } drop{ctx};
    memcpy(ctx->main_cam, main_cam, sizeof(float) * 16);
    ctx->main_store_slot = main_slot;
    ctx->have_main = true;
    ctx->pad_pitch = ((W + 2 + 63) / 64) * 64;
    ctx->pad_slab = (size_t)ctx->pad_pitch * (H + 2);
    ctx->V = 0;
    ctx->q_host.assign((size_t)nside * 12, 0.f);
    ctx->view_slots_host.assign(side_slots, side_slots + nside);
    if (nside > 0) {
        if ((rc = ensure(ctx, ctx->qmats, sizeof(float) * 12 * nside))) return rc;
        if ((rc = ensure(ctx, ctx->view_slots, sizeof(int) * 256))) return rc;
        for (int v = 0; v < nside; v++) view_matrix(ctx->main_cam, side_cams + 16 * v, W, H, ctx->q_host.data() + 12 * v);
        MVS_HIP(ctx, hipMemcpyAsync(ctx->qmats.ptr, ctx->q_host.data(), sizeof(float) * 12 * nside, hipMemcpyHostToDevice, ctx->stream));
        MVS_HIP(ctx, hipMemcpyAsync(ctx->view_slots.ptr, ctx->view_slots_host.data(), sizeof(int) * nside, hipMemcpyHostToDevice, ctx->stream));
    }
    ctx->V = nside;
    ctx->views_in_store = nside > 0;
    ctx->have_views = true;
    if ((rc = sweep_set_planes_impl(ctx, nplanes, z_lo, z_hi, false))) return rc;
    if ((rc = mvs_sweep_run(ctx, 0, nside, MVS_SWEEP_FUSED_ARGMIN))) return rc;  // plans on demand (rectified or general kernel)
    if ((rc = mvs_sweep_fetch(ctx, depth_hw, cost_hw, nullptr, nullptr))) return rc;
    drop.armed = false;
    return MVS_OK;
}

int mvs_warp_by_depth(mvs_ctx *ctx, const float main_cam[16], const float *depth_hw, const float side_cam[16],
                      const uint8_t *frame_hw, uint8_t *out_hw2)
{
    if (!ctx || !main_cam || !depth_hw || !side_cam || !frame_hw || !out_hw2) return fail(ctx, MVS_EINVAL, "mvs_warp_by_depth: null argument");
    MVS_HIP(ctx, hipSetDevice(ctx->device));
    const int W = ctx->W, H = ctx->H;
    const size_t P = (size_t)W * H;
    const int pitch = ((W + 2 + 63) / 64) * 64;
    int rc;
    if ((rc = ensure(ctx, ctx->upload, P))) return rc;
    if ((rc = ensure(ctx, ctx->r_frame, (size_t)pitch * (H + 2) + 64))) return rc;
    if ((rc = ensure(ctx, ctx->r_zbuf, P * sizeof(float)))) return rc;
    if ((rc = ensure(ctx, ctx->r_out3, 3 * P))) return rc;
    if ((rc = ensure(ctx, ctx->r_tmp0, P * sizeof(int) > 64 ? P * sizeof(int) : 64))) return rc;
    float Q[12];
    view_matrix(main_cam, side_cam, W, H, Q);
    MVS_HIP(ctx, hipMemcpyAsync(ctx->r_tmp0.ptr, Q, sizeof(Q), hipMemcpyHostToDevice, ctx->stream));
    MVS_HIP(ctx, hipMemcpyAsync(ctx->upload.ptr, frame_hw, P, hipMemcpyHostToDevice, ctx->stream));
    MVS_HIP(ctx, hipMemcpyAsync(ctx->r_zbuf.ptr, depth_hw, P * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    pad_wrap_kernel<<<dim3(div_up(pitch, 256), H + 2), 256, 0, ctx->stream>>>((const uint8_t *)ctx->upload.ptr,
                                                                              (uint8_t *)ctx->r_frame.ptr, W, H, pitch);
    if (ctx->sampler == MVS_SAMPLER_FIXED) {
        if ((rc = warp_by_depth_fx_launch(ctx, (const float *)ctx->r_zbuf.ptr, (const float *)ctx->r_tmp0.ptr, (const uint8_t *)ctx->r_frame.ptr, pitch,
                                          (uint8_t *)ctx->r_out3.ptr)))
            return rc;
    } else {
        warp_by_depth_kernel<<<dim3(div_up(W, 64), div_up(H, 4)), 256, 0, ctx->stream>>>(
            (const float *)ctx->r_zbuf.ptr, (const float *)ctx->r_tmp0.ptr, (const uint8_t *)ctx->r_frame.ptr, pitch, W, H,
            1.0f / (float)W, 1.0f / (float)H, (uint8_t *)ctx->r_out3.ptr);
    }
    MVS_HIP(ctx, hipGetLastError());
    MVS_HIP(ctx, hipMemcpyAsync(out_hw2, ctx->r_out3.ptr, 2 * P, hipMemcpyDeviceToHost, ctx->stream));
    MVS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return MVS_OK;
}

// test hook (not in mvs.h): exhaustive check of the Newton reciprocal for one exponent
int mvs_test_rcp(mvs_ctx *ctx, unsigned exp_bits, unsigned long long *mismatch, unsigned long long *allones)
{
    if (!ctx || !mismatch || !allones) return MVS_EINVAL;
    MVS_HIP(ctx, hipSetDevice(ctx->device));
    unsigned long long *d = nullptr;
    MVS_HIP(ctx, hipMalloc(&d, 16));
    MVS_HIP(ctx, hipMemsetAsync(d, 0, 16, ctx->stream));
    rcp_check_kernel<<<(1u << 23) / 256, 256, 0, ctx->stream>>>(exp_bits, d);
    unsigned long long h[2] = {0, 0};
    MVS_HIP(ctx, hipMemcpyAsync(h, d, 16, hipMemcpyDeviceToHost, ctx->stream));
    MVS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    MVS_HIP(ctx, hipFree(d));
    *mismatch = h[0];
    *allones = h[1];
    return MVS_OK;
}

}  // extern "C"
