// sweep_fx.hip -- plane-sweep cost volume + depth selection with the FIXED-POINT sampler (arithmetic contract v2) for gfx950.
//
// Same warp as sweep.hip (the D-plane evaluation of shader.frag:11-25); the texture unit is modelled
// the way fixed-function samplers and OpenCV's own fixed-point remap (INTER_BITS = 5: the path the reference's flowRemap takes
// through cv::remap, util.cpp:401) work: positions quantised to 1/32 texel, bilinear weights from a 32 x 32 table of 8-bit
// integers that sum to 255.  Identical in oracle/sweep_oracle.c (orc_sweep_fx); DESIGN.md section 2b:
//   s, r   as in contract v1:  s = fma(z, B, A),  r = RN(1 / s.w)  (v_rcp_f32 + one FMA Newton step)
//   u      = RNE(s.xy * (256 r) + 4)        one rounding of the real product, done by an FMA whose addend is 1.5 * 2^23 + 4
//   in frame <=> s.w > 0 and 132 < ux < 256 W + 132 and 132 < uy < 256 H + 132
//   i, k   = u >> 8, (u >> 3) & 31
//   dot    = table[ky][kx] . (t00, t01, t10, t11)                                  v_dot4_u32_u8, in [0, 65025]
//   cell  += (1 << 24) + |dot - 255 I_main|                                         v_sad_u16; integer, order independent
// Why: the exact-f32 sampler costs ~50 VALU cycles per sample on gfx950 (4 conversions, 2 fract, 2 address ops, 2 v_fma_mix, cvt,
// v_sad_u32 at 4 cycles each on top of the projection); here a sample is 4 FMAs + v_perm_b32 + 2 shifts + 2 ands + v_dot4 +
// v_sad_u16 = ~28 cycles, and the two LDS reads (weights, texel quad) hide behind it at 4 waves per SIMD
// (tools/sweep_v2_probe.hip, profiles/r02/probe_*.txt).
//
// LDS image of a workgroup (32 KiB): 32 rows of 1 KiB.  Row r = [ weight table row ky = r : 32 dwords | 224 texel quads: row r of one region in
// columns 0..111 and of the next view's region in columns 112..223, or of one wide region ].
// A quad is the four u8 texels (t00, t01, t10, t11) of one bilinear footprint, so the fetch is one ds_read_b32.  With
// P = v_perm_b32(Ty, Tx) = [iy : ix : fy8 : fx8] both addresses are a shift and a mask away:
//   weights  (P >> 1) & 0x7c7c = ky << 10 | kx << 2          quad  ((P >> 14) & 0x7ffc) + 128 = iy << 10 | ix << 2 (+ 128)
#include "sweep_shared.hpp"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace mvs {

#ifndef MVS_FX_ROWS
#define MVS_FX_ROWS 32
#endif
#ifndef MVS_FX_OCC
#define MVS_FX_OCC 4
#endif
constexpr int FX_ROWS = MVS_FX_ROWS;       // region rows (and weight-table rows) per LDS image (timing experiments: -DMVS_FX_ROWS=24 -DMVS_FX_OCC=5, wrong weights for ky >= 24)
constexpr int FX_ROW_DW = 256;    // dwords per LDS row
constexpr int FX_LUT_DW = 32;     // of which the first 32 hold the weight-table row
constexpr int FX_MAX_RW = FX_ROW_DW - FX_LUT_DW;  // 224 quads
constexpr int FX_HALF_COL = FX_MAX_RW / 2;          // two regions of up to 112 quads side by side (region look-ahead)
constexpr float FX_MAGIC = 12582912.0f;            // 1.5 * 2^23: floats in [2^23, 2^24) have ulp 1
constexpr int FX_TILE_H = 8, FX_PC = 16, FX_NPX = 2;  // 64x8-pixel tiles, 16 planes per chunk: 32 accumulators per thread
constexpr int FX_GS = 2;          // samples per software-pipeline group (4: 12 more VGPRs, spills, 1.505 vs 1.479 ms at c3)
constexpr int FX_VB = 64;         // views per batch of LDS-resident per-view constants
constexpr int FX_WG_PER_CU = MVS_FX_OCC;   // launch bound (waves per SIMD): 128 VGPRs, 39.5 KiB of LDS (5 would need <= 96 VGPRs: the kernel needs ~125)

enum FxMode : unsigned { FX_SKIP = 0, FX_FAST = 1, FX_BORDER = 2, FX_GENERIC = 3 };

typedef float f32x2 __attribute__((ext_vector_type(2)));

// acc + |a.lo16 - b.lo16| + |a.hi16 - b.hi16| (v_sad_u16).  Through the builtin, NOT inline asm: the first operand comes straight
// from v_dot4_u32_u8, and on gfx950 a VALU instruction that overwrites a dot instruction's destination needs wait states that
// the compiler inserts only for instructions it can see -- with inline asm the sums came out wrong (and differently per run).
__device__ __forceinline__ uint32_t sad_u16(uint32_t a, uint32_t b, uint32_t acc) { return __builtin_amdgcn_sad_u16(a, b, acc); }

// one sample with every check, taps and weights from global memory (planner mode GENERIC, the un-tiled kernel, warp_by_depth)
// WSCALED: the caller's w row (A.aw, bw) is the view matrix's divided by 256, so RN(1 / s.w) already is 256 r -- bit for bit, a
// power of two commutes with every rounding involved (the tiled kernel keeps its view matrices in that form)
template <bool WSCALED = false>
__device__ __forceinline__ uint32_t sample_global_fx(const Affine &A, float bx, float by, float bw, float z, const uint8_t *__restrict__ pad,
                                                     int pitch, float hix, float hiy, const uint32_t *__restrict__ lut, uint32_t Im255,
                                                     uint32_t *dot_out = nullptr)
{
    const float sx = __builtin_fmaf(z, bx, A.ax);
    const float sy = __builtin_fmaf(z, by, A.ay);
    const float sw = __builtin_fmaf(z, bw, A.aw);
    if (!(sw > 0.0f)) return 0u;
    const float r256 = WSCALED ? rcp_rn(sw) : rcp_rn(sw) * 256.0f;
    const float tx = __builtin_fmaf(sx, r256, FX_MAGIC + 4.0f), ty = __builtin_fmaf(sy, r256, FX_MAGIC + 4.0f);
    if (!(tx > FX_MAGIC + 132.0f && tx < hix && ty > FX_MAGIC + 132.0f && ty < hiy)) return 0u;
    const uint32_t ux = __builtin_bit_cast(uint32_t, tx) & 0x3fffffu, uy = __builtin_bit_cast(uint32_t, ty) & 0x3fffffu;  // tx - magic
    const uint8_t *q = pad + (size_t)(uy >> 8) * pitch + (ux >> 8);
    const uint32_t quad = (uint32_t)q[0] | ((uint32_t)q[1] << 8) | ((uint32_t)q[pitch] << 16) | ((uint32_t)q[pitch + 1] << 24);
    const uint32_t w = lut[(((uy >> 3) & 31u) << 5) | ((ux >> 3) & 31u)];
    const uint32_t dot = __builtin_amdgcn_udot4(quad, w, 0u, false);
    if (dot_out) *dot_out = dot;
    return sad_u16(dot, Im255, 1u << 24);
}

// The same sample for the tiled kernel's GENERIC regions (warped footprint larger than the LDS image): the texel quad is ONE dword
// of the view's quad image (instead of four byte gathers from the padded frame) and the weights come from the table rows the LDS
// image carries anyway (instead of a gather from global memory); no divergent branch -- a sample outside the frame reads quad 0 and
// is dropped by the select.  The w row is pre-divided by 256 (see the kernel).
__device__ __forceinline__ uint32_t sample_quads_fx(const Affine &A, float bx, float by, float bw, float z, const uint32_t *__restrict__ quads, int pitch,
                                                   float hix, float hiy, const uint32_t *__restrict__ lds, uint32_t Im255)
{
    const float sx = __builtin_fmaf(z, bx, A.ax), sy = __builtin_fmaf(z, by, A.ay), sw = __builtin_fmaf(z, bw, A.aw);
    const float r256 = rcp_rn(sw);
    const float tx = __builtin_fmaf(sx, r256, FX_MAGIC + 4.0f), ty = __builtin_fmaf(sy, r256, FX_MAGIC + 4.0f);
    const bool ok = sw > 0.0f && tx > FX_MAGIC + 132.0f && tx < hix && ty > FX_MAGIC + 132.0f && ty < hiy;
    const uint32_t ux = __builtin_bit_cast(uint32_t, tx) & 0x3fffffu, uy = __builtin_bit_cast(uint32_t, ty) & 0x3fffffu;  // t - magic
    const uint32_t quad = quads[ok ? (uy >> 8) * (uint32_t)pitch + (ux >> 8) : 0u];
    const uint32_t w = lds[(((uy >> 3) & 31u) << 8) | ((ux >> 3) & 31u)];  // table row ky is the first 32 dwords of LDS row ky
    const uint32_t cell = sad_u16(__builtin_amdgcn_udot4(quad, w, 0u, false), Im255, 1u << 24);
    return ok ? cell : 0u;
}

// ------------------------------------------------------------------------------------------------------
// un-tiled kernel: one pixel per thread, global gathers (MVS_SWEEP_FORCE_GENERIC, V == 0)
// ------------------------------------------------------------------------------------------------------
template <bool WRITE_VOLUME, bool FUSED>
__global__ __launch_bounds__(256) void sweep_fx_generic(SweepParams p, const uint32_t *__restrict__ lut)
{
    const int col = blockIdx.x * 64 + (threadIdx.x & 63);
    const int row = p.row_begin + blockIdx.y * 4 + (threadIdx.x >> 6);
    if (col >= p.W || row >= p.row_end) return;
    const size_t P = (size_t)p.W * p.H;
    const size_t pix = (size_t)row * p.W + col;
    const float xn = __builtin_fmaf((float)(2 * col + 1), p.invW, -1.0f);
    const float yn = __builtin_fmaf(-(float)(2 * row + 1), p.invH, 1.0f);
    const uint32_t Im255 = 255u * p.main_img[pix];
    const float hix = FX_MAGIC + 132.0f + 256.0f * (float)p.W, hiy = FX_MAGIC + 132.0f + 256.0f * (float)p.H;
    uint32_t best = 0;
    int bi = -1;
    for (int d0 = p.plane_begin; d0 < p.plane_end; d0 += PCG) {
        uint32_t acc[PCG];
#pragma unroll
        for (int k = 0; k < PCG; k++) acc[k] = 0u;
        for (int v = p.v0; v < p.v0 + p.vcount; v++) {
            const float *q = p.Q + 12 * v;
            const Affine A = view_affine(q, xn, yn);
            const uint8_t *pad = p.pads + p.pad_slab * (p.view_slot ? p.view_slot[v] : v);
#pragma unroll
            for (int k = 0; k < PCG; k++)
                if (d0 + k < p.plane_end) acc[k] += sample_global_fx(A, q[2], q[6], q[10], p.z[d0 + k], pad, p.pitch, hix, hiy, lut, Im255);
        }
#pragma unroll
        for (int k = 0; k < PCG; k++)
            if (d0 + k < p.plane_end) {
                if (WRITE_VOLUME) p.volume[(size_t)(d0 + k) * P + pix] = acc[k];
                if (FUSED) argmin_update_packed<CS_FIXED>(acc[k], d0 + k, best, bi);
            }
    }
    if (FUSED) store_best<CS_FIXED>(p, pix, best & 0xffffffu, best >> 24, bi);
}

// ------------------------------------------------------------------------------------------------------
// region planner: where does tile t land in side view v over the 16 planes of chunk c?
// ------------------------------------------------------------------------------------------------------
// A workgroup's parameter block out of an array of them, through the constant address space: scalar loads into SGPRs (a per-thread
// copy of the 280-byte block from global memory lands in scratch: the planner of 8 frames took 5 ms that way).
__device__ __forceinline__ SweepParams load_params(const SweepParams *src)
{
    static_assert(sizeof(SweepParams) % 4 == 0, "copied dword by dword");
    SweepParams p;
    const __attribute__((address_space(4))) uint32_t *s = (const __attribute__((address_space(4))) uint32_t *)(uintptr_t)src;
    uint32_t *d = (uint32_t *)&p;
#pragma unroll
    for (size_t i = 0; i < sizeof(SweepParams) / 4; i++) d[i] = s[i];
    return p;
}

// (bits 19-31 of a descriptor's second word: the slab of the view's quad image -- the view's own index, or its slot in the frame
// store when the views are given by slots, mvs_sweep_batch)
__device__ __forceinline__ void plan_regions_fx_body(const SweepParams &p, uint2 *__restrict__ plan)
{
    const int total = p.tiles_x * p.tiles_y * p.nchunks * p.V;
    const bool live = (int)(blockIdx.x * blockDim.x + threadIdx.x) < total;  // (no early return: the counters below are reduced per wavefront)
    const int tid = min((int)(blockIdx.x * blockDim.x + threadIdx.x), total - 1);
    const int v = tid % p.V;
    const int rest = tid / p.V;
    const int chunk = rest % p.nchunks;
    const int tile = rest / p.nchunks;
    const int tx = tile % p.tiles_x, ty = tile / p.tiles_x;
    const int c0 = tx * TILE_W, c1 = min(c0 + TILE_W, p.W) - 1;
    const int r0 = ty * FX_TILE_H, r1 = min(r0 + FX_TILE_H, p.H) - 1;
    const int d0 = chunk * FX_PC, d1 = min(d0 + FX_PC, p.D) - 1;
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
    int x0 = 0, y0 = 0, rw = 0, rh = 0;
    int too_large = 0, staged_w = 0, staged_h = 0;
    // the sampled position is (256 c + 4 +- 1/2) / 256 <= c + 0.018: covered by the margin like the f32 rounding of c itself
    const float m = PLAN_MARGIN;
    if (behind || !(xmin == xmin) || !(ymin == ymin) || !(xmax < 1.0e9f) || !(ymax < 1.0e9f) || !(xmin > -1.0e9f) || !(ymin > -1.0e9f)) {
        mode = FX_GENERIC;  // w is affine in (col, row, z): positive at the eight corners <=> positive for every sample of the box
    } else if (xmax < 0.5f - m || xmin > p.Wp + m || ymax < 0.5f - m || ymin > p.Hp + m) {
        mode = FX_SKIP;  // the hull of the warped box is out of frame (projective maps keep segments: DESIGN.md)
    } else {
        x0 = max(0, (int)floorf(xmin - m)) & ~3;
        const int x1 = min(p.W, (int)floorf(xmax + m));
        y0 = max(0, (int)floorf(ymin - m));
        const int y1 = min(p.H, (int)floorf(ymax + m));
        rw = ((x1 - x0 + 1) + 3) & ~3;
        rh = y1 - y0 + 1;
        if (rw > FX_MAX_RW || rw <= 0 || rh <= 0 || rh > FX_ROWS) {
            mode = FX_GENERIC;
            too_large = 1;
        } else {
            mode = (xmin > 0.5f + m && xmax < p.Wp - m && ymin > 0.5f + m && ymax < p.Hp - m) ? FX_FAST : FX_BORDER;
            staged_w = rw;  // largest staged region (diagnostic)
            staged_h = rh;
        }
    }
    uint2 d;
    d.x = (unsigned)x0 | ((unsigned)y0 << 16);
    d.y = (unsigned)rw | ((unsigned)rh << 8) | (mode << 16) | ((unsigned)(p.view_slot ? p.view_slot[v] : v) << 19);
    if (live) plan[tid] = d;
    if (p.plan_stats) {  // one atomic per wavefront and counter, not one per thread: they all hit the same four addresses
        const int n_large = wave_sum_i32(live ? too_large : 0), n_regions = wave_sum_i32(live && mode != FX_SKIP ? 1 : 0);
        const int w_max = wave_max_i32(live ? staged_w : 0), h_max = wave_max_i32(live ? staged_h : 0);
        if ((threadIdx.x & 63) == 0) {
            if (n_large) atomicAdd(p.plan_stats, n_large);
            if (n_regions) atomicAdd(p.plan_stats + 1, n_regions);
            if (w_max) atomicMax(p.plan_stats + 2, w_max);
            if (h_max) atomicMax(p.plan_stats + 3, h_max);
        }
    }
}

__global__ __launch_bounds__(256) void plan_regions_fx(SweepParams p, uint2 *__restrict__ plan) { plan_regions_fx_body(p, plan); }

// one main frame per blockIdx.y, each with its own parameter block (mvs_sweep_batch)
__global__ __launch_bounds__(256) void plan_regions_fx_batch(const SweepParams *__restrict__ pp)
{
    // the block's parameters through the constant address space: scalar loads into SGPRs (a per-thread copy of the 280-byte block from
    // global memory lands in scratch: the planner of 8 frames took 5 ms that way)
    const SweepParams p = load_params(pp + blockIdx.y);
    plan_regions_fx_body(p, (uint2 *)p.plan);
}

// ------------------------------------------------------------------------------------------------------
// tiled kernel
// ------------------------------------------------------------------------------------------------------
// Stage quads [x0, x0+rw) x [y0, y0+rh) of a view's quad image: one global_load_lds_dwordx4 per region row (each active lane
// copies 4 quads = 16 bytes straight into LDS; the destination is the row's base + lane * 16, hence no register, no shuffle).
// Rows are dealt to the four wavefronts; completion is awaited by the caller's __syncthreads() (it drains vmcnt).
// `col` = first quad column of the LDS rows the region goes to: a row holds 224 quads and a typical region is 70-90 wide, so two
// regions sit side by side (columns 0 and FX_HALF_COL) and the NEXT view's region is copied while the current one is sampled.
__device__ __forceinline__ void stage_region_fx(const uint32_t *__restrict__ quads, int pitch, int x0, int y0, int rw, int rh, int col,
                                                uint32_t *__restrict__ lds)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int units = rw >> 2;
    // buffer form of the LDS copy: the view's quad image as a resource (SGPRs), the row's byte offset wave-uniform, 16 bytes per lane --
    // through a plain pointer the compiler forms a 64-bit per-lane address for every row (2 VGPRs + 2 VALU instructions per copy)
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)quads, 0, (int)0xffffffffu, 0x00020000);  // a slab is below 2^32 bytes (images up to 16383^2)
    const uint32_t row0 = 4u * (uint32_t)(y0 * pitch + x0), rstep = 4u * (uint32_t)pitch;
    const uint32_t lane16 = 16u * (uint32_t)lane;
    if (lane < units) {  // one exec-mask change around the whole loop, not one per row
        for (int ry = wave; ry < rh; ry += 4)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void *)(lds + ry * FX_ROW_DW + FX_LUT_DW + col), 16, lane16, row0 + rstep * (uint32_t)ry, 0, 0);
    }
}

// All planes [K0, K0 + KN) of one (pixel, view) when every sample is known to be in frame, software-pipelined by hand: the two
// LDS reads of group g + 1 are issued before the dot4 / sad of group g (inline asm reads; the wait is tied to the loaded
// registers and to the next group's addresses so that neither consumers nor the next address stage can cross it).
// WCONST: the view's w row does not depend on the plane (b.w == 0), so r is one number per (pixel, view).
template <int K0, int KN, bool WCONST>
__device__ __forceinline__ void sample_range_fx(const Affine &A, float bx, float by, float bw, float r256c, const float (&zc)[FX_PC], float offx,
                                                float offy, uint32_t lds_base, uint32_t Im255, uint32_t (&acc)[FX_PC])
{
    static_assert(KN % FX_GS == 0, "plane range must be a multiple of the group size");
#ifdef MVS_FX_NO_ASM
    // Build-time fallback without inline asm (make CXXFLAGS+=-DMVS_FX_NO_ASM): the same arithmetic with compiler-managed LDS reads and
    // waits, bit-identical and slower.  The pipelined form below issues ds_read_b32 from inline asm and places its own s_waitcnt, which
    // the compiler's wait-count insertion does not model: should a future hipcc copy or spill the loaded registers between issue and
    // wait, this is the path to ship until the asm is revisited (the GPU parity suite is the gate either way).
#pragma unroll
    for (int k = 0; k < KN; k++) {
        const float z = zc[K0 + k];
        const float sx = __builtin_fmaf(z, bx, A.ax), sy = __builtin_fmaf(z, by, A.ay);
        const float r256 = WCONST ? r256c : rcp_rn(__builtin_fmaf(z, bw, A.aw));
        const float Tx = __builtin_fmaf(sx, r256, offx), Ty = __builtin_fmaf(sy, r256, offy);
        const uint32_t P = __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, Ty), __builtin_bit_cast(uint32_t, Tx), 0x05010400u);
        typedef const __attribute__((address_space(3))) uint32_t *lds_ptr;  // lds_base is an LDS byte address, not a flat one
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wint-to-pointer-cast"  // the host pass sees a 64-bit pointer type here; LDS pointers are 32 bits wide
        const uint32_t w = *(lds_ptr)(lds_base + ((P >> 1) & 0x7c7cu));
        const uint32_t quad = *(lds_ptr)(lds_base + ((P >> 14) & 0x7ffcu) + 4u * FX_LUT_DW);
#pragma clang diagnostic pop
        acc[K0 + k] = sad_u16(__builtin_amdgcn_udot4(quad, w, 0u, false), Im255, acc[K0 + k]);
    }
    return;
#endif
    uint32_t la[FX_GS], ta[FX_GS];      // byte addresses of the group whose reads are issued next
    uint32_t lw[2][FX_GS], lq[2][FX_GS];  // weights and texel quads, double-buffered: group g is consumed while g + 1 is in flight
    auto address_stage = [&](int g) {
#pragma unroll
        for (int i = 0; i < FX_GS; i += 2) {
            const f32x2 z = {zc[K0 + g * FX_GS + i], zc[K0 + g * FX_GS + i + 1]};
            const f32x2 sx = __builtin_elementwise_fma(z, (f32x2)(bx), (f32x2)(A.ax));
            const f32x2 sy = __builtin_elementwise_fma(z, (f32x2)(by), (f32x2)(A.ay));
            f32x2 r256;
            if (WCONST) {
                r256 = (f32x2)(r256c);
            } else {
                const f32x2 sw = __builtin_elementwise_fma(z, (f32x2)(bw), (f32x2)(A.aw));
                const f32x2 r0 = {__builtin_amdgcn_rcpf(sw.x), __builtin_amdgcn_rcpf(sw.y)};
                const f32x2 e = __builtin_elementwise_fma(-sw, r0, (f32x2)(1.0f));
                r256 = __builtin_elementwise_fma(e, r0, r0);  // the w row is pre-divided by 256 (see the kernel)
            }
            const f32x2 Tx = __builtin_elementwise_fma(sx, r256, (f32x2)(offx));
            const f32x2 Ty = __builtin_elementwise_fma(sy, r256, (f32x2)(offy));
            // through float temporaries: __builtin_bit_cast applied directly to `vec.y` reads element 0 with this hipcc (ROCm 7.2)
            const float tx0 = Tx.x, tx1 = Tx.y, ty0 = Ty.x, ty1 = Ty.y;
            const uint32_t P0 = __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, ty0), __builtin_bit_cast(uint32_t, tx0), 0x05010400u);
            const uint32_t P1 = __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, ty1), __builtin_bit_cast(uint32_t, tx1), 0x05010400u);
            la[i] = ((P0 >> 1) & 0x7c7cu) + lds_base;
            ta[i] = ((P0 >> 14) & 0x7ffcu) + lds_base;
            la[i + 1] = ((P1 >> 1) & 0x7c7cu) + lds_base;
            ta[i + 1] = ((P1 >> 14) & 0x7ffcu) + lds_base;
        }
    };
    auto issue_reads = [&](int buf) {
#pragma unroll
        for (int i = 0; i < FX_GS; i++) {
            asm volatile("ds_read_b32 %0, %1" : "=v"(lw[buf][i]) : "v"(la[i]));
            asm volatile("ds_read_b32 %0, %1 offset:128" : "=v"(lq[buf][i]) : "v"(ta[i]));
        }
    };
    address_stage(0);
    issue_reads(0);
#pragma unroll
    for (int g = 0; g < KN / FX_GS; g++) {
        const int buf = g & 1;
        if (g + 1 < KN / FX_GS) address_stage(g + 1);
        // the loaded registers and the next group's addresses are operands: neither the consumers nor that address stage can cross the wait
        static_assert(FX_GS == 4 || FX_GS == 2, "the operand list of the wait below names every register of a group");
        if constexpr (FX_GS == 4)
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(lw[buf][0]), "+v"(lw[buf][1]), "+v"(lw[buf][2]), "+v"(lw[buf][3]), "+v"(lq[buf][0]), "+v"(lq[buf][1]), "+v"(lq[buf][2]),
                           "+v"(lq[buf][3]), "+v"(la[0]), "+v"(ta[0]), "+v"(la[FX_GS - 1]), "+v"(ta[FX_GS - 1]));
        else
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(lw[buf][0]), "+v"(lw[buf][1]), "+v"(lq[buf][0]), "+v"(lq[buf][1]), "+v"(la[0]), "+v"(ta[0]), "+v"(la[1]), "+v"(ta[1]));
        if (g + 1 < KN / FX_GS) issue_reads(buf ^ 1);
#pragma unroll
        for (int i = 0; i < FX_GS; i++) {
            const uint32_t dot = __builtin_amdgcn_udot4(lq[buf][i], lw[buf][i], 0u, false);
            acc[K0 + g * FX_GS + i] = sad_u16(dot, Im255, acc[K0 + g * FX_GS + i]);
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// the separable path (round 5): views with the main camera's orientation
// ------------------------------------------------------------------------------------------------------
// A side camera that is a pure TRANSLATION of the main one (any direction: along the optical axis too) has q1 = q4 = q8 = q9 = 0, and the
// contract's own expressions then separate: s.w = fma(z, q10, q11) depends on the plane only, Tx = fma(fma(z, q2, fma(q0, xn, q3)), 256 r, .) on
// (column, plane), Ty on (row, plane) -- the same floating-point operations on the same operands as the general form (fma(0, t, u) = u).
// So RN(1 / s.w) is a table entry per (view, plane), the row part of both LDS addresses -- ky << 10 for the weight word, iy << 10 for the
// quad -- a table entry per (view, image row, plane), both read through scalar loads, and a thread's column part is shared by its rows:
// a sample is two address adds + two ds_read_b32 + v_dot4 + v_sad_u16, no reciprocal, no projection (general form: 14 vector
// instructions per sample, 11.5 of them the projection).  (sweep_fx_rect needs q10 = 0 and unit scale on top: every pixel at the same
// sub-texel phase.  Here the phase varies across the tile; the weight word stays a per-sample LDS read.)  Regions of mode FAST only;
// BORDER and GENERIC regions take the general form.  Bit-identical by construction; tests/test_sweep_gpu.py compares with the hook
// MVS_NO_SEP=1 and with the oracle.
__device__ __forceinline__ bool sep_view(const float *q) { return q[1] == 0.0f && q[4] == 0.0f && q[8] == 0.0f && q[9] == 0.0f; }

__global__ __launch_bounds__(256) void plan_sep_tables(SweepParams p, float *__restrict__ rtab, uint2 *__restrict__ ytab, int dpad)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (size_t)p.V * p.H * dpad) return;
    const int d = (int)(t % dpad), row = (int)((t / dpad) % p.H), v = (int)(t / ((size_t)dpad * p.H));
    const float *q = p.Q + 12 * v;
    const float z = p.z[min(d, p.D - 1)];
    // as the tiled kernel forms them: the w row divided by 256 (exact), A.aw = fma(q8', xn, fma(q9', yn, q11')) = q11' for q8 = q9 = 0
    const float r256 = rcp_rn(__builtin_fmaf(z, q[10] * 0.00390625f, q[11] * 0.00390625f));
    const float yn = __builtin_fmaf(-(float)(2 * row + 1), p.invH, 1.0f);
    const float ay = __builtin_fmaf(q[5], yn, q[7]);  // = fma(q4, xn, fma(q5, yn, q7)) for q4 = 0
    const float Ty = __builtin_fmaf(__builtin_fmaf(z, q[6], ay), r256, FX_MAGIC + 4.0f);
    const uint32_t b = __builtin_bit_cast(uint32_t, Ty);
    ytab[t] = make_uint2(((b >> 3) & 31u) << 10, ((b >> 8) & 0x3fffu) << 10);
    if (row == 0) rtab[(size_t)v * dpad + d] = r256;
}

// all 16 planes of one view for the two rows of a thread (a FAST region: every sample in frame).  ax / bx: the x row of the view at this
// column; zc: the chunk's planes; rt / yt[j]: the view's tables at the chunk's first plane (and at row j of the thread; a row below the
// image reads the last row's entries and its sums are never used); offx: as in sample_range_fx; xq0 = LDS address of quad column 0 of
// region row 0 minus (y0 << 10).  The table entries of four planes (20 SGPRs: three scalar loads) are requested one quarter ahead of
// their use; groups of two planes x two rows, software-pipelined over the whole chunk as in sample_range_fx (the 8 LDS reads of group
// g + 1 are issued before group g is consumed).
__device__ __forceinline__ void sample_view_sep(float ax, float bx, const float (&zc)[FX_PC], const float *__restrict__ rt, const uint2 *const (&yt)[FX_NPX], float offx,
                                                uint32_t lds_base, uint32_t xq0, const uint32_t (&Im255)[FX_NPX], uint32_t (&acc)[FX_NPX][FX_PC])
{
    static_assert(FX_NPX == 2 && FX_PC == 16, "written for two rows per thread and chunks of 16 planes");
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    typedef uint32_t u32x8 __attribute__((ext_vector_type(8)));
    typedef const __attribute__((address_space(4))) u32x4 *c4;
    typedef const __attribute__((address_space(4))) u32x8 *c8;
    constexpr int GS = 2, QP = 4, NS = GS * FX_NPX, NG = FX_PC / GS;  // planes per group, planes per table load, samples per group, groups
    u32x4 rq[2];
    u32x8 yq0[2], yq1[2];
    auto load_tables = [&](int quarter) {
        rq[quarter & 1] = *(c4)(uintptr_t)(rt + QP * quarter);
        yq0[quarter & 1] = *(c8)(uintptr_t)(yt[0] + QP * quarter);
        yq1[quarter & 1] = *(c8)(uintptr_t)(yt[1] + QP * quarter);
    };
    uint32_t la[NS], ta[NS], lw[2][NS], lq[2][NS];
    auto address_stage = [&](int g) {
        const int slot = (g * GS / QP) & 1;
#pragma unroll
        for (int i = 0; i < GS; i++) {
            const int k = g * GS + i, kk = k % QP;
            const float sx = __builtin_fmaf(zc[k], bx, ax);
            const uint32_t rbits = rq[slot][kk];  // (through a temporary: __builtin_bit_cast applied directly to a vector element reads element 0 with this hipcc)
            const float Tx = __builtin_fmaf(sx, __builtin_bit_cast(float, rbits), offx);
            const uint32_t b = __builtin_bit_cast(uint32_t, Tx);
            const uint32_t xw = ((b >> 1) & 0x7cu) + lds_base;
            const uint32_t xq = ((b >> 6) & 0x3fcu) + xq0;
            la[2 * i] = xw + yq0[slot][2 * kk];
            ta[2 * i] = xq + yq0[slot][2 * kk + 1];
            la[2 * i + 1] = xw + yq1[slot][2 * kk];
            ta[2 * i + 1] = xq + yq1[slot][2 * kk + 1];
        }
    };
    auto issue_reads = [&](int buf) {
#pragma unroll
        for (int n = 0; n < NS; n++) {
            asm volatile("ds_read_b32 %0, %1" : "=v"(lw[buf][n]) : "v"(la[n]));
            asm volatile("ds_read_b32 %0, %1" : "=v"(lq[buf][n]) : "v"(ta[n]));
        }
    };
    load_tables(0);
    address_stage(0);
    issue_reads(0);
#pragma unroll
    for (int g = 0; g < NG; g++) {
        const int buf = g & 1;
        if ((g * GS) % QP == 0 && g * GS / QP + 1 < FX_PC / QP) load_tables(g * GS / QP + 1);  // the next quarter's entries: in flight during this quarter's first group
        if (g + 1 < NG) address_stage(g + 1);
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(lw[buf][0]), "+v"(lw[buf][1]), "+v"(lw[buf][2]), "+v"(lw[buf][3]), "+v"(lq[buf][0]), "+v"(lq[buf][1]), "+v"(lq[buf][2]), "+v"(lq[buf][3]),
                       "+v"(la[0]), "+v"(ta[0]), "+v"(la[NS - 1]), "+v"(ta[NS - 1]));
        if (g + 1 < NG) issue_reads(buf ^ 1);
#pragma unroll
        for (int i = 0; i < GS; i++) {
            const int k = g * GS + i;
#pragma unroll
            for (int j = 0; j < FX_NPX; j++) {
                const uint32_t dot = __builtin_amdgcn_udot4(lq[buf][2 * i + j], lw[buf][2 * i + j], 0u, false);
                acc[j][k] = sad_u16(dot, Im255[j], acc[j][k]);
            }
        }
    }
}

// FX_MAGIC + k as a float, for an integer |k| < 2^22: in the binade [2^23, 2^24) one ulp is 1, so the bit pattern is 0x4B400000 + k.
// With a wave-uniform k this is scalar integer arithmetic (gfx950's SALU has no float unit; written with floats, every region
// constant below cost a v_cvt + v_mul + v_add per wavefront and region: ~20 VALU instructions, 5 % of the kernel).
__device__ __forceinline__ float magic_plus(int k) { return __builtin_bit_cast(float, 0x4B400000 + k); }

struct FxRegion {
    float offx, offy;            // magic + 4 - 256 * region origin: T - magic is the position relative to the region, in 1/256 texel
    float lox, hix, loy, hiy;    // in-frame test on T (strict)
};

// the same planes with the in-frame test per sample (region mode BORDER, for the wavefronts that straddle the frame edge).
// A sample outside the frame reads whatever its masked address holds (the masks keep every address inside the LDS image, so no
// clamp is needed) and is then dropped by the select.  (A variant with the hoisted reciprocal of the plane-independent-w case
// doubles the instantiations and made the register allocator spill 29 VGPRs in the whole kernel: not worth 1 % of the samples.)
template <int K0, int KN>
__device__ __forceinline__ void sample_range_fx_checked(const Affine &A, float bx, float by, float bw, const float (&zc)[FX_PC], const FxRegion &rg,
                                                        const uint32_t *__restrict__ lds, uint32_t Im255, uint32_t (&acc)[FX_PC])
{
#pragma unroll
    for (int k = K0; k < K0 + KN; k++) {
        const float z = zc[k];
        const float sx = __builtin_fmaf(z, bx, A.ax), sy = __builtin_fmaf(z, by, A.ay), sw = __builtin_fmaf(z, bw, A.aw);
        const float r256 = rcp_rn(sw);  // w row pre-divided by 256
        const float Tx = __builtin_fmaf(sx, r256, rg.offx), Ty = __builtin_fmaf(sy, r256, rg.offy);
        const bool ok = Tx > rg.lox && Tx < rg.hix && Ty > rg.loy && Ty < rg.hiy;
        const uint32_t P = __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, Ty), __builtin_bit_cast(uint32_t, Tx), 0x05010400u);
        const uint32_t w = lds[((P >> 1) & 0x7c7cu) >> 2];
        const uint32_t quad = lds[(((P >> 14) & 0x7ffcu) >> 2) + FX_LUT_DW];
        const uint32_t dot = __builtin_amdgcn_udot4(quad, w, 0u, false);
        const uint32_t t = sad_u16(dot, Im255, acc[k] + (1u << 24));
        acc[k] = ok ? t : acc[k];
    }
}

// 4 workgroups per CU (<= 128 VGPRs, 39.5 KiB of LDS each): at 2 per CU the LDS reads no longer hide behind the VALU work
// (19.0 vs 14.6 ns per wave-sample, tools/sweep_v2_probe.hip).
// (Rounds 2-5 carried timing experiments here -- a section profile with s_memtime, run-time cuts on p.debug, compile-time cuts -- behind
// -DMVS_FX_EXPERIMENTS / -DMVS_FX_CUT; what they measured is docs/experiments.md.  Round 6 removed them: the kernel is the product.)

template <bool WRITE_VOLUME, bool FUSED, bool SEP = false>  // SEP: with the separable path compiled in (its own instantiation: the path's registers would cost the others 3 %)
__device__ __forceinline__ void sweep_fx_tiled_body(const SweepParams &p, const uint32_t *__restrict__ lut_g)
{
    constexpr int NPX = FX_NPX, PC = FX_PC, TILE_H = FX_TILE_H;
    // one object, so the texel image sits at LDS address 0 and its byte offsets are the ds_read addresses
    __shared__ __attribute__((aligned(16))) uint32_t smem[FX_ROWS * FX_ROW_DW + FX_VB * 14 + 2 + (FUSED ? 2 * 256 * NPX : 0)];
    uint32_t *lds = smem;
    // per-view constants of a batch of up to FX_VB views: view matrix (12 floats) and this chunk's region descriptor (2 dwords).
    // Read from here, a view's constants cost an LDS round trip (~100 cycles) instead of a dependent global load (~1 us) per region.
    float *qtab = (float *)(smem + FX_ROWS * FX_ROW_DW);
    uint2 *dtab = (uint2 *)(smem + FX_ROWS * FX_ROW_DW + FX_VB * 12);
    uint2 *best_state = (uint2 *)(smem + FX_ROWS * FX_ROW_DW + FX_VB * 14 + 2);  // (packed best cell, best index) per (pixel j, thread)

    const int band_tile = (p.debug & 2) ? ((int)blockIdx.x < p.tiles_x * p.tyn ? (int)blockIdx.x : -1) : grouped_tile(blockIdx.x, p.tiles_x, p.tyn);
    if (band_tile < 0) return;
    const int tx = band_tile % p.tiles_x, ty = band_tile / p.tiles_x + p.ty0;
    const int tile = ty * p.tiles_x + tx;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = tx * TILE_W + lane;
    const int row0 = ty * TILE_H + wave * NPX;
    const bool col_ok = col < p.W;
    const size_t P = (size_t)p.W * p.H;
    const float xn = __builtin_fmaf((float)(2 * col + 1), p.invW, -1.0f);
    const uint32_t lds_base = (uint32_t)(uintptr_t)lds;

    // the weight table goes into the first 32 dwords of every row (visible after the first staging barrier)
    for (int i = threadIdx.x; i < FX_ROWS * FX_LUT_DW; i += 256) lds[(i >> 5) * FX_ROW_DW + (i & 31)] = lut_g[i];

    float yn[NPX];
    uint32_t Im255[NPX];
    bool ok[NPX];
#pragma unroll
    for (int j = 0; j < NPX; j++) {
        const int row = row0 + j;
        ok[j] = col_ok && row < p.H;
        yn[j] = __builtin_fmaf(-(float)(2 * row + 1), p.invH, 1.0f);
        Im255[j] = ok[j] ? 255u * (uint32_t)p.main_img[(size_t)row * p.W + col] : 0u;
        if (FUSED) best_state[j * 256 + threadIdx.x] = make_uint2(1u, 0xffffffffu);  // (sum 1, count 0), no plane; own slot only: no barrier needed
    }
    const float fhix = FX_MAGIC + 132.0f + 256.0f * (float)p.W, fhiy = FX_MAGIC + 132.0f + 256.0f * (float)p.H;

    const int chunk_first = p.chunk0 + (int)blockIdx.y * p.cps;
    const int chunk_last = min(p.chunk1, chunk_first + p.cps);
    // Region pipeline.  `qcol` = first quad column of the region being sampled; `ahead` = the region of the view at hand was requested
    // while the previous staged view was being sampled (its copy is in flight or has landed).  When the descriptors of ALL this
    // workgroup's (chunk, view) regions fit the table (`whole_wg`: c3 has 4 chunks x 16 views), they and the view constants are loaded
    // once, and the pipeline runs on across chunk boundaries: no table reload and no cold first region per chunk.
    int qcol = 0;
    bool ahead = false;
    uint2 dnext = make_uint2(0u, 0u);
    const int vend = p.v0 + p.vcount;
    const bool whole_wg = p.vcount > 0 && (chunk_last - chunk_first) * p.vcount <= FX_VB;
    if (whole_wg) {
        const int nd = (chunk_last - chunk_first) * p.vcount;
        const uint2 *plan0 = p.plan + ((size_t)tile * p.nchunks + chunk_first) * p.V;
        for (int i = threadIdx.x; i < p.vcount * 12; i += 256) qtab[i] = p.Q[12 * p.v0 + i] * ((i % 12) >= 8 ? 0.00390625f : 1.0f);
        for (int i = threadIdx.x; i <= nd; i += 256) dtab[i] = i < nd ? plan0[(size_t)(i / p.vcount) * p.V + p.v0 + i % p.vcount] : make_uint2(0u, 0u);
        __syncthreads();
        dnext = dtab[0];
    }
    for (int chunk = chunk_first; chunk < chunk_last; chunk++) {
        const int d0 = chunk * PC;
        // SGPRs, by scalar loads through the constant address space (as per-lane global loads + v_readfirstlane each chunk began with 16
        // vector-memory round trips on its critical path: a tenth of a 4-view chunk's time at 640 x 480); planes past D are evaluated on
        // a clamped z and never stored
        float zc[PC];
        {
            const __attribute__((address_space(4))) float *zs = (const __attribute__((address_space(4))) float *)(uintptr_t)p.z;
#pragma unroll
            for (int k = 0; k < PC; k++) zc[k] = zs[min(d0 + k, p.D - 1)];
        }

        uint32_t acc[NPX][PC];
#pragma unroll
        for (int j = 0; j < NPX; j++)
#pragma unroll
            for (int k = 0; k < PC; k++) acc[j][k] = 0u;
        uint32_t row_checked = 0u;      // bit j: some view of this chunk took the per-sample in-frame test for row j of this wavefront (wave-uniform)
        uint32_t fast_views = 0u;       // views whose whole (tile, chunk) region is in frame: one count for every cell
        uint32_t lane_views[NPX];       // views of a BORDER region in which this pixel's whole plane range is in frame
#pragma unroll
        for (int j = 0; j < NPX; j++) lane_views[j] = 0u;
        const uint2 *plan = p.plan + ((size_t)tile * p.nchunks + chunk) * p.V;

        const int dbase = whole_wg ? (chunk - chunk_first) * p.vcount : 0;  // this chunk's first descriptor in the table
        for (int v = p.v0; v < vend; v++) {
            const int vi = (v - p.v0) & (FX_VB - 1);
            if (vi == 0 && !whole_wg) {  // next batch of per-view constants (wave-uniform branch; no region is in flight here: ahead == false)
                __syncthreads();
                const int nb = min(FX_VB, vend - v);
                // the w row goes in divided by 256: then RN(1 / s.w) IS 256 r, bit for bit (a power of two commutes with the roundings of
                // fma, v_rcp_f32 and the Newton step), and the sample loop needs no multiplication by 256
                for (int i = threadIdx.x; i < nb * 12; i += 256) qtab[i] = p.Q[12 * v + i] * ((i % 12) >= 8 ? 0.00390625f : 1.0f);
                for (int i = threadIdx.x; i <= nb; i += 256) dtab[i] = i < nb ? plan[v + i] : make_uint2(0u, 0u);  // + a SKIP sentinel
                __syncthreads();
                dnext = dtab[0];
            }
            // the per-view bookkeeping below is a chain of dependent scalar work and LDS round trips: let it overtake the other
            // wavefronts' sample loops on this SIMD (priority back to 0 before this wavefront's own sample loop)
            __builtin_amdgcn_s_setprio(3);
            const uint2 desc = dnext;
            dnext = dtab[dbase + vi + 1];  // the next region's descriptor (or the sentinel): in flight with this view's constants, one wait for all
            unsigned mode = (unsigned)__builtin_amdgcn_readfirstlane((int)((desc.y >> 16) & 7u));
            if (mode == FX_SKIP) continue;
            float q[12];  // wave-uniform values, kept in VGPRs: they are only ever VALU operands (v_fma allows one SGPR, and that is z)
            {
                const float4 qa = *(const float4 *)(qtab + 12 * vi), qb = *(const float4 *)(qtab + 12 * vi + 4), qc = *(const float4 *)(qtab + 12 * vi + 8);
                q[0] = qa.x; q[1] = qa.y; q[2] = qa.z; q[3] = qa.w; q[4] = qb.x; q[5] = qb.y; q[6] = qb.z; q[7] = qb.w;
                q[8] = qc.x; q[9] = qc.y; q[10] = qc.z; q[11] = qc.w;
            }
            const float bx = q[2], by = q[6], bw = q[10];
            if (__builtin_expect(mode == FX_GENERIC, 0)) {
                row_checked = ~0u;  // (its samples carry their own counts)
                const uint32_t *qv = p.quads + p.pad_slab * (size_t)__builtin_amdgcn_readfirstlane((int)(desc.y >> 19));
#pragma unroll
                for (int j = 0; j < NPX; j++) {
                    if (ok[j]) {
                        const Affine A = view_affine(q, xn, yn[j]);
#pragma unroll
                        for (int k = 0; k < PC; k++) acc[j][k] += sample_quads_fx(A, bx, by, bw, zc[k], qv, p.pitch, fhix, fhiy, lds, Im255[j]);
                    }
                }
                continue;
            }
            const int x0 = __builtin_amdgcn_readfirstlane((int)(desc.x & 0xffffu));
            const int y0 = __builtin_amdgcn_readfirstlane((int)(desc.x >> 16));
            const int rw = __builtin_amdgcn_readfirstlane((int)(desc.y & 0xffu));
            const int rh = __builtin_amdgcn_readfirstlane((int)((desc.y >> 8) & 0xffu));
            if (!ahead) {
                __syncthreads();  // every wavefront is done with the region these rows held
                qcol = 0;
                stage_region_fx(p.quads + p.pad_slab * (size_t)__builtin_amdgcn_readfirstlane((int)(desc.y >> 19)), p.pitch, x0, y0, rw, rh, qcol, lds);
            }
            __syncthreads();  // this view's region has landed (the barrier drains vmcnt) and the previous one is no longer read
            // Request the next staged view's region into the other half of the rows, if both regions are at most FX_HALF_COL quads
            // wide: its copy (L2 / Infinity Cache latency, 1-2 us) then overlaps this view's sampling (0.5 us of work for the
            // workgroup) and one barrier per region goes away.  Debug bit 3 switches it off (tests: bit-identical either way).
            ahead = false;
            int nqcol = 0;
            if (!(p.debug & 8) && rw <= FX_HALF_COL) {
                const unsigned mn = (unsigned)__builtin_amdgcn_readfirstlane((int)((dnext.y >> 16) & 7u));
                const int rwn = __builtin_amdgcn_readfirstlane((int)(dnext.y & 0xffu));
                if ((mn == FX_FAST || mn == FX_BORDER) && rwn <= FX_HALF_COL) {
                    nqcol = qcol ? 0 : FX_HALF_COL;  // (the next view, or after the last view the first view of the workgroup's next chunk: its slab is in the descriptor)
                    stage_region_fx(p.quads + p.pad_slab * (size_t)__builtin_amdgcn_readfirstlane((int)(dnext.y >> 19)), p.pitch, __builtin_amdgcn_readfirstlane((int)(dnext.x & 0xffffu)),
                                        __builtin_amdgcn_readfirstlane((int)(dnext.x >> 16)), rwn, __builtin_amdgcn_readfirstlane((int)((dnext.y >> 8) & 0xffu)), nqcol, lds);
                    ahead = true;
                }
            }
            FxRegion rg;
            const int kx0 = 256 * (x0 - qcol), ky0 = 256 * y0;  // region column rx sits in quad column qcol + rx of the LDS rows
            rg.offx = magic_plus(4 - kx0);
            rg.offy = magic_plus(4 - ky0);
            const bool wconst = uniform_f(bw) == 0.0f && !(p.debug & 4);  // wave-uniform: plane-independent w
            if (SEP && p.sep_y && mode == FX_FAST && uniform_f(q[1]) == 0.0f && uniform_f(q[4]) == 0.0f && uniform_f(q[8]) == 0.0f && uniform_f(q[9]) == 0.0f) {
                // the separable path (above): this view has the main camera's orientation and the whole region is in frame
                fast_views += 1u << 24;
                __builtin_amdgcn_s_setprio(0);
                const int wrow = ty * TILE_H + __builtin_amdgcn_readfirstlane(wave) * NPX;  // (the wavefront's first row, as a scalar)
                const uint2 *yt[NPX];
#pragma unroll
                for (int j = 0; j < NPX; j++) yt[j] = p.sep_y + ((size_t)v * p.H + min(wrow + j, p.H - 1)) * p.sep_dpad + d0;
                const Affine A0 = view_affine(q, xn, yn[0]);
                sample_view_sep(A0.ax, bx, zc, p.sep_r + (size_t)v * p.sep_dpad + d0, yt, rg.offx, lds_base, lds_base + 4u * FX_LUT_DW - ((uint32_t)y0 << 10), Im255, acc);
                qcol = nqcol;
                continue;
            }
#pragma unroll
            for (int j = 0; j < NPX; j++) {
                const Affine A = view_affine(q, xn, yn[j]);
                bool checked = false;
                if (mode == FX_BORDER) {
                    rg.lox = magic_plus(132 - kx0);
                    rg.loy = magic_plus(132 - ky0);
                    rg.hix = magic_plus(132 - kx0 + 256 * p.W);
                    rg.hiy = magic_plus(132 - ky0 + 256 * p.H);
                    // A pixel's samples over the chunk lie on a segment of the side image, monotone in z (w > 0 in the whole box): if
                    // both end planes are inside the frame by more than one 1/256-texel step (far above the f32 noise of the
                    // coordinates), every plane between them is in frame.  One wavefront-uniform decision per (row, view).
                    bool inside = true;
#pragma unroll
                    for (int e = 0; e < 2; e++) {
                        const float z = e ? zc[PC - 1] : zc[0];
                        const float sx = __builtin_fmaf(z, bx, A.ax), sy = __builtin_fmaf(z, by, A.ay), sw = __builtin_fmaf(z, bw, A.aw);
                        const float r256 = rcp_rn(sw);
                        const float Tx = __builtin_fmaf(sx, r256, rg.offx), Ty = __builtin_fmaf(sy, r256, rg.offy);
                        inside = inside && Tx > magic_plus(134 - kx0) && Tx < magic_plus(130 - kx0 + 256 * p.W) && Ty > magic_plus(134 - ky0) &&
                                 Ty < magic_plus(130 - ky0 + 256 * p.H);
                    }
                    checked = __builtin_amdgcn_ballot_w64(!inside && ok[j]) != 0ull;
                }
                if (checked) {
                    row_checked |= 1u << j;
                    if (ok[j]) sample_range_fx_checked<0, PC>(A, bx, by, bw, zc, rg, lds, Im255[j], acc[j]);
                    continue;
                }
                // every sample of this row's lanes is in frame: count once per view
                if (mode == FX_FAST) {
                    if (j == 0) fast_views += 1u << 24;
                } else {
                    lane_views[j] += 1u << 24;
                }
                __builtin_amdgcn_s_setprio(0);
                if (wconst) {
                    const float r256c = rcp_rn(__builtin_fmaf(zc[0], bw, A.aw));
                    sample_range_fx<0, PC, true>(A, bx, by, bw, r256c, zc, rg.offx, rg.offy, lds_base, Im255[j], acc[j]);
                } else {
                    sample_range_fx<0, PC, false>(A, bx, by, bw, 0.0f, zc, rg.offx, rg.offy, lds_base, Im255[j], acc[j]);
                }
            }
            qcol = nqcol;
        }

        // Epilogue of the chunk.  The plane base is wave-uniform (SGPR pair) and the pixel a 32-bit lane offset, so a store needs no
        // 64-bit per-lane pointer; the running best is kept as (sum, count) with the start value (1, 0), which makes
        // "s * bc < bs * c" true for the first cell with a view in frame and false for every empty cell: no other test per plane.
        uint32_t *const vol_chunk = WRITE_VOLUME ? p.volume + (size_t)d0 * P : nullptr;
        const bool whole = d0 + PC <= p.D;  // uniform: every plane of the chunk exists
        // the chunk's 16 planes as one buffer resource (plane k at byte offset 4 P k: below 2^32 for frames up to 8192^2; larger frames
        // take the pointer form): a store is resource + 32-bit lane offset + wave-uniform plane offset, no 64-bit per-lane pointer
        const bool vol_rsrc = WRITE_VOLUME && (unsigned long long)P * 4ull * (unsigned long long)PC < (1ull << 32);
        const __amdgpu_buffer_rsrc_t rvol = __builtin_amdgcn_make_buffer_rsrc((void *)vol_chunk, 0, (int)0xffffffffu, 0x00020000);
        const uint32_t plane_bytes = (uint32_t)(4u * (uint32_t)P);
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
                const uint32_t views = fast_views + lane_views[j];
                if (whole) {
                    // (the store form is decided ONCE per row: inside the plane loop the compiler kept both forms and a branch per store)
                    if (WRITE_VOLUME) {  // non-temporal: written once, read by a later kernel: keep it out of the L2 the quad images live in
                        if (vol_rsrc) {
#pragma unroll
                            for (int k = 0; k < PC; k++) __builtin_amdgcn_raw_buffer_store_b32(acc[j][k] + views, rvol, 4u * pix, plane_bytes * (uint32_t)k, 2);
                        } else {
#pragma unroll
                            for (int k = 0; k < PC; k++) __builtin_nontemporal_store(acc[j][k] + views, vol_chunk + (size_t)k * P + pix);
                        }
                    }
                    if (FUSED) {
                        if (!((row_checked >> j) & 1u)) {
                            // No view of this chunk tested this row's samples one by one: every accumulator is a bare sum (count field 0) and
                            // all 16 cells of a pixel get the same count `views`, so inside the chunk they compare like their sums.  With the
                            // plane in the low byte one v_min per cell keeps the lowest sum and, among equal sums, the lowest plane -- what
                            // the cross-multiplied comparison, strict and in plane order, decides; it is then needed once per chunk, not
                            // once per cell (2 instructions per cell instead of 6, and no 16-deep dependent chain).
                            uint32_t key[PC];
#pragma unroll
                            for (int k = 0; k < PC; k++) key[k] = (acc[j][k] << 8) | (uint32_t)k;
#pragma unroll
                            for (int w = PC / 2; w >= 1; w >>= 1)
#pragma unroll
                                for (int k = 0; k < w; k++) key[k] = min(key[k], key[k + w]);
                            const uint32_t cell = (key[0] >> 8) + views;
                            const bool better = umul24u(cell & 0xffffffu, best >> 24) < umul24u(best & 0xffffffu, cell >> 24);
                            best = better ? cell : best;
                            bi = better ? d0 + (int)(key[0] & 0xffu) : bi;
                        } else {
#pragma unroll
                            for (int k = 0; k < PC; k++) {
                                const uint32_t cell = acc[j][k] + views;
                                const bool better = umul24u(cell & 0xffffffu, best >> 24) < umul24u(best & 0xffffffu, cell >> 24);
                                best = better ? cell : best;
                                bi = better ? d0 + k : bi;
                            }
                        }
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < PC; k++) {
                        if (d0 + k < p.D) {
                            const uint32_t cell = acc[j][k] + views;
                            if (WRITE_VOLUME) __builtin_nontemporal_store(cell, vol_chunk + (size_t)k * P + pix);  // written once, read by a later kernel: keep it out of the L2 the quad images live in
                            if (FUSED) {
                                const bool better = umul24u(cell & 0xffffffu, best >> 24) < umul24u(best & 0xffffffu, cell >> 24);
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
                if ((int)st.y < 0) st.x = 0u;  // no plane had a view in frame: the empty cell, as argmin_update_packed leaves it
                const size_t pix = (size_t)(row0 + j) * p.W + col;
                if (p.part)
                    p.part[(size_t)blockIdx.y * P + pix] = st;
                else
                    store_best<CS_FIXED>(p, pix, st.x & 0xffffffu, st.x >> 24, (int)st.y);
            }
    }
}

template <bool WRITE_VOLUME, bool FUSED, bool SEP = false>
__global__ __launch_bounds__(256, FX_WG_PER_CU) void sweep_fx_tiled(SweepParams p, const uint32_t *__restrict__ lut_g)
{
    sweep_fx_tiled_body<WRITE_VOLUME, FUSED, SEP>(p, lut_g);
}

// several main frames in one launch (mvs_sweep_batch): blockIdx.z selects the frame's parameter block; depth selection only
__global__ __launch_bounds__(256, FX_WG_PER_CU) void sweep_fx_tiled_batch(const SweepParams *__restrict__ pp, const uint32_t *__restrict__ lut_g)
{
    const SweepParams p = load_params(pp + blockIdx.z);
    sweep_fx_tiled_body<false, true>(p, lut_g);
}

// the fixed sampler at one plane per pixel, z = depth[p]: (round(dot / 255), mask)
__global__ __launch_bounds__(256) void warp_by_depth_fx_kernel(const float *__restrict__ depth, const float *__restrict__ Q, const uint8_t *__restrict__ pad,
                                                               int pitch, int W, int H, float invW, float invH, const uint32_t *__restrict__ lut,
                                                               uint8_t *__restrict__ out2)
{
    const int col = blockIdx.x * 64 + (threadIdx.x & 63);
    const int row = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (col >= W || row >= H) return;
    const size_t pix = (size_t)row * W + col;
    const float z = depth[pix];
    uint32_t cell = 0u, dot = 0u;
    if (z != MVS_BACKGROUND_DEPTH) {
        const float xn = __builtin_fmaf((float)(2 * col + 1), invW, -1.0f);
        const float yn = __builtin_fmaf(-(float)(2 * row + 1), invH, 1.0f);
        const Affine A = view_affine(Q, xn, yn);
        cell = sample_global_fx(A, Q[2], Q[6], Q[10], z, pad, pitch, FX_MAGIC + 132.0f + 256.0f * (float)W, FX_MAGIC + 132.0f + 256.0f * (float)H, lut, 0u,
                                &dot);
    }
    out2[2 * pix] = cell ? (uint8_t)((dot + 127u) / 255u) : 0;
    out2[2 * pix + 1] = cell ? 255 : 0;
}

// the 32 x 32 weight table (oracle: orc_fx_weight_table): entry [ky][kx] = (w00, w01, w10, w11) packed little-endian, sum 255
void fx_weight_table(uint32_t *lut)
{
    for (int b = 0; b < 32; b++)
        for (int a = 0; a < 32; a++) {
            const int pr[4] = {(32 - a) * (32 - b), a * (32 - b), (32 - a) * b, a * b};
            int w[4], sum = 0, big = 0;
            for (int k = 0; k < 4; k++) {
                w[k] = (255 * pr[k] + 512) >> 10;
                sum += w[k];
                if (pr[k] > pr[big]) big = k;
            }
            w[big] += 255 - sum;
            lut[b * 32 + a] = (uint32_t)w[0] | ((uint32_t)w[1] << 8) | ((uint32_t)w[2] << 16) | ((uint32_t)w[3] << 24);
        }
}

int ensure_fx_lut(mvs_ctx *ctx)
{
    if (ctx->fx_lut.ptr) return MVS_OK;
    int rc = ensure(ctx, ctx->fx_lut, 4096);
    if (rc) return rc;
    uint32_t host[1024];
    fx_weight_table(host);
    MVS_HIP(ctx, hipMemcpyAsync(ctx->fx_lut.ptr, host, sizeof(host), hipMemcpyHostToDevice, ctx->stream));
    MVS_HIP(ctx, hipStreamSynchronize(ctx->stream));  // `host` is on this stack frame
    return MVS_OK;
}

// launch of the fixed-sampler sweep; `p` carries the plane / row / view ranges (sweep_run_impl in sweep.hip)
int sweep_fx_launch(mvs_ctx *ctx, SweepParams &p, bool vol, bool fused, bool generic, unsigned flags)
{
    int rc = ensure_fx_lut(ctx);
    if (rc) return rc;
    const uint32_t *lut = (const uint32_t *)ctx->fx_lut.ptr;
    if (generic || ctx->V == 0) {
        dim3 grid(div_up(ctx->W, 64), div_up(p.row_end - p.row_begin, 4));
        if (vol && fused)
            sweep_fx_generic<true, true><<<grid, 256, 0, ctx->stream>>>(p, lut);
        else if (vol)
            sweep_fx_generic<true, false><<<grid, 256, 0, ctx->stream>>>(p, lut);
        else
            sweep_fx_generic<false, true><<<grid, 256, 0, ctx->stream>>>(p, lut);
        MVS_HIP(ctx, hipGetLastError());
        return MVS_OK;
    }
    if (ctx->sep_ok && ctx->fx_general_planned) {
        p.sep_y = (const uint2 *)ctx->sep_tab.ptr;
        p.sep_r = (const float *)((const char *)ctx->sep_tab.ptr + ctx->sep_r_offset);
        p.sep_dpad = ctx->sep_dpad;
    }
    const int groups = div_up(p.tiles_x, 2) * div_up(p.tyn, 4);
    const int nch = p.chunk1 - p.chunk0, tiles = p.tiles_x * p.tyn;
    int want = (int)((flags >> 16) & 0xffu);  // undocumented: forced split count for timing experiments
    if (!want) want = div_up(16 * ctx->num_cus, tiles);  // ~16 workgroups per CU (4 rounds of 4): c3 2 splits 1.44 ms (1: 1.46, 4: 1.46, 8: 1.50), c2 0.21 ms, c1 0.037 ms
    p.cps = div_up(nch, max(1, min(want, nch)));
    const int nsplit = div_up(nch, p.cps);
    if (fused && nsplit > 1) {
        if ((rc = ensure(ctx, ctx->best_parts, (size_t)nsplit * ctx->W * ctx->H * sizeof(uint2)))) return rc;
        p.part = (uint2 *)ctx->best_parts.ptr;
    }
    const dim3 grid((unsigned)(div_up(groups, 8) * 64), (unsigned)nsplit);
    if (p.sep_y) {  // some view qualifies for the separable path: the instantiation that has it
        if (vol && fused)
            sweep_fx_tiled<true, true, true><<<grid, 256, 0, ctx->stream>>>(p, lut);
        else if (vol)
            sweep_fx_tiled<true, false, true><<<grid, 256, 0, ctx->stream>>>(p, lut);
        else
            sweep_fx_tiled<false, true, true><<<grid, 256, 0, ctx->stream>>>(p, lut);
    } else if (vol && fused)
        sweep_fx_tiled<true, true><<<grid, 256, 0, ctx->stream>>>(p, lut);
    else if (vol)
        sweep_fx_tiled<true, false><<<grid, 256, 0, ctx->stream>>>(p, lut);
    else
        sweep_fx_tiled<false, true><<<grid, 256, 0, ctx->stream>>>(p, lut);
    MVS_HIP(ctx, hipGetLastError());
    return nsplit;  // > 0: the caller merges the partial bests when p.part is set
}

// The plan of the general tiled kernel (one descriptor per (tile, chunk, view)).  When every view is rectified the sweep runs on
// sweep_fx_rect's own tables and this plan is made only if a launch asks for the general kernel (MVS_SWEEP_NO_RECT, a run over zero
// views): a third of a millisecond at c3 that the one-call path does not have to wait for.
int sweep_fx_plan_general(mvs_ctx *ctx)
{
    int rc;
    if ((rc = ensure(ctx, ctx->plan_stats, 64))) return rc;
    SweepParams q;
    fill_params(ctx, q, 0, ctx->V, FX_TILE_H, FX_PC);
    const size_t n = (size_t)q.tiles_x * q.tiles_y * q.nchunks * q.V;
    if ((rc = ensure(ctx, ctx->plan, n * sizeof(uint2)))) return rc;
    q.plan = (const uint2 *)ctx->plan.ptr;
    // the planner's counters (oversize regions, regions not skipped, widest / tallest staged region) are a diagnostic nobody reads on
    // this path: 32 000 wavefronts' atomics on four addresses were most of the planner's 0.28 ms at c3.  MVS_PLAN_DUMP keeps them.
    q.plan_stats = !ctx->hooks.plan_dump.empty() ? (int *)ctx->plan_stats.ptr : nullptr;
    if (q.plan_stats) MVS_HIP(ctx, hipMemsetAsync(q.plan_stats, 0, 4 * sizeof(int), ctx->stream));
    plan_regions_fx<<<(unsigned)((n + 255) / 256), 256, 0, ctx->stream>>>(q, (uint2 *)ctx->plan.ptr);
    MVS_HIP(ctx, hipGetLastError());
    if (const char *path = ctx->hooks.plan_dump.empty() ? nullptr : ctx->hooks.plan_dump.c_str()) {  // diagnostic (tools/plan_hist.py): header {tiles_x, tiles_y, nchunks, V}, then the descriptors
        std::vector<uint2> host(n);
        MVS_HIP(ctx, hipMemcpyAsync(host.data(), ctx->plan.ptr, n * sizeof(uint2), hipMemcpyDeviceToHost, ctx->stream));
        MVS_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (FILE *f = fopen(path, "wb")) {
            const int hdr[4] = {q.tiles_x, q.tiles_y, q.nchunks, q.V};
            fwrite(hdr, sizeof(int), 4, f);
            fwrite(host.data(), sizeof(uint2), n, f);
            fclose(f);
        }
    }
    // the separable path's tables, when a view qualifies (plan_sep_tables above): 8 bytes per (view, image row, plane)
    ctx->sep_ok = false;
    if (!ctx->hooks.no_sep) {
        bool any = false;
        for (int v = 0; v < ctx->V && !any; v++) {
            const float *m = ctx->q_host.data() + 12 * v;
            any = m[1] == 0.0f && m[4] == 0.0f && m[8] == 0.0f && m[9] == 0.0f;
        }
        const int dpad = q.nchunks * FX_PC;
        const size_t ny = (size_t)ctx->V * ctx->H * dpad, nr = (size_t)ctx->V * dpad;
        if (any && ny * sizeof(uint2) <= ((size_t)256 << 20)) {
            const size_t off_r = (ny * sizeof(uint2) + 255) & ~(size_t)255;
            if ((rc = ensure(ctx, ctx->sep_tab, off_r + nr * sizeof(float)))) return rc;
            plan_sep_tables<<<(unsigned)((ny + 255) / 256), 256, 0, ctx->stream>>>(q, (float *)((char *)ctx->sep_tab.ptr + off_r), (uint2 *)ctx->sep_tab.ptr, dpad);
            MVS_HIP(ctx, hipGetLastError());
            ctx->sep_ok = true;
            ctx->sep_dpad = dpad;
            ctx->sep_r_offset = off_r;
        }
    }
    ctx->fx_general_planned = true;
    return MVS_OK;
}

int sweep_fx_plan(mvs_ctx *ctx, PlanHook *between)
{
    // The plan depends on the view matrices (main and side cameras), the planes and, for store-resident views, the slots -- not on the
    // frames.  A fixed rig that delivers its next set of frames (the commonest "new view set" of all) therefore needs no planning: the
    // tables in memory are the ones this call would compute.  (mvs_sweep_set_plan_cache(ctx, 0) plans regardless: bench.py's cold step times the plan.)
    const bool hit = ctx->snap_valid && ctx->snap_q == ctx->q_host && ctx->snap_z == ctx->z_host && ctx->snap_in_store == ctx->views_in_store &&
                     (!ctx->views_in_store || ctx->snap_slots == ctx->view_slots_host) && ctx->hooks.plan_dump.empty() && ctx->plan_cache;
    if (hit) return between ? between->run() : MVS_OK;
    ctx->snap_valid = false;
    ctx->fx_general_planned = false;
    int rc = sweep_rect_plan(ctx, between);  // (runs `between` exactly once, whatever it decides)
    if (rc) return rc;
    if (!ctx->rect_ok || !ctx->hooks.plan_dump.empty()) {
        if ((rc = sweep_fx_plan_general(ctx))) return rc;
    }
    ctx->snap_q = ctx->q_host;
    ctx->snap_z = ctx->z_host;
    ctx->snap_in_store = ctx->views_in_store;
    ctx->snap_slots = ctx->view_slots_host;
    ctx->snap_valid = true;
    return MVS_OK;
}

int warp_by_depth_fx_launch(mvs_ctx *ctx, const float *depth_dev, const float *q_dev, const uint8_t *pad_dev, int pitch, uint8_t *out2_dev)
{
    int rc = ensure_fx_lut(ctx);
    if (rc) return rc;
    const int W = ctx->W, H = ctx->H;
    warp_by_depth_fx_kernel<<<dim3(div_up(W, 64), div_up(H, 4)), 256, 0, ctx->stream>>>(depth_dev, q_dev, pad_dev, pitch, W, H, 1.0f / (float)W, 1.0f / (float)H,
                                                                                       (const uint32_t *)ctx->fx_lut.ptr, out2_dev);
    MVS_HIP(ctx, hipGetLastError());
    return MVS_OK;
}

// ------------------------------------------------------------------------------------------------------
// batched sweep over the frame store (recon.cpp:65-117: every main frame of a sequence against its neighbours)
// ------------------------------------------------------------------------------------------------------
// all queued batches have delivered their results (and their slots are free again)
int sweep_batch_wait_impl(mvs_ctx *ctx)
{
    if (!ctx) return MVS_EINVAL;
    int rc = MVS_OK;
    for (auto &b : ctx->batch_slot)
        if (b.busy) {
            if (hipEventSynchronize(b.landed) != hipSuccess) rc = fail(ctx, MVS_EHIP, "mvs_sweep_batch_wait: hipEventSynchronize failed");
            b.busy = false;
        }
    return rc;
}

// Queues one batch and returns: parameter blocks and plan on the context's stream, the sweep behind them, the results' way home on the
// copy stream (ordered behind the sweep by an event) -- so the NEXT batch's planner and sweep start while this batch's depth maps are
// still crossing PCIe, and the host prepares that next batch meanwhile.  Two batches in flight; a third call waits for the oldest.
int sweep_batch_impl(mvs_ctx *ctx, int nmain, const int *main_slots, const float *main_cams, int nside, const int *side_slots, const float *side_cams,
                     int nplanes, float z_lo, float z_hi, float *depth_out, float *cost_out)
{
    if (!ctx) return MVS_EINVAL;
    if (nmain < 1 || nside < 1 || nside > 255 || !main_slots || !main_cams || !side_slots || !side_cams || !depth_out)
        return fail(ctx, MVS_EINVAL, "mvs_sweep_batch: bad arguments (nmain=%d, nside=%d: 1..255 side views per main frame)", nmain, nside);
    if (nplanes < 1 || nplanes > 4096) return fail(ctx, MVS_EINVAL, "mvs_sweep_batch: nplanes=%d out of range 1..4096", nplanes);
    if (ctx->sampler != MVS_SAMPLER_FIXED) return fail(ctx, MVS_EINVAL, "mvs_sweep_batch: implemented for MVS_SAMPLER_FIXED (the library default)");
    if (ctx->W > 16383 || ctx->H > 16383) return fail(ctx, MVS_EINVAL, "mvs_sweep_batch: images of up to 16383 x 16383");
    for (int m = 0; m < nmain; m++) {
        if (main_slots[m] < 0 || main_slots[m] >= ctx->store_cap || !ctx->store_have[main_slots[m]])
            return fail(ctx, MVS_EINVAL, "mvs_sweep_batch: main frame %d: slot %d holds no frame", m, main_slots[m]);
        for (int v = 0; v < nside; v++) {
            const int sl = side_slots[(size_t)m * nside + v];
            if (sl < 0 || sl >= ctx->store_cap || !ctx->store_have[sl]) return fail(ctx, MVS_EINVAL, "mvs_sweep_batch: main frame %d, side view %d: slot %d holds no frame", m, v, sl);
        }
    }
    MVS_HIP(ctx, hipSetDevice(ctx->device));
    const int W = ctx->W, H = ctx->H;
    const size_t P = (size_t)W * H;
    const int pitch = ((W + 2 + 63) / 64) * 64;
    const size_t slab = (size_t)pitch * (H + 2);
    int rc = ensure_fx_lut(ctx);
    if (rc) return rc;

    // one device block: [parameter blocks][view matrices][plane table][view slots][region plans][depth, cost, index per frame]
    const int tiles_x = div_up(W, TILE_W), tiles_y = div_up(H, FX_TILE_H), nchunks = div_up(nplanes, FX_PC);
    const size_t plan_entries = (size_t)tiles_x * tiles_y * nchunks * nside;
    auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t o_params = 0, o_q = up(o_params + sizeof(SweepParams) * nmain), o_z = up(o_q + sizeof(float) * 12 * nside * nmain),
                 o_slots = up(o_z + sizeof(float) * nplanes), o_stats = up(o_slots + sizeof(int) * nside * nmain), o_plan = up(o_stats + 64),
                 o_out = up(o_plan + sizeof(uint2) * plan_entries * nmain), total = up(o_out + 3 * P * sizeof(float) * nmain);
    if (!ctx->copy_stream) MVS_HIP(ctx, hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
    mvs_ctx::BatchSlot &slot = ctx->batch_slot[ctx->batch_next];
    ctx->batch_next ^= 1;
    if (slot.busy) {  // the batch before the previous one used this slot: its results must have landed before its buffers are reused
        MVS_HIP(ctx, hipEventSynchronize(slot.landed));
        slot.busy = false;
    }
    if (!slot.swept) {
        MVS_HIP(ctx, hipEventCreateWithFlags(&slot.swept, hipEventDisableTiming));
        MVS_HIP(ctx, hipEventCreateWithFlags(&slot.landed, hipEventDisableTiming));
    }
    if (total > slot.buf.bytes && slot.buf.ptr) {  // (ensure() below frees the old block after joining the context's stream only)
        MVS_HIP(ctx, hipStreamSynchronize(ctx->copy_stream));
    }
    if ((rc = ensure(ctx, slot.buf, total))) return rc;
    char *base = (char *)slot.buf.ptr;
    // host staging: lives in the slot until the slot is reused (the copy below is asynchronous)
    std::vector<char> &host = slot.host;
    host.assign(o_plan, 0);
    float *zt = (float *)(host.data() + o_z);
    plane_table(nplanes, z_lo, z_hi, zt);
    memcpy(host.data() + o_slots, side_slots, sizeof(int) * nside * nmain);
    for (int m = 0; m < nmain; m++) {
        float *qm = (float *)(host.data() + o_q) + (size_t)12 * nside * m;
        for (int v = 0; v < nside; v++) view_matrix(main_cams + 16 * m, side_cams + 16 * ((size_t)m * nside + v), W, H, qm + 12 * v);
        SweepParams &p = ((SweepParams *)(host.data() + o_params))[m];
        memset(&p, 0, sizeof(p));
        p.main_img = (const uint8_t *)ctx->store_raw.ptr + P * main_slots[m];
        p.pads = nullptr;  // (the tiled kernel reads quad images only)
        p.pad_slab = slab;
        p.quads = (const uint32_t *)ctx->store_quads.ptr;
        p.quads16 = nullptr;
        p.pitch = pitch;
        p.W = W;
        p.H = H;
        p.D = nplanes;
        p.V = nside;
        p.v0 = 0;
        p.vcount = nside;
        p.Q = (const float *)(base + o_q) + (size_t)12 * nside * m;
        p.z = (const float *)(base + o_z);
        p.volume = nullptr;
        p.depth = (float *)(base + o_out) + (size_t)m * P;                   // [depth of every frame][cost of every frame][index of every frame]
        p.cost = (float *)(base + o_out) + (size_t)(nmain + m) * P;
        p.index = (int *)(base + o_out) + (size_t)(2 * nmain + m) * P;
        p.invW = 1.0f / (float)W;
        p.invH = 1.0f / (float)H;
        p.Wp = (float)W + 0.5f;
        p.Hp = (float)H + 0.5f;
        p.plan = (const uint2 *)(base + o_plan) + plan_entries * m;
        p.tiles_x = tiles_x;
        p.tiles_y = tiles_y;
        p.nchunks = nchunks;
        p.chunk0 = 0;
        p.chunk1 = nchunks;
        p.ty0 = 0;
        p.tyn = tiles_y;
        p.tile_h = FX_TILE_H;
        p.pc = FX_PC;
        p.row_begin = 0;
        p.row_end = H;
        p.plane_begin = 0;
        p.plane_end = nplanes;
        p.plan_stats = nullptr;  // (planner counters: not needed here)
        p.cps = nchunks;  // several frames fill the chip: no plane split, no partial bests
        p.part = nullptr;
        p.view_slot = (const int *)(base + o_slots) + (size_t)nside * m;
        p.debug = 0;
    }
    MVS_HIP(ctx, hipMemcpyAsync(base, host.data(), o_plan, hipMemcpyHostToDevice, ctx->stream));
    MVS_HIP(ctx, hipMemsetAsync(base + o_stats, 0, 64, ctx->stream));
    {
        ProfileScope ps(ctx, MVS_K_PLAN);
        plan_regions_fx_batch<<<dim3((unsigned)((plan_entries + 255) / 256), (unsigned)nmain), 256, 0, ctx->stream>>>((const SweepParams *)(base + o_params));
        MVS_HIP(ctx, hipGetLastError());
    }
    {
        ProfileScope ps(ctx, MVS_K_SWEEP);
        const int groups = div_up(tiles_x, 2) * div_up(tiles_y, 4);
        sweep_fx_tiled_batch<<<dim3((unsigned)(div_up(groups, 8) * 64), 1, (unsigned)nmain), 256, 0, ctx->stream>>>((const SweepParams *)(base + o_params),
                                                                                                             (const uint32_t *)ctx->fx_lut.ptr);
        MVS_HIP(ctx, hipGetLastError());
    }
    MVS_HIP(ctx, hipEventRecord(slot.swept, ctx->stream));
    MVS_HIP(ctx, hipStreamWaitEvent(ctx->copy_stream, slot.swept, 0));
    MVS_HIP(ctx, hipMemcpyAsync(depth_out, base + o_out, P * sizeof(float) * nmain, hipMemcpyDeviceToHost, ctx->copy_stream));
    if (cost_out) MVS_HIP(ctx, hipMemcpyAsync(cost_out, base + o_out + P * sizeof(float) * nmain, P * sizeof(float) * nmain, hipMemcpyDeviceToHost, ctx->copy_stream));
    MVS_HIP(ctx, hipEventRecord(slot.landed, ctx->copy_stream));
    slot.busy = true;
    return MVS_OK;
}

}  // namespace mvs

extern "C" int mvs_sweep_batch(mvs_ctx *ctx, int nmain, const int *main_slots, const float *main_cams, int nside, const int *side_slots, const float *side_cams, int nplanes,
                               float z_lo, float z_hi, float *depth_out, float *cost_out)
{
    const int rc = mvs::sweep_batch_impl(ctx, nmain, main_slots, main_cams, nside, side_slots, side_cams, nplanes, z_lo, z_hi, depth_out, cost_out);
    const int rw = mvs::sweep_batch_wait_impl(ctx);  // (also after an error: nothing of an earlier asynchronous batch stays in flight)
    if (rc) return rc;
    if (rw) return rw;
    if (ctx && hipStreamSynchronize(ctx->stream) != hipSuccess) return mvs::fail(ctx, MVS_EHIP, "mvs_sweep_batch: hipStreamSynchronize failed");  // "synchronises": queued uploads included
    return MVS_OK;
}

extern "C" int mvs_sweep_batch_async(mvs_ctx *ctx, int nmain, const int *main_slots, const float *main_cams, int nside, const int *side_slots, const float *side_cams,
                                     int nplanes, float z_lo, float z_hi, float *depth_out, float *cost_out)
{
    return mvs::sweep_batch_impl(ctx, nmain, main_slots, main_cams, nside, side_slots, side_cams, nplanes, z_lo, z_hi, depth_out, cost_out);
}

extern "C" int mvs_sweep_batch_wait(mvs_ctx *ctx) { return mvs::sweep_batch_wait_impl(ctx); }
