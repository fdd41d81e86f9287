// sweep_shared.hpp -- what the two sweep samplers (sweep.hip: exact f32, contract v1; sweep_fx.hip: fixed point, contract v2)
// have in common: launch parameters, the projective arithmetic up to s and r = RN(1/s.w), the XCD-aware tile order and the
// packed-cell helpers.  Cells are  count << CS | sum  with CS = 16 (exact sampler: sum of |u8 - u8|, <= 257 views) or
// CS = 24 (fixed sampler: sum of |weights.texels - 255 I_main|, <= 255 views).
#pragma once
#include "mvs_internal.hpp"

namespace mvs {

constexpr int CS_EXACT = 16, CS_FIXED = 24;

typedef _Float16 half4_t __attribute__((ext_vector_type(4)));

constexpr int TILE_W = 64;   // one wavefront spans a tile row
// sweep_tiled is instantiated for two thread shapes with 64 accumulators each (NPX pixels x PC planes per thread, tile
// height 4 * NPX): 2 x 32 amortises the per-(pixel, view) set-up and the staged texels over twice as many planes
// (c3: 2.20 -> 2.09 ms) but needs the warped footprint of 32 consecutive planes to fit the LDS region; 4 x 16 is the
// fallback when the planner reports oversize regions (wide baselines with few planes).  Chosen per plan, see sweep_run_impl.
constexpr int PCG = 16;          // planes per accumulator batch of the un-tiled generic kernel
constexpr int ROW_GRAN = 16;     // public row granularity: a multiple of both tile heights
constexpr int PLANE_GRAN = 32;   // public plane granularity: a multiple of both chunk sizes
constexpr int LDS_QUADS = 5120;  // 40 KiB of 8-byte quads per staging buffer
constexpr int MAX_RW = 192;
constexpr float PLAN_MARGIN = 0.0625f;

enum RegionMode : unsigned { R_SKIP = 0, R_FAST = 1, R_BORDER = 2, R_GENERIC = 3 };

struct SweepParams {
    const uint8_t *__restrict__ main_img;
    const uint8_t *__restrict__ pads;
    size_t pad_slab;
    int pitch;
    const uint32_t *__restrict__ quads;  // fixed sampler: per view (H+2) x pitch packed bilinear footprints (t00, t01, t10, t11); pad_slab dwords per view
    const uint2 *__restrict__ quads16;   // exact sampler: per view (H+2) x pitch f16 quads {t00 + 0.5, t01 - t00, t10 - t00, dxy}; pad_slab quads per view
    int W, H, D, V;
    int v0, vcount;
    const float *__restrict__ Q;  // V * 12
    const float *__restrict__ z;  // D
    uint32_t *__restrict__ volume;
    float *__restrict__ depth;
    float *__restrict__ cost;
    int *__restrict__ index;
    float invW, invH;
    float Wp, Hp;  // W + 0.5, H + 0.5
    const uint2 *__restrict__ plan;
    int tiles_x, tiles_y, nchunks;
    int chunk0, chunk1;  // plane chunks [chunk0, chunk1) processed by this launch
    int ty0, tyn;        // tile rows [ty0, ty0 + tyn) processed by this launch (row-band sharding)
    int tile_h, pc;      // shape of the tiled kernel this plan was made for (tile height, planes per chunk)
    int row_begin, row_end, plane_begin, plane_end;  // the same ranges in pixels / planes (generic kernel)
    int *__restrict__ plan_stats;  // [0] regions too large for LDS, [1] regions not skipped (planner output)
    int cps;             // plane chunks per workgroup: blockIdx.y selects chunks [chunk0 + y*cps, +cps) of a tile
    uint2 *__restrict__ part;  // plane-split launches with fused depth selection: [gridDim.y][P] partial bests
    const int *__restrict__ view_slot;  // nullable: slab of view v's padded / quad image (frame store slots, mvs_sweep_batch); null = slab v
    // separable path of the fixed sampler's general kernel (sweep_fx.hip: plan_sep_tables; null = not available): for views whose matrix has
    // q1 = q4 = q8 = q9 = 0 (a camera with the main camera's orientation, anywhere), RN(1 / s.w) per (view, plane) and the row part of both LDS
    // addresses per (view, image row, plane)
    const float *__restrict__ sep_r;   // [V][sep_dpad]
    const uint2 *__restrict__ sep_y;   // [V][H][sep_dpad]: (ky << 10, iy << 10)
    int sep_dpad, sep_reserved;
    int debug;  // timing experiments only (bit 0: skip LDS staging -> wrong results; bit 1: linear tile order; bit 2: never use the plane-independent-w path; bit 3: exact sampler: force the 4 x 16 shape, fixed sampler: no region look-ahead; fixed sampler only: bit 4: BORDER regions as FAST, bit 5: skip the sample loop)
};

// ------------------------------------------------------------------------------------------------------
// shared sample arithmetic
// ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float rcp_rn(float w)
{
    // v_rcp_f32 is accurate to 1 ulp; one Newton step with FMA yields the correctly rounded
    // reciprocal (Markstein) for every w whose significand is not all ones -- verified
    // exhaustively on the device by tests/test_sweep_gpu.py::test_rcp_newton_exact.
    const float r0 = __builtin_amdgcn_rcpf(w);
    const float e = __builtin_fmaf(-w, r0, 1.0f);
    return __builtin_fmaf(e, r0, r0);
}

struct Affine {
    float ax, ay, aw;
};

__device__ __forceinline__ Affine view_affine(const float *__restrict__ q, float xn, float yn)
{
    Affine a;
    a.ax = __builtin_fmaf(q[0], xn, __builtin_fmaf(q[1], yn, q[3]));
    a.ay = __builtin_fmaf(q[4], xn, __builtin_fmaf(q[5], yn, q[7]));
    a.aw = __builtin_fmaf(q[8], xn, __builtin_fmaf(q[9], yn, q[11]));
    return a;
}

// HIP declares __umul24 as returning int: compared as such, products >= 2^31 (fixed sampler, > 181 views at maximal cost) would order wrongly
__device__ __forceinline__ uint32_t umul24u(uint32_t a, uint32_t b) { return (uint32_t)__umul24(a, b); }

// running best plane: s/c < bs/bc  <=>  s*bc < bs*c; the products stay below 2^32 (s < 2^16, c < 2^16 or s < 2^24, c < 2^8)
template <int CS = CS_EXACT>
__device__ __forceinline__ void argmin_update(uint32_t cell, int d, uint32_t &bs, uint32_t &bc, int &bi)
{
    const uint32_t s = cell & ((1u << CS) - 1u), c = cell >> CS;  // s < 2^24, c < 2^16: the 24-bit multiplies below are exact
    const bool better = (c != 0u) && (bi < 0 || umul24u(s, bc) < umul24u(bs, c));
    bs = better ? s : bs;
    bc = better ? c : bc;
    bi = better ? d : bi;
}

// the same with the best (count, sum) kept as one packed cell: two registers of state per pixel
template <int CS = CS_EXACT>
__device__ __forceinline__ void argmin_update_packed(uint32_t cell, int d, uint32_t &best, int &bi)
{
    constexpr uint32_t M = (1u << CS) - 1u;
    const uint32_t s = cell & M, c = cell >> CS, bs = best & M, bc = best >> CS;
    const bool better = (c != 0u) && (bi < 0 || umul24u(s, bc) < umul24u(bs, c));
    best = better ? cell : best;
    bi = better ? d : bi;
}

// mean cost in grey levels: sum / count (exact sampler) or sum / (255 count) (fixed sampler: sums are in 1/255 grey levels)
template <int CS = CS_EXACT>
__device__ __forceinline__ float cell_cost(uint32_t bs, uint32_t bc)
{
    return CS == CS_EXACT ? (float)bs / (float)bc : (float)bs / (float)(255u * bc);
}

template <int CS = CS_EXACT>
__device__ __forceinline__ void store_best(const SweepParams &p, size_t pix, uint32_t bs, uint32_t bc, int bi)
{
    p.depth[pix] = bi >= 0 ? p.z[bi] : MVS_BACKGROUND_DEPTH;
    p.cost[pix] = bi >= 0 ? cell_cost<CS>(bs, bc) : __builtin_inff();
    p.index[pix] = bi;
}


// wave-uniform value -> SGPR
__device__ __forceinline__ float uniform_f(float x)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, x)));
}

// Workgroups are dealt round-robin over the 8 XCDs (blockIdx % 8 shares an XCD, MI355X_MICROARCH.md), each with
// its own 4 MiB L2.  Tiles are grouped 2 wide x 4 tall (128 x 64 pixels); the 8 tiles of a group get block ids
// with equal (id % 8), so a group's overlapping side-image regions share one L2, and consecutive groups go to
// consecutive XCDs, so border tiles (which take the slower per-sample-test path) spread evenly over the chip
// (a contiguous band per XCD cut HBM fetches 3.2x but ran 5 % slower: profiles/r01).  Bijective onto the padded
// group grid; ids that fall outside the image exit.  Placement affects speed and traffic only.
constexpr int GROUP_W = 2, GROUP_H = 4;

__device__ __forceinline__ int grouped_tile(int bid, int tiles_x, int tiles_y)
{
    const int gx = (tiles_x + GROUP_W - 1) / GROUP_W;
    const int sb = bid >> 6, x = bid & 7, m = (bid >> 3) & 7;
    const int g = sb * 8 + x;
    const int tx = (g % gx) * GROUP_W + (m & (GROUP_W - 1));
    const int ty = (g / gx) * GROUP_H + (m >> 1);
    return (tx < tiles_x && ty < tiles_y) ? ty * tiles_x + tx : -1;
}

// the main view: the context's own copy, or a slot of the frame store (mvs_sweep_handles)
inline const uint8_t *main_image_ptr(const mvs_ctx *ctx)
{
    return ctx->main_store_slot >= 0 ? (const uint8_t *)ctx->store_raw.ptr + (size_t)ctx->W * ctx->H * (size_t)ctx->main_store_slot : (const uint8_t *)ctx->main_img.ptr;
}

// host: launch parameters of the whole image / all planes; callers narrow the ranges
inline int fill_params(mvs_ctx *ctx, SweepParams &p, int v0, int vcount, int tile_h, int pc)
{
    p.main_img = main_image_ptr(ctx);
    p.pads = ctx->pads_valid ? (const uint8_t *)ctx->side_pads.ptr : nullptr;  // rebuilt on demand (ensure_pads) by the paths that gather single texels
    p.pad_slab = ctx->pad_slab;
    p.quads = ctx->views_in_store ? (const uint32_t *)ctx->store_quads.ptr : (const uint32_t *)ctx->side_quads.ptr;
    p.quads16 = (const uint2 *)ctx->side_quads16.ptr;
    p.pitch = ctx->pad_pitch;
    p.W = ctx->W;
    p.H = ctx->H;
    p.D = ctx->D;
    p.V = ctx->V;
    p.v0 = v0;
    p.vcount = vcount;
    p.Q = (const float *)ctx->qmats.ptr;
    p.z = (const float *)ctx->ztab.ptr;
    p.volume = ctx->volume;
    p.depth = (float *)ctx->depth.ptr;
    p.cost = (float *)ctx->cost.ptr;
    p.index = (int *)ctx->index.ptr;
    p.invW = 1.0f / (float)ctx->W;
    p.invH = 1.0f / (float)ctx->H;
    p.Wp = (float)ctx->W + 0.5f;
    p.Hp = (float)ctx->H + 0.5f;
    p.plan = (const uint2 *)ctx->plan.ptr;
    p.tiles_x = div_up(ctx->W, TILE_W);
    p.tile_h = tile_h;
    p.pc = pc;
    p.tiles_y = div_up(ctx->H, tile_h);
    p.nchunks = div_up(ctx->D, pc);
    p.chunk0 = 0;
    p.chunk1 = p.nchunks;
    p.ty0 = 0;
    p.tyn = p.tiles_y;
    p.row_begin = 0;
    p.row_end = ctx->H;
    p.plane_begin = 0;
    p.plane_end = ctx->D;
    p.cps = p.nchunks;
    p.part = nullptr;
    p.plan_stats = nullptr;
    p.view_slot = ctx->views_in_store ? (const int *)ctx->view_slots.ptr : nullptr;  // mvs_sweep_handles: view v is slot view_slot[v] of the frame store
    p.sep_r = nullptr;
    p.sep_y = nullptr;
    p.sep_dpad = 0;
    p.sep_reserved = 0;
    p.debug = 0;
    return MVS_OK;
}



// Planner counters: every thread of a launch used to hit the same two or three addresses with an atomic (0.2-0.3 ms of serialised L2
// atomics for a 0.01 ms kernel); one atomic per wavefront instead.  All 64 lanes must call (inactive contributions: the identity).
__device__ __forceinline__ int wave_max_i32(int v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = max(v, __shfl_xor(v, m, 64));
    return v;
}
__device__ __forceinline__ int wave_sum_i32(int v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

}  // namespace mvs
