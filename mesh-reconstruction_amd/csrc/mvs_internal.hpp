// mvs_internal.hpp -- context layout and helpers shared by the HIP translation units of libmvs_hip.so.
// Not part of the ABI (include/mvs.h is).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../include/mvs.h"
#include "hooks.hpp"

namespace mvs {

// Device buffer that only ever grows; freed with the context.
struct DevBuf {
    void *ptr = nullptr;
    size_t bytes = 0;
};

struct ProfileSlot {
    hipEvent_t start = nullptr, stop = nullptr;
    int kind = -1;
};

}  // namespace mvs

struct mvs_ctx {
    int device = 0;
    int W = 0, H = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;  // the stream work is queued on (own_stream unless mvs_set_stream)
    char err[512] = {0};
    char info[256] = {0};
    int num_cus = 0;
    mvs::Hooks hooks;                // environment switches as read by mvs_create (all defaults unless MVS_TEST_HOOKS=1: hooks.hpp)
    bool plan_cache = true;          // mvs_sweep_set_plan_cache

    // ---- sweep state (HBM resident) -------------------------------------------------------------
    // main image: H*W u8 tight.  side images: V slabs of (H+2) rows x pad_pitch bytes, 1-pixel
    // GL_REPEAT wrap padding so the sampler never needs wrap logic.
    mvs::DevBuf main_img, side_pads, qmats, ztab, plan, upload;
    mvs::DevBuf volume_own;          // packed (cnt<<16 | sum) cells, [D][H][W] u32
    uint32_t *volume = nullptr;      // volume_own.ptr or caller memory (mvs_sweep_use_volume)
    size_t volume_bytes = 0;
    bool volume_external = false;
    mvs::DevBuf depth, cost, index;  // H*W each
    int pad_pitch = 0;
    size_t pad_slab = 0;             // bytes per padded side image
    int V = 0, D = 0;
    float main_cam[16] = {0};
    bool have_main = false, have_views = false, have_planes = false;
    bool plan_valid = false;         // region plan matches current (views, planes)
    int sampler = MVS_SAMPLER_FIXED;  // arithmetic contract of the sweep's texture fetch (mvs_sweep_set_sampler)
    mvs::DevBuf side_quads;          // fixed sampler: quad image of every padded side view (4 bytes per texel), built by mvs_sweep_set_views
    mvs::DevBuf side_quads16;        // exact sampler: f16 quad image (8 bytes per texel), built on demand by ensure_quads16
    bool quads16_valid = false;
    bool pads_valid = false;         // side_pads (wrap-padded u8 frames) matches the current quad images: rebuilt on demand by ensure_pads
    mvs::DevBuf frame_ptrs;          // mvs_sweep_set_views_device: device table of the caller's frame pointers
    std::vector<const uint8_t *> frame_ptrs_host;
    // mvs_sweep_handles: the current main / side views are slots of the frame store (no copies: the kernels read the store)
    bool views_in_store = false;
    int main_store_slot = -1;
    mvs::DevBuf view_slots;          // device: slot of view v
    std::vector<int> view_slots_host;
    mvs::DevBuf fx_lut;              // fixed sampler: 32 x 32 table of packed 8-bit bilinear weights
    int plan_shape = 2;              // what the plan was made for: 1 = exact sampler, 2 px x 32 planes; 2 = exact, 4 px x 16 planes; 3 = fixed sampler
    bool plan_forced = false;        // plan made with the 4 x 16 shape forced (timing experiments)
    mvs::DevBuf plan_stats;          // planner counters (oversize regions, regions not skipped, widest / tallest staged region)
    // rectified fast path of the fixed sampler (sweep_rect.hip): per (tile column | tile row, view, plane) texel + phase + certificates
    mvs::DevBuf rect_tab;
    bool rect_ok = false;            // the current plan can be served by sweep_fx_rect
    // rectified path of the exact sampler (sweep_xrect.hip): box tables, LDS row stride and slot size of the current plan
    mvs::DevBuf xrect_tab;
    bool xrect_ok = false;
    int xrect_rs = 0, xrect_slot_bytes = 0;
    bool exact_tiled_planned = false;  // sweep_tiled's region plan exists for the current (views, planes) (made on demand when xrect_ok)
    int exact_last_shape = 0;          // what served the exact sampler's last run: 1 / 2 (sweep_tiled's thread shapes) or 5 (sweep_exact_rect)
    // what the fixed sampler's plan in memory (rectified tables, general plan) was made for: a new set of views with the SAME view matrices,
    // planes and slots -- a fixed camera rig delivering its next frames -- reuses it instead of planning again
    bool snap_valid = false, snap_in_store = false;
    std::vector<float> snap_q, snap_z;
    std::vector<int> snap_slots;
    mvs::DevBuf sep_tab;             // the separable path's tables for the current (views, planes): sweep_fx.hip, plan_sep_tables
    bool sep_ok = false;
    int sep_dpad = 0;
    size_t sep_r_offset = 0;
    bool fx_general_planned = false; // the general tiled kernel's plan exists for the current (views, planes) (made on demand when rect_ok)
    int rect_rs = 0, rect_slot_dw = 0, rect_dpad = 0;
    std::vector<unsigned char> rect_cold_host;  // host copy of the kernel's cold block (sweep_rect.hip: RectCold)
    bool rect_cold_sent = false;
    hipEvent_t plan_event = nullptr;  // the rectified planner's read-back has landed (the host waits for this, not for the whole stream)
    mvs::DevBuf probe_buf;           // mvs_depth_probe: pixel coordinates in, depths out
    mvs::DevBuf raster_bins;         // face binning of large meshes: per-bin counts / offsets / lists, shared list of large faces
    mvs::DevBuf filter_sort;         // mvs_filter_points, dense clouds: keys and a second copy of the upper lists for the global sorts
    double *filter_pinned = nullptr; // mvs_filter_points: pinned host slots for the convergence value of two iterations in flight
    hipEvent_t filter_ev[2] = {nullptr, nullptr};
    std::vector<float> q_host;       // V*12
    std::vector<float> z_host;       // D

    // ---- renderer state -----------------------------------------------------------------------------
    mvs::DevBuf soup;                // 9 floats per face, dehomogenised triangle soup
    int nfaces = 0;
    mvs::DevBuf r_zbuf, r_shadow, r_frame, r_out3, r_tmp0, r_tmp1, r_tmp2, r_mips;
    mvs::DevBuf r_tris_main;         // Render::projected's main pass: the main camera's triangle records (raster.hip: projected_main_pass)
    // Render::projected's frame textures as raster.hip last made them (projected_textures): bytes per frame in r_frame / r_mips, how many frames were
    // prepared up front (mvs_process_frame: all side views at once), and the mip-chain description of one frame (raster.hip's MipArgs, opaque here)
    size_t tex_frame_bytes = 0, tex_mips_bytes = 0;
    int tex_prepared = 0;
    unsigned char tex_mip[256] = {0};
    int texture_filter = MVS_FILTER_MIPMAP;  // Render::projected's frame texture: mip chain + trilinear (the reference's request) or level 0 only
    mvs::DevBuf cubic_tab;           // Q15 bicubic weights for remap (32*32*16 shorts)
    mvs::DevBuf flow_arena;          // optical-flow pyramids and work buffers
    hipGraphExec_t flow_graph[2] = {nullptr, nullptr};  // test hook MVS_FLOW_GRAPH only (tools/graph_repro.py): calculateFlow per algorithm as a replayed graph
    void *flow_graph_arena[2] = {nullptr, nullptr};
    mvs::DevBuf flow_batch_arena;    // the same for the batched Farneback of mvs_process_frame (all side views of a main frame per launch)
    mvs::DevBuf frame_buf;           // mvs_process_frame: frames, depth, warped image, flows of one main frame
    mvs::DevBuf best_parts;          // plane-split sweeps: partial (best cell, best index) per split and pixel
    // frame store (mvs_frame_store / mvs_frame_upload / mvs_sweep_batch): the frames of a sequence, uploaded once, each as raw frame,
    // quad image (5 bytes per pixel); main and side views of the batched sweep are slots of it
    mvs::DevBuf store_raw, store_quads;
    // mvs_sweep_batch_async: two batches in flight, each with its own device block and host staging; the results of a batch travel on
    // the copy stream while the next batch is planned and swept on the context's stream
    struct BatchSlot {
        mvs::DevBuf buf;
        std::vector<char> host;
        hipEvent_t swept = nullptr, landed = nullptr;  // kernel done (copy stream waits for it) / results in the caller's memory
        bool busy = false;
    } batch_slot[2];
    int batch_next = 0;
    hipStream_t copy_stream = nullptr;
    int onecall_bands_last = 0;           // row bands of the last mvs_sweep (test hook mvs_test_onecall_bands)
    std::vector<hipEvent_t> band_events;  // mvs_sweep's band pipeline: rows uploaded / band swept, per band
    int store_cap = 0;
    std::vector<unsigned char> store_have;
    // mvs_process_frame runs the flows of one main frame's side views concurrently, one lane each:
    // A lane: the stream one side view's flow runs on, a SHADOW context that stands for this context on that stream (device, size, hooks; its own flow
    // arena; nothing else is used through it) and a host thread that queues the flow's launches while the calling thread goes on with the next side
    // view (pipeline.hip: a main frame is ~150 launches at ~6 us of host time each, the calling thread alone was the bottleneck).
    struct FlowLane {
        hipStream_t stream = nullptr;
        mvs_ctx *shadow = nullptr;
        void *worker = nullptr;   // pipeline.hip: LaneWorker
    };
    static constexpr int kFlowLanes = 4;
    FlowLane lanes[kFlowLanes];
    std::vector<hipEvent_t> lane_events;  // 2 per side view: inputs ready, flow done

    // ---- profiling -----------------------------------------------------------------------------------
    bool profiling = false;
    std::vector<mvs::ProfileSlot> slots;
    size_t slots_used = 0;
};

namespace mvs {

int fail(mvs_ctx *ctx, int code, const char *fmt, ...);
void set_global_error(const char *msg);

#define MVS_HIP(ctx, call)                                                                         \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return mvs::fail((ctx), MVS_EHIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), \
                             __FILE__, __LINE__);                                                  \
    } while (0)

// grow-only device allocation
int ensure(mvs_ctx *ctx, DevBuf &b, size_t bytes);

// bracket a kernel launch with events when profiling is on
struct ProfileScope {
    mvs_ctx *ctx;
    int slot;
    ProfileScope(mvs_ctx *c, int kind);
    ~ProfileScope();
};

// host camera math (double precision; order of operations is part of the parity contract)
void invert4(const double m[16], double out[16]);
void view_matrix(const float main_cam[16], const float side_cam[16], int W, int H, float Q[12]);
void plane_table(int D, float z_lo, float z_hi, float *z);

inline int div_up(int a, int b) { return (a + b - 1) / b; }

// the sweep's input setters with the final synchronisation optional (context.hip; mvs_sweep queues all of them and waits once)
int sweep_set_main_impl(mvs_ctx *ctx, const float main_cam[16], const uint8_t *main_hw, bool sync, bool device = false);
int sweep_set_views_impl(mvs_ctx *ctx, int nviews, const float *side_cams, const uint8_t *const *side_frames, bool sync, bool defer_frames = false, bool device = false);
int sweep_upload_frames_impl(mvs_ctx *ctx, const uint8_t *const *side_frames, bool device = false);
struct PlanHook {  // work a caller wants queued between the planner's launches and the planner's one host read-back (it then runs while the host waits)
    virtual int run() = 0;
    virtual ~PlanHook() = default;
};
int sweep_fx_plan(mvs_ctx *ctx, PlanHook *between = nullptr);  // sweep_fx.hip: the fixed sampler's region plan (rectified tables or the general plan)
int ensure_pads(mvs_ctx *ctx);     // wrap-padded u8 frames of the current side views, rebuilt from the quad images when a path needs them
int sweep_upload_rows_impl(mvs_ctx *ctx, const uint8_t *const *side_frames, int r0, int r1, hipStream_t s);  // context.hip: band pipeline of mvs_sweep
int sweep_build_quads_impl(mvs_ctx *ctx, int q0, int q1);
int sweep_rect_plan(mvs_ctx *ctx, PlanHook *between = nullptr);   // sweep_rect.hip: tables + eligibility of the rectified kernel for the current fixed-sampler plan
struct SweepParams;
int ensure_quads16(mvs_ctx *ctx);  // exact sampler's f16 quad image of the current side views  // the deferred half of sweep_set_views_impl
int sweep_set_planes_impl(mvs_ctx *ctx, int nplanes, float z_lo, float z_hi, bool sync);

// device-buffer forms used by the flow stage (photometric.hip)
int compare_device(mvs_ctx *ctx, const uint8_t *prev8, const uint8_t *next8, float *out);
int remap_device(mvs_ctx *ctx, const float *flow, int stride, const uint8_t *img, uint8_t *out);
// the same for B pairs per launch (photometric.hip): compare(prev, next_i), flowRemap(flow_i, img_i)
int compare_batch_device(mvs_ctx *ctx, const uint8_t *prev8, const uint8_t *next8, int B, float *out);
int remap_batch_device(mvs_ctx *ctx, const float *flow, int stride, ptrdiff_t flow_z, const uint8_t *img, int B, uint8_t *out);
int ensure_cubic_table(mvs_ctx *ctx);
// raster.hip / flow.hip / triangulate.hip on device buffers (pipeline.hip strings them together)
int depth_device(mvs_ctx *ctx, const float cam[16], float *out_dev);
int projected_device(mvs_ctx *ctx, const float cam[16], const uint8_t *frame_dev, const float projector[16], uint8_t *out3_dev);
int projected_main_pass(mvs_ctx *ctx, const float cam[16]);   // the half of projected() that does not depend on the side view ...
int projected_side_pass(mvs_ctx *ctx, const uint8_t *frame_dev, const float projector[16], uint8_t *out3_dev, int prepared_view = -1, const uint8_t *mix_bg = nullptr,
                        float *mix_depth = nullptr, uint8_t *mix_out = nullptr);   // ... and the half that does
int projected_prepare_views(mvs_ctx *ctx, const uint8_t *const *frames_dev, int nframes);   // the frame textures (wrap padding + mip chain) of nframes (<= 32) side frames, a device pointer each, per launch
int mix_background_device(mvs_ctx *ctx, const uint8_t *img3_dev, const uint8_t *bg_dev, float *depth_dev, uint8_t *out_dev);
int flow_device(mvs_ctx *ctx, const uint8_t *prev_dev, const uint8_t *next_dev, int use_farneback, float *out4_dev);
int flow_only_device(mvs_ctx *ctx, const uint8_t *prev_dev, const uint8_t *next_dev, int use_farneback, float *flow2_dev);  // without the variance channel ...
int flow_variance_batch_device(mvs_ctx *ctx, const uint8_t *prev8, const uint8_t *next8, const float *flow2, int B, uint8_t *r8, float *var, float *out4);  // ... which this adds for B flows per launch
int flow_farneback_batch_device(mvs_ctx *ctx, const uint8_t *prev_dev, const uint8_t *next_dev, int B, float *out4_dev);  // next: B frames, W*H bytes apart
int triangulate_impl(mvs_ctx *ctx, int nviews, const float *const *flows, bool on_device, const float main_cam[16],
                     const float *side_cams, const float *depth, float *out_points7, int *out_count);
void lanes_shutdown(mvs_ctx *ctx);   // pipeline.hip: joins the lane threads, frees the shadows' arenas (mvs_destroy)
int compare_prepare(mvs_ctx *ctx, int pairs = 1);  // allocates compare_device's arena (for `pairs` image pairs per launch: compare_batch_device)

}  // namespace mvs
