/*
 * mvs.h -- C ABI of the MI355X-native dense-MVS depth engine (libmvs_hip.so).
 *
 * This is the drop-in boundary for the hot path of addam/mesh-reconstruction: every entry point
 * replaces one call the reference's driver makes through its link-time renderer seam
 * (recon.hpp:93-100, `class Render` + `spawnRender`; Makefile:2,16,21 `render_${SYSTEM_OPENGL}.cpp`)
 * or through the free functions of recon.hpp:40-50.  Plain pointers and sizes only; no C++ or
 * torch types.  The C++ shim that turns these into `class RenderHIP : public Render`,
 * `spawnRender` and `calculateFlow` lives in mesh-reconstruction_amd/host/ (see INTEGRATION.md).
 *
 * Conventions:
 *   - camera matrices: 4x4 float, row-major, applied as P*(x,y,z,1)^T       (render_glx.cpp:265,338,375)
 *   - images: top-down, row-major, tightly packed                           (cv::flip at render_glx.cpp:365,392)
 *   - depth maps: NDC z in [-1,1], 1.0f == backgroundDepth == "no geometry" (recon.hpp:30)
 *   - every function returns 0 on success, a negative MVS_E* code on error; the message is
 *     available from mvs_last_error().  Nothing aborts, asserts or exits (the reference does:
 *     recon.cpp:49, util.cpp:442).
 *   - a context is bound to one GPU; calls on one context must be serialised by the caller
 *     (the reference is single-threaded: one GL context, render_glx.cpp:152-208).
 *   - the library never retains caller pointers after a call returns.
 *   - there is NO CPU fallback: without a usable HIP device mvs_create() fails.
 */
#ifndef MVS_H
#define MVS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MVS_OK 0
#define MVS_EINVAL (-1)   /* bad argument */
#define MVS_EHIP (-2)     /* HIP runtime error */
#define MVS_ESTATE (-3)   /* call sequence error (e.g. sweep run before inputs were set) */
#define MVS_ENOMEM (-4)   /* device or host allocation failed */

#define MVS_BACKGROUND_DEPTH 1.0f /* recon.hpp:30 */

typedef struct mvs_ctx mvs_ctx;

/* ---- life cycle: replaces RenderGLX::RenderGLX / ~RenderGLX (render_glx.cpp:152-227) -------- */
mvs_ctx *mvs_create(int device, int width, int height);
void mvs_destroy(mvs_ctx *ctx);
/* last error text of `ctx`; with ctx == NULL the last error of a failed mvs_create() */
const char *mvs_last_error(const mvs_ctx *ctx);
/* run all subsequent work of `ctx` on the caller's hipStream_t (NULL = the context's own stream) */
int mvs_set_stream(mvs_ctx *ctx, void *hip_stream);
/* page-locked host memory for frames / results handed to the entry points below (optional: any host pointer works; from pinned
 * memory the one-call entries upload at the full PCIe rate and overlap with the device-side preparation).  NULL on failure. */
void *mvs_host_alloc(size_t bytes);
void mvs_host_free(void *p);
/* block until all queued work of the context has finished */
int mvs_synchronize(mvs_ctx *ctx);
int mvs_width(const mvs_ctx *ctx);
int mvs_height(const mvs_ctx *ctx);

/* Render::depth restricted to the pixels a caller actually reads: Heuristic::chooseCameras renders 200 face cameras per
 * outer iteration (heuristic.cpp:445-456) and filterCameras looks ONE pixel up per real camera in each map
 * (heuristic.cpp:307-312).  Rasterises the depth map for `cam` on the device and returns out[i] = depth[rows[i]][cols[i]]
 * (same values as mvs_depth, without moving the map across PCIe).  Pixels outside the map are an error. */
int mvs_depth_probe(mvs_ctx *ctx, const float cam[16], int n, const int32_t *rows, const int32_t *cols, float *out);

/* ---- renderer: replaces class Render (recon.hpp:93-99) ---------------------------------------- */
/* Render::loadMesh, render_glx.cpp:230-258: verts4 = nverts homogeneous rows (x,y,z,w), faces3 = int32 triples */
int mvs_load_mesh(mvs_ctx *ctx, const float *verts4, int nverts, const int32_t *faces3, int nfaces);
/* Render::depth, render_glx.cpp:369-397: out_hw = H*W float NDC z, empty pixels 1.0 */
int mvs_depth(mvs_ctx *ctx, const float cam[16], float *out_hw);
/* Render::projected, render_glx.cpp:261-367: out_hw3 = H*W*3 u8 (warped side intensity, mask, mask) */
int mvs_projected(mvs_ctx *ctx, const float cam[16], const uint8_t *frame_hw, const float projector[16],
                  uint8_t *out_hw3);

/* ---- photometric helpers: replace util.cpp:332-403 -------------------------------------------- */
/* mixBackground, util.cpp:366-387: depth_hw_inout is mutated (masked pixels := 1.0) */
int mvs_mix_background(mvs_ctx *ctx, const uint8_t *img_hw3, const uint8_t *bg_hw, float *depth_hw_inout,
                       uint8_t *out_hw);
/* compare, util.cpp:332-361: multi-scale L1 pyramid difference, out_hw = H*W float */
int mvs_compare(mvs_ctx *ctx, const uint8_t *prev_hw, const uint8_t *next_hw, float *out_hw);
/* flowRemap, util.cpp:390-403: bicubic remap by flow (flow_stride floats per pixel, first two used) */
int mvs_flow_remap(mvs_ctx *ctx, const float *flow, int flow_stride, const uint8_t *image_hw, uint8_t *out_hw);
/* calculateFlow, flow.cpp:19-42: out_hw4 = H*W*4 float (u, v, variance, 0) */
int mvs_flow(mvs_ctx *ctx, const uint8_t *prev_hw, const uint8_t *next_hw, int use_farneback, float *out_hw4);

/* ---- per-pixel triangulation + normals: replaces triangulatePixels (util.cpp:167-329, recon.cpp:114) -------------- */
/* flows_hw4: nviews pointers to H*W*4 floats (u, v, variance, 0) as mvs_flow writes them; depth_hw: the (mixBackground-
 * updated) depth map; out_points7: room for H*W rows of (x, y, z, w, nx, ny, nz); *out_count = rows written, in pixel
 * scan order like the reference's pixelId. */
int mvs_triangulate(mvs_ctx *ctx, int nviews, const float *const *flows_hw4, const float main_cam[16],
                    const float *side_cams /* nviews*16 */, const float *depth_hw, float *out_points7, int *out_count);

/* ---- one main frame, device resident: the body of the loop recon.cpp:65-117 ------------------------------------------ */
/* depth(main) -> for every side view: projected -> mixBackground (depth mutated cumulatively) -> calculateFlow ->
 * triangulatePixels, with every intermediate kept in HBM: frames go up once, only the points come back.  Bit-identical
 * to calling mvs_depth / mvs_projected / mvs_mix_background / mvs_flow / mvs_triangulate in that order.
 * out_points7 needs room for H*W rows; depth_after_hw (nullable) receives the depth map after the last mixBackground. */
int mvs_process_frame(mvs_ctx *ctx, const float main_cam[16], const uint8_t *main_frame_hw, int nside,
                      const float *side_cams /* nside*16 */, const uint8_t *const *side_frames_hw, int use_farneback,
                      float *out_points7, int *out_count, float *depth_after_hw);
/* The same with the frames taken from the context's frame store (mvs_frame_store / mvs_frame_upload, below): recon.cpp:65-117 reads every frame of a
 * sequence about five times -- once as a main frame (configuration.cpp frame(i), recon.cpp:68), four times as a side view (:80) -- and through the
 * store it crosses PCIe once.  main_slot / side_slots name filled slots (MVS_ESTATE otherwise); results are those of mvs_process_frame on the
 * same frames, bit for bit; stream-ordered behind the uploads that filled the slots. */
int mvs_process_frame_slots(mvs_ctx *ctx, const float main_cam[16], int main_slot, int nside, const float *side_cams /* nside*16 */,
                            const int *side_slots, int use_farneback, float *out_points7, int *out_count, float *depth_after_hw);

/* cv::resize(frame, Size(dw, dh)) as Configuration applies it to every decoded frame when -s / the clip size asks for it
 * (configuration.cpp:233: INTER_LINEAR -- the CV_INTER_AREA in that call lands in the ignored fx argument): OpenCV's fixed-point
 * bilinear resize on u8, 1 or 3 interleaved channels, host buffers in and out. */
int mvs_resize_u8(mvs_ctx *ctx, const uint8_t *src, int src_w, int src_h, int channels, uint8_t *dst, int dst_w, int dst_h);

/* Texture filter of Render::projected's frame texture.  The reference uploads the side frame with glGenerateMipmap and samples it
 * with GL_LINEAR_MIPMAP_LINEAR (render_glx.cpp:83-85): MVS_FILTER_MIPMAP (default) builds the mip chain (2 x 2 box, u8 levels) per
 * frame and blends the two levels the pixel's footprint calls for (fine derivatives on the 2 x 2 pixel quad, isotropic; DESIGN.md
 * section 5 states the arithmetic, the oracle restates it); under magnification and at 1 : 1 it IS level-0 bilinear.
 * MVS_FILTER_LEVEL0 samples level 0 only (rounds 1-2 of this library; the sweep's samplers always do). */
#define MVS_FILTER_MIPMAP 0
#define MVS_FILTER_LEVEL0 1
int mvs_set_texture_filter(mvs_ctx *ctx, int filter);
int mvs_texture_filter(const mvs_ctx *ctx);

/* ---- point-cloud filter: replaces Heuristic::filterPoints (heuristic.cpp:55-176, recon.cpp:125) ------------------- */
/* points4: npoints homogeneous rows; alpha = the reference's alphaVals.back() (radius = alpha/4, compared with squared
 * distances as the reference does).  keep_out receives the ascending indices of the retained points, *out_count how
 * many; the caller compacts points and normals with them (heuristic.cpp:166-175). */
int mvs_filter_points(mvs_ctx *ctx, const float *points4, int npoints, float alpha, int32_t *keep_out, int *out_count);

/* ---- plane sweep: the D-plane generalisation of shader.frag:11-25 ------ */
/*
 * One-call form on host buffers.  For every pixel of the main view and every plane
 * z_d = z_lo + (z_hi - z_lo)(d + 1/2)/nplanes (main-camera NDC z) the side frames are warped into the
 * main view exactly as shader.frag does for the mesh position, quantised to u8 like the RGB8
 * read-back (render_glx.cpp:359), and the cost is the mean over in-frame views of |I_main - I_warp|.
 * depth_hw: H*W float, NDC z of the lowest-cost plane (ties -> lowest d), 1.0 where no view is in frame.
 * cost_hw (nullable): H*W float best cost.  volume_dhw (nullable): nplanes*H*W float normalised
 * cost, +inf where no view is in frame.
 */
int mvs_sweep(mvs_ctx *ctx, const float main_cam[16], const uint8_t *main_hw, int nviews,
              const float *side_cams /* nviews*16 */, const uint8_t *const *side_frames, int nplanes,
              float z_lo, float z_hi, float *depth_hw, float *cost_hw, float *volume_dhw);

/* The sweep's sampler at ONE plane per pixel, z = depth_hw[p] (e.g. the output of mvs_depth): out_hw2 = H*W pairs
 * (warped u8 intensity, mask 255/0); background pixels and out-of-frame samples give (0, 0).  With the renderer's depth
 * map this reproduces Render::projected (render_glx.cpp:261-367) minus its shadow test -- the parity hook between the
 * D-plane sweep and the reference's single-hypothesis warp. */
int mvs_warp_by_depth(mvs_ctx *ctx, const float main_cam[16], const float *depth_hw, const float side_cam[16],
                      const uint8_t *frame_hw, uint8_t *out_hw2);

/* Texture-fetch arithmetic of the sweep.  The reference leaves this to the OpenGL driver (GL_LINEAR on a
 * GL_RED8 texture, render_glx.cpp:75-85, read back as RGB8, :359), which pins neither sub-texel precision nor weight width.
 *   MVS_SAMPLER_FIXED (default): positions quantised to 1/32 texel and 8-bit weights from a 32 x 32 table -- the model of
 *     fixed-function samplers and of OpenCV's fixed-point remap (INTER_BITS = 5, the path of the reference's flowRemap,
 *     util.cpp:401); the warped intensity keeps the table's precision and a packed cell is count << 24 | sum |w.t - 255 I_main|
 *     (sums in 1/255 grey levels; at most 255 views; images of up to 16383 x 16383).  About 30 % fewer instructions per sample on gfx950.
 *   MVS_SAMPLER_EXACT_F32: bilinear interpolation in f32 with one rounding per operation, result rounded to u8 like the RGB8
 *     read-back; a packed cell is count << 16 | sum |u8 - I_main| (at most 257 views).
 * Both are restated bit for bit by the CPU oracle; they select the same plane except where two planes' costs are within the
 * quantisation step (measured: DESIGN.md).  Changing the sampler invalidates the region plan, not the uploaded inputs. */
#define MVS_SAMPLER_FIXED 0
#define MVS_SAMPLER_EXACT_F32 1
int mvs_sweep_set_sampler(mvs_ctx *ctx, int sampler);
int mvs_sweep_sampler(const mvs_ctx *ctx);

/* Staged form: inputs stay resident in HBM between runs (bench, multi-GPU view sharding). */
int mvs_sweep_set_main(mvs_ctx *ctx, const float main_cam[16], const uint8_t *main_hw);
int mvs_sweep_set_views(mvs_ctx *ctx, int nviews, const float *side_cams, const uint8_t *const *side_frames);
int mvs_sweep_set_planes(mvs_ctx *ctx, int nplanes, float z_lo, float z_hi);
/* The same with frames that already live in the memory of the context's GPU (a decoder's output, a previous stage's result; no counterpart
 * in the reference, whose frames are host cv::Mat: Configuration::frame, configuration.cpp:437-440): H*W u8,
 * tightly packed.  Stream-ordered like a kernel launch: nothing crosses PCIe, nothing synchronises, and the frames are read by work
 * queued on the context's stream -- they must stay unchanged until that work has run (mvs_synchronize, or the caller's own stream
 * order when mvs_set_stream shares a stream).  The side views' quad images are written straight from the raw frames in one pass.
 * (Any of the view setters: when the new set has the view matrices and planes of the last planned one -- a fixed camera rig
 * delivering its next frames -- the region plan in memory is reused; it depends on the cameras, not on the frames.) */
int mvs_sweep_set_main_device(mvs_ctx *ctx, const float main_cam[16], const void *main_dev);
int mvs_sweep_set_views_device(mvs_ctx *ctx, int nviews, const float *side_cams, const void *const *side_frames_dev);

#define MVS_SWEEP_VOLUME 1u       /* materialise the packed cost volume in HBM */
#define MVS_SWEEP_FUSED_ARGMIN 2u /* select depth inside the sweep kernel (no volume read-back pass) */
#define MVS_SWEEP_FORCE_GENERIC 4u /* use the un-tiled global-gather kernel (test / fallback path) */
#define MVS_SWEEP_NO_RECT 8u       /* never take the rectified-view kernels (sweep_fx_rect / sweep_exact_rect); results are bit-identical either way */
/* accumulate views [view_first, view_first + view_count) into the packed volume / fused outputs (async) */
int mvs_sweep_run(mvs_ctx *ctx, int view_first, int view_count, unsigned flags);
/* the same for planes [plane_first, plane_first + plane_count) only (MVS_SWEEP_VOLUME; boundaries on multiples of
 * mvs_sweep_plane_granularity(), the last group may end at nplanes): lets a view-sharded job all-reduce one plane group
 * over xGMI while the next one is being swept (bench.py --shard views --plane-groups G) */
int mvs_sweep_run_planes(mvs_ctx *ctx, int view_first, int view_count, int plane_first, int plane_count, unsigned flags);
int mvs_sweep_plane_granularity(void);
/* the same for pixel rows [row_first, row_first + row_count) only, all planes (any flags; boundaries on multiples of
 * mvs_sweep_row_granularity(), the last band may end at the image height): rows are independent (SURVEY 8e sharding 2),
 * so G ranks sweep one band each of the SAME main view and exchange only the depth rows (bench.py --shard rows).
 * Volume cells and depth / cost / index outside the band are left untouched. */
int mvs_sweep_run_rows(mvs_ctx *ctx, int view_first, int view_count, int row_first, int row_count, unsigned flags);
int mvs_sweep_row_granularity(void);                    /* valid for either sampler (16) */
int mvs_sweep_row_granularity_of(const mvs_ctx *ctx);   /* what the context's current sampler needs (fixed sampler: 8): finer bands balance better */
/* diagnostic: thread shape the region planner chose for the current (views, planes): 0 = no plan yet,
 * 1 = exact sampler, 2 pixels x 32 planes per thread (64x8-pixel tiles), 2 = exact sampler, 4 pixels x 16 planes (64x16 tiles:
 * bit-identical to 1; the choice follows how many warped 32-plane footprints fit the LDS staging buffer), 3 = fixed sampler
 * (2 pixels x 16 planes, 64x8-pixel tiles), 4 = fixed sampler, every side view rectified against the main view (pure translation
 * in its focal plane, equal intrinsics): the kernel sweep_fx_rect (8 pixels x 4 planes per thread, wave-uniform sampling
 * positions; bit-identical to 3, which MVS_SWEEP_NO_RECT selects), 5 = exact sampler on rectified views: the kernel sweep_exact_rect
 * (same thread shape; projection, reciprocal, trunc / fract and frame tests once per (column, plane, view) and per (row, plane, view);
 * bit-identical to 1 / 2).  For the exact sampler the value names what served the LAST run. */
int mvs_sweep_plan_shape(const mvs_ctx *ctx);
/* diagnostic: the fixed sampler reuses its region plan when a new view set has the view matrices, planes and slots of the last planned
 * one (a fixed rig's next frames; the plan depends on the cameras, not on the frames).  enable = 0 makes every view set plan again
 * (bench.py's cold step times the planner that way); results are identical either way.  Default: enabled. */
int mvs_sweep_set_plan_cache(mvs_ctx *ctx, int enable);
/* per-pixel depth selection over the packed volume (async); valid after MVS_SWEEP_VOLUME runs or
 * after the caller has reduced the volume across ranks in place */
int mvs_sweep_argmin(mvs_ctx *ctx);
/* Sub-plane refinement (optional): replaces the selected plane's depth in the depth map by the vertex of
 * the parabola through the mean costs of that plane and its two neighbours (clamped to half a plane step; planes at either end, or
 * with a neighbour that no view sees, keep their depth).  Needs the packed volume and a depth selection of the same run
 * (MVS_SWEEP_VOLUME | MVS_SWEEP_FUSED_ARGMIN, or MVS_SWEEP_VOLUME + mvs_sweep_argmin); best cost and index are unchanged.
 * The plane step is what makes two samplers that pick neighbouring planes of equal cost differ by 1/D in depth; refined depths of the
 * two samplers agree to a small fraction of it. */
int mvs_sweep_refine_depth(mvs_ctx *ctx);
/* The same selection in two steps, for a view-sharded job that REDUCE-SCATTERS the packed volume instead of all-reducing it
 * (half the bytes over xGMI): rank r owns the summed cells of planes [plane_first, plane_first + plane_count) in
 * `volume_slice_dev` ([plane_count][H][W] u32) and selects a partial best per pixel over them -- `partial_out_dev` receives
 * H*W records of 8 bytes (packed best cell, best absolute plane index; 0xffffffff = none).  The ranks all-gather their
 * records into [nparts][H*W] in ascending plane order and every rank merges them with mvs_sweep_combine_partials, a later part
 * winning only if strictly better (ties -> lowest plane, like mvs_sweep_argmin).  Both are asynchronous on the context's stream. */
int mvs_sweep_argmin_partial(mvs_ctx *ctx, const void *volume_slice_dev, int plane_first, int plane_count, void *partial_out_dev);
int mvs_sweep_combine_partials(mvs_ctx *ctx, const void *partials_dev, int nparts);
/* device pointer + size of the packed volume: nplanes*H*W uint32 cells (count << 24 | sum with the fixed sampler,
 * count << 16 | sum with the exact one); sums and counts add exactly, so an integer sum-all-reduce across view
 * shards is bit-identical to the single-GPU result */
void *mvs_sweep_volume_device(mvs_ctx *ctx, size_t *bytes);
/* make the sweep write into caller-owned device memory (e.g. a torch tensor handed to RCCL) */
int mvs_sweep_use_volume(mvs_ctx *ctx, void *device_ptr, size_t bytes);
/* device pointers of the per-pixel results (H*W each): depth f32, best cost f32, best index i32 */
void *mvs_sweep_depth_device(mvs_ctx *ctx);
void *mvs_sweep_cost_device(mvs_ctx *ctx);
void *mvs_sweep_index_device(mvs_ctx *ctx);
/* copy results to host (synchronises); any pointer may be NULL */
int mvs_sweep_fetch(mvs_ctx *ctx, float *depth_hw, float *cost_hw, int32_t *index_hw, uint32_t *packed_volume_dhw);
/* read back the f32 view matrices the sweep uses (nviews*12), for parity checks */
int mvs_sweep_view_matrices(mvs_ctx *ctx, float *q_out);

/* ---- a sequence on one GPU: frames uploaded once, several main frames per launch (recon.cpp:65-117) -------------------------
 * The reference sweeps every chosen main frame against a handful of neighbouring frames; through mvs_sweep each frame of the
 * sequence crosses PCIe once per main frame that uses it, and a 640 x 480 frame leaves most of the chip idle.  The frame store keeps
 * every frame of the sequence resident (raw and as quad image: 5 bytes per pixel); mvs_sweep_batch then sweeps nmain
 * main frames, each against its own nside side views -- all of them slots of the store -- in ONE launch of the general tiled kernel
 * (fixed sampler) and returns the nmain depth maps (and best costs) tightly packed.  Results are bit-identical to mvs_sweep on the
 * same frames and cameras.
 *   mvs_frame_store(ctx, capacity)   sizes the store (1..8191 frames); re-sizing empties it
 *   mvs_frame_upload(ctx, slot, f)   copies frame f (H*W u8) into a slot and prepares it (asynchronous; `f` must stay valid until the
 *                                    next synchronising call on the context, e.g. mvs_sweep_batch or mvs_synchronize)
 *   mvs_sweep_batch(...)             main_slots[nmain], main_cams[nmain*16], side_slots[nmain*nside], side_cams[nmain*nside*16];
 *                                    depth_out[nmain*H*W], cost_out nullable; synchronises */
int mvs_frame_store(mvs_ctx *ctx, int capacity);
int mvs_frame_upload(mvs_ctx *ctx, int slot, const uint8_t *frame_hw);
int mvs_frame_upload_device(mvs_ctx *ctx, int slot, const void *frame_dev); /* the frame is already on the context's GPU (stream-ordered) */
/* ONE main view over the frame store -- the resident-handle form of mvs_sweep for a caller that keeps its sequence on the device
 * (the loop of recon.cpp:65-117 through RenderHIP): main and side views are slots, nothing is uploaded, copied or
 * re-prepared (the quad images were built by mvs_frame_upload); the call pays the view matrices, the region plan, the sweep with depth
 * selection and the download of depth_hw (and cost_hw, nullable).  Bit-identical to mvs_sweep on the same frames and cameras; fixed
 * sampler; synchronises.  Afterwards the context's staged state (mvs_sweep_run, mvs_sweep_depth_device ...) refers to these views
 * until the next setter; re-uploading one of the slots in between changes the frame under it. */
int mvs_sweep_handles(mvs_ctx *ctx, int main_slot, const float main_cam[16], int nside, const int *side_slots, const float *side_cams /* nside*16 */,
                      int nplanes, float z_lo, float z_hi, float *depth_hw, float *cost_hw);
int mvs_sweep_batch(mvs_ctx *ctx, int nmain, const int *main_slots, const float *main_cams, int nside, const int *side_slots,
                    const float *side_cams, int nplanes, float z_lo, float z_hi, float *depth_out, float *cost_out);
/* The same without waiting: the batch is queued and the call returns; depth_out / cost_out (which must stay valid, and should be
 * page-locked: mvs_host_alloc) are complete when mvs_sweep_batch_wait returns.  Two batches can be in flight -- the results of one
 * cross PCIe on a second stream while the next is planned and swept, and the host prepares that next one meanwhile; a third call waits
 * for the oldest.  Frames a queued batch uses must not be re-uploaded before it has been waited for.  (mvs_sweep_batch = this + wait.) */
int mvs_sweep_batch_async(mvs_ctx *ctx, int nmain, const int *main_slots, const float *main_cams, int nside, const int *side_slots,
                          const float *side_cams, int nplanes, float z_lo, float z_hi, float *depth_out, float *cost_out);
int mvs_sweep_batch_wait(mvs_ctx *ctx);

/* ---- one main view on several GPUs of one node (SURVEY.md section 8b "multi-GPU", 8e, north_star) -----------------------------
 * A communicator owns one context per listed device and one RCCL communicator across them (librccl is loaded when the first
 * communicator is created; the library has no link dependency on it).  mvs_sweep_sharded runs ONE main view on all of them, one host
 * thread per GPU, and returns the depth map (and best cost) of the whole view -- bit-identical to mvs_sweep on one GPU in every mode
 * (cells are integers, sums are exact):
 *   MVS_SHARD_ROWS (default)  every GPU holds all side views and sweeps a band of the main view's pixel rows, depth selected inside
 *                             the kernel; each band is copied from its GPU into depth_hw / cost_hw.  4 bytes per pixel and map, no
 *                             collective: the split that scales.
 *   MVS_SHARD_VIEWS           the north_star's split: the side views are dealt to the GPUs, every GPU builds the packed volume of its
 *                             views, the volumes are all-reduced over xGMI per plane group (mvs_comm_set_plane_groups, default 4) on
 *                             a second stream while the next group is swept, depth is selected from the sum.  2 (n-1)/n x 4 P D
 *                             bytes per GPU on the links: cannot scale at the BASELINE sizes, built because
 *                             the north_star names it.
 *   MVS_SHARD_VIEWS_SCATTER   the same split, half the bytes, no overlap: reduce-scatter by plane slices + partial selection +
 *                             all-gather of 8-byte partials (nplanes a multiple of the GPU count; otherwise as MVS_SHARD_VIEWS).
 * A rank that fails before the exchange makes every rank skip it (host barrier + shared error flag); a failure inside the exchange
 * aborts all communicators (ncclCommAbort) so that no rank is left waiting; mvs_sweep_sharded then returns MVS_ESTATE until a new
 * communicator is created.  The sampler of a communicator's contexts is set through mvs_comm_context(). */
#define MVS_SHARD_ROWS 0
#define MVS_SHARD_VIEWS 1
#define MVS_SHARD_VIEWS_SCATTER 2
typedef struct mvs_comm mvs_comm;
mvs_comm *mvs_comm_create(const int *devices, int n, int width, int height); /* NULL on error: mvs_comm_last_error(NULL) */
void mvs_comm_destroy(mvs_comm *comm);
int mvs_comm_size(const mvs_comm *comm);
mvs_ctx *mvs_comm_context(mvs_comm *comm, int rank); /* borrowed; NULL if rank is out of range */
const char *mvs_comm_last_error(const mvs_comm *comm);
int mvs_comm_set_mode(mvs_comm *comm, int mode);   /* MVS_SHARD_* */
int mvs_comm_mode(const mvs_comm *comm);
int mvs_comm_set_plane_groups(mvs_comm *comm, int groups); /* MVS_SHARD_VIEWS: plane groups of the all-reduce pipeline, 1..64 */
int mvs_sweep_sharded(mvs_comm *comm, const float main_cam[16], const uint8_t *main_hw, int nviews, const float *side_cams /* nviews*16 */,
                      const uint8_t *const *side_frames, int nplanes, float z_lo, float z_hi, float *depth_hw, float *cost_hw /* nullable */);
/* The RESIDENT form (the staged mvs_sweep_set_* / mvs_sweep_run of one context, for a communicator): a sequence's caller -- the loop of
 * recon.cpp:65-117 over main frames -- uploads once and sweeps many times; nothing crosses PCIe, is re-planned or re-prepared between runs.
 *   mvs_comm_set_planes / _set_main / _set_views   upload to EVERY rank (each keeps all side views and the plan, so any mode can run on them;
 *                        host pointers are not retained: the calls return when the copies have landed).  Like a context, a new main view
 *                        invalidates the side views (their matrices depend on the main camera): set planes, then main, then views.
 *   mvs_comm_run(flags)  one sweep of what is resident, in the current mode; returns when every rank has finished.  The depth and best-cost
 *                        maps of the WHOLE view are left on rank 0's GPU (mvs_sweep_depth_device / _cost_device of mvs_comm_context(comm, 0);
 *                        not the index map): MVS_SHARD_ROWS copies each rank's band there by peer copies over xGMI behind the rank's sweep
 *                        (4 bytes per pixel and map in total, no collective, no host memory); the views modes end with the selection on
 *                        every rank.  flags: 0, or MVS_SWEEP_VOLUME to make MVS_SHARD_ROWS materialise each rank's band of the packed
 *                        volume as well (the views modes always build it).  The persistent rank threads do the work: no thread is created,
 *                        no frame uploaded and no plan made per call.
 *   mvs_comm_fetch       download rank 0's maps (either pointer may be NULL).
 * mvs_sweep_sharded = set_planes + set_main + set_views + run + fetch in one pass over the ranks (in a views mode each rank uploads only its
 * own views, and in rows mode the bands go straight to the caller's host maps); what it uploaded stays resident for mvs_comm_run in the same mode. */
int mvs_comm_set_planes(mvs_comm *comm, int nplanes, float z_lo, float z_hi);
int mvs_comm_set_main(mvs_comm *comm, const float main_cam[16], const uint8_t *main_hw);
int mvs_comm_set_views(mvs_comm *comm, int nviews, const float *side_cams /* nviews*16 */, const uint8_t *const *side_frames);
int mvs_comm_run(mvs_comm *comm, unsigned flags);
int mvs_comm_fetch(mvs_comm *comm, float *depth_hw /* nullable */, float *cost_hw /* nullable */);
/* Two calls in flight (the loop of recon.cpp:65-117 over main views, or a fixed rig's next frames: the gather of view k beside the sweep of
 * view k + 1).  mvs_comm_run_async(comm, 0) QUEUES one sweep of what is resident and returns when every rank's launches are queued -- nobody
 * waits for a GPU: in MVS_SHARD_ROWS each rank's band leaves its context's maps by a device copy behind the sweep and travels to rank 0 on a
 * second stream, into one of two (depth, cost) result pairs on rank 0's device, so the next call's sweep runs beside it; at most two calls are
 * in flight (a third is refused with MVS_ESTATE until mvs_comm_wait).  mvs_comm_wait(comm) waits for the OLDEST call in flight and makes its
 * maps the ones mvs_comm_fetch downloads; they stay valid until the second mvs_comm_run_async after that call.  The view-sharded modes end in
 * collectives that every rank thread drives: there mvs_comm_run_async completes the call before it returns (one in flight at most) and
 * mvs_comm_wait only publishes it.  While calls are in flight mvs_comm_run, mvs_comm_set_* and mvs_sweep_sharded return MVS_ESTATE.
 * Launch failures are returned by mvs_comm_run_async, failures on a GPU by mvs_comm_wait.  mvs_comm_pending: calls in flight (0..2). */
int mvs_comm_run_async(mvs_comm *comm, unsigned flags /* 0 */);
int mvs_comm_wait(mvs_comm *comm);
int mvs_comm_pending(const mvs_comm *comm);
/* Where a rank's band copies travel (the resident MVS_SHARD_ROWS gather: rank r -> rank 0).  mvs_comm_create enables peer access between
 * every rank's device and rank 0's in both directions (hipDeviceCanAccessPeer + hipDeviceEnablePeerAccess; "already enabled" accepted) and
 * keeps the outcome: mvs_comm_peer_access(comm, r) = 1 when rank r's copies go GPU to GPU (xGMI, or r shares rank 0's device; rank 0 itself:
 * 1), 0 when the runtime stages them through host memory -- nothing is refused, a staged band is slow, not wrong; right after creation
 * mvs_comm_last_error(comm) names the ranks concerned ("note: ...").  mvs_comm_device(comm, r): the HIP device ordinal of rank r. */
int mvs_comm_peer_access(const mvs_comm *comm, int rank);
int mvs_comm_device(const mvs_comm *comm, int rank);

/* ---- kernel timing (HIP events on the context's stream) ---------------------------------------- */
#define MVS_K_SWEEP 0
#define MVS_K_ARGMIN 1
#define MVS_K_PLAN 2
#define MVS_K_RASTER 3
#define MVS_K_PROJECT 4
#define MVS_K_FLOW 5
#define MVS_K_COUNT 8
int mvs_profile_enable(mvs_ctx *ctx, int on);
/* synchronises, then returns summed elapsed ms and launch count per kernel class since the last reset */
int mvs_profile_read(mvs_ctx *ctx, float ms_sum[MVS_K_COUNT], int launches[MVS_K_COUNT], int reset);

/* ---- surface meshing: replaces poissonSurface (recon.hpp:37, cgal_poisson.cpp:47-136, pcl.cpp:193-228) ----
 * Poisson reconstruction of oriented samples on a regular grid (csrc/poisson.hip): normals splatted with 64-bit fixed-point atomics,
 * laplace(chi) = div V solved with hipFFT, level set through the samples meshed by surface nets.  Context-free: runs on the calling
 * thread's current HIP device (device 0 unless the caller has set another), on a stream of its own; no CPU path.
 *   points     n rows x, y, z, w (homogeneous, as recon.cpp:121 hands them over);  normals  n rows nx, ny, nz (pointing out of the solid;
 *              a component that is NaN or beyond 1e4 in magnitude makes the sample vote for nothing).  Their lengths act as confidences,
 *              as given -- the reference's semantics on both of its backends (cgal_poisson.cpp:58-69; pcl.cpp:23, 198-202).  The wrappers
 *              with the reference's signature, poissonSurface (host/poisson.cpp) and mvs_amd.poisson_surface, normalise them first BY
 *              DEFAULT: a deliberate divergence (see there for what the pdf lengths of triangulatePixels do to this solver), switchable.
 *              The level is the median of chi over the samples (CGAL's Poisson_reconstruction_function: median value at the input points)
 *   grid_log2  log2 of the nodes per axis, 4..9; 0 = from the samples' average 6-nearest-neighbour spacing, the reference's own
 *              yardstick (CGAL::compute_average_spacing(points, 6), cgal_poisson.cpp:77): the coarsest of 32..512 nodes per axis whose
 *              node spacing is at most 0.75 x that spacing, which keeps the surface within the reference's approximation bound of
 *              0.375 x average spacing (cgal_poisson.cpp:52) on the analytic test surfaces; mvs_surface_spacing reports the average
 *              spacing, the node spacing and whether the ratio could be kept (512 nodes per axis may still be too coarse)
 *   smooth_cells  standard deviation (grid cells) of the Gaussian low-pass applied to chi; 1.0 is a good default
 *   keep_fields  non-zero: keep chi and the splatted integer fields for mvs_surface_grid (tests)
 * The alpha shape of the first iteration (alphaShapeFaces, recon.hpp:33-34) is host-only code: libmvs_host.so, host/alpha_shapes.cpp.
 *   support_spacings (mvs_poisson_surface_ex)  the level set is meshed only in cells within this many average spacings (rounded up to whole
 *              nodes, Chebyshev distance) of a grid node that collected sample weight; 0 = everywhere.  Away from the samples the solved
 *              field is flat and hovers around the level: on open or noisy clouds (what recon.cpp:121 collects from the flows) the level
 *              set there is a closing sheet no sample supports plus numerical fuzz -- 97 % of 16 M vertices on the config-5 cloud of
 *              tests/test_meshing_gpu.py.  The reference's mesher returns such sheets too (coarse ones: its Delaunay refinement has no
 *              resolution where there are no points); this grid would return them at full resolution, so it does not return them at
 *              all.  Closed, evenly sampled surfaces are not affected (every cell of the surface is next to a sample).
 *              mvs_poisson_surface uses MVS_POISSON_SUPPORT_DEFAULT. */
#define MVS_POISSON_SUPPORT_DEFAULT 8.0f
typedef struct mvs_surface mvs_surface;
/* The first hipFFT plan of a grid size in a process costs about 2 s (rocFFT compiles its kernels for the size at run time); every
 * later call of that size finds the plans cached.  mvs_poisson_warmup(grid_log2) pays that cost when the caller chooses -- at start-up,
 * beside mvs_create -- instead of inside the first poissonSurface of a reconstruction: it builds and keeps the two plans of a
 * 2^grid_log2 grid (5..9) on the calling thread's current device.  Optional; returns 0, or a negative MVS_E* code. */
int mvs_poisson_warmup(int grid_log2);
int mvs_poisson_surface(const float *points, const float *normals, int n, int grid_log2, float smooth_cells, int keep_fields, mvs_surface **out);
int mvs_poisson_surface_ex(const float *points, const float *normals, int n, int grid_log2, float smooth_cells, float support_spacings, int keep_fields,
                           mvs_surface **out);
int mvs_surface_support(const mvs_surface *s, int *support_nodes /* the radius used, in nodes; 0: no trimming */);
/* The normals' lengths are confidences (triangulatePixels scales them by a pdf, util.cpp:322-327: 1e-6 .. 1e-4 on real frames) and only
 * their ratios matter; before the fixed-point splat they are multiplied by the power of two that brings their median size into [0.5, 1)
 * (exact; 2^0 for unit normals).  chi and the level of mvs_surface_grid carry that factor. */
int mvs_surface_normal_scale(const mvs_surface *s, int *scale_log2);
int mvs_surface_counts(const mvs_surface *s, int *vertices, int *faces);
int mvs_surface_fetch(const mvs_surface *s, float *vertices /* V x 4, w = 1 */, int32_t *faces /* F x 3, normals along the samples' */);
int mvs_surface_grid(const mvs_surface *s, int *nodes_per_axis, float origin3[3], float *spacing, float *level, float *chi /* G^3, nullable */,
                     int64_t *splat /* 4 G^3: vx, vy, vz, weight in units of 2^-16; nullable */);
int mvs_surface_spacing(const mvs_surface *s, float *average_spacing /* of the samples, 6 nearest neighbours */, float *node_spacing, int *ratio_kept);
/* The facet criteria the reference hands its mesher (cgal_poisson.cpp:50-52, 95-97: CGAL::Surface_mesh_default_criteria_3(sm_angle = 20
 * degrees, sm_radius = 300 x average spacing, sm_distance = 0.375 x average spacing)), as a pass over the surface's triangles
 * (csrc/surface_criteria.cpp; host code, like the reference's mesher): facets whose smallest angle is below min_angle_deg are removed by
 * edge collapses and edge flips between the vertices the mesher placed (no vertex moves, none is added; the link condition keeps the
 * surface a manifold wherever it was one), each guarded so that the surface moves by at most a quarter of max_distance; facets whose
 * circumradius exceeds max_radius are counted (the grid rule of mvs_poisson_surface keeps every facet two orders of magnitude below the
 * reference's bound, so nothing is done about them).  Vertices that lose all their facets are dropped; the order of the others, and of
 * the facets, is kept.  Last, facets that are still below the angle bound and lie on the BORDER of an open surface (where the samples'
 * support cut the level set) are removed when that only moves the border: a facet with two or three border edges, or with one and an
 * interior opposite vertex.  report (nullable) says what was done and what is left.  poissonSurface (host/poisson.cpp) and
 * mvs_amd.poisson_surface apply it with the reference's three numbers. */
typedef struct mvs_criteria_report {
    int collapses, flips;          /* operations applied */
    int facets_below_angle;        /* facets still below min_angle_deg afterwards */
    int facets_above_radius;       /* facets whose circumradius exceeds max_radius */
    float min_angle_deg;           /* smallest facet angle of the result (180 for an empty mesh) */
    float max_circumradius;        /* largest facet circumradius of the result */
    int facets_trimmed;            /* facets below min_angle_deg ON THE BORDER of an open surface that were removed (the border moved; see below) */
} mvs_criteria_report;
int mvs_surface_enforce_criteria(mvs_surface *s, float min_angle_deg /* [0, 60) */, float max_radius, float max_distance, mvs_criteria_report *report);
/* The other half of what those criteria mean: they bound a facet's angles from below and its distance from the surface from above -- NOT
 * its size (sm_radius = 300 spacings) -- so the reference's mesher returns as few facets as the curvature allows, where the grid mesher
 * returns the grid's density.  mvs_surface_simplify removes the vertices the criteria do not need: edge collapses u -> v, shortest edge
 * first, under the guards of mvs_surface_enforce_criteria (link condition, no facet turning over) with every rewritten facet keeping
 * min_angle_deg and every vertex keeping an ACCUMULATED displacement of at most max_distance (a vertex inherits what was merged into it);
 * no vertex moves, none is added, vertices on a border or on a non-manifold edge stay.  Apply after mvs_surface_enforce_criteria.
 * poissonSurface (host/poisson.cpp) and mvs_amd.poisson_surface run it with sm_angle and sm_distance. */
typedef struct mvs_simplify_report {
    int collapses;
    int vertices_before, vertices_after, facets_before, facets_after;
    float max_accumulated_distance;
} mvs_simplify_report;
int mvs_surface_simplify(mvs_surface *s, float min_angle_deg /* [0, 60) */, float max_distance, mvs_simplify_report *report);
/* a surface object over a caller's triangle mesh (vertices V x 4 with w = 1, faces F x 3), e.g. to apply the criteria to it; needs no GPU */
int mvs_surface_from_mesh(const float *vertices, int vertex_count, const int32_t *faces, int face_count, float average_spacing, mvs_surface **out);
void mvs_surface_free(mvs_surface *s);
const char *mvs_surface_last_error(void); /* of the calling thread */

/* library / device info string, e.g. "libmvs_hip gfx950 AMD Instinct MI355X" */
const char *mvs_device_info(mvs_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif /* MVS_H */
