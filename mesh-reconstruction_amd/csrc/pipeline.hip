// pipeline.hip -- the body of the reference's main-camera loop (recon.cpp:65-117) as ONE call on device-resident data.
//
// The reference runs, per main frame `fa`:      depth = render->depth(camera(fa))                       recon.cpp:70
//   for every side frame `fb`:                  projected = render->projected(camera(fa), frame(fb), camera(fb))    :85
//                                               projected = mixBackground(projected, frame(fa), depth)  (depth mutated) :86
//                                               flow = calculateFlow(frame(fa), projected, useFarneback)             :89
//   then                                        triangData = triangulatePixels(flows, camera(fa), cameras, depth)   :114
// Through the one-stage entry points of mvs.h each of these crosses PCIe twice (cv::Mat in, cv::Mat out), as the
// reference's own GL path does (two glReadPixels + two uploads per pair, render_glx.cpp:286,325,359,75).  Here the
// frames go up once, every intermediate (depth, warped image, mask, flows) stays in HBM, and only the points come
// back.  Same kernels, same arithmetic: the result equals the stage-by-stage calls bit for bit
// (tests/test_pipeline_gpu.py).
#include "mvs_internal.hpp"

using namespace mvs;

extern "C" {

int mvs_process_frame(mvs_ctx *ctx, const float main_cam[16], const uint8_t *main_frame_hw, int nside, const float *side_cams,
                      const uint8_t *const *side_frames_hw, int use_farneback, float *out_points7, int *out_count,
                      float *depth_after_hw /* nullable */)
{
    if (!ctx || !main_cam || !main_frame_hw || !out_points7 || !out_count || nside < 0 || (nside > 0 && (!side_cams || !side_frames_hw)))
        return fail(ctx, MVS_EINVAL, "mvs_process_frame: bad arguments");
    if (nside > 32) return fail(ctx, MVS_EINVAL, "mvs_process_frame: at most 32 side views");
    if (!ctx->soup.ptr) return fail(ctx, MVS_ESTATE, "mvs_process_frame: no mesh loaded (mvs_load_mesh)");
    MVS_HIP(ctx, hipSetDevice(ctx->device));
    const size_t P = (size_t)ctx->W * ctx->H;
    hipStream_t st = ctx->stream;
    // frame_buf: main u8 | side u8 | mixed u8 | out3 (3P u8) | pad to 256 | depth (P f32) | flows (nside * 4P f32)
    const size_t bytes_u8 = (6 * P + 255) & ~(size_t)255;
    int rc = ensure(ctx, ctx->frame_buf, bytes_u8 + sizeof(float) * P * (1 + 4 * (size_t)nside) + 256);
    if (rc) return rc;
    uint8_t *d_main = (uint8_t *)ctx->frame_buf.ptr, *d_side = d_main + P, *d_mixed = d_side + P, *d_out3 = d_mixed + P;
    float *d_depth = (float *)((uint8_t *)ctx->frame_buf.ptr + bytes_u8), *d_flows = d_depth + P;

    MVS_HIP(ctx, hipMemcpyAsync(d_main, main_frame_hw, P, hipMemcpyHostToDevice, st));
    if ((rc = depth_device(ctx, main_cam, d_depth))) return rc;  // recon.cpp:70
    std::vector<const float *> flow_ptrs((size_t)(nside > 0 ? nside : 1), nullptr);
    for (int i = 0; i < nside; i++) {
        if (!side_frames_hw[i]) return fail(ctx, MVS_EINVAL, "mvs_process_frame: side_frames[%d] is null", i);
        MVS_HIP(ctx, hipMemcpyAsync(d_side, side_frames_hw[i], P, hipMemcpyHostToDevice, st));
        if ((rc = projected_device(ctx, main_cam, d_side, side_cams + 16 * i, d_out3))) return rc;   // :85
        if ((rc = mix_background_device(ctx, d_out3, d_main, d_depth, d_mixed))) return rc;           // :86
        float *fl = d_flows + (size_t)i * 4 * P;
        if ((rc = flow_device(ctx, d_main, d_mixed, use_farneback, fl))) return rc;                   // :89
        flow_ptrs[i] = fl;
    }
    if (depth_after_hw) MVS_HIP(ctx, hipMemcpyAsync(depth_after_hw, d_depth, sizeof(float) * P, hipMemcpyDeviceToHost, st));
    return triangulate_impl(ctx, nside, flow_ptrs.data(), true, main_cam, side_cams, d_depth, out_points7, out_count);  // :114
}

}  // extern "C"
