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
// back.  The flows of the side views are independent of each other, so each runs in one of four lanes -- a stream, a shadow context with
// the lane's flow arena, and (round 6) a host thread that queues the flow's launches -- while the calling thread and the main stream go on
// with the next view; what does not depend on the side view (the depth map, the main-camera half of projected(), every frame's texture) is
// done once up front, and the variance channels of all flows in one batched pass at the end.  Same kernels' arithmetic: the result equals
// the stage-by-stage calls bit for bit (tests/test_pipeline_gpu.py).
#include "mvs_internal.hpp"

#include <algorithm>
#include <condition_variable>
#include <cstdlib>
#include <deque>
#include <functional>
#include <mutex>
#include <new>
#include <thread>

using namespace mvs;

namespace {

// One host thread per flow lane: takes jobs (queue one side view's flow on the lane's stream) in order; idle() returns when everything handed over has
// been queued.  Started by the first call that uses the lane, joined by lanes_shutdown (mvs_destroy).
struct LaneWorker {
    std::thread th;
    std::mutex m;
    std::condition_variable cv, done;
    std::deque<std::function<void()>> jobs;
    bool busy = false, quit = false;
    int device = 0;

    void loop()
    {
        (void)hipSetDevice(device);
        for (;;) {
            std::function<void()> job;
            {
                std::unique_lock<std::mutex> lock(m);
                cv.wait(lock, [&] { return quit || !jobs.empty(); });
                if (jobs.empty()) return;  // quit
                job = std::move(jobs.front());
                jobs.pop_front();
                busy = true;
            }
            job();
            {
                std::lock_guard<std::mutex> lock(m);
                busy = false;
                if (jobs.empty()) done.notify_all();
            }
        }
    }
    void submit(std::function<void()> job)
    {
        {
            std::lock_guard<std::mutex> lock(m);
            if (!th.joinable()) th = std::thread(&LaneWorker::loop, this);
            jobs.push_back(std::move(job));
        }
        cv.notify_one();
    }
    void idle()
    {
        std::unique_lock<std::mutex> lock(m);
        done.wait(lock, [&] { return jobs.empty() && !busy; });
    }
    void stop()
    {
        {
            std::lock_guard<std::mutex> lock(m);
            quit = true;
        }
        cv.notify_all();
        if (th.joinable()) th.join();
    }
};

}  // namespace

namespace mvs {

void lanes_shutdown(mvs_ctx *ctx)
{
    for (auto &lane : ctx->lanes) {
        if (lane.worker) {
            LaneWorker *w = (LaneWorker *)lane.worker;
            w->stop();
            delete w;
            lane.worker = nullptr;
        }
        if (lane.stream) (void)hipStreamSynchronize(lane.stream);
        if (lane.shadow) {
            if (lane.shadow->flow_arena.ptr) (void)hipFree(lane.shadow->flow_arena.ptr);
            delete lane.shadow;
            lane.shadow = nullptr;
        }
    }
}

}  // namespace mvs

// One main frame.  The frames are host buffers (main_frame_hw / side_frames_hw: mvs_process_frame) or slots of the frame store (main_slot /
// side_slots: mvs_process_frame_slots -- a sequence's frames cross PCIe once instead of once per main frame they take part in).
static int process_frame_impl(mvs_ctx *ctx, const float main_cam[16], const uint8_t *main_frame_hw, int main_slot, int nside, const float *side_cams,
                              const uint8_t *const *side_frames_hw, const int *side_slots, int use_farneback, float *out_points7, int *out_count,
                              float *depth_after_hw /* nullable */)
{
    const bool slots = !main_frame_hw;
    if (nside > 32) return fail(ctx, MVS_EINVAL, "mvs_process_frame: at most 32 side views");
    if (!ctx->soup.ptr) return fail(ctx, MVS_ESTATE, "mvs_process_frame: no mesh loaded (mvs_load_mesh)");
    if (slots) {
        for (int i = -1; i < nside; i++) {
            const int slot = i < 0 ? main_slot : side_slots[i];
            if (slot < 0 || slot >= ctx->store_cap) return fail(ctx, MVS_EINVAL, "mvs_process_frame_slots: slot %d outside the store (capacity %d: mvs_frame_store first)", slot, ctx->store_cap);
            if (!ctx->store_have[(size_t)slot]) return fail(ctx, MVS_ESTATE, "mvs_process_frame_slots: slot %d holds no frame (mvs_frame_upload)", slot);
        }
    }
    MVS_HIP(ctx, hipSetDevice(ctx->device));
    const size_t P = (size_t)ctx->W * ctx->H;
    hipStream_t st = ctx->stream;
    // frame_buf: main u8 | side u8 x nside | out3 (3P u8) | mixed u8 x nside | remapped u8 x nside | pad to 256 | depth (P f32) | flows (nside * 4P f32) |
    // raw flows (nside * 2P f32) | variances (nside * P f32)
    const size_t bytes_u8 = ((4 + 3 * (size_t)nside) * P + 255) & ~(size_t)255;
    int rc = ensure(ctx, ctx->frame_buf, bytes_u8 + sizeof(float) * P * (1 + 7 * (size_t)nside) + 256);
    if (rc) return rc;
    uint8_t *d_main = (uint8_t *)ctx->frame_buf.ptr, *d_side0 = d_main + P, *d_out3 = d_side0 + (size_t)nside * P, *d_mixed0 = d_out3 + 3 * P;
    uint8_t *d_remapped = d_mixed0 + (size_t)nside * P;
    float *d_depth = (float *)((uint8_t *)ctx->frame_buf.ptr + bytes_u8), *d_flows = d_depth + P, *d_flow2 = d_flows + 4 * (size_t)nside * P, *d_var = d_flow2 + 2 * (size_t)nside * P;

    // The flows of the side views depend only on (main frame, mixed_i): each runs in a lane of its own while the main stream goes on
    // rasterising the next view; a flow is a chain of small kernels that fills 150 of 256 CUs at best, so up to four of them overlap.
    // Everything joins before the batched variance pass and triangulatePixels.
    const bool serial = ctx->hooks.serial_flows;  // A/B: all flows in the main stream, as before
    // Farneback (-f): the flows of all side views in ONE pass after the last mixed image (every launch covers all of them; what
    // depends on the main frame alone is computed once) -- a Farneback flow is ~150 launches of a few microseconds, and concurrency
    // between lanes does not buy what sharing the launches does.  MVS_FB_LANES=1 keeps the per-view chains on the lanes (A/B).
    const bool fb_lanes = ctx->hooks.fb_lanes;
    const bool fb_batch = use_farneback && nside > 0 && !serial && !fb_lanes;
    // The LAST side view's flow stays on the main stream: after that view's side pass the main stream has nothing to do until the flows join, and HIP
    // serves a process's streams from four hardware queues -- main stream + four lanes made the fourth lane share a queue with another one, and its flow
    // started when that one's had finished (200 us of a 1.5 ms call at 640 x 480 with four side views, profiles/r06).
    const int nlanes = (serial || fb_batch) ? 0 : std::max(0, std::min(nside - 1, (int)mvs_ctx::kFlowLanes));
    for (int l = 0; l < nlanes; l++)
        if (!ctx->lanes[l].stream) MVS_HIP(ctx, hipStreamCreateWithFlags(&ctx->lanes[l].stream, hipStreamNonBlocking));
    while ((int)ctx->lane_events.size() < 2 * nside) {
        hipEvent_t e;
        MVS_HIP(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        ctx->lane_events.push_back(e);
    }
    // a lane's shadow context: this context's stand-in on the lane's stream (flow_only_device reads device, size, hooks, stream and its own flow arena)
    for (int l = 0; l < nlanes; l++) {
        mvs_ctx::FlowLane &lane = ctx->lanes[l];
        if (!lane.shadow) {
            lane.shadow = new (std::nothrow) mvs_ctx();
            if (!lane.shadow) return fail(ctx, MVS_ENOMEM, "mvs_process_frame: out of host memory");
            lane.shadow->device = ctx->device;
            lane.shadow->W = ctx->W;
            lane.shadow->H = ctx->H;
            lane.shadow->num_cus = ctx->num_cus;
            lane.shadow->hooks = ctx->hooks;
            snprintf(lane.shadow->err, sizeof(lane.shadow->err), "no error");
        }
        lane.shadow->stream = lane.stream;
        if (!lane.worker) {
            LaneWorker *w = new (std::nothrow) LaneWorker();
            if (!w) return fail(ctx, MVS_ENOMEM, "mvs_process_frame: out of host memory");
            w->device = ctx->device;
            lane.worker = w;
        }
    }
    std::vector<int> lane_rc((size_t)(nside > 0 ? nside : 1), MVS_OK);

    // Any early return below leaves kernels and host-to-device copies of the caller's frames in flight on the main stream
    // and on the lanes: the guard joins them all first, so neither the caller's buffers nor this context's arenas (which a
    // later ensure() or mvs_destroy may free) are still in use when an error is reported.
    struct JoinOnError {
        mvs_ctx *c;
        hipStream_t st;
        int nlanes;
        bool armed = true;
        ~JoinOnError()
        {
            if (!armed) return;
            for (int l = 0; l < nlanes; l++)
                if (c->lanes[l].worker) ((LaneWorker *)c->lanes[l].worker)->idle();  // nothing is being queued any more ...
            (void)hipStreamSynchronize(st);
            for (int l = 0; l < nlanes; l++)
                if (c->lanes[l].stream) (void)hipStreamSynchronize(c->lanes[l].stream);  // ... and nothing queued is still running
        }
    } join{ctx, st, nlanes};

    if (!slots)
        for (int i = 0; i < nside; i++)
            if (!side_frames_hw[i]) return fail(ctx, MVS_EINVAL, "mvs_process_frame: side_frames[%d] is null", i);
    // What needs no frame goes first -- the depth map and the half of projected() that rasterises the mesh from the MAIN camera (once per main frame,
    // not once per side view: round 6) -- so the GPU works while the host stages the frames; then all frames go up back to back (a copy from pageable
    // memory in between two kernels cost 14 us of idle stream per side view)
    if ((rc = depth_device(ctx, main_cam, d_depth))) return rc;  // recon.cpp:70
    if (nside > 0 && (rc = projected_main_pass(ctx, main_cam))) return rc;
    // where the frames are on the device: the slots of the frame store as they lie (stream-ordered behind the uploads that filled them; nothing is copied),
    // or this call's upload buffer
    const uint8_t *main_dev = d_main;
    std::vector<const uint8_t *> side_dev((size_t)(nside > 0 ? nside : 1), nullptr);
    if (slots) {
        const uint8_t *raw = (const uint8_t *)ctx->store_raw.ptr;
        main_dev = raw + P * (size_t)main_slot;
        for (int i = 0; i < nside; i++) side_dev[(size_t)i] = raw + P * (size_t)side_slots[i];
    } else {
        MVS_HIP(ctx, hipMemcpyAsync(d_main, main_frame_hw, P, hipMemcpyHostToDevice, st));
        for (int i = 0; i < nside; i++) {
            MVS_HIP(ctx, hipMemcpyAsync(d_side0 + (size_t)i * P, side_frames_hw[i], P, hipMemcpyHostToDevice, st));
            side_dev[(size_t)i] = d_side0 + (size_t)i * P;
        }
    }
    if (nside > 0 && (rc = projected_prepare_views(ctx, side_dev.data(), nside))) return rc;   // every side frame's texture (wrap padding, mip chain): five launches in all
    std::vector<const float *> flow_ptrs((size_t)(nside > 0 ? nside : 1), nullptr);
    for (int i = 0; i < nside; i++) {
        uint8_t *d_mixed = d_mixed0 + (size_t)i * P;
        if ((rc = projected_side_pass(ctx, side_dev[(size_t)i], side_cams + 16 * i, d_out3, i, main_dev, d_depth, d_mixed))) return rc;   // :85 + :86 (mixBackground inside the fragment program's launch)
        float *fl = d_flows + (size_t)i * 4 * P;
        if (fb_batch) {
            // (after the loop)
        } else if (nlanes == 0 || i == nside - 1) {
            if ((rc = flow_only_device(ctx, main_dev, d_mixed, use_farneback, d_flow2 + (size_t)i * 2 * P))) return rc;               // :89 (the flow; its variance channel below)
        } else {
            // the lane's host thread queues this view's flow (a dozen launches for the variational refinement, ~100 for a Farneback chain) while this thread
            // goes on with the next side view; the lane's stream waits for `ready` (this view's mixed image), the main stream later for `done`
            mvs_ctx::FlowLane &lane = ctx->lanes[i % nlanes];
            hipEvent_t ready = ctx->lane_events[2 * i], done = ctx->lane_events[2 * i + 1];
            MVS_HIP(ctx, hipEventRecord(ready, st));
            mvs_ctx *shadow = lane.shadow;
            hipStream_t ls = lane.stream;
            float *flow2_i = d_flow2 + (size_t)i * 2 * P;
            int *rc_i = &lane_rc[(size_t)i];
            ((LaneWorker *)lane.worker)->submit([=]() {
                int r = MVS_OK;
                if (hipStreamWaitEvent(ls, ready, 0) != hipSuccess) r = MVS_EHIP;
                if (r == MVS_OK) r = flow_only_device(shadow, main_dev, d_mixed, use_farneback, flow2_i);   // :89 (the flow; its variance channel below)
                if (r == MVS_OK && hipEventRecord(done, ls) != hipSuccess) r = MVS_EHIP;
                *rc_i = r;
            });
        }
        flow_ptrs[i] = fl;
    }
    if (fb_batch && (rc = flow_farneback_batch_device(ctx, main_dev, d_mixed0, nside, d_flows))) return rc;                                            // :89, all side views
    if (nlanes > 0) {
        for (int l = 0; l < nlanes; l++) ((LaneWorker *)ctx->lanes[l].worker)->idle();   // every flow is queued (and every `done` recorded)
        for (int i = 0; i < nside - 1; i++)
            if (lane_rc[(size_t)i] != MVS_OK) return fail(ctx, lane_rc[(size_t)i], "mvs_process_frame: flow of side view %d: %s", i, ctx->lanes[i % nlanes].shadow->err);
        for (int i = 0; i < nside - 1; i++) MVS_HIP(ctx, hipStreamWaitEvent(st, ctx->lane_events[2 * i + 1], 0));
    }
    // flow.cpp:34-41 for every side view at once: the variance channels are twelve launches per flow of ~5 us each -- on the main stream, all flows per
    // launch, they are twelve per main frame (round 6; the batched Farneback pass does the same inside flow_farneback_batch_device)
    if (!fb_batch && nside > 0 && (rc = flow_variance_batch_device(ctx, main_dev, d_mixed0, d_flow2, nside, d_remapped, d_var, d_flows))) return rc;
    if (depth_after_hw) MVS_HIP(ctx, hipMemcpyAsync(depth_after_hw, d_depth, sizeof(float) * P, hipMemcpyDeviceToHost, st));
    rc = triangulate_impl(ctx, nside, flow_ptrs.data(), true, main_cam, side_cams, d_depth, out_points7, out_count);  // :114
    if (rc == MVS_OK) join.armed = false;  // triangulate_impl synchronised the main stream, which had joined every lane
    return rc;
}

extern "C" {

int mvs_process_frame(mvs_ctx *ctx, const float main_cam[16], const uint8_t *main_frame_hw, int nside, const float *side_cams,
                      const uint8_t *const *side_frames_hw, int use_farneback, float *out_points7, int *out_count,
                      float *depth_after_hw /* nullable */)
{
    if (!ctx || !main_cam || !main_frame_hw || !out_points7 || !out_count || nside < 0 || (nside > 0 && (!side_cams || !side_frames_hw)))
        return fail(ctx, MVS_EINVAL, "mvs_process_frame: bad arguments");
    return process_frame_impl(ctx, main_cam, main_frame_hw, -1, nside, side_cams, side_frames_hw, nullptr, use_farneback, out_points7, out_count, depth_after_hw);
}

int mvs_process_frame_slots(mvs_ctx *ctx, const float main_cam[16], int main_slot, int nside, const float *side_cams, const int *side_slots,
                            int use_farneback, float *out_points7, int *out_count, float *depth_after_hw /* nullable */)
{
    if (!ctx || !main_cam || !out_points7 || !out_count || nside < 0 || (nside > 0 && (!side_cams || !side_slots)))
        return fail(ctx, MVS_EINVAL, "mvs_process_frame_slots: bad arguments");
    return process_frame_impl(ctx, main_cam, nullptr, main_slot, nside, side_cams, nullptr, side_slots, use_farneback, out_points7, out_count, depth_after_hw);
}

}  // extern "C"
