// driver.cpp -- the loop of the reference's driver (recon.cpp:42-136) as two functions of the host mirror, so that the D-plane sweep
// can sit inside it: trackMainFrame = one pass of the `fa` loop body (recon.cpp:65-117), reconstructPoints = the outer iteration
// (recon.cpp:42-136).  recon.cpp itself stays what it is: a reference build linked per INTEGRATION.md runs its own main() against
// RenderHIP / calculateFlow exactly as before.  These functions are for a caller who wants the benchmarked capability (cost volume +
// depth selection over resident frames, mvs_sweep_handles) in that loop: `--sweep-planes N` (Configuration::sweepPlanes, default 0 =
// off) makes trackMainFrame hand the flows and triangulatePixels the swept depth instead of the proxy mesh's z-buffer.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <exception>
#include <memory>
#include <mutex>
#include <thread>

#include "recon.hpp"

namespace {

// the planes of a sweep span the proxy's own depth range seen from the main camera, widened by a quarter of it on both sides (and never
// beyond the NDC cube): the proxy says roughly where the surface is, the sweep decides where exactly
bool sweepRange(const Mat &proxyDepth, float &zLo, float &zHi)
{
    float lo = 2.f, hi = -2.f;
    const float *d = proxyDepth.ptr<float>();
    for (size_t i = 0; i < proxyDepth.total(); i++)
        if (d[i] != backgroundDepth) {
            lo = std::min(lo, d[i]);
            hi = std::max(hi, d[i]);
        }
    if (lo > hi) return false;  // the proxy covers nothing of this view
    const float margin = std::max(0.25f * (hi - lo), 1e-4f);
    zLo = std::max(-1.f, lo - margin);
    zHi = std::min(1.f, hi + margin);
    return zHi > zLo;
}

// a renderer keeps a sequence's frames in its frame store while the whole sequence stays below this (per renderer; 5 bytes per pixel and frame)
const size_t storeBudgetBytes = (size_t)8 << 30;

}  // namespace

Mat trackMainFrame(Configuration &config, Render *render, int fa, const std::vector<int> &sideFrames, Mat *depthUsed)
{
    const Mat originalImage = config.frame(fa);
    const Mat mainCamera = config.camera(fa);
    Mat depth = render->depth(mainCamera);  // recon.cpp:70
    DepthSweep *sweeper = config.sweepPlanes > 0 ? dynamic_cast<DepthSweep *>(render) : nullptr;
    float zLo = -1.f, zHi = 1.f;
    bool swept = false;
    if (sweeper && !sideFrames.empty() && sweepRange(depth, zLo, zHi)) {
        // frames go to the device once per sequence: whatever this view needs and the store does not hold yet
        if (sweeper->storeCapacity() < config.frameCount()) sweeper->storeFrames(config.frameCount());
        std::vector<Mat> sideCameras;
        for (int fb : sideFrames) {
            if (!sweeper->frameStored(fb)) sweeper->storeFrame(fb, config.frame(fb));
            sideCameras.push_back(config.camera(fb));
        }
        if (!sweeper->frameStored(fa)) sweeper->storeFrame(fa, originalImage);
        Mat sweptDepth = sweeper->sweepDepth(fa, mainCamera, sideFrames, sideCameras, config.sweepPlanes, zLo, zHi);
        // the proxy still says WHERE there is a surface (recon.cpp reconstructs what the mesh covers); the sweep says how deep it is
        float *dp = depth.ptr<float>();
        const float *sp = sweptDepth.ptr<float>();
        for (size_t i = 0; i < depth.total(); i++)
            if (dp[i] != backgroundDepth) dp[i] = sp[i];
        swept = true;
    }
    if (!swept) {
        // the reference's calls exactly; through the renderer's one-call form when it has one (the same rows bit for bit, tests/test_pipeline_gpu.py:
        // every intermediate stays on the device instead of crossing PCIe after each stage)
        if (FrameTracker *tracker = dynamic_cast<FrameTracker *>(render)) {
            std::vector<Mat> sideCams, sideImgs;
            for (int fb : sideFrames) sideCams.push_back(config.camera(fb));
            // a renderer with a frame store keeps the sequence's frames on the device (a frame is a main frame once and a side view about four times:
            // it crosses PCIe once) -- as long as the whole sequence fits the budget below; otherwise the frames travel with every call
            DepthSweep *store = dynamic_cast<DepthSweep *>(render);
            const size_t storeBytes = (size_t)config.frameCount() * originalImage.total() * 5;   // raw frame + the sweep's quad image per slot
            if (store && storeBytes <= storeBudgetBytes) {
                if (store->storeCapacity() < config.frameCount()) store->storeFrames(config.frameCount());
                if (!store->frameStored(fa)) store->storeFrame(fa, originalImage);
                for (int fb : sideFrames)
                    if (!store->frameStored(fb)) store->storeFrame(fb, config.frame(fb));
                return tracker->trackStoredFrame(mainCamera, fa, sideCams, sideFrames, config.useFarneback, depthUsed);
            }
            for (int fb : sideFrames) sideImgs.push_back(config.frame(fb));
            return tracker->trackFrame(mainCamera, originalImage, sideCams, sideImgs, config.useFarneback, depthUsed);
        }
    }
    MatList flows, cameras;
    for (int fb : sideFrames) {  // recon.cpp:81-112
        Mat projectedImage = swept ? sweeper->projectedByDepth(mainCamera, depth, config.frame(fb), config.camera(fb))
                                   : render->projected(mainCamera, config.frame(fb), config.camera(fb));
        projectedImage = mixBackground(projectedImage, originalImage, depth);
        flows.push_back(calculateFlow(originalImage, projectedImage, config.useFarneback));
        cameras.push_back(config.camera(fb));
    }
    if (depthUsed) *depthUsed = depth.clone();
    return triangulatePixels(flows, mainCamera, cameras, depth);  // recon.cpp:114
}

std::vector<Mat> trackMainFrames(Configuration &config, Heuristic &hint, Render *render, const Mesh &mesh, const std::vector<numberedVector> &schedule)
{
    std::vector<Mat> blocks(schedule.size());
    const int n = std::max(1, std::min(config.threads, (int)schedule.size()));
    if (n == 1) {
        for (size_t i = 0; i < schedule.size(); i++) blocks[i] = trackMainFrame(config, render, schedule[i].first, schedule[i].second);
        return blocks;
    }
    // One main frame keeps a fraction of the GPU busy and one host thread queues its ~200 launches: N threads, each with a renderer
    // (a context) of its own and the free functions' per-thread contexts, take main frames from a shared counter.  Thread 0 is the caller with `render`.
    // The extra renderers live as long as the process (a context takes milliseconds to create and its arenas grow on first use: not once per outer
    // iteration); the mesh of THIS iteration goes into each.  One sequence at a time: the pool is locked for the call.
    static std::mutex pool_mutex;
    static std::vector<std::unique_ptr<Render>> pool;
    static mvs::Size pool_size;
    std::lock_guard<std::mutex> pool_lock(pool_mutex);
    const mvs::Size size = hint.renderSize();
    if (size.width != pool_size.width || size.height != pool_size.height) {
        pool.clear();
        pool_size = size;
    }
    while ((int)pool.size() < n - 1) pool.emplace_back(spawnRender(hint));
    std::vector<std::unique_ptr<Render>> &extra = pool;
    for (int k = 1; k < n; k++) extra[(size_t)k - 1]->loadMesh(mesh);
    // every renderer's frame store is sized before the threads start (trackMainFrame would do it at a renderer's first frame: an allocation of the whole
    // sequence's size in the middle of the loop)
    if (config.sweepPlanes == 0 && !schedule.empty()) {
        const Mat first = config.frame(schedule[0].first);
        if ((size_t)config.frameCount() * first.total() * 5 <= storeBudgetBytes)
            for (int k = 0; k < n; k++)
                if (DepthSweep *store = dynamic_cast<DepthSweep *>(k ? extra[(size_t)k - 1].get() : render))
                    if (store->storeCapacity() < config.frameCount()) store->storeFrames(config.frameCount());
    }
    // Main frames are handed out in CHUNKS of consecutive schedule entries (two chunks per thread): neighbouring main frames share their side frames, and a
    // renderer uploads a frame once for all the main frames IT tracks -- one frame at a time from a shared counter would send every frame to every renderer.
    const size_t chunk = std::max<size_t>(1, schedule.size() / (2 * (size_t)n));
    std::atomic<size_t> next{0};
    std::mutex err_mutex;
    std::exception_ptr err;
    auto work = [&](Render *r) {
        try {
            for (size_t c = next.fetch_add(chunk); c < schedule.size(); c = next.fetch_add(chunk))
                for (size_t i = c; i < std::min(schedule.size(), c + chunk); i++) blocks[i] = trackMainFrame(config, r, schedule[i].first, schedule[i].second);
        } catch (...) {
            std::lock_guard<std::mutex> lock(err_mutex);
            if (!err) err = std::current_exception();
            next.store(schedule.size());  // nobody starts another chunk
        }
    };
    std::vector<std::thread> threads;
    for (int k = 1; k < n; k++) threads.emplace_back(work, extra[(size_t)k - 1].get());
    work(render);
    for (auto &t : threads) t.join();
    if (err) std::rethrow_exception(err);
    return blocks;
}

void reconstructPoints(Configuration &config, Heuristic &hint, Render *render, Mat &points, Mat &normals)
{
    while (hint.notHappy(points)) {  // recon.cpp:42
        const Mesh mesh = hint.tessellate(points, normals);
        render->loadMesh(mesh);
        if (hint.chooseCameras(mesh, config.allCameras(), *render) == 0) throw std::runtime_error("Heuristic has chosen no cameras");  // recon.cpp:54-57
        std::vector<numberedVector> schedule;
        for (int fa = hint.beginMain(); fa != Heuristic::sentinel; fa = hint.nextMain()) {
            std::vector<int> sides;
            for (int fb = hint.beginSide(fa); fb != Heuristic::sentinel; fb = hint.nextSide(fa)) sides.push_back(fb);
            schedule.emplace_back(fa, sides);
        }
        const std::vector<Mat> blocks = trackMainFrames(config, hint, render, mesh, schedule);
        for (size_t i = 0; i < blocks.size(); i++) {
            const Mat &tri = blocks[i];
            // recon.cpp:115-116: columns 0-3 are the points, 4-6 the normals
            Mat p(tri.rows, 4, mvs::F32C1), n(tri.rows, 3, mvs::F32C1);
            for (int r = 0; r < tri.rows; r++) {
                const float *row = tri.ptr<float>(r);
                std::copy(row, row + 4, p.ptr<float>(r));
                std::copy(row + 4, row + 7, n.ptr<float>(r));
            }
            points.push_back(p);
            normals.push_back(n);
            if (config.verbosity >= 2) printf(" After processing main frame %i: %i points\n", schedule[i].first, points.rows);
        }
        hint.filterPoints(points, normals);  // recon.cpp:123
    }
}
