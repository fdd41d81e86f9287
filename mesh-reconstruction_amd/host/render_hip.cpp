// render_hip.cpp -- `class RenderHIP : public Render` + `spawnRender`: the link-time renderer seam of the
// reference (render_${SYSTEM_OPENGL}.cpp, Makefile:2,16,21; "render_<whatever>.cpp in the future", recon.hpp:92)
// implemented on libmvs_hip.so.  Also the free functions of recon.hpp:40-50 that sit on the hot path.
//
// Ownership as in the reference: returned Mats are freshly allocated and owned by the caller; the Render*
// from spawnRender is a raw owning pointer the caller deletes (recon.cpp:130); inputs are copied to the
// device inside each call and never retained.
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "../../include/mvs.h"
#include "recon.hpp"

namespace {

[[noreturn]] void raise(mvs_ctx *ctx, const char *what)
{
    throw std::runtime_error(std::string(what) + ": " + mvs_last_error(ctx));
}

int device_from_env()
{
    const char *e = getenv("MVS_DEVICE");
    return e ? atoi(e) : 0;
}

// the reference's free functions carry no context argument; they share one context per image size AND HOST THREAD: calls on one context are the
// caller's to serialise (include/mvs.h), and reconstructPoints runs trackMainFrame on several threads (host/driver.cpp) -- each thread's calculateFlow,
// mixBackground, triangulatePixels ... then work on contexts of their own, destroyed when the thread ends (the main thread's at process exit)
struct ThreadContexts {
    std::map<std::pair<int, int>, mvs_ctx *> ctx;
    ~ThreadContexts()
    {
        for (auto &kv : ctx) mvs_destroy(kv.second);
    }
};
thread_local ThreadContexts t_ctx;

mvs_ctx *shared_ctx(int w, int h)
{
    auto key = std::make_pair(w, h);
    auto it = t_ctx.ctx.find(key);
    if (it != t_ctx.ctx.end()) return it->second;
    mvs_ctx *c = mvs_create(device_from_env(), w, h);
    if (!c) raise(nullptr, "mvs_create");
    t_ctx.ctx[key] = c;
    return c;
}

void expect(const Mat &m, int type, const char *what)
{
    if (m.empty() || m.type() != type) throw std::runtime_error(std::string(what) + ": unexpected matrix type or empty matrix");
}

}  // namespace

class RenderHIP : public Render, public DepthProbe, public DepthSweep, public FrameTracker {
public:
    RenderHIP(int width, int height) : w(width), h(height)
    {
        ctx = mvs_create(device_from_env(), width, height);  // replaces the GLX pbuffer + GL 3.0 context (render_glx.cpp:152-208)
        if (!ctx) raise(nullptr, "RenderHIP: mvs_create");
    }
    ~RenderHIP() override { mvs_destroy(ctx); }

    void loadMesh(const Mesh mesh) override  // render_glx.cpp:230-258
    {
        expect(mesh.vertices, mvs::F32C1, "loadMesh vertices");
        if (mesh.vertices.cols != 4) throw std::runtime_error("loadMesh: vertices must be N x 4 homogeneous rows");
        const int nf = mesh.faces.rows;
        if (nf > 0) expect(mesh.faces, mvs::S32C1, "loadMesh faces");
        if (mvs_load_mesh(ctx, mesh.vertices.ptr<float>(), mesh.vertices.rows, nf ? mesh.faces.ptr<int32_t>() : nullptr, nf))
            raise(ctx, "loadMesh");
    }

    Mat projected(const Mat camera, const Mat frame, const Mat projector) override  // render_glx.cpp:261-367
    {
        expect(camera, mvs::F32C1, "projected camera");
        expect(projector, mvs::F32C1, "projected projector");
        expect(frame, mvs::U8C1, "projected frame");  // assert(image.channels() == 1), render_glx.cpp:66
        if (frame.cols != w || frame.rows != h) throw std::runtime_error("projected: frame size differs from the render size");
        Mat result(h, w, mvs::U8C3);
        if (mvs_projected(ctx, camera.ptr<float>(), frame.ptr<uint8_t>(), projector.ptr<float>(), result.ptr<uint8_t>()))
            raise(ctx, "projected");
        return result;
    }

    Mat depth(const Mat camera) const override  // render_glx.cpp:369-397
    {
        expect(camera, mvs::F32C1, "depth camera");
        Mat result(h, w, mvs::F32C1);
        if (mvs_depth(ctx, camera.ptr<float>(), result.ptr<float>())) raise(ctx, "depth");
        return result;
    }

    void depthAt(const Mat camera, int n, const int32_t *rows, const int32_t *cols, float *out) const override
    {
        expect(camera, mvs::F32C1, "depthAt camera");
        if (mvs_depth_probe(ctx, camera.ptr<float>(), n, rows, cols, out)) raise(ctx, "depthAt");
    }

    // ---- DepthSweep: the frame store + mvs_sweep_handles (one main view over stored frames: nothing uploaded, copied or re-prepared per view) ----
    void storeFrames(int frameCount) override
    {
        if (mvs_frame_store(ctx, frameCount)) raise(ctx, "storeFrames");
        stored.assign((size_t)frameCount, 0);
    }
    int storeCapacity() const override { return (int)stored.size(); }
    void storeFrame(int frameNo, const Mat gray) override
    {
        expect(gray, mvs::U8C1, "storeFrame frame");
        if (gray.cols != w || gray.rows != h) throw std::runtime_error("storeFrame: frame size differs from the render size");
        if (frameNo < 0 || frameNo >= (int)stored.size()) throw std::runtime_error("storeFrame: frame number outside the store (storeFrames)");
        // the upload is asynchronous and the caller's Mat may go away: wait here (once per frame of the sequence, not per sweep)
        if (mvs_frame_upload(ctx, frameNo, gray.ptr<uint8_t>()) || mvs_synchronize(ctx)) raise(ctx, "storeFrame");
        stored[(size_t)frameNo] = 1;
    }
    bool frameStored(int frameNo) const override { return frameNo >= 0 && frameNo < (int)stored.size() && stored[(size_t)frameNo]; }
    Mat sweepDepth(int mainFrame, const Mat mainCamera, const std::vector<int> &sideFrames, const std::vector<Mat> &sideCameras, int planes, float zLo, float zHi,
                   Mat *bestCost) override
    {
        expect(mainCamera, mvs::F32C1, "sweepDepth mainCamera");
        if (sideFrames.size() != sideCameras.size()) throw std::runtime_error("sweepDepth: one camera per side frame expected");
        if (!frameStored(mainFrame)) throw std::runtime_error("sweepDepth: the main frame is not in the store (storeFrame)");
        std::vector<float> cams;
        for (size_t i = 0; i < sideFrames.size(); i++) {
            if (!frameStored(sideFrames[i])) throw std::runtime_error("sweepDepth: a side frame is not in the store (storeFrame)");
            expect(sideCameras[i], mvs::F32C1, "sweepDepth side camera");
            cams.insert(cams.end(), sideCameras[i].ptr<float>(), sideCameras[i].ptr<float>() + 16);
        }
        Mat depth(h, w, mvs::F32C1);
        if (bestCost) bestCost->create(h, w, mvs::F32C1);
        if (mvs_sweep_handles(ctx, mainFrame, mainCamera.ptr<float>(), (int)sideFrames.size(), sideFrames.data(), cams.data(), planes, zLo, zHi, depth.ptr<float>(),
                              bestCost ? bestCost->ptr<float>() : nullptr))
            raise(ctx, "sweepDepth");
        return depth;
    }
    Mat projectedByDepth(const Mat camera, const Mat depth, const Mat frame, const Mat projector) override
    {
        expect(camera, mvs::F32C1, "projectedByDepth camera");
        expect(projector, mvs::F32C1, "projectedByDepth projector");
        expect(depth, mvs::F32C1, "projectedByDepth depth");
        expect(frame, mvs::U8C1, "projectedByDepth frame");
        if (frame.cols != w || frame.rows != h || depth.cols != w || depth.rows != h) throw std::runtime_error("projectedByDepth: size differs from the render size");
        std::vector<uint8_t> pairs((size_t)w * h * 2);
        if (mvs_warp_by_depth(ctx, camera.ptr<float>(), depth.ptr<float>(), projector.ptr<float>(), frame.ptr<uint8_t>(), pairs.data())) raise(ctx, "projectedByDepth");
        Mat result(h, w, mvs::U8C3);
        uint8_t *out = result.ptr<uint8_t>();
        for (size_t p = 0; p < (size_t)w * h; p++) {
            out[3 * p] = pairs[2 * p];
            out[3 * p + 1] = out[3 * p + 2] = pairs[2 * p + 1];
        }
        return result;
    }

    // ---- FrameTracker: mvs_process_frame on this renderer's context (its mesh is the one loadMesh put there) ----
    Mat trackFrame(const Mat mainCamera, const Mat mainFrame, const std::vector<Mat> &sideCameras, const std::vector<Mat> &sideFrames, bool useFarneback, Mat *depthAfter) override
    {
        expect(mainCamera, mvs::F32C1, "trackFrame mainCamera");
        expect(mainFrame, mvs::U8C1, "trackFrame mainFrame");
        if (mainFrame.cols != w || mainFrame.rows != h) throw std::runtime_error("trackFrame: frame size differs from the render size");
        if (sideCameras.size() != sideFrames.size()) throw std::runtime_error("trackFrame: one camera per side frame expected");
        std::vector<float> cams;
        std::vector<const uint8_t *> frames;
        for (size_t i = 0; i < sideFrames.size(); i++) {
            expect(sideCameras[i], mvs::F32C1, "trackFrame side camera");
            expect(sideFrames[i], mvs::U8C1, "trackFrame side frame");
            if (sideFrames[i].cols != w || sideFrames[i].rows != h) throw std::runtime_error("trackFrame: frame size differs from the render size");
            cams.insert(cams.end(), sideCameras[i].ptr<float>(), sideCameras[i].ptr<float>() + 16);
            frames.push_back(sideFrames[i].ptr<uint8_t>());
        }
        // room for one row per pixel, kept with the renderer (a fresh 8.6 MB Mat per main frame would be zero-filled and then copied again)
        if (rows_scratch.size() < (size_t)w * h * 7) rows_scratch.resize((size_t)w * h * 7);
        if (depthAfter) depthAfter->create(h, w, mvs::F32C1);
        int n = 0;
        if (mvs_process_frame(ctx, mainCamera.ptr<float>(), mainFrame.ptr<uint8_t>(), (int)frames.size(), cams.data(), frames.data(), useFarneback ? 1 : 0, rows_scratch.data(), &n,
                              depthAfter ? depthAfter->ptr<float>() : nullptr))
            raise(ctx, "trackFrame");
        Mat rows(n, 7, mvs::F32C1);
        if (n > 0) std::memcpy(rows.data, rows_scratch.data(), (size_t)n * 7 * sizeof(float));
        return rows;
    }

    Mat trackStoredFrame(const Mat mainCamera, int mainFrame, const std::vector<Mat> &sideCameras, const std::vector<int> &sideFrames, bool useFarneback, Mat *depthAfter) override
    {
        expect(mainCamera, mvs::F32C1, "trackStoredFrame mainCamera");
        if (sideCameras.size() != sideFrames.size()) throw std::runtime_error("trackStoredFrame: one camera per side frame expected");
        if (!frameStored(mainFrame)) throw std::runtime_error("trackStoredFrame: the main frame is not in the store (storeFrame)");
        std::vector<float> cams;
        for (size_t i = 0; i < sideFrames.size(); i++) {
            if (!frameStored(sideFrames[i])) throw std::runtime_error("trackStoredFrame: a side frame is not in the store (storeFrame)");
            expect(sideCameras[i], mvs::F32C1, "trackStoredFrame side camera");
            cams.insert(cams.end(), sideCameras[i].ptr<float>(), sideCameras[i].ptr<float>() + 16);
        }
        if (rows_scratch.size() < (size_t)w * h * 7) rows_scratch.resize((size_t)w * h * 7);
        if (depthAfter) depthAfter->create(h, w, mvs::F32C1);
        int n = 0;
        if (mvs_process_frame_slots(ctx, mainCamera.ptr<float>(), mainFrame, (int)sideFrames.size(), cams.data(), sideFrames.data(), useFarneback ? 1 : 0, rows_scratch.data(), &n,
                                    depthAfter ? depthAfter->ptr<float>() : nullptr))
            raise(ctx, "trackStoredFrame");
        Mat rows(n, 7, mvs::F32C1);
        if (n > 0) std::memcpy(rows.data, rows_scratch.data(), (size_t)n * 7 * sizeof(float));
        return rows;
    }

    mvs_ctx *context() const { return ctx; }

protected:
    mvs_ctx *ctx;
    int w, h;
    std::vector<unsigned char> stored;
    std::vector<float> rows_scratch;
};

// render_glx.cpp:57-62
Render *spawnRender(Heuristic hint)
{
    mvs::Size size = hint.renderSize();
    return new RenderHIP(size.width, size.height);
}

// flow.cpp:19-42
Mat calculateFlow(const Mat prev, const Mat next, bool useFarneback)
{
    expect(prev, mvs::U8C1, "calculateFlow prev");
    expect(next, mvs::U8C1, "calculateFlow next");
    mvs_ctx *ctx = shared_ctx(prev.cols, prev.rows);
    Mat mixed(prev.rows, prev.cols, mvs::F32C4);
    if (mvs_flow(ctx, prev.ptr<uint8_t>(), next.ptr<uint8_t>(), useFarneback ? 1 : 0, mixed.ptr<float>())) raise(ctx, "calculateFlow");
    return mixed;
}

// util.cpp:332-361
Mat compare(const Mat prev, const Mat next)
{
    expect(prev, mvs::U8C1, "compare prev");
    expect(next, mvs::U8C1, "compare next");
    mvs_ctx *ctx = shared_ctx(prev.cols, prev.rows);
    Mat out(prev.rows, prev.cols, mvs::F32C1);
    if (mvs_compare(ctx, prev.ptr<uint8_t>(), next.ptr<uint8_t>(), out.ptr<float>())) raise(ctx, "compare");
    return out;
}

// util.cpp:366-387
Mat mixBackground(const Mat image, const Mat background, Mat &depth)
{
    expect(image, mvs::U8C3, "mixBackground image");       // assert(image.channels() == 3)
    expect(background, mvs::U8C1, "mixBackground background");
    expect(depth, mvs::F32C1, "mixBackground depth");
    mvs_ctx *ctx = shared_ctx(depth.cols, depth.rows);
    Mat result(depth.rows, depth.cols, mvs::U8C1);
    if (mvs_mix_background(ctx, image.ptr<uint8_t>(), background.ptr<uint8_t>(), depth.ptr<float>(), result.ptr<uint8_t>()))
        raise(ctx, "mixBackground");
    return result;
}

// util.cpp:390-403
Mat flowRemap(const Mat flow, const Mat image)
{
    expect(image, mvs::U8C1, "flowRemap image");
    if (flow.empty() || (flow.type() != mvs::F32C2 && flow.type() != mvs::F32C4)) throw std::runtime_error("flowRemap: flow must be 2 or 4 channel f32");
    mvs_ctx *ctx = shared_ctx(image.cols, image.rows);
    Mat out(image.rows, image.cols, mvs::U8C1);
    if (mvs_flow_remap(ctx, flow.ptr<float>(), flow.channels(), image.ptr<uint8_t>(), out.ptr<uint8_t>())) raise(ctx, "flowRemap");
    return out;
}

// util.cpp:167-329 (recon.cpp:114)
Mat triangulatePixels(const MatList flows, const Mat mainCamera, const MatList cameras, const Mat depth)
{
    expect(depth, mvs::F32C1, "triangulatePixels depth");
    expect(mainCamera, mvs::F32C1, "triangulatePixels mainCamera");
    if (flows.size() != cameras.size()) throw std::runtime_error("triangulatePixels: one flow per side camera expected");
    mvs_ctx *ctx = shared_ctx(depth.cols, depth.rows);
    std::vector<const float *> fl;
    std::vector<float> cams;
    auto cam = cameras.begin();
    for (auto f = flows.begin(); f != flows.end(); ++f, ++cam) {
        expect(*f, mvs::F32C4, "triangulatePixels flow");
        expect(*cam, mvs::F32C1, "triangulatePixels camera");
        fl.push_back(f->ptr<float>());
        cams.insert(cams.end(), cam->ptr<float>(), cam->ptr<float>() + 16);
    }
    Mat all(depth.rows * depth.cols, 7, mvs::F32C1);
    int n = 0;
    if (mvs_triangulate(ctx, (int)fl.size(), fl.data(), mainCamera.ptr<float>(), cams.data(), depth.ptr<float>(), all.ptr<float>(), &n))
        raise(ctx, "triangulatePixels");
    return all.rowRange(0, n);  // points.resize(pixelId), util.cpp:254
}

// the device half of Heuristic::filterPoints (heuristic.cpp:55-163)
void filterPointsIndices(const Mat &points, float alpha, int32_t *keep, int *kept, int width, int height)
{
    expect(points, mvs::F32C1, "filterPoints points");
    mvs_ctx *ctx = shared_ctx(width, height);
    if (mvs_filter_points(ctx, points.ptr<float>(), points.rows, alpha, keep, kept)) raise(ctx, "filterPoints");
}

// util.cpp:16-29
Mat dehomogenize(Mat points)
{
    Mat result(points.rows, 3, mvs::F32C1);
    for (int i = 0; i < points.rows; i++) {
        const float *inp = points.ptr<float>(i);
        float *out = result.ptr<float>(i);
        out[0] = inp[0] / inp[3];
        out[1] = inp[1] / inp[3];
        out[2] = inp[2] / inp[3];
    }
    return result;
}

// util.cpp:33-41: the reference takes rows 0,1,3 of the 4x4 and asks cv::decomposeProjectionMatrix for the
// camera position, i.e. the null vector of that 3x4 (unit length, sign unspecified).  Closed form: the signed
// 3x3 minors.  Every caller divides by the 4th component (heuristic.cpp:293-295,317-320), so scale and sign cancel.
Mat extractCameraCenter(const Mat camera)
{
    const int rows[3] = {0, 1, 3};
    double p[3][4];
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 4; c++) p[r][c] = camera.at<float>(rows[r], c);
    auto det3 = [&](int c0, int c1, int c2) {
        return p[0][c0] * (p[1][c1] * p[2][c2] - p[1][c2] * p[2][c1]) - p[0][c1] * (p[1][c0] * p[2][c2] - p[1][c2] * p[2][c0]) +
               p[0][c2] * (p[1][c0] * p[2][c1] - p[1][c1] * p[2][c0]);
    };
    double c[4] = {det3(1, 2, 3), -det3(0, 2, 3), det3(0, 1, 3), -det3(0, 1, 2)};
    const double n = std::sqrt(c[0] * c[0] + c[1] * c[1] + c[2] * c[2] + c[3] * c[3]);
    Mat T(4, 1, mvs::F32C1);
    for (int i = 0; i < 4; i++) T.at<float>(i, 0) = (float)(n > 0 ? c[i] / n : c[i]);
    return T;
}
