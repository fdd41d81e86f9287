// render_hip_cv.cpp -- the translation unit a maintainer of addam/mesh-reconstruction adds to the REFERENCE tree: it defines, against
// the reference's own recon.hpp and the real cv::Mat, the symbols render_glx.cpp and flow.cpp define today, on top of the C ABI of
// libmvs_hip.so (include/mvs.h).  Build it in place of render_${SYSTEM_OPENGL}.cpp (Makefile:2,16,21,29) and flow.o:
//
//     g++ -c render_hip_cv.cpp -I<reference tree> -I<this repo>/include `pkg-config --cflags opencv`
//     ... -L<this repo>/mesh-reconstruction_amd/lib -lmvs_hip           (and drop -lGL -lGLEW -lX11, render_glx.o, flow.o)
//
//   class RenderHIP : public Render      recon.hpp:93-99      replaces class RenderGLX, render_glx.cpp:19-53
//   Render *spawnRender(Heuristic)       recon.hpp:100        replaces render_glx.cpp:57-62
//   Mat calculateFlow(prev, next, bool)  recon.hpp:40         replaces flow.cpp:19-42
// With -DMVS_HIP_UTIL it also defines compare, flowRemap, mixBackground and triangulatePixels (recon.hpp:44-50) for a maintainer
// who deletes those four definitions from util.cpp (util.cpp:167-403); recon.cpp itself is not touched in either case.
// This image has no OpenCV, so the file is only COMPILED here, against the declarations-only header tests/cv_decl (tests/
// test_host_cpu.py); the same calls are exercised at run time through host/render_hip.cpp, its twin written against mvs::Mat.
#include "recon.hpp"  // the reference's header

#include <cassert>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include <mvs.h>

namespace {

// reference style: unrecoverable set-up errors print and exit(1) (configuration.cpp:136,141; recon.cpp:49)
void die(const char *what, mvs_ctx *ctx)
{
    fprintf(stderr, "%s: %s\n", what, mvs_last_error(ctx));
    exit(1);
}

void check(int rc, const char *what, mvs_ctx *ctx)
{
    if (rc != MVS_OK) die(what, ctx);
}

// a 4x4 CV_32FC1 camera as the 16 row-major floats the C ABI takes (render_glx.cpp:265 uploads it with transpose = GL_TRUE)
const float *cam16(const Mat &camera)
{
    assert(camera.rows == 4 && camera.cols == 4 && camera.type() == CV_32FC1 && camera.isContinuous());
    return camera.ptr<float>();
}

// calculateFlow, compare, ... are free functions without a renderer: they share one context per frame size
mvs_ctx *shared_context(int width, int height)
{
    static mvs_ctx *ctx = NULL;
    if (ctx && (mvs_width(ctx) != width || mvs_height(ctx) != height)) {
        mvs_destroy(ctx);
        ctx = NULL;
    }
    if (!ctx && !(ctx = mvs_create(0, width, height))) die("mvs_create", NULL);
    return ctx;
}

}  // namespace

class RenderHIP : public Render {
public:
    RenderHIP(int width, int height) : width(width), height(height)
    {
        ctx = mvs_create(/*device*/ 0, width, height);
        if (!ctx) die("mvs_create", NULL);
    }
    ~RenderHIP() { mvs_destroy(ctx); }
    void loadMesh(const Mesh mesh)
    {
        assert(mesh.vertices.isContinuous() && mesh.faces.isContinuous());  // render_glx.cpp:231-232
        assert(mesh.vertices.cols == 4 && mesh.vertices.type() == CV_32FC1 && mesh.faces.cols == 3 && mesh.faces.type() == CV_32SC1);
        check(mvs_load_mesh(ctx, mesh.vertices.ptr<float>(), mesh.vertices.rows, mesh.faces.ptr<int>(), mesh.faces.rows), "mvs_load_mesh", ctx);
    }
    Mat projected(const Mat camera, const Mat frame, const Mat projector)
    {
        assert(frame.channels() == 1 && frame.rows == height && frame.cols == width && frame.isContinuous());  // render_glx.cpp:66
        Mat result(height, width, CV_8UC3);
        check(mvs_projected(ctx, cam16(camera), frame.data, cam16(projector), result.data), "mvs_projected", ctx);
        return result;
    }
    Mat depth(const Mat camera) const
    {
        Mat result(height, width, CV_32FC1);
        check(mvs_depth(ctx, cam16(camera), result.ptr<float>()), "mvs_depth", ctx);  // NDC z, empty pixels = backgroundDepth
        return result;
    }

private:
    mvs_ctx *ctx;
    int width, height;
};

Render *spawnRender(Heuristic hint)
{
    cv::Size size = hint.renderSize();
    return new RenderHIP(size.width, size.height);
}

Mat calculateFlow(const Mat prev, const Mat next, bool useFarneback)
{
    assert(prev.type() == CV_8UC1 && next.type() == CV_8UC1 && prev.rows == next.rows && prev.cols == next.cols);
    assert(prev.isContinuous() && next.isContinuous());
    mvs_ctx *ctx = shared_context(prev.cols, prev.rows);
    Mat mixed(prev.rows, prev.cols, CV_32FC4);  // (u, v, variance, 0), flow.cpp:37-41
    check(mvs_flow(ctx, prev.data, next.data, useFarneback ? 1 : 0, mixed.ptr<float>()), "mvs_flow", ctx);
    return mixed;
}

#ifdef MVS_HIP_UTIL
Mat compare(const Mat prev, const Mat next)
{
    assert(prev.type() == CV_8UC1 && next.type() == CV_8UC1 && prev.isContinuous() && next.isContinuous());
    mvs_ctx *ctx = shared_context(prev.cols, prev.rows);
    Mat result(prev.rows, prev.cols, CV_32FC1);
    check(mvs_compare(ctx, prev.data, next.data, result.ptr<float>()), "mvs_compare", ctx);
    return result;
}

Mat flowRemap(const Mat flow, const Mat image)
{
    assert(flow.isContinuous() && image.type() == CV_8UC1 && image.isContinuous());
    mvs_ctx *ctx = shared_context(image.cols, image.rows);
    Mat result(image.rows, image.cols, CV_8UC1);
    check(mvs_flow_remap(ctx, flow.ptr<float>(), flow.channels(), image.data, result.data), "mvs_flow_remap", ctx);
    return result;
}

Mat mixBackground(const Mat image, const Mat background, Mat &depth)
{
    assert(image.type() == CV_8UC3 && background.type() == CV_8UC1 && depth.type() == CV_32FC1);
    mvs_ctx *ctx = shared_context(background.cols, background.rows);
    Mat result(background.rows, background.cols, CV_8UC1);
    check(mvs_mix_background(ctx, image.data, background.data, depth.ptr<float>(), result.data), "mvs_mix_background", ctx);  // mutates depth, util.cpp:380
    return result;
}

Mat triangulatePixels(const MatList flows, const Mat mainCamera, const MatList cameras, const Mat depth)
{
    assert(flows.size() == cameras.size() && depth.type() == CV_32FC1);
    mvs_ctx *ctx = shared_context(depth.cols, depth.rows);
    std::vector<const float *> flowPtrs;
    std::vector<float> cams;
    MatList::const_iterator camera = cameras.begin();
    for (MatList::const_iterator flow = flows.begin(); flow != flows.end(); flow++, camera++) {
        assert(flow->type() == CV_32FC4 && flow->isContinuous());
        flowPtrs.push_back(flow->ptr<float>());
        cams.insert(cams.end(), cam16(*camera), cam16(*camera) + 16);
    }
    Mat all(depth.rows * depth.cols, 7, CV_32FC1);
    int count = 0;
    check(mvs_triangulate(ctx, (int)flowPtrs.size(), flowPtrs.data(), cam16(mainCamera), cams.data(), depth.ptr<float>(), all.ptr<float>(), &count),
          "mvs_triangulate", ctx);
    return all.rowRange(0, count).clone();  // rows (x, y, z, w, nx, ny, nz), util.cpp:327
}
#endif
