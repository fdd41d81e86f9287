// selftest.cpp -- exercises the host mirror (recon.hpp interface) the way recon.cpp uses it; driven by tests/.
//   host_selftest cpu <tracks dir>            RNG known answers, YAML reader, camera centre, filterPoints, getopt
//   host_selftest gpu <tracks dir> <out dir>  spawnRender -> loadMesh -> depth/projected -> mixBackground ->
//                                             compare/flowRemap, chooseCameras; raw outputs for the oracle check
#include <chrono>
#include <getopt.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>

#include "recon.hpp"

static int fails = 0;
#define CHECK(cond, ...)                       \
    do {                                       \
        if (!(cond)) {                         \
            fails++;                           \
            printf("FAIL %s:%d: ", __FILE__, __LINE__); \
            printf(__VA_ARGS__);               \
            printf("\n");                      \
        }                                      \
    } while (0)

static void writeRaw(const std::string &path, const Mat &m)
{
    std::ofstream f(path, std::ios::binary);
    f.write(reinterpret_cast<const char *>(m.data), (std::streamsize)(m.total() * m.elemSize()));
}

static Mesh heightfield(int n, float extent)
{
    Mat v(n * n, 4, mvs::F32C1), f(2 * (n - 1) * (n - 1), 3, mvs::S32C1);
    for (int j = 0; j < n; j++)
        for (int i = 0; i < n; i++) {
            const double x = -extent + 2.0 * extent * i / (n - 1), y = -extent + 2.0 * extent * j / (n - 1);
            float *p = v.ptr<float>(j * n + i);
            p[0] = (float)x;
            p[1] = (float)y;
            p[2] = (float)(-3.0 - 0.4 * std::sin(1.3 * x + 0.7) * std::cos(1.1 * y - 0.2));
            p[3] = 1.f;
        }
    int k = 0;
    for (int j = 0; j < n - 1; j++)
        for (int i = 0; i < n - 1; i++) {
            const int a = j * n + i, b = a + 1, c = a + n, d = c + 1;
            int32_t *t0 = f.ptr<int32_t>(k++), *t1 = f.ptr<int32_t>(k++);
            t0[0] = a; t0[1] = b; t0[2] = c;
            t1[0] = b; t1[1] = d; t1[2] = c;
        }
    return Mesh(v, f);
}

static int run_cpu(const std::string &tracks)
{
    // cv::theRNG() default stream: known answers
    HeuristicRNG rng;
    const float expect[4] = {0.030282794f, 0.6992592f, 0.90105945f, 0.3143851f};
    for (int i = 0; i < 4; i++) {
        const float u = rng.uniform();
        CHECK(std::fabs(u - expect[i]) < 1e-7f, "rng[%d] = %.9g, expected %.9g", i, u, expect[i]);
    }
    {
        // a colour frame without -e goes through cvtColor(BGR2GRAY) (configuration.cpp:244): OpenCV's 8-bit fixed-point weights
        // give 29 / 150 / 76 for pure blue / green / red and leave greys unchanged
        Configuration c(tracks + "/koule-tr.yaml");
        Mat bgr = Mat::zeros(c.height, c.width, mvs::U8C3);
        const uint8_t px[4][3] = {{255, 0, 0}, {0, 255, 0}, {0, 0, 255}, {200, 200, 200}};
        for (int k = 0; k < 4; k++)
            for (int ch = 0; ch < 3; ch++) bgr.at<uint8_t>(0, 3 * k + ch) = px[k][ch];
        c.setFrameColor(0, bgr);
        const Mat g = c.frame(0);
        CHECK(g.at<uint8_t>(0, 0) == 29 && g.at<uint8_t>(0, 1) == 150 && g.at<uint8_t>(0, 2) == 76 && g.at<uint8_t>(0, 3) == 200,
              "BGR2GRAY gave %d %d %d %d", g.at<uint8_t>(0, 0), g.at<uint8_t>(0, 1), g.at<uint8_t>(0, 2), g.at<uint8_t>(0, 3));
    }
    // YAML reader on the four bundled calibration files
    struct { const char *name; int w, h, frames, bundles; } files[] = {
        {"koberec-.yaml", 640, 480, 55, 30}, {"koberec.yaml", 640, 480, 173, 18}, {"koule-tr.yaml", 640, 480, 31, 21}, {"zatisi.yaml", 640, 480, 120, 23}};
    for (auto &f : files) {
        Configuration c(tracks + "/" + f.name);
        CHECK(c.width == f.w && c.height == f.h, "%s size %dx%d", f.name, c.width, c.height);
        CHECK(c.frameCount() == f.frames, "%s frames %d", f.name, c.frameCount());
        CHECK(c.reconstructedPoints().rows == f.bundles, "%s bundles %d", f.name, c.reconstructedPoints().rows);
        // projection[2][2] = -(far+near)/(far-near), [2][3] = 2 far near/(near-far) for the identity-pose frame 1
        const Mat P = c.camera(0);
        const float n = c.nearVal(0), fa = c.farVal(0);
        CHECK(std::fabs(std::fabs(P.at<float>(2, 2)) - (fa + n) / (fa - n)) < 2e-3f, "%s P22 %g", f.name, P.at<float>(2, 2));
        CHECK(std::fabs(P.at<float>(1, 1) / P.at<float>(0, 0) - 4.f / 3.f) < 1e-3f, "%s aspect", f.name);
        // every camera centre is annihilated by rows 0,1,3 of its matrix
        for (int i = 0; i < c.frameCount(); i += 7) {
            const Mat cam = c.camera(i), ctr = extractCameraCenter(cam);
            const Mat s = mvs::matmul(cam, ctr);
            CHECK(std::fabs(s.at<float>(0, 0)) < 1e-4f && std::fabs(s.at<float>(1, 0)) < 1e-4f && std::fabs(s.at<float>(3, 0)) < 1e-4f,
                  "%s centre of camera %d", f.name, i);
        }
    }
    {   // skipFrames re-indexing (configuration.cpp:186-187, 210-212)
        Configuration c(tracks + "/zatisi.yaml", 4);
        CHECK(c.frameCount() == 30, "skip 4 -> %d frames", c.frameCount());
        Configuration full(tracks + "/zatisi.yaml");
        CHECK(std::memcmp(c.camera(3).data, full.camera(12).data, 64) == 0, "skip re-index");
    }
    {   // command line (configuration.cpp:35-131)
        std::string y = tracks + "/koule-tr.yaml";
        const char *argv[] = {"recon", "-n", "3", "-c", "7.5", "-f", "-v", "-o", "x.obj", y.c_str()};
        Configuration c(10, const_cast<char **>(argv));
        CHECK(c.iterationCount == 3 && c.cameraThreshold == 7.5f && c.useFarneback && c.verbosity == 2 && c.outFileName == "x.obj", "getopt");
        CHECK(c.sweepPlanes == 0 && c.threads == 1, "the two long options the reference does not have default to the reference's behaviour");
        const char *argv2[] = {"recon", "--sweep-planes", "48", "--threads", "4", "--input", y.c_str()};
        Configuration c2(7, const_cast<char **>(argv2));
        CHECK(c2.sweepPlanes == 48 && c2.threads == 4 && !c2.useFarneback && c2.iterationCount == 2, "--sweep-planes / --threads");
        Heuristic h(&c);
        CHECK(h.notHappy(Mat()) && h.notHappy(Mat()) && h.notHappy(Mat()) && !h.notHappy(Mat()), "notHappy counts iterations");
        CHECK(h.renderSize().width == 640 && h.renderSize().height == 480, "renderSize");
        bool threw = false;
        try {
            const char *bad[] = {"recon"};
            Configuration b(1, const_cast<char **>(bad));
        } catch (const std::exception &) { threw = true; }
        CHECK(threw, "missing YAML must throw");
    }
    {
        // Heuristic::tessellate: dispatch and alphaVals bookkeeping (heuristic.cpp:525-545) with stand-in meshers
        Configuration c(tracks + "/zatisi.yaml");
        Heuristic h(&c);
        int alphaCalls = 0, poissonCalls = 0, readCalls = 0;
        h.meshers.alphaShapeFaces = [&](const Mat pts, float *alpha) {
            alphaCalls++;
            *alpha = 0.75f;
            Mat f(1, 3, mvs::S32C1);
            f.at<int32_t>(0, 0) = 0; f.at<int32_t>(0, 1) = 1; f.at<int32_t>(0, 2) = 2;
            return f;
        };
        h.meshers.poissonSurface = [&](const Mat pts, const Mat) {
            poissonCalls++;
            return Mesh(pts, Mat(0, 3, mvs::S32C1));
        };
        h.meshers.readMesh = [&](const char *) {
            readCalls++;
            return Mesh(Mat(3, 4, mvs::F32C1), Mat(1, 3, mvs::S32C1));
        };
        Mat pts = c.reconstructedPoints(), nrm(pts.rows, 3, mvs::F32C1);
        bool threw = false;
        try {
            Mat p2 = pts.clone(), n2 = nrm.clone();
            h.filterPoints(p2, n2);
        } catch (const std::exception &) { threw = true; }
        CHECK(threw, "filterPoints before tessellate must fail: no alpha value yet");
        CHECK(h.notHappy(pts), "first iteration");                    // iteration = 1 (recon.cpp:27)
        Mesh m1 = h.tessellate(pts, nrm);                             // alpha shapes
        CHECK(alphaCalls == 1 && h.alphaVals.size() == 1 && h.alphaVals.back() == 0.75f && m1.faces.rows == 1 && m1.vertices.rows == pts.rows, "tessellate: alpha shapes on the first iteration");
        CHECK(h.notHappy(pts), "second iteration");                   // iteration = 2
        (void)h.tessellate(pts, nrm);                                 // Poisson, alpha halved
        CHECK(poissonCalls == 1 && h.alphaVals.size() == 2 && h.alphaVals.back() == 0.375f, "tessellate: Poisson surface halves alpha");
        Configuration c2(tracks + "/zatisi.yaml");
        c2.inMeshFile = "some.obj";
        Heuristic h2(&c2);
        h2.meshers = h.meshers;
        (void)h2.notHappy(pts);
        (void)h2.tessellate(pts, nrm);
        CHECK(readCalls == 1 && h2.alphaVals.size() == 1 && h2.alphaVals.back() == 1.f, "tessellate: initial mesh file gives alpha 1");
        // no meshers installed: the library's own alpha shapes (host/alpha_shapes.cpp) of the bundle's points, as recon.cpp:36 does
        Heuristic h3(&c);
        (void)h3.notHappy(pts);
        Mesh m3 = h3.tessellate(pts, nrm);
        bool ok = m3.faces.rows > 0 && m3.vertices.rows == pts.rows && h3.alphaVals.size() == 1 && h3.alphaVals.back() > 0.f;
        for (int i = 0; ok && i < m3.faces.rows; i++)
            for (int k = 0; k < 3; k++) ok = ok && m3.faces.at<int32_t>(i, k) >= 0 && m3.faces.at<int32_t>(i, k) < pts.rows;
        CHECK(ok, "tessellate with the built-in alpha shapes: %d faces over %d points, alpha %g", m3.faces.rows, pts.rows, h3.alphaVals.empty() ? -1.0 : (double)h3.alphaVals.back());
        printf("alpha shape of the zatisi bundle: %d points, %d faces, alpha %g\n", pts.rows, m3.faces.rows, (double)h3.alphaVals.back());
    }
    printf("cpu selftest: %d failures\n", fails);
    return fails ? 1 : 0;
}

static int run_gpu(const std::string &tracks, const std::string &out)
{
    Configuration config(tracks + "/zatisi.yaml");
    Heuristic hint(&config);
    Render *render = spawnRender(hint);  // recon.cpp:21
    const Mesh mesh = heightfield(40, 1.6f);
    render->loadMesh(mesh);  // recon.cpp:42
    const int W = config.width, H = config.height;
    // synthetic frames (the clip is missing): a smooth pattern per frame index
    for (int fi = 0; fi < config.frameCount(); fi++) {
        Mat g(H, W, mvs::U8C1);
        for (int y = 0; y < H; y++)
            for (int x = 0; x < W; x++) g.at<uint8_t>(y, x) = (uint8_t)(127 + 100 * std::sin((x + 3 * fi) / 23.0) * std::cos((y - fi) / 17.0));
        config.setFrame(fi, g);
    }
    auto t0 = std::chrono::steady_clock::now();
    const int cameraCount = hint.chooseCameras(mesh, config.allCameras(), *render);  // recon.cpp:46
    const double ms_probe = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    printf("chooseCameras: %d pairs, %zu main cameras\n", cameraCount, hint.chosen().size());
    CHECK(cameraCount > 0, "heuristic chose no cameras");
    {
        // the same 200 shots through the reference's interface only (whole depth maps): identical choices
        struct PlainRender : Render {
            Render *inner;
            explicit PlainRender(Render *r) : inner(r) {}
            void loadMesh(const Mesh m) override { inner->loadMesh(m); }
            Mat projected(const Mat c, const Mat f, const Mat p) override { return inner->projected(c, f, p); }
            Mat depth(const Mat c) const override { return inner->depth(c); }
        } plain(render);
        Heuristic hint2(&config);
        t0 = std::chrono::steady_clock::now();
        const int count2 = hint2.chooseCameras(mesh, config.allCameras(), plain);
        const double ms_plain = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        CHECK(count2 == cameraCount && hint2.chosen() == hint.chosen(), "chooseCameras differs between probed and whole-map depth");
        printf("chooseCameras (200 shots): %.1f ms with depth probes, %.1f ms with whole depth maps\n", ms_probe, ms_plain);
    }
    {
        // recon.cpp:136 on a later iteration: Heuristic::tessellate -> the library's Poisson surface (csrc/poisson.hip) of oriented samples
        Heuristic hp(&config);
        const int n = 20000;
        Mat pts(n, 4, mvs::F32C1), nrm(n, 3, mvs::F32C1);
        for (int i = 0; i < n; i++) {  // a Fibonacci sphere of radius 0.7 around (0.1, -0.2, 3)
            const double z = 1.0 - 2.0 * (i + 0.5) / n, r = std::sqrt(1.0 - z * z), phi = i * 2.399963229728653;
            const double d[3] = {r * std::cos(phi), r * std::sin(phi), z};
            const double c[3] = {0.1, -0.2, 3.0};
            for (int k = 0; k < 3; k++) pts.at<float>(i, k) = (float)(2.0 * (c[k] + 0.7 * d[k])), nrm.at<float>(i, k) = (float)d[k];
            pts.at<float>(i, 3) = 2.0f;  // homogeneous rows, w != 1
        }
        (void)hp.notHappy(pts);
        (void)hp.notHappy(pts);
        hp.alphaVals.push_back(0.5f);
        const Mesh m = hp.tessellate(pts, nrm);
        double worst = 0.0, vol = 0.0;
        for (int i = 0; i < m.vertices.rows; i++) {
            const float *v = m.vertices.ptr<float>(i);
            const double dx = v[0] / v[3] - 0.1, dy = v[1] / v[3] + 0.2, dz = v[2] / v[3] - 3.0;
            worst = std::max(worst, std::fabs(std::sqrt(dx * dx + dy * dy + dz * dz) - 0.7));
        }
        for (int i = 0; i < m.faces.rows; i++) {
            const float *a = m.vertices.ptr<float>(m.faces.at<int32_t>(i, 0)), *b = m.vertices.ptr<float>(m.faces.at<int32_t>(i, 1)), *c = m.vertices.ptr<float>(m.faces.at<int32_t>(i, 2));
            vol += (a[0] * (b[1] * c[2] - b[2] * c[1]) - a[1] * (b[0] * c[2] - b[2] * c[0]) + a[2] * (b[0] * c[1] - b[1] * c[0])) / 6.0;
        }
        // cgal_poisson.cpp:50: sm_angle = 20 -- the smallest angle of any facet (vertices have w = 1 here)
        double max_cos = -1.0;
        for (int i = 0; i < m.faces.rows; i++) {
            const float *q[3] = {m.vertices.ptr<float>(m.faces.at<int32_t>(i, 0)), m.vertices.ptr<float>(m.faces.at<int32_t>(i, 1)), m.vertices.ptr<float>(m.faces.at<int32_t>(i, 2))};
            for (int k = 0; k < 3; k++) {
                double u[3], w[3], uu = 0.0, ww = 0.0, uw = 0.0;
                for (int c = 0; c < 3; c++) {
                    u[c] = (double)q[(k + 1) % 3][c] - (double)q[k][c], w[c] = (double)q[(k + 2) % 3][c] - (double)q[k][c];
                    uu += u[c] * u[c], ww += w[c] * w[c], uw += u[c] * w[c];
                }
                max_cos = std::max(max_cos, uw / std::sqrt(uu * ww));
            }
        }
        const double min_angle = std::acos(std::min(1.0, max_cos)) * 180.0 / 3.14159265358979;
        const double sphere = 4.0 / 3.0 * 3.14159265358979 * 0.7 * 0.7 * 0.7;
        printf("poissonSurface: %d vertices, %d faces, worst radial error %.4f, volume %.4f (sphere %.4f), smallest facet angle %.2f degrees\n", m.vertices.rows,
               m.faces.rows, worst, vol, sphere, min_angle);
        // (poissonSurface ends with the simplification pass: a few hundred facets where the grid made tens of thousands; its chords lie inside the sphere)
        CHECK(m.faces.rows > 200 && worst < 0.02 && std::fabs(vol - sphere) < 0.03 * sphere && hp.alphaVals.back() == 0.25f, "tessellate with the built-in Poisson surface");
        CHECK(min_angle >= 20.0 - 1e-4, "poissonSurface keeps the reference's angle bound (cgal_poisson.cpp:50)");
        // which default is intended: the two-argument call normalises the normals -- lengths varying over two decades give the
        // mesh of unit normals, byte for byte -- and POISSON_CONFIDENCE_NORMALS (the reference's semantics) uses them: another mesh
        Mat scaled(n, 3, mvs::F32C1);
        for (int i = 0; i < n; i++) {
            const float s = std::ldexp(1.0f, -(i % 7)) * (1.0f + 0.25f * (float)(i % 3));
            for (int k = 0; k < 3; k++) scaled.at<float>(i, k) = nrm.at<float>(i, k) * s;
        }
        CHECK(poissonNormals() == POISSON_UNIT_NORMALS, "the default of poissonSurface(points, normals) is unit normals");
        setPoissonSimplify(false);  // (vertex-for-vertex comparisons of meshes whose fields differ in the last float bit: the grid's mesh, not the simplified one)
        const Mesh unit_mesh = poissonSurface(pts, nrm), scaled_mesh = poissonSurface(pts, scaled), conf_mesh = poissonSurface(pts, scaled, POISSON_CONFIDENCE_NORMALS);
        bool same = unit_mesh.vertices.rows == scaled_mesh.vertices.rows && unit_mesh.faces.rows == scaled_mesh.faces.rows;
        double dmax = 0.0;
        for (int i = 0; same && i < unit_mesh.vertices.rows; i++)
            for (int k = 0; k < 4; k++) dmax = std::max(dmax, (double)std::fabs(unit_mesh.vertices.at<float>(i, k) - scaled_mesh.vertices.at<float>(i, k)));
        CHECK(same && dmax < 1e-4, "poissonSurface normalises the normals by default");
        setPoissonNormals(POISSON_CONFIDENCE_NORMALS);
        const Mesh conf2 = poissonSurface(pts, scaled);
        setPoissonNormals(POISSON_UNIT_NORMALS);
        CHECK(conf2.vertices.rows == conf_mesh.vertices.rows && conf2.faces.rows == conf_mesh.faces.rows, "setPoissonNormals switches the two-argument call");
        CHECK(conf_mesh.vertices.rows != unit_mesh.vertices.rows || conf_mesh.faces.rows != unit_mesh.faces.rows, "confidence normals give another mesh than unit normals");
        setPoissonSimplify(true);
    }
    std::ofstream sel(out + "/chosen.txt");
    int mains = 0, pairs = 0;
    for (int fa = hint.beginMain(); fa != Heuristic::sentinel; fa = hint.nextMain()) {  // recon.cpp:65
        sel << fa << ":";
        mains++;
        for (int fb = hint.beginSide(fa); fb != Heuristic::sentinel; fb = hint.nextSide(fa)) {  // recon.cpp:81
            sel << " " << fb;
            pairs++;
            CHECK(fb != fa && fb >= 0 && fb < config.frameCount(), "side camera index");
        }
        sel << "\n";
    }
    CHECK(mains == (int)hint.chosen().size() && pairs >= mains, "iterator protocol");
    // one (main, side) pair exactly as recon.cpp:69-89 runs it
    const int fa = hint.beginMain();
    const int fb = hint.beginSide(fa);
    Mat originalImage = config.frame(fa);
    Mat depth = render->depth(config.camera(fa));
    writeRaw(out + "/depth.f32", depth);
    Mat projectedImage = render->projected(config.camera(fa), config.frame(fb), config.camera(fb));
    writeRaw(out + "/projected.u8", projectedImage);
    Mat mixed = mixBackground(projectedImage, originalImage, depth);
    writeRaw(out + "/mixed.u8", mixed);
    writeRaw(out + "/depth_after_mix.f32", depth);
    Mat var = compare(originalImage, mixed);
    writeRaw(out + "/compare.f32", var);
    Mat flow = Mat::zeros(H, W, mvs::F32C4);
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            flow.at<float>(y, 4 * x) = 1.5f * std::sin(y / 40.0f);
            flow.at<float>(y, 4 * x + 1) = -0.75f;
        }
    writeRaw(out + "/remap.u8", flowRemap(flow, mixed));
    {
        std::ofstream meta(out + "/meta.txt");
        meta << fa << " " << fb << " " << W << " " << H << " " << mesh.faces.rows << "\n";
        writeRaw(out + "/mesh_verts.f32", mesh.vertices);
        writeRaw(out + "/mesh_faces.i32", mesh.faces);
        writeRaw(out + "/frame_a.u8", originalImage);
        writeRaw(out + "/frame_b.u8", config.frame(fb));
        writeRaw(out + "/cam_a.f32", config.camera(fa));
        writeRaw(out + "/cam_b.f32", config.camera(fb));
        writeRaw(out + "/flow.f32", flow);
    }
    {   // recon.cpp:89-116: flows for every side view of this main frame, then triangulatePixels
        MatList flows, cameras;
        Mat d2 = render->depth(config.camera(fa));
        int used = 0;
        for (int s = hint.beginSide(fa); s != Heuristic::sentinel && used < 2; s = hint.nextSide(fa), used++) {
            Mat proj = render->projected(config.camera(fa), config.frame(s), config.camera(s));
            Mat mix = mixBackground(proj, originalImage, d2);
            flows.push_back(calculateFlow(originalImage, mix, config.useFarneback));
            cameras.push_back(config.camera(s));
        }
        Mat tri = triangulatePixels(flows, config.camera(fa), cameras, d2);
        printf("triangulatePixels: %d points from %d side views\n", tri.rows, used);
        CHECK(tri.cols == 7 && tri.rows > 0, "triangulatePixels produced %d x %d", tri.rows, tri.cols);
        writeRaw(out + "/tri.f32", tri);
        writeRaw(out + "/tri_depth.f32", d2);
        int idx = 0;
        for (const Mat &f : flows) writeRaw(out + "/tri_flow" + std::to_string(idx++) + ".f32", f);
        idx = 0;
        for (const Mat &c : cameras) writeRaw(out + "/tri_cam" + std::to_string(idx++) + ".f32", c);
        std::ofstream(out + "/tri_meta.txt") << used << " " << tri.rows << "\n";
    }
    {   // filterPoints: a dense cluster survives thinned, isolated outliers go (heuristic.cpp:55-176)
        Heuristic h(&config);
        h.alphaVals.push_back(0.16f);  // radius 0.04 -> reach 0.2
        HeuristicRNG r;
        const int N = 400;
        Mat pts(N + 5, 4, mvs::F32C1), nrm = Mat::zeros(N + 5, 3, mvs::F32C1);
        for (int i = 0; i < N; i++) {
            float *p = pts.ptr<float>(i);
            p[0] = r.uniform(); p[1] = r.uniform(); p[2] = 0.05f * r.uniform(); p[3] = 1.f;
        }
        for (int i = 0; i < 5; i++) {
            float *p = pts.ptr<float>(N + i);
            p[0] = 10.f + 3 * i; p[1] = -7.f; p[2] = 4.f; p[3] = 1.f;
        }
        h.filterPoints(pts, nrm);
        CHECK(pts.rows > 10 && pts.rows < N, "filterPoints kept %d of %d", pts.rows, N + 5);
        bool outlier = false;
        for (int i = 0; i < pts.rows; i++) outlier |= pts.at<float>(i, 0) > 5.f;
        CHECK(!outlier, "isolated points must be removed");
        CHECK(nrm.rows == pts.rows, "normals follow points");
    }
    bool threw = false;
    try {
        render->projected(config.camera(fa), Mat(10, 10, mvs::U8C1), config.camera(fb));
    } catch (const std::exception &) { threw = true; }
    CHECK(threw, "wrong frame size must throw");
    delete render;  // recon.cpp:130
    printf("gpu selftest: %d failures\n", fails);
    return fails ? 1 : 0;
}

// `-e`: Configuration::estimateExposure on the colour frames found next to the clip (<clip>.frames/%06d.ppm); dumps the
// exposure matrix (channels x frames, f32) and the normalised grey frames for tests/test_host_cpu.py
static int run_exposure(const std::string &yaml, const std::string &out)
{
    std::string a0 = "recon", a1 = "-e", a2 = yaml;
    char *argv[] = {&a0[0], &a1[0], &a2[0], nullptr};
    optind = 1;
    Configuration config(3, argv);
    if (config.exposure.empty()) {
        printf("exposure selftest: the colour frames are incomplete, nothing estimated\n");
        return 1;
    }
    writeRaw(out + "/exposure.f32", config.exposure);
    for (int i = 0; i < config.frameCount(); i++) {
        char name[64];
        snprintf(name, sizeof(name), "/gray_%03d.u8", i);
        writeRaw(out + name, config.frame(i));
    }
    printf("exposure selftest: %d channels x %d frames\n", config.exposure.rows, config.exposure.cols);
    return 0;
}

// host_selftest frames <tracks yaml> <out dir> [skip]: the grey frames Configuration ends up with (clip decoding -> resize -> BGR2GRAY)
static int run_frames(const std::string &yaml, const std::string &out, int skip)
{
    Configuration config(yaml, skip);
    int have = 0;
    for (int i = 0; i < config.frameCount(); i++) {
        Mat g;
        try { g = config.frame(i); } catch (const std::exception &) { continue; }
        char name[64];
        snprintf(name, sizeof(name), "/gray_%03d.u8", i);
        writeRaw(out + name, g);
        have++;
    }
    printf("frames selftest: %d of %d frames, %d x %d\n", have, config.frameCount(), config.width, config.height);
    return 0;
}

// host_selftest choose <tracks yaml> <verts.f32> <faces.i32> <camera threshold> <out.txt> [nodepth]
// Heuristic::chooseCameras on a mesh from raw files, chosen schedule written as "main: side side ..." lines preceded by the pair
// count and followed by the generator's state -- compared pair for pair with tests/policy_mirror.py.  With `nodepth` the renderer is
// a stand-in that sees no geometry (every depth = backgroundDepth), which needs no GPU: the policy arithmetic alone.
static int run_choose(int argc, char **argv)
{
    Configuration config(argv[2]);
    config.cameraThreshold = (float)atof(argv[5]);
    auto slurp = [](const char *path) {
        std::ifstream f(path, std::ios::binary);
        return std::vector<char>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    };
    const std::vector<char> vb = slurp(argv[3]), fb = slurp(argv[4]);
    Mesh mesh(Mat((int)(vb.size() / 16), 4, mvs::F32C1), Mat((int)(fb.size() / 12), 3, mvs::S32C1));
    std::memcpy(mesh.vertices.data, vb.data(), vb.size());
    std::memcpy(mesh.faces.data, fb.data(), fb.size());
    Heuristic hint(&config);
    struct BlindRender : Render {
        int w, h;
        BlindRender(int w_, int h_) : w(w_), h(h_) {}
        void loadMesh(const Mesh) override {}
        Mat projected(const Mat, const Mat, const Mat) override { return Mat(); }
        Mat depth(const Mat) const override
        {
            Mat d(h, w, mvs::F32C1);
            for (size_t i = 0; i < d.total(); i++) d.ptr<float>()[i] = backgroundDepth;
            return d;
        }
    };
    Render *render = (argc > 7 && !strcmp(argv[7], "nodepth")) ? (Render *)new BlindRender(config.width, config.height) : spawnRender(hint);
    render->loadMesh(mesh);
    const int count = hint.chooseCameras(mesh, config.allCameras(), *render);
    std::ofstream out(argv[6]);
    out << count << "\n";
    for (const auto &entry : hint.chosen()) {
        out << entry.first << ":";
        for (int s : entry.second) out << " " << s;
        out << "\n";
    }
    out << "rng " << hint.rng.state << "\n";
    delete render;
    return 0;
}

// host_selftest sweep <tracks yaml> <verts.f32> <faces.i32> <out dir> <planes> <farneback 0|1> <main frame> <side frame> [<side frame> ...]
// The sweep behind the C++ seam (recon.hpp: DepthSweep, trackMainFrame): the frames come from the YAML's `<clip>.frames` directory, the proxy
// mesh from raw files.  Writes (a) RenderHIP::sweepDepth of the main frame against the side frames over the whole NDC range -- compared bit for
// bit with mvs_sweep through the Python binding; (b) trackMainFrame's point rows and the depth map it used, with Configuration::sweepPlanes = 0
// (the reference's path) and = <planes> (the swept depth) -- compared with the surface the frames were rendered from (tests/test_host_gpu.py).
static int run_sweep(int argc, char **argv)
{
    Configuration config(argv[2]);
    const std::string out = argv[5];
    const int planes = atoi(argv[6]);
    config.useFarneback = atoi(argv[7]) != 0;
    const int fa = atoi(argv[8]);
    std::vector<int> sides;
    for (int i = 9; i < argc; i++) sides.push_back(atoi(argv[i]));
    auto slurp = [](const char *path) {
        std::ifstream f(path, std::ios::binary);
        return std::vector<char>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    };
    const std::vector<char> vb = slurp(argv[3]), fb = slurp(argv[4]);
    Mesh mesh(Mat((int)(vb.size() / 16), 4, mvs::F32C1), Mat((int)(fb.size() / 12), 3, mvs::S32C1));
    std::memcpy(mesh.vertices.data, vb.data(), vb.size());
    std::memcpy(mesh.faces.data, fb.data(), fb.size());
    Heuristic hint(&config);
    Render *render = spawnRender(hint);
    render->loadMesh(mesh);
    DepthSweep *sweeper = dynamic_cast<DepthSweep *>(render);
    CHECK(sweeper != nullptr, "spawnRender's renderer implements DepthSweep");
    if (!sweeper) return 1;
    // (a) the extension itself
    CHECK(sweeper->storeCapacity() == 0 && !sweeper->frameStored(fa), "an empty store");
    sweeper->storeFrames(config.frameCount());
    std::vector<Mat> sideCameras;
    for (int s : sides) {
        sweeper->storeFrame(s, config.frame(s));
        sideCameras.push_back(config.camera(s));
    }
    bool threw = false;
    try { sweeper->sweepDepth(fa, config.camera(fa), sides, sideCameras, planes); } catch (const std::exception &) { threw = true; }
    CHECK(threw, "sweepDepth refuses a main frame that is not in the store");
    sweeper->storeFrame(fa, config.frame(fa));
    Mat cost;
    const Mat whole = sweeper->sweepDepth(fa, config.camera(fa), sides, sideCameras, planes, -1.f, 1.f, &cost);
    CHECK(whole.rows == config.height && whole.cols == config.width && cost.rows == config.height, "sweepDepth's maps have the render size");
    writeRaw(out + "/sweep_depth.f32", whole);
    writeRaw(out + "/sweep_cost.f32", cost);
    // (b) the driver's loop body, both ways
    for (int pass = 0; pass < 2; pass++) {
        config.sweepPlanes = pass ? planes : 0;
        Mat used;
        const auto t0 = std::chrono::steady_clock::now();
        const Mat tri = trackMainFrame(config, render, fa, sides, &used);
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        printf("trackMainFrame(sweepPlanes = %d): %d points, %.2f ms\n", config.sweepPlanes, tri.rows, ms);
        CHECK(tri.cols == 7 && tri.rows > 0, "trackMainFrame produced %d x %d", tri.rows, tri.cols);
        const std::string tag = pass ? "swept" : "proxy";
        writeRaw(out + "/points_" + tag + ".f32", tri);
        writeRaw(out + "/depth_" + tag + ".f32", used);
        std::ofstream(out + "/meta_" + tag + ".txt") << tri.rows << "\n";
    }
    delete render;
    printf("%s\n", fails ? "SELFTEST FAILED" : "sweep selftest OK");
    return fails ? 1 : 0;
}

// host_selftest sequence <tracks yaml> <verts.f32> <faces.i32> <threads> <farneback 0|1> <main frame> [<main frame> ...]
// trackMainFrames (host/driver.cpp) over a schedule of main frames, each with its neighbours at -10, -5, +5, +10 frames (frames from the YAML's
// `<clip>.frames` directory): once on one thread and once on <threads> threads with a renderer and contexts each -- the point blocks must be the same
// bytes in the same order whatever the thread count; prints both times.
static int run_sequence(int argc, char **argv)
{
    Configuration config(argv[2]);
    const int threads = atoi(argv[5]);
    config.useFarneback = atoi(argv[6]) != 0;
    auto slurp = [](const char *path) {
        std::ifstream f(path, std::ios::binary);
        return std::vector<char>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    };
    const std::vector<char> vb = slurp(argv[3]), fb = slurp(argv[4]);
    Mesh mesh(Mat((int)(vb.size() / 16), 4, mvs::F32C1), Mat((int)(fb.size() / 12), 3, mvs::S32C1));
    std::memcpy(mesh.vertices.data, vb.data(), vb.size());
    std::memcpy(mesh.faces.data, fb.data(), fb.size());
    std::vector<numberedVector> schedule;
    for (int i = 7; i < argc; i++) {
        const int fa = atoi(argv[i]);
        std::vector<int> sides;
        for (int o : {-10, -5, 5, 10}) sides.push_back(std::min(config.frameCount() - 1, std::max(0, fa + o)));
        schedule.emplace_back(fa, sides);
    }
    Heuristic hint(&config);
    Render *render = spawnRender(hint);
    render->loadMesh(mesh);
    std::vector<Mat> ref;
    double ms[2] = {0, 0};
    for (int pass = 0; pass < 2; pass++) {
        config.threads = pass ? threads : 1;
        trackMainFrames(config, hint, render, mesh, schedule);  // warm-up: contexts, arenas
        const auto t0 = std::chrono::steady_clock::now();
        const std::vector<Mat> blocks = trackMainFrames(config, hint, render, mesh, schedule);
        ms[pass] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / (double)schedule.size();
        if (!pass) {
            ref = blocks;
            continue;
        }
        CHECK(blocks.size() == ref.size(), "one block per main frame");
        for (size_t i = 0; i < blocks.size() && i < ref.size(); i++) {
            CHECK(blocks[i].rows == ref[i].rows && blocks[i].rows > 0, "main frame %d: %d points on %d threads, %d on one", schedule[i].first, blocks[i].rows, threads, ref[i].rows);
            if (blocks[i].rows == ref[i].rows)
                CHECK(!std::memcmp(blocks[i].data, ref[i].data, blocks[i].total() * blocks[i].elemSize()), "main frame %d: the points differ between 1 and %d threads", schedule[i].first, threads);
        }
    }
    // and the first main frame through the reference's own calls, stage by stage with host frames (recon.cpp:65-117 as written): trackMainFrame took its
    // frames from the renderer's frame store and ran the whole body in one device-resident call -- the same rows, bit for bit
    if (!schedule.empty() && !ref.empty()) {
        const int fa = schedule[0].first;
        const Mat originalImage = config.frame(fa), mainCamera = config.camera(fa);
        Mat depth = render->depth(mainCamera);
        MatList flows, cameras;
        for (int fb2 : schedule[0].second) {
            Mat projectedImage = render->projected(mainCamera, config.frame(fb2), config.camera(fb2));
            projectedImage = mixBackground(projectedImage, originalImage, depth);
            flows.push_back(calculateFlow(originalImage, projectedImage, config.useFarneback));
            cameras.push_back(config.camera(fb2));
        }
        const Mat rows = triangulatePixels(flows, mainCamera, cameras, depth);
        CHECK(rows.rows == ref[0].rows, "main frame %d: %d points stage by stage, %d from the frame store", fa, rows.rows, ref[0].rows);
        if (rows.rows == ref[0].rows) CHECK(!std::memcmp(rows.data, ref[0].data, rows.total() * rows.elemSize()), "main frame %d: stage-by-stage points differ from the stored-frame call's", fa);
    }
    printf("trackMainFrames, %zu main frames, %s flow: %.2f ms per main frame on one thread, %.2f ms on %d threads\n", schedule.size(), config.useFarneback ? "Farneback" : "variational",
           ms[0], ms[1], threads);
    delete render;
    printf("%s\n", fails ? "SELFTEST FAILED" : "sequence selftest OK");
    return fails ? 1 : 0;
}

int main(int argc, char **argv)
{
    try {
        if (argc >= 8 && !strcmp(argv[1], "sequence")) return run_sequence(argc, argv);
        if (argc >= 10 && !strcmp(argv[1], "sweep")) return run_sweep(argc, argv);
        if (argc >= 4 && !strcmp(argv[1], "exposure")) return run_exposure(argv[2], argv[3]);
        if (argc >= 4 && !strcmp(argv[1], "frames")) return run_frames(argv[2], argv[3], argc > 4 ? atoi(argv[4]) : 1);
        if (argc >= 7 && !strcmp(argv[1], "choose")) return run_choose(argc, argv);
        if (argc >= 3 && !strcmp(argv[1], "cpu")) return run_cpu(argv[2]);
        if (argc >= 4 && !strcmp(argv[1], "gpu")) return run_gpu(argv[2], argv[3]);
    } catch (const std::exception &e) {
        printf("exception: %s\n", e.what());
        return 2;
    }
    printf("usage: host_selftest cpu <tracks dir> | gpu <tracks dir> <out dir> | exposure <tracks yaml> <out dir> | sweep <tracks yaml> <verts> <faces> <out dir> <planes> <farneback> <main> <sides...>\n");
    return 64;
}
