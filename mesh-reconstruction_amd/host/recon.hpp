// recon.hpp -- host-side mirror of the reference's public interface for the hot path (recon.hpp:17-123 of
// addam/mesh-reconstruction), backed by libmvs_hip.so (include/mvs.h).
//
// Same names, argument meaning and ownership rules as the reference so that code written against the
// reference's header (recon.cpp:12-141) reads the same here:
//   Mat / Mesh / MatList / backgroundDepth            recon.hpp:17-30
//   class Render { loadMesh, projected, depth }       recon.hpp:93-99     -> RenderHIP (render_hip.cpp)
//   Render *spawnRender(Heuristic)                    recon.hpp:100, render_glx.cpp:57-62
//   calculateFlow / compare / mixBackground / flowRemap / extractCameraCenter / dehomogenize   recon.hpp:40-50
//   class Heuristic                                   recon.hpp:104-123, heuristic.cpp
//   class Configuration                               recon.hpp:58-90, configuration.cpp
// Differences, all forced by the missing OpenCV: Mat is mvs::Mat (matlite.hpp) -- the same
// seam written against the reference's own header and the real cv::Mat is host/render_hip_cv.cpp; errors are C++ exceptions (std::runtime_error) instead of
// assert/exit(1) (recon.cpp:49, configuration.cpp:136,141,172); video decoding is replaced by a frame
// directory (the clips are not in the reference checkout either: .MISSING_LARGE_BLOBS).
#pragma once

#include <functional>
#include <list>
#include <map>
#include <set>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "matlite.hpp"

typedef unsigned char uchar;
typedef mvs::Mat Mat;
typedef struct Mesh {
    Mat vertices, faces;  // N x 4 f32 homogeneous rows; F x 3 i32
    Mesh() {}
    Mesh(Mat v, Mat f) : vertices(v), faces(f) {}
} Mesh;
typedef std::list<Mat> MatList;

class Configuration;
class Heuristic;

const float backgroundDepth = 1.0;  // recon.hpp:30

// == flow (flow.cpp:19-42) ==
Mat calculateFlow(const Mat prev, const Mat next, bool useFarneback);

// == util (util.cpp) ==
Mat extractCameraCenter(const Mat camera);                 // util.cpp:33-41: homogeneous 4x1 centre
// util.cpp:167-329: rows (x, y, z, w, nx, ny, nz), one per triangulated pixel in scan order
Mat triangulatePixels(const MatList flows, const Mat mainCamera, const MatList cameras, const Mat depth);
Mat compare(const Mat prev, const Mat next);               // util.cpp:332-361
Mat dehomogenize(Mat points);                              // util.cpp:16-29
Mat mixBackground(const Mat image, const Mat background, Mat &depth);  // util.cpp:366-387 (mutates depth)
Mat flowRemap(const Mat flow, const Mat image);            // util.cpp:390-403
float sampleImage(const Mat image, float radiusSquared, const float x, const float y, char c);  // util.cpp:408-433 (recon.hpp:47)

// == surface meshing (alpha_shapes.cpp:36-104, cgal_poisson.cpp:47-136): host/alpha_shapes.cpp, host/poisson.cpp ==
Mat alphaShapeFaces(const Mat points);                // recon.hpp:33
Mat alphaShapeFaces(const Mat points, float *alpha);  // recon.hpp:34: faces F x 3 i32 of the alpha shape that is one solid component; *alpha = the value chosen
Mesh poissonSurface(const Mat points, const Mat normals);  // recon.hpp:37: points N x 4 homogeneous, normals N x 3
// Not part of the reference's interface: what the call above does with the LENGTHS of the normals (triangulatePixels scales them by a
// pdf, util.cpp:322-327).  The reference uses them as confidences on both backends (cgal_poisson.cpp:58-69; pcl.cpp:23 + 198-202); this
// library's default normalises them first -- a deliberate divergence, measured on the pipeline's own clouds (host/poisson.cpp, DESIGN.md
// section 9).  POISSON_CONFIDENCE_NORMALS restores the reference's semantics.
enum PoissonNormals { POISSON_UNIT_NORMALS = 0, POISSON_CONFIDENCE_NORMALS = 1 };
Mesh poissonSurface(const Mat points, const Mat normals, PoissonNormals mode);
void setPoissonNormals(PoissonNormals mode);  // what the two-argument call does from now on (process-wide; the reference is single-threaded)
PoissonNormals poissonNormals();
// poissonSurface ends with a simplification pass (mvs_surface_simplify: the facet count the reference's criteria ask for instead of the
// grid's); setPoissonSimplify(false) returns the criteria pass's mesh as it is
void setPoissonSimplify(bool on);

// == configuration (configuration.cpp) ==
class Configuration {
public:
    Configuration(int argc, char **argv);      // same 12 getopt options as configuration.cpp:37-53
    explicit Configuration(const std::string &yamlPath, int skipFrames = 1);
    Mat reconstructedPoints();                 // bundles, N x 4            configuration.cpp:432-435
    const Mat frame(int frameNo) const;        // H x W u8                  configuration.cpp:437-440
    const Mat camera(int frameNo) const;       // 4 x 4 f32                 configuration.cpp:442-445
    const std::vector<Mat> allCameras() const; // configuration.cpp:447-450
    float nearVal(int frameNo) const { return nearVals.at(frameNo); }
    float farVal(int frameNo) const { return farVals.at(frameNo); }
    int frameCount() const { return (int)cameras.size(); }
    void setFrame(int frameNo, const Mat gray); // supplies a decoded frame (replaces cv::VideoCapture)
    void setFrameColor(int frameNo, const Mat bgr); // the same for a colour frame (H x W x 3 u8, B G R): converted like
                                               // cvtColor(BGR2GRAY) (configuration.cpp:244) unless doEstimateExposure is set
    void estimateExposure();                   // configuration.cpp:270-426; runs by itself once every colour frame is present
    const Mat projectPoints(int frameNo);      // configuration.cpp:262-267
    Mat exposure;                              // channels x frames, filled by estimateExposure (the reference keeps it local)
    int iterationCount = 2;        // configuration.cpp:28
    char verbosity = 0;
    bool useFarneback = false;     // configuration.cpp:26
    float cameraThreshold = 10.f;  // configuration.cpp:30
    float sceneResolution = 1.f;
    float scalingFactor = 1.f;
    unsigned skipFrames = 1;
    int width = 0, height = 0;
    std::string outFileName = "output.obj";
    std::string inMeshFile;
    std::string clipPath;
    std::vector<float> lensDistortion;
    float centerX = 0, centerY = 0;
    bool doEstimateExposure = false;
    int sweepPlanes = 0;           // --sweep-planes N (long option only; not in the reference): 0 = the reference's path, N > 0 = trackMainFrame's swept depth
    int threads = 1;               // --threads N (long option only; not in the reference): reconstructPoints runs the main frames of one outer iteration on N host
                                   // threads, each with a renderer and contexts of its own on the one GPU (the `fa` loop's iterations are independent)

protected:
    void parseYaml(const std::string &path);
    std::vector<Mat> frames;
    std::vector<Mat> colorFrames;  // kept until the exposure estimate has turned them into `frames`
    void colorFramesReady();
    Mat resizedToClipSize(const Mat &frame) const;  // configuration.cpp:232-233 (cv::resize, bilinear) through mvs_resize_u8
    std::vector<Mat> cameras;
    std::vector<float> nearVals, farVals;
    Mat bundles;
    std::vector<std::set<int>> bundlesEnabled;
};

// == renderer ==
class Render {
public:
    virtual ~Render() {}
    virtual void loadMesh(const Mesh) = 0;
    virtual Mat projected(const Mat camera, const Mat frame, const Mat projector) = 0;
    virtual Mat depth(const Mat camera) const = 0;
};
Render *spawnRender(Heuristic hint);

// Optional extension a renderer may also implement (RenderHIP does): depth(camera) restricted to the n pixels the caller
// reads.  Heuristic::chooseCameras uses it when present -- filterCameras reads one pixel per real camera out of each of
// its 200 depth maps (heuristic.cpp:307-312, 445-456) -- and falls back to depth() otherwise.  Not part of the
// reference's interface; a renderer that lacks it changes nothing but speed.
class DepthProbe {
public:
    virtual ~DepthProbe() {}
    virtual void depthAt(const Mat camera, int n, const int32_t *rows, const int32_t *cols, float *out) const = 0;
};

// Optional extension a renderer may also implement (RenderHIP does): the D-plane sweep -- plane-sweep photometric cost volume + per-pixel depth
// selection, the generalisation of shader.frag:11-25 from the mesh's one depth per pixel to `planes` hypotheses -- over frames kept on the device.
// Not part of the reference's interface (the reference has no D-plane path); trackMainFrame below uses it when Configuration::sweepPlanes > 0.
class DepthSweep {
public:
    virtual ~DepthSweep() {}
    // the sequence's frames, uploaded ONCE (Configuration load) and swept many times: size the store (emptying it), then one call per frame
    virtual void storeFrames(int frameCount) = 0;
    virtual int storeCapacity() const = 0;                      // frames the store was sized for (0 before storeFrames)
    virtual void storeFrame(int frameNo, const Mat gray) = 0;   // H x W u8 of the render size
    virtual bool frameStored(int frameNo) const = 0;
    // depth map of the main view (H x W f32, main-camera NDC z like Render::depth, backgroundDepth where no side view sees the pixel on any plane):
    // the plane of lowest mean |I_main - I_side warped| among `planes` planes spread evenly over (zLo, zHi); main and side views are stored frames
    virtual Mat sweepDepth(int mainFrame, const Mat mainCamera, const std::vector<int> &sideFrames, const std::vector<Mat> &sideCameras, int planes,
                           float zLo = -1.f, float zHi = 1.f, Mat *bestCost = nullptr) = 0;
    // Render::projected with a depth map in place of the mesh (no shadow test: a depth map has one surface per pixel): H x W x 3 u8,
    // (intensity, 255, 255) where the pixel's point lands inside `frame`, (0, 0, 0) elsewhere -- what mixBackground expects (util.cpp:376-377)
    virtual Mat projectedByDepth(const Mat camera, const Mat depth, const Mat frame, const Mat projector) = 0;
};

// Optional extension a renderer may also implement (RenderHIP does): the whole body of the `fa` loop (recon.cpp:65-117) in one call on the renderer's own
// context -- depth, per side view projected -> mixBackground -> calculateFlow, triangulatePixels -- with every intermediate kept on the device
// (mvs_process_frame): the same rows (x, y, z, w, nx, ny, nz), bit for bit, as the stage-by-stage calls, without their seven host round trips per side view.
// Not part of the reference's interface; trackMainFrame uses it when present.
class FrameTracker {
public:
    virtual ~FrameTracker() {}
    virtual Mat trackFrame(const Mat mainCamera, const Mat mainFrame, const std::vector<Mat> &sideCameras, const std::vector<Mat> &sideFrames, bool useFarneback,
                           Mat *depthAfter = nullptr) = 0;
    // the same with the frames named by number in the renderer's frame store (DepthSweep::storeFrame): the `fa` loop reads every frame of a sequence about
    // five times (once as a main frame, four times as a side view), through the store it crosses PCIe once (mvs_process_frame_slots)
    virtual Mat trackStoredFrame(const Mat mainCamera, int mainFrame, const std::vector<Mat> &sideCameras, const std::vector<int> &sideFrames, bool useFarneback,
                                 Mat *depthAfter = nullptr) = 0;
};

typedef std::pair<int, std::vector<int>> numberedVector;   // (main frame, its side frames): recon.hpp:102

// == the driver's loop (recon.cpp:42-136) as functions, so that the sweep can sit inside it ==
// recon.cpp:65-117 for ONE main frame: depth map, per side view projected -> mixBackground -> calculateFlow, then triangulatePixels; returns the
// rows (x, y, z, w, nx, ny, nz).  With config.sweepPlanes == 0 (the default) these are exactly the reference's calls.  With sweepPlanes > 0 and a
// renderer that implements DepthSweep the depth map handed to the flows and to triangulatePixels is the SWEPT one -- `sweepPlanes` planes across the
// proxy mesh's own depth range widened by a quarter on both sides, kept only where the proxy covers the pixel -- and the side frames are warped
// through it (projectedByDepth) instead of through the mesh: the flow then only has to correct what is left after depth selection.
Mat trackMainFrame(Configuration &config, Render *render, int mainFrame, const std::vector<int> &sideFrames, Mat *depthUsed = nullptr);
// recon.cpp:42-136: the outer iteration (tessellate -> loadMesh -> chooseCameras -> every main frame -> filterPoints) until the heuristic is happy.
// With config.threads > 1 the main frames of an iteration are tracked by that many host threads (renderers of their own from spawnRender, the mesh loaded
// into each); the point blocks are appended in the order of the `fa` loop whatever thread produced them, so the result does not depend on the thread count.
void reconstructPoints(Configuration &config, Heuristic &hint, Render *render, Mat &points, Mat &normals);
// the same for a given schedule of (main frame, side frames), mesh already loaded into `render`: what one iteration's tracking phase does (recon.cpp:65-117 over
// all main frames); returns the point blocks in schedule order.  `mesh` is loaded into the extra renderers of threads 2 .. N.
std::vector<Mat> trackMainFrames(Configuration &config, Heuristic &hint, Render *render, const Mesh &mesh, const std::vector<numberedVector> &schedule);

// == heuristic ==

// cv::theRNG() as the reference uses it through cv::randu<float>() (heuristic.cpp:207,365,400,450): OpenCV's
// multiply-with-carry generator, default state 0xffffffff, never seeded by the reference -> a fixed stream.
struct HeuristicRNG {
    uint64_t state = 0xffffffffULL;
    unsigned next()
    {
        state = (uint64_t)(unsigned)state * 4164903690U + (unsigned)(state >> 32);
        return (unsigned)state;
    }
    float uniform() { return next() * 2.3283064365386963e-10f; }
};

class Heuristic {
public:
    Heuristic(Configuration *iconfig);
    int chooseCameras(const Mesh mesh, const std::vector<Mat> cameras, const Render &);
    bool notHappy(const Mat points);
    int beginMain();
    int nextMain();
    int beginSide(int mainNumber);
    int nextSide(int mainNumber);
    void filterPoints(Mat &points, Mat &normals);
    // heuristic.cpp:525-545: first iteration -> the mesh file given with -m (alpha 1) or alpha shapes of the points (alpha from the
    // mesher); later iterations -> Poisson surface (alpha halved).  The meshers default to this library's own (alphaShapeFaces,
    // poissonSurface above: SURVEY.md section 8f-4 without CGAL); a caller that links CGAL / PCL installs its own here.
    struct Meshers {
        std::function<Mat(const Mat points, float *alpha)> alphaShapeFaces;       // alpha_shapes.cpp:36-99
        std::function<Mesh(const Mat points, const Mat normals)> poissonSurface;  // cgal_poisson.cpp:47-136 or pcl.cpp
        std::function<Mesh(const char *fileName)> readMesh;                       // util.cpp (Wavefront OBJ)
    };
    Meshers meshers;
    Mesh tessellate(const Mat points, const Mat normals);
    mvs::Size renderSize();
    static const int sentinel = -1;
    HeuristicRNG rng;  // explicit and seedable (SURVEY 8b "determinism hook")
    const std::vector<numberedVector> &chosen() const { return chosenCameras; }
    std::vector<float> alphaVals;

protected:
    Configuration *config;
    int iteration;
    int mainIdx, sideIdx;
    std::vector<numberedVector> chosenCameras;
};
