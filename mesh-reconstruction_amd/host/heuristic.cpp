// heuristic.cpp -- view-selection / iteration policy of the reference (heuristic.cpp:23-551), host side.
//
// All of this is scalar f32 policy code whose one heavy callee is Render::depth (200 depth renders per
// iteration from cameras placed ON mesh faces, heuristic.cpp:445-456) -- that callee is the HIP rasteriser.
// The logic below restates the reference statement by statement, quirks included:
//   * filterCameras looks the depth map up with an un-flipped y and accepts col == cols (heuristic.cpp:307-309);
//     the read is clamped here instead of running one past the row;
//   * the random stream is cv::theRNG()'s default, never seeded (HeuristicRNG in recon.hpp);
//   * FLANN's L2_Simple returns SQUARED distances which filterPoints uses as distances (heuristic.cpp:81-89);
//     FLANN's randomised KD-tree is replaced by an exact grid search (a superset of what an approximate
//     search returns; the reference already filters by index to restore symmetry, heuristic.cpp:86).
#include <algorithm>
#include <cmath>
#include <cstdio>

#include "recon.hpp"

// render_hip.cpp: mvs_filter_points on the shared context
void filterPointsIndices(const Mat &points, float alpha, int32_t *keep, int *kept, int width, int height);

namespace {

const float focal = 0.5f;  // heuristic.cpp:9

struct CameraLabel {  // heuristic.cpp:12-17
    int index;
    float cosFromViewer, distance;
    float viewX, viewY;
};
const CameraLabel dummyLabel = {-1, 0, 0, 0, 0};  // heuristic.cpp:19
typedef std::vector<std::pair<CameraLabel, Mat>> LabelledCameras;

inline float pow2(float x) { return x * x; }
inline unsigned compact(unsigned short i, unsigned short j) { return (unsigned(i) << 16) + unsigned(j); }  // heuristic.cpp:43-46

struct V3 {
    float x, y, z;
};
inline V3 vertex(const Mat &verts, int i)
{
    const float *p = verts.ptr<float>(i);
    return {p[0] / p[3], p[1] / p[3], p[2] / p[3]};
}
inline V3 sub(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline float norm(V3 a) { return std::sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }

// heuristic.cpp:179-190
float faceArea(const Mat &points, int ia, int ib, int ic)
{
    V3 a = vertex(points, ia), b = vertex(points, ib), c = vertex(points, ic);
    return norm(cross(sub(b, a), sub(c, b))) / 2;
}

Mat mat44(const float (&m)[16])
{
    Mat r(4, 4, mvs::F32C1);
    std::memcpy(r.data, m, sizeof(m));
    return r;
}

// heuristic.cpp:193-247: a camera sitting on the face, looking along its normal
Mat faceCamera(const Mesh &mesh, int faceIdx, float far, float focalLen, HeuristicRNG &rng)
{
    const int32_t *vi = mesh.faces.ptr<int32_t>(faceIdx);
    V3 a = vertex(mesh.vertices, vi[0]), b = vertex(mesh.vertices, vi[1]), c = vertex(mesh.vertices, vi[2]);
    V3 n = cross(sub(b, a), sub(c, b));
    const float len = norm(n);
    n = {n.x / len, n.y / len, n.z / len};
    float u1 = rng.uniform(), u2 = rng.uniform();  // heuristic.cpp:207
    if (u1 + u2 > 1) {
        u1 = 1 - u1;
        u2 = 1 - u2;
    }
    const float u3 = 1 - u1 - u2;
    const float ce[3] = {a.x * u1 + b.x * u2 + c.x * u3, a.y * u1 + b.y * u2 + c.y * u3, a.z * u1 + b.z * u2 + c.z * u3};
    const float x = n.x, y = n.y, z = n.z;
    const float xys = x * x + y * y, xy = std::sqrt(xys);
    float RT[16];
    if (xy > 0) {  // heuristic.cpp:221-227
        const float m[16] = {z * x / xy, z * y / xy, xy, -z * (ce[0] * x + ce[1] * y) / xy - ce[2] * xy,
                             -y / xy,    x / xy,     0,  (ce[0] * y - ce[1] * x) / xy,
                             -x,         -y,         z,  ce[0] * x + ce[1] * y - ce[2] * z,
                             0,          0,          0,  1};
        std::memcpy(RT, m, sizeof(m));
    } else {  // heuristic.cpp:228-236
        const float s = (z > 0) ? 1.f : -1.f;
        const float m[16] = {1, 0, 0, -ce[0], 0, s, 0, -ce[1], 0, 0, s, -ce[2], 0, 0, 0, 1};
        std::memcpy(RT, m, sizeof(m));
    }
    const float near = 0.001f;  // heuristic.cpp:239
    const float K[16] = {focalLen, 0, 0, 0, 0, focalLen, 0, 0, 0, 0, (near + far) / (far - near), 2 * near * far / (near - far), 0, 0, 1, 0};
    return mvs::matmul(mat44(K), mat44(RT));
}

// heuristic.cpp:250-258: linear search; returns list.size() when nothing exceeds `choice`
int bisect(const std::vector<float> &list, float choice)
{
    for (int i = 0; i < (int)list.size(); i++)
        if (list[i] > choice) return i - 1;
    return (int)list.size();
}

// the same on a list the caller has verified to be non-decreasing: "first entry greater than choice" is std::upper_bound.
// chooseCameras looks its F + 1-entry cumulative-area table up 200 times (F ~ 10^6 for the later iterations' Poisson meshes).
int bisectSorted(const std::vector<float> &list, float choice)
{
    const auto it = std::upper_bound(list.begin(), list.end(), choice);
    return it == list.end() ? (int)list.size() : (int)(it - list.begin()) - 1;
}

int myFind(const std::vector<numberedVector> &list, int index)
{
    for (int i = 0; i < (int)list.size(); i++)
        if (list[i].first == index) return i;
    return -1;
}
int myFind(const std::vector<int> &list, int index)
{
    for (int i = 0; i < (int)list.size(); i++)
        if (list[i] == index) return i;
    return -1;
}

// heuristic.cpp:285-341
// The depth lookups are separated from the rest so that a renderer with the DepthProbe extension serves them without
// moving the whole map to the host: pass 1 applies the tests that precede the lookup and collects (row, col), the
// depths arrive in one call, pass 2 applies the remaining tests in the original order.  Same decisions as the one-pass form.
LabelledCameras filterCameras(const Mat &viewer, const Render &render, int rows, int cols, const std::vector<Mat> &cameras,
                              const std::vector<Mat> &centers)
{
    struct Candidate {
        int index;
        float cfv[4];
        int row, col;
    };
    std::vector<Candidate> cand;
    const Mat viewerCenter = extractCameraCenter(viewer);
    for (int i = 0; i < (int)cameras.size(); i++) {
        Candidate c;
        c.index = i;
        Mat cfvM = mvs::matmul(viewer, centers[i]);  // centers[i] = extractCameraCenter(cameras[i]), hoisted out of the 200 shots
        for (int k = 0; k < 4; k++) c.cfv[k] = cfvM.at<float>(k, 0) / cfvM.at<float>(3, 0);
        if (c.cfv[2] > 1 || c.cfv[2] < -1) continue;  // wrong side of the face
        c.row = (int)((c.cfv[1] + 1) * rows / 2);
        c.col = (int)((c.cfv[0] + 1) * cols / 2);
        if (c.row < 0 || c.row >= rows || c.col < 0 || c.col > cols) continue;  // `col > cols`, heuristic.cpp:309
        c.col = std::min(c.col, cols - 1);                                      // the read is clamped (see the header note)
        cand.push_back(c);
    }
    std::vector<float> obstacle(cand.size());
    if (!cand.empty()) {
        if (const DepthProbe *probe = dynamic_cast<const DepthProbe *>(&render)) {
            std::vector<int32_t> r(cand.size()), c(cand.size());
            for (size_t k = 0; k < cand.size(); k++) {
                r[k] = cand[k].row;
                c[k] = cand[k].col;
            }
            probe->depthAt(viewer, (int)cand.size(), r.data(), c.data(), obstacle.data());
        } else {
            const Mat depth = render.depth(viewer);  // the reference's path: the whole map for a handful of pixels
            for (size_t k = 0; k < cand.size(); k++) obstacle[k] = depth.at<float>(cand[k].row, cand[k].col);
        }
    }
    LabelledCameras filtered;
    for (size_t k = 0; k < cand.size(); k++) {
        const Candidate &c = cand[k];
        const Mat &camera = cameras[c.index];
        CameraLabel label = dummyLabel;
        label.index = c.index;
        label.viewX = c.cfv[0];
        label.viewY = c.cfv[1];
        if (obstacle[k] != backgroundDepth && obstacle[k] <= c.cfv[2]) continue;
        Mat vfcM = mvs::matmul(camera, viewerCenter);
        label.distance = vfcM.at<float>(3, 0) / viewerCenter.at<float>(3, 0);
        if (label.distance < 0) continue;
        const float w = vfcM.at<float>(3, 0);
        const float vfc0 = vfcM.at<float>(0, 0) / w, vfc1 = vfcM.at<float>(1, 0) / w;
        if (vfc0 < -1 || vfc0 > 1 || vfc1 < -1 || vfc1 > 1) continue;
        label.cosFromViewer = std::sqrt(1 / (1 + (c.cfv[0] * c.cfv[0] + c.cfv[1] * c.cfv[1]) / (focal * focal)));
        filtered.push_back(std::make_pair(label, camera));
    }
    return filtered;
}

// heuristic.cpp:345-369
CameraLabel chooseMain(std::map<unsigned, float> &weights, const LabelledCameras &fc, float *outWeightSum, float boostFactor,
                       HeuristicRNG &rng)
{
    if (fc.empty()) throw std::runtime_error("chooseMain: no camera passed the visibility tests");
    std::vector<float> weightSum(fc.size() + 1, 0.f);
    *outWeightSum = 0;
    for (int i = 0; i < (int)fc.size(); i++) {
        const CameraLabel &label = fc[i].first;
        float weight = label.cosFromViewer / pow2(label.distance);
        *outWeightSum += weight;
        if (weights.count(compact(label.index, label.index))) weight += weight * boostFactor * fc.size();
        weightSum[i + 1] = weightSum[i] + weight;
    }
    const float choice = rng.uniform() * weightSum.back();  // heuristic.cpp:365
    const int index = bisect(weightSum, choice);
    return fc[std::min(std::max(index, 0), (int)fc.size() - 1)].first;
}

// heuristic.cpp:372-426
CameraLabel chooseSide(std::map<unsigned, float> &weights, CameraLabel mainCamera, float threshold, float boostFactor,
                       const LabelledCameras &fc, HeuristicRNG &rng)
{
    if (fc.size() < 2) throw std::runtime_error("chooseSide: fewer than two cameras");
    std::vector<float> weightSum(fc.size(), 0.f);
    std::vector<CameraLabel> labels;
    float actualWeightSum = 0;
    int i = 0;
    for (const auto &entry : fc) {
        const CameraLabel &label = entry.first;
        if (label.index == mainCamera.index) continue;
        const float parallaxSqr = (pow2(label.viewX - mainCamera.viewX) + pow2(label.viewY - mainCamera.viewY)) / focal;
        float weight = label.cosFromViewer * parallaxSqr / pow2(label.distance);
        actualWeightSum += weight;
        const unsigned key = compact(mainCamera.index, label.index);
        if (weights.count(key) && weights[key] >= 1) weight += weight * boostFactor * fc.size();
        if (i + 1 < (int)weightSum.size()) weightSum[i + 1] = weightSum[i] + weight;
        labels.push_back(label);
        i++;
    }
    if (labels.empty()) return dummyLabel;
    const float choice = rng.uniform() * weightSum.back();  // heuristic.cpp:400
    int index = bisect(weightSum, choice);
    index = std::min(std::max(index, 0), i - 1);  // assert(index >= 0 && index < i), heuristic.cpp:402
    const unsigned key = compact(mainCamera.index, labels[index].index);
    if (weights[key] >= 1) return dummyLabel;  // already selected
    weights[compact(mainCamera.index, mainCamera.index)] = 1;
    const float addWeight = (weightSum[index + 1] - weightSum[index]) / (threshold * actualWeightSum);
    weights[key] += addWeight;
    return weights[key] >= 1 ? labels[index] : dummyLabel;
}

}  // namespace

Heuristic::Heuristic(Configuration *iconfig) : config(iconfig), iteration(0), mainIdx(0), sideIdx(0) {}

// heuristic.cpp:31-35
bool Heuristic::notHappy(const Mat)
{
    iteration++;
    return iteration <= config->iterationCount;
}

// heuristic.cpp:429-486
int Heuristic::chooseCameras(const Mesh mesh, const std::vector<Mat> cameras, const Render &render)
{
    chosenCameras.clear();
    int cameraCount = 0;
    const int F = mesh.faces.rows;
    std::vector<float> areaSum(F + 1, 0.f);
    for (int i = 0; i < F; i++) {
        const int32_t *vi = mesh.faces.ptr<int32_t>(i);
        areaSum[i + 1] = areaSum[i] + faceArea(mesh.vertices, vi[0], vi[1], vi[2]);
    }
    const float totalArea = areaSum.back();
    const float samplingResolution = std::sqrt((float)cameras.size()) * config->width * config->height / (totalArea * config->cameraThreshold);
    const int shotCount = 200;  // heuristic.cpp:445
    std::map<unsigned, float> weights;
    // the reference decomposes every real camera again for every shot (heuristic.cpp:293: 200 x N_cam calls); the centre
    // of a camera does not depend on the shot, so it is computed once per call with the same function
    std::vector<Mat> centers;
    centers.reserve(cameras.size());
    for (const Mat &camera : cameras) centers.push_back(extractCameraCenter(camera));
    const bool areasSorted = std::is_sorted(areaSum.begin(), areaSum.end());  // false only with NaN areas: then the linear walk
    for (int i = 0; i < shotCount; i++) {
        const float choice = rng.uniform() * totalArea;  // heuristic.cpp:450
        int chosenIdx = (areasSorted && choice == choice) ? bisectSorted(areaSum, choice) : bisect(areaSum, choice);
        chosenIdx = std::min(std::max(chosenIdx, 0), F - 1);
        const float far = 10;  // heuristic.cpp:454
        const Mat viewer = faceCamera(mesh, chosenIdx, far, focal, rng);
        // the hot callee: one depth map per shot (render.depth, or its probed form when the renderer offers it)
        const LabelledCameras filtered = filterCameras(viewer, render, config->height, config->width, cameras, centers);
        if (filtered.size() >= 2) {
            float mainWeightSum;
            const CameraLabel mainCamera = chooseMain(weights, filtered, &mainWeightSum, config->cameraThreshold, rng);
            const CameraLabel sideCamera = chooseSide(weights, mainCamera, shotCount * mainWeightSum / samplingResolution,
                                                      config->cameraThreshold / 10, filtered, rng);
            if (sideCamera.index == dummyLabel.index) continue;
            cameraCount += 1;
            const int positionMain = myFind(chosenCameras, mainCamera.index);
            if (positionMain == -1)
                chosenCameras.push_back(numberedVector(mainCamera.index, std::vector<int>(1, sideCamera.index)));
            else if (myFind(chosenCameras[positionMain].second, sideCamera.index) == -1)
                chosenCameras[positionMain].second.push_back(sideCamera.index);
        }
    }
    std::sort(chosenCameras.begin(), chosenCameras.end());
    return cameraCount;
}

// heuristic.cpp:489-522
int Heuristic::beginMain() { return chosenCameras.empty() ? sentinel : chosenCameras[mainIdx = 0].first; }
int Heuristic::nextMain() { return ++mainIdx < (int)chosenCameras.size() ? chosenCameras[mainIdx].first : sentinel; }
int Heuristic::beginSide(int imain)
{
    if (imain != chosenCameras[mainIdx].first || chosenCameras[mainIdx].second.empty()) return sentinel;
    return chosenCameras[mainIdx].second[sideIdx = 0];
}
int Heuristic::nextSide(int imain)
{
    if (imain != chosenCameras[mainIdx].first || ++sideIdx >= (int)chosenCameras[mainIdx].second.size()) return sentinel;
    return chosenCameras[mainIdx].second[sideIdx];
}

// heuristic.cpp:525-545
Mesh Heuristic::tessellate(const Mat points, const Mat normals)
{
    if (iteration <= 1) {
        if (!config->inMeshFile.empty()) {
            if (!meshers.readMesh) throw std::runtime_error("tessellate: no readMesh callback installed");
            Mesh result = meshers.readMesh(config->inMeshFile.c_str());
            alphaVals.push_back(1);  // "TODO: estimate some alpha value from the geometry", heuristic.cpp:530
            return result;
        }
        float alpha = 0;
        Mat faces = meshers.alphaShapeFaces ? meshers.alphaShapeFaces(points, &alpha) : alphaShapeFaces(points, &alpha);
        alphaVals.push_back(alpha);
        return Mesh(points, faces);
    }
    if (alphaVals.empty()) throw std::runtime_error("tessellate: Poisson iteration before any alpha value was recorded");
    Mesh result = meshers.poissonSurface ? meshers.poissonSurface(points, normals) : poissonSurface(points, normals);
    alphaVals.push_back(alphaVals.back() / 2);
    return result;
}

// heuristic.cpp:548-551
mvs::Size Heuristic::renderSize() { return mvs::Size(config->width, config->height); }

// heuristic.cpp:55-176: neighbour table, power iteration and the greedy pass run behind mvs_filter_points (filter.hip);
// here only the in-place compaction of heuristic.cpp:166-175
void Heuristic::filterPoints(Mat &points, Mat &normals)
{
    const int pointCount = points.rows;
    if (pointCount == 0) return;
    if (alphaVals.empty()) throw std::runtime_error("filterPoints: no alpha value recorded (tessellate was never called)");
    std::vector<int32_t> keep((size_t)pointCount);
    int kept = 0;
    filterPointsIndices(points, alphaVals.back(), keep.data(), &kept, config->width, config->height);
    Mat np(kept, 4, mvs::F32C1), nn(kept, 3, mvs::F32C1);
    for (int i = 0; i < kept; i++) {
        std::memcpy(np.ptr<float>(i), points.ptr<float>(keep[i]), 4 * sizeof(float));
        std::memcpy(nn.ptr<float>(i), normals.ptr<float>(keep[i]), 3 * sizeof(float));
    }
    points = np;
    normals = nn;
}
