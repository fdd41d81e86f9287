// configuration.cpp -- Configuration: command line + YAML-tracks input of the reference (configuration.cpp:18-465),
// without OpenCV.
//
// What is kept: the 12 getopt options and their defaults (configuration.cpp:20-53), the YAML schema written by the
// Blender exporter (io_export_tracks.py:34-96: `clip`, `camera[]` with !!opencv-matrix `projection`, `tracks[]` with
// `bundle` + `frames-enabled`), 1-based frame numbers, `skipFrames` re-indexing and `-s` down-scaling of
// width/height/centre (configuration.cpp:160-165, 186-187, 207-212), and the accessors of recon.hpp:62-69.
// What is replaced: cv::FileStorage by a reader for exactly that YAML subset (block maps / block sequences /
// single-line flow sequences / `!!opencv-matrix` with rows, cols, dt, data); cv::VideoCapture by an uncompressed YUV4MPEG2 stream (readY4m), setFrame() or a
// directory of binary PGM files `<clip path>.frames/%06d.pgm` (the clips are absent from the reference checkout:
// .MISSING_LARGE_BLOBS) or binary PPM files `%06d.ppm` for colour; exit(1) by exceptions.  estimateExposure
// (configuration.cpp:270-426, option -e) is a one-time host stage over the sparse bundle points and runs here on the host
// too, with the pseudo-inverse of each frame's n x 3 sample matrix taken through its 3 x 3 normal matrix.
#include <getopt.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <sstream>

#include "../../include/mvs.h"
#include "recon.hpp"

namespace {

struct Node {
    enum Kind { Scalar, FlowSeq, Map, Seq } kind = Scalar;
    std::string scalar;
    std::vector<std::string> flow;
    std::vector<std::pair<std::string, Node>> map;
    std::vector<Node> seq;
    const Node *get(const std::string &key) const
    {
        for (const auto &kv : map)
            if (kv.first == key) return &kv.second;
        return nullptr;
    }
    const Node &at(const std::string &key) const
    {
        const Node *n = get(key);
        if (!n) throw std::runtime_error("tracks YAML: missing key '" + key + "'");
        return *n;
    }
    double num() const { return atof(scalar.c_str()); }
    // an integer the file is allowed to say: anything else -- not a number, NaN, beyond [lo, hi] -- is a rejected file, not a cast
    // of an out-of-range double (undefined behaviour; found by tools/fuzz/fuzz_readers.cpp in its first second)
    int integer(int lo, int hi, const char *what) const
    {
        const double v = atof(scalar.c_str());
        if (!(v >= (double)lo && v <= (double)hi)) throw std::runtime_error(std::string("tracks YAML: ") + what + " out of range: '" + scalar + "'");
        return (int)v;
    }
};

// an integer token of a flow sequence, same rule
int integerToken(const std::string &s, int lo, int hi, const char *what)
{
    const double v = atof(s.c_str());
    if (!(v >= (double)lo && v <= (double)hi)) throw std::runtime_error(std::string("tracks YAML: ") + what + " out of range: '" + s + "'");
    return (int)v;
}

// the bytes of `path` from the stream's current position to its end (what a header may promise at most)
size_t bytesLeft(std::ifstream &f)
{
    const std::streampos here = f.tellg();
    f.seekg(0, std::ios::end);
    const std::streampos end = f.tellg();
    f.seekg(here);
    return (here < 0 || end < here) ? 0 : (size_t)(end - here);
}

constexpr int kMaxYamlDepth = 32;      // nesting of blocks (the exporter writes 4 levels)
constexpr int kMaxFrameNumber = 1 << 22;  // 1-based frame numbers of a clip (46 hours at 25 frames per second)

struct Line {
    int indent;
    bool dash;
    std::string key, value;
};

std::string trim(const std::string &s)
{
    size_t a = s.find_first_not_of(" \t\r\n"), b = s.find_last_not_of(" \t\r\n");
    return a == std::string::npos ? std::string() : s.substr(a, b - a + 1);
}

std::vector<Line> tokenize(std::istream &in)
{
    std::vector<Line> lines;
    std::string raw;
    while (std::getline(in, raw)) {
        if (raw.empty() || raw[0] == '%' || raw[0] == '#' || trim(raw).empty() || trim(raw) == "---") continue;
        Line l;
        size_t i = 0;
        while (i < raw.size() && raw[i] == ' ') i++;
        l.indent = (int)i;
        l.dash = false;
        if (raw.compare(i, 2, "- ") == 0) {
            l.dash = true;
            i += 2;
            l.indent = (int)i;  // the entry's keys align with the text after "- "
        }
        std::string rest = raw.substr(i);
        const size_t colon = rest.find(':');
        if (colon == std::string::npos) throw std::runtime_error("tracks YAML: cannot parse line: " + raw);
        l.key = trim(rest.substr(0, colon));
        l.value = trim(rest.substr(colon + 1));
        // a flow sequence may continue on following lines
        if (!l.value.empty() && l.value[0] == '[') {
            while (l.value.find(']') == std::string::npos && std::getline(in, raw)) l.value += " " + trim(raw);
        }
        lines.push_back(l);
    }
    return lines;
}

Node parseValue(const std::vector<Line> &lines, size_t &i, int parentIndent, int depth);

// parse a block (map or sequence) whose entries sit at `indent`
Node parseBlock(const std::vector<Line> &lines, size_t &i, int indent, int depth = 0)
{
    if (depth > kMaxYamlDepth) throw std::runtime_error("tracks YAML: blocks nested deeper than 32 levels");  // (recursion bounded: a file cannot overflow the stack)
    Node n;
    if (i < lines.size() && lines[i].dash && lines[i].indent == indent) {
        n.kind = Node::Seq;
        while (i < lines.size() && lines[i].indent == indent && lines[i].dash) {
            Node item;
            item.kind = Node::Map;
            bool first = true;
            while (i < lines.size() && lines[i].indent == indent && (first || !lines[i].dash)) {
                first = false;
                const std::string key = lines[i].key;
                item.map.push_back(std::make_pair(key, parseValue(lines, i, indent, depth)));
            }
            n.seq.push_back(item);
        }
        return n;
    }
    n.kind = Node::Map;
    while (i < lines.size() && lines[i].indent == indent && !lines[i].dash) {
        const std::string key = lines[i].key;
        n.map.push_back(std::make_pair(key, parseValue(lines, i, indent, depth)));
    }
    return n;
}

Node parseValue(const std::vector<Line> &lines, size_t &i, int parentIndent, int depth)
{
    const Line &l = lines[i++];
    Node n;
    std::string v = l.value;
    if (v.compare(0, 2, "!!") == 0) {  // type tag such as !!opencv-matrix: the value is the nested block
        const size_t sp = v.find(' ');
        v = sp == std::string::npos ? std::string() : trim(v.substr(sp));
    }
    if (v.empty()) {
        if (i < lines.size() && lines[i].indent > parentIndent) return parseBlock(lines, i, lines[i].indent, depth + 1);
        return n;
    }
    if (v[0] == '[') {
        n.kind = Node::FlowSeq;
        const size_t close = v.rfind(']');
        std::stringstream ss(v.substr(1, close == std::string::npos ? std::string::npos : close - 1));
        std::string tok;
        while (std::getline(ss, tok, ',')) {
            tok = trim(tok);
            if (!tok.empty()) n.flow.push_back(tok);
        }
        return n;
    }
    n.scalar = v;
    if (n.scalar.size() >= 2 && (n.scalar[0] == '"' || n.scalar[0] == '\'')) n.scalar = n.scalar.substr(1, n.scalar.size() - 2);
    return n;
}

Mat matrixOf(const Node &n)
{
    const int rows = n.at("rows").integer(0, 4096, "matrix rows"), cols = n.at("cols").integer(0, 4096, "matrix cols");
    const Node &data = n.at("data");
    if (data.flow.size() != (size_t)rows * (size_t)cols) throw std::runtime_error("tracks YAML: opencv-matrix data size mismatch");
    if (n.at("dt").scalar != "f") throw std::runtime_error("tracks YAML: only dt: f matrices are supported");
    Mat m(rows, cols, mvs::F32C1);
    for (int i = 0; i < rows * cols; i++) m.ptr<float>()[i] = (float)atof(data.flow[i].c_str());
    return m;
}

std::string dirName(const std::string &path)
{
    const size_t s = path.find_last_of('/');
    return s == std::string::npos ? std::string(".") : path.substr(0, s);
}

bool readPgm(const std::string &path, int w, int h, Mat &out)
{
    std::ifstream f(path, std::ios::binary);
    if (!f) return false;
    std::string magic;
    int pw = 0, ph = 0, maxv = 0;
    f >> magic >> pw >> ph >> maxv;
    f.get();
    if (magic != "P5" || pw < 1 || ph < 1 || pw > 16384 || ph > 16384 || maxv != 255) throw std::runtime_error("frame " + path + ": expected a binary 8-bit PGM of at most 16384 x 16384");
    if (bytesLeft(f) < (size_t)pw * ph) throw std::runtime_error("frame " + path + ": truncated (the header promises more pixels than the file holds)");  // before anything is allocated
    (void)w;  // a frame of another size than the (scaled) clip is resized by the caller, as configuration.cpp:232-233 does
    (void)h;
    out.create(ph, pw, mvs::U8C1);
    f.read(reinterpret_cast<char *>(out.data), (std::streamsize)pw * ph);
    return (bool)f;
}

// binary PPM (R G B per pixel) -> the B G R order cv::VideoCapture delivers
bool readPpm(const std::string &path, int w, int h, Mat &out)
{
    std::ifstream f(path, std::ios::binary);
    if (!f) return false;
    std::string magic;
    int pw = 0, ph = 0, maxv = 0;
    f >> magic >> pw >> ph >> maxv;
    f.get();
    if (magic != "P6" || pw < 1 || ph < 1 || pw > 16384 || ph > 16384 || maxv != 255) throw std::runtime_error("frame " + path + ": expected a binary 8-bit PPM of at most 16384 x 16384");
    if (bytesLeft(f) < (size_t)pw * ph * 3) throw std::runtime_error("frame " + path + ": truncated (the header promises more pixels than the file holds)");
    (void)w;
    (void)h;
    out.create(ph, pw, mvs::U8C3);
    f.read(reinterpret_cast<char *>(out.data), (std::streamsize)pw * ph * 3);
    uint8_t *p = out.ptr<uint8_t>();
    for (size_t i = 0; i < (size_t)pw * ph; i++) std::swap(p[3 * i], p[3 * i + 2]);
    return (bool)f;
}

// Uncompressed video: a YUV4MPEG2 stream (`ffmpeg -i clip.mkv clip.mkv.y4m`; 8-bit C420* / C422 / C444 / Cmono) read frame by frame the
// way configuration.cpp:228-238 reads the clip: frame fi * skipFrames of the stream becomes tracked frame fi, the ones between are
// skipped.  A decoder hands cv::VideoCapture B G R pixels: Y'CbCr -> R'G'B' here is ITU-R BT.601 limited range in 16.16 fixed point
// with the chroma sample that covers the pixel (no chroma interpolation) -- FFmpeg's swscale output differs by its chroma filter; the
// reference does not pin a decoder either.  Returns false when `path` is not such a stream.
bool readY4m(const std::string &path, int skipFrames, int count, std::vector<Mat> &bgr)
{
    std::ifstream f(path, std::ios::binary);
    if (!f) return false;
    std::string header;
    std::getline(f, header);
    if (header.compare(0, 10, "YUV4MPEG2 ") != 0) return false;
    int w = 0, h = 0, cw = 2, ch = 2;  // chroma subsampling factors; 0 = no chroma planes
    std::istringstream tags(header.substr(10));
    std::string tag;
    while (tags >> tag) {
        if (tag[0] == 'W') w = (int)std::min(1000000L, std::max(0L, strtol(tag.c_str() + 1, nullptr, 10)));   // (atoi of a number beyond int is undefined)
        else if (tag[0] == 'H') h = (int)std::min(1000000L, std::max(0L, strtol(tag.c_str() + 1, nullptr, 10)));
        else if (tag[0] == 'C') {
            const std::string c = tag.substr(1);
            // exact tags only: C420p10 / C420p12 / C420p16 carry two bytes per sample and would decode as garbage
            if (c == "420" || c == "420jpeg" || c == "420mpeg2" || c == "420paldv") cw = 2, ch = 2;
            else if (c == "422") cw = 2, ch = 1;
            else if (c == "444") cw = 1, ch = 1;
            else if (c == "mono") cw = 0, ch = 0;
            else throw std::runtime_error("clip " + path + ": unsupported YUV4MPEG2 colour space " + c + " (8-bit 420 / 422 / 444 / mono only)");
        }
    }
    if (w < 1 || h < 1 || w > 16384 || h > 16384) throw std::runtime_error("clip " + path + ": YUV4MPEG2 header without a size (or beyond 16384 x 16384)");
    const size_t cpw = cw ? (size_t)(w + cw - 1) / cw : 0, cph = ch ? (size_t)(h + ch - 1) / ch : 0;
    if (count < 0 || skipFrames < 1) throw std::runtime_error("clip " + path + ": bad frame count / skip");
    if (bytesLeft(f) < (size_t)w * h + 2 * cpw * cph) {  // not one whole frame: nothing is allocated for a header's promise
        bgr.assign(count, Mat());
        return true;
    }
    std::vector<uint8_t> Y((size_t)w * h), U(cpw * cph), V(cpw * cph);
    bgr.assign(count, Mat());
    int next = 0;  // tracked frame waiting for stream frame next * skipFrames
    for (int si = 0; next < count; si++) {
        std::string fh;
        if (!std::getline(f, fh)) break;  // a clip shorter than the tracks: the remaining frames stay empty (frame() says so when asked)
        if (fh.compare(0, 5, "FRAME") != 0) throw std::runtime_error("clip " + path + ": expected a FRAME marker");
        f.read((char *)Y.data(), (std::streamsize)Y.size());
        if (cw) {
            f.read((char *)U.data(), (std::streamsize)U.size());
            f.read((char *)V.data(), (std::streamsize)V.size());
        }
        if (!f) throw std::runtime_error("clip " + path + ": truncated frame");
        if (si != next * skipFrames) continue;
        Mat m(h, w, mvs::U8C3);
        for (int y = 0; y < h; y++) {
            uint8_t *row = m.ptr<uint8_t>(y);
            for (int x = 0; x < w; x++) {
                const int c = 298 * ((int)Y[(size_t)y * w + x] - 16);
                int d = 0, e = 0;
                if (cw) d = (int)U[(size_t)(y / ch) * cpw + x / cw] - 128, e = (int)V[(size_t)(y / ch) * cpw + x / cw] - 128;
                auto clip8 = [](int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); };
                row[3 * x + 2] = clip8((c + 409 * e + 128) >> 8);            // R
                row[3 * x + 1] = clip8((c - 100 * d - 208 * e + 128) >> 8);  // G
                row[3 * x + 0] = clip8((c + 516 * d + 128) >> 8);            // B
            }
        }
        bgr[next++] = m;
    }
    return true;
}

// cv::cvtColor(BGR2GRAY) for 8-bit images: fixed point, 14 fractional bits (OpenCV 3.x color.cpp: R 4899, G 9617, B 1868)
Mat bgrToGray(const Mat &bgr)
{
    Mat g(bgr.rows, bgr.cols, mvs::U8C1);
    const uint8_t *s = bgr.ptr<uint8_t>();
    uint8_t *d = g.ptr<uint8_t>();
    for (size_t i = 0; i < (size_t)bgr.rows * bgr.cols; i++)
        d[i] = (uint8_t)((s[3 * i] * 1868 + s[3 * i + 1] * 9617 + s[3 * i + 2] * 4899 + (1 << 13)) >> 14);
    return g;
}

// configuration.cpp:248-259: radial distortion applied to cartesian points (rows), z scaled along like the reference does
void cameraToScreen(Mat points, const std::vector<float> &lensDistortion, float aspect)
{
    for (int i = 0; i < points.rows; i++) {
        float *p = points.ptr<float>(i);
        const float radSquared = (p[0] * p[0] + p[1] * p[1] * aspect * aspect) / 4;
        const float k = 1 + radSquared * (lensDistortion[0] + radSquared * lensDistortion[1]);
        for (int c = 0; c < points.cols; c++) p[c] *= k;
    }
}

// least-squares solution of A x = b of minimal norm (A: n x m row-major, m <= 3): what validSamples[i].inv(DECOMP_SVD) * b
// computes (configuration.cpp:389), through the eigen-decomposition of the m x m normal matrix (Jacobi); directions whose
// eigenvalue is negligible are dropped, like the zero singular values of the pseudo-inverse
void pinvSolve(const float *A, int n, int m, const float *b, float *x)
{
    double N[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}, r[3] = {0, 0, 0}, V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int k = 0; k < n; k++)
        for (int a = 0; a < m; a++) {
            r[a] += (double)A[k * m + a] * b[k];
            for (int c = 0; c < m; c++) N[a][c] += (double)A[k * m + a] * A[k * m + c];
        }
    for (int sweep = 0; sweep < 64; sweep++) {
        double off = 0;
        for (int a = 0; a < m; a++)
            for (int c = a + 1; c < m; c++) off += std::fabs(N[a][c]);
        if (off < 1e-300) break;
        for (int p = 0; p < m; p++)
            for (int q = p + 1; q < m; q++) {
                if (N[p][q] == 0.0) continue;
                const double theta = (N[q][q] - N[p][p]) / (2.0 * N[p][q]);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                const double c = 1.0 / std::sqrt(t * t + 1.0), sn = t * c;
                for (int k = 0; k < m; k++) {
                    const double kp = N[k][p], kq = N[k][q];
                    N[k][p] = c * kp - sn * kq;
                    N[k][q] = sn * kp + c * kq;
                }
                for (int k = 0; k < m; k++) {
                    const double pk = N[p][k], qk = N[q][k];
                    N[p][k] = c * pk - sn * qk;
                    N[q][k] = sn * pk + c * qk;
                }
                for (int k = 0; k < m; k++) {
                    const double vp = V[k][p], vq = V[k][q];
                    V[k][p] = c * vp - sn * vq;
                    V[k][q] = sn * vp + c * vq;
                }
            }
    }
    double lmax = 0;
    for (int a = 0; a < m; a++) lmax = std::max(lmax, std::fabs(N[a][a]));
    double sol[3] = {0, 0, 0};
    for (int e = 0; e < m; e++) {
        if (!(N[e][e] > 1e-12 * lmax)) continue;
        double proj = 0;
        for (int a = 0; a < m; a++) proj += V[a][e] * r[a];
        for (int a = 0; a < m; a++) sol[a] += V[a][e] * proj / N[e][e];
    }
    for (int a = 0; a < m; a++) x[a] = (float)sol[a];
}

}  // namespace

// util.cpp:408-433: mean of the unclipped values (0 < v < 255) of one channel inside a disc; -1 when there is none
float sampleImage(const Mat image, float radiusSquared, const float x, const float y, char channel)
{
    const int ch = image.channels();
    float sum = 0.f;
    int weightSum = 0;
    const float radius = std::sqrt(radiusSquared);
    for (int ny = (int)std::max(0.f, y - radius); ny < std::min(y + radius + 1, (float)image.rows); ny++) {
        const uint8_t *row = image.ptr<uint8_t>(ny);
        for (int nx = (int)std::max(0.f, x - radius); nx < std::min(x + radius + 1, (float)image.cols); nx++) {
            const float dx = nx - x, dy = ny - y;
            const uint8_t val = row[nx * ch + channel];
            if (dx * dx + dy * dy <= radiusSquared && val > 0 && val < 255) {
                sum += val;
                weightSum += 1;
            }
        }
    }
    return weightSum > 0 ? sum / weightSum : -1.f;
}

void Configuration::parseYaml(const std::string &path)
{
    std::ifstream in(path);
    if (!in) throw std::runtime_error("Cannot read file " + path);  // configuration.cpp:139-142
    const std::vector<Line> lines = tokenize(in);
    size_t i = 0;
    const Node root = parseBlock(lines, i, 0);

    const Node &clip = root.at("clip");  // configuration.cpp:145-166
    width = clip.at("width").integer(2, 16384, "clip width");   // (what mvs_create accepts)
    height = clip.at("height").integer(2, 16384, "clip height");
    clipPath = dirName(path) + "/" + clip.at("path").scalar;
    centerX = (float)clip.at("center-x").num();
    centerY = (float)clip.at("center-y").num();
    if (scalingFactor != 1 && scalingFactor != 0) {
        const double sw = width / scalingFactor, sh = height / scalingFactor;
        if (!(sw >= 2 && sw <= 16384 && sh >= 2 && sh <= 16384)) throw std::runtime_error("tracks YAML: the scaled clip size is outside 2..16384");
        width = (int)sw;
        height = (int)sh;
        centerX /= scalingFactor;
        centerY /= scalingFactor;
    }
    for (const std::string &s : clip.at("distortion").flow) lensDistortion.push_back((float)atof(s.c_str()));

    bundles = Mat(0, 4, mvs::F32C1);  // configuration.cpp:176-197
    const Node *tracks = root.get("tracks");
    if (tracks)
        for (const Node &t : tracks->seq) {
            const Mat bundle = matrixOf(t.at("bundle"));  // 4 x 1
            if (bundle.rows != 4 || bundle.cols != 1) throw std::runtime_error("tracks YAML: a bundle must be a 4 x 1 matrix");
            Mat row(1, 4, mvs::F32C1);
            for (int k = 0; k < 4; k++) row.at<float>(0, k) = bundle.at<float>(k, 0);
            bundles.push_back(row);
            std::set<int> enabled;
            for (const std::string &s : t.at("frames-enabled").flow) {
                const int f = integerToken(s, -kMaxFrameNumber, kMaxFrameNumber, "frames-enabled entry");
                if ((f - 1) % (int)skipFrames == 0) enabled.insert((f - 1) / (int)skipFrames);
            }
            bundlesEnabled.push_back(enabled);
        }

    int trackedFrameCount = -1;  // configuration.cpp:200-224
    std::map<int, const Node *> byIndex;
    for (const Node &c : root.at("camera").seq) {
        int fi = c.at("frame").integer(-kMaxFrameNumber, kMaxFrameNumber, "camera frame number");   // (the arrays below are sized by the largest one)
        if (fi <= 0) throw std::runtime_error("tracks YAML: frame numbers are 1-based");
        fi -= 1;
        if (fi % (int)skipFrames) continue;
        fi /= (int)skipFrames;
        byIndex[fi] = &c;
        if (trackedFrameCount <= fi) trackedFrameCount = fi + 1;
    }
    if (trackedFrameCount < 0) trackedFrameCount = 0;
    cameras.assign(trackedFrameCount, Mat());
    nearVals.assign(trackedFrameCount, 0.f);
    farVals.assign(trackedFrameCount, 0.f);
    for (const auto &kv : byIndex) {
        nearVals[kv.first] = (float)kv.second->at("near").num();
        farVals[kv.first] = (float)kv.second->at("far").num();
        cameras[kv.first] = matrixOf(kv.second->at("projection"));
        if (cameras[kv.first].rows != 4 || cameras[kv.first].cols != 4) throw std::runtime_error("tracks YAML: a projection must be a 4 x 4 matrix");
        if (!(nearVals[kv.first] > 0 && farVals[kv.first] > 0)) throw std::runtime_error("tracks YAML: near/far must be positive");
    }
    frames.assign(trackedFrameCount, Mat());
    colorFrames.assign(trackedFrameCount, Mat());
    // frames: optional directory of PGMs (grey) or PPMs (colour) next to the clip (stands in for cv::VideoCapture,
    // configuration.cpp:169-238)
    {
        // the clip itself if it is an uncompressed YUV4MPEG2 stream, or one next to it (`<clip>.y4m`)
        std::vector<Mat> decoded;
        if (readY4m(clipPath, (int)skipFrames, trackedFrameCount, decoded) || readY4m(clipPath + ".y4m", (int)skipFrames, trackedFrameCount, decoded))
            for (int fi = 0; fi < trackedFrameCount; fi++)
                if (!decoded[fi].empty()) colorFrames[fi] = (decoded[fi].cols != width || decoded[fi].rows != height) ? resizedToClipSize(decoded[fi]) : decoded[fi];
    }
    for (int fi = 0; fi < trackedFrameCount; fi++) {
        if (!colorFrames[fi].empty()) continue;
        char name[64];
        snprintf(name, sizeof(name), "/%06d.pgm", fi * (int)skipFrames + 1);
        Mat g;
        if (readPgm(clipPath + ".frames" + name, width, height, g)) {
            setFrame(fi, g);  // (resized when its size is not the clip's, configuration.cpp:232-233)
            continue;
        }
        snprintf(name, sizeof(name), "/%06d.ppm", fi * (int)skipFrames + 1);
        if (readPpm(clipPath + ".frames" + name, width, height, g)) {
            if (g.cols != width || g.rows != height) g = resizedToClipSize(g);
            colorFrames[fi] = g;
        }
    }
    colorFramesReady();
}

// configuration.cpp:240-245: colour frames become grey ones, either through the exposure estimate (needs all of them) or by
// cvtColor(BGR2GRAY)
void Configuration::colorFramesReady()
{
    if (doEstimateExposure) {
        for (const Mat &c : colorFrames)
            if (c.empty()) return;  // wait until every frame has been supplied
        if (!colorFrames.empty()) estimateExposure();
        return;
    }
    for (size_t i = 0; i < colorFrames.size(); i++)
        if (!colorFrames[i].empty()) {
            frames[i] = bgrToGray(colorFrames[i]);
            colorFrames[i] = Mat();
        }
}

// configuration.cpp:262-267
const Mat Configuration::projectPoints(const int frameNo)
{
    Mat projected(bundles.rows, 4, mvs::F32C1);
    const Mat cam = camera(frameNo);
    for (int j = 0; j < bundles.rows; j++)
        for (int r = 0; r < 4; r++) {
            float s = 0.f;  // (camera * bundles.t()).t(): float accumulation like cv::gemm on CV_32F
            for (int k = 0; k < 4; k++) s += cam.at<float>(r, k) * bundles.at<float>(j, k);
            projected.at<float>(j, r) = s;
        }
    Mat cartesian = dehomogenize(projected);
    cameraToScreen(cartesian, lensDistortion, (float)height / (float)width);
    return cartesian;
}

// configuration.cpp:270-426
void Configuration::estimateExposure()
{
    const int frameCount = (int)cameras.size(), pointCount = bundles.rows;
    if (frameCount == 0 || (int)colorFrames.size() != frameCount) throw std::runtime_error("estimateExposure: colour frames missing");
    const int ch = colorFrames[0].channels();
    if (lensDistortion.size() < 2) throw std::runtime_error("estimateExposure: the tracks file gives no lens distortion");
    std::vector<float> sampledColor;                     // rows of `ch` brightness values, one per valid (frame, point) sample
    std::vector<int> sampleIds((size_t)frameCount * pointCount, -1);
    std::vector<int> rowBegin(frameCount + 1, 0);
    int rowId = 0;
    for (int i = 0; i < frameCount; i++) {
        const Mat &image = colorFrames[i];
        if (image.empty() || image.channels() != ch) throw std::runtime_error("estimateExposure: colour frame " + std::to_string(i) + " missing");
        const Mat reprojected = projectPoints(i);
        rowBegin[i] = rowId;
        for (int j = 0; j < pointCount; j++) {
            if (!bundlesEnabled[j].count(i)) continue;
            const float *re = reprojected.ptr<float>(j);
            const float imageX = centerX + re[0] * width * 0.5f, imageY = height - centerY - re[1] * height * 0.5f;
            float sc[4];
            bool valid = true;
            for (int c = 0; c < ch && valid; c++) {
                sc[c] = sampleImage(image, 16, imageX, imageY, (char)c);
                valid = sc[c] != -1;
            }
            if (!valid) continue;
            sampledColor.insert(sampledColor.end(), sc, sc + ch);
            sampleIds[(size_t)i * pointCount + j] = rowId++;
        }
        if (rowId - rowBegin[i] < ch)  // `assert(false)` in the reference (configuration.cpp:318-321)
            throw std::runtime_error("estimateExposure: frame " + std::to_string(i) + " has fewer valid samples than colour channels");
    }
    rowBegin[frameCount] = rowId;

    double sumBrightness = 0;
    for (size_t k = 0; k < sampledColor.size(); k++) sumBrightness += sampledColor[k];
    sumBrightness *= 1. / ch;

    exposure.create(ch, frameCount, mvs::F32C1);
    for (int c = 0; c < ch; c++)
        for (int i = 0; i < frameCount; i++) exposure.at<float>(c, i) = 1.f / ch;
    std::vector<float> pointBrightness(pointCount, 1.f), valid;
    for (int iteration = 0; iteration < 100; iteration++) {
        double error = 0, currentSumBrightness = 0;
        for (int j = 0; j < pointCount; j++) {  // imagine that the exposure is correct
            float sum = 0.f;
            int weightSum = 0;
            for (int i = 0; i < frameCount; i++) {
                const int row = sampleIds[(size_t)i * pointCount + j];
                if (row == -1) continue;
                weightSum += 1;
                for (int c = 0; c < ch; c++) sum += sampledColor[(size_t)row * ch + c] * exposure.at<float>(c, i);
            }
            currentSumBrightness += sum;
            pointBrightness[j] = weightSum > 0 ? sum / weightSum : 0.f;
        }
        const float scale = (float)(sumBrightness / currentSumBrightness);  // back to the original scale
        for (float &b : pointBrightness) b *= scale;
        for (int i = 0; i < frameCount; i++) {  // imagine that the point brightness is correct
            valid.clear();
            for (int j = 0; j < pointCount; j++)
                if (sampleIds[(size_t)i * pointCount + j] >= 0) valid.push_back(pointBrightness[j]);
            const float *A = sampledColor.data() + (size_t)rowBegin[i] * ch;
            const int n = rowBegin[i + 1] - rowBegin[i];
            float x[3] = {0, 0, 0};
            pinvSolve(A, n, ch, valid.data(), x);
            const float omega = 0.4f;  // "strongly overrelax"
            double norm2 = 0;
            for (int c = 0; c < ch; c++) exposure.at<float>(c, i) = x[c] * (1 + omega) - exposure.at<float>(c, i) * omega;
            for (int k = 0; k < n; k++) {
                float fit = 0.f;
                for (int c = 0; c < ch; c++) fit += A[(size_t)k * ch + c] * exposure.at<float>(c, i);
                norm2 += (double)(fit - valid[k]) * (fit - valid[k]);
            }
            error += std::sqrt(norm2) / n;
        }
        if (error / frameCount < 0.1) break;
    }
    if (verbosity >= 3) {  // configuration.cpp:396-416
        if (FILE *exlog = fopen("exposure.tab", "w+")) {
            for (int i = 0; i < frameCount; i++) {
                double stddev = 0;
                int weightSum = 0;
                for (int j = 0; j < pointCount; j++) {
                    const int row = sampleIds[(size_t)i * pointCount + j];
                    if (row == -1) continue;
                    for (int c = 0; c < ch; c++) {
                        const float d = sampledColor[(size_t)row * ch + c] - exposure.at<float>(c, i) * pointBrightness[j];
                        stddev += d * d;
                        weightSum += 1;
                    }
                }
                for (int c = 0; c < 3; c++) fprintf(exlog, "%f\t", c < ch ? exposure.at<float>(c, i) : 0.f);
                fprintf(exlog, "%f\n", std::sqrt(stddev / weightSum));
            }
            fclose(exlog);
        }
    }
    // normalise the frames: frames[i] = sum_c channel_c * exposure[c][i], each step saturate_cast<uchar>(cvRound(.)) like
    // `frames[i] += channels[c] * exposure` on a CV_8U matrix (configuration.cpp:418-425)
    for (int i = 0; i < frameCount; i++) {
        Mat g(height, width, mvs::U8C1);
        const uint8_t *s = colorFrames[i].ptr<uint8_t>();
        uint8_t *d = g.ptr<uint8_t>();
        for (size_t p = 0; p < (size_t)width * height; p++) {
            int acc = 0;
            for (int c = 0; c < ch; c++) {
                const long v = std::lrint((double)((float)s[p * ch + c] * exposure.at<float>(c, i) + (float)acc));
                acc = (int)std::min(255l, std::max(0l, v));
            }
            d[p] = (uint8_t)acc;
        }
        frames[i] = g;
        colorFrames[i] = Mat();
    }
}

Configuration::Configuration(const std::string &yamlPath, int skip)
{
    skipFrames = skip > 0 ? skip : 1;
    parseYaml(yamlPath);
}

Configuration::Configuration(int argc, char **argv)
{
    const char *inFileName = nullptr;
    optind = 1;
    static struct option long_options[] = {{"input", required_argument, 0, 'i'},        {"initial-mesh", required_argument, 0, 'm'},
                                           {"output", required_argument, 0, 'o'},       {"camera-threshold", required_argument, 0, 'c'},
                                           {"estimate-exposure", no_argument, 0, 'e'},  {"iterations", required_argument, 0, 'n'},
                                           {"scale", required_argument, 0, 's'},        {"skip-frames", required_argument, 0, 'k'},
                                           {"farneback", no_argument, 0, 'f'},          {"verbose", no_argument, 0, 'v'},
                                           {"hyper-verbose", no_argument, 0, 'V'},      {"help", no_argument, 0, 'h'},
                                           {"sweep-planes", required_argument, 0, 1000},  // (not in the reference: recon.hpp's sweepPlanes)
                                           {"threads", required_argument, 0, 1001},       // (not in the reference: recon.hpp's threads)
                                           {0, 0, 0, 0}};
    for (;;) {
        int option_index = 0;
        const int c = getopt_long(argc, argv, "i:m:o:c:en:s:k:fvVh", long_options, &option_index);
        if (c == -1) break;
        switch (c) {
        case 'i': inFileName = optarg; break;
        case 'm': inMeshFile = optarg; break;
        case 'o': outFileName = optarg; break;
        case 'c': cameraThreshold = (float)atof(optarg); break;
        case 'e': doEstimateExposure = true; break;
        case 'n': iterationCount = atoi(optarg); break;
        case 's': {
            const float tmp = (float)atof(optarg);
            if (tmp > 1) scalingFactor = tmp;  // configuration.cpp:83-87
        } break;
        case 'k': skipFrames = (unsigned)std::max(1, atoi(optarg)); break;
        case 'f': useFarneback = true; break;
        case 'v':
            if (verbosity < 2) verbosity = 2;
            break;
        case 'V': verbosity = 99; break;
        case 1000: sweepPlanes = std::max(0, atoi(optarg)); break;
        case 1001: threads = std::min(64, std::max(1, atoi(optarg))); break;
        default:
            throw std::runtime_error("Usage: recon [OPTIONS] [INPUT_FILE]  (options: -c f, -e, -f, -h, -i s, -k i, -m s, -n i, -o s, -s f, -v, -V)");
        }
    }
    if (optind < argc) inFileName = argv[optind];  // configuration.cpp:129-131
    if (!inFileName) throw std::runtime_error("No configuration YAML file given");
    parseYaml(inFileName);
}

Mat Configuration::reconstructedPoints() { return bundles.clone(); }

const Mat Configuration::frame(int frameNo) const
{
    const Mat &f = frames.at(frameNo);
    if (f.empty()) throw std::runtime_error("frame " + std::to_string(frameNo) + " was not supplied (setFrame / .frames directory)");
    return f;
}

const Mat Configuration::camera(int frameNo) const { return cameras.at(frameNo); }
const std::vector<Mat> Configuration::allCameras() const { return cameras; }

static Mat resizedToClip(const Mat &frame, int width, int height);

// configuration.cpp:232-235: a decoded frame whose size is not (width, height) -- the -s option divides those -- goes through
// cv::resize(frame, frames[fi], cv::Size(width, height), CV_INTER_AREA), i.e. INTER_LINEAR (the constant lands in the ignored fx
// argument): mvs_resize_u8, on a context of its own per call
Mat Configuration::resizedToClipSize(const Mat &frame) const { return resizedToClip(frame, width, height); }

static Mat resizedToClip(const Mat &frame, int width, int height)
{
    // a context per call, destroyed before returning: no process-wide handle that is never released or shared between threads
    // (a frame is resized once, when it is loaded; creating a context is a stream creation)
    mvs_ctx *ctx = mvs_create(0, width, height);
    if (!ctx) throw std::runtime_error(std::string("resize: ") + mvs_last_error(nullptr));
    Mat out(height, width, frame.type());
    const int rc = mvs_resize_u8(ctx, frame.ptr<uchar>(0), frame.cols, frame.rows, frame.channels(), out.ptr<uchar>(0), width, height);
    const std::string msg = rc ? mvs_last_error(ctx) : "";
    mvs_destroy(ctx);
    if (rc) throw std::runtime_error("resize: " + msg);
    return out;
}

void Configuration::setFrameColor(int frameNo, const Mat bgr_in)
{
    if (bgr_in.type() != mvs::U8C3) throw std::runtime_error("setFrameColor: frame must be H x W x 3 u8");
    const Mat bgr = (bgr_in.cols != width || bgr_in.rows != height) ? resizedToClip(bgr_in, width, height) : bgr_in;
    colorFrames.at(frameNo) = bgr;
    colorFramesReady();
}

void Configuration::setFrame(int frameNo, const Mat gray_in)
{
    if (gray_in.type() != mvs::U8C1) throw std::runtime_error("setFrame: frame must be H x W u8");
    // (the reference resizes the decoded BGR frame and converts to grey afterwards; a grey frame of another size is resized as is)
    const Mat gray = (gray_in.cols != width || gray_in.rows != height) ? resizedToClip(gray_in, width, height) : gray_in;
    frames.at(frameNo) = gray;
}
