// configuration.cpp -- Configuration: command line + YAML-tracks input of the reference (configuration.cpp:18-465),
// without OpenCV.
//
// What is kept: the 12 getopt options and their defaults (configuration.cpp:20-53), the YAML schema written by the
// Blender exporter (io_export_tracks.py:34-96: `clip`, `camera[]` with !!opencv-matrix `projection`, `tracks[]` with
// `bundle` + `frames-enabled`), 1-based frame numbers, `skipFrames` re-indexing and `-s` down-scaling of
// width/height/centre (configuration.cpp:160-165, 186-187, 207-212), and the accessors of recon.hpp:62-69.
// What is replaced: cv::FileStorage by a reader for exactly that YAML subset (block maps / block sequences /
// single-line flow sequences / `!!opencv-matrix` with rows, cols, dt, data); cv::VideoCapture by setFrame() or a
// directory of binary PGM files `<clip path>.frames/%06d.pgm` (the clips are absent from the reference checkout:
// .MISSING_LARGE_BLOBS); exit(1) by exceptions.  estimateExposure (configuration.cpp:270-426) is not on the hot
// path and is not built; requesting it (-e) throws.
#include <getopt.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <sstream>

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
};

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

Node parseValue(const std::vector<Line> &lines, size_t &i, int parentIndent);

// parse a block (map or sequence) whose entries sit at `indent`
Node parseBlock(const std::vector<Line> &lines, size_t &i, int indent)
{
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
                item.map.push_back(std::make_pair(key, parseValue(lines, i, indent)));
            }
            n.seq.push_back(item);
        }
        return n;
    }
    n.kind = Node::Map;
    while (i < lines.size() && lines[i].indent == indent && !lines[i].dash) {
        const std::string key = lines[i].key;
        n.map.push_back(std::make_pair(key, parseValue(lines, i, indent)));
    }
    return n;
}

Node parseValue(const std::vector<Line> &lines, size_t &i, int parentIndent)
{
    const Line &l = lines[i++];
    Node n;
    std::string v = l.value;
    if (v.compare(0, 2, "!!") == 0) {  // type tag such as !!opencv-matrix: the value is the nested block
        const size_t sp = v.find(' ');
        v = sp == std::string::npos ? std::string() : trim(v.substr(sp));
    }
    if (v.empty()) {
        if (i < lines.size() && lines[i].indent > parentIndent) return parseBlock(lines, i, lines[i].indent);
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
    const int rows = (int)n.at("rows").num(), cols = (int)n.at("cols").num();
    const Node &data = n.at("data");
    if ((int)data.flow.size() != rows * cols) throw std::runtime_error("tracks YAML: opencv-matrix data size mismatch");
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
    if (magic != "P5" || pw != w || ph != h || maxv != 255) throw std::runtime_error("frame " + path + ": expected binary PGM of the clip size");
    out.create(h, w, mvs::U8C1);
    f.read(reinterpret_cast<char *>(out.data), (std::streamsize)w * h);
    return (bool)f;
}

}  // namespace

void Configuration::parseYaml(const std::string &path)
{
    std::ifstream in(path);
    if (!in) throw std::runtime_error("Cannot read file " + path);  // configuration.cpp:139-142
    const std::vector<Line> lines = tokenize(in);
    size_t i = 0;
    const Node root = parseBlock(lines, i, 0);

    const Node &clip = root.at("clip");  // configuration.cpp:145-166
    width = (int)clip.at("width").num();
    height = (int)clip.at("height").num();
    clipPath = dirName(path) + "/" + clip.at("path").scalar;
    centerX = (float)clip.at("center-x").num();
    centerY = (float)clip.at("center-y").num();
    if (scalingFactor != 1 && scalingFactor != 0) {
        width = (int)(width / scalingFactor);
        height = (int)(height / scalingFactor);
        centerX /= scalingFactor;
        centerY /= scalingFactor;
    }
    for (const std::string &s : clip.at("distortion").flow) lensDistortion.push_back((float)atof(s.c_str()));

    bundles = Mat(0, 4, mvs::F32C1);  // configuration.cpp:176-197
    const Node *tracks = root.get("tracks");
    if (tracks)
        for (const Node &t : tracks->seq) {
            const Mat bundle = matrixOf(t.at("bundle"));  // 4 x 1
            Mat row(1, 4, mvs::F32C1);
            for (int k = 0; k < 4; k++) row.at<float>(0, k) = bundle.at<float>(k, 0);
            bundles.push_back(row);
            std::set<int> enabled;
            for (const std::string &s : t.at("frames-enabled").flow) {
                const int f = atoi(s.c_str());
                if ((f - 1) % (int)skipFrames == 0) enabled.insert((f - 1) / (int)skipFrames);
            }
            bundlesEnabled.push_back(enabled);
        }

    int trackedFrameCount = -1;  // configuration.cpp:200-224
    std::map<int, const Node *> byIndex;
    for (const Node &c : root.at("camera").seq) {
        int fi = (int)c.at("frame").num();
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
        if (!(nearVals[kv.first] > 0 && farVals[kv.first] > 0)) throw std::runtime_error("tracks YAML: near/far must be positive");
    }
    frames.assign(trackedFrameCount, Mat());
    // frames: optional directory of PGMs next to the clip (stands in for cv::VideoCapture, configuration.cpp:169-238)
    for (int fi = 0; fi < trackedFrameCount; fi++) {
        char name[64];
        snprintf(name, sizeof(name), "/%06d.pgm", fi * (int)skipFrames + 1);
        Mat g;
        if (readPgm(clipPath + ".frames" + name, width, height, g)) frames[fi] = g;
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
        default:
            throw std::runtime_error("Usage: recon [OPTIONS] [INPUT_FILE]  (options: -c f, -e, -f, -h, -i s, -k i, -m s, -n i, -o s, -s f, -v, -V)");
        }
    }
    if (optind < argc) inFileName = argv[optind];  // configuration.cpp:129-131
    if (!inFileName) throw std::runtime_error("No configuration YAML file given");
    if (doEstimateExposure) throw std::runtime_error("--estimate-exposure is not part of the MI355X hot path build");
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

void Configuration::setFrame(int frameNo, const Mat gray)
{
    if (gray.type() != mvs::U8C1 || gray.cols != width || gray.rows != height) throw std::runtime_error("setFrame: frame must be H x W u8 of the clip size");
    frames.at(frameNo) = gray;
}
