// dump_goldens.cpp -- runs the REFERENCE's renderer, mixBackground, compare, flowRemap and calculateFlow on the inputs written by
// make_inputs.py and dumps the results as .npy, for tests/test_ref_goldens.py (README.md in this directory).
// Built against the reference tree (make REF=...): this file includes the reference's render_glx.cpp, whose RenderGLX class is
// local to that file, and links its util.cpp / flow.cpp.  It needs OpenCV, GLEW and an X display; it cannot be built in this
// repository's image and has never been run.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include REF_RENDER_GLX  // brings recon.hpp, cv::Mat, class RenderGLX (render_glx.cpp:38-52)

// Heuristic / Configuration are referenced by render_glx.cpp's spawnRender only: never called here, satisfied for the linker
Heuristic::Heuristic(Configuration *) {}
cv::Size Heuristic::renderSize() { return cv::Size(0, 0); }

namespace {

// minimal .npy (version 1.0, little endian, C order) reader / writer for f32, i32 and u8
struct Npy {
    std::vector<int> shape;
    std::string descr;
    std::vector<char> data;
};

Npy load(const std::string &path)
{
    std::ifstream f(path, std::ios::binary);
    if (!f) {
        fprintf(stderr, "cannot open %s\n", path.c_str());
        exit(1);
    }
    char magic[8];
    f.read(magic, 8);
    unsigned short hlen = 0;
    f.read((char *)&hlen, 2);
    std::string hdr(hlen, ' ');
    f.read(&hdr[0], hlen);
    Npy a;
    const size_t d = hdr.find("'descr': '");
    a.descr = hdr.substr(d + 10, hdr.find('\'', d + 10) - d - 10);
    const size_t s = hdr.find('(', hdr.find("'shape'")), e = hdr.find(')', s);
    std::string dims = hdr.substr(s + 1, e - s - 1);
    size_t n = 1;
    for (char *tok = strtok(&dims[0], ", "); tok; tok = strtok(nullptr, ", ")) {
        a.shape.push_back(atoi(tok));
        n *= (size_t)atoi(tok);
    }
    const size_t item = a.descr == "|u1" ? 1 : 4;
    a.data.resize(n * item);
    f.read(a.data.data(), (std::streamsize)a.data.size());
    return a;
}

void save(const std::string &path, const cv::Mat &m_in)
{
    cv::Mat m = m_in.isContinuous() ? m_in : m_in.clone();
    const char *descr = m.depth() == CV_8U ? "|u1" : (m.depth() == CV_32S ? "<i4" : "<f4");
    char shape[64];
    if (m.channels() == 1)
        snprintf(shape, sizeof(shape), "(%d, %d)", m.rows, m.cols);
    else
        snprintf(shape, sizeof(shape), "(%d, %d, %d)", m.rows, m.cols, m.channels());
    std::string hdr = std::string("{'descr': '") + descr + "', 'fortran_order': False, 'shape': " + shape + ", }";
    while ((10 + hdr.size() + 1) % 64) hdr += ' ';
    hdr += '\n';
    std::ofstream f(path, std::ios::binary);
    f.write("\x93NUMPY\x01\x00", 8);
    const unsigned short hlen = (unsigned short)hdr.size();
    f.write((const char *)&hlen, 2);
    f.write(hdr.data(), (std::streamsize)hdr.size());
    f.write((const char *)m.data, (std::streamsize)(m.total() * m.elemSize()));
}

cv::Mat as_mat(const Npy &a, int type)
{
    const int rows = a.shape[0], cols = a.shape.size() > 1 ? a.shape[1] : 1;
    return cv::Mat(rows, cols, type, (void *)a.data.data()).clone();
}

}  // namespace

int main(int argc, char **argv)
{
    const std::string io = argc > 1 ? argv[1] : "io";
    const cv::Mat points = as_mat(load(io + "/in_points.npy"), CV_32FC1), faces = as_mat(load(io + "/in_faces.npy"), CV_32SC1);
    const cv::Mat mvp = as_mat(load(io + "/in_mvp.npy"), CV_32FC1), side = as_mat(load(io + "/in_side_mvp.npy"), CV_32FC1);
    const cv::Mat frame_a = as_mat(load(io + "/in_frame_a.npy"), CV_8UC1), frame_b = as_mat(load(io + "/in_frame_b.npy"), CV_8UC1);

    RenderGLX r(frame_a.cols, frame_a.rows, getenv("DISPLAY") ? getenv("DISPLAY") : (char *)":0");  // render_glx.cpp:152
    r.loadMesh(Mesh(points, faces));                                                                 // :230
    cv::Mat depth = r.depth(mvp);                                                                    // :369
    save(io + "/depth.npy", depth);
    save(io + "/depth_side.npy", r.depth(side));
    const cv::Mat projected = r.projected(mvp, frame_b, side);                                       // :261 (frame_b seen from the side camera)
    save(io + "/projected.npy", projected);
    const cv::Mat mixed = mixBackground(projected, frame_a, depth);                                  // util.cpp:366 (mutates depth)
    save(io + "/mixed.npy", mixed);
    save(io + "/depth_after_mix.npy", depth);
    save(io + "/compare.npy", compare(frame_a, frame_b));                                            // util.cpp:332
    const cv::Mat fb = calculateFlow(frame_a, frame_b, true), fv = calculateFlow(frame_a, frame_b, false);  // flow.cpp:19
    save(io + "/flow_farneback.npy", fb);
    save(io + "/flow_variational.npy", fv);
    save(io + "/flow_remap.npy", flowRemap(fv, frame_b));                                            // util.cpp:390
    printf("wrote 10 arrays to %s\n", io.c_str());
    return 0;
}
