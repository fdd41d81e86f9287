// fuzz_readers.cpp -- libFuzzer harness over the host code that parses UNTRUSTED files: the tracks YAML reader of
// Configuration (configuration.cpp:138-246 of the reference: cv::FileStorage there, a hand-written reader here) and the frame readers that
// stand in for cv::VideoCapture (binary PGM / PPM, YUV4MPEG2).  Built by `make sanitize` with -fsanitize=fuzzer,address,undefined; the
// first input byte selects the reader, the rest is the file.  A rejected file is a C++ exception (expected); anything the sanitizers
// report, a crash, a timeout or an allocation beyond the limits given on the command line is a finding.
// The translation unit under test is included as source so that its file-local readers are reachable.
#include "../../mesh-reconstruction_amd/host/configuration.cpp"

#include <unistd.h>

#include <cstdio>

static std::string scratch_dir()
{
    static std::string dir;
    if (dir.empty()) {
        char tmpl[] = "/tmp/mvs_fuzz_XXXXXX";
        const char *d = mkdtemp(tmpl);
        dir = d ? d : "/tmp";
    }
    return dir;
}

extern "C" int LLVMFuzzerTestOneInput(const uint8_t *data, size_t size)
{
    if (size < 1) return 0;
    const std::string path = scratch_dir() + "/input";
    if (FILE *f = fopen(path.c_str(), "wb")) {
        fwrite(data + 1, 1, size - 1, f);
        fclose(f);
    } else {
        return 0;
    }
    try {
        switch (data[0] & 3) {
        case 0: {
            Mat m;
            (void)readPgm(path, 0, 0, m);
            break;
        }
        case 1: {
            Mat m;
            (void)readPpm(path, 0, 0, m);
            break;
        }
        case 2: {
            std::vector<Mat> frames;
            (void)readY4m(path, 1 + ((data[0] >> 2) & 3), 1 + ((data[0] >> 4) & 7), frames);
            break;
        }
        default: {
            Configuration c(path, 1 + ((data[0] >> 2) & 3));
            (void)c.frameCount();
            if (c.frameCount() > 0) (void)c.camera(0);
            break;
        }
        }
    } catch (const std::exception &) {  // a rejected file
    }
    return 0;
}
