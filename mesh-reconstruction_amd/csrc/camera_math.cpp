// camera_math.cpp -- host-side double-precision camera algebra for libmvs_hip.so.
//
// The sweep never projects through the two 4x4 cameras on the device.  The host folds
//   Q_v = S * side_cam_v * inverse(main_cam)
// into one 3x4 f32 matrix per view, where S scales clip coordinates to wrap-padded pixel units
// (texel centres and the vertical flip of render_glx.cpp:69 / SURVEY.md Appendix A-7 included), so the
// kernel evaluates shader.frag:13-15 (`sideMVP * vec4(pos,1)`) as 3 FMAs per sample.
//
// Arithmetic contract (DESIGN.md "sweep arithmetic"): inverse by cofactor expansion with the
// term order below, products accumulated k = 0..3, one rounding to f32 at the end.  The CPU oracle
// restates the same contract so both sides start from identical f32 matrices.
#include "mvs_internal.hpp"

namespace mvs {

namespace {
// signed 3x3 minor of m with row r and column c removed, expanded along the first remaining row in
// the order (+a(ei-fh) written as aei - afh - bdi + bfg + cdh - ceg); signs folded by the caller
struct Term {
    int a, b, c;
};
}  // namespace

void invert4(const double m[16], double out[16])
{
    double c[16];
    c[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] + m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
    c[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] - m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
    c[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] + m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
    c[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] - m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
    c[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] - m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
    c[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] + m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
    c[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] - m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
    c[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] + m[8] * m[2] * m[13] + m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
    c[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] + m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
    c[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] - m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
    c[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] + m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
    c[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] - m[4] * m[2] * m[13] - m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
    c[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] - m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
    c[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] + m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
    c[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] - m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
    c[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] + m[4] * m[2] * m[9] + m[8] * m[1] * m[6] - m[8] * m[2] * m[5];
    const double det = m[0] * c[0] + m[1] * c[4] + m[2] * c[8] + m[3] * c[12];
    const double rdet = 1.0 / det;
    for (int i = 0; i < 16; i++) out[i] = c[i] * rdet;
}

void view_matrix(const float main_cam[16], const float side_cam[16], int W, int H, float Q[12])
{
    double M[16], Mi[16], C[16], T[16];
    for (int i = 0; i < 16; i++) {
        M[i] = main_cam[i];
        C[i] = side_cam[i];
    }
    invert4(M, Mi);
    for (int r = 0; r < 4; r++)
        for (int col = 0; col < 4; col++) {
            double s = 0.0;
            for (int k = 0; k < 4; k++) s += C[4 * r + k] * Mi[4 * k + col];
            T[4 * r + col] = s;
        }
    // clip (x, y, w) -> padded pixel * w:  cx_p = (x/w) W/2 + W/2 - 1/2 + 1,  cy_p = -(y/w) H/2 + H/2 - 1/2 + 1
    const double hw = 0.5 * W, hh = 0.5 * H;
    for (int col = 0; col < 4; col++) {
        Q[0 + col] = (float)(hw * T[0 + col] + (hw + 0.5) * T[12 + col]);
        Q[4 + col] = (float)(-hh * T[4 + col] + (hh + 0.5) * T[12 + col]);
        Q[8 + col] = (float)(T[12 + col]);
    }
}

void plane_table(int D, float z_lo, float z_hi, float *z)
{
    for (int d = 0; d < D; d++)
        z[d] = (float)((double)z_lo + ((double)z_hi - (double)z_lo) * ((double)d + 0.5) / (double)D);
}

}  // namespace mvs
