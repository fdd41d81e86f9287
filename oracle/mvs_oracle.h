/*
 * mvs_oracle.h -- CPU restatement of the dense-MVS hot path of addam/mesh-reconstruction.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ may be imported, linked or executed by the
 * product path (mesh-reconstruction_amd/, include/).  Allowed users: tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg -- always as the checker / reported baseline, never as the thing
 * shipped or measured as the product.
 *
 * PARITY UNPINNED: the reference holds no golden outputs, known-answer tests or fixtures for this
 * path and cannot be compiled in this image (OpenCV, GLEW, GLX server and
 * CGAL are absent), so this restatement is anchored only on the reference's source semantics
 * (file:line cited per function) plus the OpenGL 3.0 rules the reference relies on.
 *
 * All images are top-down, row-major, tightly packed.  Camera matrices are 4x4 float, row-major,
 * applied as P * (x, y, z, 1)^T (reference: glUniformMatrix4fv(..., GL_TRUE, ...), render_glx.cpp:265,338,375).
 */
#ifndef MVS_ORACLE_H
#define MVS_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_BACKGROUND_DEPTH 1.0f /* recon.hpp:30 */

/* ---- plane-sweep cost volume (D-plane generalisation of shader.frag:11-25) ------------------- */

/* 3x4 "view matrix" Q = S * side_cam * inverse(main_cam) rounded to f32 (row-major, 12 floats):
 * maps main-camera NDC (x, y, z, 1) to (cx*w, cy*w, w) where (cx, cy) is the sampling position in
 * the 1-pixel wrap-padded side image (DESIGN.md "sweep arithmetic"). */
void orc_view_matrix(const float main_cam[16], const float side_cam[16], int W, int H, float Q[12]);

/* wrap-pad (GL_REPEAT, render_glx.cpp:81-82) a WxH u8 image to (H+2) rows of `pitch` bytes */
void orc_pad_image(const uint8_t *img, int W, int H, uint8_t *pad, int pitch);

/* plane table: z_d = z_lo + (z_hi - z_lo) * (d + 0.5) / D, evaluated in double, rounded to f32 */
void orc_plane_table(int D, float z_lo, float z_hi, float *z);

/* one sample of the sweep: returns 1 if in frame and writes the u8-rounded warped intensity */
int orc_sweep_sample(const float Q[12], float xn, float yn, float z, const uint8_t *pad, int pitch,
                     int W, int H, int *Iq);

/* NDC centre of pixel (col,row): Appendix A-2 of SURVEY.md (GL window coordinates, top-down image) */
float orc_pixel_xn(int col, int W);
float orc_pixel_yn(int row, int H);

/*
 * Full sweep for one main view.  volume (nullable) is D*H*W packed cells (cnt << 16) | sum where
 * sum = sum over in-frame views of |I_main - round(bilinear I_v)| and cnt = number of in-frame views.
 * first_view/num_views select the sub-range of side views accumulated (a rank's shard); depth
 * selection is done on exactly that range.  nthreads <= 1: scalar loop (the reference has no threads).
 */
void orc_sweep(const float main_cam[16], const uint8_t *main_img, int W, int H,
               int V, const float *side_cams /* V*16 */, const uint8_t *const *side_imgs,
               int D, float z_lo, float z_hi,
               uint32_t *volume /* nullable */, float *depth /* H*W */, float *best_cost /* nullable */,
               int32_t *best_idx /* nullable */, int nthreads);

/* the sweep's sampler at one plane per pixel (z = depth map): out_hw2 = (warped u8, mask) pairs; see sweep_oracle.c */
void orc_warp_by_depth(const float main_cam[16], const float *depth, const float side_cam[16], const uint8_t *frame,
                       int W, int H, uint8_t *out_hw2);

/* per-pixel depth selection over a packed volume: lowest d minimising sum/cnt (exact integer
 * cross-multiplied comparison); no valid plane -> depth 1.0 (backgroundDepth), idx -1, cost +inf */
void orc_argmin(const uint32_t *volume, int W, int H, int D, const float *z,
                float *depth, float *best_cost, int32_t *best_idx);

/* ---- the same sweep with the "fixed" sampler (contract v2: 1/32-texel positions, 8-bit weight table, cells count<<24 | sum of
 * |weights.texels - 255 I_main|); see the block comment in sweep_oracle.c ---- */
void orc_fx_weight_table(uint8_t *lut /* 32*32*4: [ky][kx][w00, w01, w10, w11], each row sums to 255 */);
int orc_sweep_sample_fx(const float Q[12], float xn, float yn, float z, const uint8_t *pad, int pitch, int W, int H, int *dot);
void orc_sweep_fx(const float main_cam[16], const uint8_t *main_img, int W, int H, int V, const float *side_cams,
                  const uint8_t *const *side_imgs, int D, float z_lo, float z_hi, uint32_t *volume, float *depth, float *best_cost,
                  int32_t *best_idx, int nthreads);
void orc_argmin_fx(const uint32_t *volume, int W, int H, int D, const float *z, float *depth, float *best_cost, int32_t *best_idx);
/* sub-plane parabola refinement of the selected plane; cs = 16 / 24: cell format of the exact / fixed sampler */
void orc_refine_depth(const uint32_t *volume, int W, int H, int D, const float *z, const int32_t *idx, int cs, float *depth);
void orc_warp_by_depth_fx(const float main_cam[16], const float *depth, const float side_cam[16], const uint8_t *frame, int W, int H,
                          uint8_t *out_hw2);

/* ---- renderer (render_glx.cpp:230-397 + shader.vert/frag) ---------------------------------- */

/* loadMesh: dehomogenise + expand to triangle soup; out_soup holds 9*nfaces floats. render_glx.cpp:230-258 */
void orc_load_mesh(const float *verts4, int nverts, const int32_t *faces3, int nfaces, float *out_soup);

/* window-space z-buffer (GL_LESS against clear value 1.0; top-left fill rule; clip -w<=z<=w via
 * per-pixel test, near-plane crossing handled with homogeneous edge functions). Output: window z
 * in [0,1] in GL orientation (row 0 = bottom). */
void orc_raster_window_z(const float *soup, int nfaces, const float cam[16], int W, int H, float *zwin_gl);

/* Render::depth: NDC z = 2*zwin-1, top-down, empty = 1.0. render_glx.cpp:369-397 */
void orc_depth(const float *soup, int nfaces, const float cam[16], int W, int H, float *depth_td);

/* the in-place 3x3 shadow "dilation" with its row-0 recursive-min quirk, literal: render_glx.cpp:287-314 */
void orc_shadow_dilate(float *zwin_gl, int W, int H);

/* Render::projected: shadow pass, dilation, projective texture pass; out is H*W*3 u8 top-down.
 * render_glx.cpp:261-367 + shader.frag:11-25 */
void orc_projected_filter(const float *soup, int nfaces, const float cam[16], const uint8_t *frame,
                          const float projector[16], int W, int H, int mipmap, uint8_t *out_hw3);
void orc_projected(const float *soup, int nfaces, const float cam[16], const uint8_t *frame,
                   const float projector[16], int W, int H, uint8_t *out_hw3);

/* ---- photometric helpers (util.cpp) --------------------------------------------------------- */

/* util.cpp:366-387; mutates depth */
void orc_mix_background(const uint8_t *img_hw3, const uint8_t *bg_hw, float *depth_hw, uint8_t *out_hw,
                        int W, int H);

/* util.cpp:332-361: multi-scale L1 pyramid difference of two u8 (or f32) images */
void orc_compare_u8(const uint8_t *prev, const uint8_t *next, int W, int H, float *out);
void orc_compare_f32(const float *prev, const float *next, int W, int H, float *out);

/* util.cpp:390-403: bicubic remap of a u8 image by a dense flow (stride = floats per pixel: 2 or 4) */
void orc_flow_remap(const float *flow, int stride, const uint8_t *image, int W, int H, uint8_t *out);
void orc_resize_linear_u8(const uint8_t *src, int sw, int sh, int channels, uint8_t *dst, int dw, int dh); /* configuration.cpp:233 */
/* the Q15 4x4 bicubic weight table (32*32 sub-pixel positions x 16 taps) remap uses; out nullable */
void orc_remap_cubic_table(short *out);

/* ---- optical flow (flow.cpp:19-42) ---------------------------------------------------------------- */

void orc_gaussian_kernel(int n, double sigma, float *k);
void orc_farneback_gaussian(int n, double sigma, float *g, float *xg, float *xxg, double ig[4]);
int orc_farneback_levels(int W, int H, int levels, double pyr_scale, int *lw, int *lh, double *lscale);
/* cv::FarnebackOpticalFlow::calc with flags 0 (box window, zero initial flow); flow_out = H*W*2 */
void orc_farneback(const uint8_t *prev, const uint8_t *next, int W, int H, int levels, double pyr_scale, int winsize,
                   int iterations, int poly_n, double poly_sigma, float *flow_out);
/* cv::optflow VariationalRefinement::calc with default parameters; flow = H*W*2, refined in place */
void orc_variational_refine(const uint8_t *I0, const uint8_t *I1, int W, int H, float *flow);
/* building blocks, exposed so tests can check them against independent formulations (scipy / least squares) */
void orc_pyr_down(const float *src, int w, int h, float *dst);
void orc_pyr_up(const float *src, int sw, int sh, float *dst, int dw, int dh);
void orc_poly_exp(const float *src, int w, int h, int n, double sigma, float *dst5);
void orc_gaussian_blur(const float *src, int w, int h, int ksize, double sigma, float *dst);
void orc_resize_linear(const float *src, int sw, int sh, int cn, float *dst, int dw, int dh);
/* calculateFlow: out4 = H*W*4 (u, v, variance, 0) */
void orc_calculate_flow(const uint8_t *prev, const uint8_t *next, int W, int H, int use_farneback, float *out4);

/* ---- per-pixel triangulation + normals (util.cpp:44-329), the consumer of depth + flows (recon.cpp:114) ---- */

void orc_inv44_f32(const float *m, float *out);
/* imageGradient (util.cpp:465-479): Sobel 3x3 of a f32 image -> H*W*2 (gx, gy) */
void orc_image_gradient(const float *img, int W, int H, float *grad2);
/* triangulatePixels: flows = V pointers to H*W*4 (u, v, variance, 0); returns N and writes N rows of
 * (x, y, z, w, nx, ny, nz) in pixel scan order; out_points7 must hold H*W*7 floats */
int orc_triangulate_pixels(const float *const *flows, const float main_cam[16], const float *side_cams, int V,
                           const float *depth, int W, int H, float *out_points7);

/* ---- point-cloud filter (heuristic.cpp:55-176) -------------------------------------------------------- */
/* returns M and writes the ascending indices of the retained points to keep_out[0..M); alpha = alphaVals.back() */
int orc_filter_points(const float *points4, int N, float alpha, int32_t *keep_out, float *density_out);

#ifdef __cplusplus
}
#endif
#endif
