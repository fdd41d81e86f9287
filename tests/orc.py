"""ctypes loader of the CPU oracle (oracle/_build/libmvs_oracle.so).  Test infrastructure only."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB = os.path.join(ORACLE_DIR, "_build", os.environ.get("MVS_BUILD_VARIANT", ""), "libmvs_oracle.so")   # (_build/san under MVS_BUILD_VARIANT=san)

_fp, _u8p, _i32p, _u32p = C.POINTER(C.c_float), C.POINTER(C.c_uint8), C.POINTER(C.c_int32), C.POINTER(C.c_uint32)


def build(force=False):
    srcs = [os.path.join(ORACLE_DIR, f) for f in os.listdir(ORACLE_DIR) if f.endswith((".c", ".h"))]
    if force or not os.path.exists(LIB) or any(os.path.getmtime(s) > os.path.getmtime(LIB) for s in srcs):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"])
    return LIB


class Oracle:
    def __init__(self, lib):
        self.lib = lib
        L = lib
        L.orc_view_matrix.argtypes = [_fp, _fp, C.c_int, C.c_int, _fp]
        L.orc_pad_image.argtypes = [_u8p, C.c_int, C.c_int, _u8p, C.c_int]
        L.orc_plane_table.argtypes = [C.c_int, C.c_float, C.c_float, _fp]
        L.orc_sweep.argtypes = [_fp, _u8p, C.c_int, C.c_int, C.c_int, _fp, C.POINTER(_u8p), C.c_int, C.c_float,
                                C.c_float, _u32p, _fp, _fp, _i32p, C.c_int]
        L.orc_argmin.argtypes = [_u32p, C.c_int, C.c_int, C.c_int, _fp, _fp, _fp, _i32p]
        L.orc_mix_background.argtypes = [_u8p, _u8p, _fp, _u8p, C.c_int, C.c_int]
        L.orc_sweep_sample.argtypes = [_fp, C.c_float, C.c_float, C.c_float, _u8p, C.c_int, C.c_int, C.c_int,
                                       C.POINTER(C.c_int)]
        L.orc_sweep_sample.restype = C.c_int
        L.orc_pixel_xn.restype = C.c_float
        L.orc_pixel_xn.argtypes = [C.c_int, C.c_int]
        L.orc_pixel_yn.restype = C.c_float
        L.orc_pixel_yn.argtypes = [C.c_int, C.c_int]
        for name, args in [
            ("orc_load_mesh", [_fp, C.c_int, _i32p, C.c_int, _fp]),
            ("orc_raster_window_z", [_fp, C.c_int, _fp, C.c_int, C.c_int, _fp]),
            ("orc_depth", [_fp, C.c_int, _fp, C.c_int, C.c_int, _fp]),
            ("orc_shadow_dilate", [_fp, C.c_int, C.c_int]),
            ("orc_projected", [_fp, C.c_int, _fp, _u8p, _fp, C.c_int, C.c_int, _u8p]),
            ("orc_projected_filter", [_fp, C.c_int, _fp, _u8p, _fp, C.c_int, C.c_int, C.c_int, _u8p]),
            ("orc_compare_u8", [_u8p, _u8p, C.c_int, C.c_int, _fp]),
            ("orc_compare_f32", [_fp, _fp, C.c_int, C.c_int, _fp]),
            ("orc_flow_remap", [_fp, C.c_int, _u8p, C.c_int, C.c_int, _u8p]),
            ("orc_resize_linear_u8", [_u8p, C.c_int, C.c_int, C.c_int, _u8p, C.c_int, C.c_int]),
        ]:
            if hasattr(L, name):
                getattr(L, name).argtypes = args

    @staticmethod
    def _p(a, t):
        return a.ctypes.data_as(t)

    def view_matrix(self, main_cam, side_cam, W, H):
        m = np.ascontiguousarray(main_cam, np.float32)
        s = np.ascontiguousarray(side_cam, np.float32)
        q = np.empty((3, 4), np.float32)
        self.lib.orc_view_matrix(self._p(m, _fp), self._p(s, _fp), W, H, self._p(q, _fp))
        return q

    def plane_table(self, D, z_lo, z_hi):
        z = np.empty(D, np.float32)
        self.lib.orc_plane_table(D, z_lo, z_hi, self._p(z, _fp))
        return z

    def pad_image(self, img):
        H, W = img.shape
        pad = np.empty((H + 2, W + 2), np.uint8)
        img = np.ascontiguousarray(img, np.uint8)
        self.lib.orc_pad_image(self._p(img, _u8p), W, H, self._p(pad, _u8p), W + 2)
        return pad

    def sweep(self, main_cam, main_img, side_cams, side_imgs, D, z_lo=-1.0, z_hi=1.0, want_volume=False,
              nthreads=1, sampler="exact"):
        """sampler "exact": contract v1 (f32 bilinear, cells count<<16 | sum); "fixed": contract v2 (1/32-texel positions,
        8-bit weight table, cells count<<24 | sum of |weights.texels - 255 I_main|)"""
        H, W = main_img.shape
        V = len(side_imgs)
        cam = np.ascontiguousarray(main_cam, np.float32)
        img = np.ascontiguousarray(main_img, np.uint8)
        cams = np.ascontiguousarray(np.asarray(side_cams, np.float32).reshape(max(V, 0), 16)) if V else np.zeros((1, 16), np.float32)
        frames = [np.ascontiguousarray(s, np.uint8) for s in side_imgs]
        arr = (_u8p * max(V, 1))(*[self._p(f, _u8p) for f in frames])
        depth = np.empty((H, W), np.float32)
        cost = np.empty((H, W), np.float32)
        idx = np.empty((H, W), np.int32)
        vol = np.empty((D, H, W), np.uint32) if want_volume else None
        fn = self.lib.orc_sweep if sampler == "exact" else self.lib.orc_sweep_fx
        fn.argtypes = self.lib.orc_sweep.argtypes
        fn(self._p(cam, _fp), self._p(img, _u8p), W, H, V, self._p(cams, _fp), arr, D, z_lo, z_hi,
           self._p(vol, _u32p) if want_volume else None, self._p(depth, _fp), self._p(cost, _fp),
           self._p(idx, _i32p), nthreads)
        return depth, cost, idx, vol

    def fx_weight_table(self):
        lut = np.empty((32, 32, 4), np.uint8)
        self.lib.orc_fx_weight_table.argtypes = [_u8p]
        self.lib.orc_fx_weight_table(self._p(lut, _u8p))
        return lut

    def sweep_sample_fx(self, q, xn, yn, z, pad):
        q = np.ascontiguousarray(q, np.float32)
        pad = np.ascontiguousarray(pad, np.uint8)
        Hp, pitch = pad.shape
        out = C.c_int(0)
        fn = self.lib.orc_sweep_sample_fx
        fn.argtypes = [_fp, C.c_float, C.c_float, C.c_float, _u8p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int)]
        fn.restype = C.c_int
        ok = fn(self._p(q, _fp), xn, yn, z, self._p(pad, _u8p), pitch, pitch - 2, Hp - 2, C.byref(out))
        return ok, out.value

    def warp_by_depth(self, main_cam, depth, side_cam, frame, sampler="exact"):
        H, W = frame.shape
        m, sc = np.ascontiguousarray(main_cam, np.float32), np.ascontiguousarray(side_cam, np.float32)
        d, f = np.ascontiguousarray(depth, np.float32), np.ascontiguousarray(frame, np.uint8)
        out = np.empty((H, W, 2), np.uint8)
        fn = self.lib.orc_warp_by_depth if sampler == "exact" else self.lib.orc_warp_by_depth_fx
        fn.argtypes = [_fp, _fp, _fp, _u8p, C.c_int, C.c_int, _u8p]
        fn(self._p(m, _fp), self._p(d, _fp), self._p(sc, _fp), self._p(f, _u8p), W, H, self._p(out, _u8p))
        return out

    def argmin(self, vol, z, sampler="exact"):
        D, H, W = vol.shape
        vol = np.ascontiguousarray(vol, np.uint32)
        z = np.ascontiguousarray(z, np.float32)
        depth = np.empty((H, W), np.float32)
        cost = np.empty((H, W), np.float32)
        idx = np.empty((H, W), np.int32)
        fn = self.lib.orc_argmin if sampler == "exact" else self.lib.orc_argmin_fx
        fn.argtypes = self.lib.orc_argmin.argtypes
        fn(self._p(vol, _u32p), W, H, D, self._p(z, _fp), self._p(depth, _fp), self._p(cost, _fp), self._p(idx, _i32p))
        return depth, cost, idx

    def refine_depth(self, vol, z, idx, sampler="exact"):
        """sub-plane parabola refinement of the selected planes `idx` over the packed volume"""
        D, H, W = vol.shape
        vol = np.ascontiguousarray(vol, np.uint32)
        z = np.ascontiguousarray(z, np.float32)
        idx = np.ascontiguousarray(idx, np.int32)
        depth = np.empty((H, W), np.float32)
        self.lib.orc_refine_depth.argtypes = [_u32p, C.c_int, C.c_int, C.c_int, _fp, _i32p, C.c_int, _fp]
        self.lib.orc_refine_depth(self._p(vol, _u32p), W, H, D, self._p(z, _fp), self._p(idx, _i32p), 16 if sampler == "exact" else 24, self._p(depth, _fp))
        return depth

    def mix_background(self, img3, bg, depth):
        H, W = bg.shape
        img3 = np.ascontiguousarray(img3, np.uint8)
        bg = np.ascontiguousarray(bg, np.uint8)
        depth = np.ascontiguousarray(depth, np.float32).copy()
        out = np.empty((H, W), np.uint8)
        self.lib.orc_mix_background(self._p(img3, _u8p), self._p(bg, _u8p), self._p(depth, _fp), self._p(out, _u8p), W, H)
        return out, depth


    # ---- renderer ------------------------------------------------------------------------------
    def load_mesh(self, verts4, faces3):
        v = np.ascontiguousarray(verts4, np.float32)
        f = np.ascontiguousarray(faces3, np.int32)
        soup = np.empty((f.shape[0], 9), np.float32)
        self.lib.orc_load_mesh(self._p(v, _fp), v.shape[0], self._p(f, _i32p), f.shape[0], self._p(soup, _fp))
        return soup

    def depth(self, soup, cam, W, H):
        cam = np.ascontiguousarray(cam, np.float32)
        out = np.empty((H, W), np.float32)
        self.lib.orc_depth(self._p(soup, _fp), soup.shape[0], self._p(cam, _fp), W, H, self._p(out, _fp))
        return out

    def raster_window_z(self, soup, cam, W, H):
        cam = np.ascontiguousarray(cam, np.float32)
        out = np.empty((H, W), np.float32)
        self.lib.orc_raster_window_z(self._p(soup, _fp), soup.shape[0], self._p(cam, _fp), W, H, self._p(out, _fp))
        return out

    def shadow_dilate(self, zwin_gl):
        a = np.ascontiguousarray(zwin_gl, np.float32).copy()
        H, W = a.shape
        self.lib.orc_shadow_dilate(self._p(a, _fp), W, H)
        return a

    def projected(self, soup, cam, frame, projector, mipmap=True):
        """Render::projected; mipmap: the frame texture with its mip chain (what the reference asks GL for) or level 0 only"""
        H, W = frame.shape
        cam = np.ascontiguousarray(cam, np.float32)
        prj = np.ascontiguousarray(projector, np.float32)
        frame = np.ascontiguousarray(frame, np.uint8)
        out = np.empty((H, W, 3), np.uint8)
        self.lib.orc_projected_filter(self._p(soup, _fp), soup.shape[0], self._p(cam, _fp), self._p(frame, _u8p),
                                      self._p(prj, _fp), W, H, 1 if mipmap else 0, self._p(out, _u8p))
        return out


    # ---- photometric helpers -------------------------------------------------------------------
    def compare(self, prev, nxt):
        H, W = prev.shape
        out = np.empty((H, W), np.float32)
        if prev.dtype == np.uint8:
            a, b = np.ascontiguousarray(prev), np.ascontiguousarray(nxt, np.uint8)
            self.lib.orc_compare_u8(self._p(a, _u8p), self._p(b, _u8p), W, H, self._p(out, _fp))
        else:
            a, b = np.ascontiguousarray(prev, np.float32), np.ascontiguousarray(nxt, np.float32)
            self.lib.orc_compare_f32(self._p(a, _fp), self._p(b, _fp), W, H, self._p(out, _fp))
        return out

    def flow_remap(self, flow, image):
        H, W = image.shape
        flow = np.ascontiguousarray(flow, np.float32)
        image = np.ascontiguousarray(image, np.uint8)
        out = np.empty((H, W), np.uint8)
        self.lib.orc_flow_remap(self._p(flow, _fp), flow.shape[2], self._p(image, _u8p), W, H, self._p(out, _u8p))
        return out

    def resize_linear(self, img, dw, dh):
        """cv::resize(img, Size(dw, dh)) INTER_LINEAR on u8, 1 or 3 channels (configuration.cpp:233)"""
        img = np.ascontiguousarray(img, np.uint8)
        sh, sw = img.shape[:2]
        ch = 1 if img.ndim == 2 else img.shape[2]
        out = np.empty((dh, dw) if img.ndim == 2 else (dh, dw, ch), np.uint8)
        self.lib.orc_resize_linear_u8(self._p(img, _u8p), sw, sh, ch, self._p(out, _u8p), dw, dh)
        return out

    def cubic_table(self):
        t = np.empty((1024, 16), np.int16)
        self.lib.orc_remap_cubic_table(t.ctypes.data_as(C.POINTER(C.c_short)))
        return t


    # ---- optical flow ---------------------------------------------------------------------------
    def farneback(self, prev, nxt, levels=10, pyr_scale=0.8, winsize=None, iterations=7, poly_n=None, poly_sigma=None):
        H, W = prev.shape
        poly_sigma = (H + W) / 1000.0 if poly_sigma is None else poly_sigma
        winsize = (H + W) // 100 if winsize is None else winsize
        poly_n = (5 if poly_sigma < 1.5 else 7) if poly_n is None else poly_n
        a, b = np.ascontiguousarray(prev, np.uint8), np.ascontiguousarray(nxt, np.uint8)
        flow = np.empty((H, W, 2), np.float32)
        f = self.lib.orc_farneback
        f.argtypes = [_u8p, _u8p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int, C.c_int, C.c_double, _fp]
        f(self._p(a, _u8p), self._p(b, _u8p), W, H, levels, pyr_scale, winsize, iterations, poly_n, poly_sigma, self._p(flow, _fp))
        return flow

    def variational_refine(self, prev, nxt, flow=None):
        H, W = prev.shape
        a, b = np.ascontiguousarray(prev, np.uint8), np.ascontiguousarray(nxt, np.uint8)
        flow = np.zeros((H, W, 2), np.float32) if flow is None else np.ascontiguousarray(flow, np.float32).copy()
        f = self.lib.orc_variational_refine
        f.argtypes = [_u8p, _u8p, C.c_int, C.c_int, _fp]
        f(self._p(a, _u8p), self._p(b, _u8p), W, H, self._p(flow, _fp))
        return flow

    def calculate_flow(self, prev, nxt, use_farneback):
        H, W = prev.shape
        a, b = np.ascontiguousarray(prev, np.uint8), np.ascontiguousarray(nxt, np.uint8)
        out = np.empty((H, W, 4), np.float32)
        f = self.lib.orc_calculate_flow
        f.argtypes = [_u8p, _u8p, C.c_int, C.c_int, C.c_int, _fp]
        f(self._p(a, _u8p), self._p(b, _u8p), W, H, 1 if use_farneback else 0, self._p(out, _fp))
        return out


    # ---- triangulation --------------------------------------------------------------------------
    def triangulate_pixels(self, flows, main_cam, side_cams, depth):
        H, W = depth.shape
        V = len(flows)
        fl = [np.ascontiguousarray(f, np.float32) for f in flows]
        arr = (_fp * max(V, 1))(*[self._p(f, _fp) for f in fl])
        cam = np.ascontiguousarray(main_cam, np.float32)
        cams = np.ascontiguousarray(np.asarray(side_cams, np.float32).reshape(V, 16)) if V else np.zeros((1, 16), np.float32)
        depth = np.ascontiguousarray(depth, np.float32)
        out = np.empty((H * W, 7), np.float32)
        f = self.lib.orc_triangulate_pixels
        f.restype = C.c_int
        f.argtypes = [C.POINTER(_fp), _fp, _fp, C.c_int, _fp, C.c_int, C.c_int, _fp]
        n = f(arr, self._p(cam, _fp), self._p(cams, _fp), V, self._p(depth, _fp), W, H, self._p(out, _fp))
        return out[:n].copy()


    # ---- point filter ---------------------------------------------------------------------------
    def filter_points(self, points4, alpha):
        pts = np.ascontiguousarray(points4, np.float32)
        N = pts.shape[0]
        keep = np.empty(max(N, 1), np.int32)
        dens = np.empty(max(N, 1), np.float32)
        f = self.lib.orc_filter_points
        f.restype = C.c_int
        f.argtypes = [_fp, C.c_int, C.c_float, _i32p, _fp]
        m = f(self._p(pts, _fp), N, alpha, self._p(keep, _i32p), self._p(dens, _fp))
        return keep[:m].copy(), dens[:N].copy()


_oracle = None


def load():
    global _oracle
    if _oracle is None:
        build()
        _oracle = Oracle(C.CDLL(LIB))
    return _oracle
