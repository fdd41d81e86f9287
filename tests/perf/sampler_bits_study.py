"""is the fixed sampler's 5-bit sub-texel grid the knee, or just the cheapest point?  

A numpy restatement of the sweep with a parametrised sampler -- positions quantised to 1/2^b texel, 8-bit weights from a 2^b x 2^b
table built like the library's 32 x 32 one -- for b = 5 (the library's fixed sampler), 6 and 8, against the exact-f32 sampler
(bilinear in double precision here, rounded to u8: the f32 rounding noise is three orders below the effects measured), on the SURVEY 8d
scene at 640 x 360, 64 planes, 8 views.  For each: how many pixels select another plane than the exact sampler (and by more than one
plane), depth RMSE between the samplers and against the analytic ground truth -- each WITHOUT and WITH the sub-plane parabola
refinement (mvs_sweep_refine_depth / orc_refine_depth's formula).  CPU only (numpy); writes JSON to stdout:
    python tests/perf/sampler_bits_study.py > profiles/r03/sampler_bits_study.json"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "mesh-reconstruction_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from mvs_amd import synth  # noqa: E402
import orc  # noqa: E402


def weight_table(bits):
    n = 1 << bits
    a = np.arange(n)
    A, B = np.meshgrid(a, a)  # A: x phase, B: y phase
    p = np.stack([(n - A) * (n - B), A * (n - B), (n - A) * B, A * B], -1).astype(np.int64)
    w = (255 * p + n * n // 2) // (n * n)
    big = p.argmax(-1)
    fix = 255 - w.sum(-1)
    np.put_along_axis(w, big[..., None], np.take_along_axis(w, big[..., None], -1) + fix[..., None], -1)
    return w  # [ky][kx][4]


def sweep(main_img, sides, Q, z, bits):
    """returns (sum, count) volumes [D, H, W]; bits None = exact sampler"""
    H, W = main_img.shape
    D, V = len(z), len(sides)
    col, row = np.arange(W, dtype=np.float64), np.arange(H, dtype=np.float64)
    xn, yn = (2 * col + 1) / W - 1, 1 - (2 * row + 1) / H
    XN, YN = np.meshgrid(xn, yn)
    tab = weight_table(bits) if bits else None
    S = np.zeros((D, H, W), np.float64)
    C = np.zeros((D, H, W), np.int32)
    I = main_img.astype(np.float64)
    for v in range(V):
        q = Q[v].astype(np.float64).reshape(3, 4)
        pad = np.pad(sides[v].astype(np.float64), 1, mode="wrap")
        ax, ay, aw = q[0, 0] * XN + q[0, 1] * YN + q[0, 3], q[1, 0] * XN + q[1, 1] * YN + q[1, 3], q[2, 0] * XN + q[2, 1] * YN + q[2, 3]
        for d in range(D):
            sw = z[d] * q[2, 2] + aw
            cx, cy = (z[d] * q[0, 2] + ax) / sw, (z[d] * q[1, 2] + ay) / sw
            if bits:
                n = 1 << bits
                ux, uy = np.rint(cx * n).astype(np.int64), np.rint(cy * n).astype(np.int64)
                ok = (sw > 0) & (ux > n // 2) & (ux < W * n + n // 2) & (uy > n // 2) & (uy < H * n + n // 2)
                ux, uy = np.clip(ux, n, W * n), np.clip(uy, n, H * n)
                ix, iy, kx, ky = ux >> bits, uy >> bits, ux & (n - 1), uy & (n - 1)
                w = tab[ky, kx]
                dot = w[..., 0] * pad[iy, ix] + w[..., 1] * pad[iy, ix + 1] + w[..., 2] * pad[iy + 1, ix] + w[..., 3] * pad[iy + 1, ix + 1]
                cost = np.abs(dot - 255.0 * I) / 255.0
            else:
                ok = (sw > 0) & (cx > 0.5) & (cx < W + 0.5) & (cy > 0.5) & (cy < H + 0.5)
                cxx, cyy = np.clip(cx, 0.5, W + 0.5), np.clip(cy, 0.5, H + 0.5)
                ix, iy = np.floor(cxx).astype(np.int64), np.floor(cyy).astype(np.int64)
                fx, fy = cxx - ix, cyy - iy
                val = (1 - fy) * ((1 - fx) * pad[iy, ix] + fx * pad[iy, ix + 1]) + fy * ((1 - fx) * pad[iy + 1, ix] + fx * pad[iy + 1, ix + 1])
                cost = np.abs(np.floor(val + 0.5) - I)
            S[d] += np.where(ok, cost, 0.0)
            C[d] += ok
    return S, C


def select(S, C, z):
    mean = np.where(C > 0, S / np.maximum(C, 1), np.inf)
    idx = mean.argmin(0)
    D = len(z)
    i = np.clip(idx, 1, D - 2)
    ca, cb, cc = np.take_along_axis(mean, (i - 1)[None], 0)[0], np.take_along_axis(mean, i[None], 0)[0], np.take_along_axis(mean, (i + 1)[None], 0)[0]
    den = ca - 2 * cb + cc
    t = np.where((den > 0) & np.isfinite(ca) & np.isfinite(cc) & (idx == i), np.clip(0.5 * (ca - cc) / np.where(den > 0, den, 1), -0.5, 0.5), 0.0)
    step = z[1] - z[0]
    return idx, z[idx], z[idx] + t * step


def main():
    W, H, D, V = 640, 360, 64, 8
    main_cam, main_img, side_cams, sides, gt = synth.make_views(W, H, V)
    o = orc.load()
    Q = [o.view_matrix(main_cam, side_cams[v], W, H) for v in range(V)]
    z = (-1 + 2 * (np.arange(D) + 0.5) / D)
    inner = np.s_[16:-16, 16:-16]
    res = {}
    ref = None
    for name, bits in (("exact", None), ("fixed_5_bits", 5), ("fixed_6_bits", 6), ("fixed_8_bits", 8)):
        S, C = sweep(main_img, sides, Q, z, bits)
        idx, d0, d1 = select(S, C, z)
        if ref is None:
            ref = (idx, d0, d1)
        r = {"depth_rmse_vs_ground_truth": float(np.sqrt(np.mean((d0 - gt)[inner] ** 2))),
             "depth_rmse_vs_ground_truth_refined": float(np.sqrt(np.mean((d1 - gt)[inner] ** 2)))}
        if name != "exact":
            r.update({"plane_flip_rate_vs_exact": float(np.mean(idx != ref[0])), "flips_by_more_than_one_plane": float(np.mean(np.abs(idx - ref[0]) > 1)),
                      "depth_rmse_vs_exact": float(np.sqrt(np.mean((d0 - ref[1]) ** 2))), "depth_rmse_vs_exact_refined": float(np.sqrt(np.mean((d1 - ref[2]) ** 2)))})
        res[name] = r
        print(name, r, file=sys.stderr)
    print(json.dumps({"workload": "SURVEY 8d scene, %dx%d, %d planes, %d views; plane step %.5f" % (W, H, D, V, 2.0 / D), "samplers": res}, indent=1))


if __name__ == "__main__":
    main()
