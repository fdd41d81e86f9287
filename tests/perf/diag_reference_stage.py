#!/usr/bin/env python3
"""Stage-by-stage GPU vs oracle comparison on the c1 reference-stage workload (diagnostic)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "mesh-reconstruction_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import mvs_amd, orc, tracks_yaml

def main():
    t = tracks_yaml.load("koberec.yaml")
    W, H = t["width"], t["height"]
    cams = t["cameras"]
    main = cams[40]; side_is = [30, 35, 45, 50]
    sides = np.stack([cams[i] for i in side_is])
    b = t["bundles"]; xyz = b[:, :3] / b[:, 3:4]; c = xyz.mean(0); ext = 1.5 * np.abs(xyz - c).max()
    n = 48; g = np.linspace(-ext, ext, n); X, Y = np.meshgrid(g, g)
    verts = np.stack([c[0] + X.ravel(), c[1] + Y.ravel(), np.full(n * n, c[2]), np.ones(n * n)], 1).astype(np.float32)
    idx = np.arange(n * n).reshape(n, n)
    a_, b_, c_, d_ = idx[:-1, :-1].ravel(), idx[:-1, 1:].ravel(), idx[1:, :-1].ravel(), idx[1:, 1:].ravel()
    faces = np.concatenate([np.stack([a_, b_, c_], 1), np.stack([b_, d_, c_], 1)]).astype(np.int32)
    rng = np.random.default_rng(5); yy, xx = np.mgrid[0:H, 0:W]
    def frame(k):
        return (127 + 60 * np.sin((xx + 3 * k) / 19.0) * np.cos((yy - 2 * k) / 23.0) + rng.normal(0, 3, (H, W))).clip(0, 255).astype(np.uint8)
    main_img = frame(0); side_imgs = [frame(k + 1) for k in range(4)]
    o = orc.load(); soup = o.load_mesh(verts, faces)
    def cmp(name, a, b):
        a = np.asarray(a); b = np.asarray(b)
        if a.shape != b.shape:
            print(f"{name}: SHAPE {a.shape} vs {b.shape}"); return
        eq = (a == b) | (np.isnan(a.astype(np.float64)) & np.isnan(b.astype(np.float64)))
        nbad = int((~eq).sum())
        msg = f"{name}: {nbad} / {a.size} differ"
        if nbad:
            w = np.argwhere(~eq)[:5]
            msg += " first " + "; ".join(f"{tuple(i)} gpu={a[tuple(i)]!r} orc={b[tuple(i)]!r}" for i in w)
            msg += f" maxabs={np.nanmax(np.abs(a.astype(np.float64) - b.astype(np.float64))[~eq])}"
        print(msg)
    with mvs_amd.Context(W, H) as ctx:
        ctx.load_mesh(verts, faces)
        d_g = ctx.depth(main); d_o = o.depth(soup, main, W, H)
        print("depth coverage", float((d_o < 1).mean()), "range", float(d_o.min()), float(d_o.max()))
        cmp("depth", d_g, d_o)
        flows_g, flows_o = [], []
        dg, do = d_g, d_o
        for k, (cam, img) in enumerate(zip(sides, side_imgs)):
            p_g = ctx.projected(main, img, cam); p_o = o.projected(soup, main, img, cam)
            cmp(f"projected[{k}]", p_g, p_o)
            m_g, dg = ctx.mix_background(p_g, main_img, dg); m_o, do = o.mix_background(p_o, main_img, do)
            cmp(f"mixed[{k}]", m_g, m_o); cmp(f"depth_after[{k}]", dg, do)
            f_g = ctx.flow(main_img, m_o, False); f_o = o.calculate_flow(main_img, m_o, False)
            cmp(f"flow[{k}] (same input)", f_g, f_o)
            flows_g.append(f_g); flows_o.append(f_o)
        t_g = ctx.triangulate(flows_o, main, sides, do); t_o = o.triangulate_pixels(flows_o, main, sides, do)
        cmp("triangulate pos (same input)", t_g[:, :4], t_o[:, :4])
        if t_g.shape == t_o.shape:
            print("  normals/pdf max rel", float(np.nanmax(np.abs(t_g[:, 4:] - t_o[:, 4:]))))
        pf = ctx.process_frame(main, main_img, sides, side_imgs, False)
        cmp("process_frame pos vs oracle chain", pf[:, :4], t_o[:, :4])
        cmp("process_frame pos vs gpu stagewise", pf[:, :4], t_g[:, :4])
main()
