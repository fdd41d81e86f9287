"""Regenerates the small golden vectors from the CPU oracle (run in the dev container):
    python tests/golden/make_golden.py
The reference holds no golden outputs for this path, so these pin the oracle
against silent drift, not against the reference."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "mesh-reconstruction_amd", "python"))
import orc  # noqa: E402
from mvs_amd import synth  # noqa: E402


def main():
    o = orc.load()
    W, H, D, V = 48, 32, 12, 3
    main_cam, main_img, side_cams, sides, gt = synth.make_views(W, H, V, radius=0.3, freq_scale=0.3)
    depth, cost, idx, vol = o.sweep(main_cam, main_img, side_cams, sides, D, want_volume=True)
    np.savez_compressed(os.path.join(HERE, "sweep_small.npz"), main_cam=main_cam, main_img=main_img,
                        side_cams=side_cams, side_imgs=np.stack(sides), D=D, depth=depth, cost=cost, idx=idx, vol=vol,
                        gt=gt)
    print("wrote sweep_small.npz", depth.shape, vol.shape)
    # the same inputs through the fixed sampler (contract v2): only the outputs are stored
    depth, cost, idx, vol = o.sweep(main_cam, main_img, side_cams, sides, D, want_volume=True, sampler="fixed")
    np.savez_compressed(os.path.join(HERE, "sweep_small_fx.npz"), D=D, depth=depth, cost=cost, idx=idx, vol=vol, lut=o.fx_weight_table())
    print("wrote sweep_small_fx.npz", depth.shape, vol.shape)
    stage_small(o)
    rect_small(o)


def rect_small(o):
    """a geometry the RECTIFIED-view kernels serve (sweep_fx_rect / sweep_exact_rect: the SURVEY 8d ring at a size whose boxes fit their LDS
    slots), both samplers: inputs, depth / index / best cost in full, the packed volume as a checksum and a 1-in-4 probe"""
    W, H, D, V = 192, 40, 32, 3
    main_cam, main_img, side_cams, sides, _ = synth.make_views(W, H, V, radius=0.1, freq_scale=0.4)
    out = {"main_cam": main_cam, "main_img": main_img, "side_cams": side_cams, "side_imgs": np.stack(sides), "D": D}
    for sampler in ("exact", "fixed"):
        depth, cost, idx, vol = o.sweep(main_cam, main_img, side_cams, sides, D, want_volume=True, sampler=sampler)
        out.update({"depth_" + sampler: depth, "cost_" + sampler: cost, "idx_" + sampler: idx.astype(np.int16), "vol_crc_" + sampler: _crc(vol),
                    "vol_probe_" + sampler: vol[::2, ::2, ::2].copy()})
    np.savez_compressed(os.path.join(HERE, "sweep_rect_small.npz"), **out)
    print("wrote sweep_rect_small.npz", depth.shape, vol.shape)


def _crc(a):
    import zlib
    return np.uint32(zlib.crc32(np.ascontiguousarray(a).tobytes()))


def stage_inputs():
    """the reference's per-frame stage (recon.cpp:65-117) at 96x64 with two side views: deterministic inputs"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import scenes
    W, H = 96, 64
    verts, faces = scenes.heightfield_mesh(24, extent=1.4)
    sc = synth.Scene(freq_scale=0.2)
    main_c, side_cs = [0.0, 0.0, 0.0], [[0.15, 0.0, 0.0], [-0.1, 0.12, 0.03]]
    main = synth.camera_at(main_c, W, H)
    sides = np.stack([synth.camera_at(c, W, H) for c in side_cs])
    return W, H, verts, faces, main, sides, sc.render(main_c, W, H), [sc.render(c, W, H) for c in side_cs]


def stage_outputs(o):
    W, H, verts, faces, main, sides, main_img, side_imgs = stage_inputs()
    soup = o.load_mesh(verts, faces)
    out = {}
    d = o.depth(soup, main, W, H)
    out["depth"] = d.copy()
    flows = []
    for k, (cam, img) in enumerate(zip(sides, side_imgs)):
        proj = o.projected(soup, main, img, cam)
        mixed, d = o.mix_background(proj, main_img, d)
        out["projected%d_crc" % k] = _crc(proj)
        out["mixed%d" % k] = mixed.copy()
        fv = o.calculate_flow(main_img, mixed, False)
        ff = o.calculate_flow(main_img, mixed, True)
        # large float maps: a checksum of every bit plus a 1-in-16 probe (enough to see WHERE a drift starts)
        for name, arr in (("compare", o.compare(main_img, mixed)), ("flow_var", fv), ("flow_fb", ff)):
            out["%s%d_crc" % (name, k)] = _crc(arr)
            out["%s%d_probe" % (name, k)] = arr[::4, ::4].copy()
        out["remap%d" % k] = o.flow_remap(fv, mixed)
        flows.append(fv)
    out["depth_after"] = d.copy()
    pts = o.triangulate_pixels(flows, main, sides, d)
    out["points_n"] = np.int32(pts.shape[0])
    out["points_xyzw_crc"] = _crc(pts[:, :4])
    out["points_probe"] = pts[::16].copy()
    cloud = pts[:, :4].copy()
    cloud = cloud[np.isfinite(cloud).all(1)]
    keep, _ = o.filter_points(cloud, 0.02)
    out["filter_keep_crc"] = _crc(np.asarray(keep, np.int32))
    out["filter_keep_n"] = np.int32(len(keep))
    out["filter_n"] = np.int32(cloud.shape[0])
    return out


def stage_small(o):
    out = stage_outputs(o)
    np.savez_compressed(os.path.join(HERE, "stage_small.npz"), **out)
    print("wrote stage_small.npz", {k: getattr(v, "shape", ()) for k, v in out.items()})


if __name__ == "__main__":
    main()
