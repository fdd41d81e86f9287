"""Deterministic synthetic inputs for the dense-MVS benchmark and tests.

Cameras follow the reference's convention (io_export_tracks.py:22-28, 59-66):
P = K [R|t] with w = -z_camera > 0 in front, NDC z = -1 at near, +1 at far.
Harness code: numpy only, no GPU, no oracle.
"""
import numpy as np

FOVX = 0.9186  # zatisi.yaml clip fov (0.91858...)
NEAR, FAR = 1.85, 7.70  # zatisi frame-1 near/far
SEED_SCENE = 0x5EED0001
SEED_NOISE = 0x5EED0002


def splitmix64(seed, n):
    """n uniform doubles in [0,1) from splitmix64(seed)"""
    out = np.empty(n, np.float64)
    x = np.uint64(seed)
    M = (1 << 64) - 1
    xi = int(x)
    for i in range(n):
        xi = (xi + 0x9E3779B97F4A7C15) & M
        z = xi
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M
        z = z ^ (z >> 31)
        out[i] = (z >> 11) * (1.0 / (1 << 53))
    return out


def perspective(fovx, aspect, near, far):
    """io_export_tracks.py:22-28 applied to a camera looking down -z (flip of io_export_tracks.py:59-62)"""
    K = np.array([[2.0 / fovx, 0, 0, 0],
                  [0, 2.0 * aspect / fovx, 0, 0],
                  [0, 0, (far + near) / (far - near), 2.0 * far * near / (near - far)],
                  [0, 0, 1.0, 0]], np.float64)
    flip = np.diag([1.0, 1.0, -1.0, 1.0])
    return K @ flip


def camera_at(center, W, H, fovx=FOVX, near=NEAR, far=FAR, rot=None):
    """4x4 projection of a camera at `center` (world), axes parallel to world unless `rot` (3x3 world->camera)"""
    Rt = np.eye(4)
    if rot is not None:
        Rt[:3, :3] = rot
    Rt[:3, 3] = -Rt[:3, :3] @ np.asarray(center, np.float64)
    return (perspective(fovx, W / H, near, far) @ Rt).astype(np.float32)


def ring_cameras(V, W, H, radius=0.15):
    """main camera at the origin + V side cameras on a circle of `radius` in the z = 0 plane"""
    main = camera_at([0, 0, 0], W, H)
    sides = []
    for v in range(V):
        a = 2.0 * np.pi * v / max(V, 1)
        sides.append(camera_at([radius * np.cos(a), radius * np.sin(a), 0.0], W, H))
    return main, np.stack(sides) if V else np.zeros((0, 4, 4), np.float32)


class Scene:
    """analytic height field z(x,y) = -3.0 - 0.4 sin(1.3x+0.7) cos(1.1y-0.2) with a 6-term sinusoid albedo"""

    def __init__(self, seed=SEED_SCENE, freq_scale=1.0):
        """freq_scale = W/1920 keeps the texture's wavelengths (22..290 px at 1080p) fixed in pixels"""
        u = splitmix64(seed, 30)
        self.a = 0.5 + u[0:6]
        self.f = (15.0 + 185.0 * u[6:12]) * freq_scale
        self.g = (15.0 + 185.0 * u[12:18]) * freq_scale
        self.phi = 2 * np.pi * u[18:24]
        self.psi = 2 * np.pi * u[24:30]

    @staticmethod
    def height(x, y):
        return -3.0 - 0.4 * np.sin(1.3 * x + 0.7) * np.cos(1.1 * y - 0.2)

    def albedo(self, X, Y):
        acc = np.zeros_like(X)
        for k in range(6):
            acc += self.a[k] * np.sin(self.f[k] * X + self.phi[k]) * np.sin(self.g[k] * Y + self.psi[k])
        return 127.5 + 127.5 * acc / self.a.sum()

    def render(self, center, W, H, fovx=FOVX, want_depth=False, near=NEAR, far=FAR):
        """ray-cast the height field from an axis-parallel camera at `center`; returns u8 image (H,W)
        and optionally the NDC depth map of that camera"""
        cx, cy, cz = [float(c) for c in center]
        aspect = W / H
        col = (np.arange(W, dtype=np.float64) * 2 + 1) / W - 1.0
        row = 1.0 - (np.arange(H, dtype=np.float64) * 2 + 1) / H
        dx = (col * fovx / 2.0)[None, :].repeat(H, 0)
        dy = (row * fovx / (2.0 * aspect))[:, None].repeat(W, 1)
        # point = c + t (dx, dy, -1); solve cz - t = height(cx + t dx, cy + t dy) by fixed-point iteration
        t = np.full((H, W), 3.0 + cz)
        for _ in range(14):
            t = cz - self.height(cx + t * dx, cy + t * dy)
        X, Y = cx + t * dx, cy + t * dy
        img = np.clip(np.rint(self.albedo(X, Y)), 0, 255).astype(np.uint8)
        if not want_depth:
            return img
        zn = (far + near) / (far - near) + (2.0 * far * near / (near - far)) / t  # NDC z for w = t
        return img, zn.astype(np.float32)


def make_views(W, H, V, radius=0.15, seed=SEED_SCENE, freq_scale=None):
    """synthetic multi-view set: (main_cam, main_img, side_cams[V], side_imgs[V], main_depth_ndc)"""
    sc = Scene(seed, W / 1920.0 if freq_scale is None else freq_scale)
    main_cam, side_cams = ring_cameras(V, W, H, radius)
    main_img, depth = sc.render([0, 0, 0], W, H, want_depth=True)
    sides = []
    for v in range(V):
        a = 2.0 * np.pi * v / max(V, 1)
        sides.append(sc.render([radius * np.cos(a), radius * np.sin(a), 0.0], W, H))
    return main_cam, main_img, side_cams, sides, depth


def noise_views(W, H, V, seed=SEED_NOISE):
    """i.i.d. uniform u8 frames (adversarial for locality; no depth meaning)"""
    rng = np.random.Generator(np.random.PCG64(seed))
    main_cam, side_cams = ring_cameras(V, W, H)
    main_img = rng.integers(0, 256, (H, W), dtype=np.uint8)
    sides = [rng.integers(0, 256, (H, W), dtype=np.uint8) for _ in range(V)]
    return main_cam, main_img, side_cams, sides
