"""numpy restatement of Configuration::estimateExposure (reference configuration.cpp:248-426, util.cpp:408-433) -- the checker
for the C++ host mirror's implementation (test infrastructure, like oracle/).  Follows the reference step by step with
float32 where it uses float; the per-frame least squares uses numpy's SVD pseudo-inverse, as `inv(DECOMP_SVD)` does."""
import numpy as np

f32 = np.float32


def sample_image(image, radius_squared, x, y, channel):
    """util.cpp:408-433: mean of the unclipped values of one channel inside a disc, -1 if there is none"""
    H, W = image.shape[:2]
    radius = np.sqrt(f32(radius_squared))
    total, count = f32(0), 0
    ny = int(max(0.0, y - radius))
    while ny < min(y + radius + 1, H):
        nx = int(max(0.0, x - radius))
        while nx < min(x + radius + 1, W):
            dx, dy = f32(nx) - x, f32(ny) - y
            val = int(image[ny, nx, channel])
            if dx * dx + dy * dy <= radius_squared and 0 < val < 255:
                total = f32(total + f32(val))
                count += 1
            nx += 1
        ny += 1
    return f32(total / f32(count)) if count else f32(-1)


def project_points(camera, bundles, distortion, width, height):
    """configuration.cpp:248-267"""
    cam, b = camera.astype(f32), bundles.astype(f32)
    proj = np.zeros((b.shape[0], 4), f32)
    for j in range(b.shape[0]):
        for r in range(4):
            s = f32(0)
            for k in range(4):
                s = f32(s + f32(cam[r, k] * b[j, k]))
            proj[j, r] = s
    cart = (proj[:, :3] / proj[:, 3:4]).astype(f32)
    aspect = f32(height) / f32(width)
    for p in cart:
        rad = f32(f32(p[0] * p[0] + f32(f32(p[1] * p[1]) * aspect) * aspect) / f32(4))
        k = f32(1) + rad * f32(f32(distortion[0]) + rad * f32(distortion[1]))
        p *= f32(k)
    return cart


def estimate_exposure(frames_bgr, cameras, bundles, enabled, distortion, center_x, center_y):
    """-> (exposure [channels, frames] f32, grey frames list of u8 arrays)"""
    F, N = len(frames_bgr), bundles.shape[0]
    H, W, ch = frames_bgr[0].shape
    sampled, ids, begin = [], -np.ones((F, N), int), [0]
    for i in range(F):
        re = project_points(cameras[i], bundles, distortion, W, H)
        for j in range(N):
            if i not in enabled[j]:
                continue
            x = f32(f32(center_x) + f32(f32(re[j, 0] * f32(W)) * f32(0.5)))
            y = f32(f32(f32(H) - f32(center_y)) - f32(f32(re[j, 1] * f32(H)) * f32(0.5)))
            sc = [sample_image(frames_bgr[i], f32(16), x, y, c) for c in range(ch)]
            if any(v == -1 for v in sc):
                continue
            ids[i, j] = len(sampled)
            sampled.append(sc)
        begin.append(len(sampled))
        assert begin[-1] - begin[-2] >= ch
    S = np.array(sampled, f32)
    sum_brightness = float(S.astype(np.float64).sum()) / ch
    exposure = np.full((ch, F), f32(1) / f32(ch), f32)
    pb = np.ones(N, f32)
    for _ in range(100):
        error, current = 0.0, 0.0
        for j in range(N):
            s, wsum = f32(0), 0
            for i in range(F):
                r = ids[i, j]
                if r == -1:
                    continue
                wsum += 1
                for c in range(ch):
                    s = f32(s + f32(S[r, c] * exposure[c, i]))
            current += float(s)
            pb[j] = f32(s / f32(wsum)) if wsum else f32(0)
        pb = (pb * f32(sum_brightness / current)).astype(f32)
        for i in range(F):
            A = S[begin[i]:begin[i + 1]]
            vb = np.array([pb[j] for j in range(N) if ids[i, j] >= 0], f32)
            x = (np.linalg.pinv(A.astype(np.float64)) @ vb.astype(np.float64)).astype(f32)
            omega = f32(0.4)
            exposure[:, i] = x * (f32(1) + omega) - exposure[:, i] * omega
            error += float(np.linalg.norm(A @ exposure[:, i] - vb)) / len(vb)
        if error / F < 0.1:
            break
    greys = []
    for i in range(F):
        acc = np.zeros((H, W), f32)
        for c in range(ch):
            acc = np.clip(np.rint(frames_bgr[i][..., c].astype(f32) * exposure[c, i] + acc), 0, 255).astype(f32)
        greys.append(acc.astype(np.uint8))
    return exposure, greys
