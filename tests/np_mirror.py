"""Independent float64 numpy restatement of one sweep sample plane (cross-check of the C oracle).

Not bit-exact by design: it evaluates the warp through the two 4x4 cameras in float64
(shader.frag:13-15, 17, 19, 22) without the oracle's folded f32 matrix, so agreement with the oracle
shows the folding/padding algebra is right; residual mismatches are confined to u8 rounding boundaries.
"""
import numpy as np


def warp_plane(main_cam, side_cam, side_img, z):
    """returns (Iq float64 rounded intensity (H,W), valid mask) for main-camera NDC plane z"""
    H, W = side_img.shape
    M = np.asarray(main_cam, np.float64)
    C = np.asarray(side_cam, np.float64)
    col = (2 * np.arange(W) + 1) / W - 1.0
    row = 1.0 - (2 * np.arange(H) + 1) / H
    xn, yn = np.meshgrid(col, row)
    ndc = np.stack([xn, yn, np.full_like(xn, z), np.ones_like(xn)], -1)  # H,W,4
    pos = ndc @ np.linalg.inv(M).T
    s = pos @ C.T
    sx, sy, sw = s[..., 0], s[..., 1], s[..., 3]
    # pos has w = 1/w_clip > 0 in front of the main camera, so sign(sw) is the side camera's w sign
    with np.errstate(divide="ignore", invalid="ignore"):
        nx, ny = sx / sw, sy / sw
    valid = (sw > 0) & (np.abs(nx) < 1) & (np.abs(ny) < 1)
    u = nx / 2 + 0.5
    v = ny / 2 + 0.5
    cf = u * W - 0.5
    rf = (1 - v) * H - 0.5
    cf = np.where(valid, cf, 0.0)
    rf = np.where(valid, rf, 0.0)
    i0 = np.floor(cf).astype(np.int64)
    j0 = np.floor(rf).astype(np.int64)
    ax = cf - i0
    ay = rf - j0
    img = side_img.astype(np.float64)

    def tap(j, i):
        return img[np.mod(j, H), np.mod(i, W)]  # GL_REPEAT

    top = tap(j0, i0) * (1 - ax) + tap(j0, i0 + 1) * ax
    bot = tap(j0 + 1, i0) * (1 - ax) + tap(j0 + 1, i0 + 1) * ax
    res = top * (1 - ay) + bot * ay
    return res, valid
