"""CPU restatements for the surface-meshing row.  TEST INFRASTRUCTURE ONLY: imported by tests/ alone.

alpha_shape(points): alphaShapeFaces (alpha_shapes.cpp:36-99 of the reference) from CGAL's published definitions, with the Delaunay
triangulation taken from scipy (Qhull) -- an implementation independent of host/alpha_shapes.cpp's own triangulator:
    REGULARIZED alpha complex: a finite cell is interior iff squared circumradius <= alpha; a facet is REGULAR iff exactly one of its
    cells is interior; find_optimal_alpha(1) as CGAL's Alpha_shape_3.h states it (see host/alpha_shapes.cpp's header).
poisson(points, normals, G, smooth): csrc/poisson.hip's grid Poisson reconstruction in float64 numpy (same box, same fixed-point
splat, numpy FFT, same surface-nets rules).
enforce_facet_criteria(vertices, faces, angle, radius, distance): csrc/surface_criteria.cpp's pass for the facet criteria the reference
hands CGAL (cgal_poisson.cpp:50-52, 95-97), operation for operation.
PARITY UNPINNED against the reference: CGAL / PCL are third-party packages that are neither in this image nor vendored by the
reference, and the reference holds no golden output for either function (its TEST_BUILD mains read test/bunny_5000, not shipped)."""
import numpy as np


def squared_circumradius(p, q, r, s):
    """CGAL's squared_radiusC3, float64, vectorised over rows."""
    qp, rp, sp = q - p, r - p, s - p
    qp2, rp2, sp2 = (qp * qp).sum(1), (rp * rp).sum(1), (sp * sp).sum(1)

    def det3(a, b, c):
        return (a[:, 0] * (b[:, 1] * c[:, 2] - b[:, 2] * c[:, 1]) - a[:, 1] * (b[:, 0] * c[:, 2] - b[:, 2] * c[:, 0]) + a[:, 2] * (b[:, 0] * c[:, 1] - b[:, 1] * c[:, 0]))

    def rows(c0, c1):
        return [np.stack([m[:, c0], m[:, c1], m2], 1) for m, m2 in ((qp, qp2), (rp, rp2), (sp, sp2))]

    num_x = det3(*rows(1, 2))
    num_y = det3(*rows(0, 2))
    num_z = det3(*rows(0, 1))
    den = det3(qp, rp, sp)
    return (num_x ** 2 + num_y ** 2 + num_z ** 2) / (4.0 * den * den)


def alpha_shape(points, forced_alpha=0.0):
    """-> (faces F x 3 of input row indices, normals out of the solid; alpha; solid components; cells)"""
    from scipy.spatial import Delaunay
    pts = np.asarray(points, np.float32)
    if pts.shape[1] == 4:
        pts = (pts[:, :3] / pts[:, 3:4]).astype(np.float32)
    xyz = pts.astype(np.float64)
    uniq, inverse = np.unique(xyz, axis=0, return_inverse=True)
    row_of = np.zeros(len(uniq), np.int64)
    row_of[inverse.ravel()] = np.arange(len(xyz))        # the last row with those coordinates wins (alpha_shapes.cpp:49)
    tri = Delaunay(uniq, qhull_options="Qt Qbb Qc")
    cells = tri.simplices
    nb = tri.neighbors                                     # -1: beyond the hull
    a = squared_circumradius(*(uniq[cells[:, k]] for k in range(4)))
    vmin = np.full(len(uniq), np.inf)
    for k in range(4):
        np.minimum.at(vmin, cells[:, k], a)
    alpha_solid = vmin.max()
    spectrum = np.unique(a[a > 0])

    def components(alpha):
        inside = a <= alpha
        seen = np.zeros(len(cells), bool)
        count = 0
        for c in np.flatnonzero(inside):
            if seen[c]:
                continue
            count += 1
            stack = [c]
            seen[c] = True
            while stack:
                x = stack.pop()
                for y in nb[x]:
                    if y >= 0 and inside[y] and not seen[y]:
                        seen[y] = True
                        stack.append(y)
        return count

    first = int(np.searchsorted(spectrum, alpha_solid, side="left"))
    first = min(first, len(spectrum) - 1)
    if components(alpha_solid) != 1:
        length = len(spectrum) - first - 1
        while length > 0:
            half = length // 2
            middle = first + half
            if components(spectrum[middle]) > 1:
                first = middle + 1
                length = length - half - 1
            else:
                length = half
    opt = first + 1 if first + 1 < len(spectrum) else first
    alpha_opt = spectrum[opt]
    alpha = forced_alpha if forced_alpha > 0 else alpha_opt
    inside = a <= alpha
    faces = []
    for c in np.flatnonzero(inside):
        for i in range(4):
            y = nb[c, i]
            if y >= 0 and inside[y]:
                continue
            f = [cells[c, k] for k in range(4) if k != i]
            p0, p1, p2, pv = uniq[f[0]], uniq[f[1]], uniq[f[2]], uniq[cells[c, i]]
            if np.dot(np.cross(p1 - p0, p2 - p0), pv - p0) > 0:
                f[1], f[2] = f[2], f[1]
            faces.append([row_of[f[0]], row_of[f[1]], row_of[f[2]]])
    return np.array(faces, np.int32).reshape(-1, 3), float(np.float32(alpha_opt)), components(alpha), row_of[cells]


# ---------------------------------------------------------------------------------------------------------------------------------
# grid Poisson reconstruction (csrc/poisson.hip)
SPLAT_SCALE = np.float32(65536.0)


def average_spacing(points, k=6):
    """CGAL::compute_average_spacing(points, k) (cgal_poisson.cpp:77): CGAL queries k + 1 neighbours per sample -- the sample itself comes
    first, at distance 0 -- and divides the sum of the distances by the k + 1 points visited (compute_average_spacing.h, as recalled: CGAL
    is not in this image); averaged over the samples; scipy's k-d tree, float64"""
    from scipy.spatial import cKDTree
    p = np.asarray(points, np.float32)
    xyz = (p[:, :3] / p[:, 3:4]).astype(np.float64)
    kk = min(k, len(xyz) - 1)
    d, _ = cKDTree(xyz).query(xyz, k=kk + 1)
    return float((d.sum(1) / (kk + 1)).mean())


def poisson_grid(points, grid_log2):
    """the box and grid csrc/poisson.hip chooses: (G, origin float32[3], h float32); grid_log2 = 0: the coarsest of 32..512 nodes per
    axis with node spacing <= 0.75 x the samples' average 6-nearest-neighbour spacing"""
    p = np.asarray(points, np.float32)
    xyz = (p[:, :3] / p[:, 3:4]).astype(np.float64)
    lo, hi = xyz.min(0), xyz.max(0)
    side = float((hi - lo).max())
    lg = grid_log2
    if lg == 0:
        sp = average_spacing(p)
        lg = 5
        while lg < 9 and 1.5 * side / float((1 << lg) - 1) > 0.75 * sp:
            lg += 1
    G = 1 << lg
    box = 1.5 * side
    h = np.float32(box / float(G - 1))
    origin = (0.5 * (lo + hi) - 0.5 * box).astype(np.float32)
    return G, origin, h


def normal_scale_log2(normals):
    """csrc/poisson.hip: the power of two the normals are multiplied by before the fixed-point splat -- the lower median of
    max(|nx|, |ny|, |nz|) over the usable (finite, every component <= 1e4), non-zero normals lands in [0.5, 1); lowered until the largest
    usable component times it stays within 1e4; 0 when there is no such normal"""
    import math
    a = np.abs(np.asarray(normals, np.float32))
    with np.errstate(invalid="ignore"):
        ok = np.all(a <= np.float32(1e4), axis=1)
    m = a[ok].max(1) if ok.any() else np.zeros(0, np.float32)
    m = np.sort(m[m > 0])
    if len(m) == 0:
        return 0
    k = max(-100, min(100, -math.frexp(float(m[(len(m) - 1) // 2]))[1]))
    while k > -100 and math.ldexp(float(m[-1]), k) > 1e4:
        k -= 1
    return k


def poisson_splat(points, normals, G, origin, h):
    """the four fixed-point fields [vx, vy, vz, weight][z][y][x], int64: the same integers as splat_kernel (float32 arithmetic in its order)"""
    p = np.asarray(points, np.float32)
    nrm = np.asarray(normals, np.float32)
    scale = np.float32(2.0) ** np.float32(normal_scale_log2(nrm))
    g = ((p[:, :3] / p[:, 3:4]) - origin[None, :]) / h                     # float32 throughout
    f = np.floor(g)
    ijk = f.astype(np.int64)
    ok = np.all((ijk >= 0) & (ijk + 1 < G), axis=1) & np.all(np.abs(nrm) <= np.float32(1e4), axis=1)   # (NaN normals fail the comparison too)
    nrm = (np.where(ok[:, None], nrm, np.float32(0.0)) * scale).astype(np.float32)   # masked BEFORE the integer cast: NaN -> int64 is undefined (and warns); the scale is exact
    t = (g - f).astype(np.float32)
    out = np.zeros((4, G, G, G), np.int64)
    one = np.float32(1.0)
    for c in range(8):
        d = np.array([c & 1, (c >> 1) & 1, c >> 2])
        w = np.ones(len(p), np.float32)
        for a in range(3):
            w = (w * (t[:, a] if d[a] else (one - t[:, a]))).astype(np.float32) if a else (t[:, a] if d[a] else (one - t[:, a])).astype(np.float32)
        q = ijk + d[None, :]
        for ch in range(3):
            val = np.rint((nrm[:, ch] * w).astype(np.float32) * SPLAT_SCALE).astype(np.int64)
            np.add.at(out[ch], (q[ok, 2], q[ok, 1], q[ok, 0]), val[ok])
        np.add.at(out[3], (q[ok, 2], q[ok, 1], q[ok, 0]), np.rint(w * SPLAT_SCALE).astype(np.int64)[ok])
    return out


def poisson_chi(splat, smooth):
    """laplace(chi) = div V in the Fourier domain, float64: chi^ = -i k.V^ / |k|^2 exp(-sigma^2 |k|^2 / 2)"""
    G = splat.shape[1]
    V = splat[:3].astype(np.float64) / 65536.0
    f = np.fft.fftfreq(G, 1.0 / G)
    f[np.abs(f) * 2 == G] = 0.0
    k = 2.0 * np.pi * f / G
    fr = np.arange(G // 2 + 1, dtype=np.float64)
    fr[fr * 2 == G] = 0.0
    kx = (2.0 * np.pi * fr / G)[None, None, :]
    ky = k[None, :, None]
    kz = k[:, None, None]
    k2 = kx * kx + ky * ky + kz * kz
    S = [np.fft.rfftn(V[c]) for c in range(3)]
    dot = kx * S[0] + ky * S[1] + kz * S[2]
    with np.errstate(divide="ignore", invalid="ignore"):
        chi_hat = np.where(k2 > 0, -1j * dot * np.exp(-0.5 * smooth * smooth * k2) / k2, 0.0)
    return np.fft.irfftn(chi_hat, s=(G, G, G), axes=(0, 1, 2))


def trilinear(field, G, origin, h, xyz):
    g = (np.asarray(xyz, np.float64) - origin[None, :].astype(np.float64)) / float(h)
    ijk = np.clip(np.floor(g).astype(np.int64), 0, G - 2)
    t = g - ijk
    acc = np.zeros(len(g))
    for c in range(8):
        d = np.array([c & 1, (c >> 1) & 1, c >> 2])
        w = np.prod(np.where(d[None, :] == 1, t, 1.0 - t), axis=1)
        q = ijk + d[None, :]
        acc += w * field[q[:, 2], q[:, 1], q[:, 0]]
    return acc


def poisson_level(chi, G, origin, h, xyz):
    """the level the surface is extracted at: the lower median of chi (trilinear) over the samples -- CGAL's Poisson_reconstruction_function
    shifts its function to the median value at the input points (compute_implicit_function, cgal_poisson.cpp:72)"""
    vals = np.sort(trilinear(chi, G, origin, h, xyz))
    return vals[(len(vals) - 1) // 2]


def poisson_support(splat, nodes):
    """the samples' support (csrc/poisson.hip step 4a): [z][y][x] bool, nodes within `nodes` nodes (Chebyshev, no wrap) of a node that collected
    sample weight; nodes = ceil(support_spacings x average spacing / h), at most G"""
    import scipy.ndimage as ndi
    seed = (np.asarray(splat[3]) > 0).astype(np.uint8)
    return ndi.maximum_filter(seed, size=2 * int(nodes) + 1, mode="constant", cval=0).astype(bool)


CELL_EDGE_A = [0, 2, 4, 6, 0, 1, 4, 5, 0, 1, 2, 3]   # the 12 edges of a cell as pairs of corner numbers (bit 0: x, bit 1: y, bit 2: z)
CELL_EDGE_B = [1, 3, 5, 7, 2, 3, 6, 7, 4, 5, 6, 7]


def cell_components():
    """csrc/poisson.hip: cell_patch_table.  For each of the 256 inside / outside patterns of a cell's corners (bit c: corner c inside) the
    patches the level set cuts out of the cell: two crossing edges belong to one patch when a face of the cell joins them -- a face with two
    crossing edges joins those two; a face with four (two inside corners on a diagonal) joins the pair AROUND EACH INSIDE CORNER, the same
    decision from either side of the face.  Returns (comp[256][12]: patch number of edge e or -1, numbered by lowest edge; count[256])."""
    comp = np.full((256, 12), -1, np.int8)
    count = np.zeros(256, np.int32)
    for m in range(256):
        ins = [(m >> c) & 1 for c in range(8)]
        cross = [ins[a] != ins[b] for a, b in zip(CELL_EDGE_A, CELL_EDGE_B)]
        parent = list(range(12))

        def find(x):
            while parent[x] != x:
                x = parent[x]
            return x

        def join(x, y):
            x, y = find(x), find(y)
            if x != y:
                parent[max(x, y)] = min(x, y)

        for axis in range(3):
            for side in range(2):
                on = [c for c in range(8) if ((c >> axis) & 1) == side]
                es = [e for e in range(12) if CELL_EDGE_A[e] in on and CELL_EDGE_B[e] in on and cross[e]]
                if len(es) == 2:
                    join(es[0], es[1])
                elif len(es) == 4:
                    for c in on:
                        if ins[c]:
                            mine = [e for e in es if CELL_EDGE_A[e] == c or CELL_EDGE_B[e] == c]
                            join(mine[0], mine[1])
        roots = sorted({find(e) for e in range(12) if cross[e]})
        for e in range(12):
            if cross[e]:
                comp[m, e] = roots.index(find(e))
        count[m] = len(roots)
    return comp, count


def surface_nets(chi, iso, origin, h, support=None):
    """csrc/poisson.hip's meshing rules on a given field: (vertices V x 4 float32, faces F x 3 int32), numbered in its order.  One vertex per
    PATCH of a cell (cell_components: a cell that two sheets of the surface pass through gets two vertices, so the sheets stay apart), at the
    mean of the patch's edge crossings; one quad per crossing grid edge, over the patches of its four cells that contain it.
    support: poisson_support's node mask (a cell is meshed when its low corner node is inside), None: everywhere"""
    chi = np.asarray(chi, np.float32)
    iso = np.float32(iso)
    G = chi.shape[0]
    C = G - 1
    inside = chi < iso                                              # [z][y][x]
    corners = [(c & 1, (c >> 1) & 1, c >> 2) for c in range(8)]
    case = np.zeros((C, C, C), np.int32)
    for c, (dx, dy, dz) in enumerate(corners):
        case |= inside[dz:dz + C, dy:dy + C, dx:dx + C].astype(np.int32) << c
    comp, count = cell_components()
    mixed = (case != 0) & (case != 255)
    if support is not None:
        mixed &= np.asarray(support, bool)[:C, :C, :C]
    per_cell = np.where(mixed, count[case], 0)
    first = np.cumsum(per_cell.ravel()) - per_cell.ravel()           # C order = k, j, i with i fastest: the kernel's cell order
    first = first.reshape(C, C, C)
    nv = int(per_cell.sum())
    kk, jj, ii = np.nonzero(mixed)
    cc = case[kk, jj, ii]
    v = [chi[kk + dz, jj + dy, ii + dx] - iso for (dx, dy, dz) in corners]
    verts = np.ones((nv, 4), np.float32)
    for patch in range(4):
        sel = count[cc] > patch
        if not sel.any():
            break
        s = np.zeros((3, len(kk)), np.float32)
        m = np.zeros(len(kk), np.int32)
        for e, (a, b) in enumerate(zip(CELL_EDGE_A, CELL_EDGE_B)):
            va, vb = v[a], v[b]
            cross = comp[cc, e] == patch
            with np.errstate(divide="ignore", invalid="ignore"):
                t = (va / (va - vb)).astype(np.float32)
            for axis in range(3):
                ca, cb = np.float32(corners[a][axis]), np.float32(corners[b][axis])
                with np.errstate(invalid="ignore"):                  # (t is inf or NaN on edges that do not cross: not selected)
                    s[axis] = np.where(cross, (s[axis] + (ca + t * (cb - ca)).astype(np.float32)).astype(np.float32), s[axis])
            m += cross
        with np.errstate(divide="ignore"):
            inv = (np.float32(1.0) / m.astype(np.float32)).astype(np.float32)
        rows = first[kk, jj, ii][sel] + patch
        for axis, base in enumerate((ii, jj, kk)):
            with np.errstate(invalid="ignore"):                      # (cells without this patch: 0 x inf, not selected)
                verts[rows, axis] = (origin[axis] + np.float32(h) * (base.astype(np.float32) + (s[axis] * inv).astype(np.float32)).astype(np.float32))[sel]
    faces = []
    du, dw = [-1, 0, 0, -1], [-1, -1, 0, 0]
    for axis in range(3):
        # node (i, j, k) -> its +axis neighbour; the kernel numbers edges axis-major, then k, j, i
        a = inside
        b = np.roll(inside, -1, axis=2 - axis)
        state = np.where(a == b, 0, np.where(a, 1, 2))
        k, j, i = np.meshgrid(np.arange(G), np.arange(G), np.arange(G), indexing="ij")
        along = (i, j, k)[axis]
        u = (j, k, i)[axis]
        w = (k, i, j)[axis]
        valid = (along + 1 <= C) & (u >= 1) & (u <= C - 1) & (w >= 1) & (w <= C - 1) & (state != 0)
        ek, ej, ei = np.nonzero(valid)
        st = state[ek, ej, ei]
        quad = np.zeros((len(ek), 4), np.int64)
        have = np.ones(len(ek), bool)
        for c in range(4):
            ci, cj, ck = ei.copy(), ej.copy(), ek.copy()
            if axis == 0:
                cj += du[c]; ck += dw[c]
                local = -du[c] + 2 * -dw[c]                          # the grid edge inside that cell: x edges 0..3 = ly + 2 lz
            elif axis == 1:
                ck += du[c]; ci += dw[c]
                local = 4 + -dw[c] + 2 * -du[c]                      # y edges 4..7 = 4 + lx + 2 lz
            else:
                ci += du[c]; cj += dw[c]
                local = 8 + -du[c] + 2 * -dw[c]                      # z edges 8..11 = 8 + lx + 2 ly
            quad[:, c] = first[ck, cj, ci] + comp[case[ck, cj, ci], local]
            have &= mixed[ck, cj, ci]
        if support is not None:                                      # all four cells around the edge have a vertex
            quad, st = quad[have], st[have]
        else:
            assert have.all()
        flip = st == 2
        q1 = np.where(flip, quad[:, 3], quad[:, 1])
        q3 = np.where(flip, quad[:, 1], quad[:, 3])
        tri = np.stack([quad[:, 0], q1, quad[:, 2], quad[:, 0], quad[:, 2], q3], 1).reshape(-1, 3)
        faces.append(tri)
    return verts, np.concatenate(faces).astype(np.int32)


# ---------------------------------------------------------------------------------------------------------------------------------
# facet criteria (csrc/surface_criteria.cpp): cgal_poisson.cpp:50-52, 95-97 hands make_surface_mesh a lower bound on the facets' angles
# (sm_angle = 20 degrees), an upper bound on their circumradius (sm_radius x average spacing) and on their distance to the surface
# (sm_distance x average spacing).  The restatement below follows the pass operation for operation in float64 Python scalars (the
# C++ is compiled with -ffp-contract=off and uses double throughout: the same IEEE operations in the same order), including the order
# of the incidence lists and of the work list, so that the result can be compared index for index on small meshes.
def simplify_surface(vertices, faces, min_angle_deg, max_distance):
    """csrc/surface_criteria.cpp: mvs_surface_simplify, operation for operation -> (vertices, faces, report dict)"""
    return enforce_facet_criteria(vertices, faces, min_angle_deg, 1e30, max_distance, _simplify=True)


def enforce_facet_criteria(vertices, faces, min_angle_deg, max_radius, max_distance, _simplify=False):
    """-> (vertices V' x 4 float32, faces F' x 3 int32, report dict).  Pure-Python loops: use on meshes of a few ten thousand facets.
    (_simplify: the driver of simplify_surface over the same mesh operations, instead of the criteria rounds)"""
    import math
    vin = np.asarray(vertices, np.float32)
    p = [(float(r[0]), float(r[1]), float(r[2])) for r in vin]
    f = [[int(x) for x in r] for r in np.asarray(faces)]
    nv, nf = len(p), len(f)
    inc = [[] for _ in range(nv)]

    def sub(a, b):
        return (a[0] - b[0], a[1] - b[1], a[2] - b[2])

    def dot(a, b):
        return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]

    def cross(a, b):
        return (a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0])

    def quality(a, b, c):
        ab, bc, ca = sub(b, a), sub(c, b), sub(a, c)
        lab, lbc, lca = dot(ab, ab), dot(bc, bc), dot(ca, ca)
        if not lab > 0.0 or not lbc > 0.0 or not lca > 0.0:
            return -1.0
        cA, cB, cC = -dot(ab, ca) / math.sqrt(lab * lca), -dot(ab, bc) / math.sqrt(lab * lbc), -dot(bc, ca) / math.sqrt(lbc * lca)
        return -max(cA, max(cB, cC))

    def q_of(i):
        return quality(p[f[i][0]], p[f[i][1]], p[f[i][2]])

    def normal_of(a, b, c):
        return cross(sub(p[b], p[a]), sub(p[c], p[a]))

    def third(i, u, v):
        for w in f[i]:
            if w != u and w != v:
                return w
        return -1

    def directed(i, u, v):
        t = f[i]
        return any(t[k] == u and t[(k + 1) % 3] == v for k in range(3))

    def edge_facets(u, v):
        return [i for i in inc[u] if v in f[i]]

    for i in range(nf):
        a, b, c = f[i]
        if a == b or b == c or a == c:
            f[i] = [-1, -1, -1]
            continue
        inc[a].append(i), inc[b].append(i), inc[c].append(i)
    q_bound = -math.cos(float(np.float32(min_angle_deg)) * 3.14159265358979323846 / 180.0)
    guard = [0.25 * float(np.float32(max_distance))]
    last_nearest = [0.0]
    count = {"collapses": 0, "flips": 0}

    def try_collapse(u, v):
        e = edge_facets(u, v)
        if len(e) != 2:
            return None
        c, d = third(e[0], u, v), third(e[1], u, v)
        if c < 0 or d < 0 or c == d:
            return None
        if len(inc[c]) <= 3 or len(inc[d]) <= 3 or len(inc[u]) + len(inc[v]) < 7:
            return None
        nbv = set(w for i in inc[v] for w in f[i])
        shared = set(w for i in inc[u] for w in f[i] if w != u and w != v and w in nbv)
        if len(shared) != 2:
            return None
        before, after, nearest, any_ = 0.0, 0.0, 1e300, False
        for i in inc[u]:
            q0 = q_of(i)
            before = min(before, q0)
            if i == e[0] or i == e[1]:
                continue
            t = [v if w == u else w for w in f[i]]
            after = min(after, quality(p[t[0]], p[t[1]], p[t[2]]))
            n0, n1 = normal_of(*f[i]), normal_of(*t)
            l0, l1 = dot(n0, n0), dot(n1, n1)
            if not l1 > 0.0:
                return None
            if q0 > -0.9962 and dot(n0, n1) < 0.3 * math.sqrt(l0 * l1):
                return None
            nearest = min(nearest, abs(dot(n1, sub(p[u], p[t[0]]))) / math.sqrt(l1))
            any_ = True
        if not any_ or nearest > guard[0]:
            return None
        last_nearest[0] = nearest
        return after, before

    journal = {"on": False, "log": [], "count": None}

    def save_face(i):
        if journal["on"]:
            journal["log"].append(("f", i, list(f[i])))

    def save_list(w):
        if journal["on"]:
            journal["log"].append(("l", w, list(inc[w])))

    def journal_begin():
        journal["on"], journal["log"], journal["count"] = True, [], dict(count)

    def journal_rollback():
        for kind, idx, old in reversed(journal["log"]):
            if kind == "f":
                f[idx] = old
            else:
                inc[idx] = old
        count.update(journal["count"])
        journal["on"] = False

    def do_collapse(u, v, touched):
        e = edge_facets(u, v)[:2]
        for i in e:
            for w in f[i]:
                if w != u:
                    save_list(w)
                    inc[w].remove(i)
            save_face(i)
            f[i] = [-1, -1, -1]
        save_list(v)
        save_list(u)
        for i in inc[u]:
            if i == e[0] or i == e[1]:
                continue
            save_face(i)
            f[i] = [v if w == u else w for w in f[i]]
            inc[v].append(i)
            touched.append(i)
        inc[u] = []
        count["collapses"] += 1

    def flip_pair(u, v):
        e = edge_facets(u, v)
        if len(e) != 2:
            return None
        f1, f2 = e
        if not directed(f1, u, v):
            f1, f2 = f2, f1
        return f1, f2

    def try_flip(u, v):
        pair = flip_pair(u, v)
        if pair is None:
            return None
        f1, f2 = pair
        if not directed(f1, u, v) or not directed(f2, v, u):
            return None
        c, d = third(f1, u, v), third(f2, u, v)
        if c < 0 or d < 0 or c == d or any(d in f[i] for i in inc[c]):
            return None
        if len(inc[u]) <= 3 or len(inc[v]) <= 3:
            return None
        before = min(q_of(f1), q_of(f2))
        after = min(quality(p[u], p[d], p[c]), quality(p[d], p[v], p[c]))
        n1, n2 = normal_of(u, v, c), normal_of(v, u, d)
        ref = (n1[0] + n2[0], n1[1] + n2[1], n1[2] + n2[2])
        m1, m2 = normal_of(u, d, c), normal_of(d, v, c)
        lr, l1, l2 = dot(ref, ref), dot(m1, m1), dot(m2, m2)
        if not lr > 0.0 or not l1 > 0.0 or not l2 > 0.0:
            return None
        if dot(m1, ref) < 0.3 * math.sqrt(l1 * lr) or dot(m2, ref) < 0.3 * math.sqrt(l2 * lr):
            return None
        x = cross(sub(p[v], p[u]), sub(p[d], p[c]))
        lx = dot(x, x)
        if not lx > 0.0 or abs(dot(x, sub(p[c], p[u]))) / math.sqrt(lx) > guard[0]:
            return None
        return after, before

    def do_flip(u, v, touched):
        f1, f2 = flip_pair(u, v)
        c, d = third(f1, u, v), third(f2, u, v)
        save_face(f1), save_face(f2)
        save_list(u), save_list(v), save_list(c), save_list(d)
        f[f1] = [u, d, c]
        f[f2] = [d, v, c]
        inc[v].remove(f1)
        inc[d].append(f1)
        inc[u].remove(f2)
        inc[c].append(f2)
        touched += [f1, f2]
        count["flips"] += 1

    def improve(i, touched):
        best = (0, 0, 0, -2.0)
        for k in range(3):
            u, v = f[i][k], f[i][(k + 1) % 3]
            for kind, a, b, fn in ((1, u, v, try_collapse), (1, v, u, try_collapse), (2, u, v, try_flip)):
                r = fn(a, b)
                if r is not None and r[0] > r[1] + 1e-12 and r[0] > best[3]:
                    best = (kind, a, b, r[0])
        if best[0] == 1:
            do_collapse(best[1], best[2], touched)
        elif best[0] == 2:
            do_flip(best[1], best[2], touched)
        return best[0] != 0

    def alive_bad(i):
        return f[i][0] >= 0 and q_of(i) < q_bound

    def rescue(i, touched):
        """csrc/surface_criteria.cpp Work::rescue: every valid operation on the facet's edges, best result first, applied on trial; the facets
        it leaves below the bound go to improve() in turn (at most 6 follow-up operations); kept only if nothing it touched ends below the
        bound, rolled back otherwise"""
        cands = []
        for k in range(3):
            u, v = f[i][k], f[i][(k + 1) % 3]
            for kind, a, b, fn in ((1, u, v, try_collapse), (1, v, u, try_collapse), (2, u, v, try_flip)):
                r = fn(a, b)
                if r is not None:
                    cands.append((kind, a, b, r[0]))
        cands.sort(key=lambda c: -c[3])   # stable, like std::stable_sort
        for kind, a, b, _ in cands:
            journal_begin()
            trial = []
            if kind == 1:
                do_collapse(a, b, trial)
            else:
                do_flip(a, b, trial)
            work = [t for t in trial if alive_bad(t)]
            ok, steps, head = True, 0, 0
            while head < len(work):
                j = work[head]
                head += 1
                if not alive_bad(j):
                    continue
                if steps >= 6:
                    ok = False
                    break
                steps += 1
                t2 = []
                if not improve(j, t2):
                    ok = False
                    break
                for t in t2:
                    trial.append(t)
                    if alive_bad(t):
                        work.append(t)
            if ok and any(alive_bad(t) for t in trial):
                ok = False
            if ok and alive_bad(i):
                ok = False
            if ok:
                journal["on"] = False
                touched += trial
                return True
            journal_rollback()
        return False

    if _simplify:
        md = float(np.float32(max_distance))
        err = [0.0] * nv
        vn = [(0.0, 0.0, 0.0)] * nv
        for i in range(nf):
            if f[i][0] < 0:
                continue
            n = normal_of(*f[i])
            for w in f[i]:
                a = vn[w]
                vn[w] = (a[0] + n[0], a[1] + n[1], a[2] + n[2])
        for i in range(nv):
            l = math.sqrt(dot(vn[i], vn[i]))
            vn[i] = (vn[i][0] / l, vn[i][1] / l, vn[i][2] / l) if l > 0.0 else (0.0, 0.0, 0.0)

        def deviation(u, v):
            e = edge_facets(u, v)[:2]
            worst = 0.0
            for i in inc[u]:
                if i == e[0] or i == e[1]:
                    continue
                t = [v if w == u else w for w in f[i]]
                n = normal_of(*t)
                ln = math.sqrt(dot(n, n))
                if not ln > 0.0:
                    return 1e300
                c, emax, l2 = 1.0, 0.0, 0.0
                for k in range(3):
                    c = min(c, dot(n, vn[t[k]]) / ln)
                    emax = max(emax, max(err[v], err[u]) if t[k] == v else err[t[k]])
                    d = sub(p[t[k]], p[t[(k + 1) % 3]])
                    l2 = max(l2, dot(d, d))
                if not c > 0.2:
                    return 1e300
                worst = max(worst, 0.5 * math.sqrt(l2) * math.sqrt(max(0.0, 1.0 - c * c)) / (1.0 + c) + emax)
            return worst

        locked = [False] * nv
        for i in range(nf):
            if f[i][0] < 0:
                continue
            for k in range(3):
                u, v = f[i][k], f[i][(k + 1) % 3]
                if len(edge_facets(u, v)) != 2:
                    locked[u] = locked[v] = True
        alive_v = sum(1 for i in range(nv) if inc[i])
        for _pass in range(16):
            edges = []
            for i in range(nf):
                if f[i][0] < 0:
                    continue
                for k in range(3):
                    u, v = f[i][k], f[i][(k + 1) % 3]
                    if u < v:
                        d = sub(p[u], p[v])
                        edges.append((dot(d, d), u, v))
            edges.sort()
            removed = 0
            for _, eu, ev in edges:
                if not inc[eu] or not inc[ev] or not any(ev in f[i] for i in inc[eu]):
                    continue
                best = None
                for u, v in ((eu, ev), (ev, eu)):
                    if locked[u]:
                        continue
                    guard[0] = md - err[u]
                    if not guard[0] > 0.0:
                        continue
                    r = try_collapse(u, v)
                    if r is not None and r[0] >= q_bound and (best is None or r[0] > best[2]) and deviation(u, v) <= md:
                        best = (u, v, r[0], last_nearest[0])
                if best is None:
                    continue
                do_collapse(best[0], best[1], [])
                err[best[1]] = max(err[best[1]], err[best[0]] + best[3])
                removed += 1
            alive_v -= removed
            if removed * 100 < alive_v:
                break
    queue = [] if _simplify else [i for i in range(nf) if f[i][0] >= 0 and q_of(i) < q_bound]
    budget, done = 2 * nf + 1000, 0
    for rnd in range(3):
        head = 0
        while head < len(queue) and done < budget:
            i = queue[head]
            head += 1
            if f[i][0] < 0 or q_of(i) >= q_bound:
                continue
            touched = []
            if improve(i, touched):
                done += 1
                queue += [t for t in touched if f[t][0] >= 0 and q_of(t) < q_bound]
        left = sorted(set(i for i in queue if f[i][0] >= 0 and q_of(i) < q_bound))
        if not left:
            break
        queue = left
        guard[0] = (0.5 if rnd == 0 else 1.0) * float(np.float32(max_distance))
    left = sorted(set(i for i in queue if alive_bad(i)))
    guard[0] = float(np.float32(max_distance))
    for i in left[:4096]:
        if alive_bad(i):
            rescue(i, [])
    # border trim (csrc/surface_criteria.cpp): facets still below the bound on the outline of an open surface go when that only moves the outline
    trimmed = 0
    if not _simplify:
        bad = [i for i in range(nf) if alive_bad(i)]

        def border_edge(a, b):
            return len(edge_facets(a, b)) == 1

        def border_vertex(v):
            return any(x != v and border_edge(v, x) for i in inc[v] for x in f[i])

        changed = True
        while changed:
            changed = False
            for i in bad:
                if f[i][0] < 0:
                    continue
                a, b, c = f[i]
                on = (len(edge_facets(a, b)), len(edge_facets(b, c)), len(edge_facets(c, a)))
                eb = tuple(n == 1 for n in on)
                nb = sum(eb)
                if nb == 0 or max(on) > 2 or (nb == 1 and border_vertex(c if eb[0] else (a if eb[1] else b))):
                    continue
                for w in (a, b, c):
                    inc[w].remove(i)
                f[i] = [-1, -1, -1]
                trimmed += 1
                changed = True
    used = np.zeros(nv, bool)
    alive = [t for t in f if t[0] >= 0]
    for t in alive:
        used[t] = True
    renum = np.cumsum(used) - 1
    fout = np.array([[renum[w] for w in t] for t in alive], np.int32).reshape(-1, 3)
    report = dict(count)
    if not _simplify:
        report["facets_trimmed"] = trimmed
    min_q, max_r2, below, above = 0.0, 0.0, 0, 0
    for t in alive:
        a, b, c = p[t[0]], p[t[1]], p[t[2]]
        q = quality(a, b, c)
        below += q < q_bound
        min_q = min(min_q, q)
        n = cross(sub(b, a), sub(c, a))
        area2 = dot(n, n)
        l = dot(sub(b, a), sub(b, a)) * dot(sub(c, b), sub(c, b)) * dot(sub(a, c), sub(a, c))
        r2 = l / (4.0 * area2) if area2 > 0.0 else 1e300
        above += r2 > float(np.float32(max_radius)) * float(np.float32(max_radius))
        if r2 < 1e299:
            max_r2 = max(max_r2, r2)
    report.update(facets_below_angle=int(below), facets_above_radius=int(above),
                  min_angle_deg=180.0 if not alive else math.degrees(math.acos(min(1.0, max(-1.0, -min_q)))), max_circumradius=math.sqrt(max_r2))
    return vin[used], fout, report
