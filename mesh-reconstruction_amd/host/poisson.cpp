// poisson.cpp -- poissonSurface (recon.hpp:37; cgal_poisson.cpp:47-136, pcl.cpp:225-228 of the reference) above the C ABI
// (include/mvs.h: mvs_poisson_surface, csrc/poisson.hip).  Same call, same Mesh layout (vertices N x 4 homogeneous f32, faces F x 3
// i32, normals out of the solid); what is inside is this library's grid Poisson solver, not CGAL's (see csrc/poisson.hip), followed by
// the reference's facet criteria as a pass over its triangles (csrc/surface_criteria.cpp).
#include <cmath>
#include <cstdlib>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/mvs.h"
#include "recon.hpp"

static PoissonNormals g_poisson_normals = POISSON_UNIT_NORMALS;  // what the reference's two-argument call does here (see below)

static bool g_poisson_simplify = true;  // the facet count the criteria ask for, not the grid's (setPoissonSimplify(false): the criteria pass's mesh as it is)
void setPoissonSimplify(bool on) { g_poisson_simplify = on; }
void setPoissonNormals(PoissonNormals mode) { g_poisson_normals = mode; }
PoissonNormals poissonNormals() { return g_poisson_normals; }

Mesh poissonSurface(const Mat points, const Mat normals) { return poissonSurface(points, normals, g_poisson_normals); }

Mesh poissonSurface(const Mat points, const Mat normals, PoissonNormals mode)
{
    if (points.rows == 0) return Mesh(Mat(0, 4, mvs::F32C1), Mat(0, 3, mvs::S32C1));
    if (points.cols != 4 || normals.cols != 3 || normals.rows != points.rows)
        throw std::runtime_error("poissonSurface: points must be N x 4 (homogeneous) and normals N x 3");  // cgal_poisson.cpp:58-59 reads p[0..3], n[0..2]
    // A DELIBERATE DIVERGENCE from the reference.  The normals arrive scaled by triangulatePixels' pdf
    // (util.cpp:322-327: lengths spanning two decades on real frames), and BOTH of the reference's backends use those lengths: the default
    // one (Makefile:4, POISSON_LIBRARY = cgal) hands them to Poisson_reconstruction_function as they are (cgal_poisson.cpp:58-69), and
    // pcl.cpp defines USE_PRECISION (pcl.cpp:23), so pcl::Poisson runs with setConfidence(true) (pcl.cpp:198-202) -- under a comment of the
    // reference's author: "may be better to disable, sometimes does more harm than use".  On this library's grid solver it does more harm:
    // on a cloud this pipeline produces (tools/criteria_on_c5.py) the level set under pdf-length confidences wanders off the weak samples
    // (591 k vertices, half of them more than two spacings from any sample); with unit normals it is the sheet (171 k vertices, median
    // distance half a spacing).  So the two-argument call of recon.hpp:37 normalises the normals first (POISSON_UNIT_NORMALS);
    // setPoissonNormals(POISSON_CONFIDENCE_NORMALS), or the three-argument call, gives the reference's semantics.  No environment
    // variable is read.
    std::vector<float> unit;
    const float *nrm = normals.ptr<float>();
    if (mode == POISSON_UNIT_NORMALS) {
        unit.assign(nrm, nrm + 3 * (size_t)normals.rows);
        for (int i = 0; i < normals.rows; i++) {
            float *n = unit.data() + 3 * (size_t)i;
            const double len = std::sqrt((double)n[0] * n[0] + (double)n[1] * n[1] + (double)n[2] * n[2]);
            if (len > 0.0 && len < 1e30)
                for (int c = 0; c < 3; c++) n[c] = (float)((double)n[c] / len);
        }
        nrm = unit.data();
    }
    mvs_surface *s = nullptr;
    const int rc = mvs_poisson_surface(points.ptr<float>(), nrm, points.rows, 0, 1.0f, 0, &s);
    if (rc != MVS_OK) throw std::runtime_error(std::string("poissonSurface: ") + mvs_surface_last_error());  // cgal_poisson.cpp:73: assert(success)
    // cgal_poisson.cpp:50-52, 95-97: the facet criteria handed to make_surface_mesh, in units of the samples' average spacing
    const float sm_angle = 20.0f, sm_radius = 300.0f, sm_distance = 0.375f;
    float spacing = 0.0f;
    mvs_surface_spacing(s, &spacing, nullptr, nullptr);
    if (spacing > 0.0f && mvs_surface_enforce_criteria(s, sm_angle, sm_radius * spacing, sm_distance * spacing, nullptr) != MVS_OK) {
        mvs_surface_free(s);
        throw std::runtime_error("poissonSurface: the facet criteria pass failed");
    }
    // ... and what those criteria do NOT ask for is the grid's facet density: vertices the criteria do not need are removed (edge collapses
    // under the same angle bound and an accumulated distance of at most sm_distance; csrc/surface_criteria.cpp: mvs_surface_simplify)
    if (g_poisson_simplify && spacing > 0.0f && mvs_surface_simplify(s, sm_angle, sm_distance * spacing, nullptr) != MVS_OK) {
        mvs_surface_free(s);
        throw std::runtime_error("poissonSurface: the simplification pass failed");
    }
    int nv = 0, nf = 0;
    mvs_surface_counts(s, &nv, &nf);
    Mesh result(Mat(nv, 4, mvs::F32C1), Mat(nf, 3, mvs::S32C1));
    mvs_surface_fetch(s, nv ? result.vertices.ptr<float>() : nullptr, nf ? result.faces.ptr<int32_t>() : nullptr);
    mvs_surface_free(s);
    return result;
}
