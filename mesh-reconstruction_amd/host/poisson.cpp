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

Mesh poissonSurface(const Mat points, const Mat normals)
{
    if (points.rows == 0) return Mesh(Mat(0, 4, mvs::F32C1), Mat(0, 3, mvs::S32C1));
    if (points.cols != 4 || normals.cols != 3 || normals.rows != points.rows)
        throw std::runtime_error("poissonSurface: points must be N x 4 (homogeneous) and normals N x 3");  // cgal_poisson.cpp:58-59 reads p[0..3], n[0..2]
    // The normals arrive scaled by triangulatePixels' pdf (util.cpp:322-327: lengths spanning two decades on real frames).  The reference's
    // PCL backend normalises them unless it is built with USE_PRECISION ("may be better to disable, sometimes does more harm than use",
    // pcl.cpp:198-202: pcl::Poisson's confidence flag off = unit normals); what CGAL does with the lengths cannot be looked up here.  Measured
    // on a cloud this pipeline produces (tools/criteria_on_c5.py): with the pdf lengths as confidences the level set wanders off the
    // weak samples (591 k vertices, half of them more than two spacings from any sample), with unit normals it is the sheet (171 k, median
    // distance half a spacing).  So: unit normals, like pcl.cpp's default; MVS_POISSON_USE_PRECISION=1 keeps the lengths.
    std::vector<float> unit;
    const float *nrm = normals.ptr<float>();
    const char *keep = std::getenv("MVS_POISSON_USE_PRECISION");
    if (!(keep && keep[0] == '1')) {
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
    int nv = 0, nf = 0;
    mvs_surface_counts(s, &nv, &nf);
    Mesh result(Mat(nv, 4, mvs::F32C1), Mat(nf, 3, mvs::S32C1));
    mvs_surface_fetch(s, nv ? result.vertices.ptr<float>() : nullptr, nf ? result.faces.ptr<int32_t>() : nullptr);
    mvs_surface_free(s);
    return result;
}
