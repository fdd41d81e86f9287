// surface_internal.hpp -- the object behind mvs_surface (include/mvs.h), shared by csrc/poisson.hip (which fills it) and
// csrc/surface_criteria.cpp (which improves its facets in place).
#pragma once
#include <cstdint>
#include <vector>

struct SurfaceGrid {
    int G;             // nodes per axis
    float ox, oy, oz;  // position of node (0, 0, 0)
    float h;           // node spacing
};

struct mvs_surface {
    SurfaceGrid grid{};
    float iso = 0.0f;
    float spacing = 0.0f;     // CGAL::compute_average_spacing(points, 6) of the samples
    int ratio_kept = 1;       // node spacing <= 0.75 x average spacing (0: the finest grid, 512^3, is coarser than that)
    int normal_scale_log2 = 0;  // the normals were multiplied by 2^this before the fixed-point splat (chi and the level scale with it)
    int support_cells = 0;    // cells are meshed within this many nodes of a node that collected sample weight (0: everywhere)
    std::vector<float> vertices;   // 4 per vertex
    std::vector<int32_t> faces;    // 3 per face
    std::vector<float> chi;        // G^3, kept when asked for (tests)
    std::vector<int64_t> splat;    // 4 G^3 (vx, vy, vz, weight), kept when asked for (tests)
};
