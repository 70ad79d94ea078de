// Geometry gate: which of the two kernel variants runs is decided ON THE DEVICE from the actual cameras and volume.
// AUTO launches both variants' kernels; k_brick_gate has counted the bricks whose pooled tap windows overflow LDS, and
// every gated kernel compares that count with its limit and returns at once when the other variant was selected.  No host
// synchronisation, graph-capturable; only speed depends on it, never results beyond the variants' common tolerance.
#pragma once

namespace mvhmr {

struct Gate {
    const int *count;   // device counter written by k_brick_gate; null = not gated
    int limit;          // brick variant runs while count <= limit
    int wants_brick;    // this kernel belongs to the brick variant (1) / the gather variant (0)
};

// Where voxel centres come from: the caller's (B,X,Y,Z,3) tensor, or -- when `ptr` is null -- the reference's cuboid recipe
// (aggregation.py:138-187) evaluated in the kernel: rot[b] @ (pos + step * (i,j,k) - center[b]) + center[b], 13 floats per
// sample instead of a tensor read.  Same rounding order as mvhmr_build_coord_volumes, so both routes give bit-equal centres.
struct Coords {
    const float *ptr;        // (B,X,Y,Z,3) fp32, or null
    const float *rot;        // (B,9) row-major fp32 (device)
    const float *center;     // (B,3) fp32 (device)
    float px, py, pz;        // cuboid corner
    float sx, sy, sz;        // step per index = sides / (S - 1)
    int Y, Z;                // volume extents needed to split a flat voxel index
};

}  // namespace mvhmr
