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

}  // namespace mvhmr
