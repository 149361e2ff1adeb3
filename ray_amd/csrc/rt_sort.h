// rt_sort.h -- sort key for secondary rays (K6).
//
// The reference sorts secondary rays before every bounce (RendererCPU.h:506 SortRays_CPU; GPU: sort_hash_rays +
// 8 radix passes + reorder, RendererVK.cpp:641-652) with a 32-bit key = direction bucket (8 b) | 24-bit Morton code
// of the origin cell (CoreRef.cpp:594-612).  Ray order never changes results (each pixel owns at most one ray per
// bounce), so the key is free to design.  Here: position first, then direction:
//
//     [ coarse origin Morton, 5 bit/axis = 15 b ][ direction octant 3 b ][ fine origin Morton, 3 bit/axis = 9 b ]
//
// Rays that start in the same ~1/32 of the scene extent and head into the same octant become neighbours, i.e. lanes
// of one wavefront walk the same BVH subtrees (fewer divergent node fetches, better L2 hit rate) and, later, shade
// the same materials.  27 significant bits -> 4 radix passes of 7 bits.
#pragma once

#include "rt_base.h"

namespace rt {

constexpr uint32_t SORT_KEY_BITS = 27;
constexpr uint32_t SORT_KEY_DEAD = 0xffffffffu; // slots beyond the live ray count sort to the end

// spread the low 8 bits of v so that there are two zero bits between each: abcdefgh -> a00b00c00d00e00f00g00h
RT_HD uint32_t part1by2_8(uint32_t v) {
    v &= 0xffu;
    v = (v | (v << 8)) & 0x0000f00fu;
    v = (v | (v << 4)) & 0x000c30c3u;
    v = (v | (v << 2)) & 0x00249249u;
    return v;
}

struct SortGrid {
    float root_min[3];
    float inv_cell[3]; // 256 / extent
};

// `mode` (tuning, RAYHIP_SORT_KEY): 0 = the key above; 1 = direction octant only (3 bits: what binning at emission could do
// without a sort pass); 2 = coarse cell, then octant (18 bits); 3 = octant, then coarse cell (18 bits)
RT_HD uint32_t ray_sort_key_bits(const int mode) { return mode == 1 ? 3u : (mode == 2 || mode == 3) ? 18u : SORT_KEY_BITS; }
RT_HD uint32_t ray_sort_key(const SortGrid &g, const f3 o, const f3 d, const int mode = 0) {
    const int x = clampi(int((o.x - g.root_min[0]) * g.inv_cell[0]), 0, 255);
    const int y = clampi(int((o.y - g.root_min[1]) * g.inv_cell[1]), 0, 255);
    const int z = clampi(int((o.z - g.root_min[2]) * g.inv_cell[2]), 0, 255);
    const uint32_t m = part1by2_8(uint32_t(x)) | (part1by2_8(uint32_t(y)) << 1) | (part1by2_8(uint32_t(z)) << 2); // 24 b
    const uint32_t oct = (d.x < 0.0f ? 1u : 0u) | (d.y < 0.0f ? 2u : 0u) | (d.z < 0.0f ? 4u : 0u);
    const uint32_t coarse = m >> 9, fine = m & 0x1ffu;
    if (mode == 1) {
        return oct;
    }
    if (mode == 2) {
        return (coarse << 3) | oct;
    }
    if (mode == 3) {
        return (oct << 15) | coarse;
    }
    return (coarse << 12) | (oct << 9) | fine;
}

} // namespace rt
