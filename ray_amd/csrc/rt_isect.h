// rt_isect.h -- ray/box and ray/triangle primitives of the traversal (shared by the BVH2 walk of rt_traverse.h and the
// 4-wide walk of rt_bvh4.h).  IntersectTri: reference internal/CoreRef.cpp:24-50; bbox_test: CoreRef.cpp:171-210.
#pragma once

#include "rt_rng.h"
#include "rt_texture.h"
#include "rt_types.h"

namespace rt {

// plain array stack (host simulation)
struct ArrayStack {
    uint32_t data[4 * MAX_STACK_SIZE]; // (the 8-wide walk of rt_bvh8.h keeps 8-byte entries, one per level with pending siblings)
    uint32_t size = 0;
    RT_HD void push(uint32_t v) { data[size++] = v; }
    RT_HD uint32_t pop() { return data[--size]; }
    // slot access used by the 4-wide walk (rt_bvh4.h), which keeps the top of the stack in a register
    RT_HD bool fast_range(uint32_t) const { return true; }
    RT_HD void write_at(uint32_t idx, uint32_t v) { data[idx] = v; }
    RT_HD void write3_fast(uint32_t idx, uint32_t a, uint32_t b, uint32_t c) { data[idx] = a, data[idx + 1] = b, data[idx + 2] = c; }
    RT_HD uint32_t read_at(uint32_t idx) const { return data[idx]; }
    // two-word entries of the 8-wide walk (rt_bvh8.h), `idx` in words
    RT_HD void write2_at(uint32_t idx, uint32_t a, uint32_t b) { data[idx] = a, data[idx + 1] = b; }
    RT_HD void read2_at(uint32_t idx, uint32_t &a, uint32_t &b) const { a = data[idx], b = data[idx + 1]; }
};

#define RT_SIGN_OF(f) (((f) >= 0) ? 1 : -1)

// One triangle fetched as three 16-byte loads issued together (a single memory round trip per triangle).  The table is the
// reference's 48-byte tri_accel_t array, or -- on the device -- the same records padded to a pitch of 64 bytes, so that
// a record never straddles two 64-byte sectors (half of the 48-byte records do; the walks are bound by the number of
// sectors they miss on).
struct TriData {
    float4 n, u, v;
};
struct TriTable {
    const float4 *rows;
    uint32_t pitch; // rows per record: 3 or 4
};
RT_HD TriData load_tri(const TriTable tris, const uint32_t i) {
    const float4 *p = tris.rows + size_t(i) * tris.pitch;
    TriData t;
    t.n = p[0], t.u = p[1], t.v = p[2];
    return t;
}

// CoreRef.cpp:24-50.  Same operations in the same order as the reference; the three early `return`s are folded
// into one predicate so that no load sits behind a branch (on the GPU every early-out used to cost a
// dependent memory round trip: n_plane -> branch -> u_plane -> branch -> v_plane).  The speculated arithmetic is
// discarded when the predicate fails, so accepted hits are bit-identical.
RT_HD void intersect_tri(const f3 ro, const f3 rd, const TriData &tri, const uint32_t prim_index, Hit &inter) {
    const float det = rd.x * tri.n.x + rd.y * tri.n.y + rd.z * tri.n.z;
    const float dett = tri.n.w - (ro.x * tri.n.x + ro.y * tri.n.y + ro.z * tri.n.z);
    bool ok = !(det == 0.0f || RT_SIGN_OF(dett) != RT_SIGN_OF(det * inter.t - dett));

    const float p0 = det * ro.x + dett * rd.x, p1 = det * ro.y + dett * rd.y, p2 = det * ro.z + dett * rd.z;

    const float detu = (p0 * tri.u.x + p1 * tri.u.y + p2 * tri.u.z) + det * tri.u.w;
    ok = ok && !(RT_SIGN_OF(detu) != RT_SIGN_OF(det - detu));

    const float detv = (p0 * tri.v.x + p1 * tri.v.y + p2 * tri.v.z) + det * tri.v.w;
    ok = ok && !(RT_SIGN_OF(detv) != RT_SIGN_OF(det - detu - detv));

    if (ok) { // rare: a whole wavefront usually skips the (correctly rounded, ~10 instruction) division
        const float rdet = (1.0f / det);
        inter.prim_index = (det < 0.0f) ? int(prim_index) : -int(prim_index) - 1;
        inter.t = dett * rdet;
        inter.u = detu * rdet;
        inter.v = detv * rdet;
    }
}
#undef RT_SIGN_OF

// CoreRef.cpp:171-210.  The reference orders (lo, hi) per axis with compare-and-swap and folds the three axes with
// compare-and-select; written here with fminf/fmaxf, which return the same values for the finite inputs of this test
// (they could only differ in the sign of a zero, and every consumer is a comparison) and compile to single
// v_min_f32 / v_max_f32 / v_max3_f32 instructions instead of v_cmp + v_cndmask pairs.
RT_HD bool bbox_test(const f3 o, const f3 inv_d, const float t, const float mn[3], const float mx[3], float &out_dist) {
    const float lo_x = inv_d.x * (mn[0] - o.x), hi_x = inv_d.x * (mx[0] - o.x);
    const float lo_y = inv_d.y * (mn[1] - o.y), hi_y = inv_d.y * (mx[1] - o.y);
    const float lo_z = inv_d.z * (mn[2] - o.z), hi_z = inv_d.z * (mx[2] - o.z);

    const float tmin = fmaxf(fmaxf(fminf(lo_x, hi_x), fminf(lo_y, hi_y)), fminf(lo_z, hi_z));
    float tmax = fminf(fminf(fmaxf(lo_x, hi_x), fmaxf(lo_y, hi_y)), fmaxf(lo_z, hi_z));
    tmax *= 1.00000024f;

    out_dist = tmin;
    return tmin <= tmax && tmin <= t && tmax > 0;
}

} // namespace rt
