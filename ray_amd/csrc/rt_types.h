// rt_types.h -- scene view, wavefront ray state (register form + SoA memory form) shared by all kernels.
#pragma once

#include "../../include/rayhip.h"
#include "rt_base.h"

#if !defined(__HIPCC__)
// host-side stand-ins so the same accessors compile under g++ (tests/hostsim)
struct float4 {
    float x, y, z, w;
};
struct uint2 {
    uint32_t x, y;
};
#endif

namespace rt {

struct Bvh4Node; // rt_bvh4.h
struct Bvh8Node; // rt_bvh8.h

// what the physical sky reads (rt_sky.h): rayhip_sky + the arrays it sizes; desc == null: the environment is not the physical sky
struct SkyView {
    const rayhip_sky *desc;
    const float *transmittance_lut, *multiscatter_lut;
    const uint32_t *dir_lights;
    uint32_t dir_lights_count;
    const uint8_t *weather, *noise3d, *curl, *moon, *cirrus;
};

// Device-side view of the flat scene (pointers into HBM).  Mirrors reference Core.h:511-535 scene_data_t.
struct SceneView {
    const rayhip_bvh2_node *nodes;
    const Bvh4Node *nodes4;     // 4-wide quantised BLAS trees built from `nodes` at upload (rt_bvh4.h), or null
    const Bvh8Node *nodes8;     // 8-wide quantised BLAS trees (rt_bvh8.h), or null; at most one of the two wide forms is set
    const uint32_t *blas_root4; // per mesh instance: root of its tree in the wide form that is set (nodes4 or nodes8)
    const rayhip_tri_accel *tris;
    uint32_t all_solid; // 1: every triangle side of the scene carries MATERIAL_SOLID_BIT -- the transparency round of IntersectScene has nothing to ask the materials
    uint32_t tri_pitch; // 16-byte rows per triangle record as the walks fetch it: 3 = the reference's 48-byte array, 4 = padded to one 64-byte sector each (rt_isect.h: load_tri)
    const uint32_t *tri_indices;
    const rayhip_tri_mat_data *tri_materials;
    const rayhip_material *materials;
    const rayhip_vertex *vertices;
    const uint32_t *vtx_indices;
    const rayhip_mesh_instance *mesh_instances;
    const rayhip_light *lights;
    const uint32_t *li_indices;
    const rayhip_light_cwbvh_node *light_cwnodes;
    const float4 *light_children; // LIGHT_CHILDREN_STRIDE float4 per light-tree node (shade_lights.h: fill_light_children)
    const float4 *tri_verts;      // 8 float4 per triangle: corners (position, normal, uv) + plane + uv area (shade_point.h: fill_tri_verts)
    const float4 *tri_bitangents; // 4 float4 per triangle: the corners' bitangents (normal-mapped materials only)
    const float4 *light_tri_geom; // 4 float4 per light: world-space corners + uvs of a TRI light (shade_lights.h: fill_light_tri_geom)
    const rayhip_texture *textures;
    const uint32_t *texels;
    const float4 *env_qtree;       // env-map importance quadtree: quads of all lods, lod 0 first (rayhip.h)
    uint32_t env_qtree_offset[16]; // first quad of each lod
    const uint32_t *pmj; // 32 dims x 4096 samples x 2 (u32), reference Core.h:363-368
    uint32_t tex_table[8];
    uint32_t tex_flags; // rayhip_scene_desc::texture_flags
    uint32_t li_indices_count;
    uint32_t light_cwnodes_count;
    uint32_t visible_lights_count;
    uint32_t blocker_lights_count;
    uint32_t tlas_root;
    rayhip_environment env;
    SkyView sky;
};

// ---- light_t bitfield accessors (Core.h:197-205; GCC packs bitfields LSB first) --------------------
RT_HD uint32_t light_type(const rayhip_light &l) { return l.flags & 7u; }
RT_HD bool light_doublesided(const rayhip_light &l) { return (l.flags >> 3) & 1u; }
RT_HD bool light_cast_shadow(const rayhip_light &l) { return (l.flags >> 4) & 1u; }
RT_HD bool light_visible(const rayhip_light &l) { return (l.flags >> 5) & 1u; }
RT_HD bool light_sky_portal(const rayhip_light &l) { return (l.flags >> 6) & 1u; }
RT_HD uint32_t light_ray_visibility(const rayhip_light &l) { return (l.flags >> 7) & 0xffu; }
// params[] aliases (Core.h:207-236):
//  sph : pos[0..2] area[3] dir[4..6] radius[7] spot[8] blend[9]
//  rect/disk: pos[0..2] area[3] u[4..6] - v[8..10] -
//  line: pos[0..2] area[3] u[4..6] radius[7] v[8..10] height[11]
//  tri : tri_index mi_index tex_index (as uint32)
//  dir : dir[0..2] cos_angle[3] tan_angle[4] angle[5]

// ---- register forms -------------------------------------------------------------------------------
// reference CoreRef.h:57-71 ray_data_t
struct Ray {
    f3 o, d;
    float pdf;
    f3 c;
    float ior[4];
    float cone_width, cone_spread;
    uint32_t xy;
    uint32_t depth;
};
// reference CoreRef.h:74-85 shadow_ray_t
struct ShadowRay {
    f3 o;
    uint32_t depth;
    f3 d;
    float dist;
    f3 c;
    uint32_t xy;
};
// reference CoreRef.h:88-105 hit_data_t
struct Hit {
    int obj_index;
    int prim_index;
    float t, u, v;
};
RT_HD Hit make_hit() { return Hit{-1, -1, MAX_DIST, 0.0f, -1.0f}; }

// ---- SoA memory forms -----------------------------------------------------------------------------
// One float4 plane per group of fields the same kernel touches together, so a wave reads/writes 1 KiB per
// instruction (16 B/lane, fully coalesced) and the traversal kernel only pulls the planes it needs
// (o/d: 32 B per ray instead of the 72-B AoS record).
struct RaySoA {
    float4 *o_pdf;        // o.xyz, pdf
    float4 *d_cw;         // d.xyz, cone_width
    float4 *c_cs;         // c.rgb, cone_spread
    float4 *ior;          // ior[4]
    uint2 *xy_depth;      // xy, depth
};
struct ShadowSoA {
    float4 *o_depth; // o.xyz, depth(bits)
    float4 *d_dist;  // d.xyz, dist
    float4 *c_xy;    // c.rgb, xy(bits)
};
struct HitSoA {
    float4 *oi_pi_t_u; // obj_index(bits), prim_index(bits), t, u
    float *v;
};


RT_HD float4 mkfloat4(float x, float y, float z, float w) {
    float4 r;
    r.x = x, r.y = y, r.z = z, r.w = w;
    return r;
}

RT_HD void load_ray_od(const RaySoA &s, uint32_t i, Ray &r) {
    const float4 a = s.o_pdf[i], b = s.d_cw[i];
    r.o = {a.x, a.y, a.z};
    r.pdf = a.w;
    r.d = {b.x, b.y, b.z};
    r.cone_width = b.w;
}
// with_ior = false: the plane is not read and the ray gets the stack the camera gives every ray (see store_ray)
RT_HD Ray load_ray(const RaySoA &s, uint32_t i, const bool with_ior = true) {
    Ray r;
    load_ray_od(s, i, r);
    const float4 c = s.c_cs[i], io = with_ior ? s.ior[i] : mkfloat4(-1.0f, -1.0f, -1.0f, -1.0f);
    const uint2 xd = s.xy_depth[i];
    r.c = {c.x, c.y, c.z};
    r.cone_spread = c.w;
    r.ior[0] = io.x, r.ior[1] = io.y, r.ior[2] = io.z, r.ior[3] = io.w;
    r.xy = xd.x, r.depth = xd.y;
    return r;
}
// with_ior = false: the stack of refractive indices is not written (a pass over a scene without refractive surfaces: every ray carries the
// stack the camera gave it, nobody reads the plane -- ShadeParams::plain_ior)
RT_HD void store_ray(const RaySoA &s, uint32_t i, const Ray &r, const bool with_ior = true) {
    s.o_pdf[i] = mkfloat4(r.o.x, r.o.y, r.o.z, r.pdf);
    s.d_cw[i] = mkfloat4(r.d.x, r.d.y, r.d.z, r.cone_width);
    s.c_cs[i] = mkfloat4(r.c.x, r.c.y, r.c.z, r.cone_spread);
    if (with_ior) {
        s.ior[i] = mkfloat4(r.ior[0], r.ior[1], r.ior[2], r.ior[3]);
    }
    uint2 xd;
    xd.x = r.xy, xd.y = r.depth;
    s.xy_depth[i] = xd;
}
RT_HD ShadowRay load_shadow(const ShadowSoA &s, uint32_t i) {
    const float4 a = s.o_depth[i], b = s.d_dist[i], c = s.c_xy[i];
    ShadowRay r;
    r.o = {a.x, a.y, a.z};
    r.depth = float_as_uint(a.w);
    r.d = {b.x, b.y, b.z};
    r.dist = b.w;
    r.c = {c.x, c.y, c.z};
    r.xy = float_as_uint(c.w);
    return r;
}
RT_HD void store_shadow(const ShadowSoA &s, uint32_t i, const ShadowRay &r) {
    s.o_depth[i] = mkfloat4(r.o.x, r.o.y, r.o.z, uint_as_float(r.depth));
    s.d_dist[i] = mkfloat4(r.d.x, r.d.y, r.d.z, r.dist);
    s.c_xy[i] = mkfloat4(r.c.x, r.c.y, r.c.z, uint_as_float(r.xy));
}
RT_HD Hit load_hit(const HitSoA &s, uint32_t i) {
    const float4 a = s.oi_pi_t_u[i];
    Hit h;
    h.obj_index = float_as_int(a.x);
    h.prim_index = float_as_int(a.y);
    h.t = a.z, h.u = a.w;
    h.v = s.v[i];
    return h;
}
RT_HD void store_hit(const HitSoA &s, uint32_t i, const Hit &h) {
    s.oi_pi_t_u[i] = mkfloat4(int_as_float(h.obj_index), int_as_float(h.prim_index), h.t, h.u);
    s.v[i] = h.v;
}

// Per-launch traversal work counters (instrumented builds only)
struct TravCount {
    uint32_t nodes, tris, instances;
    uint32_t max_stack; // deepest stack use (entries), to size the LDS stack
    uint32_t nodes4;    // 4-wide quantised BLAS nodes fetched (rt_bvh4.h; 64 B each) -- the WIDE walk only
};

} // namespace rt
