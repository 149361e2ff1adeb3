// rt_lights.h -- next-event estimation: light-tree descent and per-type light sampling.
//
// Restates reference internal/CoreRef.cpp:
//   calc_lnode_importance(light_cwbvh_node_t)   :1004-1066  (+ decode_oct_dir :935-947, decode_cosines :949-956,
//                                                             cos/sin_sub_clamped :900-912)
//   SampleSphericalRectangle / Triangle         :1288-1354 / :1356-1416 (+ orthogonalize, slerp, angle_between)
//   map_to_cone, create_tbn, sphere_intersection :679-723
//   SampleLightSource                            :3264-3614
//   EvalTriLightFactor(light_cwbvh_node_t)       :4692-4736
//   SampleLatlong_RGBE                           :2995-3039
// The reference evaluates the 8 children of a light-tree node in two 4-lane SSE groups; every lane is
// independent, so the per-child scalar form below performs the same operations in the same order.
#pragma once

#include "rt_rng.h"
#include "rt_texture.h"
#include "rt_types.h"

namespace rt {

struct LightSample { // CoreRef.h:117-125 light_sample_t
    f3 col, L, lp;
    float area, dist_mul, pdf;
    bool cast_shadow, from_env;
    uint32_t ray_flags;
};
RT_HD LightSample make_light_sample() {
    LightSample ls;
    ls.col = ls.L = ls.lp = {0.0f, 0.0f, 0.0f};
    ls.area = 0.0f, ls.dist_mul = 1.0f, ls.pdf = 0.0f;
    ls.cast_shadow = false, ls.from_env = false, ls.ray_flags = 0;
    return ls;
}

// CoreRef.cpp:679-689
RT_HD void create_tbn(const f3 N, f3 &out_T, f3 &out_B) {
    f3 U;
    if (fabsf(N.y) < 0.999f) {
        U = {0.0f, 1.0f, 0.0f};
    } else {
        U = {1.0f, 0.0f, 0.0f};
    }
    out_T = normalize(cross(U, N));
    out_B = cross(N, out_T);
}

// CoreRef.cpp:691-714
RT_HD f3 map_to_cone(float r1, float r2, f3 N, float radius) {
    const f2 offset = {2.0f * r1 - 1.0f, 2.0f * r2 - 1.0f};
    if (offset.x == 0.0f && offset.y == 0.0f) {
        return N;
    }
    float theta, r;
    if (fabsf(offset.x) > fabsf(offset.y)) {
        r = offset.x;
        theta = 0.25f * PI * (offset.y / offset.x);
    } else {
        r = offset.y;
        theta = 0.5f * PI * (1.0f - 0.5f * (offset.x / offset.y));
    }
    const f2 sincos_theta = portable_sincos(theta);
    const f2 uv = {radius * r * sincos_theta.y, radius * r * sincos_theta.x};

    f3 LT, LB;
    create_tbn(normalize(N), LT, LB);
    return N + uv.x * LT + uv.y * LB;
}

// CoreRef.cpp:716-723
RT_HD float sphere_intersection(const f3 center, const float radius, const f3 ro, const f3 rd) {
    const f3 oc = ro - center;
    const float a = dot(rd, rd);
    const float b = 2 * dot(oc, rd);
    const float c = dot(oc, oc) - radius * radius;
    const float discriminant = b * b - 4 * a * c;
    return (-b - sqrtf(fmaxf(discriminant, 0.0f))) / (2 * a);
}

// CoreRef.cpp:1104-1129
RT_HD f3 orthogonalize(const f3 a, const f3 b) { return normalize(b - dot(a, b) * a); }
RT_HD f3 slerp(const f3 start, const f3 end, const float percent) {
    float cos_theta = dot(start, end);
    cos_theta = clampf(cos_theta, -1.0f, 1.0f);
    const float theta = acosf(cos_theta) * percent;
    const f3 relative_vec = safe_normalize(end - start * cos_theta);
    const f2 sincos_theta = portable_sincos(theta);
    return start * sincos_theta.y + relative_vec * sincos_theta.x;
}
// CoreRef.cpp:1274-1280
RT_HD float angle_between(const f3 v1, const f3 v2) {
    if (dot(v1, v2) < 0) {
        return PI - 2 * portable_asinf(length(v1 + v2) / 2);
    } else {
        return 2 * portable_asinf(length(v2 - v1) / 2);
    }
}

// CoreRef.cpp:1288-1354.  out_p may be null (pdf-only evaluation)
RT_HD float sample_spherical_rectangle(const f3 P, const f3 light_pos, const f3 axis_u, const f3 axis_v, const f2 Xi,
                                       f3 *out_p) {
    const f3 corner = light_pos - 0.5f * axis_u - 0.5f * axis_v;

    float axisu_len, axisv_len;
    const f3 x = normalize_len(axis_u, axisu_len), y = normalize_len(axis_v, axisv_len);
    f3 z = cross(x, y);

    // compute rectangle coords in local reference system
    const f3 dir = corner - P;
    float z0 = dot(dir, z);
    // flip z to make it point against Q
    if (z0 > 0.0f) {
        z = -z;
        z0 = -z0;
    }
    const float x0 = dot(dir, x);
    const float y0 = dot(dir, y);
    const float x1 = x0 + axisu_len;
    const float y1 = y0 + axisv_len;
    // compute internal angles (gamma_i)
    const f4 diff = mk4(x0, y1, x1, y0) - mk4(x1, y0, x0, y1);
    f4 nz = mk4(y0, x1, y1, x0) * diff;
    {
        const float z0z0 = z0 * z0;
        const f4 den = {sqrtf(z0z0 * diff.x * diff.x + nz.x * nz.x), sqrtf(z0z0 * diff.y * diff.y + nz.y * nz.y),
                        sqrtf(z0z0 * diff.z * diff.z + nz.z * nz.z), sqrtf(z0z0 * diff.w * diff.w + nz.w * nz.w)};
        nz = nz / den;
    }
    const float g0 = portable_acosf(clampf(-nz.x * nz.y, -1.0f, 1.0f));
    const float g1 = portable_acosf(clampf(-nz.y * nz.z, -1.0f, 1.0f));
    const float g2 = portable_acosf(clampf(-nz.z * nz.w, -1.0f, 1.0f));
    const float g3 = portable_acosf(clampf(-nz.w * nz.x, -1.0f, 1.0f));
    // compute predefined constants
    const float b0 = nz.x;
    const float b1 = nz.z;
    const float b0sq = b0 * b0;
    const float k = 2 * PI - g2 - g3;
    // compute solid angle from internal angles
    const float area = g0 + g1 - k;
    if (area <= SPHERICAL_AREA_THRESHOLD) {
        return 0.0f;
    }

    if (out_p) {
        // compute cu
        const float au = Xi.x * area + k;
        const f2 sincos_au = portable_sincos(au);
        const float fu = safe_div((sincos_au.y * b0 - b1), sincos_au.x);
        float cu = 1.0f / sqrtf(fu * fu + b0sq) * (fu > 0.0f ? 1.0f : -1.0f);
        cu = clampf(cu, -1.0f, 1.0f);
        // compute xu
        float xu = -(cu * z0) / fmaxf(sqrtf(1.0f - cu * cu), 1e-7f);
        xu = clampf(xu, x0, x1);
        // compute yv
        const float z0sq = z0 * z0;
        const float y0sq = y0 * y0;
        const float y1sq = y1 * y1;
        const float d = sqrtf(xu * xu + z0sq);
        const float h0 = y0 / sqrtf(d * d + y0sq);
        const float h1 = y1 / sqrtf(d * d + y1sq);
        const float hv = h0 + Xi.y * (h1 - h0), hv2 = hv * hv;
        const float yv = (hv2 < 1.0f - 1e-6f) ? (hv * d) / sqrtf(1.0f - hv2) : y1;

        // transform (xu, yv, z0) to world coords
        (*out_p) = P + xu * x + yv * y + z0 * z;
    }
    return (1.0f / area);
}

// CoreRef.cpp:1356-1416.  out_dir may be null (pdf-only evaluation)
RT_HD float sample_spherical_triangle(const f3 P, const f3 p1, const f3 p2, const f3 p3, const f2 Xi, f3 *out_dir) {
    // Setup spherical triangle
    const f3 A = normalize(p1 - P), B = normalize(p2 - P), C = normalize(p3 - P);

    // calculate internal angles of spherical triangle: alpha, beta and gamma
    const f3 BA = orthogonalize(A, B - A);
    const f3 CA = orthogonalize(A, C - A);
    const f3 AB = orthogonalize(B, A - B);
    const f3 CB = orthogonalize(B, C - B);
    const f3 BC = orthogonalize(C, B - C);
    const f3 AC = orthogonalize(C, A - C);

    const float alpha = angle_between(BA, CA);
    const float beta = angle_between(AB, CB);
    const float gamma = angle_between(BC, AC);

    const float area = alpha + beta + gamma - PI;
    if (area <= SPHERICAL_AREA_THRESHOLD) {
        return 0.0f;
    }

    if (out_dir) {
        // calculate arc lengths for edges of spherical triangle
        const float b = portable_acosf(clampf(dot(C, A), -1.0f, 1.0f));
        const float c = portable_acosf(clampf(dot(A, B), -1.0f, 1.0f));

        // Use one random variable to select the new area
        const float area_S = Xi.x * area;

        // Save the sine and cosine of the angle delta
        const f2 sincos_area = portable_sincos(area_S - alpha);
        const float p = sincos_area.x;
        const float q = sincos_area.y;

        // Compute the pair(u; v) that determines sin(beta_s) and cos(beta_s)
        const f2 sincos_alpha = portable_sincos(alpha);
        const float u = q - sincos_alpha.y;
        const float v = p + sincos_alpha.x * portable_cos(c);

        // Compute the s coordinate as normalized arc length from A to C_s
        const float denom = ((v * p + u * q) * sincos_alpha.x);
        const float s = safe_div(1.0f, b) *
                        portable_acosf(clampf(safe_div(((v * q - u * p) * sincos_alpha.y - v), denom), -1.0f, 1.0f));

        // Compute the third vertex of the sub - triangle
        const f3 C_s = slerp(A, C, s);

        // Compute the t coordinate using C_s and Xi[1]
        const float denom2 = portable_acosf(clampf(dot(C_s, B), -1.0f, 1.0f));
        const float t = safe_div(portable_acosf(clampf(1.0f - Xi.y * (1.0f - dot(C_s, B)), -1.0f, 1.0f)), denom2);

        // Construct the corresponding point on the sphere.
        (*out_dir) = slerp(B, C_s, t);
    }
    // return pdf
    return (1.0f / area);
}

// ---- light tree ---------------------------------------------------------------------------------------
// unpacked child box of a quantised node (shared prologue of calc_lnode_importance / bbox_test_oct)
RT_HD void cw_child_bounds(const rayhip_light_cwbvh_node &n, const int i, float bmin[3], float bmax[3]) {
    const float ext[3] = {(n.bbox_max[0] - n.bbox_min[0]) / 255.0f, (n.bbox_max[1] - n.bbox_min[1]) / 255.0f,
                          (n.bbox_max[2] - n.bbox_min[2]) / 255.0f};
    bmin[0] = bmin[1] = bmin[2] = -MAX_DIST;
    bmax[0] = bmax[1] = bmax[2] = MAX_DIST;
    if (n.ch_bbox_min[0][i] != 0xff || n.ch_bbox_max[0][i] != 0) {
        bmin[0] = n.bbox_min[0] + float(int(n.ch_bbox_min[0][i])) * ext[0];
        bmin[1] = n.bbox_min[1] + float(int(n.ch_bbox_min[1][i])) * ext[1];
        bmin[2] = n.bbox_min[2] + float(int(n.ch_bbox_min[2][i])) * ext[2];

        bmax[0] = n.bbox_min[0] + float(int(n.ch_bbox_max[0][i])) * ext[0];
        bmax[1] = n.bbox_min[1] + float(int(n.ch_bbox_max[1][i])) * ext[1];
        bmax[2] = n.bbox_min[2] + float(int(n.ch_bbox_max[2][i])) * ext[2];
    }
}

RT_HD float cos_sub_clamped(float sin_a, float cos_a, float sin_b, float cos_b) {
    return (cos_a > cos_b) ? 1.0f : (cos_a * cos_b + sin_a * sin_b);
}
RT_HD float sin_sub_clamped(float sin_a, float cos_a, float sin_b, float cos_b) {
    return (cos_a > cos_b) ? 0.0f : (sin_a * cos_b - cos_a * sin_b);
}

// calc_lnode_importance (CoreRef.cpp:1004-1066) is split in two:
//   * decode_lnode_child: everything that depends on the node only (box un-quantisation, decode_oct_dir :935-947,
//     decode_cosines :949-956, box centre / half-diagonal) -- about half of the arithmetic, 8 divisions and 3 square
//     roots per child.  It is evaluated ONCE per scene into the `light_children` table (fill_light_children below) when
//     the scene is uploaded, by this very function compiled for the host (same IEEE operations, so the same bits the
//     device would produce), instead of 8 x depth times per shade point;
//   * lnode_child_importance: the part that depends on the shade point P.
// Measured before the split: light-tree work was 56 % of the shade kernel's time (41 % NEE + 15 % emissive-hit MIS).
struct LNodeChild {
    float4 axis_extent; // decoded cone axis, half-diagonal of the child box
    float4 pc_valid;    // box centre, 1 if the slot has a finite box (else importance = flux)
    float4 cosines;     // cos_omega_n, sin_omega_n, cos_omega_e, flux
};

RT_HD LNodeChild decode_lnode_child(const rayhip_light_cwbvh_node &n, const int i) {
    float bmin[3], bmax[3];
    cw_child_bounds(n, i, bmin, bmax);

    LNodeChild o;
    o.axis_extent = mkfloat4(0.0f, 0.0f, 0.0f, 0.0f);
    o.pc_valid = mkfloat4(0.0f, 0.0f, 0.0f, 0.0f);
    o.cosines = mkfloat4(0.0f, 0.0f, 0.0f, n.flux[i]);
    if (bmin[0] > -MAX_DIST) {
        // decode_oct_dir, CoreRef.cpp:935-947
        const uint32_t oct = n.axis[i];
        float ax = -1.0f + 2.0f * float((oct >> 16) & 0x0000ffff) / 65535.0f;
        float ay = -1.0f + 2.0f * float(oct & 0x0000ffff) / 65535.0f;
        float az = 1.0f - fabsf(ax) - fabsf(ay);
        if (az < 0.0f) {
            const float temp = ax;
            ax = (1.0f - fabsf(ay)) * copysignf(1.0f, temp);
            ay = (1.0f - fabsf(temp)) * copysignf(1.0f, ay);
        }
        {
            const float l = sqrtf(ax * ax + ay * ay + az * az);
            ax = ax / l, ay = ay / l, az = az / l;
        }
        const float ext[3] = {bmax[0] - bmin[0], bmax[1] - bmin[1], bmax[2] - bmin[2]};
        const float extent = 0.5f * sqrtf(ext[0] * ext[0] + ext[1] * ext[1] + ext[2] * ext[2]);

        // decode_cosines, CoreRef.cpp:949-956
        const uint32_t cv = n.cos_omega_ne[i];
        const float cos_omega_n = 2.0f * (float((cv >> 16) & 0x0000ffff) / 65534.0f) - 1.0f;
        const float cos_omega_e = 2.0f * (float(cv & 0x0000ffff) / 65534.0f) - 1.0f;
        const float sin_omega_n = sqrtf(1.0f - cos_omega_n * cos_omega_n);

        o.axis_extent = mkfloat4(ax, ay, az, extent);
        o.pc_valid = mkfloat4(0.5f * (bmin[0] + bmax[0]), 0.5f * (bmin[1] + bmax[1]), 0.5f * (bmin[2] + bmax[2]), 1.0f);
        o.cosines = mkfloat4(cos_omega_n, sin_omega_n, cos_omega_e, n.flux[i]);
    }
    return o;
}

// Division and square root of the importance heuristic.  The importance of a light-tree child only steers WHICH light is
// sampled and with what probability (the estimator divides by that same probability), it is not a radiometric quantity:
// on the device these use the hardware reciprocal / square root (1 ulp) instead of the correctly rounded sequences
// (10-12 instructions each; five divisions and four square roots per child, eight children per level: measured 26 %
// of the shade kernel).  The host build keeps the IEEE operations and stays bit-exact with RendererRef; the device
// result moves by ~1e-7 relative in the pick probability, far inside the image tolerance (tests/test_gpu_parity.py).
// -DRT_EXACT_IMPORTANCE restores the IEEE operations on the device.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(RT_EXACT_IMPORTANCE)
RT_HD float imp_div(const float a, const float b) { return a * __builtin_amdgcn_rcpf(b); }
RT_HD float imp_sqrt(const float a) { return __builtin_amdgcn_sqrtf(a); }
#else
RT_HD float imp_div(const float a, const float b) { return a / b; }
RT_HD float imp_sqrt(const float a) { return sqrtf(a); }
#endif

// importance of one child as seen from P
RT_HD float lnode_child_importance(const LNodeChild &ch, const f3 P) {
    float imp = ch.cosines.w;
    // (zero flux -- an empty slot or a black light -- stays zero whatever the geometric term: 0 * mul, mul finite
    // unless P sits exactly on a degenerate box; skipping it lets a wavefront whose lanes are all at sparsely filled
    // nodes, e.g. the root, jump over the slot)
    if (ch.pc_valid.w != 0.0f && imp != 0.0f) {
        const float ax = ch.axis_extent.x, ay = ch.axis_extent.y, az = ch.axis_extent.z, extent = ch.axis_extent.w;
        float wi[3] = {P.x - ch.pc_valid.x, P.y - ch.pc_valid.y, P.z - ch.pc_valid.z};
        const float dist2 = wi[0] * wi[0] + wi[1] * wi[1] + wi[2] * wi[2];
        const float dist = imp_sqrt(dist2);
        wi[0] = imp_div(wi[0], dist), wi[1] = imp_div(wi[1], dist), wi[2] = imp_div(wi[2], dist);

        const float v_len2 = sse_max(dist2, extent);

        const float cos_omega_w = ax * wi[0] + ay * wi[1] + az * wi[2];
        const float sin_omega_w = imp_sqrt(sse_max(1.0f - cos_omega_w * cos_omega_w, 0.0f));

        float cos_omega_b = imp_sqrt(sse_max(1.0f - imp_div(extent * extent, dist2), 0.0f));
        if (dist2 < extent * extent) {
            cos_omega_b = -1.0f;
        }
        const float sin_omega_b = imp_sqrt(1.0f - cos_omega_b * cos_omega_b);

        const float cos_omega_n = ch.cosines.x, sin_omega_n = ch.cosines.y, cos_omega_e = ch.cosines.z;

        const float cos_omega_x = cos_sub_clamped(sin_omega_w, cos_omega_w, sin_omega_n, cos_omega_n);
        const float sin_omega_x = sin_sub_clamped(sin_omega_w, cos_omega_w, sin_omega_n, cos_omega_n);
        const float cos_omega = cos_sub_clamped(sin_omega_x, cos_omega_x, sin_omega_b, cos_omega_b);

        float mul = 0.0f;
        if (cos_omega > cos_omega_e) {
            mul = imp_div(cos_omega, v_len2);
        }
        imp = imp * mul;
    }
    return imp;
}

// `light_children` table: LIGHT_CHILDREN_STRIDE float4 per light-tree node -- [0..1] the eight fluxes, then three float4
// (LNodeChild) per child.  The fluxes come first because the reference's 8-wide tree is sparsely filled (Sponza-class
// scene: 31 of 41 nodes hold two children, 57 % of all slots are empty) and an empty or black slot needs nothing else:
// two loads tell a lane which of the 24 others to issue.  The descent is bound by the texture addresser, not by
// arithmetic (384 B per lane and level before), so the loads not issued are the saving.
constexpr int LIGHT_CHILDREN_STRIDE = 26;
RT_HD void fill_light_children(const rayhip_light_cwbvh_node &n, float4 *out /* [LIGHT_CHILDREN_STRIDE] */) {
    out[0] = mkfloat4(n.flux[0], n.flux[1], n.flux[2], n.flux[3]);
    out[1] = mkfloat4(n.flux[4], n.flux[5], n.flux[6], n.flux[7]);
    for (int i = 0; i < 8; ++i) {
        const LNodeChild ch = decode_lnode_child(n, i);
        out[2 + 3 * i + 0] = ch.axis_extent, out[2 + 3 * i + 1] = ch.pc_valid, out[2 + 3 * i + 2] = ch.cosines;
    }
}

// `light_tri_geom` table: what sampling a TRI light needs from the scene, resolved once per scene instead of once per
// sample -- the reference walks light -> mesh instance (xform) -> vtx_indices -> three 44-byte vertices and transforms the
// corners (CoreRef.cpp:3530-3545): three dependent round trips and ~25 scattered loads per shade point.  Four float4 per
// light: (p1, uv1.x) (p2, uv1.y) (p3, uv2.x) (uv2.y, uv3.x, uv3.y, -), corners already in world space, computed by this
// very function on the host (same IEEE operations as the device would perform, so the same bits).
RT_HD void fill_light_tri_geom(const rayhip_light &l, const rayhip_mesh_instance *mesh_instances, const uint32_t *vtx_indices,
                               const rayhip_vertex *vertices, float4 *out /* [4] */) {
    out[0] = out[1] = out[2] = out[3] = mkfloat4(0.0f, 0.0f, 0.0f, 0.0f);
    if (light_type(l) != LIGHT_TYPE_TRI) {
        return;
    }
    const uint32_t ltri_index = float_as_uint(l.params[0]);
    const rayhip_mesh_instance &lmi = mesh_instances[float_as_uint(l.params[1])];
    const rayhip_vertex &v1 = vertices[vtx_indices[ltri_index * 3 + 0]], &v2 = vertices[vtx_indices[ltri_index * 3 + 1]],
                        &v3 = vertices[vtx_indices[ltri_index * 3 + 2]];
    const f3 p1 = transform_point(mk3(v1.p), lmi.xform), p2 = transform_point(mk3(v2.p), lmi.xform),
             p3 = transform_point(mk3(v3.p), lmi.xform);
    out[0] = mkfloat4(p1.x, p1.y, p1.z, v1.t[0]);
    out[1] = mkfloat4(p2.x, p2.y, p2.z, v1.t[1]);
    out[2] = mkfloat4(p3.x, p3.y, p3.z, v2.t[0]);
    out[3] = mkfloat4(v2.t[1], v3.t[0], v3.t[1], 0.0f);
}

// importance of the eight children of light-tree node `node_index` (through the table fill_light_children filled)
RT_HD void calc_lnode_importance(const SceneView &sc, const uint32_t node_index, const f3 P, float importance[8]) {
    const float4 *t = sc.light_children + size_t(node_index) * LIGHT_CHILDREN_STRIDE;
    const float4 f0 = t[0], f1 = t[1];
    const float flux[8] = {f0.x, f0.y, f0.z, f0.w, f1.x, f1.y, f1.z, f1.w};
    for (int i = 0; i < 8; ++i) {
        importance[i] = flux[i]; // (zero flux: lnode_child_importance returns the flux itself)
        if (flux[i] != 0.0f) {
            LNodeChild ch;
            ch.axis_extent = t[2 + 3 * i + 0], ch.pc_valid = t[2 + 3 * i + 1], ch.cosines = t[2 + 3 * i + 2];
            importance[i] = lnode_child_importance(ch, P);
        }
    }
}
// hsum(fvec4{imp[0..3]} + fvec4{imp[4..7]}), SSE2 hsum order
RT_HD float total_importance8(const float imp[8]) {
    return (((imp[0] + imp[4]) + (imp[1] + imp[5])) + (imp[2] + imp[6])) + (imp[3] + imp[7]);
}

// ---- env-map importance quadtree -----------------------------------------------------------------------
// Core.cpp:110-141 (libm sinf / cosf / atan2f there, i.e. the device's versions here: last-ulp differences)
RT_HD f3 canonical_to_dir(const f2 p, const float y_rotation) {
    const float cos_theta = 2 * p.x - 1;
    float phi = 2 * PI * p.y + y_rotation;
    if (phi < 0) {
        phi += 2 * PI;
    }
    if (phi > 2 * PI) {
        phi -= 2 * PI;
    }
    const float sin_theta = sqrtf(1 - cos_theta * cos_theta);
    const float sin_phi = sinf(phi);
    const float cos_phi = cosf(phi);
    return f3{sin_theta * cos_phi, cos_theta, -sin_theta * sin_phi};
}
RT_HD f2 dir_to_canonical(const f3 d, const float y_rotation) {
    const float cos_theta = fminf(fmaxf(d.y, -1.0f), 1.0f);
    float phi = -atan2f(d.z, d.x) + y_rotation;
    if (phi < 0) {
        phi += 2 * PI;
    }
    if (phi > 2 * PI) {
        phi -= 2 * PI;
    }
    return f2{(cos_theta + 1.0f) / 2.0f, phi / (2.0f * PI)};
}
RT_HD float quad_at(const float4 q, const int i) { return i == 0 ? q.x : (i == 1 ? q.y : (i == 2 ? q.z : q.w)); }

// pdf of direction L under the quadtree, CoreRef.cpp:4738-4771
RT_HD float evaluate_env_qtree(const SceneView &sc, const float y_rotation, const f3 L) {
    const int qtree_levels = sc.env.qtree_levels;
    int res = 2;
    int lod = qtree_levels - 1;

    const f2 p = dir_to_canonical(L, -y_rotation);
    float factor = 1.0f;

    while (lod >= 0) {
        const int x = clampi(int(p.x * float(res)), 0, res - 1);
        const int y = clampi(int(p.y * float(res)), 0, res - 1);

        int index = 0;
        index |= (x & 1) << 0;
        index |= (y & 1) << 1;

        const int qx = x / 2;
        const int qy = y / 2;

        const float4 quad = sc.env_qtree[sc.env_qtree_offset[lod] + uint32_t(qy * res / 2 + qx)];
        const float total = quad.x + quad.y + quad.z + quad.w;
        if (total <= 0.0f) {
            break;
        }
        factor *= 4.0f * quad_at(quad, index) / total;

        --lod;
        res *= 2;
    }
    return factor / (4.0f * PI);
}

// importance-sampled direction + pdf, CoreRef.cpp:4773-4839
RT_HD f4 sample_env_qtree(const SceneView &sc, const float y_rotation, const float rand, const float rx, const float ry) {
    const int qtree_levels = sc.env.qtree_levels;
    int res = 2;
    float step = 1.0f / float(res);

    float sample = rand;
    int lod = qtree_levels - 1;

    f2 origin = {0.0f, 0.0f};
    float factor = 1.0f;

    while (lod >= 0) {
        const int qx = int(origin.x * float(res)) / 2;
        const int qy = int(origin.y * float(res)) / 2;

        const float4 quad = sc.env_qtree[sc.env_qtree_offset[lod] + uint32_t(qy * res / 2 + qx)];

        const float top_left = quad.x;
        const float top_right = quad.y;
        float partial = top_left + quad.z;
        const float total = partial + top_right + quad.w;
        if (total <= 0.0f) {
            break;
        }

        float boundary = partial / total;

        int index = 0;
        if (sample < boundary) {
            sample /= boundary;
            boundary = top_left / partial;
        } else {
            partial = total - partial;
            origin.x = origin.x + step;
            sample = (sample - boundary) / (1.0f - boundary);
            boundary = top_right / partial;
            index |= (1 << 0);
        }

        if (sample < boundary) {
            sample /= boundary;
        } else {
            origin.y = origin.y + step;
            sample = (sample - boundary) / (1.0f - boundary);
            index |= (1 << 1);
        }

        factor *= 4.0f * quad_at(quad, index) / total;

        --lod;
        res *= 2;
        step *= 0.5f;
    }

    origin.x += 2 * step * rx;
    origin.y += 2 * step * ry;

    const f3 dir = canonical_to_dir(origin, y_rotation);
    return f4{dir.x, dir.y, dir.z, factor / (4.0f * PI)};
}

// CoreRef.cpp:2995-3039 (stochastic branch)
RT_HD f3 sample_latlong_rgbe(const SceneView &sc, const uint32_t handle, const f3 dir, const float y_rotation, const f2 rnd) {
    const float theta = acosf(clampf(dir.y, -1.0f, 1.0f)) / PI;
    float phi = atan2f(dir.z, dir.x) + y_rotation;
    if (phi < 0) {
        phi += 2 * PI;
    }
    if (phi > 2 * PI) {
        phi -= 2 * PI;
    }
    const float u = fractf(0.5f * phi / PI);

    const rayhip_texture &t = sc.textures[sc.tex_table[0] + (handle & 0x00ffffffu)];
    f2 uvs = {u * float(t.width[0]), theta * float(t.height[0])};
    uvs = uvs + rnd;
    int ix = int(uvs.x), iy = int(uvs.y);
    ix %= int(t.width[0]);
    iy %= int(t.height[0]);
    const uint32_t px = sc.texels[t.offset[0] + uint32_t(iy) * t.width[0] + uint32_t(ix)];
    // rgbe_to_rgb, CoreRef.h:229-232 + to_norm_float Core.h:411-418
    const float f = exp2f(float((px >> 24) & 0xffu) - 128.0f);
    f3 ret;
    {
        const uint32_t v0 = px & 0xffu, v1 = (px >> 8) & 0xffu, v2 = (px >> 16) & 0xffu;
        ret.x = (uint_as_float(0x3f800000u + v0 * 0x8080u + (v0 + 1) / 2) - 1.0f) * f;
        ret.y = (uint_as_float(0x3f800000u + v1 * 0x8080u + (v1 + 1) / 2) - 1.0f) * f;
        ret.z = (uint_as_float(0x3f800000u + v2 * 0x8080u + (v2 + 1) / 2) - 1.0f) * f;
    }
    return ret;
}

// Ref::SampleLightSource, CoreRef.cpp:3264-3614
RT_HD void sample_light_source(const SceneView &sc, const f3 P, const f3 T, const f3 B, const f3 N,
                               const float rand_pick_light, const f2 rand_light_uv, const f2 rand_tex_uv, LightSample &ls) {
    float u1 = rand_pick_light;

    uint32_t light_index;
    float factor;
    { // USE_HIERARCHICAL_NEE
        factor = 1.0f;
        uint32_t i = 0; // start from root
        while ((i & LEAF_NODE_BIT) == 0) {
            const rayhip_light_cwbvh_node &node = sc.light_cwnodes[i];
            float importance[8];
            calc_lnode_importance(sc, i, P, importance);

            const float total_importance = total_importance8(importance);
            if (total_importance == 0.0f) {
                // failed to find lightsource for sampling
                return;
            }

            float factors[8];
            for (int j = 0; j < 8; ++j) {
                factors[j] = imp_div(importance[j], total_importance);
            }
            float factors_cdf[9];
            factors_cdf[0] = 0.0f;
            for (int j = 0; j < 8; ++j) {
                factors_cdf[j + 1] = factors_cdf[j] + factors[j];
            }
            // make sure cdf ends with 1.0
            for (int j = 0; j < 8; ++j) {
                if (factors_cdf[j + 1] == factors_cdf[8]) {
                    factors_cdf[j + 1] = 1.01f;
                }
            }
            int next = 0;
            for (int j = 1; j < 9; ++j) {
                next += (factors_cdf[j] <= u1) ? 1 : 0;
            }
            // factors_cdf[next], factors[next] through selects: with a run-time index the three arrays would have to
            // live in scratch memory (8 + 9 + 8 stores and the loads back, per tree level)
            float cdf_next = factors_cdf[0], factor_next = factors[0];
            for (int j = 1; j < 8; ++j) {
                cdf_next = (next == j) ? factors_cdf[j] : cdf_next;
                factor_next = (next == j) ? factors[j] : factor_next;
            }

            u1 = fractf((u1 - cdf_next) / factor_next);
            i = node.child[next];
            factor *= factor_next;
        }
        light_index = (i & PRIM_INDEX_BITS);
        factor = 1.0f / factor;
    }
    RT_PROF(15)
    const rayhip_light &l = sc.lights[light_index];
    const uint32_t ltype = light_type(l);

    ls.col = mk3(l.col);
    ls.cast_shadow = light_cast_shadow(l);
    ls.from_env = false;

    if (ltype == LIGHT_TYPE_SPHERE) {
        const float r1 = rand_light_uv.x, r2 = rand_light_uv.y;
        const float radius = l.params[7];

        const f3 center = mk3(&l.params[0]);
        float d;
        const f3 light_normal = normalize_len(center - P, d);

        if (d > radius) {
            const float temp = sqrtf(d * d - radius * radius);
            const float disk_radius = (temp * radius) / d;
            float disk_dist = radius > 0.0f ? ((temp * disk_radius) / radius) : d;
            const f3 sampled_dir = normalize_len(map_to_cone(r1, r2, disk_dist * light_normal, disk_radius), disk_dist);

            if (radius > 0.0f) {
                const float ls_dist = sphere_intersection(center, radius, P, sampled_dir);

                const f3 light_surf_pos = P + sampled_dir * ls_dist;
                const f3 light_forward = normalize(light_surf_pos - center);

                const float sampled_area = PI * disk_radius * disk_radius;
                const float cos_theta = dot(sampled_dir, light_normal);

                ls.lp = offset_ray(light_surf_pos, light_forward);
                ls.pdf = (disk_dist * disk_dist) / (sampled_area * cos_theta);
            } else {
                ls.lp = center;
                ls.pdf = (disk_dist * disk_dist) / PI;
            }
            ls.L = sampled_dir;
            ls.area = PI * disk_radius * disk_radius;
            ls.ray_flags = light_ray_visibility(l);

            if (!light_visible(l)) {
                ls.area = 0.0f;
            }

            const float spot = l.params[8], blend = l.params[9];
            if (spot > 0.0f) {
                const float _dot = -dot(ls.L, mk3(&l.params[4]));
                if (_dot > 0.0f) {
                    const float _angle = acosf(saturatef(_dot));
                    ls.col *= saturatef((spot - _angle) / blend);
                } else {
                    ls.col *= 0.0f;
                }
            }
        }
    } else if (ltype == LIGHT_TYPE_DIR) {
        const f3 ldir = mk3(&l.params[0]);
        const float tan_angle = l.params[4];
        ls.L = ldir;
        ls.area = 0.0f;
        ls.pdf = 1.0f;
        if (tan_angle != 0.0f) {
            const float r1 = rand_light_uv.x, r2 = rand_light_uv.y;

            const float radius = tan_angle;
            ls.L = normalize(map_to_cone(r1, r2, ls.L, radius));
            ls.area = PI * radius * radius;

            const float cos_theta = dot(ls.L, ldir);
            ls.pdf = 1.0f / (ls.area * cos_theta);
        }
        ls.lp = P + ls.L;
        ls.dist_mul = MAX_DIST;
        ls.ray_flags = light_ray_visibility(l);

        if (!light_visible(l)) {
            ls.area = 0.0f;
        }
    } else if (ltype == LIGHT_TYPE_RECT) {
        const f3 light_pos = mk3(&l.params[0]);
        const f3 light_u = mk3(&l.params[4]), light_v = mk3(&l.params[8]);
        const f3 light_forward = normalize(cross(light_u, light_v));
        const float rect_area = l.params[3];

        f3 lp = {0.0f, 0.0f, 0.0f};
        float pdf = sample_spherical_rectangle(P, light_pos, light_u, light_v, rand_light_uv, &lp);
        if (pdf <= 0.0f) {
            const float r1 = rand_light_uv.x - 0.5f, r2 = rand_light_uv.y - 0.5f;
            lp = light_pos + light_u * r1 + light_v * r2;
        }

        float ls_dist;
        ls.L = normalize_len(lp - P, ls_dist);
        ls.ray_flags = light_ray_visibility(l);

        const float cos_theta = dot(-ls.L, light_forward);
        if (cos_theta > 0.0f) {
            ls.lp = offset_ray(lp, light_forward);
            ls.pdf = (pdf > 0.0f) ? pdf : (ls_dist * ls_dist) / (rect_area * cos_theta);
            ls.area = light_visible(l) ? rect_area : 0.0f;
            if (light_sky_portal(l)) {
                f3 env_col = mk3(sc.env.env_col);
                if (sc.env.env_map != 0xffffffff) {
                    env_col *= sample_latlong_rgbe(sc, sc.env.env_map, ls.L, sc.env.env_map_rotation, rand_tex_uv);
                }
                ls.col *= env_col;
                ls.from_env = true;
            }
        }
    } else if (ltype == LIGHT_TYPE_DISK) {
        const f3 light_pos = mk3(&l.params[0]);
        const f3 light_u = mk3(&l.params[4]), light_v = mk3(&l.params[8]);

        const float r1 = rand_light_uv.x, r2 = rand_light_uv.y;

        f2 offset = {2.0f * r1 - 1.0f, 2.0f * r2 - 1.0f};
        if (offset.x != 0.0f && offset.y != 0.0f) {
            float theta, r;
            if (fabsf(offset.x) > fabsf(offset.y)) {
                r = offset.x;
                theta = 0.25f * PI * (offset.y / offset.x);
            } else {
                r = offset.y;
                theta = 0.5f * PI - 0.25f * PI * (offset.x / offset.y);
            }
            const f2 sincos_theta = portable_sincos(theta);
            offset.x = 0.5f * r * sincos_theta.y;
            offset.y = 0.5f * r * sincos_theta.x;
        }

        const f3 lp = light_pos + light_u * offset.x + light_v * offset.y;
        const f3 light_forward = normalize(cross(light_u, light_v));

        ls.lp = offset_ray(lp, light_forward);
        float ls_dist;
        ls.L = normalize_len(lp - P, ls_dist);
        ls.area = l.params[3];
        ls.ray_flags = light_ray_visibility(l);

        const float cos_theta = dot(-ls.L, light_forward);
        if (cos_theta > 0.0f) {
            ls.pdf = (ls_dist * ls_dist) / (ls.area * cos_theta);
        }
        if (!light_visible(l)) {
            ls.area = 0.0f;
        }
        if (light_sky_portal(l)) {
            f3 env_col = mk3(sc.env.env_col);
            if (sc.env.env_map != 0xffffffff) {
                env_col *= sample_latlong_rgbe(sc, sc.env.env_map, ls.L, sc.env.env_map_rotation, rand_tex_uv);
            }
            ls.col *= env_col;
            ls.from_env = true;
        }
    } else if (ltype == LIGHT_TYPE_LINE) {
        const f3 light_pos = mk3(&l.params[0]);
        const f3 light_dir = mk3(&l.params[8]);
        const float line_radius = l.params[7], line_height = l.params[11];

        const float r1 = rand_light_uv.x, r2 = rand_light_uv.y;

        const f3 center_to_surface = P - light_pos;

        const f3 light_u = normalize(cross(center_to_surface, light_dir));
        const f3 light_v = cross(light_u, light_dir);

        const float phi = PI * r1;
        const f2 sincos_phi = portable_sincos(phi);
        const f3 normal = sincos_phi.y * light_u + sincos_phi.x * light_v;

        const f3 lp = light_pos + normal * line_radius + (r2 - 0.5f) * light_dir * line_height;

        ls.lp = lp;
        float ls_dist;
        ls.L = normalize_len(lp - P, ls_dist);
        ls.area = l.params[3];
        ls.ray_flags = light_ray_visibility(l);

        const float cos_theta = 1.0f - fabsf(dot(ls.L, light_dir));
        if (cos_theta != 0.0f) {
            ls.pdf = (ls_dist * ls_dist) / (ls.area * cos_theta);
        }
        if (!light_visible(l)) {
            ls.area = 0.0f;
        }
    } else if (ltype == LIGHT_TYPE_TRI) {
        const uint32_t tex_index = float_as_uint(l.params[2]);

        // world-space corners and uvs from the per-light table (fill_light_tri_geom)
        const float4 *tg = sc.light_tri_geom + size_t(light_index) * 4;
        const float4 g0 = tg[0], g1 = tg[1], g2 = tg[2], g3 = tg[3];
        const f3 p1 = {g0.x, g0.y, g0.z}, p2 = {g1.x, g1.y, g1.z}, p3 = {g2.x, g2.y, g2.z};
        const f2 uv1 = mk2(g0.w, g1.w), uv2 = mk2(g2.w, g3.x), uv3 = mk2(g3.y, g3.z);

        const f3 e1 = p2 - p1, e2 = p3 - p1;
        float light_fwd_len;
        const f3 light_forward = normalize_len(cross(e1, e2), light_fwd_len);
        ls.area = 0.5f * light_fwd_len;
        ls.ray_flags = light_ray_visibility(l);

        f3 lp;
        f2 luvs;
        // Spherical triangle sampling
        float pdf = sample_spherical_triangle(P, p1, p2, p3, rand_light_uv, &ls.L);
        if (pdf > 0.0f) {
            // find u, v of intersection point
            const f3 pvec = cross(ls.L, e2);
            const f3 tvec = P - p1, qvec = cross(tvec, e1);

            const float inv_det = 1.0f / dot(e1, pvec);
            const float tri_u = dot(tvec, pvec) * inv_det, tri_v = dot(ls.L, qvec) * inv_det;

            lp = (1.0f - tri_u - tri_v) * p1 + tri_u * p2 + tri_v * p3;
            luvs = (1.0f - tri_u - tri_v) * uv1 + tri_u * uv2 + tri_v * uv3;
        } else {
            // Simple area sampling
            const float r1 = sqrtf(rand_light_uv.x), r2 = rand_light_uv.y;
            luvs = uv1 * (1.0f - r1) + r1 * (uv2 * (1.0f - r2) + uv3 * r2);
            lp = p1 * (1.0f - r1) + r1 * (p2 * (1.0f - r2) + p3 * r2);

            float ls_dist;
            ls.L = normalize_len(lp - P, ls_dist);

            const float cos_theta = -dot(ls.L, light_forward);
            pdf = safe_div_pos(ls_dist * ls_dist, ls.area * cos_theta);
        }

        float cos_theta = -dot(ls.L, light_forward);
        ls.lp = offset_ray(lp, cos_theta >= 0.0f ? light_forward : -light_forward);
        if (light_doublesided(l)) {
            cos_theta = fabsf(cos_theta);
        }

        if (cos_theta > 0.0f) {
            ls.pdf = pdf;
            if (tex_index != 0xffffffff) {
                const f4 tex_color = sample_color(sc, tex_index, luvs, 0 /* lod */, rand_tex_uv);
                ls.col *= xyz(tex_color);
            }
        }
    } else if (ltype == LIGHT_TYPE_ENV) {
        const float rx = rand_light_uv.x, ry = rand_light_uv.y;

        float env_pdf;
        if (sc.env.qtree_levels) {
            // Sample environment using quadtree
            const f4 dir_and_pdf = sample_env_qtree(sc, sc.env.env_map_rotation, u1, rx, ry);
            ls.L = f3{dir_and_pdf.x, dir_and_pdf.y, dir_and_pdf.z};
            env_pdf = dir_and_pdf.w;
        } else {
            // Sample environment as hemishpere
            const float phi = 2 * PI * ry;
            const f2 sincos_phi = portable_sincos(phi);
            const float cos_phi = sincos_phi.y, sin_phi = sincos_phi.x;

            const float dir = sqrtf(1.0f - rx * rx);
            const f3 V = {dir * cos_phi, dir * sin_phi, rx}; // in tangent-space

            ls.L = world_from_tangent(T, B, N, V);
            env_pdf = 0.5f / PI;
        }
        ls.col *= mk3(sc.env.env_col);

        if (sc.env.env_map != 0xffffffff) {
            ls.col *= sample_latlong_rgbe(sc, sc.env.env_map, ls.L, sc.env.env_map_rotation, rand_tex_uv);
        }

        ls.area = 1.0f;
        ls.lp = P + ls.L;
        ls.dist_mul = MAX_DIST;
        ls.pdf = env_pdf;
        ls.from_env = true;
        ls.ray_flags = light_ray_visibility(l);
    }

    ls.pdf /= factor;
}

// point-in-box test over the children of a quantised node: CoreRef.cpp:246-278 (bbox_test_oct(p, cwbvh))
RT_HD uint32_t cw_point_mask(const rayhip_light_cwbvh_node &n, const f3 p) {
    uint32_t mask = 0;
    for (int i = 0; i < 8; ++i) {
        float bmin[3], bmax[3];
        cw_child_bounds(n, i, bmin, bmax);
        const bool in = (bmin[0] <= p.x) && (bmin[1] <= p.y) && (bmin[2] <= p.z) && (bmax[0] >= p.x) && (bmax[1] >= p.y) &&
                        (bmax[2] >= p.z);
        mask |= (in ? 1u : 0u) << i;
    }
    return mask;
}

// Ref::EvalTriLightFactor(light_cwbvh_node_t), CoreRef.cpp:4692-4736
RT_HD float eval_tri_light_factor(const SceneView &sc, const f3 P, const f3 ro, const uint32_t tri_index) {
    uint32_t stack[MAX_STACK_SIZE];
    float stack_factors[MAX_STACK_SIZE];
    uint32_t stack_size = 0;

    stack_factors[stack_size] = 1.0f;
    stack[stack_size++] = 0;

    while (stack_size) {
        const uint32_t cur = stack[--stack_size];
        const float cur_factor = stack_factors[stack_size];

        if ((cur & LEAF_NODE_BIT) == 0) {
            const rayhip_light_cwbvh_node &node = sc.light_cwnodes[cur];
            uint32_t mask = cw_point_mask(node, P);
            if (mask) {
                float importance[8];
                calc_lnode_importance(sc, cur, ro, importance);

                const float total_importance = total_importance8(importance);
                if (total_importance == 0.0f) {
                    continue;
                }
                for (int i = 0; i < 8; ++i) { // GetFirstBit/ClearBit loop: ascending bit order
                    if ((mask >> i) & 1u) {
                        if (importance[i] > 0.0f) {
                            stack_factors[stack_size] = imp_div(cur_factor * importance[i], total_importance);
                            stack[stack_size++] = node.child[i];
                        }
                    }
                }
            }
        } else {
            const int light_index = int(cur & PRIM_INDEX_BITS);
            const rayhip_light &l = sc.lights[light_index];
            if (light_type(l) == LIGHT_TYPE_TRI && float_as_uint(l.params[0]) == tri_index) {
                // needed triangle found
                return 1.0f / cur_factor;
            }
        }
    }
    return 1.0f;
}

} // namespace rt
