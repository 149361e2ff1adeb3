// shade_lights.h -- next-event estimation: which light (a stochastic descent of the 8-wide light tree), then where on it.
//
// Two steps that the device runs as two kernels (kernels.hip.h):
//   pick_light      a pointer chase that only needs the shade position P and one random number -> (light, 1 / probability)
//   sample_light    per-kind sampling of the picked emitter -> direction, distance, density, radiance
// plus what the terminal cases of the shade stage need from the same data: the density with which next-event estimation
// WOULD have picked an emissive triangle that a BSDF-sampled ray hit (multiple importance sampling), the env-map importance
// quadtree, the lat-long RGBE lookup.
//
// Order-of-operations source (bit-exact parity with the oracle on the host build): reference internal/CoreRef.cpp --
// SampleLightSource :3264-3614, calc_lnode_importance :1004-1066 (+ :900-956), SampleSphericalRectangle / Triangle
// :1288-1416, EvalTriLightFactor :4692-4736, Evaluate/Sample_EnvQTree :4738-4839, SampleLatlong_RGBE :2995-3039,
// helpers :679-723, :1104-1129, :1274-1280.
#pragma once

#include "rt_rng.h"
#include "rt_texture.h"
#include "rt_types.h"

namespace rt {

// ---- what sampling a light yields ----------------------------------------------------------------------------------------
struct LightSample {
    f3 radiance;  // emitted colour towards the shade point (spot cone, portal / env lookups, textures applied)
    f3 dir;       // unit direction from the shade point to the sampled point
    f3 point;     // the sampled point, nudged off the emitter
    float area;   // > 0: BSDF sampling can hit this emitter as well, i.e. the estimate is MIS-weighted
    float reach;  // multiplier of the shadow-ray length (MAX_DIST for emitters at infinity)
    float pdf;    // solid-angle density, pick probability included; 0 = no sample
    bool casts_shadow, is_env;
    uint32_t ray_mask; // which ray types may be lit by it (RAY_TYPE_*_BIT)
};
RT_HD LightSample no_light_sample() {
    LightSample s;
    s.radiance = s.dir = s.point = f3{0.0f, 0.0f, 0.0f};
    s.area = 0.0f, s.reach = 1.0f, s.pdf = 0.0f;
    s.casts_shadow = false, s.is_env = false, s.ray_mask = 0;
    return s;
}

// ---- small geometry helpers ----------------------------------------------------------------------------------------------
// any tangent / bitangent pair for N (the reference's pick of the helper axis, CoreRef.cpp:679-689)
RT_HD void tangent_pair(const f3 n, f3 &t, f3 &b) {
    const f3 helper = (fabsf(n.y) < 0.999f) ? f3{0.0f, 1.0f, 0.0f} : f3{1.0f, 0.0f, 0.0f};
    t = normalize(cross(helper, n));
    b = cross(n, t);
}
// Shirley-Chiu concentric square -> disk, (radius, angle)
RT_HD bool concentric_polar(const f2 square, const bool half_pi_minus, float &r, float &theta) {
    if (fabsf(square.x) > fabsf(square.y)) {
        r = square.x;
        theta = 0.25f * PI * (square.y / square.x);
    } else {
        r = square.y;
        theta = half_pi_minus ? 0.5f * PI - 0.25f * PI * (square.x / square.y) : 0.5f * PI * (1.0f - 0.5f * (square.x / square.y));
    }
    return true;
}
// a direction inside the cone of half-width `radius` (at unit distance along `axis`'s length) around `axis` (CoreRef.cpp:691-714)
RT_HD f3 cone_direction(const float u1, const float u2, const f3 axis, const float radius) {
    const f2 square = {2.0f * u1 - 1.0f, 2.0f * u2 - 1.0f};
    if (square.x == 0.0f && square.y == 0.0f) {
        return axis;
    }
    float r, theta;
    concentric_polar(square, false, r, theta);
    const f2 sc = portable_sincos(theta);
    const f2 disk = {radius * r * sc.y, radius * r * sc.x};
    f3 t, b;
    tangent_pair(normalize(axis), t, b);
    return axis + disk.x * t + disk.y * b;
}
// nearer root of |o + t d - c| = radius (CoreRef.cpp:716-723)
RT_HD float sphere_entry_distance(const f3 centre, const float radius, const f3 o, const f3 d) {
    const f3 oc = o - centre;
    const float a = dot(d, d);
    const float b = 2 * dot(oc, d);
    const float c = dot(oc, oc) - radius * radius;
    const float disc = b * b - 4 * a * c;
    return (-b - sqrtf(fmaxf(disc, 0.0f))) / (2 * a);
}
RT_HD f3 gram_schmidt(const f3 a, const f3 b) { return normalize(b - dot(a, b) * a); }
// spherical interpolation from unit vector `from` towards `to` by the fraction `s` of their angle (CoreRef.cpp:1108-1129)
RT_HD f3 arc_lerp(const f3 from, const f3 to, const float s) {
    const float c = clampf(dot(from, to), -1.0f, 1.0f);
    const float theta = acosf(c) * s;
    const f3 side = safe_normalize(to - from * c);
    const f2 sc = portable_sincos(theta);
    return from * sc.y + side * sc.x;
}
// numerically safe angle between unit vectors (Kahan; CoreRef.cpp:1274-1280)
RT_HD float unit_angle(const f3 a, const f3 b) {
    if (dot(a, b) < 0) {
        return PI - 2 * portable_asinf(length(a + b) / 2);
    }
    return 2 * portable_asinf(length(b - a) / 2);
}

// ---- solid-angle sampling of a rectangle (Urena et al. 2013) -------------------------------------------------------------
// returns the density 1 / solid angle (0: too small, the caller falls back to area sampling); `out_point` may be null
// (density only).  CoreRef.cpp:1288-1354.
RT_HD float solid_angle_rect(const f3 P, const f3 centre, const f3 side_u, const f3 side_v, const f2 u, f3 *out_point) {
    const f3 corner = centre - 0.5f * side_u - 0.5f * side_v;
    float len_u, len_v;
    const f3 ex = normalize_len(side_u, len_u), ey = normalize_len(side_v, len_v);
    f3 ez = cross(ex, ey);
    // rectangle in the local frame, z pointing away from P
    const f3 to_corner = corner - P;
    float z0 = dot(to_corner, ez);
    if (z0 > 0.0f) {
        ez = -ez;
        z0 = -z0;
    }
    const float x0 = dot(to_corner, ex), y0 = dot(to_corner, ey);
    const float x1 = x0 + len_u, y1 = y0 + len_v;
    // normals of the four planes through P and the edges, then the internal angles between them
    const f4 d = mk4(x0, y1, x1, y0) - mk4(x1, y0, x0, y1);
    f4 n = mk4(y0, x1, y1, x0) * d;
    {
        const float zz = z0 * z0;
        const f4 len = {sqrtf(zz * d.x * d.x + n.x * n.x), sqrtf(zz * d.y * d.y + n.y * n.y), sqrtf(zz * d.z * d.z + n.z * n.z),
                        sqrtf(zz * d.w * d.w + n.w * n.w)};
        n = n / len;
    }
    const float g0 = portable_acosf(clampf(-n.x * n.y, -1.0f, 1.0f));
    const float g1 = portable_acosf(clampf(-n.y * n.z, -1.0f, 1.0f));
    const float g2 = portable_acosf(clampf(-n.z * n.w, -1.0f, 1.0f));
    const float g3 = portable_acosf(clampf(-n.w * n.x, -1.0f, 1.0f));
    const float b0 = n.x, b1 = n.z, b0_sq = b0 * b0;
    const float k = 2 * PI - g2 - g3;
    const float solid_angle = g0 + g1 - k;
    if (solid_angle <= SPHERICAL_AREA_THRESHOLD) {
        return 0.0f;
    }
    if (out_point) {
        const float au = u.x * solid_angle + k;
        const f2 sc = portable_sincos(au);
        const float fu = safe_div((sc.y * b0 - b1), sc.x);
        float cu = 1.0f / sqrtf(fu * fu + b0_sq) * (fu > 0.0f ? 1.0f : -1.0f);
        cu = clampf(cu, -1.0f, 1.0f);
        float xu = -(cu * z0) / fmaxf(sqrtf(1.0f - cu * cu), 1e-7f);
        xu = clampf(xu, x0, x1);
        const float z0_sq = z0 * z0, y0_sq = y0 * y0, y1_sq = y1 * y1;
        const float dd = sqrtf(xu * xu + z0_sq);
        const float h0 = y0 / sqrtf(dd * dd + y0_sq);
        const float h1 = y1 / sqrtf(dd * dd + y1_sq);
        const float hv = h0 + u.y * (h1 - h0), hv_sq = hv * hv;
        const float yv = (hv_sq < 1.0f - 1e-6f) ? (hv * dd) / sqrtf(1.0f - hv_sq) : y1;
        (*out_point) = P + xu * ex + yv * ey + z0 * ez;
    }
    return (1.0f / solid_angle);
}

// ---- solid-angle sampling of a triangle (Arvo 1995) ----------------------------------------------------------------------
// returns 1 / solid angle or 0 (too small); `out_dir` may be null (density only).  CoreRef.cpp:1356-1416.
RT_HD float solid_angle_triangle(const f3 P, const f3 p1, const f3 p2, const f3 p3, const f2 u, f3 *out_dir) {
    const f3 A = normalize(p1 - P), B = normalize(p2 - P), C = normalize(p3 - P);
    // internal angles: between the tangents of the two arcs meeting at each vertex
    const f3 tBA = gram_schmidt(A, B - A), tCA = gram_schmidt(A, C - A);
    const f3 tAB = gram_schmidt(B, A - B), tCB = gram_schmidt(B, C - B);
    const f3 tBC = gram_schmidt(C, B - C), tAC = gram_schmidt(C, A - C);
    const float alpha = unit_angle(tBA, tCA);
    const float beta = unit_angle(tAB, tCB);
    const float gamma = unit_angle(tBC, tAC);
    const float solid_angle = alpha + beta + gamma - PI;
    if (solid_angle <= SPHERICAL_AREA_THRESHOLD) {
        return 0.0f;
    }
    if (out_dir) {
        const float arc_b = portable_acosf(clampf(dot(C, A), -1.0f, 1.0f));
        const float arc_c = portable_acosf(clampf(dot(A, B), -1.0f, 1.0f));
        // sub-triangle of area u.x * solid_angle: its vertex C' on the arc AC
        const float sub_area = u.x * solid_angle;
        const f2 sc_delta = portable_sincos(sub_area - alpha);
        const float p = sc_delta.x, q = sc_delta.y;
        const f2 sc_alpha = portable_sincos(alpha);
        const float uu = q - sc_alpha.y;
        const float vv = p + sc_alpha.x * portable_cos(arc_c);
        const float denom = ((vv * p + uu * q) * sc_alpha.x);
        const float s = safe_div(1.0f, arc_b) * portable_acosf(clampf(safe_div(((vv * q - uu * p) * sc_alpha.y - vv), denom), -1.0f, 1.0f));
        const f3 C_sub = arc_lerp(A, C, s);
        // then a point on the arc B C' by u.y
        const float arc_bc = portable_acosf(clampf(dot(C_sub, B), -1.0f, 1.0f));
        const float t = safe_div(portable_acosf(clampf(1.0f - u.y * (1.0f - dot(C_sub, B)), -1.0f, 1.0f)), arc_bc);
        (*out_dir) = arc_lerp(B, C_sub, t);
    }
    return (1.0f / solid_angle);
}

// ---- light tree ----------------------------------------------------------------------------------------------------------
// child box of a quantised node; an "infinite" slot (directional / env emitters) has lo = 0xff, hi = 0 and decodes to
// +-MAX_DIST (CoreRef.cpp:1005-1022)
RT_HD void light_child_box(const rayhip_light_cwbvh_node &n, const int i, float lo[3], float hi[3]) {
    const float step[3] = {(n.bbox_max[0] - n.bbox_min[0]) / 255.0f, (n.bbox_max[1] - n.bbox_min[1]) / 255.0f,
                           (n.bbox_max[2] - n.bbox_min[2]) / 255.0f};
    lo[0] = lo[1] = lo[2] = -MAX_DIST;
    hi[0] = hi[1] = hi[2] = MAX_DIST;
    if (n.ch_bbox_min[0][i] != 0xff || n.ch_bbox_max[0][i] != 0) {
        for (int a = 0; a < 3; ++a) {
            lo[a] = n.bbox_min[a] + float(int(n.ch_bbox_min[a][i])) * step[a];
            hi[a] = n.bbox_min[a] + float(int(n.ch_bbox_max[a][i])) * step[a];
        }
    }
}

// The importance of a child (Conty & Kulla 2018: flux x cosine bound / distance^2) splits into a part that depends on the
// node only -- box centre and half-diagonal, the decoded emission-cone axis and its two cosines: 8 divisions and 3 square
// roots per child -- and a part that depends on the shade point.  The first is evaluated ONCE per scene into the
// `light_children` table (fill_light_children, run on the host at upload with these very functions: same IEEE operations,
// same bits), LIGHT_CHILDREN_STRIDE float4 per node (layout: fill_light_children).
struct LightChild {
    float4 axis_extent;  // emission-cone axis (unit), half-diagonal of the box
    float4 centre_valid; // box centre, 1 if the box is finite (else importance = flux)
    float4 cosines;      // cos(theta_o), sin(theta_o), cos(theta_e), flux
};
constexpr int LIGHT_CHILDREN_STRIDE = 26;

RT_HD LightChild decode_light_child(const rayhip_light_cwbvh_node &n, const int i) {
    float lo[3], hi[3];
    light_child_box(n, i, lo, hi);
    LightChild c;
    c.axis_extent = mkfloat4(0.0f, 0.0f, 0.0f, 0.0f);
    c.centre_valid = mkfloat4(0.0f, 0.0f, 0.0f, 0.0f);
    c.cosines = mkfloat4(0.0f, 0.0f, 0.0f, n.flux[i]);
    if (lo[0] > -MAX_DIST) {
        // octahedral cone axis, 2 x 16 bit (CoreRef.cpp:935-947)
        const uint32_t oct = n.axis[i];
        float ax = -1.0f + 2.0f * float((oct >> 16) & 0x0000ffff) / 65535.0f;
        float ay = -1.0f + 2.0f * float(oct & 0x0000ffff) / 65535.0f;
        float az = 1.0f - fabsf(ax) - fabsf(ay);
        if (az < 0.0f) {
            const float ax0 = ax;
            ax = (1.0f - fabsf(ay)) * copysignf(1.0f, ax0);
            ay = (1.0f - fabsf(ax0)) * copysignf(1.0f, ay);
        }
        const float len = sqrtf(ax * ax + ay * ay + az * az);
        ax = ax / len, ay = ay / len, az = az / len;
        const float e[3] = {hi[0] - lo[0], hi[1] - lo[1], hi[2] - lo[2]};
        const float half_diag = 0.5f * sqrtf(e[0] * e[0] + e[1] * e[1] + e[2] * e[2]);
        // two cosines, 2 x 16 bit (CoreRef.cpp:949-956)
        const uint32_t packed = n.cos_omega_ne[i];
        const float cos_o = 2.0f * (float((packed >> 16) & 0x0000ffff) / 65534.0f) - 1.0f;
        const float cos_e = 2.0f * (float(packed & 0x0000ffff) / 65534.0f) - 1.0f;
        const float sin_o = sqrtf(1.0f - cos_o * cos_o);
        c.axis_extent = mkfloat4(ax, ay, az, half_diag);
        c.centre_valid = mkfloat4(0.5f * (lo[0] + hi[0]), 0.5f * (lo[1] + hi[1]), 0.5f * (lo[2] + hi[2]), 1.0f);
        c.cosines = mkfloat4(cos_o, sin_o, cos_e, n.flux[i]);
    }
    return c;
}
// Rows [0..1]: the eight child links (what the descent follows: no second table in the loop); rows [2 + 8 r + i]: row r (axis +
// extent, centre + valid, cosines + flux) of child i -- row-type major, so that eight lanes fetching the same row of the eight
// children of a node (k_light_pick: one child per lane) read 128 consecutive bytes.
RT_HD void fill_light_children(const rayhip_light_cwbvh_node &n, float4 *out /* [LIGHT_CHILDREN_STRIDE] */) {
    out[0] = mkfloat4(uint_as_float(n.child[0]), uint_as_float(n.child[1]), uint_as_float(n.child[2]), uint_as_float(n.child[3]));
    out[1] = mkfloat4(uint_as_float(n.child[4]), uint_as_float(n.child[5]), uint_as_float(n.child[6]), uint_as_float(n.child[7]));
    for (int i = 0; i < 8; ++i) {
        const LightChild c = decode_light_child(n, i);
        out[2 + 0 * 8 + i] = c.axis_extent, out[2 + 1 * 8 + i] = c.centre_valid, out[2 + 2 * 8 + i] = c.cosines;
    }
}
// the same two accessors on the rows of ONE node wherever they are (k_light_pick_first keeps the top of the table in LDS)
RT_HD LightChild load_light_child_rows(const float4 *node_rows, const int i) {
    const float4 *t = node_rows + 2;
    LightChild c;
    c.axis_extent = t[0 * 8 + i], c.centre_valid = t[1 * 8 + i], c.cosines = t[2 * 8 + i];
    return c;
}
RT_HD uint32_t light_child_link_rows(const float4 *node_rows, const int i) { return reinterpret_cast<const uint32_t *>(node_rows)[i]; }
RT_HD LightChild load_light_child(const SceneView &sc, const uint32_t node, const int i) {
    const float4 *t = sc.light_children + size_t(node) * LIGHT_CHILDREN_STRIDE + 2;
    LightChild c;
    c.axis_extent = t[0 * 8 + i], c.centre_valid = t[1 * 8 + i], c.cosines = t[2 * 8 + i];
    return c;
}
RT_HD uint32_t light_child_link(const SceneView &sc, const uint32_t node, const int i) {
    return reinterpret_cast<const uint32_t *>(sc.light_children + size_t(node) * LIGHT_CHILDREN_STRIDE)[i];
}

// Division and square root of the importance heuristic.  A child's importance only steers WHICH light is sampled and with
// what probability -- the estimator divides by that same probability -- so on the device they are the hardware reciprocal
// and square root (1 ulp, one instruction instead of 10-12; five divisions and four roots per child and level).  The host
// build keeps the IEEE operations and stays bit-exact with the oracle; -DRT_EXACT_IMPORTANCE restores them on the device
// (tests/test_gpu_parity.py measures the difference: pick probabilities move by ~1e-7 relative).
#if defined(__HIP_DEVICE_COMPILE__) && !defined(RT_EXACT_IMPORTANCE)
RT_HD float steer_div(const float a, const float b) { return a * __builtin_amdgcn_rcpf(b); }
RT_HD float steer_sqrt(const float a) { return __builtin_amdgcn_sqrtf(a); }
#else
RT_HD float steer_div(const float a, const float b) { return a / b; }
RT_HD float steer_sqrt(const float a) { return sqrtf(a); }
#endif

// cos / sin of max(a - b, 0) from the sines and cosines of a and b
RT_HD float cos_of_clamped_difference(float sin_a, float cos_a, float sin_b, float cos_b) {
    return (cos_a > cos_b) ? 1.0f : (cos_a * cos_b + sin_a * sin_b);
}
RT_HD float sin_of_clamped_difference(float sin_a, float cos_a, float sin_b, float cos_b) {
    return (cos_a > cos_b) ? 0.0f : (sin_a * cos_b - cos_a * sin_b);
}
// importance of one child as seen from P: flux * cos(max(theta_w - theta_o - theta_b, 0)) / max(d^2, half_diag), zero outside
// the emission cone widened by theta_e (CoreRef.cpp:1024-1066)
// (written without an early exit: an empty or box-less child yields its flux through the final select, the arithmetic above it
// runs on the zero rows of its table entry and is discarded.  That keeps the eight evaluations of a node free of control
// flow, so the compiler can issue the table loads of all children before the first use -- with the branch the descent
// paid one memory round trip per child, 8 in a row per level, and k_light_pick sat 75 % of its time in s_waitcnt.)
RT_HD float light_child_importance(const LightChild &c, const f3 P) {
    const float flux = c.cosines.w;
    const float half_diag = c.axis_extent.w;
    float w[3] = {P.x - c.centre_valid.x, P.y - c.centre_valid.y, P.z - c.centre_valid.z};
    const float d2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
    const float d = steer_sqrt(d2);
    w[0] = steer_div(w[0], d), w[1] = steer_div(w[1], d), w[2] = steer_div(w[2], d);
    const float falloff_d2 = sse_max(d2, half_diag);
    const float cos_w = c.axis_extent.x * w[0] + c.axis_extent.y * w[1] + c.axis_extent.z * w[2];
    const float sin_w = steer_sqrt(sse_max(1.0f - cos_w * cos_w, 0.0f));
    // angle under which the box is seen (inside the bounding sphere: everything)
    float cos_b = steer_sqrt(sse_max(1.0f - steer_div(half_diag * half_diag, d2), 0.0f));
    cos_b = (d2 < half_diag * half_diag) ? -1.0f : cos_b;
    const float sin_b = steer_sqrt(1.0f - cos_b * cos_b);
    const float cos_o = c.cosines.x, sin_o = c.cosines.y, cos_e = c.cosines.z;
    const float cos_x = cos_of_clamped_difference(sin_w, cos_w, sin_o, cos_o);
    const float sin_x = sin_of_clamped_difference(sin_w, cos_w, sin_o, cos_o);
    const float cos_min = cos_of_clamped_difference(sin_x, cos_x, sin_b, cos_b);
    const float geometric = (cos_min > cos_e) ? steer_div(cos_min, falloff_d2) : 0.0f;
    return (c.centre_valid.w != 0.0f && flux != 0.0f) ? flux * geometric : flux;
}
// the eight importances of a node (their sum is taken in the oracle's SSE association order, sum8_sse_order)
RT_HD void light_node_importances(const SceneView &sc, const uint32_t node, const f3 P, float imp[8]) {
    LightChild c[8];
    for (int i = 0; i < 8; ++i) {
        c[i] = load_light_child(sc, node, i);
    }
    for (int i = 0; i < 8; ++i) {
        imp[i] = light_child_importance(c[i], P);
    }
}
RT_HD void light_node_importances_rows(const float4 *node_rows, const f3 P, float imp[8]) {
    LightChild c[8];
    for (int i = 0; i < 8; ++i) {
        c[i] = load_light_child_rows(node_rows, i);
    }
    for (int i = 0; i < 8; ++i) {
        imp[i] = light_child_importance(c[i], P);
    }
}
RT_HD float sum8_sse_order(const float v[8]) { return (((v[0] + v[4]) + (v[1] + v[5])) + (v[2] + v[6])) + (v[3] + v[7]); }

// ---- step 1: pick a light -------------------------------------------------------------------------------------------------
struct LightPick {
    uint32_t light;   // index into the light array
    float inv_prob;   // 1 / probability of this pick; 0 when no light can reach P
    float u_left;     // what is left of the random number after the descent (the env quadtree continues with it)
};
// Descend from the root: at every node the children's importances form a discrete distribution, the running random number
// selects a child and is re-stretched to [0, 1) inside its interval (CoreRef.cpp:3273-3312).  Everything is kept in
// registers: the selection is a chain of selects, not an indexed local array.
// one level of the descent, given the eight importances: the child the running random number selects (u is re-stretched inside
// its interval, prob multiplied by its share); false when nothing below this node can light the point
RT_HD bool light_level_choice(const float imp[8], float &u, float &prob, int &chosen) {
    const float total = sum8_sse_order(imp);
    if (total == 0.0f) {
        return false;
    }
    float share[8], upper[9];
    upper[0] = 0.0f;
    for (int j = 0; j < 8; ++j) {
        share[j] = steer_div(imp[j], total);
        upper[j + 1] = upper[j] + share[j];
    }
    for (int j = 0; j < 8; ++j) { // the trailing entries that already equal the total become 1.01: u < 1 never passes them
        if (upper[j + 1] == upper[8]) {
            upper[j + 1] = 1.01f;
        }
    }
    chosen = 0;
    for (int j = 1; j < 9; ++j) {
        chosen += (upper[j] <= u) ? 1 : 0;
    }
    float lower = upper[0], width = share[0];
    for (int j = 1; j < 8; ++j) {
        lower = (chosen == j) ? upper[j] : lower;
        width = (chosen == j) ? share[j] : width;
    }
    u = fractf((u - lower) / width);
    prob *= width;
    return true;
}
RT_HD LightPick pick_light(const SceneView &sc, const f3 P, float u) {
    LightPick pick;
    pick.light = 0, pick.inv_prob = 0.0f, pick.u_left = u;
    float prob = 1.0f;
    uint32_t cur = 0;
    while ((cur & LEAF_NODE_BIT) == 0) {
        float imp[8];
        light_node_importances(sc, cur, P, imp);
        int chosen;
        if (!light_level_choice(imp, u, prob, chosen)) {
            return pick; // nothing in this subtree can light P
        }
        cur = light_child_link(sc, cur, chosen);
    }
    pick.light = (cur & PRIM_INDEX_BITS);
    pick.inv_prob = 1.0f / prob;
    pick.u_left = u;
    return pick;
}

// `light_tri_geom` table: world-space corners + uvs of every triangle emitter, resolved once per scene (the reference walks
// light -> instance transform -> index triple -> three 44-byte vertices per sample, CoreRef.cpp:3530-3545).  Four float4 per
// light: (p1, uv1.x) (p2, uv1.y) (p3, uv2.x) (uv2.y, uv3.x, uv3.y, -); filled on the host by this function.
RT_HD void fill_light_tri_geom(const rayhip_light &l, const rayhip_mesh_instance *instances, const uint32_t *vtx_indices,
                               const rayhip_vertex *vertices, float4 *out /* [4] */) {
    out[0] = out[1] = out[2] = out[3] = mkfloat4(0.0f, 0.0f, 0.0f, 0.0f);
    if (light_type(l) != LIGHT_TYPE_TRI) {
        return;
    }
    const uint32_t tri = float_as_uint(l.params[0]);
    const rayhip_mesh_instance &inst = instances[float_as_uint(l.params[1])];
    const rayhip_vertex &a = vertices[vtx_indices[tri * 3 + 0]], &b = vertices[vtx_indices[tri * 3 + 1]], &c = vertices[vtx_indices[tri * 3 + 2]];
    const f3 p1 = transform_point(mk3(a.p), inst.xform), p2 = transform_point(mk3(b.p), inst.xform), p3 = transform_point(mk3(c.p), inst.xform);
    out[0] = mkfloat4(p1.x, p1.y, p1.z, a.t[0]);
    out[1] = mkfloat4(p2.x, p2.y, p2.z, a.t[1]);
    out[2] = mkfloat4(p3.x, p3.y, p3.z, b.t[0]);
    out[3] = mkfloat4(b.t[1], c.t[0], c.t[1], 0.0f);
}

// ---- environment: lat-long RGBE lookup, importance quadtree ----------------------------------------------------------------------
RT_HD float wrap_two_pi(float phi) {
    if (phi < 0) {
        phi += 2 * PI;
    }
    if (phi > 2 * PI) {
        phi -= 2 * PI;
    }
    return phi;
}
// unit-square point <-> direction of the equal-area cylindrical map (Core.cpp:110-141; libm sinf / cosf / atan2f there)
RT_HD f3 square_to_direction(const f2 p, const float y_rotation) {
    const float cos_theta = 2 * p.x - 1;
    const float phi = wrap_two_pi(2 * PI * p.y + y_rotation);
    const float sin_theta = sqrtf(1 - cos_theta * cos_theta);
    const float sin_phi = sinf(phi), cos_phi = cosf(phi);
    return f3{sin_theta * cos_phi, cos_theta, -sin_theta * sin_phi};
}
RT_HD f2 direction_to_square(const f3 d, const float y_rotation) {
    const float cos_theta = fminf(fmaxf(d.y, -1.0f), 1.0f);
    const float phi = wrap_two_pi(-atan2f(d.z, d.x) + y_rotation);
    return f2{(cos_theta + 1.0f) / 2.0f, phi / (2.0f * PI)};
}
RT_HD float quad_cell(const float4 q, const int i) { return i == 0 ? q.x : (i == 1 ? q.y : (i == 2 ? q.z : q.w)); }
RT_HD float4 env_quad(const SceneView &sc, const int lod, const int res, const int qx, const int qy) {
    return sc.env_qtree[sc.env_qtree_offset[lod] + uint32_t(qy * res / 2 + qx)];
}
// density of direction `dir` under the quadtree: the product of the cell shares on the way down (CoreRef.cpp:4738-4771)
RT_HD float env_quadtree_pdf(const SceneView &sc, const float y_rotation, const f3 dir) {
    const f2 p = direction_to_square(dir, -y_rotation);
    float share = 1.0f;
    int res = 2;
    for (int lod = sc.env.qtree_levels - 1; lod >= 0; --lod, res *= 2) {
        const int x = clampi(int(p.x * float(res)), 0, res - 1), y = clampi(int(p.y * float(res)), 0, res - 1);
        const int cell = (x & 1) | ((y & 1) << 1);
        const float4 quad = env_quad(sc, lod, res, x / 2, y / 2);
        const float total = quad.x + quad.y + quad.z + quad.w;
        if (total <= 0.0f) {
            break;
        }
        share *= 4.0f * quad_cell(quad, cell) / total;
    }
    return share / (4.0f * PI);
}
// one binary decision of the descent: `u` against the share of the first half; `u` is rescaled into the half it fell in
RT_HD bool second_half(float &u, const float first_share) {
    if (u < first_share) {
        u /= first_share;
        return false;
    }
    u = (u - first_share) / (1.0f - first_share);
    return true;
}
// draws a direction: `u` walks down the tree -- per level first the column of the quad, then the row inside that column --, (jx, jy)
// jitter inside the final cell; returns (direction, density).  The quotients are the reference's (CoreRef.cpp:4773-4839: column share
// of the quad, then the upper cell's share of the column), so the same u lands in the same cell bit for bit.
RT_HD f4 env_quadtree_draw(const SceneView &sc, const float y_rotation, float u, const float jx, const float jy) {
    int res = 2;
    float cell_size = 1.0f / float(res);
    f2 corner = {0.0f, 0.0f};
    float share = 1.0f;
    for (int lod = sc.env.qtree_levels - 1; lod >= 0; --lod, res *= 2, cell_size *= 0.5f) {
        const float4 quad = env_quad(sc, lod, res, int(corner.x * float(res)) / 2, int(corner.y * float(res)) / 2);
        const float left = quad.x + quad.z, total = left + quad.y + quad.w;
        if (total <= 0.0f) {
            break;
        }
        const bool right = second_half(u, left / total);
        const bool lower = second_half(u, right ? quad.y / (total - left) : quad.x / left);
        if (right) {
            corner.x = corner.x + cell_size;
        }
        if (lower) {
            corner.y = corner.y + cell_size;
        }
        share *= 4.0f * quad_cell(quad, int(right) | (int(lower) << 1)) / total;
    }
    corner.x += 2 * cell_size * jx;
    corner.y += 2 * cell_size * jy;
    const f3 dir = square_to_direction(corner, y_rotation);
    return f4{dir.x, dir.y, dir.z, share / (4.0f * PI)};
}
// one stochastic tap of a lat-long RGBE map in direction `dir` (CoreRef.cpp:2995-3039)
RT_HD f3 latlong_rgbe(const SceneView &sc, const uint32_t handle, const f3 dir, const float y_rotation, const f2 jitter) {
    const float theta = acosf(clampf(dir.y, -1.0f, 1.0f)) / PI;
    const float phi = wrap_two_pi(atan2f(dir.z, dir.x) + y_rotation);
    const float u = fractf(0.5f * phi / PI);
    const rayhip_texture &t = sc.textures[sc.tex_table[0] + (handle & 0x00ffffffu)];
    f2 texel = {u * float(t.width[0]), theta * float(t.height[0])};
    texel = texel + jitter;
    int ix = int(texel.x), iy = int(texel.y);
    ix %= int(t.width[0]);
    iy %= int(t.height[0]);
    const uint32_t px = sc.texels[t.offset[0] + uint32_t(iy) * t.width[0] + uint32_t(ix)];
    // shared-exponent decode: mantissas through the exact u8 -> [0, 1] bit trick, times 2^(e - 128) (CoreRef.h:229-232, Core.h:411-418)
    const float scale = exp2f(float((px >> 24) & 0xffu) - 128.0f);
    const uint32_t m0 = px & 0xffu, m1 = (px >> 8) & 0xffu, m2 = (px >> 16) & 0xffu;
    return f3{(uint_as_float(0x3f800000u + m0 * 0x8080u + (m0 + 1) / 2) - 1.0f) * scale,
              (uint_as_float(0x3f800000u + m1 * 0x8080u + (m1 + 1) / 2) - 1.0f) * scale,
              (uint_as_float(0x3f800000u + m2 * 0x8080u + (m2 + 1) / 2) - 1.0f) * scale};
}
// environment radiance seen through a sky portal / by an env sample
RT_HD f3 env_radiance_towards(const SceneView &sc, const f3 dir, const f2 jitter) {
    f3 c = mk3(sc.env.env_col);
    if (sc.env.env_map != 0xffffffff) {
        c *= latlong_rgbe(sc, sc.env.env_map, dir, sc.env.env_map_rotation, jitter);
    }
    return c;
}

// ---- step 2: sample the picked light ----------------------------------------------------------------------------------------
// One function per emitter kind; each fills dir / point / pdf / area (pdf stays 0 when the emitter cannot light P).
// T, B, N: the shading frame (hemisphere sampling of an unstructured environment only).

RT_HD void sample_sphere_light(const rayhip_light &l, const f3 P, const f2 u, LightSample &s) {
    const float radius = l.params[7];
    const f3 centre = mk3(&l.params[0]);
    float d;
    const f3 towards = normalize_len(centre - P, d);
    if (!(d > radius)) {
        return; // inside the emitter
    }
    // the sphere as the disk it subtends
    const float tangent_len = sqrtf(d * d - radius * radius);
    const float disk_radius = (tangent_len * radius) / d;
    float disk_dist = radius > 0.0f ? ((tangent_len * disk_radius) / radius) : d;
    const f3 dir = normalize_len(cone_direction(u.x, u.y, disk_dist * towards, disk_radius), disk_dist);
    if (radius > 0.0f) {
        const float t = sphere_entry_distance(centre, radius, P, dir);
        const f3 on_sphere = P + dir * t;
        const f3 outward = normalize(on_sphere - centre);
        const float disk_area = PI * disk_radius * disk_radius;
        const float cos_theta = dot(dir, towards);
        s.point = offset_ray(on_sphere, outward);
        s.pdf = (disk_dist * disk_dist) / (disk_area * cos_theta);
    } else {
        s.point = centre;
        s.pdf = (disk_dist * disk_dist) / PI;
    }
    s.dir = dir;
    s.area = PI * disk_radius * disk_radius;
    s.ray_mask = light_ray_visibility(l);
    if (!light_visible(l)) {
        s.area = 0.0f;
    }
    const float spot = l.params[8], blend = l.params[9];
    if (spot > 0.0f) {
        const float along = -dot(s.dir, mk3(&l.params[4]));
        if (along > 0.0f) {
            const float angle = acosf(saturatef(along));
            s.radiance *= saturatef((spot - angle) / blend);
        } else {
            s.radiance *= 0.0f;
        }
    }
}

RT_HD void sample_directional_light(const rayhip_light &l, const f3 P, const f2 u, LightSample &s) {
    const f3 axis = mk3(&l.params[0]);
    const float tan_half_angle = l.params[4];
    s.dir = axis;
    s.area = 0.0f;
    s.pdf = 1.0f;
    if (tan_half_angle != 0.0f) {
        s.dir = normalize(cone_direction(u.x, u.y, s.dir, tan_half_angle));
        s.area = PI * tan_half_angle * tan_half_angle;
        const float cos_theta = dot(s.dir, axis);
        s.pdf = 1.0f / (s.area * cos_theta);
    }
    s.point = P + s.dir;
    s.reach = MAX_DIST;
    s.ray_mask = light_ray_visibility(l);
    if (!light_visible(l)) {
        s.area = 0.0f;
    }
}

template <class Jitter>
RT_HD void sample_rect_light(const SceneView &sc, const rayhip_light &l, const f3 P, const f2 u, Jitter &&tex_jitter, LightSample &s) {
    const f3 centre = mk3(&l.params[0]);
    const f3 side_u = mk3(&l.params[4]), side_v = mk3(&l.params[8]);
    const f3 facing = normalize(cross(side_u, side_v));
    const float rect_area = l.params[3];
    f3 point = {0.0f, 0.0f, 0.0f};
    const float sa_pdf = solid_angle_rect(P, centre, side_u, side_v, u, &point);
    if (sa_pdf <= 0.0f) { // solid angle too small: uniform on the area
        point = centre + side_u * (u.x - 0.5f) + side_v * (u.y - 0.5f);
    }
    float dist;
    s.dir = normalize_len(point - P, dist);
    s.ray_mask = light_ray_visibility(l);
    const float cos_theta = dot(-s.dir, facing);
    if (cos_theta > 0.0f) {
        s.point = offset_ray(point, facing);
        s.pdf = (sa_pdf > 0.0f) ? sa_pdf : (dist * dist) / (rect_area * cos_theta);
        s.area = light_visible(l) ? rect_area : 0.0f;
        if (light_sky_portal(l)) {
            s.radiance *= env_radiance_towards(sc, s.dir, tex_jitter());
            s.is_env = true;
        }
    }
}

template <class Jitter>
RT_HD void sample_disk_light(const SceneView &sc, const rayhip_light &l, const f3 P, const f2 u, Jitter &&tex_jitter, LightSample &s) {
    const f3 centre = mk3(&l.params[0]);
    const f3 side_u = mk3(&l.params[4]), side_v = mk3(&l.params[8]);
    f2 disk = {2.0f * u.x - 1.0f, 2.0f * u.y - 1.0f};
    if (disk.x != 0.0f && disk.y != 0.0f) {
        float r, theta;
        concentric_polar(disk, true, r, theta);
        const f2 sc_t = portable_sincos(theta);
        disk.x = 0.5f * r * sc_t.y;
        disk.y = 0.5f * r * sc_t.x;
    }
    const f3 point = centre + side_u * disk.x + side_v * disk.y;
    const f3 facing = normalize(cross(side_u, side_v));
    s.point = offset_ray(point, facing);
    float dist;
    s.dir = normalize_len(point - P, dist);
    s.area = l.params[3];
    s.ray_mask = light_ray_visibility(l);
    const float cos_theta = dot(-s.dir, facing);
    if (cos_theta > 0.0f) {
        s.pdf = (dist * dist) / (s.area * cos_theta);
    }
    if (!light_visible(l)) {
        s.area = 0.0f;
    }
    if (light_sky_portal(l)) {
        s.radiance *= env_radiance_towards(sc, s.dir, tex_jitter());
        s.is_env = true;
    }
}

RT_HD void sample_line_light(const rayhip_light &l, const f3 P, const f2 u, LightSample &s) {
    const f3 centre = mk3(&l.params[0]);
    const f3 axis = mk3(&l.params[8]);
    const float radius = l.params[7], height = l.params[11];
    // a point on the half of the cylinder that faces P
    const f3 from_axis = P - centre;
    const f3 side = normalize(cross(from_axis, axis));
    const f3 front = cross(side, axis);
    const f2 sc = portable_sincos(PI * u.x);
    const f3 radial = sc.y * side + sc.x * front;
    const f3 point = centre + radial * radius + (u.y - 0.5f) * axis * height;
    s.point = point;
    float dist;
    s.dir = normalize_len(point - P, dist);
    s.area = l.params[3];
    s.ray_mask = light_ray_visibility(l);
    const float cos_theta = 1.0f - fabsf(dot(s.dir, axis));
    if (cos_theta != 0.0f) {
        s.pdf = (dist * dist) / (s.area * cos_theta);
    }
    if (!light_visible(l)) {
        s.area = 0.0f;
    }
}

template <class Jitter>
RT_HD void sample_triangle_light(const SceneView &sc, const rayhip_light &l, const uint32_t light_index, const f3 P, const f2 u,
                                 Jitter &&tex_jitter, LightSample &s) {
    const uint32_t texture = float_as_uint(l.params[2]);
    const float4 *geom = sc.light_tri_geom + size_t(light_index) * 4;
    const float4 g0 = geom[0], g1 = geom[1], g2 = geom[2], g3 = geom[3];
    const f3 p1 = {g0.x, g0.y, g0.z}, p2 = {g1.x, g1.y, g1.z}, p3 = {g2.x, g2.y, g2.z};
    const f2 uv1 = mk2(g0.w, g1.w), uv2 = mk2(g2.w, g3.x), uv3 = mk2(g3.y, g3.z);
    const f3 e1 = p2 - p1, e2 = p3 - p1;
    float twice_area;
    const f3 facing = normalize_len(cross(e1, e2), twice_area);
    s.area = 0.5f * twice_area;
    s.ray_mask = light_ray_visibility(l);

    f3 point;
    f2 uv;
    float pdf = solid_angle_triangle(P, p1, p2, p3, u, &s.dir);
    if (pdf > 0.0f) {
        // where the drawn direction meets the triangle's plane (Moller-Trumbore without the range checks)
        const f3 pv = cross(s.dir, e2);
        const f3 tv = P - p1, qv = cross(tv, e1);
        const float inv_det = 1.0f / dot(e1, pv);
        const float bu = dot(tv, pv) * inv_det, bv = dot(s.dir, qv) * inv_det;
        point = (1.0f - bu - bv) * p1 + bu * p2 + bv * p3;
        uv = (1.0f - bu - bv) * uv1 + bu * uv2 + bv * uv3;
    } else { // solid angle too small: uniform on the area
        const float r1 = sqrtf(u.x), r2 = u.y;
        uv = uv1 * (1.0f - r1) + r1 * (uv2 * (1.0f - r2) + uv3 * r2);
        point = p1 * (1.0f - r1) + r1 * (p2 * (1.0f - r2) + p3 * r2);
        float dist;
        s.dir = normalize_len(point - P, dist);
        const float cos_area = -dot(s.dir, facing);
        pdf = safe_div_pos(dist * dist, s.area * cos_area);
    }
    float cos_theta = -dot(s.dir, facing);
    s.point = offset_ray(point, cos_theta >= 0.0f ? facing : -facing);
    if (light_doublesided(l)) {
        cos_theta = fabsf(cos_theta);
    }
    if (cos_theta > 0.0f) {
        s.pdf = pdf;
        if (texture != 0xffffffff) {
            s.radiance *= xyz(sample_color(sc, texture, uv, 0, tex_jitter()));
        }
    }
}

template <class Jitter>
RT_HD void sample_env_light(const SceneView &sc, const rayhip_light &l, const f3 P, const f3 T, const f3 B, const f3 N, const float u_tree,
                            const f2 u, Jitter &&tex_jitter, LightSample &s) {
    float pdf;
    if (sc.env.qtree_levels) {
        const f4 d = env_quadtree_draw(sc, sc.env.env_map_rotation, u_tree, u.x, u.y);
        s.dir = f3{d.x, d.y, d.z};
        pdf = d.w;
    } else { // no importance map: uniform over the hemisphere of the shading normal
        const f2 sc_phi = portable_sincos(2 * PI * u.y);
        const float r = sqrtf(1.0f - u.x * u.x);
        s.dir = world_from_tangent(T, B, N, f3{r * sc_phi.y, r * sc_phi.x, u.x});
        pdf = 0.5f / PI;
    }
    s.radiance *= mk3(sc.env.env_col);
    if (sc.env.env_map != 0xffffffff) {
        s.radiance *= latlong_rgbe(sc, sc.env.env_map, s.dir, sc.env.env_map_rotation, tex_jitter());
    }
    s.area = 1.0f;
    s.point = P + s.dir;
    s.reach = MAX_DIST;
    s.pdf = pdf;
    s.is_env = true;
    s.ray_mask = light_ray_visibility(l);
}

// `tex_jitter()`: the texture-lookup random pair of this path vertex, asked for only by emitters that look something up
// (textured triangles, sky portals, the environment)
template <class Jitter>
RT_HD LightSample sample_light(const SceneView &sc, const LightPick &pick, const f3 P, const f3 T, const f3 B, const f3 N, const f2 u,
                               Jitter &&tex_jitter) {
    LightSample s = no_light_sample();
    if (pick.inv_prob == 0.0f) {
        return s;
    }
    RT_PROF_SHADE_LANES(2)
    const rayhip_light &l = sc.lights[pick.light];
    s.radiance = mk3(l.col);
    s.casts_shadow = light_cast_shadow(l);
    switch (light_type(l)) {
    case LIGHT_TYPE_SPHERE:
        sample_sphere_light(l, P, u, s);
        break;
    case LIGHT_TYPE_DIR:
        sample_directional_light(l, P, u, s);
        break;
    case LIGHT_TYPE_RECT:
        sample_rect_light(sc, l, P, u, tex_jitter, s);
        break;
    case LIGHT_TYPE_DISK:
        sample_disk_light(sc, l, P, u, tex_jitter, s);
        break;
    case LIGHT_TYPE_LINE:
        sample_line_light(l, P, u, s);
        break;
    case LIGHT_TYPE_TRI:
        RT_PROF_SHADE_LANES(4)
        sample_triangle_light(sc, l, pick.light, P, u, tex_jitter, s);
        break;
    case LIGHT_TYPE_ENV:
        RT_PROF_SHADE_LANES(6)
        sample_env_light(sc, l, P, T, B, N, pick.u_left, u, tex_jitter, s);
        break;
    default:
        break;
    }
    s.pdf /= pick.inv_prob;
    return s;
}

// ---- the reverse question: with what probability would the tree have picked triangle emitter `tri` for a path at `origin`
// that has just hit it at P?  Walks every branch whose box contains P (CoreRef.cpp:4692-4736, :246-278); returns 1 / probability.
RT_HD uint32_t children_containing(const rayhip_light_cwbvh_node &n, const f3 p) {
    uint32_t mask = 0;
    for (int i = 0; i < 8; ++i) {
        float lo[3], hi[3];
        light_child_box(n, i, lo, hi);
        const bool inside = (lo[0] <= p.x) && (lo[1] <= p.y) && (lo[2] <= p.z) && (hi[0] >= p.x) && (hi[1] >= p.y) && (hi[2] >= p.z);
        mask |= (inside ? 1u : 0u) << i;
    }
    return mask;
}
RT_HD float triangle_light_inv_pick_prob(const SceneView &sc, const f3 P, const f3 origin, const uint32_t tri) {
    uint32_t todo[MAX_STACK_SIZE];
    float todo_prob[MAX_STACK_SIZE];
    uint32_t n_todo = 0;
    todo_prob[n_todo] = 1.0f;
    todo[n_todo++] = 0;
    while (n_todo) {
        const uint32_t cur = todo[--n_todo];
        const float prob = todo_prob[n_todo];
        if ((cur & LEAF_NODE_BIT) != 0) {
            const rayhip_light &l = sc.lights[cur & PRIM_INDEX_BITS];
            if (light_type(l) == LIGHT_TYPE_TRI && float_as_uint(l.params[0]) == tri) {
                return 1.0f / prob;
            }
            continue;
        }
        const rayhip_light_cwbvh_node &node = sc.light_cwnodes[cur];
        const uint32_t mask = children_containing(node, P);
        if (mask == 0) {
            continue;
        }
        float imp[8];
        light_node_importances(sc, cur, origin, imp);
        const float total = sum8_sse_order(imp);
        if (total == 0.0f) {
            continue;
        }
        for (int i = 0; i < 8; ++i) {
            if (((mask >> i) & 1u) && imp[i] > 0.0f) {
                todo_prob[n_todo] = steer_div(prob * imp[i], total);
                todo[n_todo++] = node.child[i];
            }
        }
    }
    return 1.0f;
}

} // namespace rt
