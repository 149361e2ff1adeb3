// shade_math.h -- the scattering models of the shade stage as stand-alone lobes: Fresnel, GGX microfacet reflection and
// refraction with visible-normal sampling, the GTR1 clearcoat, Oren-Nayar and the Disney diffuse + sheen term.
//
// Every lobe is a pair  eval(wo, h, wi) -> (rgb, pdf)  and  draw(u) -> wi  in the TANGENT frame of the shade point
// (z = shading normal); the caller (shade_lobes.h) builds that frame once per shade point and shares it between the
// next-event estimate and the continuation ray.  The formulas are the published ones (Walter 2007, Heitz 2018, Dupuy &
// Benyoub 2023 for the spherical-cap / bounded VNDF, Burley 2012); what is pinned to the oracle is the ORDER of the fp32
// operations, because parity with RendererRef is checked bit for bit on the host build.  Order-of-operations source for
// each function: reference internal/ShadeRef.cpp, line ranges in the comments.
#pragma once

#include "rt_rng.h"
#include "rt_types.h"

namespace rt {

struct LobeValue { // what evaluating a lobe yields
    f3 f;          // bsdf * cosine, rgb
    float pdf;     // solid-angle density with which draw() would have produced this direction
};
RT_HD LobeValue no_lobe_value() { return LobeValue{f3{0.0f, 0.0f, 0.0f}, 0.0f}; }

RT_HD float pow5(const float v) { return (v * v) * (v * v) * v; }
// Schlick's (1 - u)^5 weight with u clamped to [0, 1]
RT_HD float schlick5(const float u) { return pow5(saturatef(1.0f - u)); }
RT_HD f3 mirror(const f3 wi, const f3 n, const float n_dot_wi) { return wi - 2 * n_dot_wi * n; }

// ---- roughness -> GGX alpha -------------------------------------------------------------------------------------------
// alpha = roughness^2, stretched by the anisotropy aspect sqrt(1 - 0.9 a); alphas below the path-regularisation floor are
// doubled and clamped into [floor / 4, floor] (ShadeRef.cpp:12-20; std::min / std::max argument order matters for NaN only)
RT_HD float regularised_alpha(const float a, const float floor_) {
    if (!(a < floor_)) {
        return a;
    }
    const float twice = 2 * a, lo = 0.25f * floor_;
    const float raised = (twice < lo) ? lo : twice;
    return (floor_ < raised) ? floor_ : raised;
}
RT_HD f2 ggx_alpha(const float roughness, const float anisotropy, const float floor_) {
    const float r2 = sqr(roughness);
    const float aspect = sqrtf(1.0f - 0.9f * anisotropy);
    return f2{regularised_alpha(r2 / aspect, floor_), regularised_alpha(r2 * aspect, floor_)};
}
// below this the lobe is treated as a perfect mirror / is skipped by next-event estimation
RT_HD bool alpha_is_singular(const f2 alpha) { return alpha.x * alpha.y < 1e-7f; }

// ---- Fresnel ---------------------------------------------------------------------------------------------------------
// unpolarised dielectric reflectance from the cosine of incidence alone (no refracted direction): with g^2 = eta^2 - 1 + c^2,
// F = 1/2 ((g - c) / (g + c))^2 (1 + ((c (g + c) - 1) / (c (g - c) + 1))^2); g^2 <= 0 is total internal reflection
// (ShadeRef.cpp:54-71)
RT_HD float fresnel_dielectric(const float cos_i, const float eta) {
    const float c = fabsf(cos_i);
    float g = eta * eta - 1 + c * c;
    if (!(g > 0)) {
        return 1.0f;
    }
    g = sqrtf(g);
    const float a = (g - c) / (g + c);
    const float b = (c * (g + c) - 1) / (c * (g - c) + 1);
    return 0.5f * a * a * (1 + b * b);
}
// the artist "specular" knob as an index of refraction, and the matching normal-incidence reflectance
RT_HD float ior_from_specular(const float specular) { return (2.0f / (1.0f - sqrtf(0.08f * specular))) - 1.0f; }
RT_HD float reflectance_at_normal(const float ior) { return fresnel_dielectric(1.0f, ior); }
// Fresnel blend factor rescaled so that it runs 0 -> 1 between normal and grazing incidence
RT_HD float fresnel_blend(const float cos_i, const float ior, const float f0) { return (fresnel_dielectric(cos_i, ior) - f0) / (1.0f - f0); }

// ---- GGX distribution, Smith masking ------------------------------------------------------------------------------------
// D(h) through the slopes of the half vector (ShadeRef.cpp:231-240)
RT_HD float ggx_D(const f3 h, const f2 alpha) {
    if (h.z == 0.0f) {
        return 0.0f;
    }
    const float slope_x = -h.x / (h.z * alpha.x);
    const float slope_y = -h.y / (h.z * alpha.y);
    const float s = 1.0f + slope_x * slope_x + slope_y * slope_y;
    const float cos4 = sqr(sqr(h.z));
    return 1.0f / (sqr(s) * PI * alpha.x * alpha.y * cos4);
}
// G1(w) = 1 / (1 + Lambda(w)), Lambda = (-1 + sqrt(1 + (ax^2 wx^2 + ay^2 wy^2) / wz^2)) / 2   (ShadeRef.cpp:201-208)
RT_HD float smith_G1(const f3 w, const f2 alpha) {
    const f2 a2 = alpha * alpha;
    const float lambda = (-1.0f + sqrtf(1.0f + safe_div_pos(a2.x * sqr(w.x) + a2.y * sqr(w.y), sqr(w.z)))) / 2.0f;
    return 1.0f / (1.0f + lambda);
}
// Berry / GTR1 distribution of the clearcoat (ShadeRef.cpp:216-223)
RT_HD float gtr1_D(const float cos_h, const float a) {
    if (a >= 1.0f) {
        return 1.0f / PI;
    }
    const float a2 = sqr(a);
    const float t = 1.0f + (a2 - 1.0f) * cos_h * cos_h;
    return (a2 - 1.0f) / (PI * logf(a2) * t);
}

// ---- visible-normal sampling -----------------------------------------------------------------------------------------------
// A visible normal is drawn by stretching the view vector into the hemisphere configuration, drawing a point of the
// spherical cap above -wo_h.z (Dupuy & Benyoub), adding wo_h and un-stretching.  `bounded` shrinks the cap to the part that
// reflects above the surface (their bounded VNDF).  (ShadeRef.cpp:126-184)
RT_HD f3 cap_point(const float lower_z, const f2 u) {
    const float phi = 2.0f * PI * u.x;
    const float z = fmaf(1.0f - u.y, 1.0f + lower_z, -lower_z);
    const float r = sqrtf(saturatef(1.0f - z * z));
    const f2 sc = portable_sincos(phi);
    return f3{r * sc.y, r * sc.x, z};
}
RT_HD float bounded_cap_scale(const f3 wo, const f2 alpha) {
    const float a = saturatef(fminf(alpha.x, alpha.y));
    const float s = 1.0f + length(mk2(wo.x, wo.y));
    const float a2 = a * a, s2 = s * s;
    return (1.0f - a2) * s2 / (s2 + a2 * wo.z * wo.z);
}
template <bool BOUNDED> RT_HD f3 draw_visible_normal(const f3 wo, const f2 alpha, const f2 u) {
    const f3 wo_h = normalize(mk3(alpha.x * wo.x, alpha.y * wo.y, wo.z));
    float lower_z = wo_h.z;
    if (BOUNDED) {
        const float k = bounded_cap_scale(wo, alpha);
        lower_z = (wo.z > 0.0f) ? k * wo_h.z : wo_h.z;
    }
    const f3 n_h = cap_point(lower_z, u) + wo_h; // (unnormalised: the un-stretch below normalises)
    return normalize(mk3(alpha.x * n_h.x, alpha.y * n_h.y, fmaxf(0.0f, n_h.z)));
}
// density of the reflected direction under the bounded VNDF (ShadeRef.cpp:186-198)
RT_HD float bounded_vndf_reflection_pdf(const float D, const f3 wo, const f2 alpha) {
    const f2 ai = alpha * mk2(wo.x, wo.y);
    const float len2 = dot(ai, ai);
    const float t = sqrtf(len2 + wo.z * wo.z);
    if (wo.z >= 0.0f) {
        return D / (2.0f * (bounded_cap_scale(wo, alpha) * wo.z + t));
    }
    return D * (t - wo.z) / (2.0f * len2);
}

// ---- GGX reflection lobe ------------------------------------------------------------------------------------------------------
struct GlossLobe {
    f2 alpha;
    float ior, f0; // Fresnel blend between tint0 (normal incidence) and tint90 (grazing)
    f3 tint0, tint90;
};
// wo = view, h = half vector, wi = reflected / light direction, all in the tangent frame (ShadeRef.cpp:493-512)
RT_HD LobeValue gloss_eval(const GlossLobe &g, const f3 wo, const f3 h, const f3 wi) {
    const float D = ggx_D(h, g.alpha);
    const float G = smith_G1(wo, g.alpha) * smith_G1(wi, g.alpha);
    f3 F = mix3(g.tint0, g.tint90, fresnel_blend(dot(wo, h), g.ior, g.f0));
    const float denom = 4.0f * fabsf(wo.z * wi.z);
    F *= (denom != 0.0f) ? (D * G / denom) : 0.0f;
    F *= fmaxf(wi.z, 0.0f);
    return LobeValue{F, bounded_vndf_reflection_pdf(D, wo, g.alpha)};
}
// the delta form: a mirror carries its Fresnel tint on a nominal density of 1e6
constexpr float DELTA_PDF = 1e6f;

// ---- GGX refraction lobe ----------------------------------------------------------------------------------------------------
// (ShadeRef.cpp:538-568; the generalised half vector's Jacobian |wi.h| / (wi.h + eta wo.h)^2)
RT_HD LobeValue refract_eval(const f2 alpha, const float eta, const f3 tint, const f3 wo, const f3 h, const f3 wi) {
    if (wi.z >= 0.0f || wo.z <= 0.0f || alpha_is_singular(alpha)) {
        return no_lobe_value();
    }
    const float D = ggx_D(h, alpha);
    const float G_wi = smith_G1(wi, alpha), G_wo = smith_G1(wo, alpha);
    const float denom = dot(wi, h) + dot(wo, h) * eta;
    const float jacobian = safe_div_pos(fmaxf(-dot(wi, h), 0.0f), denom * denom);
    const float value = D * G_wo * G_wi * fmaxf(dot(wo, h), 0.0f) * jacobian / (wo.z);
    const float pdf = D * G_wi * fmaxf(dot(wo, h), 0.0f) * jacobian / wo.z;
    return LobeValue{value * tint, pdf};
}
// refracted direction for half vector h (Snell); false on total internal reflection
RT_HD bool refract_through(const f3 wo, const f3 h, const float eta, f3 &wi) {
    const float cos_i = dot(wo, h);
    const float cos_t2 = 1.0f - eta * eta * (1.0f - cos_i * cos_i);
    if (cos_t2 < 0) {
        return false;
    }
    const float m = eta * cos_i - sqrtf(cos_t2);
    wi = normalize(-eta * wo + m * h);
    return true;
}

// ---- clearcoat lobe -------------------------------------------------------------------------------------------------------------
// GTR1 distribution, masking of a fixed alpha 0.25^2, 4 % base reflectance (ShadeRef.cpp:608-628)
RT_HD LobeValue coat_eval(const float coat_alpha, const float ior, const float f0, const f3 wo, const f3 h, const f3 wi) {
    const float D = gtr1_D(h.z, coat_alpha);
    const f2 mask_alpha = {0.25f * 0.25f, 0.25f * 0.25f};
    const float G = smith_G1(wo, mask_alpha) * smith_G1(wi, mask_alpha);
    float F = mixf(0.04f, 1.0f, fresnel_blend(dot(wi, h), ior, f0));
    const float denom = 4.0f * fabsf(wo.z) * fabsf(wi.z);
    F *= (denom != 0.0f) ? D * G / denom : 0.0f;
    F *= fmaxf(wi.z, 0.0f);
    return LobeValue{f3{F, F, F}, bounded_vndf_reflection_pdf(D, wo, mask_alpha)};
}

// ---- diffuse lobes (world space: they only need N) --------------------------------------------------------------------------------
// qualitative Oren-Nayar (Fujii's form), uniform-hemisphere density (ShadeRef.cpp:403-427)
RT_HD LobeValue oren_nayar_eval(const f3 wo, const f3 n, const f3 wi, const float sigma, const f3 albedo) {
    const float norm = 1.0f / (PI + ((3.0f * PI - 4.0f) / 6.0f) * sigma);
    const float a = 1.0f * norm, b = sigma * norm;
    const float n_wi = fmaxf(dot(n, wi), 0.0f), n_wo = fmaxf(dot(n, wo), 0.0f);
    float t = dot(wi, wo) - n_wi * n_wo;
    if (t > 0.0f) {
        t /= fmaxf(n_wi, n_wo) + FLT_MIN;
    }
    const float shape = n_wi * (a + b * t);
    return LobeValue{shape * albedo, 0.5f / PI};
}
// tangent-frame direction for u (the reference draws sqrt(1 - u.x u.y) here, and so does this: ShadeRef.cpp:429-441)
RT_HD f3 oren_nayar_draw(const f2 u) {
    const f2 sc = portable_sincos(2 * PI * u.y);
    const float r = sqrtf(1.0f - u.x * u.y);
    return f3{r * sc.y, r * sc.x, u.x};
}

// Disney diffuse with retro-reflection (Fd90 = 1/2 + 2 rough (wi.h)^2) plus the sheen term, cosine-weighted density
// (ShadeRef.cpp:385-401, 443-468; the uniform-sampling variant of the reference is never requested by its callers)
RT_HD LobeValue disney_diffuse_eval(const f3 wo, const f3 n, const f3 wi, const float roughness, const f3 albedo, const f3 sheen) {
    const float pdf = dot(n, wi) / PI;
    f3 h = normalize(wi + wo);
    if (dot(wo, h) < 0.0f) {
        h = -h;
    }
    float retro = 0.0f;
    {
        const float n_wi = dot(n, wi), n_wo = dot(n, wo);
        if (!(n_wi <= 0.0f)) {
            const float f_wi = schlick5(n_wi), f_wo = schlick5(n_wo);
            const float wi_h = dot(wi, h);
            const float fd90 = 0.5f + 2.0f * wi_h * wi_h * roughness;
            retro = mixf(1.0f, fd90, f_wi) * mixf(1.0f, fd90, f_wo);
        }
    }
    f3 value = albedo * (1.0f * retro);
    value += (PI * schlick5(dot(wi, h))) * sheen;
    return LobeValue{value, pdf};
}
RT_HD f3 cosine_hemisphere_draw(const f2 u) {
    const f2 sc = portable_sincos(2 * PI * u.y);
    const float r = sqrtf(u.x), up = sqrtf(1.0f - u.x);
    return f3{r * sc.y, r * sc.x, up};
}

// ---- shading-normal repair, tangent rotation, the nested-dielectric stack ----------------------------------------------------------
// Bend shading normal `n` towards the geometric normal `ng` just enough that the mirror direction of `wo` about it stays
// `threshold` above the surface (the construction of Cycles' ensure_valid_reflection: solve for the normal in the plane
// spanned by ng and n whose reflection has exactly that elevation; ShadeRef.cpp:245-335)
RT_HD f3 keep_reflection_above_surface(const f3 ng, const f3 wo, const f3 n) {
    const f3 r = 2 * dot(n, wo) * n - wo;
    const float threshold = fminf(0.9f * dot(ng, wo), 0.01f);
    if (dot(ng, r) >= threshold) {
        return n;
    }
    // 2-D frame: z = ng, x = the part of n orthogonal to ng
    const float n_ng = dot(n, ng);
    const f3 x = normalize(n - n_ng * ng);
    const float wx = dot(wo, x), wz = dot(wo, ng);
    const float wx2 = (wx * wx), wz2 = (wz * wz);
    const float a = wx2 + wz2;
    const float b = safe_sqrt(wx2 * (a - (threshold * threshold)));
    const float c = wz * threshold + a;
    // the two candidate solutions for the squared z component of the bent normal
    const float half_inv_a = 0.5f / a;
    const float z2_plus = half_inv_a * (b + c), z2_minus = half_inv_a * (-b + c);
    bool ok_plus = (z2_plus > 1e-5f) && (z2_plus <= (1.0f + 1e-5f));
    bool ok_minus = (z2_minus > 1e-5f) && (z2_minus <= (1.0f + 1e-5f));
    f2 bent;
    if (ok_plus && ok_minus) { // both geometrically possible: compare the elevations of their reflections
        const f2 n_plus = {safe_sqrt(1.0f - z2_plus), safe_sqrt(z2_plus)};
        const f2 n_minus = {safe_sqrt(1.0f - z2_minus), safe_sqrt(z2_minus)};
        const float up_plus = 2 * (n_plus.x * wx + n_plus.y * wz) * n_plus.y - wz;
        const float up_minus = 2 * (n_minus.x * wx + n_minus.y * wz) * n_minus.y - wz;
        ok_plus = (up_plus >= 1e-5f);
        ok_minus = (up_minus >= 1e-5f);
        if (ok_plus && ok_minus) {
            bent = (up_plus < up_minus) ? n_plus : n_minus;
        } else {
            bent = (up_plus > up_minus) ? n_plus : n_minus;
        }
    } else if (ok_plus || ok_minus) {
        const float z2 = ok_plus ? z2_plus : z2_minus;
        bent = {safe_sqrt(1.0f - z2), safe_sqrt(z2)};
    } else {
        return ng;
    }
    return bent.x * x + bent.y * ng;
}

// Rodrigues rotation of p about the unit axis k, written as the rotation matrix applied row by row (ShadeRef.cpp:337-358)
RT_HD f3 rotate_about_axis(const f3 p, const f3 k, const float angle) {
    const f2 sc = portable_sincos(angle);
    const float c = sc.y, s = sc.x;
    f3 r;
    r.x = ((c + (1.0f - c) * k.x * k.x) * p.x) + (((1.0f - c) * k.x * k.y - k.z * s) * p.y) + (((1.0f - c) * k.x * k.z + k.y * s) * p.z);
    r.y = (((1.0f - c) * k.x * k.y + k.z * s) * p.x) + ((c + (1.0f - c) * k.y * k.y) * p.y) + (((1.0f - c) * k.y * k.z - k.x * s) * p.z);
    r.z = (((1.0f - c) * k.x * k.z - k.y * s) * p.x) + (((1.0f - c) * k.y * k.z + k.x * s) * p.y) + ((c + (1.0f - c) * k.z * k.z) * p.z);
    return r;
}

// The four-entry stack of refractive indices a ray is inside of (ray_data_t::ior; negative = free slot; ShadeRef.cpp:360-391)
// (written with selects over all four entries, not "find the slot, then store": a store through a computed index puts the whole
// array -- i.e. every ray's stack, in the common case where nothing refracts -- into scratch memory: four stores and four loads per
// shade point in both halves of k_scatter)
RT_HD void ior_stack_enter(float stack[4], const float ior) {
    // the first free entry from the bottom; a full stack has its top entry overwritten
    const bool at0 = stack[0] < 0.0f, at1 = !at0 && stack[1] < 0.0f, at2 = !at0 && !at1 && stack[2] < 0.0f, at3 = !at0 && !at1 && !at2;
    stack[0] = at0 ? ior : stack[0];
    stack[1] = at1 ? ior : stack[1];
    stack[2] = at2 ? ior : stack[2];
    stack[3] = at3 ? ior : stack[3];
}
RT_HD void ior_stack_leave(float stack[4]) {
    // the topmost occupied entry, if any
    const bool at3 = stack[3] > 0.0f, at2 = !at3 && stack[2] > 0.0f, at1 = !at3 && !at2 && stack[1] > 0.0f,
               at0 = !at3 && !at2 && !at1 && stack[0] > 0.0f;
    stack[3] = at3 ? -1.0f : stack[3];
    stack[2] = at2 ? -1.0f : stack[2];
    stack[1] = at1 ? -1.0f : stack[1];
    stack[0] = at0 ? -1.0f : stack[0];
}
// the medium on the far side of the surface: the top of the stack, or the entry below it when the ray is leaving the
// top medium (skip_top); vacuum when there is none
RT_HD float peek_ior_stack(const float stack[4], bool skip_top) {
    for (int i = 3; i >= 0; --i) {
        if (stack[i] > 0.0f) {
            if (!skip_top) {
                return stack[i];
            }
            skip_top = false;
        }
    }
    return 1.0f;
}

} // namespace rt
