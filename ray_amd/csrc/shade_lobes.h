// shade_lobes.h -- stage 3 of shading: a shade point scatters light.
//
// One pass over the material's lobes produces BOTH estimates a path-tracing vertex needs --
//   next-event estimation   the picked light sample weighted by the material (and by the power heuristic against the
//                           material's own sampling density) -> a shadow ray carrying that radiance, or radiance at once
//                           for lights that cast no shadow
//   continuation            a direction drawn from one lobe -> the secondary ray with its throughput and density
// -- from one set of per-point quantities: the tangent frame, the view direction in it, the GGX alphas of the lobes.
// (The reference evaluates them in separate Evaluate_*Node / Sample_*Node passes, ShadeRef.cpp:659-1026; the arithmetic
// of every value is kept in its order, because parity is checked bit for bit on the host build.)
//
// Materials are lobe sets:   Diffuse = Oren-Nayar;   Glossy = GGX reflection;   Refractive = GGX refraction;
// Principled = Disney diffuse + sheen | GGX reflection (anisotropic, tinted) | clearcoat | Fresnel-split reflection /
// refraction, one of them picked for the continuation in proportion to its luminance weight.
#pragma once

#include "shade_point.h"

namespace rt {

// what stage 3 emits
struct Scatter {
    f3 direct;       // radiance of shadow-less lights, not yet multiplied by the path throughput
    Ray next;        // valid when has_next
    ShadowRay shadow; // valid when has_shadow
    bool has_next, has_shadow;
};

// the shade point's frame + the incoming ray, shared by all lobes
struct ScatterFrame {
    f3 P, T, B, N, Ng;
    f3 I;            // ray direction (into the surface); the "view" vector of the BSDFs is -I
    f3 wo;           // -I in the tangent frame (not normalised: the reference evaluates with the raw projection)
    float floor_alpha; // path regularisation: minimum GGX alpha after the first diffuse bounce
    float mix_weight;
    bool mis;        // the path may continue, so BSDF sampling competes with next-event estimation
};
RT_HD f3 to_tangent(const ScatterFrame &fr, const f3 v) { return tangent_from_world(fr.T, fr.B, fr.N, v); }
RT_HD f3 to_world(const ScatterFrame &fr, const f3 v) { return world_from_tangent(fr.T, fr.B, fr.N, v); }

// ---- next-event bookkeeping --------------------------------------------------------------------------------------------------
// weight of the light sample against the density `bsdf_pdf` with which the material would have drawn the same direction
RT_HD float nee_mis_weight(const ScatterFrame &fr, const LightSample &ls, const float bsdf_pdf) {
    return (fr.mis && ls.area > 0.0f) ? power_heuristic(ls.pdf, bsdf_pdf) : 1.0f;
}
// hand the weighted light radiance to a shadow ray leaving from `side` of the surface, or book it at once
RT_HD void deliver(const ScatterFrame &fr, const LightSample &ls, const f3 radiance, const f3 side, Scatter &out) {
    if (!ls.casts_shadow) {
        out.direct += radiance;
        return;
    }
    out.shadow.o = offset_ray(fr.P, side);
    out.shadow.c = radiance;
}

// ---- continuation bookkeeping -------------------------------------------------------------------------------------------------
enum Bounce { BOUNCE_DIFFUSE, BOUNCE_SPECULAR, BOUNCE_REFRACT };
RT_HD void count_bounce(Ray &next, const uint32_t depth_in, const Bounce kind) {
    const int type = kind == BOUNCE_DIFFUSE ? RAY_TYPE_DIFFUSE : (kind == BOUNCE_SPECULAR ? RAY_TYPE_SPECULAR : RAY_TYPE_REFR);
    next.depth = pack_ray_type(type);
    next.depth |= mask_ray_depth(depth_in) + pack_ray_depth(kind == BOUNCE_DIFFUSE, kind == BOUNCE_SPECULAR, kind == BOUNCE_REFRACT, 0);
}
RT_HD void scale_throughput(Ray &next, const f3 f, const float k) { next.c = f3{f.x * k, f.y * k, f.z * k}; }

// ---- drawing from the microfacet lobes (world-space result + value) ----------------------------------------------------------------
// GGX reflection: a visible normal (bounded cap), mirrored view; a singular alpha degenerates to the mirror direction
// carrying the Fresnel tint (ShadeRef.cpp:514-536)
RT_HD LobeValue gloss_draw(const ScatterFrame &fr, const GlossLobe &g, const f2 u, f3 &dir) {
    if (alpha_is_singular(g.alpha)) {
        dir = mirror(fr.I, fr.N, dot(fr.N, fr.I));
        const f3 F = mix3(g.tint0, g.tint90, fresnel_blend(dot(dir, fr.N), g.ior, g.f0));
        return LobeValue{f3{F.x * DELTA_PDF, F.y * DELTA_PDF, F.z * DELTA_PDF}, DELTA_PDF};
    }
    const f3 wo = normalize(fr.wo);
    const f3 h = draw_visible_normal<true>(wo, g.alpha, u);
    const f3 wi = normalize(mirror(-wo, h, -dot(h, wo)));
    dir = to_world(fr, wi);
    return gloss_eval(g, wo, h, wi);
}
// GGX refraction: a visible normal (full cap), Snell; zero value on total internal reflection (ShadeRef.cpp:570-606)
RT_HD LobeValue refract_draw(const ScatterFrame &fr, const f2 alpha, const float eta, const f3 tint, const f2 u, f3 &dir) {
    dir = f3{0.0f, 0.0f, 0.0f};
    if (alpha_is_singular(alpha)) {
        const float cos_i = -dot(fr.I, fr.N);
        const float cos_t2 = 1.0f - eta * eta * (1.0f - cos_i * cos_i);
        if (cos_t2 < 0) {
            return no_lobe_value();
        }
        const float m = eta * cos_i - sqrtf(cos_t2);
        dir = normalize(eta * fr.I + m * fr.N);
        return LobeValue{f3{tint.x * DELTA_PDF, tint.y * DELTA_PDF, tint.z * DELTA_PDF}, DELTA_PDF};
    }
    const f3 wo = normalize(fr.wo);
    const f3 h = draw_visible_normal<false>(wo, alpha, u);
    f3 wi;
    if (!refract_through(wo, h, eta, wi)) {
        return no_lobe_value();
    }
    const LobeValue v = refract_eval(alpha, eta, tint, wo, h, wi);
    dir = to_world(fr, wi);
    return v;
}
// clearcoat: visible normals of a GGX with the coat's alpha (Cycles draws it that way), GTR1 value (ShadeRef.cpp:630-657)
RT_HD LobeValue coat_draw(const ScatterFrame &fr, const float coat_alpha, const float ior, const float f0, const f2 u, f3 &dir) {
    if (sqr(coat_alpha) < 1e-7f) {
        dir = mirror(fr.I, fr.N, dot(fr.N, fr.I));
        const float F = mixf(0.04f, 1.0f, fresnel_blend(dot(dir, fr.N), ior, f0));
        return LobeValue{f3{F * DELTA_PDF, F * DELTA_PDF, F * DELTA_PDF}, DELTA_PDF};
    }
    const f3 wo = normalize(fr.wo);
    const f3 h = draw_visible_normal<true>(wo, mk2(coat_alpha, coat_alpha), u);
    const f3 wi = normalize(mirror(-wo, h, -dot(h, wo)));
    dir = to_world(fr, wi);
    return coat_eval(coat_alpha, ior, f0, wo, h, wi);
}

// ---- single-lobe materials ----------------------------------------------------------------------------------------------------------
// half vector of reflection (eta = 1) / refraction in the tangent frame, for a light direction L
RT_HD f3 half_vector_ts(const ScatterFrame &fr, const f3 L, const float eta_scale, const bool scaled) {
    return to_tangent(fr, normalize(scaled ? L - fr.I * eta_scale : L - fr.I));
}

RT_HD void scatter_diffuse(const ScatterFrame &fr, const ShadePoint &pt, const LightSample &ls, const bool light_usable, const bool may_continue,
                           const uint32_t depth_in, const f2 u, Scatter &out) {
    if (light_usable && (ls.ray_mask & RAY_TYPE_DIFFUSE_BIT) != 0 && dot(fr.N, ls.dir) > 0.0f) {
        const LobeValue v = oren_nayar_eval(-fr.I, fr.N, ls.dir, pt.roughness, pt.base);
        deliver(fr, ls, ls.radiance * v.f * (fr.mix_weight * nee_mis_weight(fr, ls, v.pdf) / ls.pdf), fr.Ng, out);
    }
    if (may_continue) {
        const f3 dir = to_world(fr, oren_nayar_draw(u));
        const LobeValue v = oren_nayar_eval(-fr.I, fr.N, dir, pt.roughness, pt.base);
        count_bounce(out.next, depth_in, BOUNCE_DIFFUSE);
        out.next.o = offset_ray(fr.P, fr.Ng);
        out.next.d = dir;
        out.next.c = f3{v.f.x * fr.mix_weight / v.pdf, v.f.y * fr.mix_weight / v.pdf, v.f.z * fr.mix_weight / v.pdf};
        out.next.pdf = v.pdf;
        out.next.cone_spread += MAX_CONE_SPREAD_INCREMENT;
    }
}

RT_HD void scatter_glossy(const ScatterFrame &fr, const ShadePoint &pt, const LightSample &ls, const bool light_usable, const bool may_continue,
                          const uint32_t depth_in, const f2 u, Scatter &out) {
    GlossLobe g;
    g.alpha = ggx_alpha(pt.roughness, 0.0f, fr.floor_alpha);
    g.ior = ior_from_specular(0.5f);
    g.f0 = reflectance_at_normal(g.ior);
    g.tint0 = g.tint90 = pt.base;
    if (light_usable && (ls.ray_mask & RAY_TYPE_SPECULAR_BIT) != 0 && dot(fr.N, ls.dir) > 0.0f && !alpha_is_singular(g.alpha)) {
        const LobeValue v = gloss_eval(g, fr.wo, half_vector_ts(fr, ls.dir, 1.0f, false), to_tangent(fr, ls.dir));
        deliver(fr, ls, ls.radiance * v.f * (fr.mix_weight * nee_mis_weight(fr, ls, v.pdf) / ls.pdf), fr.Ng, out);
    }
    if (may_continue) {
        f3 dir;
        const LobeValue v = gloss_draw(fr, g, u, dir);
        count_bounce(out.next, depth_in, BOUNCE_SPECULAR);
        out.next.o = offset_ray(fr.P, fr.Ng);
        out.next.d = dir;
        scale_throughput(out.next, v.f, safe_div_pos(fr.mix_weight, v.pdf));
        out.next.pdf = v.pdf;
        out.next.cone_spread += MAX_CONE_SPREAD_INCREMENT * fminf(g.alpha.x, g.alpha.y);
    }
}

RT_HD void scatter_refractive(const ScatterFrame &fr, const ShadePoint &pt, const float inside_ior, const float outside_ior, const LightSample &ls,
                              const bool light_usable, const bool may_continue, const uint32_t depth_in, const f2 u, Scatter &out) {
    const f2 alpha = ggx_alpha(pt.roughness, 0.0f, fr.floor_alpha);
    const float eta = pt.backfacing ? (inside_ior / outside_ior) : (outside_ior / inside_ior);
    if (light_usable && (ls.ray_mask & RAY_TYPE_REFR_BIT) != 0 && dot(fr.N, ls.dir) < 0.0f) {
        const LobeValue v = refract_eval(alpha, eta, pt.base, fr.wo, half_vector_ts(fr, ls.dir, eta, true), to_tangent(fr, ls.dir));
        deliver(fr, ls, ls.radiance * v.f * (fr.mix_weight * nee_mis_weight(fr, ls, v.pdf) / ls.pdf), -fr.Ng, out);
    }
    if (may_continue) {
        f3 dir;
        const LobeValue v = refract_draw(fr, alpha, eta, pt.base, u, dir);
        count_bounce(out.next, depth_in, BOUNCE_REFRACT);
        scale_throughput(out.next, v.f, safe_div_pos(fr.mix_weight, v.pdf));
        out.next.pdf = v.pdf;
        if (!pt.backfacing) {
            ior_stack_enter(out.next.ior, inside_ior);
        } else {
            ior_stack_leave(out.next.ior);
        }
        out.next.o = offset_ray(fr.P, -fr.Ng);
        out.next.d = dir;
        out.next.cone_spread += MAX_CONE_SPREAD_INCREMENT * fminf(alpha.x, alpha.y);
    }
}

// ---- Principled ------------------------------------------------------------------------------------------------------------------------
struct PrincipledLobes {
    // continuation probabilities of the four lobes (sum to 1 unless the material is black)
    float p_diffuse, p_gloss, p_coat, p_transmit;
    // diffuse + sheen
    f3 sheen;
    float diffuse_scale; // (1 - metallic)(1 - transmission)
    // tinted anisotropic reflection
    GlossLobe gloss;
    // clearcoat
    float coat_alpha, coat_ior, coat_f0;
    bool coat_singular;
    // transmission: a Fresnel-weighted pair of an untinted reflection and a refraction
    GlossLobe clear_gloss;
    f2 transmit_alpha;
    float eta, fresnel, inside_ior;
};
// lobe weights from the luminances of what each lobe would reflect (Cycles' heuristic; ShadeRef.cpp:32-52)
RT_HD void principled_probabilities(PrincipledLobes &L, const float diffuse_lum, const float gloss_lum, const float specular, const float metallic,
                                    const float transmission, const float clearcoat) {
    L.p_diffuse = diffuse_lum * (1.0f - metallic) * (1.0f - transmission);
    const float transmitted = transmission * (1.0f - metallic);
    L.p_gloss = (specular != 0.0f || metallic != 0.0f) ? gloss_lum * (1.0f - transmitted) : 0.0f;
    L.p_coat = 0.25f * clearcoat * (1.0f - metallic);
    L.p_transmit = transmitted * diffuse_lum;
    const float total = L.p_diffuse + L.p_gloss + L.p_coat + L.p_transmit;
    if (total != 0.0f) {
        L.p_diffuse /= total;
        L.p_gloss /= total;
        L.p_coat /= total;
        L.p_transmit /= total;
    }
}
// material constants + the point's textured parameters -> lobes (ShadeRef.cpp:1539-1598)
RT_HD PrincipledLobes principled_lobes(const ScatterFrame &fr, const ShadePoint &pt, const rayhip_material &m, const float outside_ior) {
    PrincipledLobes L;
    const float unorm = 65535.0f;
    const float specular_tint = float(m.specular_tint_unorm) / unorm, transmission = float(m.transmission_unorm) / unorm;
    const float clearcoat = float(m.clearcoat_unorm) / unorm, clearcoat_roughness = float(m.clearcoat_roughness_unorm) / unorm;
    const float sheen = 2.0f * (float(m.sheen_unorm) / unorm), sheen_tint = float(m.sheen_tint_unorm) / unorm;

    // hue of the base colour at unit luminance
    f3 hue = {0.0f, 0.0f, 0.0f};
    const float base_lum = lum(pt.base);
    if (base_lum > 0.0f) {
        hue = pt.base / base_lum;
    }
    L.sheen = sheen * mix3(splat3(1.0f), hue, sheen_tint);
    L.diffuse_scale = (1.0f - pt.metallic) * (1.0f - transmission);

    L.gloss.tint0 = mix3(splat3(1.0f), hue, specular_tint);
    L.gloss.tint0 = mix3(pt.specular * 0.08f * L.gloss.tint0, pt.base, pt.metallic);
    L.gloss.tint90 = splat3(1.0f);
    L.gloss.ior = ior_from_specular(pt.specular);
    L.gloss.f0 = reflectance_at_normal(L.gloss.ior);
    L.gloss.alpha = ggx_alpha(pt.roughness, float(m.anisotropic_unorm) / unorm, fr.floor_alpha);

    // the clearcoat and transmission lobes exist only when the material switches them on; their parameters are never read
    // otherwise (probability 0), so they are not computed (each costs a square root, divisions and a Fresnel term)
    L.coat_alpha = L.coat_ior = L.coat_f0 = 0.0f;
    L.coat_singular = true;
    if (clearcoat != 0.0f) {
        L.coat_ior = ior_from_specular(clearcoat);
        L.coat_f0 = reflectance_at_normal(L.coat_ior);
        const f2 coat_alpha2 = ggx_alpha(clearcoat_roughness, 0.0f, fr.floor_alpha);
        L.coat_alpha = coat_alpha2.x;
        L.coat_singular = alpha_is_singular(coat_alpha2);
    }
    L.inside_ior = m.ior;
    L.eta = 1.0f, L.fresnel = 0.0f;
    L.transmit_alpha = L.clear_gloss.alpha = f2{0.0f, 0.0f};
    L.clear_gloss.ior = 1.0f, L.clear_gloss.f0 = 0.0f;
    L.clear_gloss.tint0 = L.clear_gloss.tint90 = splat3(1.0f);
    if (transmission != 0.0f) {
        const float transmit_roughness = 1.0f - (1.0f - pt.roughness) * (1.0f - float(m.transmission_roughness_unorm) / unorm);
        L.eta = pt.backfacing ? (m.ior / outside_ior) : (outside_ior / m.ior);
        L.fresnel = fresnel_dielectric(dot(fr.I, fr.N), 1.0f / L.eta);
        L.transmit_alpha = ggx_alpha(transmit_roughness, 0.0f, fr.floor_alpha);
        L.clear_gloss.alpha = ggx_alpha(pt.roughness, 0.0f, fr.floor_alpha);
    }

    // luminance the reflection lobe shows at this view angle (shading normal standing in for the half vector)
    const float grazing = fresnel_blend(dot(fr.I, fr.N), L.gloss.ior, L.gloss.f0);
    const float gloss_lum = lum(mix3(L.gloss.tint0, splat3(1.0f), grazing));
    principled_probabilities(L, mixf(base_lum, 1.0f, sheen), gloss_lum, pt.specular, pt.metallic, transmission, clearcoat);
    return L;
}

// next-event estimate: every lobe that can see the light adds its value; the densities add up weighted by the lobes'
// continuation probabilities (ShadeRef.cpp:810-901)
RT_HD void principled_nee(const ScatterFrame &fr, const ShadePoint &pt, const PrincipledLobes &L, const LightSample &ls, Scatter &out) {
    const float n_wi = dot(fr.N, ls.dir);
    const bool above = n_wi > 0.0f;
    f3 radiance = {0.0f, 0.0f, 0.0f};
    float bsdf_pdf = 0.0f;

    if (L.p_diffuse > 0.0f && above && (ls.ray_mask & RAY_TYPE_DIFFUSE_BIT) != 0) {
        const LobeValue v = disney_diffuse_eval(-fr.I, fr.N, ls.dir, pt.roughness, pt.base, L.sheen);
        bsdf_pdf += L.p_diffuse * v.pdf;
        radiance += ls.radiance * n_wi * (v.f * L.diffuse_scale) / (PI * ls.pdf);
    }
    const f3 h = half_vector_ts(fr, ls.dir, L.eta, !above);
    const f3 wi = to_tangent(fr, ls.dir);
    const bool mirror_side = above && (ls.ray_mask & RAY_TYPE_SPECULAR_BIT) != 0;

    if (L.p_gloss > 0.0f && !alpha_is_singular(L.gloss.alpha) && mirror_side) {
        const LobeValue v = gloss_eval(L.gloss, fr.wo, h, wi);
        bsdf_pdf += L.p_gloss * v.pdf;
        radiance += ls.radiance * v.f / ls.pdf;
    }
    if (L.p_coat > 0.0f && !L.coat_singular && mirror_side) {
        const LobeValue v = coat_eval(L.coat_alpha, L.coat_ior, L.coat_f0, fr.wo, h, wi);
        bsdf_pdf += L.p_coat * v.pdf;
        radiance += 0.25f * ls.radiance * v.f / ls.pdf;
    }
    if (L.p_transmit > 0.0f) {
        if (L.fresnel != 0.0f && !alpha_is_singular(L.clear_gloss.alpha) && mirror_side) {
            const LobeValue v = gloss_eval(L.clear_gloss, fr.wo, h, wi);
            bsdf_pdf += L.p_transmit * L.fresnel * v.pdf;
            radiance += ls.radiance * v.f * (L.fresnel / ls.pdf);
        }
        if (L.fresnel != 1.0f && !alpha_is_singular(L.transmit_alpha) && n_wi < 0.0f && (ls.ray_mask & RAY_TYPE_REFR_BIT) != 0) {
            const LobeValue v = refract_eval(L.transmit_alpha, L.eta, pt.base, fr.wo, h, wi);
            bsdf_pdf += L.p_transmit * (1.0f - L.fresnel) * v.pdf;
            radiance += ls.radiance * v.f * ((1.0f - L.fresnel) / ls.pdf);
        }
    }
    radiance *= fr.mix_weight * nee_mis_weight(fr, ls, bsdf_pdf);
    deliver(fr, ls, radiance, n_wi < 0.0f ? -fr.Ng : fr.Ng, out);
}

// continuation: `pick` selects the lobe in proportion to the probabilities; the bounce budgets of the pass gate each kind
// (ShadeRef.cpp:903-1026)
RT_HD void principled_continue(const ScatterFrame &fr, const ShadePoint &pt, const PrincipledLobes &L, const PassLimits &ps, const uint32_t depth_in,
                               const f2 u, float pick, Scatter &out) {
    const int n_diff = get_diff_depth(depth_in), n_spec = get_spec_depth(depth_in), n_refr = get_refr_depth(depth_in);
    const bool budget = (n_diff + n_spec + n_refr) < ps.max_total_depth; // (transparency crossings do not count)
    Ray &next = out.next;
    f3 dir;
    if (pick < L.p_diffuse) {
        if (n_diff < ps.max_diff_depth && budget) {
            RT_PROF_SHADE_LANES(14)
            const f3 d = to_world(fr, cosine_hemisphere_draw(u));
            LobeValue v = disney_diffuse_eval(-fr.I, fr.N, d, pt.roughness, pt.base, L.sheen);
            v.f *= L.diffuse_scale;
            count_bounce(next, depth_in, BOUNCE_DIFFUSE);
            next.o = offset_ray(fr.P, fr.Ng);
            next.d = d;
            scale_throughput(next, v.f, safe_div_pos(fr.mix_weight, L.p_diffuse));
            next.pdf = v.pdf;
            next.cone_spread += MAX_CONE_SPREAD_INCREMENT;
        }
    } else if (pick < L.p_diffuse + L.p_gloss) {
        if (n_spec < ps.max_spec_depth && budget) {
            RT_PROF_SHADE_LANES(16)
            const LobeValue v = gloss_draw(fr, L.gloss, u, dir);
            const float pdf = v.pdf * L.p_gloss;
            count_bounce(next, depth_in, BOUNCE_SPECULAR);
            scale_throughput(next, v.f, safe_div_pos(fr.mix_weight, pdf));
            next.pdf = pdf;
            next.o = offset_ray(fr.P, fr.Ng);
            next.d = dir;
            next.cone_spread += MAX_CONE_SPREAD_INCREMENT * fminf(L.gloss.alpha.x, L.gloss.alpha.y);
        }
    } else if (pick < L.p_diffuse + L.p_gloss + L.p_coat) {
        if (n_spec < ps.max_spec_depth && budget) {
            RT_PROF_SHADE_LANES(18)
            const LobeValue v = coat_draw(fr, L.coat_alpha, L.coat_ior, L.coat_f0, u, dir);
            const float pdf = v.pdf * L.p_coat;
            count_bounce(next, depth_in, BOUNCE_SPECULAR);
            const float k = safe_div_pos(fr.mix_weight, pdf);
            next.c = f3{0.25f * v.f.x * k, 0.25f * v.f.y * k, 0.25f * v.f.z * k};
            next.pdf = pdf;
            next.o = offset_ray(fr.P, fr.Ng);
            next.d = dir;
            next.cone_spread += MAX_CONE_SPREAD_INCREMENT * L.coat_alpha;
        }
    } else { // transmission: reflect with probability `fresnel`, else refract
        pick -= L.p_diffuse + L.p_gloss + L.p_coat;
        pick = safe_div_pos(pick, L.p_transmit);
        const bool reflect = pick < L.fresnel;
        if (((!reflect && n_refr < ps.max_refr_depth) || (reflect && n_spec < ps.max_spec_depth)) && budget) {
            RT_PROF_SHADE_LANES(20)
            LobeValue v;
            if (reflect) {
                v = gloss_draw(fr, L.clear_gloss, u, dir);
                count_bounce(next, depth_in, BOUNCE_SPECULAR);
                next.o = offset_ray(fr.P, fr.Ng);
                next.cone_spread += MAX_CONE_SPREAD_INCREMENT * fminf(L.clear_gloss.alpha.x, L.clear_gloss.alpha.y);
            } else {
                v = refract_draw(fr, L.transmit_alpha, L.eta, pt.base, u, dir);
                count_bounce(next, depth_in, BOUNCE_REFRACT);
                next.o = offset_ray(fr.P, -fr.Ng);
                next.cone_spread += MAX_CONE_SPREAD_INCREMENT * fminf(L.transmit_alpha.x, L.transmit_alpha.y);
                if (!pt.backfacing) {
                    ior_stack_enter(next.ior, L.inside_ior);
                } else {
                    ior_stack_leave(next.ior);
                }
            }
            const float pdf = v.pdf * L.p_transmit;
            scale_throughput(next, v.f, safe_div_pos(fr.mix_weight, pdf));
            next.pdf = pdf;
            next.d = dir;
        }
    }
}

// ---- the scatter stage ------------------------------------------------------------------------------------------------------------------
// `ray`: the ray that produced the shade point (direction, throughput, ior stack, cone, pixel, depth counters)
template <bool NEE = true, bool CONTINUE = true>
RT_HD void scatter_stage(const SceneView &sc, const ShadeParams &sp, const Ray &ray, const ShadePoint &pt, const LightPick &pick, Scatter &out,
                         const VertexRandoms *ahead = nullptr) {
    RT_PROF_SHADE_LANES(0)
    const PassLimits &ps = sp.ps;
    const rayhip_material &mat = sc.materials[pt.material];
    const PathRandom rnd = path_random(sc, sp, ray.xy, ray.depth);

    ScatterFrame fr;
    fr.P = pt.P, fr.N = pt.N, fr.B = pt.B, fr.Ng = pt.plane_N;
    fr.T = tangent_of(pt);
    fr.I = ray.d;
    fr.wo = to_tangent(fr, -fr.I);
    fr.floor_alpha = (get_diff_depth(ray.depth) > 0) ? ps.regularize_alpha : 0.0f;
    fr.mix_weight = pt.mix_weight;
    const int n_diff = get_diff_depth(ray.depth), n_spec = get_spec_depth(ray.depth), n_refr = get_refr_depth(ray.depth);
    const int n_total = n_diff + n_spec + n_refr;
    fr.mis = n_total < ps.max_total_depth;

    // the light sample (stage 2 picked the light)
    LightSample ls = no_light_sample();
    if (NEE && sc.light_cwnodes_count != 0) {
        ls = sample_light(sc, pick, pt.P, fr.T, fr.B, fr.N, rnd.get(RAND_DIM_LIGHT), [&]() { return rnd.get(RAND_DIM_TEX); });
    }
    const bool light_usable = NEE && ls.pdf > 0.0f;

    out.direct = f3{0.0f, 0.0f, 0.0f};
    out.has_next = out.has_shadow = false;
    Ray &next = out.next;
    next.ior[0] = ray.ior[0], next.ior[1] = ray.ior[1], next.ior[2] = ray.ior[2], next.ior[3] = ray.ior[3];
    next.cone_width = pt.cone_width;
    next.cone_spread = ray.cone_spread;
    next.xy = ray.xy;
    next.pdf = 0.0f;
    next.o = next.d = next.c = f3{0.0f, 0.0f, 0.0f};
    next.depth = 0;
    ShadowRay &shadow = out.shadow;
    shadow.c = f3{0.0f, 0.0f, 0.0f};
    shadow.depth = ray.depth;
    shadow.xy = ray.xy;
    shadow.o = shadow.d = f3{0.0f, 0.0f, 0.0f};
    shadow.dist = 0.0f;

    const f2 u = ahead ? ahead->bsdf : rnd.get(RAND_DIM_BSDF);
    const bool budget = n_total < ps.max_total_depth;
    const float outside_ior = peek_ior_stack(ray.ior, pt.backfacing);
    switch (mat.type) {
    case NODE_DIFFUSE:
        RT_PROF_SHADE_LANES(22)
        scatter_diffuse(fr, pt, ls, light_usable, CONTINUE && n_diff < ps.max_diff_depth && budget, ray.depth, u, out);
        break;
    case NODE_GLOSSY:
        RT_PROF_SHADE_LANES(24)
        scatter_glossy(fr, pt, ls, light_usable, CONTINUE && n_spec < ps.max_spec_depth && budget, ray.depth, u, out);
        break;
    case NODE_REFRACTIVE:
        RT_PROF_SHADE_LANES(26)
        scatter_refractive(fr, pt, mat.ior, outside_ior, ls, light_usable, CONTINUE && n_refr < ps.max_refr_depth && budget, ray.depth, u, out);
        break;
    case NODE_PRINCIPLED: {
        RT_PROF_SHADE_LANES(8)
        const PrincipledLobes L = principled_lobes(fr, pt, mat, outside_ior);
        if (light_usable) {
            RT_PROF_SHADE_LANES(10)
            principled_nee(fr, pt, L, ls, out);
        }
        if (CONTINUE) {
            RT_PROF_SHADE_LANES(12)
            principled_continue(fr, pt, L, ps, ray.depth, u, pt.mix_pick, out);
        }
    } break;
    default:
        break;
    }

    if (CONTINUE) { // Russian roulette on the continuation's throughput once the path is past its guaranteed length
        next.c *= ray.c;
        const float brightest = fmaxf(next.c.x, fmaxf(next.c.y, next.c.z));
        const float survive_u = ahead ? ahead->bsdf_pick.y : rnd.get(RAND_DIM_BSDF_PICK).y;
        const float q = (n_total > ps.min_total_depth) ? fmaxf(0.05f, 1.0f - brightest) : 0.0f;
        if (survive_u >= q && brightest > 0.0f && next.pdf > 0.0f) {
            RT_PROF_SHADE_LANES(28)
            next.pdf = fminf(next.pdf, 1e6f);
            next.c.x /= (1.0f - q);
            next.c.y /= (1.0f - q);
            next.c.z /= (1.0f - q);
            out.has_next = true;
        }
    }
    if (NEE) {
        shadow.c *= ray.c;
        if (fmaxf(shadow.c.x, fmaxf(shadow.c.y, shadow.c.z)) > 0.0f) {
            RT_PROF_SHADE_LANES(30)
            // direction and length between the two nudged end points; a negative length marks "towards the environment"
            shadow.d = normalize_len(ls.point - shadow.o, shadow.dist);
            shadow.dist *= ls.reach;
            if (ls.is_env) {
                shadow.dist = -shadow.dist;
            }
            out.has_shadow = true;
        }
    }
}

} // namespace rt
