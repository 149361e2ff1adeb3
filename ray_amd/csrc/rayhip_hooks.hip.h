// rayhip_hooks.hip.h -- part of librayhip's host side (one translation unit: included by rayhip.hip, in this order, after the kernels):
// kernel-level test hooks (rayhip_k_*): single stages on caller-supplied rays, for the parity tests.
#pragma once

// ---- kernel-level hooks ---------------------------------------------------------------------------------------

int rayhip_k_generate_primary_rays(rayhip_ctx *c, const rayhip_camera *cam, const int rect[4], int iteration,
                                   rayhip_ray *out_rays, rayhip_hit *out_hits, int *out_count) {
    if (use_device(c)) {
        return 1;
    }
    if (!c->w || !c->pmj.p || !c->filter_table.p) {
        return fail("k_generate_primary_rays needs resize + upload_static + set_filter_table first");
    }
    hipStream_t s = c->stream;
    const RayGenTiling tiling = make_raygen_tiling(c->w, c->h, rect[2], rect[3], c->shard);
    const size_t nslots = size_t(tiling.tiles) * 64u;
    if (c->clear_queues(1, s)) {
        return fail("queue counter clear failed");
    }
    const RayGenParams rg = make_raygen_params(*cam, c->w, c->h, rect, iteration, c->shard);
    // kernel-level hooks use one dense stripe so that the host sees a plain array
    k_raygen<<<grid_for(c, nslots, 256), 256, 0, s>>>(rg, c->sc.pmj, c->filter_table.as<float>(), c->px.required_samples,
                                                      c->rays[0], c->hits, c->ray_queue(0, nslots, 1), single_layer(c->w, c->h), tiling);
    HIP_TRY(hipGetLastError());
    uint32_t n = 0;
    HIP_TRY(hipMemcpyAsync(&n, c->ray_count(0), 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    std::vector<float4> pl[4], hp(n);
    std::vector<uint2> xd(n);
    std::vector<float> hv(n);
    for (int k = 0; k < 4; ++k) {
        pl[k].resize(n);
        HIP_TRY(hipMemcpyAsync(pl[k].data(), c->ray_planes[0][k].p, size_t(n) * 16, hipMemcpyDeviceToHost, s));
    }
    HIP_TRY(hipMemcpyAsync(xd.data(), c->ray_planes[0][4].p, size_t(n) * 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(hp.data(), c->hit_planes[0].p, size_t(n) * 16, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(hv.data(), c->hit_planes[1].p, size_t(n) * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    for (uint32_t i = 0; i < n; ++i) {
        rayhip_ray &r = out_rays[i];
        r.o[0] = pl[0][i].x, r.o[1] = pl[0][i].y, r.o[2] = pl[0][i].z, r.pdf = pl[0][i].w;
        r.d[0] = pl[1][i].x, r.d[1] = pl[1][i].y, r.d[2] = pl[1][i].z, r.cone_width = pl[1][i].w;
        r.c[0] = pl[2][i].x, r.c[1] = pl[2][i].y, r.c[2] = pl[2][i].z, r.cone_spread = pl[2][i].w;
        r.ior[0] = pl[3][i].x, r.ior[1] = pl[3][i].y, r.ior[2] = pl[3][i].z, r.ior[3] = pl[3][i].w;
        r.xy = xd[i].x, r.depth = xd[i].y;
        rayhip_hit &h = out_hits[i];
        memcpy(&h.obj_index, &hp[i].x, 4), memcpy(&h.prim_index, &hp[i].y, 4);
        h.t = hp[i].z, h.u = hp[i].w, h.v = hv[i];
    }
    *out_count = int(n);
    return 0;
}

int rayhip_k_intersect_closest(rayhip_ctx *c, const rayhip_camera *cam, rayhip_ray *rays, rayhip_hit *hits, int count,
                               int iteration, uint32_t flags, rayhip_trav_counters *out_counters) {
    if (use_device(c)) {
        return 1;
    }
    if (!c->have_scene || !c->pmj.p) {
        return fail("k_intersect_closest needs a scene and the PMJ table");
    }
    if (size_t(count) > size_t(c->w) * size_t(c->h)) {
        return fail("ray count exceeds the wavefront buffers (w*h)");
    }
    hipStream_t s = c->stream;
    std::vector<float4> pl[4], hp;
    hp.resize(size_t(count));
    std::vector<uint2> xd;
    std::vector<float> hv{};
    hv.resize(size_t(count));
    rays_to_soa(rays, count, pl, xd);
    for (int i = 0; i < count; ++i) {
        float oi, pi;
        memcpy(&oi, &hits[i].obj_index, 4), memcpy(&pi, &hits[i].prim_index, 4);
        hp[i] = make_float4(oi, pi, hits[i].t, hits[i].u);
        hv[i] = hits[i].v;
    }
    for (int k = 0; k < 4; ++k) {
        HIP_TRY(hipMemcpyAsync(c->ray_planes[0][k].p, pl[k].data(), size_t(count) * 16, hipMemcpyHostToDevice, s));
    }
    HIP_TRY(hipMemcpyAsync(c->ray_planes[0][4].p, xd.data(), size_t(count) * 8, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(c->hit_planes[0].p, hp.data(), size_t(count) * 16, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(c->hit_planes[1].p, hv.data(), size_t(count) * 4, hipMemcpyHostToDevice, s));
    const uint32_t n = uint32_t(count);
    HIP_TRY(hipMemcpyAsync(c->ray_count(0), &n, 4, hipMemcpyHostToDevice, s));
    unsigned long long *tc = c->trav_counters.as<unsigned long long>();
    unsigned long long before[TRAV_COUNTER_WORDS], after[TRAV_COUNTER_WORDS];
    HIP_TRY(hipMemcpyAsync(before, tc, sizeof(before), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemsetAsync(tc, 0, sizeof(before), s));
    const TraceParams tp = make_trace_params(*cam, c->sc.tlas_root, iteration);
    const int g = int(std::min<size_t>(size_t(c->grid_waves), (size_t(count) + WAVE - 1) / WAVE));
    const RayQueue q = c->ray_queue(0, size_t(count), 1);
    {
        const int gg = g ? g : 1;
#define KK_ARGS c->sc, tp, c->rays[0], c->hits, q, 0, c->stack_spill.as<uint32_t>(), tc, single_layer(c->w, c->h)
        if ((flags & RAYHIP_FLAG_COUNT_WIDE) && c->wide == 8) { // the product walk with counters
            k_trace_closest<true, 8><<<gg, WAVE, 0, s>>>(KK_ARGS);
        } else if ((flags & RAYHIP_FLAG_COUNT_WIDE) && c->wide == 4) {
            k_trace_closest<true, 4><<<gg, WAVE, 0, s>>>(KK_ARGS);
        } else if (flags & RAYHIP_FLAG_COUNT_TRAVERSAL) { // instrumented walk of the reference's BVH2
            k_trace_closest<true, 0><<<gg, WAVE, 0, s>>>(KK_ARGS);
        } else if (c->wide == 4 && c->refill_pool && c->pool_scene) { // what rayhip_render launches (the pooled form also takes preset hits)
            k_trace_closest_pool<><<<std::max(1, std::min(g, c->pool_waves)), WAVE, 0, s>>>(c->sc, tp, c->rays[0], c->hits, q, 0, c->stack_spill.as<uint32_t>(), single_layer(c->w, c->h));
        } else if (c->wide == 8 && c->refill_waves) {
            k_trace_closest_refill<8><<<std::max(1, std::min(g, c->refill_waves)), WAVE, 0, s>>>(c->sc, tp, c->rays[0], c->hits, q, 0, c->stack_spill.as<uint32_t>(), single_layer(c->w, c->h), nullptr);
        } else if (c->wide == 4 && c->refill_waves) {
            k_trace_closest_refill<4><<<std::max(1, std::min(g, c->refill_waves)), WAVE, 0, s>>>(c->sc, tp, c->rays[0], c->hits, q, 0, c->stack_spill.as<uint32_t>(), single_layer(c->w, c->h), nullptr);
        } else if (c->wide == 8) {
            k_trace_closest<false, 8><<<gg, WAVE, 0, s>>>(KK_ARGS);
        } else if (c->wide == 4) {
            k_trace_closest<false, 4><<<gg, WAVE, 0, s>>>(KK_ARGS);
        } else {
            k_trace_closest<false, 0><<<gg, WAVE, 0, s>>>(KK_ARGS);
        }
#undef KK_ARGS
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(s));
    HIP_TRY(hipMemcpyAsync(after, tc, sizeof(after), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(tc, before, sizeof(before), hipMemcpyHostToDevice, s));
    HIP_TRY(hipStreamSynchronize(s));
    if (out_counters) {
        out_counters->rays = after[0], out_counters->nodes = after[1];
        out_counters->tris = after[2], out_counters->instances = after[3];
        out_counters->max_stack = after[4], out_counters->nodes4 = after[5];
    }
    HIP_TRY(hipMemcpyAsync(pl[2].data(), c->ray_planes[0][2].p, size_t(count) * 16, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(xd.data(), c->ray_planes[0][4].p, size_t(count) * 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(hp.data(), c->hit_planes[0].p, size_t(count) * 16, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(hv.data(), c->hit_planes[1].p, size_t(count) * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    for (int i = 0; i < count; ++i) {
        rays[i].c[0] = pl[2][i].x, rays[i].c[1] = pl[2][i].y, rays[i].c[2] = pl[2][i].z;
        rays[i].depth = xd[i].y;
        memcpy(&hits[i].obj_index, &hp[i].x, 4), memcpy(&hits[i].prim_index, &hp[i].y, 4);
        hits[i].t = hp[i].z, hits[i].u = hp[i].w, hits[i].v = hv[i];
    }
    return 0;
}

int rayhip_k_intersect_shadow(rayhip_ctx *c, const rayhip_camera *cam, const rayhip_shadow_ray *rays, int count,
                              int iteration, float *out_rc, rayhip_trav_counters *out_counters) {
    if (use_device(c)) {
        return 1;
    }
    if (!c->have_scene || !c->pmj.p) {
        return fail("k_intersect_shadow needs a scene and the PMJ table");
    }
    if (size_t(count) > size_t(c->w) * size_t(c->h)) {
        return fail("ray count exceeds the wavefront buffers (w*h)");
    }
    hipStream_t s = c->stream;
    std::vector<float4> pl[3];
    for (int k = 0; k < 3; ++k) {
        pl[k].resize(size_t(count));
    }
    for (int i = 0; i < count; ++i) {
        float depth_f, xy_f;
        memcpy(&depth_f, &rays[i].depth, 4), memcpy(&xy_f, &rays[i].xy, 4);
        pl[0][i] = make_float4(rays[i].o[0], rays[i].o[1], rays[i].o[2], depth_f);
        pl[1][i] = make_float4(rays[i].d[0], rays[i].d[1], rays[i].d[2], rays[i].dist);
        pl[2][i] = make_float4(rays[i].c[0], rays[i].c[1], rays[i].c[2], xy_f);
    }
    for (int k = 0; k < 3; ++k) {
        HIP_TRY(hipMemcpyAsync(c->shadow_planes[k].p, pl[k].data(), size_t(count) * 16, hipMemcpyHostToDevice, s));
    }
    const uint32_t n = uint32_t(count);
    HIP_TRY(hipMemcpyAsync(c->shadow_count(0), &n, 4, hipMemcpyHostToDevice, s));
    unsigned long long *tc = c->trav_counters.as<unsigned long long>() + TRAV_COUNTER_WORDS;
    unsigned long long before[TRAV_COUNTER_WORDS], after[TRAV_COUNTER_WORDS];
    HIP_TRY(hipMemcpyAsync(before, tc, sizeof(before), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemsetAsync(tc, 0, sizeof(before), s));
    const TraceParams tp = make_trace_params(*cam, c->sc.tlas_root, iteration);
    const int g = int(std::min<size_t>(size_t(c->grid_waves), (size_t(count) + WAVE - 1) / WAVE));
    // results land in the (otherwise idle) hit plane
    if (getenv("RAYHIP_HOOK_SHADOW_REFILL") && c->wide == 4) { // the product's flat persistent form over the 4-wide tree (no counters)
        k_trace_shadow_refill<<<g ? g : 1, WAVE, 0, s>>>(c->sc, tp, c->shadow, c->shadow_queue(0, size_t(count), 1), FLT_MAX, c->w, c->px.temp,
                                                       c->hit_planes[0].as<float4>(), c->stack_spill.as<uint32_t>(), single_layer(c->w, c->h), nullptr);
    } else {
        k_trace_shadow<true, 0><<<g ? g : 1, WAVE, 0, s>>>(c->sc, tp, c->shadow, c->shadow_queue(0, size_t(count), 1), FLT_MAX, c->w, c->px.temp,
                                                        c->hit_planes[0].as<float4>(), c->stack_spill.as<uint32_t>(), tc, single_layer(c->w, c->h));
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(s));
    HIP_TRY(hipMemcpyAsync(after, tc, sizeof(after), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(tc, before, sizeof(before), hipMemcpyHostToDevice, s));
    HIP_TRY(hipStreamSynchronize(s));
    if (out_counters) {
        out_counters->rays = after[0], out_counters->nodes = after[1];
        out_counters->tris = after[2], out_counters->instances = after[3];
        out_counters->max_stack = after[4], out_counters->nodes4 = after[5];
    }
    HIP_TRY(hipMemcpyAsync(out_rc, c->hit_planes[0].p, size_t(count) * 16, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return 0;
}

int rayhip_k_shade(rayhip_ctx *c, const rayhip_camera *cam, int bounce, int iteration, const rayhip_ray *rays, const rayhip_hit *hits,
                   int count, float *inout_color, rayhip_ray *out_secondary, int *out_secondary_count, rayhip_shadow_ray *out_shadow,
                   int *out_shadow_count) {
    if (use_device(c)) {
        return 1;
    }
    if (!c->have_scene || !c->pmj.p || !c->w) {
        return fail("k_shade needs resize + upload_static + scene_upload first");
    }
    if (count < 0 || size_t(count) > size_t(c->w) * size_t(c->h)) {
        return fail("ray count exceeds the wavefront buffers (w*h)");
    }
    if (bounce < 0 || bounce + 2 > MAX_BOUNCE_SLOTS || iteration < 1) {
        return fail("bad bounce / iteration");
    }
    hipStream_t s = c->stream;
    const size_t npix = size_t(c->w) * size_t(c->h);
    std::vector<float4> pl[4], hp(static_cast<size_t>(count));
    std::vector<uint2> xd;
    std::vector<float> hv(static_cast<size_t>(count));
    rays_to_soa(rays, count, pl, xd);
    for (int i = 0; i < count; ++i) {
        float oi, pi;
        memcpy(&oi, &hits[i].obj_index, 4), memcpy(&pi, &hits[i].prim_index, 4);
        hp[i] = make_float4(oi, pi, hits[i].t, hits[i].u);
        hv[i] = hits[i].v;
    }
    for (int k = 0; k < 4; ++k) {
        HIP_TRY(hipMemcpyAsync(c->ray_planes[0][k].p, pl[k].data(), size_t(count) * 16, hipMemcpyHostToDevice, s));
    }
    HIP_TRY(hipMemcpyAsync(c->ray_planes[0][4].p, xd.data(), size_t(count) * 8, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(c->hit_planes[0].p, hp.data(), size_t(count) * 16, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(c->hit_planes[1].p, hv.data(), size_t(count) * 4, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(c->px.temp, inout_color, npix * 16, hipMemcpyHostToDevice, s));
    if (c->clear_queues(bounce + 2, s)) {
        return fail("queue counter clear failed");
    }
    const uint32_t n = uint32_t(count);
    HIP_TRY(hipMemcpyAsync(c->ray_count(bounce), &n, 4, hipMemcpyHostToDevice, s));
    const int g = std::max(1, int(std::min<size_t>(size_t(c->grid_waves), (size_t(count) + WAVE - 1) / WAVE)));
    // one dense stripe, so the host sees plain arrays; the kernels are the ones rayhip_render launches
    launch_shade(c, *cam, iteration, bounce, 0, size_t(count), 1, g, c->w, 1.0f / float(iteration), single_layer(c->w, c->h));
    HIP_TRY(hipGetLastError());
    uint32_t n_sec = 0, n_sh = 0;
    HIP_TRY(hipMemcpyAsync(&n_sec, c->ray_count(bounce + 1), 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(&n_sh, c->shadow_count(bounce), 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(inout_color, c->px.temp, npix * 16, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    for (int k = 0; k < 4; ++k) {
        pl[k].resize(n_sec);
        HIP_TRY(hipMemcpyAsync(pl[k].data(), c->ray_planes[1][k].p, size_t(n_sec) * 16, hipMemcpyDeviceToHost, s));
    }
    xd.resize(n_sec);
    HIP_TRY(hipMemcpyAsync(xd.data(), c->ray_planes[1][4].p, size_t(n_sec) * 8, hipMemcpyDeviceToHost, s));
    std::vector<float4> sp_[3];
    for (int k = 0; k < 3; ++k) {
        sp_[k].resize(n_sh);
        HIP_TRY(hipMemcpyAsync(sp_[k].data(), c->shadow_planes[k].p, size_t(n_sh) * 16, hipMemcpyDeviceToHost, s));
    }
    HIP_TRY(hipStreamSynchronize(s));
    for (uint32_t i = 0; i < n_sec; ++i) {
        rayhip_ray &r = out_secondary[i];
        r.o[0] = pl[0][i].x, r.o[1] = pl[0][i].y, r.o[2] = pl[0][i].z, r.pdf = pl[0][i].w;
        r.d[0] = pl[1][i].x, r.d[1] = pl[1][i].y, r.d[2] = pl[1][i].z, r.cone_width = pl[1][i].w;
        r.c[0] = pl[2][i].x, r.c[1] = pl[2][i].y, r.c[2] = pl[2][i].z, r.cone_spread = pl[2][i].w;
        r.ior[0] = pl[3][i].x, r.ior[1] = pl[3][i].y, r.ior[2] = pl[3][i].z, r.ior[3] = pl[3][i].w;
        r.xy = xd[i].x, r.depth = xd[i].y;
    }
    for (uint32_t i = 0; i < n_sh; ++i) {
        rayhip_shadow_ray &r = out_shadow[i];
        r.o[0] = sp_[0][i].x, r.o[1] = sp_[0][i].y, r.o[2] = sp_[0][i].z, memcpy(&r.depth, &sp_[0][i].w, 4);
        r.d[0] = sp_[1][i].x, r.d[1] = sp_[1][i].y, r.d[2] = sp_[1][i].z, r.dist = sp_[1][i].w;
        r.c[0] = sp_[2][i].x, r.c[1] = sp_[2][i].y, r.c[2] = sp_[2][i].z, memcpy(&r.xy, &sp_[2][i].w, 4);
    }
    *out_secondary_count = int(n_sec), *out_shadow_count = int(n_sh);
    return 0;
}

int rayhip_k_scrambled_rand(rayhip_ctx *c, const uint32_t *dims, const uint32_t *seeds, const int32_t *samples, int count,
                            float *out_xy) {
    if (use_device(c)) {
        return 1;
    }
    if (!c->pmj.p) {
        return fail("k_scrambled_rand needs the PMJ table");
    }
    hipStream_t s = c->stream;
    DevBuf d, sd, sm, o;
    if (d.alloc(size_t(count) * 4) || sd.alloc(size_t(count) * 4) || sm.alloc(size_t(count) * 4) || o.alloc(size_t(count) * 8)) {
        return 1;
    }
    HIP_TRY(hipMemcpyAsync(d.p, dims, size_t(count) * 4, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(sd.p, seeds, size_t(count) * 4, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(sm.p, samples, size_t(count) * 4, hipMemcpyHostToDevice, s));
    k_scrambled_rand<<<(count + 255) / 256, 256, 0, c->stream>>>(d.as<uint32_t>(), sd.as<uint32_t>(), sm.as<int32_t>(), count,
                                                                 c->sc.pmj, o.as<float2>());
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemcpyAsync(out_xy, o.p, size_t(count) * 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    d.release(), sd.release(), sm.release(), o.release();
    return 0;
}

#if defined(RT_PROFILE_SHADE) || defined(RT_PROFILE_TRACE)
// tuning build only (tools/variants.py): cycles per shade-kernel section, see RT_PROF in kernels.hip.h
__attribute__((visibility("default"))) int rayhip_tuning_read_profile(rayhip_ctx *c, unsigned long long out[32], int reset) {
    if (use_device(c)) {
        return 1;
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemcpyFromSymbol(out, HIP_SYMBOL(rt::g_prof_acc), 32 * sizeof(unsigned long long)));
    if (reset) {
        unsigned long long z[32] = {};
        HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(rt::g_prof_acc), z, sizeof(z)));
    }
    return 0;
}
#endif

} // extern "C"
