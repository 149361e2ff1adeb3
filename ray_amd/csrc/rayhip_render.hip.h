// rayhip_render.hip.h -- part of librayhip's host side (one translation unit: included by rayhip.hip, in this order, after the kernels):
// one wavefront pass (render_pass: ray generation, K2 / shade / K3 per bounce, accumulate), layered passes, rayhip_render[_batch].
#pragma once

// do the allocated per-iteration pixel buffers and wavefront state hold a pass of `n` iterations over `rect`?
static bool pass_fits(const rayhip_ctx *c, const int rect[4], int n) {
    const Layering L = make_layering(c->w, c->h, n);
    const size_t vbytes = size_t(c->w) * size_t(c->h) * size_t(L.cols) * size_t(layer_rows(L)) * 16u;
    return vbytes <= c->px_temp.bytes && (n <= 1 || (vbytes <= c->px_aux_base.bytes && vbytes <= c->px_aux_dn.bytes)) &&
           pass_slots(c, c->w, c->h, rect[2], rect[3], n) + size_t(WAVE) * QUEUE_MAX_STRIPES <= c->slots_cap;
}
// grow the per-iteration pixel buffers / the wavefront state if a pass of `n` iterations over `rect` needs more
static bool rect_inside(const rayhip_ctx *c, const int rect[4]) {
    return rect[0] >= 0 && rect[1] >= 0 && rect[2] > 0 && rect[3] > 0 && rect[0] <= c->w - rect[2] && rect[1] <= c->h - rect[3];
}
static int ensure_pass(rayhip_ctx *c, const int rect[4], int n) {
    if (!rect_inside(c, rect)) { // before anything is (re)allocated for it
        return fail("rect outside the frame");
    }
    if (!pass_fits(c, rect, n)) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (alloc_frame(c, c->w, c->h, n)) {
            return 1;
        }
    }
    return 0;
}

// K5 of one bounce: shade the rays of queue `bounce` in ray buffer `cur` -> secondary rays into queue bounce + 1 of the
// other ray buffer, shadow rays into shadow queue `bounce`, radiance into the per-iteration pixel buffer.  One place for
// rayhip_render and the kernel-level hook rayhip_k_shade.
static void launch_shade(rayhip_ctx *c, const rayhip_camera &cam, int iteration, int bounce, int cur, size_t nslots, uint32_t stripes,
                         int gtrace, int vw, float mix_factor, const Layering &layers, bool plain_ior = false, bool sized = false, int split = -1) {
    ShadeLaunch a;
    a.sc = c->sc;
    a.sp = make_shade_params(cam, iteration, bounce);
    a.sp.plain_ior = plain_ior ? 1u : 0u;
    a.rays_in = c->rays[cur], a.rays_out = c->rays[cur ^ 1];
    a.hits = c->hits, a.shadow = c->shadow, a.deferred = c->deferred, a.points = c->points;
    a.in = c->ray_queue(bounce, nslots, stripes), a.pts = c->point_queue(bounce, nslots, stripes);
    a.out_rays = c->ray_queue(bounce + 1, nslots, stripes), a.out_shadow = c->shadow_queue(bounce, nslots, stripes);
    a.out_deferred = c->deferred_queue(bounce, nslots, stripes), a.nee = c->nee_queue(bounce, nslots, stripes);
    a.out_sky = c->sky_queue(bounce, nslots, stripes), a.sky_index = c->sky_index.as<uint32_t>();
    a.px = c->px, a.layers = layers, a.vw = vw, a.mix_factor = mix_factor;
    a.bounce = bounce, a.grid = gtrace, a.split = split >= 0 ? split : c->shade_split, a.stream = c->stream;
    a.picks = c->pick_plane.as<float4>(), a.record_rays = c->record_rays, a.tag = c->next_shade_tag();
    a.pick_lds = c->pick_lds;
    a.no_textures = c->textures_count == 0 && c->notex_kernels;
    if (sized) { // a pass (not a kernel-level hook): grids from the queue census, the persistent pick with a work counter
        a.expect[EXPECT_RAYS] = bounce == 0 ? uint32_t(nslots / WAVE + stripes) : c->expect_chunks(bounce, 0, nslots, stripes);
        a.expect[EXPECT_POINTS] = c->expect_chunks(bounce, 3, nslots, stripes), a.expect[EXPECT_LIT] = c->expect_chunks(bounce, 4, nslots, stripes);
        a.expect[EXPECT_DEFERRED] = c->expect_chunks(bounce, 2, nslots, stripes), a.expect[EXPECT_SKY] = c->expect_chunks(bounce, 5, nslots, stripes);
        a.chunks_per_block = c->chunks_per_block;
        a.work = c->next_work(), a.chunks = uint32_t((nslots + WAVE - 1) / WAVE + stripes), a.dyn_mult = c->dyn_mult;
    }
    shade::launch(a);
}

// One wavefront pass over `count` consecutive iterations of the rect (count == 1: the plain case; > 1: layered, see
// Layering in rt_base.h).  The caller has checked that a batch is admissible.
static int render_pass(rayhip_ctx *c, const rayhip_camera *cam, const int rect[4], int iteration, int count_iterations,
                       uint32_t flags, rayhip_stats *stats) {
    if (use_device(c)) {
        return 1;
    }
    if (!c->w || !c->have_scene || !c->pmj.p || !c->filter_table.p) {
        return fail("rayhip_render needs resize + upload_static + scene_upload + set_filter_table first");
    }
    if (iteration < 1) {
        return fail("iteration is 1-based");
    }
    if (cam->type != 0 /* eCamType::Persp */) {
        return fail("only perspective cameras are supported");
    }
    if (cam->view_transform != 0 /* eViewTransform::Standard */ && cam->view_transform != c->lut_transform) {
        return fail("view transform %d needs its look-up table: rayhip_set_tonemap_lut", int(cam->view_transform));
    }
    if (!rect_inside(c, rect)) {
        return fail("rect outside the frame");
    }
    const int max_depth = cam->pass_settings.max_total_depth;
    if (max_depth + 2 > MAX_BOUNCE_SLOTS) {
        return fail("max_total_depth too large");
    }
    const bool count_wide = (flags & RAYHIP_FLAG_COUNT_WIDE) != 0;
    const bool count = !count_wide && (flags & RAYHIP_FLAG_COUNT_TRAVERSAL) != 0;
    const bool sort_rays = (flags & RAYHIP_FLAG_SORT_RAYS) != 0;
    if (cam->pass_settings.variance_threshold != 0.0f) {
        c->adaptive_dirty = true;
    }
    hipStream_t s = c->stream;
    const size_t npix = size_t(rect[2]) * size_t(rect[3]);
    const Layering layers = make_layering(c->w, c->h, count_iterations);
    const int vw = virtual_width(layers); // row pitch of the per-iteration pixel buffers
    // ray slots: the 8x8 tiles the ray generator walks (this rank's share under a shard), one set per iteration in flight
    RayGenTiling tiling = make_raygen_tiling(c->w, c->h, rect[2], rect[3], c->shard);
    if (c->raygen_samples_per_wave > 1 && layers.count % c->raygen_samples_per_wave == 0) {
        tiling.samples_per_wave = uint32_t(c->raygen_samples_per_wave);
    }
    const size_t nslots = size_t(tiling.tiles) * 64u * size_t(count_iterations);
    if (!pass_fits(c, rect, count_iterations)) {
        return fail("internal: pass of %d iterations exceeds the allocated wavefront state", count_iterations);
    }
    const int gw = c->grid_waves;
    const int gtrace = int(std::min<size_t>(size_t(gw), nslots / WAVE));
    unsigned long long *tc = c->trav_counters.as<unsigned long long>();
    uint32_t *spill = c->stack_spill.as<uint32_t>();
    const TraceParams tp_ = make_trace_params(*cam, c->sc.tlas_root, iteration);

    // striped queues unless a stage needs one dense ray array (the sort)
    const uint32_t stripes = sort_rays ? 1u : QUEUE_MAX_STRIPES;
    // The queue census of the previous pass, if it has arrived (rayhip_ctx: census): fractions of that pass's ray slots
    if (c->census_on && c->census_pending && hipEventQuery(c->census_event) == hipSuccess) {
        const double inv = c->census_slots ? 1.0 / double(c->census_slots) : 0.0;
        for (int k = 0; k < c->census_bounces * rayhip_ctx::QUEUES_PER_BOUNCE; ++k) {
            c->census[size_t(k)] = float(double(c->census_host[k]) * inv);
        }
        c->census_valid = true, c->census_pending = false;
    }
    int bounce_now = 0; // (the launchers below size their grids for the bounce being enqueued)
    // grid of a persistent kernel whose chunks are handed out dynamically: what the device holds at once (x RAYHIP_DYN_MULT), never more blocks
    // than chunks are expected (every block takes chunk blockIdx.x first), never more than the spill slabs allow
    auto dyn_grid = [&](int resident, uint32_t expect) {
        const size_t chunks = expect ? size_t(expect) : (nslots + WAVE - 1) / WAVE + stripes;
        return int(std::max<size_t>(1, std::min<size_t>({size_t(gw), size_t(std::max(resident, 1)) * size_t(c->dyn_mult), chunks})));
    };
    // ... and of one that walks statically: chunks_per_block live chunks per block, at least a block per wave slot while there are chunks
    auto static_grid = [&](int full, int resident, uint32_t expect) {
        if (!expect) {
            return full;
        }
        const uint32_t want = std::max(std::min(expect, uint32_t(std::max(resident, 1))), expect / uint32_t(c->chunks_per_block));
        return int(std::min<uint32_t>(uint32_t(full), std::max(want, 1u)));
    };
    // K2 launcher (instrumented variant on request)
    auto launch_closest = [&](const RaySoA &r, const RayQueue &q, int init_hits) {
        const int wide = c->wide;
#define K2_ARGS c->sc, tp_, r, c->hits, q, init_hits, spill, tc, layers
        if (count_wide && wide == 8) {
            k_trace_closest<true, 8><<<gtrace, WAVE, 0, s>>>(K2_ARGS);
        } else if (count_wide && wide == 4) {
            k_trace_closest<true, 4><<<gtrace, WAVE, 0, s>>>(K2_ARGS);
        } else if (count) {
            k_trace_closest<true, 0><<<gtrace, WAVE, 0, s>>>(K2_ARGS);
        } else if (wide && c->refill_waves && c->refill_primary_whole && !init_hits) {
            // primary rays (coherent): the flat kernel, chunks taken whole (RAYHIP_REFILL=3)
            // (static walk: this launch takes a chunk every ~8 ns chip-wide, more than one counter can hand out -- wavefront.hip.h)
            if (wide == 8) {
                k_trace_closest_refill<8, WAVE><<<gtrace, WAVE, 0, s>>>(c->sc, tp_, r, c->hits, q, init_hits, spill, layers, nullptr);
            } else {
                k_trace_closest_refill<4, WAVE><<<gtrace, WAVE, 0, s>>>(c->sc, tp_, r, c->hits, q, init_hits, spill, layers, nullptr);
            }
        } else if (wide == 4 && c->refill_pool && c->pool_scene && init_hits) {
            // secondary bounces, pooled kernel (grid: as for the refill kernel below)
            const int want = int(std::min<size_t>(size_t(c->pool_waves), std::max<size_t>(size_t(c->pool_resident), nslots / WAVE / 8)));
            k_trace_closest_pool<><<<std::min(gtrace, want), WAVE, 0, s>>>(c->sc, tp_, r, c->hits, q, init_hits, spill, layers);
        } else if (wide && c->refill_waves && !(c->refill_secondary_only && !init_hits)) {
            // blocks: enough to even out the end of the launch (16 per wave slot on a full-size pass), but never so many that a
            // block gets fewer than ~8 chunks of 64 rays -- below that the kernel degenerates into the plain one with extra
            // set-up per block (a rank of 8 at 20 spp: 5.2 M rays per pass; 16 blocks per slot 6.6 ms, 4: 5.97, plain 5.98)
            const int want = int(std::min<size_t>(size_t(c->refill_waves), std::max<size_t>(size_t(c->refill_resident), nslots / WAVE / 8)));
            // round 5: chunks handed out dynamically -- as many blocks as the device holds, every wavefront refills its lanes until the
            // queue is empty and drains once (wavefront.hip.h: ChunkWalk)
            uint32_t *work = c->next_work();
            const uint32_t expect = c->expect_chunks(bounce_now, 0, nslots, stripes);
            const int resident = wide == 4 ? c->refill_resident4 : c->refill_resident;
            const int grid = work ? dyn_grid(resident, expect) : static_grid(std::min(gtrace, want), resident, expect);
            if (wide == 8) {
                k_trace_closest_refill<8><<<grid, WAVE, 0, s>>>(c->sc, tp_, r, c->hits, q, init_hits, spill, layers, work);
            } else {
                k_trace_closest_refill<4><<<grid, WAVE, 0, s>>>(c->sc, tp_, r, c->hits, q, init_hits, spill, layers, work);
            }
        } else if (wide == 8 && (c->small_scene || c->tune_primary_waves == 5)) {
            k_trace_closest<false, 8, RT_TRACE_SMALL_WAVES><<<gtrace, WAVE, 0, s>>>(K2_ARGS);
        } else if (wide == 8) {
            k_trace_closest<false, 8><<<gtrace, WAVE, 0, s>>>(K2_ARGS);
        } else if (wide == 4 && (c->small_scene || c->tune_primary_waves == 5)) {
            k_trace_closest<false, 4, RT_TRACE_SMALL_WAVES><<<gtrace, WAVE, 0, s>>>(K2_ARGS);
        } else if (wide == 4) {
            k_trace_closest<false, 4><<<gtrace, WAVE, 0, s>>>(K2_ARGS);
        } else {
            k_trace_closest<false, 0><<<gtrace, WAVE, 0, s>>>(K2_ARGS);
        }
#undef K2_ARGS
    };

    StageTimer tm(c, stats != nullptr || (flags & RAYHIP_FLAG_TIME_STAGES) != 0);

    if (c->clear_queues(max_depth + 2, s) || c->clear_work(max_depth + 2, s)) {
        return fail("queue counter clear failed");
    }

    RayGenParams rg = make_raygen_params(*cam, c->w, c->h, rect, iteration, c->shard);
    rg.skip_ior = c->plain_ior ? 1 : 0; // (the scatter stage will not read the plane: rayhip_upload.hip.h)
    const TraceParams tp = make_trace_params(*cam, c->sc.tlas_root, iteration);
    const float mix_factor = 1.0f / float(iteration);

    const bool trace_launch = getenv("RAYHIP_TRACE_LAUNCH") != nullptr; // (diagnostics: host time of the first calls of a pass)
    const auto h0 = std::chrono::steady_clock::now();
    if (tm.mark(ST_GEN, -1)) {
        return 1;
    }
    const auto h1 = std::chrono::steady_clock::now();
    k_raygen<<<grid_for(c, nslots, 256), 256, 0, s>>>(rg, c->sc.pmj, c->filter_table.as<float>(), c->px.required_samples,
                                                      c->rays[0], c->hits, c->ray_queue(0, nslots, stripes), layers, tiling);
    const auto h2 = std::chrono::steady_clock::now();
    if (tm.mark(ST_PTRACE, 0)) {
        return 1;
    }
    if (trace_launch) {
        const auto h3 = std::chrono::steady_clock::now();
        auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
        fprintf(stderr, "rayhip pass start (host): event record %.0f us, k_raygen launch %.0f us, event record %.0f us\n", us(h0, h1), us(h1, h2), us(h2, h3));
    }
    if (c->sc.tlas_root != 0xffffffffu) {
        launch_closest(c->rays[0], c->ray_queue(0, nslots, stripes), 0);
    }
    const int shade_form = c->shade_split_for_pass(); // (one form for the whole pass: rayhip_ctx.hip.h)
    int cur = 0;
    bool side_pending = false; // a K3 launch on the second stream that the main stream has not waited for yet
    bool side_forked = false;  // anything was sent to the second stream in this pass
    // every error exit below leaves through here: a shadow launch still running on the second stream would race with the next pass's queue
    // clears and shade launches on the shadow planes, the pixel buffer and the counters -- drain it (error path: blocking is fine)
    struct SideDrain {
        rayhip_ctx *c;
        const bool *forked, *pending;
        bool armed = true;
        ~SideDrain() {
            if (armed && *forked && *pending && c->stream2) {
                (void)hipStreamSynchronize(c->stream2);
            }
        }
    } side_drain{c, &side_forked, &side_pending};
    for (int bounce = 0; bounce <= max_depth; ++bounce) {
        bounce_now = bounce;
        if (bounce > 0) {
            if (sort_rays) {
                // K6-K8 (RendererVK.cpp:641-652): key -> radix sort of (key, index) -> gather into the idle ray buffer
                if (tm.mark(ST_SORT, -1)) {
                    return 1;
                }
                k_ray_keys<<<grid_for(c, npix, 256), 256, 0, s>>>(c->rays[cur], c->ray_count(bounce), uint32_t(npix), c->sort_grid,
                                                                  c->sort_keys[0].as<uint32_t>(), c->sort_idx[0].as<uint32_t>(), c->sort_key_mode);
                HIP_TRY(sort_pairs(c->sort_temp.p, c->sort_temp.bytes, c->sort_keys[0].as<uint32_t>(),
                                   c->sort_keys[1].as<uint32_t>(), c->sort_idx[0].as<uint32_t>(), c->sort_idx[1].as<uint32_t>(),
                                   npix, ray_sort_key_bits(c->sort_key_mode), s));
                k_reorder_rays<<<grid_for(c, npix, 256), 256, 0, s>>>(c->rays[cur], c->rays[cur ^ 1], c->sort_idx[1].as<uint32_t>(),
                                                                      c->ray_count(bounce));
                cur ^= 1;
            }
            if (tm.mark(ST_STRACE, 0)) {
                return 1;
            }
            launch_closest(c->rays[cur], c->ray_queue(bounce, nslots, stripes), 1);
            // K2's own interval ends here.  What follows -- until the shadow launch of the bounce before, which ran beside K2, is through -- is what
            // the frame spends on shadow rays beyond the trace: that wait is the shadow stage's entry in RendererBase::stats_t, so that the stages
            // PARTITION the frame again (the reference's GPU backends report exclusive intervals: RendererVK.cpp:452-487, RendererBase.h:230-244;
            // round 5 booked K3's whole elapsed time, which overlaps the trace interval: the fields summed to 1.45 x the frame)
            if (side_pending && tm.mark(bounce - 1 == 0 ? ST_PSHADOW : ST_SSHADOW, -1)) {
                return 1;
            }
            if (c->sc.visible_lights_count != 0) {
                k_intersect_area_lights<<<gtrace, WAVE, 0, s>>>(c->sc, c->rays[cur], c->hits, c->ray_queue(bounce, nslots, stripes));
            }
        }
        if (side_pending) { // the shadow rays of the bounce before have to be through: pixel additions keep the reference's order
            HIP_TRY(hipStreamWaitEvent(s, c->join_event, 0));
            side_pending = false;
        }
        if (tm.mark(bounce == 0 ? ST_PSHADE : ST_SSHADE, -1)) {
            return 1;
        }
        launch_shade(c, *cam, iteration, bounce, cur, nslots, stripes, gtrace, vw, mix_factor, layers, c->plain_ior, true, shade_form);
        // K3 of this bounce: on the second stream, next to the closest-hit launch of the next bounce (rayhip_ctx: stream2), when it is the flat
        // persistent form and nothing is being counted
        const bool side = c->overlap_shadow && c->wide == 4 && c->shadow_refill && !count && !count_wide;
        hipStream_t ss = side ? c->stream2 : s;
        long side_t0 = -1;
        if (side) {
            HIP_TRY(hipEventRecord(c->fork_event, s));
            HIP_TRY(hipStreamWaitEvent(ss, c->fork_event, 0));
            side_forked = side_pending = true; // (from here on an error exit has to drain the second stream)
            if (tm.on) {
                side_t0 = tm.stamp(ss);
            }
        } else if (tm.mark(bounce == 0 ? ST_PSHADOW : ST_SSHADOW, 1)) {
            return 1;
        }
        const float limit = shadow_clamp_limit(*cam, bounce);
        if (c->sc.blocker_lights_count != 0) {
            k_shadow_blockers<<<gtrace, WAVE, 0, ss>>>(c->sc, c->shadow, c->shadow_queue(bounce, nslots, stripes));
        }
        {
            const int wide = c->wide;
#define K3_ARGS c->sc, tp, c->shadow, c->shadow_queue(bounce, nslots, stripes), limit, vw, c->px.temp, nullptr, spill, tc + TRAV_COUNTER_WORDS, layers
            if (count_wide && wide == 8) {
                k_trace_shadow<true, 8><<<gtrace, WAVE, 0, s>>>(K3_ARGS);
            } else if (count_wide && wide == 4) {
                k_trace_shadow<true, 4><<<gtrace, WAVE, 0, s>>>(K3_ARGS);
            } else if (count) {
                k_trace_shadow<true, 0><<<gtrace, WAVE, 0, s>>>(K3_ARGS);
            } else if (wide == 4 && c->shadow_refill) { // the flat persistent form (kernels_shadow.hip.h)
                uint32_t *work = c->next_work();
                const uint32_t expect = c->expect_chunks(bounce, 1, nslots, stripes);
                k_trace_shadow_refill<<<work ? dyn_grid(c->shadow_resident, expect) : static_grid(gtrace, c->shadow_resident, expect), WAVE, 0, ss>>>(
                    c->sc, tp, c->shadow, c->shadow_queue(bounce, nslots, stripes), limit, vw, c->px.temp, nullptr, side ? c->stack_spill2.as<uint32_t>() : spill, layers, work);
            } else if (wide == 8 && (c->small_scene || c->tune_shadow_waves == 5)) {
                k_trace_shadow<false, 8, RT_TRACE_SMALL_WAVES><<<gtrace, WAVE, 0, s>>>(K3_ARGS);
            } else if (wide == 8) {
                k_trace_shadow<false, 8><<<gtrace, WAVE, 0, s>>>(K3_ARGS);
            } else if (wide == 4 && (c->small_scene || c->tune_shadow_waves == 5)) {
                k_trace_shadow<false, 4, RT_TRACE_SMALL_WAVES><<<gtrace, WAVE, 0, s>>>(K3_ARGS);
            } else if (wide == 4) {
                k_trace_shadow<false, 4><<<gtrace, WAVE, 0, s>>>(K3_ARGS);
            } else {
                k_trace_shadow<false, 0><<<gtrace, WAVE, 0, s>>>(K3_ARGS);
            }
#undef K3_ARGS
        }
        if (side) {
            if (tm.on) {
                const long side_t1 = tm.stamp(ss);
                if (side_t0 >= 0 && side_t1 >= 0) {
                    c->pending2.push_back({size_t(side_t0), size_t(side_t1), -1, 1}); // (elapsed time -> rayhip_get_trav_timing only: not a stage)
                }
            }
            HIP_TRY(hipEventRecord(c->join_event, ss));
            side_pending = true;
        }
        cur ^= 1;
    }
    if (side_pending) { // the last shadow launch has nothing beside it: all of it is the shadow stage's
        if (tm.mark(max_depth == 0 ? ST_PSHADOW : ST_SSHADOW, -1)) {
            return 1;
        }
        HIP_TRY(hipStreamWaitEvent(s, c->join_event, 0));
        side_pending = false;
    }
    if (tm.mark(-1, -1)) {
        return 1;
    }
    AccumParams ap = make_accum_params(*cam, c->w, rect, iteration, c->shard);
    ap.lut = c->tonemap_lut.as<uint32_t>(), ap.lut_dims = c->lut_dims;
    for (int base = 0; base < count_iterations; base += MAX_BATCH) { // the layers are folded in iteration order
        const int n = std::min(MAX_BATCH, count_iterations - base);
        AccumLayers per_layer = {};
        for (int k = 0; k < n; ++k) {
            const AccumParams al = make_accum_params(*cam, c->w, rect, iteration + base + k, c->shard);
            per_layer.l[k] = AccumLayer{al.iteration, al.mix_factor, al.half_mix_factor, al.is_class_a, al.variance_threshold};
        }
        k_accumulate<<<grid_for(c, npix, 256), 256, 0, s>>>(ap, c->px, layers, per_layer, base, n);
    }
    if (c->census_on && !count && !count_wide) { // what the queues of this pass held: the next pass sizes its launches from it
        const int queues = (max_depth + 1) * rayhip_ctx::QUEUES_PER_BOUNCE;
        k_queue_totals<<<queues, WAVE, 0, s>>>(c->counters.as<uint32_t>(), c->census_dev);
        HIP_TRY(hipEventRecord(c->census_event, s));
        c->census_pending = true, c->census_bounces = max_depth + 1, c->census_slots = nslots;
    }
    HIP_TRY(hipGetLastError());
    if (tm.mark(-1, -1)) {
        return 1;
    }
    if (stats) {
        // synchronous mode: resolve now and hand this call's stage times to the caller
        double before[11];
        memcpy(before, c->stage_us, sizeof(before));
        if (resolve_timing(c)) {
            return 1;
        }
        unsigned long long *slots = reinterpret_cast<unsigned long long *>(stats);
        for (int i = 0; i < 11; ++i) {
            slots[i] += (unsigned long long)(c->stage_us[i] - before[i]);
        }
    }
    return 0;
}

int rayhip_max_batch(rayhip_ctx *c) {
    if (!c || !c->h) {
        return 0;
    }
    return max_layers_for(c->w, c->h);
}

int rayhip_reserve_batch(rayhip_ctx *c, int count) {
    if (use_device(c)) {
        return 1;
    }
    if (!c->w) {
        return fail("rayhip_reserve_batch before rayhip_resize");
    }
    const int rect[4] = {0, 0, c->w, c->h};
    return ensure_pass(c, rect, std::max(1, std::min(count, rayhip_max_batch(c))));
}

int rayhip_render_batch(rayhip_ctx *c, const rayhip_camera *cam, const int rect[4], int first_iteration, int count,
                        uint32_t flags, rayhip_stats *stats) {
    if (use_device(c)) {
        return 1;
    }
    if (count < 1) {
        return fail("batch of %d iterations", count);
    }
    if (!c->w) {
        return fail("rayhip_render needs resize + upload_static + scene_upload + set_filter_table first");
    }
    // A batch is exact only while adaptive sampling is inert (the reference re-queues every pixel every iteration when
    // variance_threshold == 0); the ray sort works on one dense ray array; pixel rows are 16-bit.
    // ... and only while no pixel can have required_samples < first_iteration: once an adaptive pass has run, pixels may be
    // parked; a one-by-one run would wake them up again after the first iteration with threshold 0, a batch would not
    // (k_raygen decides liveness once per pass).  adaptive_dirty is cleared by Clear / Resize.
    int max_layers = rayhip_max_batch(c);
    if (cam->pass_settings.variance_threshold != 0.0f || c->adaptive_dirty || (flags & RAYHIP_FLAG_SORT_RAYS) != 0) {
        max_layers = 1;
    }
    int done = 0;
    while (done < count) {
        const int n = std::min(count - done, max_layers);
        if (ensure_pass(c, rect, n)) {
            return 1;
        }
        if (render_pass(c, cam, rect, first_iteration + done, n, flags, stats)) {
            return 1;
        }
        done += n;
    }
    return 0;
}

int rayhip_render(rayhip_ctx *c, const rayhip_camera *cam, const int rect[4], int iteration, uint32_t flags,
                  rayhip_stats *stats) {
    if (use_device(c)) {
        return 1;
    }
    if (c->w && ensure_pass(c, rect, 1)) {
        return 1;
    }
    return render_pass(c, cam, rect, iteration, 1, flags, stats);
}
