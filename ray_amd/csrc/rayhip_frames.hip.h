// rayhip_frames.hip.h -- part of librayhip's host side (one translation unit: included by rayhip.hip, in this order, after the kernels):
// shards, frame read-back / import, synchronisation, counters and stage times.
#pragma once

int rayhip_set_shard(rayhip_ctx *c, int tile, int shard_count, int shard_index) {
    if (tile <= 0 || shard_count <= 0 || shard_index < 0 || shard_index >= shard_count) {
        return fail("bad shard (tile %d, %d of %d)", tile, shard_index, shard_count);
    }
    c->shard = Shard{tile, shard_count, shard_index};
    return 0;
}

static float4 *pick_buffer(rayhip_ctx *c, int which) {
    switch (which) {
    case RAYHIP_BUF_FINAL:
        return c->px.final_;
    case RAYHIP_BUF_RAW:
        return c->px.raw;
    case RAYHIP_BUF_BASE_COLOR:
        return c->px.base_color;
    case RAYHIP_BUF_DEPTH_NORMALS:
        return c->px.depth_normals;
    case RAYHIP_BUF_VARIANCE:
        return c->px.variance;
    default:
        return nullptr;
    }
}

int rayhip_readback(rayhip_ctx *c, int which, float *dst_rgba, int pitch_px) {
    if (use_device(c)) {
        return 1;
    }
    float4 *src = pick_buffer(c, which);
    if (!src) {
        return fail("bad buffer id %d", which);
    }
    if (pitch_px == c->w) { // one linear copy (a 2-D copy of 1080 rows into pageable memory is staged row by row)
        HIP_TRY(hipMemcpyAsync(dst_rgba, src, size_t(c->w) * size_t(c->h) * 16, hipMemcpyDeviceToHost, c->stream));
    } else {
        HIP_TRY(hipMemcpy2DAsync(dst_rgba, size_t(pitch_px) * 16, src, size_t(c->w) * 16, size_t(c->w) * 16, size_t(c->h),
                                 hipMemcpyDeviceToHost, c->stream));
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}

int rayhip_readback_device(rayhip_ctx *c, int which, void *dst_device_rgba, int pitch_px) {
    if (use_device(c)) {
        return 1;
    }
    float4 *src = pick_buffer(c, which);
    if (!src) {
        return fail("bad buffer id %d", which);
    }
    HIP_TRY(hipMemcpy2DAsync(dst_device_rgba, size_t(pitch_px) * 16, src, size_t(c->w) * 16, size_t(c->w) * 16, size_t(c->h),
                             hipMemcpyDeviceToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}

int rayhip_set_raw_device(rayhip_ctx *c, const void *src_device_rgba, int pitch_px, const rayhip_camera *cam) {
    if (use_device(c)) {
        return 1;
    }
    HIP_TRY(hipMemcpy2DAsync(c->px.full, size_t(c->w) * 16, src_device_rgba, size_t(pitch_px) * 16, size_t(c->w) * 16,
                             size_t(c->h), hipMemcpyDeviceToDevice, c->stream));
    const int rect[4] = {0, 0, c->w, c->h};
    if (cam->view_transform != 0 && cam->view_transform != c->lut_transform) {
        return fail("view transform %d needs its look-up table: rayhip_set_tonemap_lut", int(cam->view_transform));
    }
    AccumParams ap = make_accum_params(*cam, c->w, rect, 1);
    ap.lut = c->tonemap_lut.as<uint32_t>(), ap.lut_dims = c->lut_dims;
    k_retonemap<<<grid_for(c, size_t(c->w) * c->h, 256), 256, 0, c->stream>>>(ap, c->px, c->h);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}

int rayhip_sync(rayhip_ctx *c) {
    if (use_device(c)) {
        return 1;
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}

int rayhip_get_trav_counters(rayhip_ctx *c, rayhip_trav_counters out[2], int reset) {
    if (use_device(c)) {
        return 1;
    }
    unsigned long long h[2 * TRAV_COUNTER_WORDS];
    HIP_TRY(hipMemcpyAsync(h, c->trav_counters.p, sizeof(h), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    for (int k = 0; k < 2; ++k) {
        const unsigned long long *w = h + TRAV_COUNTER_WORDS * k;
        out[k].rays = w[0], out[k].nodes = w[1], out[k].tris = w[2], out[k].instances = w[3], out[k].max_stack = w[4], out[k].nodes4 = w[5];
    }
    if (reset) {
        HIP_TRY(hipMemsetAsync(c->trav_counters.p, 0, sizeof(h), c->stream));
    }
    return 0;
}

int rayhip_get_stage_times(rayhip_ctx *c, rayhip_stats *out, int reset) {
    if (use_device(c) || resolve_timing(c)) {
        return 1;
    }
    unsigned long long *slots = reinterpret_cast<unsigned long long *>(out);
    for (int i = 0; i < 11; ++i) {
        slots[i] = (unsigned long long)(c->stage_us[i]);
        if (reset) {
            c->stage_us[i] = 0.0;
        }
    }
    return 0;
}

int rayhip_get_trav_timing(rayhip_ctx *c, double out_ms[2], unsigned long long out_launches[2], int reset) {
    if (use_device(c) || resolve_timing(c)) {
        return 1;
    }
    for (int k = 0; k < 2; ++k) {
        out_ms[k] = c->trav_ms[k];
        out_launches[k] = c->trav_launches[k];
        if (reset) {
            c->trav_ms[k] = 0.0, c->trav_launches[k] = 0;
        }
    }
    return 0;
}
