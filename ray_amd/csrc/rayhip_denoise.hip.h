// rayhip_denoise.hip.h -- part of librayhip's host side (one translation unit: included by rayhip.hip, in this order, after the kernels):
// DenoiseImage: the NLM filter and the sixteen UNet passes.
#pragma once

int rayhip_denoise_nlm(rayhip_ctx *c, const rayhip_camera *cam, const int rect[4], int iteration) {
    if (use_device(c)) {
        return 1;
    }
    if (!c->w) {
        return fail("rayhip_denoise_nlm before rayhip_resize");
    }
    if (rect[0] < 0 || rect[1] < 0 || rect[2] <= 0 || rect[3] <= 0 || rect[0] + rect[2] > c->w || rect[1] + rect[3] > c->h) {
        return fail("rect outside the frame");
    }
    if (iteration < 1) {
        return fail("iteration is 1-based (the RegionContext::iteration of the last RenderScene)");
    }
    if (cam->view_transform != 0 && cam->view_transform != c->lut_transform) {
        return fail("view transform %d needs its look-up table: rayhip_set_tonemap_lut", int(cam->view_transform));
    }
    DenoiseParams p;
    p.w = c->w, p.h = c->h;
    for (int i = 0; i < 4; ++i) {
        p.rect[i] = rect[i];
    }
    p.ext_w = rect[2] + 2 * NLM_EXT_RADIUS, p.ext_h = rect[3] + 2 * NLM_EXT_RADIUS;
    p.iteration = iteration;
    AccumParams tone = make_accum_params(*cam, c->w, rect, iteration, c->shard);
    tone.lut = c->tonemap_lut.as<uint32_t>(), tone.lut_dims = c->lut_dims;
    p.variance_threshold = tone.variance_threshold; // what the last RenderScene left in variance_threshold_ (RendererCPU.h:583-604)
    const size_t n_ext = size_t(p.ext_w) * size_t(p.ext_h);
    if (c->nlm_tm.alloc(n_ext * 16) || c->nlm_var_h.alloc(n_ext * 16) || c->nlm_var.alloc(n_ext * 16)) {
        return 1;
    }
    hipStream_t s = c->stream;
    k_nlm_prepare_h<<<grid_for(c, n_ext, 256), 256, 0, s>>>(p, c->px, c->nlm_tm.as<float4>(), c->nlm_var_h.as<float4>());
    k_nlm_prepare_v<<<grid_for(c, n_ext, 256), 256, 0, s>>>(p, c->nlm_var_h.as<float4>(), c->nlm_var.as<float4>());
    const size_t tiles = size_t((rect[2] + 15) / 16) * size_t((rect[3] + 15) / 16);
    k_nlm_filter<<<int(std::min<size_t>(tiles, size_t(c->props.multiProcessorCount) * 32u)), 256, 0, s>>>(
        p, tone, c->px, c->nlm_tm.as<float4>(), c->nlm_var.as<float4>());
    HIP_TRY(hipGetLastError());
    return 0;
}

// ---- UNet denoiser: InitUNetFilter / DenoiseImage(pass, region) (RendererCPU.h:1261-1310, 790-1007) ------------------------------
namespace {
// the sixteen passes: inputs (tensor ids, -1 = none; `up`: the first input is read through the nearest-neighbour upsample;
// `img`: the renderer's three images as nine more channels), output, resolution divider of the pass, pooling
struct UNetPassDesc {
    int a, a_ch, up, b, b_ch, img, cout, out, div, pool;
};
constexpr UNetPassDesc UNET_PASSES[16] = {
    {-1, 0, 0, -1, 0, 1, 32, 0, 1, 0},     // enc_conv0     images -> encConv0
    {0, 32, 0, -1, 0, 0, 32, 1, 1, 1},     // enc_conv1     -> pool1 (1/2)
    {1, 32, 0, -1, 0, 0, 48, 2, 2, 1},     // enc_conv2     -> pool2 (1/4)
    {2, 48, 0, -1, 0, 0, 64, 3, 4, 1},     // enc_conv3     -> pool3 (1/8)
    {3, 64, 0, -1, 0, 0, 80, 4, 8, 1},     // enc_conv4     -> pool4 (1/16)
    {4, 80, 0, -1, 0, 0, 96, 5, 16, 0},    // enc_conv5a
    {5, 96, 0, -1, 0, 0, 96, 6, 16, 0},    // enc_conv5b    -> upsample4 (still 1/16: the consumer upsamples)
    {6, 96, 1, 3, 64, 0, 112, 7, 8, 0},    // dec_conv4a    up(upsample4) ++ pool3
    {7, 112, 0, -1, 0, 0, 112, 8, 8, 0},   // dec_conv4b    -> upsample3
    {8, 112, 1, 2, 48, 0, 96, 9, 4, 0},    // dec_conv3a    up(upsample3) ++ pool2
    {9, 96, 0, -1, 0, 0, 96, 10, 4, 0},    // dec_conv3b    -> upsample2
    {10, 96, 1, 1, 32, 0, 64, 11, 2, 0},   // dec_conv2a    up(upsample2) ++ pool1
    {11, 64, 0, -1, 0, 0, 64, 12, 2, 0},   // dec_conv2b    -> upsample1
    {12, 64, 1, -1, 0, 1, 64, 13, 1, 0},   // dec_conv1a    up(upsample1) ++ images
    {13, 64, 0, -1, 0, 0, 32, 14, 1, 0},   // dec_conv1b
    {14, 32, 0, -1, 0, 0, 3, -1, 1, 0},    // dec_conv0     -> the filtered image
};
// resolution divider and channels of the fifteen tensors (UNetFilter.cpp:118-134)
constexpr int UNET_TENSOR_DIV[15] = {1, 2, 4, 8, 16, 16, 16, 8, 8, 4, 4, 2, 2, 1, 1};
constexpr int UNET_TENSOR_CH[15] = {32, 32, 48, 64, 80, 96, 96, 112, 112, 96, 96, 64, 64, 64, 32};
int round_up16(const int v) { return 16 * ((v + 15) / 16); }
} // namespace

// `weights` / `offsets`: what the reference's SetupUNetWeights<float>(alignment, &offsets, weights) produces (UNetFilter.cpp:296-570:
// per output channel three rows of round_up(3 * in_channels, alignment) floats, [ky][kx * in_channels + c]; a concatenating
// convolution keeps the rows of its two inputs one after the other) -- `offsets` is unet_weight_offsets_t as 32 ints.
int rayhip_unet_init(rayhip_ctx *c, const float *weights, int weights_count, const int32_t offsets[32], int alignment) {
    if (use_device(c)) {
        return 1;
    }
    if (!weights || !offsets || alignment < 1) {
        return fail("rayhip_unet_init: bad arguments");
    }
    using rt::unet::CHUNK;
    auto rt_of = [&](const int ch) { return alignment * ((3 * ch + alignment - 1) / alignment); };
    for (int pass = 0; pass < 16; ++pass) {
        const UNetPassDesc &d = UNET_PASSES[pass];
        const int c1 = d.a >= 0 ? d.a_ch : 0, c2 = d.b >= 0 ? d.b_ch : (d.img ? 9 : 0);
        const int n_tiles = (d.cout + 15) / 16, wp = rt::unet::weight_pitch(n_tiles);
        const int chunks = c1 / CHUNK + (d.b >= 0 ? d.b_ch / CHUNK : 0) + (d.img ? 1 : 0);
        // rows of one output channel in the reference blob: the first input's, then the second's (images count as one input)
        const int rt1 = c1 ? rt_of(c1) : 0, rt2 = c2 ? rt_of(c2) : 0;
        const int per_out = 3 * (rt1 + rt2);
        const int64_t w_off = offsets[2 * pass], b_off = offsets[2 * pass + 1];
        if (w_off < 0 || b_off < 0 || w_off + int64_t(per_out) * d.cout > weights_count || b_off + d.cout > weights_count) {
            return fail("rayhip_unet_init: pass %d reads outside the weight blob", pass);
        }
        std::vector<float> w(size_t(chunks) * 9 * CHUNK * wp, 0.0f), b(size_t(n_tiles) * 16, 0.0f);
        for (int n = 0; n < d.cout; ++n) {
            b[size_t(n)] = weights[b_off + n];
            for (int tap = 0; tap < 9; ++tap) {
                const int ky = tap / 3, kx = tap % 3;
                for (int cin = 0; cin < c1 + c2; ++cin) {
                    float v;
                    int chunk, cc;
                    if (cin < c1) {
                        v = weights[w_off + int64_t(n) * per_out + ky * rt1 + kx * c1 + cin];
                        chunk = cin / CHUNK, cc = cin % CHUNK;
                    } else {
                        const int k = cin - c1;
                        v = weights[w_off + int64_t(n) * per_out + 3 * rt1 + ky * rt2 + kx * c2 + k];
                        chunk = c1 / CHUNK + k / CHUNK, cc = k % CHUNK;
                    }
                    w[((size_t(chunk) * 9 + tap) * CHUNK + cc) * wp + n] = v;
                }
            }
        }
        // the f16 form's weights (unet.h: ConvParamsH): per input its own chunks of 32 channels (the last one of an odd multiple of 16 half
        // empty), [chunk][tap][out channel (16 n_tiles)][32], the 16-byte units of every 64-byte row in the order the kernel's operand
        // reads expect them (swizzle_h) -- the reference's fp16 backends convert the same blob (SetupUNetWeights<uint16_t>, RendererGPU.h:533-545)
        using rt::unet::CHUNK_H;
        const int ch1 = c1 ? (c1 + CHUNK_H - 1) / CHUNK_H : 0, c2_dev = d.b >= 0 ? d.b_ch : (d.img ? CHUNK : 0), ch2 = c2_dev ? (c2_dev + CHUNK_H - 1) / CHUNK_H : 0;
        const int rows = 9 * n_tiles * 16;
        std::vector<_Float16> wh(size_t(ch1 + ch2) * size_t(rows) * CHUNK_H, _Float16(0.0f));
        for (int n = 0; n < d.cout; ++n) {
            for (int tap = 0; tap < 9; ++tap) {
                const int ky = tap / 3, kx = tap % 3, L = tap * (n_tiles * 16) + n;
                for (int cin = 0; cin < c1 + c2; ++cin) {
                    float v;
                    int chunk, k;
                    if (cin < c1) {
                        v = weights[w_off + int64_t(n) * per_out + ky * rt1 + kx * c1 + cin];
                        chunk = cin / CHUNK_H, k = cin % CHUNK_H;
                    } else {
                        const int kk = cin - c1;
                        v = weights[w_off + int64_t(n) * per_out + 3 * rt1 + ky * rt2 + kx * c2 + kk];
                        chunk = ch1 + kk / CHUNK_H, k = kk % CHUNK_H;
                    }
                    wh[(size_t(chunk) * size_t(rows) + size_t(L)) * CHUNK_H + size_t(rt::unet::swizzle_h(n, k / 8) * 8 + k % 8)] = _Float16(v);
                }
            }
        }
        if (upload(c, c->unet_pass[pass].weights, w.data(), w.size() * sizeof(float)) ||
            upload(c, c->unet_pass[pass].bias, b.data(), b.size() * sizeof(float)) ||
            upload(c, c->unet_pass[pass].weights_h, wh.data(), wh.size() * sizeof(_Float16))) {
            return 1;
        }
        c->unet_pass[pass].n_tiles = n_tiles;
        HIP_TRY(hipStreamSynchronize(c->stream)); // (w, b, wh go out of scope)
    }
    c->unet_ready = true;
    return 0;
}

namespace {
// the activation tensors of the current frame size: one-pixel border, zero (only interiors are ever written)
int unet_tensors(rayhip_ctx *c) {
    if (c->unet_w == c->w && c->unet_h == c->h) {
        return 0;
    }
    const int wr = round_up16(c->w), hr = round_up16(c->h);
    c->unet_w = c->unet_h = 0; // (a failure part-way leaves tensors of two frame sizes: none of them counts as sized)
    for (int t = 0; t < 15; ++t) {
        const size_t n = size_t(wr / UNET_TENSOR_DIV[t] + 2) * size_t(hr / UNET_TENSOR_DIV[t] + 2) * size_t(UNET_TENSOR_CH[t]);
        c->unet_tensor[t].release(); // a fresh, zeroed allocation: the borders must be zero
        if (c->unet_tensor[t].alloc(n * sizeof(float))) {
            return 1;
        }
    }
    c->unet_images.release(); // the three images as a tensor (zero outside the image: never written there)
    if (c->unet_images.alloc(size_t(wr + 2) * size_t(hr + 2) * size_t(rt::unet::CHUNK) * sizeof(float))) {
        return 1;
    }
    c->unet_w = c->w, c->unet_h = c->h;
    return 0;
}
float *unet_interior(rayhip_ctx *c, const int t) {
    const int wr = round_up16(c->w);
    return c->unet_tensor[t].as<float>() + size_t(wr / UNET_TENSOR_DIV[t] + 3) * size_t(UNET_TENSOR_CH[t]);
}
// ... of the f16 form
int unet_tensors_h(rayhip_ctx *c) {
    if (c->unet_h_w == c->w && c->unet_h_h == c->h) {
        return 0;
    }
    const int wr = round_up16(c->w), hr = round_up16(c->h);
    c->unet_h_w = c->unet_h_h = 0;
    for (int t = 0; t < 15; ++t) {
        const size_t n = size_t(wr / UNET_TENSOR_DIV[t] + 2) * size_t(hr / UNET_TENSOR_DIV[t] + 2) * size_t(UNET_TENSOR_CH[t]);
        c->unet_tensor_h[t].release(); // a fresh, zeroed allocation: the borders must be zero
        if (c->unet_tensor_h[t].alloc(n * sizeof(uint16_t) + 64)) { // (+ 64: a 16-byte piece read of the last border pixel's upper channels stays inside)
            return 1;
        }
    }
    c->unet_images_h.release();
    if (c->unet_images_h.alloc(size_t(wr + 2) * size_t(hr + 2) * size_t(rt::unet::CHUNK) * sizeof(uint16_t) + 64)) {
        return 1;
    }
    c->unet_h_w = c->w, c->unet_h_h = c->h;
    return 0;
}
uint16_t *unet_interior_h(rayhip_ctx *c, const int t) {
    const int wr = round_up16(c->w);
    return c->unet_tensor_h[t].as<uint16_t>() + size_t(wr / UNET_TENSOR_DIV[t] + 3) * size_t(UNET_TENSOR_CH[t]);
}
} // namespace

int rayhip_denoise_unet(rayhip_ctx *c, const rayhip_camera *cam, const int rect[4], int pass) {
    if (use_device(c)) {
        return 1;
    }
    if (!c->unet_ready) {
        return fail("rayhip_denoise_unet before rayhip_unet_init (RendererBase::InitUNetFilter)");
    }
    if (!c->w) {
        return fail("rayhip_denoise_unet before rayhip_resize");
    }
    if (pass < -1 || pass > 15) {
        return fail("the UNet filter has passes 0 .. 15 (-1: all of them)");
    }
    if (rect[0] < 0 || rect[1] < 0 || rect[2] <= 0 || rect[3] <= 0 || rect[0] + rect[2] > c->w || rect[1] + rect[3] > c->h) {
        return fail("rect outside the frame");
    }
    if ((rect[0] % 16) != 0 || (rect[1] % 16) != 0) {
        return fail("the UNet filter works on regions whose corner is a multiple of 16 pixels (the network pools four times)");
    }
    if (cam->view_transform != 0 && cam->view_transform != c->lut_transform) {
        return fail("view transform %d needs its look-up table: rayhip_set_tonemap_lut", int(cam->view_transform));
    }
    if (c->unet_half ? unet_tensors_h(c) : unet_tensors(c)) {
        return 1;
    }
    const int wr = round_up16(c->w), hr = round_up16(c->h);
    for (int p = (pass < 0 ? 0 : pass); c->unet_half && p <= (pass < 0 ? 15 : pass); ++p) { // the f16 form (unet.h: ConvParamsH): the same schedule
        const UNetPassDesc &d = UNET_PASSES[p];
        int rx = rect[0], ry = rect[1], rw = rect[2], rh = rect[3];
        if (p < 15) {
            rw = round_up16(rw), rh = round_up16(rh);
        }
        rx /= d.div, ry /= d.div, rw = (rw + d.div - 1) / d.div, rh = (rh + d.div - 1) / d.div;
        rt::unet::ConvParamsH cp = {};
        if (d.a >= 0) {
            cp.a = unet_interior_h(c, d.a), cp.a_stride = wr / UNET_TENSOR_DIV[d.a] + 2, cp.a_ch = d.a_ch, cp.a_up = d.up;
        }
        if (d.b >= 0) {
            cp.b = unet_interior_h(c, d.b), cp.b_stride = wr / UNET_TENSOR_DIV[d.b] + 2, cp.b_ch = d.b_ch;
        }
        if (d.img) {
            uint16_t *img16 = c->unet_images_h.as<uint16_t>() + size_t(wr + 3) * size_t(rt::unet::CHUNK);
            if (pass >= 0 || p == 0) { // (all passes in one call: the tensor pass 0 made is still there for dec_conv1a -- nothing in between touches the images)
                HIP_TRY(rt::unet::launch_image_inputs_h(c->px.full, c->px.base_color, c->px.depth_normals, c->w, c->h, img16, wr + 2,
                                                        grid_for(c, size_t(c->w) * size_t(c->h), 256), c->stream));
            }
            if (d.a >= 0) {
                cp.b = img16, cp.b_stride = wr + 2, cp.b_ch = rt::unet::CHUNK;
            } else {
                cp.a = img16, cp.a_stride = wr + 2, cp.a_ch = rt::unet::CHUNK, cp.a_up = 0;
            }
        }
        cp.weights = c->unet_pass[p].weights_h.p, cp.bias = c->unet_pass[p].bias.as<float>();
        cp.x0 = rx, cp.y0 = ry, cp.w = rw, cp.h = rh;
        cp.in_w = wr / d.div, cp.in_h = hr / d.div;
        cp.pool = d.pool;
        if (d.out >= 0) {
            cp.out = unet_interior_h(c, d.out), cp.out_stride = wr / UNET_TENSOR_DIV[d.out] + 2, cp.out_ch = d.cout;
        } else {
            cp.out = c->px.raw, cp.out_stride = c->w, cp.out_ch = 3, cp.final_image = 1;
        }
        HIP_TRY(rt::unet::launch_conv_h(cp, c->unet_pass[p].n_tiles, c->stream));
        if (p == 15) {
            AccumParams tone = make_accum_params(*cam, c->w, rect, 1, c->shard);
            tone.lut = c->tonemap_lut.as<uint32_t>(), tone.lut_dims = c->lut_dims;
            k_tonemap_raw_rect<<<grid_for(c, size_t(rect[2]) * size_t(rect[3]), 256), 256, 0, c->stream>>>(tone, c->px);
            HIP_TRY(hipGetLastError());
        }
    }
    for (int p = (pass < 0 ? 0 : pass); !c->unet_half && p <= (pass < 0 ? 15 : pass); ++p) {
        const UNetPassDesc &d = UNET_PASSES[p];
        // the region of this pass in its own resolution (RendererCPU.h:797-802 and the head of every case)
        int rx = rect[0], ry = rect[1], rw = rect[2], rh = rect[3];
        if (p < 15) {
            rw = round_up16(rw), rh = round_up16(rh);
        }
        rx /= d.div, ry /= d.div, rw = (rw + d.div - 1) / d.div, rh = (rh + d.div - 1) / d.div;
        rt::unet::ConvParams cp = {};
        if (d.a >= 0) {
            cp.a = unet_interior(c, d.a), cp.a_stride = wr / UNET_TENSOR_DIV[d.a] + 2, cp.a_ch = d.a_ch, cp.a_up = d.up;
        }
        if (d.b >= 0) {
            cp.b = unet_interior(c, d.b), cp.b_stride = wr / UNET_TENSOR_DIV[d.b] + 2, cp.b_ch = d.b_ch;
        }
        if (d.img) { // the three images as a 16-channel tensor: the first input of pass 0, the second one of dec_conv1a
            float *img16 = c->unet_images.as<float>() + size_t(wr + 3) * size_t(rt::unet::CHUNK);
            HIP_TRY(rt::unet::launch_image_inputs(c->px.full, c->px.base_color, c->px.depth_normals, c->w, c->h, img16, wr + 2,
                                                  grid_for(c, size_t(c->w) * size_t(c->h), 256), c->stream));
            if (d.a >= 0) {
                cp.b = img16, cp.b_stride = wr + 2, cp.b_ch = rt::unet::CHUNK;
            } else {
                cp.a = img16, cp.a_stride = wr + 2, cp.a_ch = rt::unet::CHUNK, cp.a_up = 0;
            }
        }
        cp.weights = c->unet_pass[p].weights.as<float>(), cp.bias = c->unet_pass[p].bias.as<float>();
        cp.x0 = rx, cp.y0 = ry, cp.w = rw, cp.h = rh;
        cp.in_w = wr / d.div, cp.in_h = hr / d.div;
        cp.pool = d.pool;
        if (d.out >= 0) {
            cp.out = unet_interior(c, d.out), cp.out_stride = wr / UNET_TENSOR_DIV[d.out] + 2, cp.out_ch = d.cout;
        } else {
            cp.out = reinterpret_cast<float *>(c->px.raw), cp.out_stride = c->w, cp.out_ch = 3, cp.final_image = 1;
        }
        HIP_TRY(rt::unet::launch_conv(cp, c->unet_pass[p].n_tiles, c->stream));
        if (p == 15) {
            AccumParams tone = make_accum_params(*cam, c->w, rect, 1, c->shard);
            tone.lut = c->tonemap_lut.as<uint32_t>(), tone.lut_dims = c->lut_dims;
            k_tonemap_raw_rect<<<grid_for(c, size_t(rect[2]) * size_t(rect[3]), 256), 256, 0, c->stream>>>(tone, c->px);
            HIP_TRY(hipGetLastError());
        }
    }
    return 0;
}

int rayhip_unet_set_precision(rayhip_ctx *c, int half) {
    if (!c || (half != 0 && half != 1)) {
        return fail("rayhip_unet_set_precision: 0 (f32, exact) or 1 (f16 tensors and weights, f32 accumulate)");
    }
    c->unet_half = half;
    return 0;
}

// test hook: one activation tensor (0 .. 14, the order of unet_filter_tensors_t) with its border, NHWC; dims = {rows, columns, channels}
int rayhip_unet_read_tensor(rayhip_ctx *c, int which, float *dst, size_t capacity_floats, int out_dims[3]) {
    if (use_device(c)) {
        return 1;
    }
    if (which < 0 || which > 14 || (c->unet_half ? (c->unet_h_w != c->w || !c->unet_tensor_h[which].p) : (c->unet_w != c->w || !c->unet_tensor[which].p))) {
        return fail("rayhip_unet_read_tensor: no such tensor (run rayhip_denoise_unet first)");
    }
    const int wr = round_up16(c->w), hr = round_up16(c->h);
    out_dims[0] = hr / UNET_TENSOR_DIV[which] + 2, out_dims[1] = wr / UNET_TENSOR_DIV[which] + 2, out_dims[2] = UNET_TENSOR_CH[which];
    const size_t n = size_t(out_dims[0]) * out_dims[1] * out_dims[2];
    if (n > capacity_floats) {
        return fail("rayhip_unet_read_tensor: %zu floats needed", n);
    }
    if (c->unet_half) { // the f16 form's tensor, widened on the host
        std::vector<_Float16> h(n);
        HIP_TRY(hipMemcpyAsync(h.data(), c->unet_tensor_h[which].p, n * sizeof(_Float16), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        for (size_t i = 0; i < n; ++i) {
            dst[i] = float(h[i]);
        }
        return 0;
    }
    HIP_TRY(hipMemcpyAsync(dst, c->unet_tensor[which].p, n * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}
