// rayhip_upload.hip.h -- part of librayhip's host side (one translation unit: included by rayhip.hip, in this order, after the kernels):
// scene upload: lights and derived tables, the scene view, rayhip_scene_upload (validation, leaf refinement, wide collapse),
// rayhip_scene_update_instances (top level rebuilt on the device), filter table, tonemap LUT, the blob forms.
#pragma once

// the physical sky (rayhip_sky + its tables and textures: 1.6 MB): device copies and the view the kernels read; the directional-light
// list is part of it, so an instance / light update sends it again
static int upload_sky(rayhip_ctx *c, const rayhip_scene_desc *d) {
    c->sky_view = SkyView{};
    if (!(d->env.sky_map_spread_angle > 0.0f) || d->sky_count == 0) {
        return 0;
    }
    if (upload(c, c->sky_desc, d->sky, sizeof(rayhip_sky)) ||
        upload(c, c->sky_transmittance_lut, d->sky_transmittance_lut, size_t(d->sky_transmittance_lut_count) * sizeof(float)) ||
        upload(c, c->sky_multiscatter_lut, d->sky_multiscatter_lut, size_t(d->sky_multiscatter_lut_count) * sizeof(float)) ||
        upload(c, c->sky_dir_lights, d->sky_dir_lights, size_t(d->sky_dir_lights_count) * sizeof(uint32_t)) ||
        upload(c, c->sky_weather, d->sky_weather_tex, d->sky_weather_tex_count) || upload(c, c->sky_noise3d, d->sky_noise3d_tex, d->sky_noise3d_tex_count) ||
        upload(c, c->sky_curl, d->sky_curl_tex, d->sky_curl_tex_count) || upload(c, c->sky_moon, d->sky_moon_tex, d->sky_moon_tex_count) ||
        upload(c, c->sky_cirrus, d->sky_cirrus_tex, d->sky_cirrus_tex_count)) {
        return 1;
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    SkyView &v = c->sky_view;
    v.desc = c->sky_desc.as<rayhip_sky>();
    v.transmittance_lut = c->sky_transmittance_lut.as<float>(), v.multiscatter_lut = c->sky_multiscatter_lut.as<float>();
    v.dir_lights = c->sky_dir_lights.as<uint32_t>(), v.dir_lights_count = d->sky_dir_lights_count;
    v.weather = c->sky_weather.as<uint8_t>(), v.noise3d = c->sky_noise3d.as<uint8_t>(), v.curl = c->sky_curl.as<uint8_t>();
    v.moon = c->sky_moon.as<uint8_t>(), v.cirrus = c->sky_cirrus.as<uint8_t>();
    return 0;
}

// lights, their index list, the light tree (+ its per-node importance table) and the world-space corners of the triangle
// lights: everything an instance / light change replaces besides the top-level tree
// The sky environment map baked ON THE DEVICE (round 5; VERDICT round 4, missing 5): what Scene::PrepareSkyEnvMap produces on the host
// (SceneCPU.cpp:1017-1056 over CalcSkyEnvTexture, SceneCommon.cpp:286-361) and the reference's GPU scene in a compute pass (SceneGPU.h:1697-1768).
// Takes the sky part of a scene description (sky, its two tables, its five textures, sky_dir_lights) and the light array; writes w x h RGBE8 texels
// to host memory.  Uses buffers of its own: the scene that is on the device is not touched.
int rayhip_bake_sky(rayhip_ctx *c, const rayhip_scene_desc *d, int w, int h, uint32_t *out_rgbe8) {
    if (use_device(c)) {
        return 1;
    }
    if (!d || !out_rgbe8 || w <= 0 || h <= 0 || w > 16384 || h > 16384) {
        return fail("rayhip_bake_sky: bad arguments");
    }
    if (d->struct_size != sizeof(rayhip_scene_desc)) {
        return fail("rayhip_scene_desc::struct_size is %u, this library's struct has %zu bytes", d->struct_size, sizeof(rayhip_scene_desc));
    }
    if (d->lights_count != 0 && d->lights == nullptr) { // (before validate_sky: it looks at lights[sky_dir_lights[i]])
        return fail("rayhip_bake_sky: lights is null");
    }
    std::string why;
    if (!(d->env.sky_map_spread_angle > 0.0f) || !rayhip_validate::validate_sky(*d, why)) {
        return fail("rayhip_bake_sky: %s", why.empty() ? "the description holds no physical sky (env.sky_map_spread_angle > 0, sky, tables, textures)" : why.c_str());
    }
    DevBuf desc, tlut, mlut, dirs, weather, noise, curl, moon, cirrus, lights, out;
    DevBuf *all[] = {&desc, &tlut, &mlut, &dirs, &weather, &noise, &curl, &moon, &cirrus, &lights, &out};
    auto done = [&](int rc) {
        for (DevBuf *b : all) {
            b->release();
        }
        return rc;
    };
    if (upload(c, desc, d->sky, sizeof(rayhip_sky)) || upload(c, tlut, d->sky_transmittance_lut, size_t(d->sky_transmittance_lut_count) * sizeof(float)) ||
        upload(c, mlut, d->sky_multiscatter_lut, size_t(d->sky_multiscatter_lut_count) * sizeof(float)) ||
        upload(c, dirs, d->sky_dir_lights, size_t(d->sky_dir_lights_count) * sizeof(uint32_t)) || upload(c, weather, d->sky_weather_tex, d->sky_weather_tex_count) ||
        upload(c, noise, d->sky_noise3d_tex, d->sky_noise3d_tex_count) || upload(c, curl, d->sky_curl_tex, d->sky_curl_tex_count) ||
        upload(c, moon, d->sky_moon_tex, d->sky_moon_tex_count) || upload(c, cirrus, d->sky_cirrus_tex, d->sky_cirrus_tex_count) ||
        upload(c, lights, d->lights, size_t(d->lights_count) * sizeof(rayhip_light)) || out.alloc(size_t(w) * size_t(h) * sizeof(uint32_t))) {
        return done(1);
    }
    SkyView v = {};
    v.desc = desc.as<rayhip_sky>();
    v.transmittance_lut = tlut.as<float>(), v.multiscatter_lut = mlut.as<float>();
    v.dir_lights = dirs.as<uint32_t>(), v.dir_lights_count = d->sky_dir_lights_count;
    v.weather = weather.as<uint8_t>(), v.noise3d = noise.as<uint8_t>(), v.curl = curl.as<uint8_t>(), v.moon = moon.as<uint8_t>(), v.cirrus = cirrus.as<uint8_t>();
    const size_t n = size_t(w) * size_t(h);
    k_bake_sky<<<unsigned((n + 63) / 64), 64, 0, c->stream>>>(v, lights.as<rayhip_light>(), w, h, out.as<uint32_t>());
    if (hipGetLastError() != hipSuccess || hipMemcpyAsync(out_rgbe8, out.p, n * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
        hipStreamSynchronize(c->stream) != hipSuccess) {
        (void)fail("rayhip_bake_sky: the bake failed on the device");
        return done(1);
    }
    return done(0);
}

// ... from a serialised scene (scene_blob.h): what tests and tools hold
int rayhip_bake_sky_blob(rayhip_ctx *c, const void *blob, size_t size, int w, int h, uint32_t *out_rgbe8) {
    rayhip_scene_desc d;
    rayhip_camera cam;
    const float *ft = nullptr;
    int ftn = 0;
    std::string err;
    if (!rayhip_blob::deserialize(blob, size, d, cam, &ft, &ftn, err, nullptr)) {
        return fail("%s", err.c_str());
    }
    return rayhip_bake_sky(c, &d, w, h, out_rgbe8);
}

static int upload_lights(rayhip_ctx *c, const rayhip_scene_desc *d) {
    if (upload(c, c->lights, d->lights, size_t(d->lights_count) * sizeof(*d->lights)) ||
        upload(c, c->li_indices, d->li_indices, size_t(d->li_indices_count) * sizeof(uint32_t)) ||
        upload(c, c->light_cwnodes, d->light_cwnodes, size_t(d->light_cwnodes_count) * sizeof(*d->light_cwnodes))) {
        return 1;
    }
    // node-only half of the light-tree importance, evaluated once per scene (shade_lights.h: decode_light_child)
    std::vector<float4> lc(size_t(d->light_cwnodes_count) * LIGHT_CHILDREN_STRIDE);
    for (uint32_t n = 0; n < d->light_cwnodes_count; ++n) {
        fill_light_children(d->light_cwnodes[n], &lc[size_t(n) * LIGHT_CHILDREN_STRIDE]);
    }
    if (upload(c, c->light_children, lc.data(), lc.size() * sizeof(float4))) {
        return 1;
    }
    // world-space corners of the TRI lights (shade_lights.h: fill_light_tri_geom)
    // (the light array is a sparse pool: only the slots li_indices[] names hold lights)
    std::vector<float4> tg(size_t(d->lights_count) * 4, make_float4(0.0f, 0.0f, 0.0f, 0.0f));
    for (uint32_t k = 0; k < d->li_indices_count; ++k) {
        const uint32_t i = d->li_indices[k];
        if (i >= d->lights_count) {
            return fail("li_indices[%u] = %u is outside the light array", k, i);
        }
        const rayhip_light &l = d->lights[i];
        if (light_type(l) == LIGHT_TYPE_TRI) {
            const uint32_t tri = float_as_uint(l.params[0]), mi = float_as_uint(l.params[1]);
            if (mi >= d->mesh_instances_count || size_t(tri) * 3 + 2 >= d->vtx_indices_count) {
                return fail("triangle light %u refers to triangle %u of instance %u: out of range", i, tri, mi);
            }
        }
        fill_light_tri_geom(l, d->mesh_instances, d->vtx_indices, d->vertices, &tg[size_t(i) * 4]);
    }
    if (upload(c, c->light_tri_geom, tg.data(), tg.size() * sizeof(float4))) {
        return 1;
    }
    HIP_TRY(hipStreamSynchronize(c->stream)); // `lc`, `tg` go out of scope
    return 0;
}


// the kernels' view of what is on the device (SceneView), after a full upload or an instance update
// distinct instances the top level of a (validated) scene holds: a leaf word of the top level stands for one instance (the reference's
// one-leaf tree is a root whose two links are the same leaf word, the second one behind a point box at the origin: Core.cpp:1191-1213)
static uint32_t count_top_level_instances(const rayhip_bvh2_node *nodes, const uint32_t nodes_count, const uint32_t root) {
    if (root == 0xffffffffu) {
        return 0;
    }
    std::vector<uint32_t> seen, todo(1, root);
    while (!todo.empty()) {
        const uint32_t w = todo.back();
        todo.pop_back();
        if ((w & BVH2_PRIM_COUNT_BITS) != 0) {
            const uint32_t mi = w & BVH2_PRIM_INDEX_BITS;
            if (std::find(seen.begin(), seen.end(), mi) == seen.end()) {
                if (seen.size() >= 2) {
                    return 3; // (more than one is all the caller asks)
                }
                seen.push_back(mi);
            }
        } else if (w < nodes_count) {
            todo.push_back(nodes[w].left_child), todo.push_back(nodes[w].right_child);
        }
    }
    return uint32_t(seen.size());
}

static void refresh_scene_view(rayhip_ctx *c, const rayhip_scene_desc *d, const uint32_t tlas_root, const rayhip_lbvh::Box &root_box,
                               const uint32_t live_instances) {
    SceneView &v = c->sc;
    // the pooled closest-hit kernel hands prepared rays from lane to lane; a ray can change lanes only while nothing is pending at the top
    // level, which is every ray of a scene with ONE instance (RAYHIP_POOL_ANY=1: the pooled kernel for any scene -- tests of its other path)
    c->pool_scene = (live_instances == 1 || getenv("RAYHIP_POOL_ANY") != nullptr) && d->mesh_instances_count < (1u << 24);
    v.nodes = c->nodes.as<rayhip_bvh2_node>(), v.tris = c->tris.as<rayhip_tri_accel>(), v.tri_pitch = c->tri_pitch, v.all_solid = getenv("RAYHIP_NO_ALL_SOLID") ? 0u : c->all_solid;
    v.tri_indices = c->tri_indices.as<uint32_t>(), v.tri_materials = c->tri_materials.as<rayhip_tri_mat_data>();
    v.materials = c->materials.as<rayhip_material>(), v.vertices = c->vertices.as<rayhip_vertex>();
    v.vtx_indices = c->vtx_indices.as<uint32_t>(), v.mesh_instances = c->mesh_instances.as<rayhip_mesh_instance>();
    v.lights = c->lights.as<rayhip_light>(), v.li_indices = c->li_indices.as<uint32_t>();
    v.light_children = c->light_children.as<float4>();
    v.light_tri_geom = c->light_tri_geom.as<float4>();
    v.tri_verts = c->tri_verts.as<float4>();
    v.tri_bitangents = c->tri_bitangents.as<float4>();
    v.env_qtree = c->env_qtree.as<float4>();
    for (int lod = 0, off = 0; lod < 16; ++lod) {
        v.env_qtree_offset[lod] = uint32_t(off);
        if (lod < d->env.qtree_levels) {
            off += 1 << (2 * (d->env.qtree_levels - 1 - lod));
        }
    }
    v.nodes4 = c->wide == 4 ? c->nodes4.as<Bvh4Node>() : nullptr;
    v.nodes8 = c->wide == 8 ? c->nodes8.as<Bvh8Node>() : nullptr;
    v.blas_root4 = c->wide ? c->blas_root4.as<uint32_t>() : nullptr;
    v.light_cwnodes = c->light_cwnodes.as<rayhip_light_cwbvh_node>(), v.textures = c->textures.as<rayhip_texture>();
    v.texels = c->texels.as<uint32_t>();
    memcpy(v.tex_table, c->tex_table, sizeof(v.tex_table));
    v.tex_flags = c->tex_flags;
    v.li_indices_count = d->li_indices_count;
    v.light_cwnodes_count = d->light_cwnodes_count;
    v.visible_lights_count = d->visible_lights_count;
    v.blocker_lights_count = d->blocker_lights_count;
    v.tlas_root = tlas_root;
    v.env = d->env;
    v.sky = c->sky_view;
    memcpy(c->bbox_min, d->bbox_min, 12), memcpy(c->bbox_max, d->bbox_max, 12);
    // ray-sort grid: true bounds of the TLAS root (Scene::GetBounds takes fminf for the max corner, SceneCPU.cpp:1553)
    for (int i = 0; i < 3; ++i) {
        const bool have = root_box.lo[i] <= root_box.hi[i];
        const float mn = have ? root_box.lo[i] : d->bbox_min[i], mx = have ? root_box.hi[i] : d->bbox_max[i];
        const float ext = mx - mn;
        c->sort_grid.root_min[i] = mn;
        c->sort_grid.inv_cell[i] = (ext > 0.0f && ext < 1e30f) ? 256.0f / ext : 0.0f;
    }
}


#define UPLOAD_TRACE(msg)                                                                                              \
    if (getenv("RAYHIP_TRACE_UPLOAD")) {                                                                               \
        fprintf(stderr, "rayhip_scene_upload: %8.1f ms  %s\n",                                                         \
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - upload_t0).count(), msg);  \
    }

int rayhip_scene_upload(rayhip_ctx *c, const rayhip_scene_desc *d_in) {
    if (use_device(c)) {
        return 1;
    }
    if (d_in->struct_size != sizeof(rayhip_scene_desc)) { // (before anything else of the struct is looked at)
        return fail("rayhip_scene_desc::struct_size is %u, this library's struct has %zu bytes (ABI version %d): the caller was built against another rayhip.h",
                    d_in->struct_size, sizeof(rayhip_scene_desc), RAYHIP_ABI_VERSION);
    }
    const auto upload_t0 = std::chrono::steady_clock::now();
    (void)upload_t0;
    UPLOAD_TRACE("begin")
    const rayhip_layout::AlignedDesc aligned(*d_in); // see bvh_layout.h
    const rayhip_scene_desc *d = &aligned.d;
    if (d->env.qtree_levels < 0 || d->env.qtree_levels > 16) {
        return fail("bad env-map quadtree depth %d", d->env.qtree_levels);
    }
    // (the physical sky, environment_t::sky_map_spread_angle > 0: narrow rays are evaluated analytically, rt_sky.h; what that needs
    // arrives as rayhip_scene_desc::sky* and is checked by the validation below)
    {
        size_t quads = 0;
        for (int lod = 0; lod < d->env.qtree_levels; ++lod) {
            quads += size_t(1) << (2 * (d->env.qtree_levels - 1 - lod));
        }
        if (size_t(d->env_qtree_count) != quads * 4) {
            return fail("env_qtree holds %u floats, %d levels need %zu", d->env_qtree_count, d->env.qtree_levels, quads * 4);
        }
    }
    bool all_sides_solid = false; // (over the triangles the trees reach: the pools are sparse, an unused slot is all zeros)
    { // every index a kernel would follow without a bound of its own (scene_validate.h)
        std::string why;
        if (!rayhip_validate::validate(*d, why, &all_sides_solid)) {
            return fail("%s", why.c_str());
        }
    }
    UPLOAD_TRACE("validated")
    // Leaf refinement (scene_rebuild.h): the scene's trees are kept, every leaf with more than `leaf_max` triangles is replaced
    // by a subtree of the linear builder.  RAYHIP_REFINE_LEAVES=<leaf_max> (0 = leave the trees as they are; default 2).
    // RAYHIP_REBUILD_BVH=<leaf_max>: both levels rebuilt from the triangles and instance transforms instead (lbvh.h).
    rayhip_rebuild::Rebuilt rebuilt;
    rayhip_scene_desc d_rebuilt = *d;
    {
        int refine = 2, rebuild = 0;
        if (const char *e = getenv("RAYHIP_REFINE_LEAVES")) {
            refine = std::max(0, std::min(8, atoi(e)));
        }
        if (const char *e = getenv("RAYHIP_REBUILD_BVH")) {
            rebuild = std::max(0, std::min(8, atoi(e)));
        }
        if (rebuild > 0 || refine > 0) {
            // the builder itself runs on the device (lbvh.hip.h); RAYHIP_BVH_BUILD_ON_HOST=1 runs the same element functions
            // as host loops instead (A/B and debugging: the two produce identical arrays)
            const bool on_host = getenv("RAYHIP_BVH_BUILD_ON_HOST") != nullptr && atoi(getenv("RAYHIP_BVH_BUILD_ON_HOST")) != 0;
            auto build = [&](const rayhip_lbvh::Input &in, rayhip_lbvh::Output &out, std::string &why) {
                if (on_host) {
                    out = rayhip_lbvh::build_host(in);
                    return true;
                }
                return rayhip_lbvh::build_device(c->stream, in, out, why);
            };
            rebuilt = rebuild > 0 ? rayhip_rebuild::rebuild_with(*d, uint32_t(rebuild), build) : rayhip_rebuild::refine_with(*d, uint32_t(refine), build);
            UPLOAD_TRACE(rebuild > 0 ? "both levels rebuilt" : "leaves refined")
            if (!rebuilt.ok) {
                return fail("acceleration-structure %s failed: %s", rebuild > 0 ? "rebuild" : "refinement", rebuilt.why.c_str());
            }
            d_rebuilt.nodes = rebuilt.nodes.data(), d_rebuilt.nodes_count = uint32_t(rebuilt.nodes.size());
            d_rebuilt.tris = rebuilt.tris.data(), d_rebuilt.tris_count = uint32_t(rebuilt.tris.size());
            d_rebuilt.tri_indices = rebuilt.tri_indices.data(), d_rebuilt.tri_indices_count = uint32_t(rebuilt.tri_indices.size());
            d_rebuilt.mesh_instances = rebuilt.mesh_instances.data();
            d_rebuilt.tlas_root = rebuilt.tlas_root;
            d = &d_rebuilt;
            std::string why;
            if (!rayhip_validate::validate(*d, why)) {
                return fail("rebuilt scene: %s", why.c_str());
            }
        }
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
#define UP(field)                                                                                                      \
    if (upload(c, c->field, d->field, size_t(d->field##_count) * sizeof(*d->field))) {                                 \
        return 1;                                                                                                      \
    }
    // HBM layout pass (bvh_layout.h): depth-first node order with sibling pairs in one 128-byte line, triangles in
    // leaf-visit order.  Off by default since round 3 (RAYHIP_LAYOUT=1 switches it on): the kernels walk the 4-wide collapse,
    // whose node order is the collapse's own, and the triangle records come out of the leaf refinement grouped leaf by leaf
    // in the order of a depth-first walk already -- measured, Bistro-class scene: K2 2.11 ms per iteration with the pass,
    // 2.13 without, Sponza-class 1.70 / 1.70 (profiles/r03/experiments/variants_layout_*.txt) -- for 181 ms of host time
    // per upload.  Without leaf refinement (RAYHIP_REFINE_LEAVES=0) the pass still runs: the reference builder's order is poor.
    rayhip_layout::Result lay;
    {
        const char *e = getenv("RAYHIP_LAYOUT"), *off = getenv("RAYHIP_NO_LAYOUT");
        const bool want_layout = e ? e[0] == '1' : !rebuilt.ok;
        if (want_layout && !(off && off[0] == '1')) {
            lay = rayhip_layout::optimize(*d);
        }
    }
    UPLOAD_TRACE(lay.applied ? "layout applied" : lay.why_not)
    uint32_t tlas_root = d->tlas_root;
    { // room behind the nodes for top-level trees rebuilt on the device later (rayhip_scene_update_instances)
        const size_t n_now = lay.applied ? lay.nodes.size() : size_t(d->nodes_count);
        c->nodes_used = uint32_t(n_now);
        c->nodes_reserved = uint32_t(std::max<size_t>(8192, 8 * size_t(d->mesh_instances_count)));
        c->tlas_half = 0;
        if (c->nodes.alloc((n_now + c->nodes_reserved) * sizeof(rayhip_bvh2_node))) {
            return 1;
        }
    }
    if (lay.applied) {
        tlas_root = lay.tlas_root;
    }
    // Wide quantised BLAS trees over the node order just decided.  RAYHIP_BVH_WIDTH: 8 (default; rt_bvh8.h -- it also decides the
    // order of the triangle records and re-bases the BVH2's leaf words onto it), 4 (rt_bvh4.h: round 2's form), 2 keeps the
    // kernels on the reference's BVH2 (RAYHIP_NO_BVH4=1 says the same; A/B measurements)
    int wide = 0;
    std::vector<uint32_t> blas_root4;
    {
        // Default 4: measured on the MI355X (profiles/r03/experiments/variants_bvh8.txt) the 8-wide walk performs 29 % fewer node
        // visits and 17 % more triangle tests per ray and takes the same time -- 2.17 vs 2.11 ms per iteration on the Bistro-class
        // scene, within 1 % on the other workloads: the kernel is bound by random cache-line fetches per second, and an 80-byte
        // node costs two 64-byte sectors.  The narrower form needs no dynamic-programming collapse at upload (0.5 s) either.
        int want = 4;
        if (const char *e = getenv("RAYHIP_BVH_WIDTH")) {
            want = atoi(e);
        }
        if (const char *e = getenv("RAYHIP_NO_BVH4")) {
            want = e[0] == '1' ? 2 : want;
        }
        std::vector<rayhip_bvh2_node> nodes2_own; // a copy the 8-wide build may re-base (the caller's arrays are const)
        rayhip_bvh2_node *n2 = nullptr;
        if (lay.applied) {
            n2 = lay.nodes.data();
        } else {
            nodes2_own.assign(d->nodes, d->nodes + d->nodes_count);
            n2 = nodes2_own.data();
        }
        const uint32_t n2_count = lay.applied ? uint32_t(lay.nodes.size()) : d->nodes_count;
        const rayhip_mesh_instance *mis = lay.applied ? lay.mesh_instances.data() : d->mesh_instances;
        const rayhip_tri_accel *tris_in = lay.applied ? lay.tris.data() : d->tris;
        const uint32_t *tri_indices_in = lay.applied ? lay.tri_indices.data() : d->tri_indices;
        size_t n_tris = lay.applied ? lay.tris.size() : size_t(d->tris_count);
        rayhip_bvh8::Result b8;
        if (want == 8) {
            b8 = rayhip_bvh8::build(n2, n2_count, mis, d->mesh_instances_count, tlas_root, tris_in, tri_indices_in, uint32_t(n_tris));
            UPLOAD_TRACE(b8.ok ? "bvh8 built" : b8.why_not)
        }
        if (b8.ok && !b8.nodes.empty()) {
            tris_in = b8.tris.data(), tri_indices_in = b8.tri_indices.data(), n_tris = b8.tris.size();
        }
        if (upload(c, c->nodes, n2, size_t(n2_count) * sizeof(rayhip_bvh2_node)) ||
            upload(c, c->tris, tris_in, n_tris * sizeof(rayhip_tri_accel)) ||
            upload(c, c->tri_indices, tri_indices_in, n_tris * sizeof(uint32_t)) ||
            upload(c, c->mesh_instances, mis, size_t(d->mesh_instances_count) * sizeof(rayhip_mesh_instance))) {
            return 1;
        }
        // the walks' triangle table: the reference's 48-byte array as it is; RAYHIP_TRI_PITCH=64 re-pitches it so that every record lies in
        // its own 64-byte sector (half of the 48-byte records straddle two) -- measured neutral (K2 2.14 against 2.12 ms,
        // profiles/r03/experiments/variants_tripitch.txt: the kernel is bound by instruction issue, not by sectors), so it stays an option
        c->tri_pitch = 3;
        if (n_tris && getenv("RAYHIP_TRI_PITCH") && atoi(getenv("RAYHIP_TRI_PITCH")) == 64) {
            DevBuf padded;
            if (padded.alloc(n_tris * 64)) {
                return 1;
            }
            k_pad_tris<<<unsigned((n_tris * 4 + 255) / 256), 256, 0, c->stream>>>(c->tris.as<float4>(), padded.as<float4>(), n_tris);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipStreamSynchronize(c->stream));
            c->tris.swap(padded);
            padded.release();
            c->tri_pitch = 4;
        }
        size_t wide_bytes = 0;
        if (b8.ok && !b8.nodes.empty()) {
            if (upload(c, c->nodes8, b8.nodes.data(), b8.nodes.size() * sizeof(Bvh8Node)) ||
                upload(c, c->blas_root4, b8.blas_root8.data(), b8.blas_root8.size() * sizeof(uint32_t))) {
                return 1;
            }
            wide = 8, blas_root4 = b8.blas_root8, wide_bytes = b8.nodes.size() * sizeof(Bvh8Node);
        } else if (want == 4 || want == 8) {
            // the collapse runs on the device over the nodes just uploaded (bvh4_build.hip.h); RAYHIP_BVH_BUILD_ON_HOST=1: the host
            // driver over the same element functions (A/B: the same tree in another node order)
            const bool on_host = getenv("RAYHIP_BVH_BUILD_ON_HOST") != nullptr && atoi(getenv("RAYHIP_BVH_BUILD_ON_HOST")) != 0;
            if (on_host) {
                rayhip_bvh4::Result b4 = rayhip_bvh4::build(n2, n2_count, mis, d->mesh_instances_count, tlas_root);
                if (b4.ok && !b4.nodes.empty()) {
                    if (upload(c, c->nodes4, b4.nodes.data(), b4.nodes.size() * sizeof(Bvh4Node)) ||
                        upload(c, c->blas_root4, b4.blas_root4.data(), b4.blas_root4.size() * sizeof(uint32_t))) {
                        return 1;
                    }
                    HIP_TRY(hipStreamSynchronize(c->stream));
                    wide = 4, blas_root4 = b4.blas_root4, wide_bytes = b4.nodes.size() * sizeof(Bvh4Node);
                }
            } else {
                std::vector<uint32_t> roots;
                uint32_t n_wide = 0;
                std::string why;
                if (rayhip_bvh4::collect_roots(n2, n2_count, mis, d->mesh_instances_count, tlas_root, roots, blas_root4) && !roots.empty()) {
                    if (c->nodes4.alloc(size_t(n2_count) * sizeof(Bvh4Node))) {
                        return 1;
                    }
                    bool unquantisable = false;
                    if (!rayhip_bvh4::build_device(c->stream, c->nodes.as<rayhip_bvh2_node>(), n2_count, roots, c->nodes4.as<Bvh4Node>(), n_wide, why, &unquantisable)) {
                        if (!unquantisable) {
                            return fail("4-wide collapse failed: %s", why.c_str()); // (a HIP error)
                        }
                        // a box the grid cannot hold: the kernels walk the BVH2 for this scene, as the host builder documents (bvh4_build.h)
                        blas_root4.clear();
                    } else {
                        if (upload(c, c->blas_root4, blas_root4.data(), blas_root4.size() * sizeof(uint32_t))) {
                            return 1;
                        }
                        wide = 4, wide_bytes = size_t(n_wide) * sizeof(Bvh4Node);
                    }
                } else {
                    blas_root4.clear();
                }
            }
            UPLOAD_TRACE(wide == 4 ? "bvh4 built" : "no wide BLAS")
        }
        HIP_TRY(hipStreamSynchronize(c->stream)); // the builders' arrays go out of scope
        // "small": the BLAS working set (nodes + triangle records) fits one XCD's 4 MB L2 with room to spare
        c->small_scene = wide != 0 && getenv("RAYHIP_NO_SMALL") == nullptr && wide_bytes + n_tris * sizeof(rayhip_tri_accel) <= (size_t(2) << 20);
        // the meshes in use, for rayhip_scene_update_instances: the roots of their trees as uploaded
        rayhip_update::collect_mesh_refs(n2, n2_count, mis, d->mesh_instances_count, tlas_root, wide ? blas_root4.data() : nullptr, c->mesh_refs);
    }
    UPLOAD_TRACE("bvh uploaded")
    UP(tri_materials)
    // is there a triangle side that is not plainly solid?  (the closest-hit kernels skip the per-hit material fetch when not; round 4: judged
    // over the reachable triangles -- the headline scene's pool has unused slots, and the flag had never been set for it)
    c->all_solid = all_sides_solid ? 1u : 0u;
    UPLOAD_TRACE(all_sides_solid ? "every reachable triangle side is solid" : "some triangle sides are not solid")
    UP(materials)
    { // can any surface change a ray's stack of refractive indices?  (ShadeParams::plain_ior: if not, the passes leave the rays' ior plane alone)
        bool refracts = false;
        for (uint32_t i = 0; i < d->materials_count; ++i) {
            const rayhip_material &m = d->materials[i];
            refracts |= m.type == NODE_REFRACTIVE || (m.type == NODE_PRINCIPLED && m.transmission_unorm != 0);
        }
        c->plain_ior = !refracts && !getenv("RAYHIP_NO_PLAIN_IOR");
        UPLOAD_TRACE(refracts ? "refractive surfaces: rays carry their ior stacks" : "no refractive surface: the ior plane of the rays is not used")
    }
    UP(vertices)
    UP(vtx_indices)
    { // vertices gathered per triangle (shade_point.h: fill_tri_verts), on the device from the arrays just uploaded
        const uint32_t n_tris = d->vtx_indices_count / 3;
        if (c->tri_verts.alloc(size_t(n_tris) * TRI_VERTS_STRIDE * sizeof(float4)) ||
            c->tri_bitangents.alloc(size_t(n_tris) * TRI_BITANGENTS_STRIDE * sizeof(float4))) {
            return 1;
        }
        if (n_tris) {
            k_fill_tri_verts<<<(n_tris + 255) / 256, 256, 0, c->stream>>>(c->vertices.as<rayhip_vertex>(), d->vertices_count,
                                                                          c->vtx_indices.as<uint32_t>(), n_tris,
                                                                          c->tri_materials.as<rayhip_tri_mat_data>(), d->tri_materials_count,
                                                                          c->tri_verts.as<float4>(), c->tri_bitangents.as<float4>());
            HIP_TRY(hipGetLastError());
        }
    }
    UPLOAD_TRACE("tri_verts done")
    if (upload_lights(c, d)) {
        return 1;
    }
    UPLOAD_TRACE("lights done")
    UP(textures)
    UP(texels)
    UP(env_qtree)
#undef UP
    HIP_TRY(hipStreamSynchronize(c->stream)); // host arrays may go away after this call
    c->wide = wide;
    memcpy(c->tex_table, d->tex_table, sizeof(c->tex_table));
    c->textures_count = d->textures_count;
    c->tex_flags = d->texture_flags;
    c->geometry = {d->vertices_count, d->vtx_indices_count, d->tri_materials_count, d->materials_count};
    {
        rayhip_lbvh::Box root_box = rayhip_lbvh::empty_box();
        if (d->tlas_root != 0xffffffffu && d->tlas_root < d->nodes_count) {
            root_box = rayhip_rebuild::node_box(d->nodes[d->tlas_root]);
        }
        if (upload_sky(c, d)) {
            return 1;
        }
        refresh_scene_view(c, d, tlas_root, root_box, count_top_level_instances(d->nodes, d->nodes_count, d->tlas_root));
    }
    c->have_scene = true;
    c->census_valid = false; // (another scene: its queues fill differently; the next pass runs at full grids and takes a new census)
    UPLOAD_TRACE("done")
    return 0;
}

// ---- instance / light / environment update without a new upload of the geometry -----------------------------------------
// What SceneBase::SetMeshInstanceTransform / AddMeshInstance / RemoveMeshInstance / AddLight / RemoveLight / SetEnvironment /
// Finalize change (SceneCPU.cpp:1004-1094, 1103-1162 RebuildTLAS, 1411-1521 RebuildLightTree): the instance array, the
// top-level tree, the light arrays and the environment.  The top level is rebuilt ON THE DEVICE by the linear builder
// (lbvh.hip.h) over the instance boxes; the host's own top-level tree in `d` (node numbering of the host arrays, which the
// device does not share after the layout pass) only tells which instance slots are alive and their world-space boxes.
// Returns 0, 1 = error, 2 = the scene needs rayhip_scene_upload (an instance of a mesh that is not on the device, geometry
// arrays of another size, no room for the tree).
int rayhip_scene_bvh_width(rayhip_ctx *c) { return !c || !c->have_scene ? 0 : c->wide ? c->wide : 2; }

int rayhip_closest_hit_form(rayhip_ctx *c) {
    if (!c || !c->have_scene || !c->wide || !c->refill_waves) {
        return 0;
    }
    return (c->wide == 4 && c->refill_pool && c->pool_scene) ? 2 : 1;
}

int rayhip_scene_update_instances(rayhip_ctx *c, const rayhip_scene_desc *d) {
    if (use_device(c)) {
        return 1;
    }
    if (d->struct_size != sizeof(rayhip_scene_desc)) {
        return fail("rayhip_scene_desc::struct_size is %u, this library's struct has %zu bytes (ABI version %d): the caller was built against another rayhip.h",
                    d->struct_size, sizeof(rayhip_scene_desc), RAYHIP_ABI_VERSION);
    }
    if (!c->have_scene) {
        (void)fail("rayhip_scene_update_instances before rayhip_scene_upload");
        return 2;
    }
    const auto upload_t0 = std::chrono::steady_clock::now();
    (void)upload_t0;
    if (d->vertices_count != c->geometry.vertices || d->vtx_indices_count != c->geometry.vtx_indices ||
        d->tri_materials_count != c->geometry.tri_materials || d->materials_count != c->geometry.materials) {
        (void)fail("geometry arrays changed size since the last upload");
        return 2;
    }
    if (d->env.qtree_levels < 0 || d->env.qtree_levels > 16) {
        return fail("bad env-map quadtree depth %d", d->env.qtree_levels);
    }
    {
        std::string why;
        if (!rayhip_validate::validate_sky(*d, why)) {
            return fail("%s", why.c_str());
        }
    }
    {
        size_t quads = 0;
        for (int lod = 0; lod < d->env.qtree_levels; ++lod) {
            quads += size_t(1) << (2 * (d->env.qtree_levels - 1 - lod));
        }
        if (size_t(d->env_qtree_count) != quads * 4) {
            return fail("env_qtree holds %u floats, %d levels need %zu", d->env_qtree_count, d->env.qtree_levels, quads * 4);
        }
        for (const uint32_t handle : {d->env.env_map, d->env.back_map}) {
            if (handle != 0xffffffffu &&
                ((handle >> 28) >= 8u || uint64_t(c->tex_table[handle >> 28]) + (handle & 0x00ffffffu) >= c->textures_count)) {
                return fail("environment map handle outside the texture table on the device");
            }
        }
    }
    rayhip_update::Plan up;
    {
        std::string why;
        const int rc = rayhip_update::plan(*d, c->mesh_refs, up, why);
        if (rc) {
            (void)fail("%s", why.c_str());
            return rc;
        }
    }
    const std::vector<uint32_t> &live = up.live;
    std::vector<rayhip_mesh_instance> &mis = up.instances;
    std::vector<uint32_t> &root4 = up.root4;
    {
        rayhip_scene_desc lights_only = *d;
        lights_only.mesh_instances = mis.data();
        std::string why;
        if (!rayhip_validate::validate_lights(lights_only, why)) {
            return fail("%s", why.c_str());
        }
    }
    uint32_t tlas_root = 0xffffffffu;
    rayhip_lbvh::Box root_box = rayhip_lbvh::empty_box();
    if (!live.empty()) {
        const std::vector<uint32_t> group(live.size(), 0);
        const rayhip_lbvh::Input ti = rayhip_update::top_level_input(up, group);
        rayhip_lbvh::Output tlas;
        std::string why;
        if (!rayhip_lbvh::build_device(c->stream, ti, tlas, why)) {
            return fail("top-level build failed: %s", why.c_str());
        }
        // two halves, used in turn: the tree the scene view still points at is never overwritten, so a failure further
        // down (rc 1) leaves a context that renders the previous top level
        const uint32_t half = c->nodes_reserved / 2;
        if (tlas.nodes.size() > half || tlas.group_root.empty() || tlas.group_root[0] == 0xffffffffu) {
            (void)fail("no room for a top-level tree of %zu nodes", tlas.nodes.size());
            return 2;
        }
        const uint32_t base = c->nodes_used + c->tlas_half * half;
        c->tlas_half ^= 1u;
        tlas_root = rayhip_update::relocate_top_level(tlas, up, base);
        root_box = tlas.bounds;
        UPLOAD_TRACE("top level built")
        // pending passes read the old tree: the caller flushed (RendererHIP) or synchronises through the stream order here
        HIP_TRY(hipMemcpyAsync(c->nodes.as<rayhip_bvh2_node>() + base, tlas.nodes.data(), tlas.nodes.size() * sizeof(rayhip_bvh2_node),
                               hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream)); // `tlas` goes out of scope
    }
    if (upload(c, c->mesh_instances, mis.data(), mis.size() * sizeof(rayhip_mesh_instance)) ||
        (c->wide && upload(c, c->blas_root4, root4.data(), root4.size() * sizeof(uint32_t)))) {
        return 1;
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    {
        rayhip_scene_desc with_roots = *d; // triangle lights are placed by their instance's transform only
        if (upload_lights(c, &with_roots) ||
            upload(c, c->env_qtree, d->env_qtree, size_t(d->env_qtree_count) * sizeof(float))) {
            return 1;
        }
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (upload_sky(c, d)) {
        return 1;
    }
    refresh_scene_view(c, d, tlas_root, root_box, uint32_t(live.size()));
    UPLOAD_TRACE("instances updated")
    return 0;
}

int rayhip_set_filter_table(rayhip_ctx *c, const float *table, int count) {
    if (use_device(c)) {
        return 1;
    }
    if (count != FILTER_TABLE_SIZE) {
        return fail("filter table must have %d entries", FILTER_TABLE_SIZE);
    }
    if (upload(c, c->filter_table, table, size_t(count) * 4)) {
        return 1;
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}

int rayhip_set_tonemap_lut(rayhip_ctx *c, int view_transform, const uint32_t *lut, int dims) {
    if (use_device(c)) {
        return 1;
    }
    if (view_transform <= 0 || !lut || dims < 2 || dims > 256) {
        return fail("bad tonemap table (view transform %d, dims %d)", view_transform, dims);
    }
    const size_t bytes = size_t(dims) * size_t(dims) * size_t(dims) * sizeof(uint32_t);
    HIP_TRY(hipStreamSynchronize(c->stream)); // a pass in flight may still read the old table
    if (c->tonemap_lut.alloc(bytes)) {
        return 1;
    }
    HIP_TRY(hipMemcpy(c->tonemap_lut.p, lut, bytes, hipMemcpyHostToDevice));
    c->lut_transform = view_transform, c->lut_dims = dims;
    return 0;
}

int rayhip_scene_upload_blob(rayhip_ctx *c, const void *blob, size_t size, rayhip_camera *out_cam) {
    rayhip_scene_desc d;
    const float *ft = nullptr;
    int ftn = 0;
    std::string err;
    rayhip_blob::Extras extras;
    if (!rayhip_blob::deserialize(blob, size, d, *out_cam, &ft, &ftn, err, &extras)) {
        return fail("%s", err.c_str());
    }
    if (extras.tonemap_lut && out_cam->view_transform != 0 &&
        rayhip_set_tonemap_lut(c, out_cam->view_transform, extras.tonemap_lut, extras.tonemap_lut_dims)) {
        return 1;
    }
    if (rayhip_scene_upload(c, &d)) {
        return 1;
    }
    if (ft && rayhip_set_filter_table(c, ft, ftn)) {
        return 1;
    }
    return 0;
}

int rayhip_scene_update_instances_blob(rayhip_ctx *c, const void *blob, size_t size, rayhip_camera *out_cam) {
    rayhip_scene_desc d;
    const float *ft = nullptr;
    int ftn = 0;
    std::string err;
    rayhip_blob::Extras extras;
    if (!rayhip_blob::deserialize(blob, size, d, *out_cam, &ft, &ftn, err, &extras)) {
        return fail("%s", err.c_str());
    }
    return rayhip_scene_update_instances(c, &d);
}
