// comm.hip.h -- the one exchange step of the path: reduce the tile-sharded framebuffers of N GPUs to one rank over
// RCCL / xGMI (rayhip_comm_create / rayhip_comm_reduce_framebuffers / rayhip_comm_destroy; 8e).
// Included by rayhip.hip (needs rayhip_ctx).
//
// What is exchanged: every rank packs the pixels it OWNS (rayhip_set_shard) of the running mean (`full`), of the two aux
// images and of the variance estimate into one staging buffer -- zero elsewhere -- and ONE ncclReduce(sum, fp32) of
// 4 x W x H x 16 B (133 MB at 1080p; 33 MB when only the radiance image is asked for) lands the frame on the root, which
// unpacks it into its own buffers and re-runs the tonemap.  A sum with zeros is exact, so the assembled images equal an
// unsharded render bit for bit, and DenoiseImage on the root sees the complete guides.  The ranks' accumulation state is
// never overwritten on pixels they own, so progressive refinement (render more, reduce again) stays exact.
//
// RCCL is loaded with dlopen when the first communicator is created: librayhip.so has no link-time dependency on it, and
// inside a process that already carries an RCCL (PyTorch) the loaded copy is reused.
#pragma once

#include <dlfcn.h>
#include <rccl/rccl.h> // types only; every call goes through the table below

namespace {
struct RcclApi {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Reduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
RcclApi g_rccl;

int load_rccl() {
    if (g_rccl.lib) {
        return 0;
    }
    const char *names[] = {getenv("RAYHIP_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *lib = nullptr;
    for (const char *n : names) {
        if (n && (lib = dlopen(n, RTLD_NOW | RTLD_LOCAL)) != nullptr) {
            break;
        }
    }
    if (!lib) {
        return fail("RCCL not found (librccl.so.1): %s", dlerror());
    }
    RcclApi a;
    a.lib = lib;
#define RCCL_SYM(field, name)                                                                                           \
    a.field = reinterpret_cast<decltype(a.field)>(dlsym(lib, name));                                                    \
    if (!a.field) {                                                                                                     \
        return fail("RCCL lacks %s", name);                                                                             \
    }
    RCCL_SYM(GetUniqueId, "ncclGetUniqueId")
    RCCL_SYM(CommInitRank, "ncclCommInitRank")
    RCCL_SYM(CommInitAll, "ncclCommInitAll")
    RCCL_SYM(CommDestroy, "ncclCommDestroy")
    RCCL_SYM(Reduce, "ncclReduce")
    RCCL_SYM(GroupStart, "ncclGroupStart")
    RCCL_SYM(GroupEnd, "ncclGroupEnd")
    RCCL_SYM(GetErrorString, "ncclGetErrorString")
#undef RCCL_SYM
    g_rccl = a;
    return 0;
}

#define RCCL_TRY(expr)                                                                                                  \
    do {                                                                                                                \
        const ncclResult_t _r = (expr);                                                                                 \
        if (_r != ncclSuccess) {                                                                                        \
            return fail("%s failed: %s", #expr, g_rccl.GetErrorString(_r));                                             \
        }                                                                                                               \
    } while (0)

constexpr int COMM_IMAGES = 4; // full, base colour, depth-normals, variance
float4 *comm_image(rayhip_ctx *c, int k) {
    switch (k) {
    case 0:
        return c->px.full;
    case 1:
        return c->px.base_color;
    case 2:
        return c->px.depth_normals;
    default:
        return c->px.variance;
    }
}
// images a RAYHIP_REDUCE_* mask selects, in staging order
int comm_selected(uint32_t mask, int out[COMM_IMAGES]) {
    if (mask == 0) {
        mask = RAYHIP_REDUCE_ALL;
    }
    int n = 0;
    for (int k = 0; k < COMM_IMAGES; ++k) {
        if (mask & (1u << k)) {
            out[n++] = k;
        }
    }
    return n;
}
} // namespace

struct rayhip_comm {
    int nranks = 0;             // ranks of the communicator
    std::vector<int> local;     // ranks this process drives (all of them for rayhip_comm_create, one for _create_rank)
    std::vector<ncclComm_t> comms;
    std::vector<rayhip_ctx *> ctx;
};

namespace {
// pack the selected images of one context into its staging buffer (enqueued on the context stream)
int comm_pack(rayhip_ctx *c, const int *sel, int n_sel) {
    HIP_TRY(hipSetDevice(c->device));
    const size_t npix = size_t(c->w) * size_t(c->h);
    if (c->shard_stage.alloc(npix * 16 * COMM_IMAGES)) {
        return 1;
    }
    for (int k = 0; k < n_sel; ++k) {
        k_pack_owned<<<grid_for(c, npix, 256), 256, 0, c->stream>>>(comm_image(c, sel[k]), c->shard_stage.as<float4>() + size_t(k) * npix, c->w,
                                                                    c->h, c->shard);
    }
    HIP_TRY(hipGetLastError());
    return 0;
}
// root: staging buffer -> the context's own images; the radiance image also becomes RAW and is tonemapped into FINAL
int comm_unpack(rayhip_ctx *c, const int *sel, int n_sel, const rayhip_camera *cam) {
    HIP_TRY(hipSetDevice(c->device));
    const size_t npix = size_t(c->w) * size_t(c->h);
    for (int k = 0; k < n_sel; ++k) {
        k_copy_f4<<<grid_for(c, npix, 256), 256, 0, c->stream>>>(c->shard_stage.as<float4>() + size_t(k) * npix, comm_image(c, sel[k]), npix);
        if (sel[k] == 0) {
            if (cam->view_transform != 0 && cam->view_transform != c->lut_transform) {
                return fail("view transform %d needs its look-up table: rayhip_set_tonemap_lut", int(cam->view_transform));
            }
            const int rect[4] = {0, 0, c->w, c->h};
            AccumParams ap = make_accum_params(*cam, c->w, rect, 1);
            ap.lut = c->tonemap_lut.as<uint32_t>(), ap.lut_dims = c->lut_dims;
            k_retonemap<<<grid_for(c, npix, 256), 256, 0, c->stream>>>(ap, c->px, c->h);
        }
    }
    HIP_TRY(hipGetLastError());
    return 0;
}
} // namespace

extern "C" {

int rayhip_comm_create(int ndev, const int *devices, rayhip_comm **out) {
    if (ndev < 1 || !devices || !out) {
        return fail("rayhip_comm_create: bad arguments");
    }
    if (load_rccl()) {
        return 1;
    }
    rayhip_comm *m = new rayhip_comm();
    m->nranks = ndev;
    m->comms.resize(size_t(ndev));
    m->ctx.assign(size_t(ndev), nullptr);
    for (int r = 0; r < ndev; ++r) {
        m->local.push_back(r);
    }
    const ncclResult_t r = g_rccl.CommInitAll(m->comms.data(), ndev, devices);
    if (r != ncclSuccess) {
        delete m;
        return fail("ncclCommInitAll failed: %s", g_rccl.GetErrorString(r));
    }
    *out = m;
    return 0;
}

int rayhip_comm_unique_id(void *out_id, size_t size) {
    if (!out_id || size < sizeof(ncclUniqueId)) {
        return fail("rayhip_comm_unique_id needs a buffer of %zu bytes", sizeof(ncclUniqueId));
    }
    if (load_rccl()) {
        return 1;
    }
    ncclUniqueId id;
    RCCL_TRY(g_rccl.GetUniqueId(&id));
    memcpy(out_id, &id, sizeof(id));
    return 0;
}

int rayhip_comm_create_rank(const void *unique_id, int nranks, int rank, rayhip_ctx *ctx, rayhip_comm **out) {
    if (!unique_id || nranks < 1 || rank < 0 || rank >= nranks || !ctx || !out) {
        return fail("rayhip_comm_create_rank: bad arguments");
    }
    if (load_rccl()) {
        return 1;
    }
    HIP_TRY(hipSetDevice(ctx->device));
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof(id));
    ncclComm_t comm = nullptr;
    RCCL_TRY(g_rccl.CommInitRank(&comm, nranks, id, rank));
    rayhip_comm *m = new rayhip_comm();
    m->nranks = nranks;
    m->local.push_back(rank);
    m->comms.push_back(comm);
    m->ctx.push_back(ctx);
    ctx->shard = Shard{ctx->shard.tile, nranks, rank};
    *out = m;
    return 0;
}

int rayhip_comm_bind(rayhip_comm *m, int rank, rayhip_ctx *ctx) {
    if (!m || !ctx) {
        return fail("rayhip_comm_bind: bad arguments");
    }
    for (size_t i = 0; i < m->local.size(); ++i) {
        if (m->local[i] == rank) {
            m->ctx[i] = ctx;
            ctx->shard = Shard{ctx->shard.tile, m->nranks, rank}; // rank r renders the tiles r, r + N, r + 2N, ...
            return 0;
        }
    }
    return fail("rank %d is not driven by this communicator handle", rank);
}

int rayhip_comm_reduce_framebuffers(rayhip_comm *m, int root, uint32_t what, const rayhip_camera *cam) {
    if (!m || root < 0 || root >= m->nranks || !cam) {
        return fail("rayhip_comm_reduce_framebuffers: bad arguments");
    }
    int sel[COMM_IMAGES];
    const int n_sel = comm_selected(what, sel);
    if (n_sel == 0) {
        return fail("rayhip_comm_reduce_framebuffers: empty image mask");
    }
    int w = 0, h = 0;
    for (rayhip_ctx *c : m->ctx) {
        if (!c || !c->w) {
            return fail("every local rank needs a bound, resized context (rayhip_comm_bind)");
        }
        if (w && (c->w != w || c->h != h)) {
            return fail("the contexts of a communicator must render the same frame size");
        }
        w = c->w, h = c->h;
    }
    for (rayhip_ctx *c : m->ctx) {
        if (comm_pack(c, sel, n_sel)) {
            return 1;
        }
    }
    const size_t count = size_t(w) * size_t(h) * 4u * size_t(n_sel); // floats
    RCCL_TRY(g_rccl.GroupStart());
    for (size_t i = 0; i < m->ctx.size(); ++i) {
        rayhip_ctx *c = m->ctx[i];
        const ncclResult_t r = g_rccl.Reduce(c->shard_stage.p, c->shard_stage.p, count, ncclFloat, ncclSum, root, m->comms[i], c->stream);
        if (r != ncclSuccess) {
            (void)g_rccl.GroupEnd();
            return fail("ncclReduce failed: %s", g_rccl.GetErrorString(r));
        }
    }
    RCCL_TRY(g_rccl.GroupEnd());
    for (size_t i = 0; i < m->ctx.size(); ++i) {
        if (m->local[i] == root && comm_unpack(m->ctx[i], sel, n_sel, cam)) {
            return 1;
        }
    }
    for (rayhip_ctx *c : m->ctx) {
        HIP_TRY(hipSetDevice(c->device));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    return 0;
}

void rayhip_comm_destroy(rayhip_comm *m) {
    if (!m) {
        return;
    }
    for (size_t i = 0; i < m->comms.size(); ++i) {
        if (m->comms[i] && g_rccl.CommDestroy) {
            (void)g_rccl.CommDestroy(m->comms[i]);
        }
    }
    delete m;
}

// The pack step alone, into DEVICE memory the caller owns: for hosts that bring their own collective (bench.py reduces a
// torch tensor over torch.distributed's RCCL communicator).
int rayhip_export_shard_device(rayhip_ctx *c, int which, void *dst_device_rgba) {
    if (use_device(c)) {
        return 1;
    }
    if (!c->w) {
        return fail("rayhip_export_shard_device before rayhip_resize");
    }
    const float4 *src = which == RAYHIP_BUF_RAW ? c->px.full
                        : which == RAYHIP_BUF_BASE_COLOR ? c->px.base_color
                        : which == RAYHIP_BUF_DEPTH_NORMALS ? c->px.depth_normals
                        : which == RAYHIP_BUF_VARIANCE ? c->px.variance : nullptr;
    if (!src) {
        return fail("bad buffer id %d", which);
    }
    const size_t npix = size_t(c->w) * size_t(c->h);
    k_pack_owned<<<grid_for(c, npix, 256), 256, 0, c->stream>>>(src, static_cast<float4 *>(dst_device_rgba), c->w, c->h, c->shard);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}

} // extern "C"
