// comm.hip.h -- the one exchange step of the path: assemble the tile-sharded framebuffers of N GPUs on one rank over
// RCCL / xGMI (rayhip_comm_create / rayhip_comm_reduce_framebuffers / rayhip_comm_destroy; 8e).
// Included by rayhip.hip (needs rayhip_ctx).
//
// What is exchanged: the shards are disjoint sets of 64 x 64 tiles, so the "reduce" of the frame is a GATHER.  Every rank
// packs the tiles it owns (rayhip_set_shard) of the running mean (`full`), of the two aux images and of the variance
// estimate densely into a staging buffer -- 1/N of the frame -- and sends it straight to the root (ncclSend / ncclRecv in
// one group); the root scatters what arrives into its own images and re-runs the tonemap.  xGMI is point-to-point: each
// of the N - 1 senders has its own link to the root, so the step moves 33 MB * (N - 1) / N at 1080p over N - 1 links in
// parallel (4.1 MB per link at N = 8), where a ring ncclReduce of zero-padded full frames (what round 2 did) pushed the
// whole 33 MB through every link in turn.  Copies are exact, so the assembled images equal an unsharded render bit for
// bit, and DenoiseImage on the root sees the complete guides.  A rank's accumulation state is never overwritten on pixels
// it owns, so progressive refinement (render more, gather again) stays exact.
//
// Two transports, one packing:
//   * rayhip_comm_create -- ONE process drives all GPUs (a C++ host using RendererHIP with RAY_HIP_DEVICES): the root pulls
//     every sender's buffer with hipMemcpyPeerAsync on its own stream, ordered behind the sender's pack kernel by an event:
//     plain peer DMA over the sender's xGMI link, no collective library involved (and two contexts may share a device,
//     which is how the path is tested on a one-GPU box);
//   * rayhip_comm_unique_id + rayhip_comm_create_rank -- one process per GPU (bench.py under torch.distributed.run): ncclSend /
//     ncclRecv in one group.  RCCL is loaded with dlopen when the first such communicator is created: librayhip.so has no
//     link-time dependency on it, and inside a process that already carries an RCCL (PyTorch) the loaded copy is reused.
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
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
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
    RCCL_SYM(CommCount, "ncclCommCount")
    RCCL_SYM(CommUserRank, "ncclCommUserRank")
    RCCL_SYM(Send, "ncclSend")
    RCCL_SYM(Recv, "ncclRecv")
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
    bool in_process = false;    // rayhip_comm_create: every rank is a context of this process, transport = peer copies
    std::vector<int> local;     // ranks this process drives (all of them for rayhip_comm_create, one for _create_rank)
    std::vector<ncclComm_t> comms; // (one-process-per-GPU form only)
    std::vector<rayhip_ctx *> ctx;
    std::vector<hipEvent_t> packed; // in-process form: "this rank's tiles are packed", recorded on its stream
    std::vector<int> devices;
};

namespace {
// floats rank `rank` contributes for `n_sel` images of a w x h frame
size_t comm_rank_floats(int w, int h, int tile, int nranks, int rank, int n_sel) {
    const ShardTiles st = shard_tiles(w, h, tile);
    return size_t(shard_owned_tiles(st.total, nranks, rank)) * size_t(tile) * size_t(tile) * 4u * size_t(n_sel);
}
// pack the owned tiles of the selected images of one context, image after image, at `dst` (enqueued on the context stream)
int comm_pack(rayhip_ctx *c, const int *sel, int n_sel, float4 *dst) {
    HIP_TRY(hipSetDevice(c->device));
    const ShardTiles st = shard_tiles(c->w, c->h, c->shard.tile);
    const int owned = shard_owned_tiles(st.total, c->shard.count, c->shard.index);
    const size_t per_image = size_t(owned) * size_t(c->shard.tile) * size_t(c->shard.tile);
    for (int k = 0; k < n_sel && per_image; ++k) {
        k_pack_owned_dense<<<grid_for(c, per_image, 256), 256, 0, c->stream>>>(comm_image(c, sel[k]), dst + size_t(k) * per_image, c->w, c->h,
                                                                             c->shard, owned);
    }
    HIP_TRY(hipGetLastError());
    return 0;
}
// root: the packed tiles of rank `from` -> the context's own images
int comm_unpack(rayhip_ctx *c, const int *sel, int n_sel, int from, const float4 *src) {
    HIP_TRY(hipSetDevice(c->device));
    const ShardTiles st = shard_tiles(c->w, c->h, c->shard.tile);
    const int owned = shard_owned_tiles(st.total, c->shard.count, from);
    const size_t per_image = size_t(owned) * size_t(c->shard.tile) * size_t(c->shard.tile);
    const Shard sender = {c->shard.tile, c->shard.count, from};
    for (int k = 0; k < n_sel && per_image; ++k) {
        k_unpack_owned_dense<<<grid_for(c, per_image, 256), 256, 0, c->stream>>>(src + size_t(k) * per_image, comm_image(c, sel[k]), c->w, c->h,
                                                                               sender, owned);
    }
    HIP_TRY(hipGetLastError());
    return 0;
}
// root, after the radiance image is complete: it becomes RAW and is tonemapped into FINAL
int comm_finish(rayhip_ctx *c, const rayhip_camera *cam) {
    HIP_TRY(hipSetDevice(c->device));
    if (cam->view_transform != 0 && cam->view_transform != c->lut_transform) {
        return fail("view transform %d needs its look-up table: rayhip_set_tonemap_lut", int(cam->view_transform));
    }
    const int rect[4] = {0, 0, c->w, c->h};
    AccumParams ap = make_accum_params(*cam, c->w, rect, 1);
    ap.lut = c->tonemap_lut.as<uint32_t>(), ap.lut_dims = c->lut_dims;
    k_retonemap<<<grid_for(c, size_t(c->w) * c->h, 256), 256, 0, c->stream>>>(ap, c->px, c->h);
    HIP_TRY(hipGetLastError());
    return 0;
}
} // namespace

extern "C" {

int rayhip_comm_create(int ndev, const int *devices, rayhip_comm **out) {
    if (ndev < 1 || !devices || !out) {
        return fail("rayhip_comm_create: bad arguments");
    }
    int have = 0;
    if (hipGetDeviceCount(&have) != hipSuccess || have == 0) {
        return fail("no HIP device available (librayhip has no CPU path)");
    }
    rayhip_comm *m = new rayhip_comm();
    m->nranks = ndev;
    m->in_process = true;
    m->ctx.assign(size_t(ndev), nullptr);
    m->packed.assign(size_t(ndev), nullptr);
    for (int r = 0; r < ndev; ++r) {
        if (devices[r] < 0 || devices[r] >= have) {
            delete m;
            return fail("rayhip_comm_create: device %d out of range (have %d)", devices[r], have);
        }
        m->local.push_back(r);
        m->devices.push_back(devices[r]);
    }
    // direct peer access where the devices allow it (the copy works either way; without it the runtime stages through the host)
    for (int a = 0; a < ndev; ++a) {
        for (int b = 0; b < ndev; ++b) {
            int can = 0;
            if (devices[a] != devices[b] && hipDeviceCanAccessPeer(&can, devices[a], devices[b]) == hipSuccess && can) {
                if (hipSetDevice(devices[a]) == hipSuccess) {
                    (void)hipDeviceEnablePeerAccess(devices[b], 0); // (already enabled: an error that does not matter)
                }
            }
        }
    }
    (void)hipGetLastError();
    *out = m;
    return 0;
}

// Can this process form the one-process-per-GPU communicator at all?  LOCAL and non-collective (dlopen + symbol resolution): the ranks of a job
// ask this first and agree on the answer before any of them enters ncclCommInitRank, which only returns when every rank has entered it
// (a rank that failed before the collective left the others waiting in it).
int rayhip_comm_probe(void) { return load_rccl(); }

// What the communicator is, as the transport itself reports it: out[0] = ranks, out[1] = first local rank, out[2] = ncclCommCount of the first
// local RCCL communicator (-1 for the in-process form, which has none), out[3] = 1 if the transport is peer copies inside one process
int rayhip_comm_info(rayhip_comm *m, int out[4]) {
    if (!m || !out) {
        return fail("rayhip_comm_info: bad arguments");
    }
    out[0] = m->nranks, out[1] = m->local.empty() ? -1 : m->local[0], out[2] = -1, out[3] = m->in_process ? 1 : 0;
    if (!m->in_process && !m->comms.empty() && m->comms[0]) {
        RCCL_TRY(g_rccl.CommCount(m->comms[0], &out[2]));
    }
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
    // what RCCL itself thinks this communicator is -- the one place a wrong rendezvous (a stale id, two jobs sharing one) shows before
    // the first gather hangs; logged once per rank so that a multi-GPU run leaves a trace of the transport it really used
    int seen_ranks = -1, seen_rank = -1;
    ncclResult_t asked = g_rccl.CommCount(comm, &seen_ranks);
    if (asked == ncclSuccess) {
        asked = g_rccl.CommUserRank(comm, &seen_rank);
    }
    if (asked != ncclSuccess || seen_ranks != nranks || seen_rank != rank) {
        g_rccl.CommDestroy(comm);
        return fail("RCCL communicator reports rank %d of %d, expected rank %d of %d", seen_rank, seen_ranks, rank, nranks);
    }
    if (getenv("RAYHIP_TRACE_COMM")) { // (like every other diagnostic of the library: on request only -- an N-rank job would print N lines per communicator)
        fprintf(stderr, "rayhip: RCCL communicator up: rank %d of %d on device %d\n", seen_rank, seen_ranks, ctx->device);
    }
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
            if (m->in_process && ctx->device != m->devices[i]) {
                return fail("rank %d was created for device %d, the context lives on device %d", rank, m->devices[i], ctx->device);
            }
            m->ctx[i] = ctx;
            ctx->shard = Shard{ctx->shard.tile, m->nranks, rank}; // rank r renders the tiles r, r + N, r + 2N, ...
            if (m->in_process && !m->packed[i]) {
                HIP_TRY(hipSetDevice(ctx->device));
                HIP_TRY(hipEventCreateWithFlags(&m->packed[i], hipEventDisableTiming));
            }
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
    int w = 0, h = 0, tile = 0;
    for (size_t i = 0; i < m->ctx.size(); ++i) {
        rayhip_ctx *c = m->ctx[i];
        if (!c || !c->w) {
            return fail("every local rank needs a bound, resized context (rayhip_comm_bind)");
        }
        if (w && (c->w != w || c->h != h || c->shard.tile != tile)) {
            return fail("the contexts of a communicator must render the same frame size with the same shard tile");
        }
        if (c->shard.count != m->nranks || c->shard.index != m->local[i]) {
            return fail("rank %d: the context's shard is (%d of %d); rayhip_set_shard was called after rayhip_comm_bind", m->local[i],
                        c->shard.index, c->shard.count);
        }
        w = c->w, h = c->h, tile = c->shard.tile;
    }
    // staging: the root holds one region per sender (rank order), every other rank its own tiles
    std::vector<size_t> region(size_t(m->nranks) + 1, 0); // float offsets on the root
    for (int r = 0; r < m->nranks; ++r) {
        region[size_t(r) + 1] = region[size_t(r)] + (r == root ? 0 : comm_rank_floats(w, h, tile, m->nranks, r, n_sel));
    }
    for (size_t i = 0; i < m->ctx.size(); ++i) {
        rayhip_ctx *c = m->ctx[i];
        HIP_TRY(hipSetDevice(c->device));
        g_touch_stream = c->stream;
        const size_t floats = m->local[i] == root ? region[size_t(m->nranks)] : comm_rank_floats(w, h, tile, m->nranks, m->local[i], n_sel);
        if (c->shard_stage.alloc(std::max<size_t>(floats, 4) * sizeof(float))) {
            return 1;
        }
        if (m->local[i] != root && comm_pack(c, sel, n_sel, c->shard_stage.as<float4>())) {
            return 1;
        }
    }
    if (m->in_process) {
        // the root pulls every sender's buffer: a peer copy on ITS stream, behind the event the sender records after packing
        rayhip_ctx *rc = nullptr;
        for (size_t i = 0; i < m->ctx.size(); ++i) {
            rc = m->local[i] == root ? m->ctx[i] : rc;
        }
        for (size_t i = 0; i < m->ctx.size(); ++i) {
            rayhip_ctx *c = m->ctx[i];
            const size_t n = comm_rank_floats(w, h, tile, m->nranks, m->local[i], n_sel);
            if (m->local[i] == root || n == 0) {
                continue;
            }
            HIP_TRY(hipSetDevice(c->device));
            HIP_TRY(hipEventRecord(m->packed[i], c->stream));
            HIP_TRY(hipSetDevice(rc->device));
            HIP_TRY(hipStreamWaitEvent(rc->stream, m->packed[i], 0));
            HIP_TRY(hipMemcpyPeerAsync(rc->shard_stage.as<float>() + region[size_t(m->local[i])], rc->device, c->shard_stage.p, c->device,
                                       n * sizeof(float), rc->stream));
        }
    } else if (m->nranks > 1) {
        RCCL_TRY(g_rccl.GroupStart());
        ncclResult_t r = ncclSuccess;
        for (size_t i = 0; i < m->ctx.size() && r == ncclSuccess; ++i) {
            rayhip_ctx *c = m->ctx[i];
            if (m->local[i] == root) {
                for (int from = 0; from < m->nranks && r == ncclSuccess; ++from) {
                    const size_t n = region[size_t(from) + 1] - region[size_t(from)];
                    if (from != root && n) {
                        r = g_rccl.Recv(c->shard_stage.as<float>() + region[size_t(from)], n, ncclFloat, from, m->comms[i], c->stream);
                    }
                }
            } else {
                const size_t n = comm_rank_floats(w, h, tile, m->nranks, m->local[i], n_sel);
                if (n) {
                    r = g_rccl.Send(c->shard_stage.p, n, ncclFloat, root, m->comms[i], c->stream);
                }
            }
        }
        if (r != ncclSuccess) {
            (void)g_rccl.GroupEnd();
            return fail("ncclSend / ncclRecv failed: %s", g_rccl.GetErrorString(r));
        }
        RCCL_TRY(g_rccl.GroupEnd());
    }
    for (size_t i = 0; i < m->ctx.size(); ++i) {
        if (m->local[i] != root) {
            continue;
        }
        rayhip_ctx *c = m->ctx[i];
        for (int from = 0; from < m->nranks; ++from) {
            if (from != root && comm_unpack(c, sel, n_sel, from, reinterpret_cast<const float4 *>(c->shard_stage.as<float>() + region[size_t(from)]))) {
                return 1;
            }
        }
        bool radiance = false;
        for (int k = 0; k < n_sel; ++k) {
            radiance |= sel[k] == 0;
        }
        if (radiance && comm_finish(c, cam)) {
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
    for (hipEvent_t e : m->packed) {
        if (e) {
            (void)hipEventDestroy(e);
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

// Bring-your-own transport (torch.distributed in ray_amd/multigpu.py, MPI in a C++ host): the same dense tile packing as
// the RCCL path, into / out of DEVICE memory the caller owns.
size_t rayhip_owned_bytes(rayhip_ctx *c, uint32_t what, int nranks, int rank) {
    int sel[COMM_IMAGES];
    const int n_sel = comm_selected(what, sel);
    if (!c || !c->w || nranks < 1 || rank < 0 || rank >= nranks) {
        return 0;
    }
    return comm_rank_floats(c->w, c->h, c->shard.tile, nranks, rank, n_sel) * sizeof(float);
}

int rayhip_export_owned(rayhip_ctx *c, uint32_t what, void *dst_device, size_t capacity_bytes) {
    if (use_device(c)) {
        return 1;
    }
    if (!c->w) {
        return fail("rayhip_export_owned before rayhip_resize");
    }
    int sel[COMM_IMAGES];
    const int n_sel = comm_selected(what, sel);
    if (rayhip_owned_bytes(c, what, c->shard.count, c->shard.index) > capacity_bytes) {
        return fail("rayhip_export_owned: the destination holds %zu bytes, %zu needed", capacity_bytes,
                    rayhip_owned_bytes(c, what, c->shard.count, c->shard.index));
    }
    if (comm_pack(c, sel, n_sel, static_cast<float4 *>(dst_device))) {
        return 1;
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}

int rayhip_import_owned(rayhip_ctx *c, uint32_t what, int from_rank, const void *src_device, size_t bytes) {
    if (use_device(c)) {
        return 1;
    }
    if (!c->w || from_rank < 0 || from_rank >= c->shard.count) {
        return fail("rayhip_import_owned: rank %d of a shard of %d", from_rank, c->shard.count);
    }
    int sel[COMM_IMAGES];
    const int n_sel = comm_selected(what, sel);
    if (bytes < rayhip_owned_bytes(c, what, c->shard.count, from_rank)) {
        return fail("rayhip_import_owned: %zu bytes, rank %d sends %zu", bytes, from_rank, rayhip_owned_bytes(c, what, c->shard.count, from_rank));
    }
    if (from_rank != c->shard.index && comm_unpack(c, sel, n_sel, from_rank, static_cast<const float4 *>(src_device))) {
        return 1;
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}

int rayhip_finish_import(rayhip_ctx *c, const rayhip_camera *cam) {
    if (use_device(c) || !cam) {
        return 1;
    }
    if (comm_finish(c, cam)) {
        return 1;
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}

} // extern "C"
