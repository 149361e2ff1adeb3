// fake_rccl.cpp -- TEST INFRASTRUCTURE, not product: a stand-in for the handful of RCCL entry points librayhip resolves with dlopen
// (ray_amd/csrc/comm.hip.h: load_rccl), loaded through the existing RAYHIP_RCCL_LIB override, so that the one-process-per-GPU exchange
// (rayhip_comm_create_rank / rayhip_comm_reduce_framebuffers: region offsets, counts, the group of one ncclSend per rank and N - 1 ncclRecv
// on the root, unpack) EXECUTES with N > 1 ranks on a box that has ONE device.  Real RCCL refuses two ranks on one device; this library does
// not care where a rank's buffer lives: a message travels  device -> shared host memory (POSIX shm named by the "unique id") -> device.
//
// Semantics kept from NCCL where the caller depends on them:
//   * ncclGetUniqueId on one rank, the id handed to the others out of band, ncclCommInitRank(nranks, id, rank) on every rank;
//   * point-to-point operations between ncclGroupStart / ncclGroupEnd are only RECORDED; ncclGroupEnd issues the sends, then the receives
//     (so a root that posts its N - 1 receives in a group cannot deadlock against senders that post theirs in a group);
//   * an operation is ordered behind the work already enqueued on its stream (here: the stream is drained first) and its data is in
//     place when ncclGroupEnd returns (stronger than NCCL, which only enqueues -- the caller synchronises its stream anyway);
//   * a send and the matching receive must agree in size: a mismatch is an error (ncclInvalidArgument), as is a peer out of range.
// One mailbox per ordered pair (src, dst) and one message in flight per mailbox -- what a gather needs.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

namespace {
constexpr size_t MAILBOX_BYTES = size_t(64) << 20; // per ordered pair: a 1080p gather of four images is 133 MB / N
constexpr int MAX_RANKS = 8;

struct Mailbox {
    std::atomic<unsigned long long> full; // bytes waiting + 1, 0 = empty
};
struct Control {
    std::atomic<int> joined;
    Mailbox box[MAX_RANKS][MAX_RANKS];
};
struct Comm {
    int nranks, rank;
    char name[64];
    Control *ctl;
    unsigned char *data; // [src][dst][MAILBOX_BYTES]
    size_t bytes;
};
struct Op {
    bool send;
    void *buf;
    size_t bytes;
    int peer;
    Comm *comm;
    hipStream_t stream;
};
thread_local int g_group = 0;
thread_local std::vector<Op> g_ops;

size_t type_bytes(ncclDataType_t t) {
    switch (t) {
    case ncclInt8:
    case ncclUint8:
        return 1;
    case ncclFloat16:
        return 2;
    case ncclInt32:
    case ncclUint32:
    case ncclFloat32:
        return 4;
    case ncclInt64:
    case ncclUint64:
    case ncclFloat64:
        return 8;
    default:
        return 0;
    }
}
unsigned char *slot(Comm *c, int src, int dst) { return c->data + (size_t(src) * MAX_RANKS + size_t(dst)) * MAILBOX_BYTES; }

ncclResult_t run(const Op &op) {
    Comm *c = op.comm;
    if (op.peer < 0 || op.peer >= c->nranks || op.peer == c->rank || op.bytes > MAILBOX_BYTES) {
        return ncclInvalidArgument;
    }
    if (hipStreamSynchronize(op.stream) != hipSuccess) {
        return ncclUnhandledCudaError;
    }
    const auto deadline = std::chrono::steady_clock::now() + std::chrono::seconds(120);
    if (op.send) {
        Mailbox &b = c->ctl->box[c->rank][op.peer];
        while (b.full.load(std::memory_order_acquire) != 0) { // the previous message has not been taken yet
            if (std::chrono::steady_clock::now() > deadline) {
                return ncclSystemError;
            }
            std::this_thread::sleep_for(std::chrono::microseconds(50));
        }
        if (hipMemcpy(slot(c, c->rank, op.peer), op.buf, op.bytes, hipMemcpyDeviceToHost) != hipSuccess) {
            return ncclUnhandledCudaError;
        }
        b.full.store(op.bytes + 1, std::memory_order_release);
    } else {
        Mailbox &b = c->ctl->box[op.peer][c->rank];
        unsigned long long have;
        while ((have = b.full.load(std::memory_order_acquire)) == 0) {
            if (std::chrono::steady_clock::now() > deadline) {
                return ncclSystemError;
            }
            std::this_thread::sleep_for(std::chrono::microseconds(50));
        }
        if (have - 1 != op.bytes) {
            fprintf(stderr, "fake_rccl: rank %d receives %zu bytes from rank %d, which sent %llu\n", c->rank, op.bytes, op.peer, have - 1);
            return ncclInvalidArgument;
        }
        if (hipMemcpy(op.buf, slot(c, op.peer, c->rank), op.bytes, hipMemcpyHostToDevice) != hipSuccess) {
            return ncclUnhandledCudaError;
        }
        b.full.store(0, std::memory_order_release);
    }
    return ncclSuccess;
}
} // namespace

extern "C" {
#define FAKE_API __attribute__((visibility("default")))

FAKE_API ncclResult_t ncclGetUniqueId(ncclUniqueId *id) {
    memset(id, 0, sizeof(*id));
    snprintf(id->internal, sizeof(id->internal), "/fake_rccl_%d_%llx", int(getpid()),
             (unsigned long long)std::chrono::steady_clock::now().time_since_epoch().count());
    return ncclSuccess;
}

FAKE_API ncclResult_t ncclCommInitRank(ncclComm_t *out, int nranks, ncclUniqueId id, int rank) {
    if (nranks < 1 || nranks > MAX_RANKS || rank < 0 || rank >= nranks || id.internal[0] != '/') {
        return ncclInvalidArgument;
    }
    Comm *c = new Comm();
    c->nranks = nranks, c->rank = rank;
    snprintf(c->name, sizeof(c->name), "%s", id.internal);
    c->bytes = sizeof(Control) + size_t(MAX_RANKS) * MAX_RANKS * MAILBOX_BYTES; // (sparse: only touched mailboxes get pages)
    const int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, off_t(c->bytes)) != 0) {
        delete c;
        return ncclSystemError;
    }
    void *p = mmap(nullptr, c->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) {
        delete c;
        return ncclSystemError;
    }
    c->ctl = static_cast<Control *>(p); // (a fresh segment is zero-filled: every mailbox empty, nobody joined)
    c->data = static_cast<unsigned char *>(p) + sizeof(Control);
    c->ctl->joined.fetch_add(1);
    const auto deadline = std::chrono::steady_clock::now() + std::chrono::seconds(120);
    while (c->ctl->joined.load() < nranks) { // ncclCommInitRank is a rendezvous
        if (std::chrono::steady_clock::now() > deadline) {
            return ncclSystemError;
        }
        std::this_thread::sleep_for(std::chrono::milliseconds(1));
    }
    *out = reinterpret_cast<ncclComm_t>(c);
    return ncclSuccess;
}

FAKE_API ncclResult_t ncclCommInitAll(ncclComm_t *, int, const int *) { return ncclInvalidUsage; } // (the in-process path of librayhip uses peer copies)

FAKE_API ncclResult_t ncclCommDestroy(ncclComm_t comm) {
    Comm *c = reinterpret_cast<Comm *>(comm);
    if (c) {
        munmap(c->ctl, c->bytes);
        shm_unlink(c->name); // (the first rank to leave removes the name; the mapping of the others lives on)
        delete c;
    }
    return ncclSuccess;
}

FAKE_API ncclResult_t ncclCommCount(const ncclComm_t comm, int *count) {
    *count = reinterpret_cast<const Comm *>(comm)->nranks;
    return ncclSuccess;
}
FAKE_API ncclResult_t ncclCommUserRank(const ncclComm_t comm, int *rank) {
    *rank = reinterpret_cast<const Comm *>(comm)->rank;
    return ncclSuccess;
}

static ncclResult_t post(bool send, void *buf, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t stream) {
    const size_t tb = type_bytes(type);
    if (!tb || !comm) {
        return ncclInvalidArgument;
    }
    const Op op{send, buf, count * tb, peer, reinterpret_cast<Comm *>(comm), stream};
    if (g_group > 0) {
        g_ops.push_back(op);
        return ncclSuccess;
    }
    return run(op);
}
FAKE_API ncclResult_t ncclSend(const void *buf, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t stream) {
    return post(true, const_cast<void *>(buf), count, type, peer, comm, stream);
}
FAKE_API ncclResult_t ncclRecv(void *buf, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t stream) {
    return post(false, buf, count, type, peer, comm, stream);
}
FAKE_API ncclResult_t ncclGroupStart() {
    ++g_group;
    return ncclSuccess;
}
FAKE_API ncclResult_t ncclGroupEnd() {
    if (g_group <= 0) {
        return ncclInvalidUsage;
    }
    if (--g_group > 0) {
        return ncclSuccess;
    }
    std::vector<Op> ops;
    ops.swap(g_ops);
    ncclResult_t r = ncclSuccess;
    for (int pass = 0; pass < 2 && r == ncclSuccess; ++pass) { // sends first: they only fill this rank's own mailboxes
        for (const Op &op : ops) {
            if (op.send == (pass == 0) && r == ncclSuccess) {
                r = run(op);
            }
        }
    }
    if (const char *log = getenv("FAKE_RCCL_LOG")) { // what the caller really asked for: the test reads it back
        if (FILE *f = fopen(log, "a")) {
            for (const Op &op : ops) {
                fprintf(f, "rank %d %s %zu bytes peer %d\n", op.comm->rank, op.send ? "send" : "recv", op.bytes, op.peer);
            }
            fclose(f);
        }
    }
    return r;
}
FAKE_API const char *ncclGetErrorString(ncclResult_t r) {
    switch (r) {
    case ncclSuccess:
        return "no error";
    case ncclInvalidArgument:
        return "invalid argument (fake_rccl)";
    case ncclInvalidUsage:
        return "invalid usage (fake_rccl)";
    case ncclSystemError:
        return "system error / timeout (fake_rccl)";
    default:
        return "error (fake_rccl)";
    }
}
}
