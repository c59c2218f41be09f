// Comm group of the C ABI (include/vq_amd.h): RCCL over xGMI for a host that is not torch.
//
// The reference has no collective to translate: its clip-level data parallelism ends in multiprocessing.Pool pickling
// the per-clip features back to the parent (calcSig_wOF.py:204-210).  Here every rank owns one GPU; the per-GPU feature
// blocks (half A -> half B hand-off) and the per-rank score slices of a sharded scan (N x 8 bytes, never the features)
// are exchanged by ONE fixed-size all-gather each.  librccl is resolved at run time (dlopen) so that libvqamd.so keeps
// a single link dependency (libamdhip64) and, inside a torch process, uses the very RCCL torch has already mapped.
#include <dlfcn.h>

#include <cstdlib>
#include <cstring>

#include "vq_common.h"

using namespace vq;

namespace {

// The slice of the RCCL API the path needs (signatures from <rccl/rccl.h>, ROCm 7.x; ncclUniqueId is 128 opaque bytes
// passed BY VALUE).
struct UniqueId {
    char bytes[VQ_COMM_ID_BYTES];
};
typedef int (*get_unique_id_fn)(UniqueId*);
typedef int (*comm_init_rank_fn)(void** comm, int nranks, UniqueId id, int rank);
typedef int (*comm_destroy_fn)(void* comm);
typedef int (*all_gather_fn)(const void* send, void* recv, size_t count, int dtype, void* comm, hipStream_t stream);
typedef int (*broadcast_fn)(const void* send, void* recv, size_t count, int dtype, int root, void* comm, hipStream_t stream);
typedef const char* (*error_string_fn)(int);
constexpr int kNcclInt8 = 0;        // ncclInt8 / ncclChar: the collectives below move opaque bytes

struct Rccl {
    void* so = nullptr;
    get_unique_id_fn get_unique_id = nullptr;
    comm_init_rank_fn comm_init_rank = nullptr;
    comm_destroy_fn comm_destroy = nullptr;
    all_gather_fn all_gather = nullptr;
    broadcast_fn broadcast = nullptr;
    error_string_fn error_string = nullptr;
    std::string why;
};

Rccl* rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* env = getenv("VQ_RCCL_LIB");
        if (env && *env) {
            r.so = dlopen(env, RTLD_NOW | RTLD_GLOBAL);          // an explicit library is the ONLY candidate
        } else {
            // an already-mapped RCCL first (torch's), then the usual names
            for (const char* n : {"librccl.so.1", "librccl.so"})
                if (!r.so) r.so = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
            for (const char* n : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"})
                if (!r.so) r.so = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        }
        if (!r.so) {
            const char* e = dlerror();           // ONE call: dlerror() clears the message it returns
            r.why = std::string("librccl not found: ") + (e ? e : "?");
            return;
        }
        r.get_unique_id = (get_unique_id_fn)dlsym(r.so, "ncclGetUniqueId");
        r.comm_init_rank = (comm_init_rank_fn)dlsym(r.so, "ncclCommInitRank");
        r.comm_destroy = (comm_destroy_fn)dlsym(r.so, "ncclCommDestroy");
        r.all_gather = (all_gather_fn)dlsym(r.so, "ncclAllGather");
        r.broadcast = (broadcast_fn)dlsym(r.so, "ncclBroadcast");
        r.error_string = (error_string_fn)dlsym(r.so, "ncclGetErrorString");
        if (!r.get_unique_id || !r.comm_init_rank || !r.comm_destroy || !r.all_gather || !r.broadcast) r.why = "librccl lacks a required symbol";
    });
    return &r;
}

int rccl_fail(Rccl* r, const char* what, int rc) {
    return fail(VQ_E_HIP, "%s failed: %s (%d)", what, r->error_string ? r->error_string(rc) : "rccl error", rc);
}

}  // namespace

struct vq_comm {
    void* comm = nullptr;
    int rank = 0, world = 1, device = 0;
    std::mutex mu;
};

#define VQ_RCCL_READY(r)                                                 \
    Rccl* r = rccl();                                                    \
    if (!r->why.empty()) return fail(VQ_E_UNSUPPORTED, "%s", r->why.c_str())

extern "C" {

int vq_comm_unique_id(void* id_out) {
    VQ_REQUIRE(id_out, "id_out is NULL");
    VQ_RCCL_READY(r);
    UniqueId id;
    const int rc = r->get_unique_id(&id);
    if (rc != 0) return rccl_fail(r, "ncclGetUniqueId", rc);
    memcpy(id_out, id.bytes, VQ_COMM_ID_BYTES);
    return VQ_OK;
}

int vq_comm_init(int32_t rank, int32_t world, const void* rccl_unique_id, int32_t device, vq_comm** out) {
    VQ_REQUIRE(out && rccl_unique_id, "NULL argument");
    VQ_REQUIRE(world >= 1 && rank >= 0 && rank < world, "rank %d outside a world of %d", rank, world);
    VQ_RCCL_READY(r);
    int ndev = 0;
    VQ_HIP(hipGetDeviceCount(&ndev));
    VQ_REQUIRE(device >= 0 && device < ndev, "device %d outside [0,%d)", device, ndev);
    VQ_HIP(hipSetDevice(device));            // stays current: the communicator is bound to the calling thread's device
    UniqueId id;
    memcpy(id.bytes, rccl_unique_id, VQ_COMM_ID_BYTES);
    vq_comm* c = new (std::nothrow) vq_comm;
    if (!c) return fail(VQ_E_NOMEM, "out of host memory");
    const int rc = r->comm_init_rank(&c->comm, world, id, rank);
    if (rc != 0) {
        delete c;
        return rccl_fail(r, "ncclCommInitRank", rc);
    }
    c->rank = rank;
    c->world = world;
    c->device = device;
    *out = c;
    return VQ_OK;
}

int vq_comm_destroy(vq_comm* c) {
    if (!c) return VQ_OK;
    Rccl* r = rccl();
    if (c->comm && r->comm_destroy) {
        DeviceGuard g(c->device);
        r->comm_destroy(c->comm);
    }
    delete c;
    return VQ_OK;
}

int vq_comm_info(vq_comm* c, int32_t* rank, int32_t* world, int32_t* device) {
    VQ_REQUIRE(c, "comm is NULL");
    if (rank) *rank = c->rank;
    if (world) *world = c->world;
    if (device) *device = c->device;
    return VQ_OK;
}

int vq_allgather_features(vq_comm* c, const void* block_dev, int64_t block_bytes, void* all_dev, void* hip_stream) {
    VQ_REQUIRE(c && block_dev && all_dev, "NULL argument");
    VQ_REQUIRE(block_bytes > 0, "block_bytes must be positive");
    VQ_RCCL_READY(r);
    std::lock_guard<std::mutex> lk(c->mu);
    DeviceGuard g(c->device);
    const int rc = r->all_gather(block_dev, all_dev, (size_t)block_bytes, kNcclInt8, c->comm, (hipStream_t)hip_stream);
    if (rc != 0) return rccl_fail(r, "ncclAllGather", rc);
    return VQ_OK;
}

int vq_allgather_scores(vq_comm* c, vq_db* db, int64_t slice_rows, double* all_scores_dev, void* hip_stream) {
    VQ_REQUIRE(c && db && all_scores_dev, "NULL argument");
    int64_t n = 0;
    int32_t S, E, D, dt;
    int rc = vq_db_shape(db, &n, &S, &E, &D, &dt);
    if (rc != VQ_OK) return rc;
    VQ_REQUIRE(slice_rows >= n, "slice_rows %lld smaller than this rank's %lld rows", (long long)slice_rows, (long long)n);
    VQ_RCCL_READY(r);
    std::lock_guard<std::mutex> lk(c->mu);
    DeviceGuard g(c->device);
    hipStream_t st = (hipStream_t)hip_stream;
    // every rank contributes slice_rows doubles: its own slice first lands in its slot of the result (the padding
    // rows beyond n are zero), then the all-gather runs IN PLACE on that slot (ncclAllGather allows
    // sendbuff == recvbuff + rank * sendcount).  The copy is ordered behind the scan that produced the scores (the
    // scan runs on the database handle's stream, which need not be hip_stream); VQ_E_STATE if there was no scan.
    double* mine = all_scores_dev + (int64_t)c->rank * slice_rows;
    VQ_HIP(hipMemsetAsync(mine, 0, (size_t)slice_rows * 8, st));
    rc = db_copy_scores_ordered(db, mine, nullptr, st);
    if (rc != VQ_OK) return rc;
    rc = r->all_gather(mine, all_scores_dev, (size_t)slice_rows * 8, kNcclInt8, c->comm, st);
    if (rc != 0) return rccl_fail(r, "ncclAllGather", rc);
    return VQ_OK;
}

int vq_broadcast_query(vq_comm* c, void* buf_dev, int64_t bytes, int32_t root, void* hip_stream) {
    VQ_REQUIRE(c && buf_dev && bytes > 0, "bad argument");
    VQ_REQUIRE(root >= 0 && root < c->world, "root %d outside a world of %d", root, c->world);
    VQ_RCCL_READY(r);
    std::lock_guard<std::mutex> lk(c->mu);
    DeviceGuard g(c->device);
    const int rc = r->broadcast(buf_dev, buf_dev, (size_t)bytes, kNcclInt8, root, c->comm, (hipStream_t)hip_stream);
    if (rc != 0) return rccl_fail(r, "ncclBroadcast", rc);
    return VQ_OK;
}

}  // extern "C"
