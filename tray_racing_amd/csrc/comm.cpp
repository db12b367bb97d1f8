// comm.cpp — the N-GPU hit-shard gather behind the C ABI (include/trx.h "multi-GPU"): one process per GPU, an
// in-place all-gather of the compact per-rank hit shards over RCCL (xGMI), and the de-interleave of the gathered
// [world][m][records] buffer into row-major frames.  RCCL is loaded on first use (dlopen), so a single-GPU host never
// pays for it and libtrx.so has no link-time dependency on it.
//
// What it replaces in the reference: nothing — tray_racing is single-GPU; this is the "RCCL only for the final
// hit-buffer gather" of the north_star, shaped so a Rust/C host can drive it.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>

#include "../../include/trx.h"

namespace trx {
int fail_msg(int code, const char *fmt, ...); // api.cpp: sets the thread-local error string
}

namespace {

// the few RCCL entry points used, resolved at run time (signatures of rccl.h; ncclUniqueId is 128 opaque bytes)
struct NcclId {
    char internal[128];
};
typedef void *NcclComm;
typedef int (*GetUniqueIdFn)(NcclId *);
typedef int (*CommInitRankFn)(NcclComm *, int, NcclId, int);
typedef int (*AllGatherFn)(const void *, void *, size_t, int /*ncclDataType_t*/, NcclComm, hipStream_t);
typedef int (*CommDestroyFn)(NcclComm);
typedef const char *(*GetErrorStringFn)(int);
typedef int (*CommCountFn)(NcclComm, int *);
typedef int (*SendFn)(const void *, size_t, int, int, NcclComm, hipStream_t);
typedef int (*RecvFn)(void *, size_t, int, int, NcclComm, hipStream_t);
typedef int (*GroupFn)(void);

struct Rccl {
    void *handle = nullptr;
    GetUniqueIdFn get_unique_id = nullptr;
    CommInitRankFn comm_init_rank = nullptr;
    AllGatherFn all_gather = nullptr;
    CommDestroyFn comm_destroy = nullptr;
    GetErrorStringFn get_error_string = nullptr;
    CommCountFn comm_count = nullptr;
    SendFn send = nullptr;
    RecvFn recv = nullptr;
    GroupFn group_start = nullptr, group_end = nullptr;
    std::string error;
};

Rccl &rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        // TRX_RCCL_LIBRARY (documented in trx.h): the library to load instead of the system's librccl.so
        const char *named = getenv("TRX_RCCL_LIBRARY");
        if (named && *named) {
            r.handle = dlopen(named, RTLD_NOW | RTLD_GLOBAL);
        } else {
            for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
                r.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
                if (r.handle) break;
            }
        }
        if (!r.handle) {
            const char *why = dlerror(); // one call: dlerror() clears the message it returns
            r.error = std::string("librccl.so not found: ") + (why ? why : "no loader message");
            return;
        }
        r.get_unique_id = (GetUniqueIdFn)dlsym(r.handle, "ncclGetUniqueId");
        r.comm_init_rank = (CommInitRankFn)dlsym(r.handle, "ncclCommInitRank");
        r.all_gather = (AllGatherFn)dlsym(r.handle, "ncclAllGather");
        r.comm_destroy = (CommDestroyFn)dlsym(r.handle, "ncclCommDestroy");
        r.get_error_string = (GetErrorStringFn)dlsym(r.handle, "ncclGetErrorString");
        r.comm_count = (CommCountFn)dlsym(r.handle, "ncclCommCount");
        r.send = (SendFn)dlsym(r.handle, "ncclSend");
        r.recv = (RecvFn)dlsym(r.handle, "ncclRecv");
        r.group_start = (GroupFn)dlsym(r.handle, "ncclGroupStart");
        r.group_end = (GroupFn)dlsym(r.handle, "ncclGroupEnd");
        if (!r.get_unique_id || !r.comm_init_rank || !r.all_gather || !r.comm_destroy) r.error = "librccl.so lacks an expected symbol";
    });
    return r;
}

constexpr int kNcclInt64 = 4; // ncclInt64 (rccl.h ncclDataType_t): one 8-byte hit record per element

// Record k of rank r's block is pixel ((lt*world + r) % tiles_x)*8 + (k&7), ((lt*world + r) / tiles_x)*8 + (k>>3) with
// lt = k / 64 (kernels.hip, TRX_LAYOUT_SHARD); one thread per pixel reads its record.
__global__ void k_assemble(const trx_hit *__restrict__ flat, trx_hit *__restrict__ frames, uint32_t width, uint32_t height,
                           uint32_t tiles_x, uint32_t world, uint32_t m, uint64_t records) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t n = (uint64_t)width * height;
    if (i >= n * m) return;
    const uint32_t f = (uint32_t)(i / n);
    const uint64_t p = i - (uint64_t)f * n;
    const uint32_t x = (uint32_t)(p % width), y = (uint32_t)(p / width);
    const uint32_t tile = (y >> 3) * tiles_x + (x >> 3);
    const uint32_t r = tile % world, lt = tile / world;
    const uint64_t rec = (uint64_t)lt * 64u + ((y & 7u) << 3) + (x & 7u);
    frames[i] = flat[((uint64_t)r * m + f) * records + rec];
}

} // namespace

struct trx_comm {
    NcclComm comm = nullptr;
    int rank = 0, world = 1, device = 0;
};

extern "C" {

int trx_comm_unique_id(void *out_id128) {
    if (!out_id128) return trx::fail_msg(TRX_ERR_INVALID, "null id buffer");
    Rccl &r = rccl();
    if (!r.error.empty()) return trx::fail_msg(TRX_ERR_NO_DEVICE, "%s", r.error.c_str());
    NcclId id;
    const int rc = r.get_unique_id(&id);
    if (rc != 0) return trx::fail_msg(TRX_ERR_NO_DEVICE, "ncclGetUniqueId: %s", r.get_error_string ? r.get_error_string(rc) : "error");
    std::memcpy(out_id128, &id, sizeof(id));
    return TRX_OK;
}

int trx_comm_create(const void *id128, int rank, int world, int device, trx_comm **out) {
    if (!id128 || !out) return trx::fail_msg(TRX_ERR_INVALID, "null argument");
    if (world < 1 || rank < 0 || rank >= world) return trx::fail_msg(TRX_ERR_INVALID, "rank %d of %d", rank, world);
    Rccl &r = rccl();
    if (!r.error.empty()) return trx::fail_msg(TRX_ERR_NO_DEVICE, "%s", r.error.c_str());
    int prev_device = -1;
    (void)hipGetDevice(&prev_device);
    if (hipSetDevice(device) != hipSuccess) return trx::fail_msg(TRX_ERR_NO_DEVICE, "no HIP device %d", device);
    NcclId id;
    std::memcpy(&id, id128, sizeof(id));
    NcclComm comm = nullptr;
    const int rc = r.comm_init_rank(&comm, world, id, rank); // binds the communicator to the current device
    if (prev_device >= 0 && prev_device != device) (void)hipSetDevice(prev_device);
    if (rc != 0) return trx::fail_msg(TRX_ERR_NO_DEVICE, "ncclCommInitRank: %s", r.get_error_string ? r.get_error_string(rc) : "error");
    trx_comm *c = new trx_comm;
    c->comm = comm;
    c->rank = rank;
    c->world = world;
    c->device = device;
    *out = c;
    return TRX_OK;
}

void trx_comm_destroy(trx_comm *comm) {
    if (!comm) return;
    if (comm->comm) (void)rccl().comm_destroy(comm->comm);
    delete comm;
}

int trx_comm_world_size(const trx_comm *comm) {
    if (!comm) return 0;
    int n = comm->world;
    Rccl &r = rccl();
    if (r.comm_count && comm->comm) (void)r.comm_count(comm->comm, &n); // what the communicator itself says
    return n;
}

int trx_gather_shards(trx_comm *comm, trx_hit *d_flat, uint64_t records_per_rank, void *stream) {
    if (!comm || !d_flat) return trx::fail_msg(TRX_ERR_INVALID, "null argument");
    if (records_per_rank == 0) return TRX_OK;
    Rccl &r = rccl();
    // in place: this rank's block already sits at rank * records_per_rank (the kernels wrote it there)
    const int rc = r.all_gather(d_flat + (uint64_t)comm->rank * records_per_rank, d_flat, (size_t)records_per_rank, kNcclInt64,
                                comm->comm, (hipStream_t)stream);
    if (rc != 0) return trx::fail_msg(TRX_ERR_NO_DEVICE, "ncclAllGather: %s", r.get_error_string ? r.get_error_string(rc) : "error");
    return TRX_OK;
}

int trx_gather_shards_root(trx_comm *comm, trx_hit *d_flat, uint64_t records_per_rank, int root, void *stream) {
    if (!comm || !d_flat) return trx::fail_msg(TRX_ERR_INVALID, "null argument");
    if (root < 0 || root >= comm->world) return trx::fail_msg(TRX_ERR_INVALID, "root %d of %d", root, comm->world);
    if (records_per_rank == 0 || comm->world == 1) return TRX_OK; // the root's own block is already in place
    Rccl &r = rccl();
    if (!r.send || !r.recv || !r.group_start || !r.group_end)
        return trx::fail_msg(TRX_ERR_NO_DEVICE, "librccl.so lacks ncclSend / ncclRecv / ncclGroupStart / ncclGroupEnd");
    // every rank's block goes to the root over its own xGMI link (world - 1 point-to-point transfers in one group);
    // the other ranks move 1/world of what the all-gather moves and receive nothing
    int rc = r.group_start();
    if (rc == 0) {
        if (comm->rank == root) {
            for (int src = 0; src < comm->world && rc == 0; src++)
                if (src != root) rc = r.recv(d_flat + (uint64_t)src * records_per_rank, (size_t)records_per_rank, kNcclInt64, src, comm->comm, (hipStream_t)stream);
        } else {
            rc = r.send(d_flat + (uint64_t)comm->rank * records_per_rank, (size_t)records_per_rank, kNcclInt64, root, comm->comm, (hipStream_t)stream);
        }
        const int rc_end = r.group_end();
        if (rc == 0) rc = rc_end;
    }
    if (rc != 0) return trx::fail_msg(TRX_ERR_NO_DEVICE, "ncclSend/ncclRecv: %s", r.get_error_string ? r.get_error_string(rc) : "error");
    return TRX_OK;
}

int trx_assemble_frames(const trx_hit *d_flat, uint64_t records_per_frame, uint32_t width, uint32_t height, uint32_t world,
                        uint32_t n_frames, trx_hit *d_frames, void *stream) {
    if (!d_flat || !d_frames) return trx::fail_msg(TRX_ERR_INVALID, "null argument");
    if (width == 0 || height == 0 || world == 0 || n_frames == 0) return trx::fail_msg(TRX_ERR_INVALID, "empty frame, world or batch");
    const trx_shard s0 = {0u, world, TRX_LAYOUT_SHARD, 0u};
    if (records_per_frame < (uint64_t)trx_shard_tiles(width, height, s0) * 64u)
        return trx::fail_msg(TRX_ERR_INVALID, "records_per_frame %llu smaller than rank 0's shard", (unsigned long long)records_per_frame);
    const uint64_t total = (uint64_t)width * height * n_frames;
    const uint32_t block = 256;
    const uint64_t grid = (total + block - 1) / block;
    if (grid > 0x7fffffffull) return trx::fail_msg(TRX_ERR_INVALID, "batch too large");
    hipLaunchKernelGGL(k_assemble, dim3((uint32_t)grid), dim3(block), 0, (hipStream_t)stream, d_flat, d_frames, width, height,
                       (width + 7u) / 8u, world, n_frames, records_per_frame);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return trx::fail_msg(TRX_ERR_NO_DEVICE, "assemble launch: %s", hipGetErrorString(e));
    return TRX_OK;
}

} // extern "C"
