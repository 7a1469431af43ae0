// shard.hip -- the multi-GPU entry points of libcofhe_hip.so: row sharding of ciphertext tensors over one
// process per GPU and the one collective the path has, the all-gather that reassembles a row-sharded result.
//
// Row i of add_ciphertext_tensors / scal_ciphertext_tensors' result needs only row i of the ciphertext operand
// (reference loops: include/x86_64/cpu_cryptosystem_tensor_ops.inl:242-264 and :396-417), so ranks work on their
// row blocks without any exchange; the reference itself has no multi-device path (its compute nodes each hold a whole
// tensor).  Records have a fixed size, so the gather is a plain byte collective on the device buffers: RCCL
// (ncclAllGather, or one ncclBroadcast per rank in a group when the row blocks are ragged) on the caller's stream.
// RCCL is bound when the first communicator is made (dlopen), so single-GPU users never load it.
#include <dlfcn.h>

#include <cstring>
#include <vector>

#include "ctx.hpp"

#include <rccl/rccl.h>

using namespace cofhe;

namespace {
struct Rccl {
    void *h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl g_rccl;
std::mutex g_rccl_mu;

int bind_rccl() {
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    if (g_rccl.h) return COFHE_HIP_OK;
    void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return fail(COFHE_HIP_EHIP, std::string("librccl.so.1 not loadable: ") + dlerror());
    Rccl r;
    bool ok = true;
    auto sym = [&](const char *n) {
        void *p = dlsym(h, n);
        if (!p) ok = false;
        return p;
    };
    r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
    r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
    r.AllGather = (decltype(r.AllGather))sym("ncclAllGather");
    r.Broadcast = (decltype(r.Broadcast))sym("ncclBroadcast");
    r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
    r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
    r.CommCount = (decltype(r.CommCount))sym("ncclCommCount");
    r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
    if (!ok) {
        dlclose(h);
        return fail(COFHE_HIP_EHIP, "librccl.so.1 lacks a collective entry point");
    }
    r.h = h;
    g_rccl = r;
    return COFHE_HIP_OK;
}
}  // namespace

#define NCCLCHK(expr)                                                                                          \
    do {                                                                                                       \
        ncclResult_t r_ = (expr);                                                                              \
        if (r_ != ncclSuccess) return fail(COFHE_HIP_EHIP, std::string(#expr) + ": " + g_rccl.GetErrorString(r_)); \
    } while (0)

struct cofhe_hip_comm {
    ncclComm_t comm;
    uint32_t world, rank;
    int device;
    bool force_grouped = false;      // cofhe_hip_comm_set_option("force_grouped_broadcast"): take the ragged route for equal blocks too
};

extern "C" {

void cofhe_hip_shard_rows(uint64_t n_rows, uint32_t world, uint32_t rank, uint64_t *row0, uint64_t *n_local) {
    // contiguous blocks, the remainder rows to the low ranks (cofhe_amd/shard.py: row_partition)
    if (world == 0) world = 1;
    const uint64_t base = n_rows / world, rem = n_rows % world;
    const uint64_t r = rank < world ? rank : world - 1;
    if (row0) *row0 = r * base + (r < rem ? r : rem);
    if (n_local) *n_local = base + (r < rem ? 1 : 0);
}

int cofhe_hip_gather_plan(uint64_t n_rows, uint64_t row_bytes, uint32_t world, uint64_t *offsets, uint64_t *counts, int *uniform) {
    // what cofhe_hip_all_gather_rows will do, as data (host only, no GPU and no RCCL): block r of the assembled tensor is
    // bytes [offsets[r], offsets[r] + counts[r]); equal blocks travel in one ncclAllGather, ragged ones as one
    // ncclBroadcast per non-empty block inside a group
    if (world == 0 || !offsets || !counts) return fail(COFHE_HIP_EINVAL, "null argument or empty world");
    if (row_bytes != 0 && n_rows > (~(uint64_t)0) / row_bytes) return fail(COFHE_HIP_EINVAL, "tensor too large");
    for (uint32_t r = 0; r < world; r++) {
        uint64_t r0, cnt;
        cofhe_hip_shard_rows(n_rows, world, r, &r0, &cnt);
        offsets[r] = r0 * row_bytes;
        counts[r] = cnt * row_bytes;
    }
    if (uniform) *uniform = (n_rows % world == 0) ? 1 : 0;
    return COFHE_HIP_OK;
}

int cofhe_hip_comm_unique_id(uint8_t id[COFHE_HIP_COMM_ID_BYTES]) {
    static_assert(sizeof(ncclUniqueId) <= COFHE_HIP_COMM_ID_BYTES, "id buffer too small");
    if (!id) return fail(COFHE_HIP_EINVAL, "null argument");
    if (int rc = bind_rccl()) return rc;
    ncclUniqueId u;
    NCCLCHK(g_rccl.GetUniqueId(&u));
    memset(id, 0, COFHE_HIP_COMM_ID_BYTES);
    memcpy(id, &u, sizeof(u));
    return COFHE_HIP_OK;
}

int cofhe_hip_comm_create(cofhe_hip_ctx *ctx, const uint8_t id[COFHE_HIP_COMM_ID_BYTES], uint32_t world, uint32_t rank,
                          cofhe_hip_comm **out) {
    if (!ctx || !id || !out) return fail(COFHE_HIP_EINVAL, "null argument");
    if (world == 0 || rank >= world) return fail(COFHE_HIP_EINVAL, "rank outside the world");
    if (int rc = bind_rccl()) return rc;
    HIPCHK(hipSetDevice(ctx->device));
    ncclUniqueId u;
    memcpy(&u, id, sizeof(u));
    ncclComm_t c;
    NCCLCHK(g_rccl.CommInitRank(&c, (int)world, u, (int)rank));
    *out = new cofhe_hip_comm{c, world, rank, ctx->device};
    return COFHE_HIP_OK;
}

int cofhe_hip_comm_info(cofhe_hip_comm *comm, uint32_t *world, uint32_t *rank, uint32_t *rccl_nranks) {
    if (!comm) return fail(COFHE_HIP_EINVAL, "null argument");
    if (world) *world = comm->world;
    if (rank) *rank = comm->rank;
    if (rccl_nranks) {
        int n = 0;
        NCCLCHK(g_rccl.CommCount(comm->comm, &n));          // what RCCL itself says the communicator spans
        *rccl_nranks = (uint32_t)n;
    }
    return COFHE_HIP_OK;
}

int cofhe_hip_comm_set_option(cofhe_hip_comm *comm, const char *name, int64_t value) {
    if (!comm || !name) return fail(COFHE_HIP_EINVAL, "null argument");
    if (strcmp(name, "force_grouped_broadcast") == 0) {
        comm->force_grouped = value != 0;
        return COFHE_HIP_OK;
    }
    return fail(COFHE_HIP_EINVAL, std::string("unknown option: ") + name);
}

void cofhe_hip_comm_destroy(cofhe_hip_comm *comm) {
    if (!comm) return;
    (void)hipSetDevice(comm->device);
    (void)g_rccl.CommDestroy(comm->comm);
    delete comm;
}

int cofhe_hip_all_gather_rows(cofhe_hip_ctx *ctx, cofhe_hip_comm *comm, const void *d_local, uint64_t n_rows, uint64_t row_bytes,
                              void *d_out, void *stream) {
    if (!ctx || !comm || !d_out) return fail(COFHE_HIP_EINVAL, "null argument");
    if (comm->device != ctx->device) return fail(COFHE_HIP_EINVAL, "communicator and context are on different devices");
    HIPCHK(hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t)stream;
    if (comm->world > 4096) return fail(COFHE_HIP_EINVAL, "world too large");
    std::vector<uint64_t> offs(comm->world), cnts(comm->world);
    int uniform = 0;
    if (int rc = cofhe_hip_gather_plan(n_rows, row_bytes, comm->world, offs.data(), cnts.data(), &uniform)) return rc;
    if (cnts[comm->rank] != 0 && !d_local) return fail(COFHE_HIP_EINVAL, "null argument");
    if (uniform && !comm->force_grouped) {
        NCCLCHK(g_rccl.AllGather(d_local, d_out, (size_t)cnts[comm->rank], ncclUint8, comm->comm, st));
        return COFHE_HIP_OK;
    }
    // ragged blocks: every rank broadcasts its block to its place in the result, one fused group
    NCCLCHK(g_rccl.GroupStart());
    for (uint32_t r = 0; r < comm->world; r++) {
        if (cnts[r] == 0) continue;
        uint8_t *dst = (uint8_t *)d_out + offs[r];
        const void *src = r == comm->rank ? d_local : (const void *)dst;
        ncclResult_t e = g_rccl.Broadcast(src, dst, (size_t)cnts[r], ncclUint8, (int)r, comm->comm, st);
        if (e != ncclSuccess) {
            (void)g_rccl.GroupEnd();
            return fail(COFHE_HIP_EHIP, std::string("ncclBroadcast: ") + g_rccl.GetErrorString(e));
        }
    }
    NCCLCHK(g_rccl.GroupEnd());
    return COFHE_HIP_OK;
}

}  // extern "C"
