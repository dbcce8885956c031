// Multi-GPU exchange: one process per GPU, RCCL over xGMI.  Only two collectives exist on this
// path: the gather of decoded state paths and the sum of Baum-Welch statistics.
#include <rccl/rccl.h>

#include <algorithm>

#include "common.h"

using namespace lnasr;

#define LN_NCCL(expr)                                                                            \
    do {                                                                                         \
        ncclResult_t r__ = (expr);                                                               \
        if (r__ != ncclSuccess)                                                                  \
            return lnasr::fail(LNASR_E_COMM, "%s:%d %s -> %s", __FILE__, __LINE__, #expr,        \
                               ncclGetErrorString(r__));                                         \
    } while (0)

static_assert(sizeof(ncclUniqueId) <= LNASR_UNIQUE_ID_BYTES, "unique id does not fit the ABI blob");

extern "C" {

int lnasr_comm_unique_id(void* id_out) {
    LN_REQUIRE(id_out, "null id_out");
    ncclUniqueId id;
    LN_NCCL(ncclGetUniqueId(&id));
    memset(id_out, 0, LNASR_UNIQUE_ID_BYTES);
    memcpy(id_out, &id, sizeof(id));
    return 0;
}

int lnasr_comm_init(lnasr_ctx* ctx, const void* idb, int rank, int nranks) {
    LN_ENTER(ctx);
    LN_REQUIRE(ctx && idb, "null argument");
    LN_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, "bad rank %d / nranks %d", rank, nranks);
    LN_REQUIRE(!ctx->comm, "communicator already initialised");
    LN_HIP(hipSetDevice(ctx->device));
    ncclUniqueId id;
    memcpy(&id, idb, sizeof(id));
    ncclComm_t comm;
    LN_NCCL(ncclCommInitRank(&comm, nranks, id, rank));
    ctx->comm = comm;
    ctx->rank = rank;
    ctx->nranks = nranks;
    return 0;
}

int lnasr_comm_destroy(lnasr_ctx* ctx) {
    LN_ENTER(ctx);
    if (!ctx || !ctx->comm) return 0;
    (void)hipStreamSynchronize(ctx->stream);
    ncclCommDestroy((ncclComm_t)ctx->comm);
    ctx->comm = nullptr;
    ctx->nranks = 1;
    ctx->rank = 0;
    return 0;
}

int lnasr_comm_info(lnasr_ctx* ctx, int* rank, int* nranks) {
    LN_ENTER(ctx);
    LN_REQUIRE(ctx, "null ctx");
    if (!ctx->comm) return fail(LNASR_E_COMM, "communicator not initialised (lnasr_comm_init)");
    int r = -1, n = 0;   // asked of RCCL itself, not echoed from lnasr_comm_init's arguments
    LN_NCCL(ncclCommUserRank((ncclComm_t)ctx->comm, &r));
    LN_NCCL(ncclCommCount((ncclComm_t)ctx->comm, &n));
    if (rank) *rank = r;
    if (nranks) *nranks = n;
    return 0;
}

int lnasr_allgather_u32(lnasr_ctx* ctx, const uint32_t* local, size_t count, uint32_t* all) {
    LN_ENTER(ctx);
    LN_REQUIRE(ctx && local && all, "null argument");
    if (!ctx->comm) return fail(LNASR_E_COMM, "communicator not initialised (lnasr_comm_init)");
    const size_t nr = (size_t)ctx->nranks;
    const void* d_in;
    void* d_out;
    LN_TRY(stage_in(ctx, local, count * 4, ctx->ws_in5, &d_in));
    LN_TRY(stage_out(ctx, all, count * nr * 4, ctx->ws_out3, &d_out));
    LN_NCCL(ncclAllGather(d_in, d_out, count, ncclUint32, (ncclComm_t)ctx->comm, ctx->stream));
    return finish_out(ctx, all, d_out, count * nr * 4);
}

// State indices below 256 travel as bytes: a quarter of the xGMI traffic of the u32 paths, same result.
__global__ void pack_u8_kernel(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, size_t nwords) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nwords; i += (size_t)gridDim.x * blockDim.x) {
        const uint4 v = reinterpret_cast<const uint4*>(src)[i];
        dst[i] = (v.x & 0xFFu) | ((v.y & 0xFFu) << 8) | ((v.z & 0xFFu) << 16) | (v.w << 24);
    }
}
__global__ void unpack_u8_kernel(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, size_t nwords) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nwords; i += (size_t)gridDim.x * blockDim.x) {
        const uint32_t w = src[i];
        reinterpret_cast<uint4*>(dst)[i] = make_uint4(w & 0xFFu, (w >> 8) & 0xFFu, (w >> 16) & 0xFFu, w >> 24);
    }
}

int lnasr_allgather_paths(lnasr_ctx* ctx, const uint32_t* local, size_t count, int n_states, uint32_t* all) {
    LN_ENTER(ctx);
    LN_REQUIRE(ctx && local && all, "null argument");
    if (!ctx->comm) return fail(LNASR_E_COMM, "communicator not initialised (lnasr_comm_init)");
    if (n_states > 256 || n_states < 1 || count % 4 != 0 || count == 0) return lnasr_allgather_u32(ctx, local, count, all);
    const size_t nr = (size_t)ctx->nranks, nwords = count / 4;
    const void* d_in;
    void* d_out;
    LN_TRY(stage_in(ctx, local, count * 4, ctx->ws_in5, &d_in));
    LN_TRY(stage_out(ctx, all, count * nr * 4, ctx->ws_out3, &d_out));
    LN_TRY(ctx->ws_w3.reserve((nwords + nwords * nr) * 4));
    uint32_t* packed = ctx->ws_w3.as<uint32_t>();
    uint32_t* gathered = packed + nwords;
    const unsigned g1 = (unsigned)std::min<size_t>((nwords + 255) / 256, 4096);
    const unsigned g2 = (unsigned)std::min<size_t>((nwords * nr + 255) / 256, 4096);
    LN_LAUNCH(ctx, "pack_paths_u8", pack_u8_kernel, dim3(g1), dim3(256), 0, (const uint32_t*)d_in, packed, nwords);
    LN_NCCL(ncclAllGather(packed, gathered, nwords, ncclUint32, (ncclComm_t)ctx->comm, ctx->stream));
    LN_LAUNCH(ctx, "unpack_paths_u8", unpack_u8_kernel, dim3(g2), dim3(256), 0, (const uint32_t*)gathered, (uint32_t*)d_out,
              nwords * nr);
    return finish_out(ctx, all, d_out, count * nr * 4);
}

int lnasr_allreduce_f64(lnasr_ctx* ctx, double* buf, size_t n) {
    LN_ENTER(ctx);
    LN_REQUIRE(ctx && buf, "null argument");
    if (!ctx->comm) return fail(LNASR_E_COMM, "communicator not initialised (lnasr_comm_init)");
    const void* d_in;
    LN_TRY(stage_in(ctx, buf, n * 8, ctx->ws_in5, &d_in));
    void* d = const_cast<void*>(d_in);
    LN_NCCL(ncclAllReduce(d, d, n, ncclDouble, ncclSum, (ncclComm_t)ctx->comm, ctx->stream));
    return finish_out(ctx, buf, d, n * 8);
}

}  // extern "C"
