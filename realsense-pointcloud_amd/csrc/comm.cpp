// comm.cpp — N-GPU support: one cloud pair sharded by source-point blocks, one process per
// GPU, RCCL all-reduce of the per-iteration sums over xGMI (include/rsreg.h "N-GPU").
// The reference has no distributed path; this is new work, not a port.
#include <rccl/rccl.h>

#include <cstring>

#include "rsreg_ctx.hpp"

using namespace rsreg;

#define RSREG_NCCL(ctx, expr)                                                          \
    do {                                                                               \
        ncclResult_t _r = (expr);                                                      \
        if (_r != ncclSuccess) {                                                       \
            return fail((ctx), RSREG_ERR_RCCL, (std::string(#expr) + ": " + ncclGetErrorString(_r)).c_str());   \
        }                                                                              \
    } while (0)

extern "C" {

int rsreg_comm_unique_id(uint8_t id[RSREG_UNIQUE_ID_BYTES])
{
    if (!id) return RSREG_ERR_INVALID_ARG;
    static_assert(sizeof(ncclUniqueId) <= RSREG_UNIQUE_ID_BYTES, "unique id size");
    ncclUniqueId u;
    if (ncclGetUniqueId(&u) != ncclSuccess) return RSREG_ERR_RCCL;
    std::memset(id, 0, RSREG_UNIQUE_ID_BYTES);
    std::memcpy(id, &u, sizeof(u));
    return RSREG_OK;
}

int rsreg_comm_init(rsreg_ctx *ctx, const uint8_t id[RSREG_UNIQUE_ID_BYTES], int rank, int nranks)
{
    if (!ctx || !id || nranks < 1 || rank < 0 || rank >= nranks) return RSREG_ERR_INVALID_ARG;
    if (ctx->comm) return fail(ctx, RSREG_ERR_STATE, "communicator already initialised");
    RSREG_HIP(ctx, hipSetDevice(ctx->device));
    ncclUniqueId u;
    std::memcpy(&u, id, sizeof(u));
    ncclComm_t comm = nullptr;
    RSREG_NCCL(ctx, ncclCommInitRank(&comm, nranks, u, rank));
    ctx->comm = comm;
    ctx->rank = rank;
    ctx->nranks = nranks;
    RSREG_HIP(ctx, ctx->d_comm.reserve(64 * sizeof(double)));
    return RSREG_OK;
}

int rsreg_comm_destroy(rsreg_ctx *ctx)
{
    if (!ctx) return RSREG_ERR_INVALID_ARG;
    if (ctx->comm) {
        (void)ncclCommDestroy(static_cast<ncclComm_t>(ctx->comm));
        ctx->comm = nullptr;
    }
    ctx->rank = 0;
    ctx->nranks = 1;
    return RSREG_OK;
}

// in-place sum of `count` doubles already in HBM, on the ctx stream (no host sync)
int rsreg_comm_allreduce_device_(rsreg_ctx *ctx, double *d_buf, int count)
{
    if (!ctx->comm) return fail(ctx, RSREG_ERR_STATE, "rsreg_comm_init not called");
    RSREG_NCCL(ctx, ncclAllReduce(d_buf, d_buf, (size_t)count, ncclDouble, ncclSum,
                                  static_cast<ncclComm_t>(ctx->comm), ctx->stream));
    return RSREG_OK;
}

int rsreg_comm_allreduce_f64(rsreg_ctx *ctx, double *host_buf, int count)
{
    if (!ctx || !host_buf || count < 1 || count > 64) return RSREG_ERR_INVALID_ARG;
    if (ctx->nranks == 1 && !ctx->comm) return RSREG_OK;
    RSREG_HIP(ctx, hipSetDevice(ctx->device));
    RSREG_HIP(ctx, hipMemcpyAsync(ctx->d_comm.ptr, host_buf, count * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    int rc = rsreg_comm_allreduce_device_(ctx, ctx->d_comm.as<double>(), count);
    if (rc) return rc;
    RSREG_HIP(ctx, hipMemcpyAsync(host_buf, ctx->d_comm.ptr, count * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    RSREG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return RSREG_OK;
}

}  // extern "C"
