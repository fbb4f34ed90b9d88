// comm.cpp — N-GPU support: one cloud pair sharded by source-point blocks, one process per
// GPU, RCCL all-reduce of the per-iteration sums over xGMI (include/rsreg.h "N-GPU").
// The reference has no distributed path; this is new work, not a port.
//
// RCCL is bound at the FIRST rsreg_comm_* call (dlopen), not when librsreg.so is loaded: librccl.so is a 570 MB library
// whose device code objects are registered with the HIP runtime by its static initialisers, and a process that registers
// clouds on one GPU -- everything the reference does (main.cpp:85) -- never needs it.  As a link-time dependency it was
// mapped, relocated and registered before main() in every such process (profiles/r06_cold_run.txt).
#include <rccl/rccl.h>   // types and prototypes only

#include <dlfcn.h>

#include <cstring>
#include <mutex>

#include "rsreg_ctx.hpp"

using namespace rsreg;

namespace {
struct Rccl {
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string error;   // why it could not be bound (empty: bound)
};

// nullptr-free: check `.error` first
const Rccl &rccl()
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        // The RCCL that belongs to the HIP runtime THIS library runs on: the file beside that libamdhip64, by path.  By soname
        // alone the loader hands out whatever "librccl.so.1" the process has mapped already -- in a Python process that is
        // the copy bundled with PyTorch, built against PyTorch's own bundled runtime, and ncclCommInitRank then fails on ours
        // ("unhandled cuda error").  Local scope: two RCCLs in one process must not bind each other's symbols.
        void *h = nullptr;
        std::string beside;
        Dl_info info;
        if (dladdr(reinterpret_cast<const void *>(&hipGetDeviceCount), &info) && info.dli_fname) {
            beside = info.dli_fname;
            const size_t slash = beside.rfind('/');
            beside = slash == std::string::npos ? std::string() : beside.substr(0, slash + 1);
        }
        for (const std::string &name : {beside + "librccl.so.1", beside + "librccl.so", std::string("/opt/rocm/lib/librccl.so.1"), std::string("librccl.so.1")}) {
            if (name.empty() || (name[0] != '/' && name != "librccl.so.1")) continue;
            h = dlopen(name.c_str(), RTLD_NOW | RTLD_LOCAL);
            if (h) break;
        }
        if (!h) {
            const char *e = dlerror();
            r.error = std::string("librccl.so.1 cannot be loaded: ") + (e ? e : "unknown error");
            return;
        }
        auto bind = [&](const char *sym) {
            void *p = dlsym(h, sym);
            if (!p && r.error.empty()) r.error = std::string("librccl.so.1 has no symbol ") + sym;
            return p;
        };
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(bind("ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(bind("ncclCommInitRank"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(bind("ncclCommDestroy"));
        r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(bind("ncclAllReduce"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(bind("ncclGetErrorString"));
    });
    return r;
}
}  // namespace

#define RSREG_NCCL(ctx, expr)                                                          \
    do {                                                                               \
        ncclResult_t _r = (expr);                                                      \
        if (_r != ncclSuccess) {                                                       \
            return fail((ctx), RSREG_ERR_RCCL, (std::string(#expr) + ": " + rccl().GetErrorString(_r)).c_str());   \
        }                                                                              \
    } while (0)

extern "C" {

int rsreg_comm_unique_id(uint8_t id[RSREG_UNIQUE_ID_BYTES])
{
    if (!id) return RSREG_ERR_INVALID_ARG;
    static_assert(sizeof(ncclUniqueId) <= RSREG_UNIQUE_ID_BYTES, "unique id size");
    ncclUniqueId u;
    if (!rccl().error.empty() || rccl().GetUniqueId(&u) != ncclSuccess) return RSREG_ERR_RCCL;
    std::memset(id, 0, RSREG_UNIQUE_ID_BYTES);
    std::memcpy(id, &u, sizeof(u));
    return RSREG_OK;
}

int rsreg_comm_init(rsreg_ctx *ctx, const uint8_t id[RSREG_UNIQUE_ID_BYTES], int rank, int nranks)
{
    if (!ctx || !id || nranks < 1 || rank < 0 || rank >= nranks) return RSREG_ERR_INVALID_ARG;
    if (ctx->comm) return fail(ctx, RSREG_ERR_STATE, "communicator already initialised");
    if (!rccl().error.empty()) return fail(ctx, RSREG_ERR_RCCL, rccl().error.c_str());
    RSREG_HIP(ctx, hipSetDevice(ctx->device));
    ncclUniqueId u;
    std::memcpy(&u, id, sizeof(u));
    ncclComm_t comm = nullptr;
    RSREG_NCCL(ctx, rccl().CommInitRank(&comm, nranks, u, rank));
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
        (void)rccl().CommDestroy(static_cast<ncclComm_t>(ctx->comm));   // (a communicator exists: RCCL is bound)
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
    RSREG_NCCL(ctx, rccl().AllReduce(d_buf, d_buf, (size_t)count, ncclDouble, ncclSum,
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
