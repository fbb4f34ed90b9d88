// ndt.hip — NDT half of the C ABI (placeholder until the kernel set lands).
#include "rsreg_ctx.hpp"
using namespace rsreg;
extern "C" {
void rsreg_ndt_params_default(rsreg_ndt_params *p)
{
    if (!p) return;
    std::memset(p, 0, sizeof(*p));
    p->max_iterations = 35; p->transformation_epsilon = 0.1; p->step_size = 0.1; p->resolution = 1.0; p->outlier_ratio = 0.55;
}
void rsreg_ndt_params_reference(rsreg_ndt_params *p)
{
    if (!p) return;
    rsreg_ndt_params_default(p);
    p->transformation_epsilon = 0.01; p->step_size = 0.1; p->resolution = 1.0; p->max_iterations = 50;
}
int rsreg_ndt_set_target(rsreg_ctx *ctx, const void *, size_t, size_t, int, double) { return fail(ctx, RSREG_ERR_STATE, "NDT not built yet"); }
int rsreg_ndt_align(rsreg_ctx *ctx, const void *, size_t, size_t, int, const float *, const rsreg_ndt_params *, rsreg_ndt_result *, void *, size_t) { return fail(ctx, RSREG_ERR_STATE, "NDT not built yet"); }
int rsreg_ndt_derivatives(rsreg_ctx *ctx, const void *, size_t, size_t, int, const double *, double *, double *, double *) { return fail(ctx, RSREG_ERR_STATE, "NDT not built yet"); }
int rsreg_ndt_get_voxels(rsreg_ctx *ctx, int32_t *, double *, int32_t *, int32_t) { return fail(ctx, RSREG_ERR_STATE, "NDT not built yet"); }
}
