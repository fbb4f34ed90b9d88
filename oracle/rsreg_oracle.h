/*
 * rsreg_oracle.h — CPU restatement of the reference's pair-registration path.
 *
 * TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may import, link or execute anything under oracle/.  The product
 * (include/rsreg.h, realsense-pointcloud_amd/) never does and has no CPU fallback.
 *
 * PARITY UNPINNED: the arithmetic of the reference path lives in PCL >= 1.9
 * (find_package(PCL 1.9 REQUIRED), reference CMakeLists.txt:11), which is neither vendored
 * under /root/reference nor installed in this image, and the reference holds no test,
 * golden vector or fixture for the path.  This file restates PCL
 * 1.9.1's published algorithms from SURVEY.md Appendix A (A.1-A.8) and is cross-checked
 * against an independent numpy/scipy implementation (oracle/make_golden.py), not against
 * PCL itself.
 *
 * Reference call sites followed:
 *   src/incremental_icp.hpp:46-49,54-64   (ICP params, voxel, align, transform, concat)
 *   src/icp_edge_based_registration.hpp:42-52,75-120
 *   src/ndt_edge_based_registration.hpp:38-50,68-108
 */
#ifndef RSREG_ORACLE_H_
#define RSREG_ORACLE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_NUM_SUMS 17

enum { ORC_CONV_NOT_CONVERGED = 0, ORC_CONV_ITERATIONS, ORC_CONV_TRANSFORM, ORC_CONV_ABS_MSE,
       ORC_CONV_REL_MSE, ORC_CONV_NO_CORRESPONDENCES, ORC_CONV_FAILURE_AFTER_MAX_ITERATIONS };

enum { ORC_CRITERIA_PCL = 0, ORC_CRITERIA_FIXED = 1 };
enum { ORC_ACCUM_F32 = 0 /* PCL-like float sums */, ORC_ACCUM_F64 = 1 };
enum { ORC_NN_KDTREE = 0, ORC_NN_BRUTE = 1 };

typedef struct orc_icp_params {
    int32_t max_iterations;
    int32_t criteria_mode;
    int32_t accum_mode;
    int32_t nn_mode;
    int32_t dedup_target;   /* drop exact-duplicate target xyz before the tree build        */
    int32_t num_threads;    /* OpenMP threads over queries (1 = PCL-like single thread)      */
    double max_correspondence_distance;
    double transformation_epsilon;
    double transformation_rotation_epsilon;
    double euclidean_fitness_epsilon;
    /* optional correspondence filters (off in the reference: its CorrespondenceRejectorTrimmed is constructed and
     * never attached, incremental_icp.hpp:38) */
    int32_t use_reciprocal;      /* CorrespondenceEstimation::determineReciprocalCorrespondences               */
    int32_t reserved2;
    double trim_overlap_ratio;   /* CorrespondenceRejectorTrimmed::setOverlapRatio, 0 < r < 1; else no rejector  */
} orc_icp_params;

typedef struct orc_icp_result {
    float transform[16];
    int32_t converged, state, iterations, reserved;
    uint64_t n_correspondences;
    double mse;
    double sums_last[ORC_NUM_SUMS];
    double sec_build, sec_search, sec_total;
} orc_icp_result;

typedef struct orc_icp orc_icp;

void orc_icp_params_default(orc_icp_params *p);
void orc_icp_params_reference(orc_icp_params *p);

orc_icp *orc_icp_create(void);
void orc_icp_destroy(orc_icp *o);
int orc_icp_set_target(orc_icp *o, const void *pts, size_t n, size_t stride, int is_dense,
                       int dedup, int num_threads);
int orc_icp_set_source(orc_icp *o, const void *pts, size_t n, size_t stride, int is_dense);
int orc_icp_begin(orc_icp *o, const float *guess, const orc_icp_params *params);
int orc_icp_search(orc_icp *o, int32_t *index_out, float *sqr_dist_out);
int orc_icp_sums(orc_icp *o, double sums[ORC_NUM_SUMS]);
int orc_icp_update(orc_icp *o, const double *sums /*NULL: estimate from correspondences*/,
                   float *t_inc_out, int *done);
int orc_icp_end(orc_icp *o, orc_icp_result *result, void *aligned_out, size_t out_stride);
int orc_icp_align(orc_icp *o, const float *guess, const orc_icp_params *params,
                  orc_icp_result *result, void *aligned_out, size_t out_stride);
/* current (transformed) source xyz, n x 3 floats */
int orc_icp_get_current(orc_icp *o, float *xyz_out);

int orc_umeyama_from_sums(const double sums[ORC_NUM_SUMS], float t_out[16]);
void orc_mat4_mul(const float *a, const float *b, float *c); /* column-major c = a*b, f32 */

int orc_transform_cloud(const void *in, void *out, size_t n, size_t stride, int is_dense,
                        const float transform[16]);
int orc_approx_voxel_grid(const void *in, size_t n, size_t stride, const float leaf[3],
                          void *out, size_t *n_out);

/* ---- edge extractor (src/edge_extractor.hpp:7-39: only the RGB-Canny edges are returned) ---- */
/* indices (ascending) of the points of an organized cloud whose colour image has a Canny edge;
 * indices_out must hold width*height entries (or be NULL to count only) */
int orc_edge_rgb_canny(const void *pts, size_t stride, uint32_t width, uint32_t height, float t_low,
                       float t_high, int32_t *indices_out, size_t *n_out);

/* ---- NDT ---- */
typedef struct orc_ndt_params {
    int32_t max_iterations, reserved;
    double transformation_epsilon, step_size, resolution, outlier_ratio;
} orc_ndt_params;

typedef struct orc_ndt_result {
    float transform[16];
    int32_t converged, iterations;
    double trans_probability, score;
    int32_t n_voxels, n_derivative_passes;
    double sec_total;
} orc_ndt_result;

typedef struct orc_ndt orc_ndt;
void orc_ndt_params_default(orc_ndt_params *p);
void orc_ndt_params_reference(orc_ndt_params *p);
orc_ndt *orc_ndt_create(void);
void orc_ndt_destroy(orc_ndt *o);
void orc_ndt_set_centroid_mode(orc_ndt *o, int mode); /* 0 PCL f32 running sum, 1 f64 mean rounded */
int orc_ndt_set_target(orc_ndt *o, const void *pts, size_t n, size_t stride, int is_dense,
                       double resolution);
int orc_ndt_get_centroids(orc_ndt *o, float *centroids /*3 each*/, int32_t capacity);
int orc_ndt_get_voxels(orc_ndt *o, int32_t *n_voxels, double *mean_cov_icov, int32_t *counts,
                       int32_t capacity);
int orc_ndt_derivatives(orc_ndt *o, const void *src, size_t n, size_t stride, int is_dense,
                        const double pose[6], const orc_ndt_params *params, double *score,
                        double gradient[6], double hessian[36]);
int orc_ndt_align(orc_ndt *o, const void *src, size_t n, size_t stride, int is_dense,
                  const float *guess, const orc_ndt_params *params, orc_ndt_result *result,
                  void *aligned_out, size_t out_stride);

#ifdef __cplusplus
}
#endif
#endif
