/*
 * rsreg.h — C ABI of the MI355X-native pairwise point-cloud registration engine.
 *
 * This is the drop-in boundary for the ICP / NDT pair-registration hot path of
 * hyunminch/realsense-pointcloud.  The reference has no FFI of its own; the calls that
 * cross this boundary are the PCL calls its three registration schemes make.  Every entry
 * point below names the reference call site (file:line under the reference's src/) and
 * the PCL method it stands in for.
 *
 * Conventions
 *   - Points are handed over as an array of records `stride` bytes apart whose first 12
 *     bytes are `float x, y, z` (pcl::PointXYZRGB: stride 32, rgb at byte 16).  Only xyz
 *     is ever sent to the GPU; colour stays on the host (PCL's ICP/NDT ignore it too).
 *   - 4x4 transforms are 16 floats, COLUMN-major (memcpy-compatible with Eigen::Matrix4f).
 *   - Every function returns an rsreg_status (0 = ok, < 0 = error).  Nothing throws across
 *     the ABI.  "Did not converge" is a successful call with result->converged == 0.
 *   - A ctx is bound to one device + one HIP stream and is not thread-safe; different
 *     ctxs are independent.  Host pointers are never retained after a call returns.
 *   - *_device variants take pointers to memory already resident in HBM (same record
 *     layout); the plain variants take host pointers and copy xyz up themselves.
 *   - There is NO CPU fallback: without a usable HIP device every compute entry point
 *     returns RSREG_ERR_NO_DEVICE.
 */
#ifndef RSREG_H_
#define RSREG_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RSREG_VERSION_MAJOR 0
#define RSREG_VERSION_MINOR 4

typedef struct rsreg_ctx rsreg_ctx;

typedef enum rsreg_status {
    RSREG_OK = 0,
    RSREG_ERR_INVALID_ARG = -1,
    RSREG_ERR_EMPTY_CLOUD = -2,
    RSREG_ERR_HIP = -3,
    RSREG_ERR_RCCL = -4,
    RSREG_ERR_NO_TARGET = -5,
    RSREG_ERR_NO_DEVICE = -6,
    RSREG_ERR_ALLOC = -7,
    RSREG_ERR_NO_SOURCE = -8,
    RSREG_ERR_STATE = -9
} rsreg_status;

/* Mirrors pcl::registration::DefaultConvergenceCriteria::ConvergenceState (same order). */
typedef enum rsreg_convergence_state {
    RSREG_CONV_NOT_CONVERGED = 0,
    RSREG_CONV_ITERATIONS = 1,
    RSREG_CONV_TRANSFORM = 2,
    RSREG_CONV_ABS_MSE = 3,
    RSREG_CONV_REL_MSE = 4,
    RSREG_CONV_NO_CORRESPONDENCES = 5,
    RSREG_CONV_FAILURE_AFTER_MAX_ITERATIONS = 6
} rsreg_convergence_state;

/* How the iteration loop decides to stop. */
typedef enum rsreg_criteria_mode {
    RSREG_CRITERIA_PCL = 0,   /* pcl DefaultConvergenceCriteria (reference behaviour)      */
    RSREG_CRITERIA_FIXED = 1  /* run exactly max_iterations iterations (benchmark mode)    */
} rsreg_criteria_mode;

/* Which kernels one ICP iteration is built from. Results are bit-identical across modes. */
typedef enum rsreg_pipeline_mode {
    RSREG_PIPELINE_STAGED = 0, /* nn_search -> cov_reduce -> transform_reject (3 kernels)   */
    RSREG_PIPELINE_FUSED = 1,  /* one fused transform+NN+reject+sums kernel per iteration  */
    RSREG_PIPELINE_DEVICE_LOOP = 2 /* FUSED, and with RSREG_CRITERIA_FIXED the 3x3 solve and the
                                  composition also run on the device: all iterations are queued
                                  without a host round trip (same arithmetic, same result);
                                  with the PCL criteria it behaves as FUSED */
} rsreg_pipeline_mode;

/*
 * ICP parameters = the setters the reference calls on pcl::IterativeClosestPoint
 *   setMaximumIterations / setMaxCorrespondenceDistance / setTransformationEpsilon /
 *   setEuclideanFitnessEpsilon — incremental_icp.hpp:46-49,
 *   icp_edge_based_registration.hpp:42-45,49-52, ndt_edge_based_registration.hpp:47-50.
 * rsreg_icp_params_default() fills PCL's defaults (10, sqrt(DBL_MAX), 0, 0, -DBL_MAX);
 * rsreg_icp_params_reference() fills the reference's constants (100, 0.01, 1, 0, 1000).
 */
typedef struct rsreg_icp_params {
    int32_t max_iterations;
    int32_t criteria_mode;                   /* rsreg_criteria_mode */
    int32_t pipeline_mode;                   /* rsreg_pipeline_mode */
    int32_t reserved0;
    double max_correspondence_distance;
    double transformation_epsilon;
    double transformation_rotation_epsilon;  /* <= 0: use 1 - transformation_epsilon (PCL) */
    double euclidean_fitness_epsilon;
    /* Optional correspondence filters (both off by default and in rsreg_icp_params_reference: the reference
     * constructs a CorrespondenceRejectorTrimmed and never attaches it, incremental_icp.hpp:38,
     * icp_edge_based_registration.hpp:36, ndt_edge_based_registration.hpp:33).  Either one makes the
     * iteration run as RSREG_PIPELINE_STAGED. */
    int32_t use_reciprocal_correspondences;  /* icp.setUseReciprocalCorrespondences(true): a pair (s, t) is kept only if s
                                                is also the nearest source point of t (lowest index among equidistant ones) */
    int32_t reserved1;
    double trim_overlap_ratio;               /* CorrespondenceRejectorTrimmed::setOverlapRatio(r), 0 < r < 1: of the gated
                                                pairs the floor(r * count) closest are kept (equal distances: lowest
                                                source index first); <= 0 or >= 1: no rejector */
} rsreg_icp_params;

/*
 * NDT parameters = the setters the reference calls on pcl::NormalDistributionsTransform
 *   setTransformationEpsilon(0.01) / setStepSize(0.1) / setResolution(1.0) /
 *   setMaximumIterations(50) — ndt_edge_based_registration.hpp:38-43.
 */
typedef struct rsreg_ndt_params {
    int32_t max_iterations;
    int32_t reserved0;
    double transformation_epsilon;
    double step_size;
    double resolution;
    double outlier_ratio;                    /* PCL default 0.55 */
} rsreg_ndt_params;

/* The 17 sums one ICP iteration reduces the correspondences to (all f64):
 *   [0] n   [1..3] sum p   [4..6] sum q   [7..15] sum q_i * p_j (row-major i,j)   [16] sum d^2
 * p = transformed source point, q = its matched target point; only pairs that pass the
 * distance gate contribute.  These are what an N-GPU run all-reduces. */
#define RSREG_NUM_SUMS 17

typedef struct rsreg_icp_result {
    float transform[16];        /* final_transformation_, column-major                      */
    int32_t converged;          /* icp.hasConverged()                                       */
    int32_t state;              /* rsreg_convergence_state                                  */
    int32_t iterations;         /* nr_iterations_                                           */
    int32_t reserved0;
    uint64_t n_correspondences; /* pairs accepted in the last iteration                     */
    double mse;                 /* mean squared distance of those pairs (last iteration)    */
    double sums_last[RSREG_NUM_SUMS]; /* the 17 sums of the last iteration                  */
    /* device-time breakdown of this call (ms, HIP events on the ctx stream); 0 if profiling off */
    double ms_total;
    double ms_nn;               /* dominant kernel: NN search (or the fused kernel)         */
    double ms_reduce;           /* staged: the sums kernels; fused: all time between consecutive search
                                   kernels (final reduce + host round trip or device solve)            */
    double ms_transform;
    int32_t n_nn_launches;
    int32_t n_scheduled_launches; /* ... of them launched from the tile schedule (fused dense kernel: DESIGN.md §4)       */
    double ms_allreduce;        /* N > 1 ranks: the all-reduce of the 17 sums, all iterations (0.3; 0 with one rank)   */
} rsreg_icp_result;

typedef struct rsreg_ndt_result {
    float transform[16];
    int32_t converged;
    int32_t iterations;
    double trans_probability;   /* ndt.getTransformationProbability()                       */
    double score;
    int32_t n_voxels;           /* valid target voxels (>= 6 points, invertible covariance)  */
    int32_t n_derivative_passes;
    double ms_total;
    double ms_derivatives;
} rsreg_ndt_result;

/* ---- library / device ------------------------------------------------------------ */
int rsreg_version(void);                        /* major*1000 + minor */
const char *rsreg_status_string(int status);
const char *rsreg_last_error(const rsreg_ctx *ctx); /* detail of the last failure on ctx   */
int rsreg_device_count(int *count);

/* stream: a hipStream_t to run on (e.g. torch's current stream), or NULL for a new one. */
int rsreg_ctx_create(int device_id, void *stream, rsreg_ctx **out);
int rsreg_ctx_destroy(rsreg_ctx *ctx);
int rsreg_ctx_synchronize(rsreg_ctx *ctx);
/* What a frame loop is about to need, requested ahead of the need (engine extra; the schemes call it when registration() starts:
 * types.hpp:19, main.cpp:85 -- the first registration() of a process otherwise creates these one by one on its critical
 * path): the context's upload / source / download streams and, with RSREG_PREPARE_SIDE_STREAMS, the three side streams
 * (hardware queues: ~12 ms each for a process's first four), the pinned staging buffers of the upload and download workers
 * for frames of `frame_bytes` (0: none), and one device buffer of `model_bytes` for a cloud that will grow to that size
 * (0: none; the merged model of IncrementalICP then grows without re-allocation).  Returns at once: a thread of the context
 * makes them while the caller goes on; every entry point that needs one of them waits for that thread first.  Call it while no
 * upload or download of the context is in flight (between registrations).  Optional: without it everything is created at first
 * use, as before. */
#define RSREG_PREPARE_SIDE_STREAMS 1u
int rsreg_ctx_prepare(rsreg_ctx *ctx, size_t frame_bytes, size_t model_bytes, unsigned flags);
int rsreg_ctx_set_profiling(rsreg_ctx *ctx, int enabled);

void rsreg_icp_params_default(rsreg_icp_params *p);
void rsreg_icp_params_reference(rsreg_icp_params *p);
void rsreg_ndt_params_default(rsreg_ndt_params *p);
void rsreg_ndt_params_reference(rsreg_ndt_params *p);

/* ---- ICP: pcl::IterativeClosestPoint<PointXYZRGB,PointXYZRGB> ------------------------ */

/* icp.setInputTarget(cloud) + the search-structure build PCL does in initCompute()
 * (incremental_icp.hpp:58, icp_edge...hpp:79,109, ndt_edge...hpp:97).  Builds the
 * uniform-grid index over the finite target points.  The grid cell size is derived from
 * max_correspondence_distance, so that must be known here.
 * The build is QUEUED on the context's stream and may not be finished when the call returns (its two counts are
 * taken over at the next call that waits for the stream): a host buffer has been read by then (a second
 * rsreg_icp_set_target first waits for the queued build that still reads the staged records); a DEVICE buffer
 * (d_points) must stay alive and unchanged until the next synchronising call on the context -- rsreg_icp_align /
 * rsreg_icp_begin, rsreg_icp_grid_info or rsreg_ctx_synchronize -- has returned. */
int rsreg_icp_set_target(rsreg_ctx *ctx, const void *points, size_t n, size_t stride,
                         int is_dense, double max_correspondence_distance);
int rsreg_icp_set_target_device(rsreg_ctx *ctx, const void *d_points, size_t n, size_t stride,
                                int is_dense, double max_correspondence_distance);

/* icp.setInputSource(cloud) (incremental_icp.hpp:57, icp_edge...hpp:78,108).  The source is put into the
 * engine's order on a stream of its own and joined when the alignment begins: called BEFORE rsreg_icp_set_target
 * (the reference's order) it runs beside the target's index build.  A host buffer is consumed before the call
 * returns (packed into pinned memory; its way over the PCIe link goes on while the caller packs the target); a device
 * buffer must stay alive and unchanged until rsreg_icp_begin / rsreg_icp_align has returned. */
int rsreg_icp_set_source(rsreg_ctx *ctx, const void *points, size_t n, size_t stride, int is_dense);
int rsreg_icp_set_source_device(rsreg_ctx *ctx, const void *d_points, size_t n, size_t stride,
                                int is_dense);

/* icp.align(out) / icp.align(out, guess) + hasConverged() + getFinalTransformation()
 * (incremental_icp.hpp:59-63, icp_edge...hpp:95,104,111-117, ndt_edge...hpp:99-105).
 * guess: 16 floats column-major, NULL = identity.  aligned_out (nullable, host): receives
 * n_source records of `out_stride` bytes: the input records with xyz <- final * xyz
 * (only xyz and, if out_stride >= 16, data[3] = 1 are written; copy colour yourself). */
int rsreg_icp_align(rsreg_ctx *ctx, const float *guess, const rsreg_icp_params *params,
                    rsreg_icp_result *result, void *aligned_out, size_t out_stride);
/* The same with PCL's `output = input` done on the way (incremental_icp.hpp:59 `icp.align(*aligned)`: PCL copies the input cloud
 * into the output and then rewrites xyz): aligned_out receives n_source WHOLE records of `stride` bytes -- the records at
 * `source_records` (the host cloud the source was set from, or any records of that layout; may be aligned_out itself) with
 * xyz <- final * xyz and, if stride >= 16, data[3] = 1.  The copy is made by the host threads that write the aligned positions
 * anyway, while those are still on the PCIe link: an adaptor no longer copies 32 bytes a point itself before the call
 * (2.5 -> 1.9 ms per 10^6-point pair). */
int rsreg_icp_align_records(rsreg_ctx *ctx, const float *guess, const rsreg_icp_params *params,
                            rsreg_icp_result *result, const void *source_records, void *aligned_out, size_t stride);

/* Step-wise form of the same loop (parity tests, N-GPU sharding by source blocks):
 *   begin -> { search -> sums -> [all-reduce the 17 sums] -> update } ... -> end     */
int rsreg_icp_begin(rsreg_ctx *ctx, const float *guess, const rsreg_icp_params *params);
/* CorrespondenceEstimation::determineCorrespondences: nearest target per current source
 * point.  Outputs (host, each nullable, n_source entries): index into the ORIGINAL target
 * array (-1: no target within the gate / non-finite source point), squared distance. */
int rsreg_icp_search(rsreg_ctx *ctx, int32_t *index_out, float *sqr_dist_out);
/* The 17 sums over the accepted correspondences of the last search (this rank's block). */
int rsreg_icp_sums(rsreg_ctx *ctx, double sums[RSREG_NUM_SUMS]);
/* TransformationEstimationSVD (Umeyama) from (possibly all-reduced) sums, transform the
 * source in place, compose final = T_inc * final, evaluate the convergence criteria.
 * t_inc_out (nullable): the incremental transform.  *done: 1 when the loop must stop. */
int rsreg_icp_update(rsreg_ctx *ctx, const double sums[RSREG_NUM_SUMS], float *t_inc_out,
                     int *done);
int rsreg_icp_end(rsreg_ctx *ctx, rsreg_icp_result *result, void *aligned_out, size_t out_stride);

/* Host-only pieces of the iteration, exposed for tests and for callers that run the
 * all-reduce themselves. */
int rsreg_umeyama_from_sums(const double sums[RSREG_NUM_SUMS], float t_out[16]);

/* ---- pcl::transformPointCloud(in, out, Matrix4f) ------------------------------------ */
/* incremental_icp.hpp:63, icp_edge...hpp:116-117, ndt_edge...hpp:104-105.  in == out is
 * allowed.  Records are copied whole (stride bytes) and xyz rewritten; when !is_dense,
 * non-finite points are copied unchanged. */
int rsreg_transform_cloud(rsreg_ctx *ctx, const void *in, void *out, size_t n, size_t stride,
                          int is_dense, const float transform[16]);

/* ---- pcl::ApproximateVoxelGrid<PointXYZRGB>::filter --------------------------------- */
/* incremental_icp.hpp:54-55, icp_edge...hpp:47,59-60,75-76, ndt_edge...hpp:45,57-58,68-69.
 * Order-dependent streaming hash-history centroiding; this entry point runs it sequentially
 * on the host (no context needed), record for record like PCL.  Records must be PointXYZRGB (stride >= 20, rgb
 * at byte 16).  out must hold n records; *n_out receives the count.  in == out allowed. */
int rsreg_approx_voxel_grid(const void *in, size_t n, size_t stride, const float leaf[3],
                            void *out, size_t *n_out);
/* The same filter on the GPU, same output record for record: the points of one hash slot are
 * an independent stream, each run of equal voxels in it gives one centroid (float sums in input
 * order), and a run is emitted where the next run of its slot begins -- all of which sorts and
 * scans reconstruct (csrc/voxel.hip).  stride must be a multiple of 4. */
int rsreg_approx_voxel_grid_gpu(rsreg_ctx *ctx, const void *in, size_t n, size_t stride,
                                const float leaf[3], void *out, size_t *n_out);

/* ---- NDT: pcl::NormalDistributionsTransform<PointXYZRGB,PointXYZRGB> ---------------- */
/* ndt.setInputTarget (ndt_edge...hpp:72): voxel binning + per-voxel mean / covariance /
 * regularised inverse covariance (VoxelGridCovariance). */
int rsreg_ndt_set_target(rsreg_ctx *ctx, const void *points, size_t n, size_t stride,
                         int is_dense, double resolution);
/* ndt.setInputSource + ndt.align(out, guess) + getFinalTransformation
 * (ndt_edge...hpp:71,83,92,104). */
int rsreg_ndt_align(rsreg_ctx *ctx, const void *source, size_t n, size_t stride, int is_dense,
                    const float *guess, const rsreg_ndt_params *params, rsreg_ndt_result *result,
                    void *aligned_out, size_t out_stride);
/* One score/gradient/Hessian pass at pose p = [tx ty tz rx ry rz] (tests; the unit an
 * N-GPU run all-reduces; on the wire 1 + 6 + 21 doubles, the Hessian being symmetric). */
int rsreg_ndt_derivatives(rsreg_ctx *ctx, const void *source, size_t n, size_t stride,
                          int is_dense, const double pose[6], double *score, double gradient[6],
                          double hessian[36]);
/* Which point a voxel is searched by (VoxelGridCovariance's centroid cloud, ndt_edge_based_registration.hpp:71-72 ->
 * setInputTarget).  0 (default): the voxel's f64 mean rounded to float.  1: PCL's own arithmetic -- a float running
 * sum over the voxel's points in input order, divided by float(n) (voxel_grid_covariance.hpp: leaf.centroid += pt;
 * leaf.centroid /= nr_points) -- one sequential chain per voxel, for bit-level agreement with a PCL build.  Applies
 * to the targets set afterwards.  Optionally reads the centroids back (3 floats per valid voxel). */
int rsreg_ndt_set_centroid_mode(rsreg_ctx *ctx, int mode);
int rsreg_ndt_get_centroids(rsreg_ctx *ctx, float *centroids /*3 each*/, int32_t capacity);
/* Read back the valid voxels: per voxel 3 (mean) + 9 (cov) + 9 (icov) doubles and a count. */
int rsreg_ndt_get_voxels(rsreg_ctx *ctx, int32_t *n_voxels, double *mean_cov_icov /*21 each*/,
                         int32_t *counts, int32_t capacity);

/* ---- N-GPU: one pair sharded by source-point blocks --------------------------------- */
/* One process per GPU.  Rank 0 calls rsreg_comm_unique_id, the caller ships the 128 bytes
 * to every rank (any side channel, e.g. torch.distributed broadcast over gloo), every rank
 * calls rsreg_comm_init.  After that rsreg_icp_align / rsreg_ndt_align all-reduce their
 * sums (17 / 28 doubles per pass: NDT ships the score, the gradient and the upper triangle
 * of the Hessian) over RCCL on the ctx stream; every rank then runs the
 * same host solve on identical numbers, so no broadcast of the transform is needed. */
#define RSREG_UNIQUE_ID_BYTES 128
int rsreg_comm_unique_id(uint8_t id[RSREG_UNIQUE_ID_BYTES]);
int rsreg_comm_init(rsreg_ctx *ctx, const uint8_t id[RSREG_UNIQUE_ID_BYTES], int rank, int nranks);
int rsreg_comm_destroy(rsreg_ctx *ctx);
/* All-reduce (sum) `count` doubles in place across the ranks of ctx's communicator. */
int rsreg_comm_allreduce_f64(rsreg_ctx *ctx, double *host_buf, int count);

/* ---- device-resident clouds: the frame loop without leaving HBM ------------------------- */
/* The reference's schemes run, per frame, ApproximateVoxelGrid::filter -> align (-> align) ->
 * transformPointCloud x2 -> operator+ on host clouds (incremental_icp.hpp:54-64,
 * icp_edge_based_registration.hpp:75-76,95-120, ndt_edge_based_registration.hpp:68-108).  A
 * rsreg_cloud holds the records of one cloud in HBM (whole records, `stride` bytes each, plus
 * width / height / is_dense); every step below takes and leaves its clouds there, so a frame is
 * uploaded once and the merged cloud downloaded once.  A cloud belongs to the ctx it was created
 * on; handles given to rsreg_icp_set_*_cloud must stay alive and unchanged until the align that
 * uses them has returned. */
typedef struct rsreg_cloud rsreg_cloud;
int rsreg_cloud_create(rsreg_ctx *ctx, rsreg_cloud **out);
int rsreg_cloud_destroy(rsreg_cloud *cloud);
int rsreg_cloud_upload(rsreg_cloud *cloud, const void *points, size_t n, size_t stride, uint32_t width,
                       uint32_t height, int is_dense);
/* The same, returning as soon as the records are staged: the PCIe copy runs on a copy stream of the context beside the
 * work of the main stream (a frame loop uploads frame k + 1 while frame k is being aligned: incremental_icp.hpp:51-66
 * hands over all frames up front).  Every call that reads or rewrites the cloud waits for the copy first; `points` may
 * be reused when the call returns. */
int rsreg_cloud_upload_async(rsreg_cloud *cloud, const void *points, size_t n, size_t stride, uint32_t width,
                             uint32_t height, int is_dense);
/* rsreg_cloud_upload_async that returns before `points` has been read: the records are staged and their copy queued by a
 * thread of the context.  `points` must stay valid and unchanged until a call that reads or rewrites the cloud (any of
 * them waits for the upload) has returned.  For callers whose frames stay put for the whole registration -- the
 * reference's schemes take the caller's vector of clouds (types.hpp:19) and read frame k + 2 while frame k is aligned. */
int rsreg_cloud_upload_deferred(rsreg_cloud *cloud, const void *points, size_t n, size_t stride, uint32_t width,
                                uint32_t height, int is_dense);
int rsreg_cloud_download(const rsreg_cloud *cloud, void *out, size_t capacity_records);
/* rsreg_cloud_download that returns at once: the records as they are when the context's stream gets here go to `out`
 * (capacity in records) on a download stream and the copy-out threads of the context; the cloud may be rewritten or
 * destroyed right away.  `out` must stay valid and untouched until rsreg_ctx_wait_downloads(ctx) has returned.  The frame
 * loops hand every frame's moved points to the host this way while the next frames are aligned (the merged cloud the
 * schemes return, types.hpp:19, is then complete when the loop ends: incremental_icp.hpp:63-64, icp_edge...hpp:116-120). */
int rsreg_cloud_download_async(const rsreg_cloud *cloud, void *out, size_t capacity_records);
int rsreg_ctx_wait_downloads(rsreg_ctx *ctx);
int rsreg_cloud_info(const rsreg_cloud *cloud, size_t *n, size_t *stride, uint32_t *width, uint32_t *height,
                     int *is_dense);
const void *rsreg_cloud_device_ptr(const rsreg_cloud *cloud);
/* Which cloud this is (`id`, unique per handle) and how often its records have been rewritten (`version`: every upload,
 * filter, transform, concatenation or alignment INTO the handle counts).  A host layer that keeps PCL's
 * "setInputSource once, align many times" habit compares the pair with what it loaded last and loads again when the
 * cloud has changed in place in between (pcl_compat.hpp; PCL itself would see the new points through its pointer:
 * incremental_icp.hpp:57-59 sets both inputs before every align anyway). */
int rsreg_cloud_version(const rsreg_cloud *cloud, uint64_t *id, uint64_t *version);
int rsreg_cloud_copy(rsreg_ctx *ctx, const rsreg_cloud *in, rsreg_cloud *out);
/* ApproximateVoxelGrid::filter, same records in the same order as the host filter; in == out allowed */
int rsreg_cloud_filter(rsreg_ctx *ctx, const rsreg_cloud *in, const float leaf[3], rsreg_cloud *out);
/* The same queued by a thread of the context on a stream and scratch of its own: returns at once; the number of output
 * records is known, and the records are there, when a call that takes `out` has waited for them (every one does).  The
 * frame loops filter the next frames this way while they align this one (incremental_icp.hpp:54-55 filters every frame
 * independently of the registration).  `in` must stay alive and unchanged until `out` has been used; in != out.  `in`
 * may be the output of rsreg_cloud_edge_features_async that has not run yet: the jobs run in the order of the calls. */
int rsreg_cloud_filter_async(rsreg_ctx *ctx, const rsreg_cloud *in, const float leaf[3], rsreg_cloud *out);
/* pcl::transformPointCloud; in == out allowed */
int rsreg_cloud_transform(rsreg_ctx *ctx, const rsreg_cloud *in, const float transform[16], rsreg_cloud *out);
/* PointCloud::operator+ : out = a followed by b (width = size, height = 1, is_dense = both); out may be a or b */
int rsreg_cloud_concat(rsreg_ctx *ctx, const rsreg_cloud *a, const rsreg_cloud *b, rsreg_cloud *out);
/* icp.setInputTarget / setInputSource / align on handles; aligned_out (nullable, may be the source
 * cloud): the source records with xyz <- final * xyz and data[3] = 1 */
int rsreg_icp_set_target_cloud(rsreg_ctx *ctx, const rsreg_cloud *cloud, double max_correspondence_distance);
int rsreg_icp_set_source_cloud(rsreg_ctx *ctx, const rsreg_cloud *cloud);
/* 1 when the ICP target index of `ctx` was built by rsreg_icp_set_target_cloud from this cloud, whose records have not
 * been rewritten since, for this gate; 0 otherwise.  The ICP edge scheme sets the same grown feature cloud as the target
 * of its coarse and of its refining ICP, one after the other (icp_edge_based_registration.hpp:94-95,108-109): a caller
 * that asks first may skip the second build (pcl_compat.hpp: setReuseTargetIndex). */
int rsreg_icp_target_is_cloud(const rsreg_ctx *ctx, const rsreg_cloud *cloud, double max_correspondence_distance);
int rsreg_icp_align_cloud(rsreg_ctx *ctx, const float *guess, const rsreg_icp_params *params,
                          rsreg_icp_result *result, rsreg_cloud *aligned_out);
/* ndt.setInputTarget / align on handles, and on raw device pointers */
int rsreg_ndt_set_target_cloud(rsreg_ctx *ctx, const rsreg_cloud *cloud, double resolution);
int rsreg_ndt_align_cloud(rsreg_ctx *ctx, const rsreg_cloud *source, const float *guess,
                          const rsreg_ndt_params *params, rsreg_ndt_result *result, rsreg_cloud *aligned_out);
int rsreg_ndt_set_target_device(rsreg_ctx *ctx, const void *d_points, size_t n, size_t stride, int is_dense,
                                double resolution);
int rsreg_ndt_align_device(rsreg_ctx *ctx, const void *d_source, size_t n, size_t stride, int is_dense,
                           const float *guess, const rsreg_ndt_params *params, rsreg_ndt_result *result,
                           void *d_aligned_out);

/* ---- edge features: extract_edge_features (src/edge_extractor.hpp:7-39) -------------------- */
/* The reference's TwoPhaseRegistrationScheme::extract_features (icp_edge...hpp:21-23, ndt_edge...hpp:18-20).
 * Of everything that function computes it returns only the points labelled EDGELABEL_RGB_CANNY
 * (label_indices[4]): pcl::Edge::detectEdgeCanny (thresholds 40 / 100) on the gray image
 * float((r + g + b) / 3) of the ORGANIZED cloud (width x height records, rgb at byte 16).  out must
 * hold width*height records; indices_out (nullable) receives the edge points' indices, ascending. */
int rsreg_extract_edge_features(rsreg_ctx *ctx, const void *points, uint32_t width, uint32_t height, size_t stride,
                                void *out, int32_t *indices_out, size_t *n_out);
int rsreg_cloud_edge_features(rsreg_ctx *ctx, const rsreg_cloud *in, rsreg_cloud *out);
/* rsreg_cloud_edge_features queued like rsreg_cloud_filter_async: the edge schemes extract (and then filter) the features
 * of frame k + 1 beside the two alignments of frame k -- the reference extracts the features of all frames before it
 * registers any (types.hpp:30-43).  `in` may still be uploading (rsreg_cloud_upload_deferred): the job waits for it, not
 * the caller.  Same lifetime rule as above. */
int rsreg_cloud_edge_features_async(rsreg_ctx *ctx, const rsreg_cloud *in, rsreg_cloud *out);

/* ---- PCD files: the LZF coder of "DATA binary_compressed" bodies (host only, no ctx) ------ */
/* pcl::io::loadPCDFile / savePCDFileBinaryCompressed as reached from main.cpp:53,81,87: the body
 * is u32 compressed size, u32 uncompressed size, then one LZF stream over the fields laid out one
 * after the other.  encode/decode return the number of bytes written, 0 on failure (capacity
 * too small, malformed stream). */
size_t rsreg_lzf_max_encoded_size(size_t n);
size_t rsreg_lzf_encode(const void *in, size_t n, void *out, size_t capacity);
size_t rsreg_lzf_decode(const void *in, size_t n, void *out, size_t capacity);

/* ---- introspection (tests, bench) --------------------------------------------------- */
typedef struct rsreg_grid_info {
    float origin[3];
    float cell_size;
    int32_t dims[3];
    uint32_t n_target_points;   /* finite points handed in                                  */
    uint32_t n_unique_points;   /* after dropping exact duplicates (same xyz bits)          */
    uint32_t n_cells;           /* occupied cells                                           */
    uint32_t max_points_per_cell;
    double ms_build;            /* device time of the last build (profiling on)             */
    uint32_t index_kind;        /* 1 = dense cell-start table, 0 = brick hash (huge extents), 2 = none: a device-cloud
                                 * target set for at most 64 source points is searched whole (IncrementalICP) */
    uint32_t n_source_distinct; /* distinct source points the iterations work on (0: no source) */
    uint64_t index_bytes;       /* HBM bytes of the index: sorted points + tables           */
} rsreg_grid_info;
/* (waits for an index build that rsreg_icp_set_target* has queued and not waited for: the two counts come with it) */
int rsreg_icp_grid_info(rsreg_ctx *ctx, rsreg_grid_info *info);

/* Where the HOST-pointer entry points (rsreg_icp_set_source, rsreg_icp_set_target, rsreg_icp_align with aligned_out: the
 * literal call surface of incremental_icp.hpp:57-63, clouds in host memory in, 4x4 and aligned cloud out) spent the host's
 * wall clock in their last call, ms.  *_stage_wait: waiting for the staging buffer's previous trip over the link;
 * *_pack: packing the caller's records into pinned memory and queueing their copies (the copies run meanwhile);
 * target_build: the index build, behind the target's copy (ends with the host's wait for the build's counts);
 * align: rsreg_icp_begin .. the last iteration; aligned_copy: the aligned cloud's way home, device -> pinned -> caller. */
typedef struct rsreg_host_timing {
    double source_stage_wait, source_pack, target_stage_wait, target_pack, target_build, align, aligned_copy;
    double loop_enqueue;   /* device-resident loop: what queueing all its launches took the calling thread (0.4: was `reserved`) */
} rsreg_host_timing;
int rsreg_ctx_host_timing(rsreg_ctx *ctx, rsreg_host_timing *out);

#ifdef __cplusplus
}
#endif
#endif /* RSREG_H_ */
