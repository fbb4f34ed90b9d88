/*
 * ndt_oracle.c — CPU restatement of pcl::NormalDistributionsTransform as the reference
 * drives it (src/ndt_edge_based_registration.hpp:38-43,71-72,83,92,104).
 * TEST INFRASTRUCTURE ONLY (see rsreg_oracle.h: who may use oracle/, "PARITY UNPINNED").
 *
 * Follows PCL 1.9.1 (restated from SURVEY.md Appendix A.6 / A.7):
 *   filters/impl/voxel_grid_covariance.hpp  applyFilter, radiusSearch
 *   registration/impl/ndt.hpp               computeTransformation, computeDerivatives,
 *                                           computeAngleDerivatives, computePointDerivatives,
 *                                           updateDerivatives, computeHessian, updateHessian,
 *                                           computeStepLengthMT, trialValueSelectionMT,
 *                                           updateIntervalMT   (Magnusson 2009; More-Thuente 1994)
 * Neighbour voxels are found by a scan over all voxel centroids (PCL: kd-tree radius
 * search over the same centroids, same f32 L2_Simple distance, same strict d2 < r2 test).
 */
#include "rsreg_oracle.h"
#include "orc_linalg.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

static double now_sec(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

static inline int finite3(const float *p) { return isfinite(p[0]) && isfinite(p[1]) && isfinite(p[2]); }

typedef struct ndt_leaf {
    long long key;
    int n;
    float csum[3];      /* f32 running centroid sum (PCL leaf.centroid) */
    float centroid[3];
    double sum[3];      /* leaf.mean_ accumulator */
    double sxx[9];      /* leaf.cov_ accumulator */
    double mean[3], cov[9], icov[9];
} ndt_leaf;

struct orc_ndt {
    ndt_leaf *leaves; /* leaves with n >= min_points_per_voxel, key-ascending */
    int n_leaves;
    double resolution;
    /* per-call constants */
    double d1, d2;
    double jang[8][3], hang[15][3];
    int passes;
    int centroid_mode; /* 0: PCL's f32 running sum / n; 1: the f64 mean rounded to f32 (what the HIP path stores) */
};

void orc_ndt_params_default(orc_ndt_params *p)
{
    memset(p, 0, sizeof(*p));
    p->max_iterations = 35;
    p->transformation_epsilon = 0.1;
    p->step_size = 0.1;
    p->resolution = 1.0;
    p->outlier_ratio = 0.55;
}

/* src/ndt_edge_based_registration.hpp:38-43 */
void orc_ndt_params_reference(orc_ndt_params *p)
{
    orc_ndt_params_default(p);
    p->transformation_epsilon = 0.01;
    p->step_size = 0.1;
    p->resolution = 1.0;
    p->max_iterations = 50;
}

orc_ndt *orc_ndt_create(void) { return (orc_ndt *)calloc(1, sizeof(orc_ndt)); }
void orc_ndt_set_centroid_mode(orc_ndt *o, int mode) { o->centroid_mode = mode; }
void orc_ndt_destroy(orc_ndt *o)
{
    if (!o) return;
    free(o->leaves);
    free(o);
}

typedef struct key_idx { long long key; int idx; } key_idx;
static int key_cmp(const void *a, const void *b)
{
    const key_idx *x = (const key_idx *)a, *y = (const key_idx *)b;
    if (x->key != y->key) return x->key < y->key ? -1 : 1;
    return x->idx - y->idx;
}

/* A.6: VoxelGridCovariance::applyFilter with leaf = resolution, min_points_per_voxel 6,
 * min_covar_eigvalue_mult 0.01 */
int orc_ndt_set_target(orc_ndt *o, const void *pts, size_t n, size_t stride, int is_dense,
                       double resolution)
{
    (void)is_dense;
    free(o->leaves);
    o->leaves = NULL;
    o->n_leaves = 0;
    o->resolution = resolution;
    const char *base = (const char *)pts;
    float leaf = (float)resolution;
    float inv_leaf = 1.0f / leaf;

    /* getMinMax3D over finite points */
    float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    size_t nfin = 0;
    for (size_t i = 0; i < n; i++) {
        const float *p = (const float *)(base + i * stride);
        if (!finite3(p)) continue;
        nfin++;
        for (int d = 0; d < 3; d++) { if (p[d] < mn[d]) mn[d] = p[d]; if (p[d] > mx[d]) mx[d] = p[d]; }
    }
    if (nfin == 0) return 0;
    int min_b[3], max_b[3], div_b[3];
    for (int d = 0; d < 3; d++) {
        min_b[d] = (int)floorf(mn[d] * inv_leaf);
        max_b[d] = (int)floorf(mx[d] * inv_leaf);
        div_b[d] = max_b[d] - min_b[d] + 1;
    }
    long long mul[3] = {1, div_b[0], (long long)div_b[0] * div_b[1]};

    key_idx *ki = (key_idx *)malloc(sizeof(key_idx) * nfin);
    size_t m = 0;
    for (size_t i = 0; i < n; i++) {
        const float *p = (const float *)(base + i * stride);
        if (!finite3(p)) continue;
        long long key = 0;
        for (int d = 0; d < 3; d++) {
            int ijk = (int)(floorf(p[d] * inv_leaf) - (float)min_b[d]);
            key += ijk * mul[d];
        }
        ki[m].key = key;
        ki[m].idx = (int)i;
        m++;
    }
    qsort(ki, m, sizeof(key_idx), key_cmp); /* key asc (std::map order), then point order */

    ndt_leaf *L = (ndt_leaf *)calloc(m, sizeof(ndt_leaf));
    int nl = 0;
    size_t a = 0;
    while (a < m) {
        size_t b = a;
        while (b < m && ki[b].key == ki[a].key) b++;
        int cnt = (int)(b - a);
        if (cnt >= 6) {
            ndt_leaf *lf = &L[nl];
            memset(lf, 0, sizeof(*lf));
            lf->key = ki[a].key;
            lf->n = cnt;
            for (size_t k = a; k < b; k++) {
                const float *p = (const float *)(base + (size_t)ki[k].idx * stride);
                for (int d = 0; d < 3; d++) { lf->csum[d] += p[d]; lf->sum[d] += (double)p[d]; }
                for (int r = 0; r < 3; r++)
                    for (int c = 0; c < 3; c++) lf->sxx[r * 3 + c] += (double)p[r] * (double)p[c];
            }
            double nn = (double)cnt;
            for (int d = 0; d < 3; d++) {
                lf->mean[d] = lf->sum[d] / nn;
                lf->centroid[d] = o->centroid_mode ? (float)lf->mean[d] : lf->csum[d] / (float)cnt;
            }
            for (int r = 0; r < 3; r++)
                for (int c = 0; c < 3; c++)
                    lf->cov[r * 3 + c] = (lf->sxx[r * 3 + c] - 2.0 * (lf->sum[r] * lf->mean[c])) / nn +
                                         lf->mean[r] * lf->mean[c];
            for (int k = 0; k < 9; k++) lf->cov[k] *= (nn - 1.0) / nn;
            /* SelfAdjointEigenSolver reads the lower triangle */
            double sym[9];
            for (int r = 0; r < 3; r++)
                for (int c = 0; c < 3; c++) sym[r * 3 + c] = r >= c ? lf->cov[r * 3 + c] : lf->cov[c * 3 + r];
            double ev[3], evec[9];
            orc_eig_sym3(sym, ev, evec);
            int ok = !(ev[0] < 0 || ev[1] < 0 || ev[2] <= 0);
            if (ok) {
                double min_ev = 0.01 * ev[2];
                if (ev[0] < min_ev) {
                    ev[0] = min_ev;
                    if (ev[1] < min_ev) ev[1] = min_ev;
                    /* cov = V diag(ev) V^-1 (V orthogonal) */
                    for (int r = 0; r < 3; r++)
                        for (int c = 0; c < 3; c++) {
                            double v = 0;
                            for (int k = 0; k < 3; k++) v += evec[r * 3 + k] * ev[k] * evec[c * 3 + k];
                            lf->cov[r * 3 + c] = v;
                        }
                }
                if (!orc_inv3(lf->cov, lf->icov)) memset(lf->icov, 0, sizeof(lf->icov));
            } else {
                /* PCL: leaf flagged nr_points = -1 but its centroid stays searchable with a
                 * zero-initialised icov_ */
                memset(lf->icov, 0, sizeof(lf->icov));
            }
            nl++;
        }
        a = b;
    }
    o->leaves = (ndt_leaf *)realloc(L, sizeof(ndt_leaf) * (nl ? nl : 1));
    o->n_leaves = nl;
    free(ki);
    return 0;
}

int orc_ndt_get_voxels(orc_ndt *o, int32_t *n_voxels, double *mci, int32_t *counts, int32_t cap)
{
    *n_voxels = o->n_leaves;
    for (int i = 0; i < o->n_leaves && i < cap; i++) {
        if (mci) {
            memcpy(mci + 21 * i, o->leaves[i].mean, 24);
            memcpy(mci + 21 * i + 3, o->leaves[i].cov, 72);
            memcpy(mci + 21 * i + 12, o->leaves[i].icov, 72);
        }
        if (counts) counts[i] = o->leaves[i].n;
    }
    return 0;
}

/* the search centroids of the valid leaves, 3 floats each (what the radius search of computeDerivatives runs on) */
int orc_ndt_get_centroids(orc_ndt *o, float *centroids, int32_t cap)
{
    for (int i = 0; i < o->n_leaves && i < cap; i++) memcpy(centroids + 3 * i, o->leaves[i].centroid, 12);
    return 0;
}

static void gauss_constants(orc_ndt *o, const orc_ndt_params *prm)
{
    double c1 = 10.0 * (1 - prm->outlier_ratio);
    double c2 = prm->outlier_ratio / pow(prm->resolution, 3);
    double d3 = -log(c2);
    o->d1 = -log(c1 + c2) - d3;
    o->d2 = -2 * log((-log(c1 * exp(-0.5) + c2) - d3) / o->d1);
}

static void angle_derivatives(orc_ndt *o, const double *p)
{
    double cx, cy, cz, sx, sy, sz;
    if (fabs(p[3]) < 10e-5) { cx = 1.0; sx = 0.0; } else { cx = cos(p[3]); sx = sin(p[3]); }
    if (fabs(p[4]) < 10e-5) { cy = 1.0; sy = 0.0; } else { cy = cos(p[4]); sy = sin(p[4]); }
    if (fabs(p[5]) < 10e-5) { cz = 1.0; sz = 0.0; } else { cz = cos(p[5]); sz = sin(p[5]); }
    double (*j)[3] = o->jang;
    double (*h)[3] = o->hang;
#define SET3(v, a, b, c) do { (v)[0] = (a); (v)[1] = (b); (v)[2] = (c); } while (0)
    SET3(j[0], (-sx * sz + cx * sy * cz), (-sx * cz - cx * sy * sz), (-cx * cy));
    SET3(j[1], (cx * sz + sx * sy * cz), (cx * cz - sx * sy * sz), (-sx * cy));
    SET3(j[2], (-sy * cz), sy * sz, cy);
    SET3(j[3], sx * cy * cz, (-sx * cy * sz), sx * sy);
    SET3(j[4], (-cx * cy * cz), cx * cy * sz, (-cx * sy));
    SET3(j[5], (-cy * sz), (-cy * cz), 0);
    SET3(j[6], (cx * cz - sx * sy * sz), (-cx * sz - sx * sy * cz), 0);
    SET3(j[7], (sx * cz + cx * sy * sz), (cx * sy * cz - sx * sz), 0);
    SET3(h[0], (-cx * sz - sx * sy * cz), (-cx * cz + sx * sy * sz), sx * cy);      /* a2 */
    SET3(h[1], (-sx * sz + cx * sy * cz), (-cx * sy * sz - sx * cz), (-cx * cy));   /* a3 */
    SET3(h[2], (cx * cy * cz), (-cx * cy * sz), (cx * sy));                         /* b2 */
    SET3(h[3], (sx * cy * cz), (-sx * cy * sz), (sx * sy));                         /* b3 */
    SET3(h[4], (-sx * cz - cx * sy * sz), (sx * sz - cx * sy * cz), 0);             /* c2 */
    SET3(h[5], (cx * cz - sx * sy * sz), (-sx * sy * cz - cx * sz), 0);             /* c3 */
    SET3(h[6], (-cy * cz), (cy * sz), (sy));                                        /* d1 */
    SET3(h[7], (-sx * sy * cz), (sx * sy * sz), (sx * cy));                         /* d2 */
    SET3(h[8], (cx * sy * cz), (-cx * sy * sz), (-cx * cy));                        /* d3 */
    SET3(h[9], (sy * sz), (sy * cz), 0);                                            /* e1 */
    SET3(h[10], (-sx * cy * sz), (-sx * cy * cz), 0);                               /* e2 */
    SET3(h[11], (cx * cy * sz), (cx * cy * cz), 0);                                 /* e3 */
    SET3(h[12], (-cy * cz), (cy * sz), 0);                                          /* f1 */
    SET3(h[13], (-cx * sz - sx * sy * cz), (-cx * cz + sx * sy * sz), 0);           /* f2 */
    SET3(h[14], (-sx * sz + cx * sy * cz), (-cx * sy * sz - sx * cz), 0);           /* f3 */
#undef SET3
}

static inline double dot3(const double *a, const double *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

/* one pass over the cloud: mode 0 = score+gradient+hessian, 1 = score+gradient, 2 = hessian only */
static double derivative_pass(orc_ndt *o, const float *src, const float *trans, int n,
                              const double *p, int mode, double *grad, double *hess)
{
    angle_derivatives(o, p);
    o->passes++;
    if (mode != 2) memset(grad, 0, 6 * sizeof(double));
    memset(hess, 0, 36 * sizeof(double));
    double score = 0;
    const float r2 = (float)(o->resolution * o->resolution);
    const double d1 = o->d1, d2 = o->d2;
    for (int idx = 0; idx < n; idx++) {
        const float *xt = trans + 3 * idx;
        if (!finite3(xt)) continue;
        double x[3] = {src[3 * idx], src[3 * idx + 1], src[3 * idx + 2]};
        /* point gradient (3x6) and hessian (18x6) */
        double pg[3][6];
        memset(pg, 0, sizeof(pg));
        pg[0][0] = pg[1][1] = pg[2][2] = 1.0;
        pg[1][3] = dot3(x, o->jang[0]); pg[2][3] = dot3(x, o->jang[1]);
        pg[0][4] = dot3(x, o->jang[2]); pg[1][4] = dot3(x, o->jang[3]); pg[2][4] = dot3(x, o->jang[4]);
        pg[0][5] = dot3(x, o->jang[5]); pg[1][5] = dot3(x, o->jang[6]); pg[2][5] = dot3(x, o->jang[7]);
        double ph[6][6][3];
        if (mode != 1) {
            memset(ph, 0, sizeof(ph));
            double a[3] = {0, dot3(x, o->hang[0]), dot3(x, o->hang[1])};
            double b[3] = {0, dot3(x, o->hang[2]), dot3(x, o->hang[3])};
            double c[3] = {0, dot3(x, o->hang[4]), dot3(x, o->hang[5])};
            double d[3] = {dot3(x, o->hang[6]), dot3(x, o->hang[7]), dot3(x, o->hang[8])};
            double e[3] = {dot3(x, o->hang[9]), dot3(x, o->hang[10]), dot3(x, o->hang[11])};
            double f[3] = {dot3(x, o->hang[12]), dot3(x, o->hang[13]), dot3(x, o->hang[14])};
            memcpy(ph[3][3], a, 24); memcpy(ph[4][3], b, 24); memcpy(ph[5][3], c, 24);
            memcpy(ph[3][4], b, 24); memcpy(ph[4][4], d, 24); memcpy(ph[5][4], e, 24);
            memcpy(ph[3][5], c, 24); memcpy(ph[4][5], e, 24); memcpy(ph[5][5], f, 24);
        }
        for (int v = 0; v < o->n_leaves; v++) {
            const ndt_leaf *lf = &o->leaves[v];
            float dx = xt[0] - lf->centroid[0], dy = xt[1] - lf->centroid[1], dz = xt[2] - lf->centroid[2];
            float dd = 0.0f;
            dd += dx * dx; dd += dy * dy; dd += dz * dz;
            if (!(dd < r2)) continue;
            double xm[3] = {(double)xt[0] - lf->mean[0], (double)xt[1] - lf->mean[1], (double)xt[2] - lf->mean[2]};
            const double *ci = lf->icov;
            double cx[3] = {ci[0] * xm[0] + ci[1] * xm[1] + ci[2] * xm[2],
                            ci[3] * xm[0] + ci[4] * xm[1] + ci[5] * xm[2],
                            ci[6] * xm[0] + ci[7] * xm[1] + ci[8] * xm[2]};
            double e_x = exp(-d2 * dot3(xm, cx) / 2);
            double score_inc = -d1 * e_x;
            e_x = d2 * e_x;
            if (e_x > 1 || e_x < 0 || e_x != e_x) continue;
            if (mode != 2) score += score_inc;
            e_x *= d1;
            double cg[6][3], xcg[6];
            for (int i = 0; i < 6; i++) {
                for (int r = 0; r < 3; r++)
                    cg[i][r] = ci[r * 3] * pg[0][i] + ci[r * 3 + 1] * pg[1][i] + ci[r * 3 + 2] * pg[2][i];
                xcg[i] = dot3(xm, cg[i]);
            }
            for (int i = 0; i < 6; i++) {
                if (mode != 2) grad[i] += xcg[i] * e_x;
                if (mode == 1) continue;
                for (int j = 0; j < 6; j++) {
                    double cph[3];
                    for (int r = 0; r < 3; r++)
                        cph[r] = ci[r * 3] * ph[i][j][0] + ci[r * 3 + 1] * ph[i][j][1] + ci[r * 3 + 2] * ph[i][j][2];
                    double gj_cgi = pg[0][j] * cg[i][0] + pg[1][j] * cg[i][1] + pg[2][j] * cg[i][2];
                    hess[i * 6 + j] += e_x * (-d2 * xcg[i] * xcg[j] + dot3(xm, cph) + gj_cgi);
                }
            }
        }
    }
    return score;
}

/* Translation(p0..2) * Rx(p3) * Ry(p4) * Rz(p5), built in f32 like PCL's Eigen::Transform<float> */
static void pose_to_matrix(const double *p, float *M)
{
    float ax = (float)p[3], ay = (float)p[4], az = (float)p[5];
    float cx = cosf(ax), sx = sinf(ax), cy = cosf(ay), sy = sinf(ay), cz = cosf(az), sz = sinf(az);
    float Rx[16] = {1, 0, 0, 0, 0, cx, sx, 0, 0, -sx, cx, 0, 0, 0, 0, 1};
    float Ry[16] = {cy, 0, -sy, 0, 0, 1, 0, 0, sy, 0, cy, 0, 0, 0, 0, 1};
    float Rz[16] = {cz, sz, 0, 0, -sz, cz, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    float T[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, (float)p[0], (float)p[1], (float)p[2], 1};
    float A[16], B[16];
    orc_mat4_mul(T, Rx, A);
    orc_mat4_mul(A, Ry, B);
    orc_mat4_mul(B, Rz, M);
}

static void transform_src(const float *M, const float *src, float *out, int n)
{
    for (int i = 0; i < n; i++) {
        const float *p = src + 3 * i;
        float *q = out + 3 * i;
        if (!finite3(p)) { memcpy(q, p, 12); continue; }
        float x = p[0], y = p[1], z = p[2];
        float ox = M[0] * x; ox = ox + M[4] * y; ox = ox + M[8] * z; ox = ox + M[12];
        float oy = M[1] * x; oy = oy + M[5] * y; oy = oy + M[9] * z; oy = oy + M[13];
        float oz = M[2] * x; oz = oz + M[6] * y; oz = oz + M[10] * z; oz = oz + M[14];
        q[0] = ox; q[1] = oy; q[2] = oz;
    }
}

/* Eigen 3.3 Matrix3f::eulerAngles(0,1,2) on the rotation block of column-major M */
static void euler_xyz(const float *M, float *res)
{
#define R(i, j) M[(j) * 4 + (i)]
    const float PI_F = 3.14159265358979323846f;
    res[0] = atan2f(R(1, 2), R(2, 2));
    float c2 = sqrtf(R(0, 0) * R(0, 0) + R(0, 1) * R(0, 1));
    if (res[0] > 0.0f) {
        res[0] -= PI_F;
        res[1] = atan2f(-R(0, 2), -c2);
    } else {
        res[1] = atan2f(-R(0, 2), c2);
    }
    float s1 = sinf(res[0]), c1 = cosf(res[0]);
    res[2] = atan2f(s1 * R(2, 0) - c1 * R(1, 0), c1 * R(1, 1) - s1 * R(2, 1));
    res[0] = -res[0]; res[1] = -res[1]; res[2] = -res[2];
#undef R
}

static float *copy_xyz(const void *pts, size_t n, size_t stride)
{
    float *x = (float *)malloc(12 * (n ? n : 1));
    for (size_t i = 0; i < n; i++) memcpy(x + 3 * i, (const char *)pts + i * stride, 12);
    return x;
}

int orc_ndt_derivatives(orc_ndt *o, const void *src, size_t n, size_t stride, int is_dense,
                        const double *pose, const orc_ndt_params *prm, double *score,
                        double *grad, double *hess)
{
    (void)is_dense;
    gauss_constants(o, prm);
    float *s = copy_xyz(src, n, stride), *t = (float *)malloc(12 * (n ? n : 1));
    float M[16];
    pose_to_matrix(pose, M);
    transform_src(M, s, t, (int)n);
    *score = derivative_pass(o, s, t, (int)n, pose, 0, grad, hess);
    free(s); free(t);
    return 0;
}

/* More-Thuente helpers (ndt.hpp) */
/* std::min(a, b) / std::max(a, b) as libstdc++ and libc++ define them (the FIRST operand unless the comparison says otherwise):
 * fmin / fmax return the non-NaN operand, std::min(a, NaN) returns a and std::min(NaN, b) returns NaN -- what PCL's
 * ndt.hpp (trialValueSelectionMT, computeStepLengthMT) does with a 0/0 trial value. */
static double std_min(double a, double b) { return (b < a) ? b : a; }
static double std_max(double a, double b) { return (a < b) ? b : a; }
static double psi_mt(double a, double f_a, double f_0, double g_0, double mu) { return f_a - f_0 - mu * g_0 * a; }
static double dpsi_mt(double g_a, double g_0, double mu) { return g_a - mu * g_0; }

static int update_interval_mt(double *a_l, double *f_l, double *g_l, double *a_u, double *f_u,
                              double *g_u, double a_t, double f_t, double g_t)
{
    if (f_t > *f_l) { *a_u = a_t; *f_u = f_t; *g_u = g_t; return 0; }
    else if (g_t * (*a_l - a_t) > 0) { *a_l = a_t; *f_l = f_t; *g_l = g_t; return 0; }
    else if (g_t * (*a_l - a_t) < 0) {
        *a_u = *a_l; *f_u = *f_l; *g_u = *g_l;
        *a_l = a_t; *f_l = f_t; *g_l = g_t;
        return 0;
    }
    return 1;
}

static double trial_value_mt(double a_l, double f_l, double g_l, double a_u, double f_u, double g_u,
                             double a_t, double f_t, double g_t)
{
    if (f_t > f_l) {
        double z = 3 * (f_t - f_l) / (a_t - a_l) - g_t - g_l;
        double w = sqrt(z * z - g_t * g_l);
        double a_c = a_l + (a_t - a_l) * (w - g_l - z) / (g_t - g_l + 2 * w);
        double a_q = a_l - 0.5 * (a_l - a_t) * g_l / (g_l - (f_l - f_t) / (a_l - a_t));
        if (fabs(a_c - a_l) < fabs(a_q - a_l)) return a_c;
        return 0.5 * (a_q + a_c);
    } else if (g_t * g_l < 0) {
        double z = 3 * (f_t - f_l) / (a_t - a_l) - g_t - g_l;
        double w = sqrt(z * z - g_t * g_l);
        double a_c = a_l + (a_t - a_l) * (w - g_l - z) / (g_t - g_l + 2 * w);
        double a_s = a_l - (a_l - a_t) / (g_l - g_t) * g_l;
        if (fabs(a_c - a_t) >= fabs(a_s - a_t)) return a_c;
        return a_s;
    } else if (fabs(g_t) <= fabs(g_l)) {
        double z = 3 * (f_t - f_l) / (a_t - a_l) - g_t - g_l;
        double w = sqrt(z * z - g_t * g_l);
        double a_c = a_l + (a_t - a_l) * (w - g_l - z) / (g_t - g_l + 2 * w);
        double a_s = a_l - (a_l - a_t) / (g_l - g_t) * g_l;
        double a_t_next = fabs(a_c - a_t) < fabs(a_s - a_t) ? a_c : a_s;
        if (a_t > a_l) return std_min(a_t + 0.66 * (a_u - a_t), a_t_next);
        return std_max(a_t + 0.66 * (a_u - a_t), a_t_next);
    } else {
        double z = 3 * (f_t - f_u) / (a_t - a_u) - g_t - g_u;
        double w = sqrt(z * z - g_t * g_u);
        return a_u + (a_t - a_u) * (w - g_u - z) / (g_t - g_u + 2 * w);
    }
}

typedef struct ndt_run {
    orc_ndt *o;
    const float *src;
    float *trans;
    int n;
    float final_t[16];
} ndt_run;

static double step_length_mt(ndt_run *r, const double *x, double *step_dir, double step_init,
                             double step_max, double step_min, double *score, double *grad,
                             double *hess)
{
    double phi_0 = -(*score);
    double d_phi_0 = 0;
    for (int i = 0; i < 6; i++) d_phi_0 -= grad[i] * step_dir[i];
    double x_t[6];
    if (d_phi_0 >= 0) {
        if (d_phi_0 == 0) return 0;
        d_phi_0 *= -1;
        for (int i = 0; i < 6; i++) step_dir[i] *= -1;
    }
    const int max_step_iterations = 10;
    int step_iterations = 0;
    const double mu = 1.e-4, nu = 0.9;
    double a_l = 0, a_u = 0;
    double f_l = psi_mt(a_l, phi_0, phi_0, d_phi_0, mu);
    double g_l = dpsi_mt(d_phi_0, d_phi_0, mu);
    double f_u = psi_mt(a_u, phi_0, phi_0, d_phi_0, mu);
    double g_u = dpsi_mt(d_phi_0, d_phi_0, mu);
    int interval_converged = (step_max - step_min) < 0, open_interval = 1;
    double a_t = step_init;
    a_t = std_min(a_t, step_max);
    a_t = std_max(a_t, step_min);
    for (int i = 0; i < 6; i++) x_t[i] = x[i] + step_dir[i] * a_t;
    pose_to_matrix(x_t, r->final_t);
    transform_src(r->final_t, r->src, r->trans, r->n);
    *score = derivative_pass(r->o, r->src, r->trans, r->n, x_t, 0, grad, hess);
    double phi_t = -(*score);
    double d_phi_t = 0;
    for (int i = 0; i < 6; i++) d_phi_t -= grad[i] * step_dir[i];
    double psi_t = psi_mt(a_t, phi_t, phi_0, d_phi_0, mu);
    double d_psi_t = dpsi_mt(d_phi_t, d_phi_0, mu);
    while (!interval_converged && step_iterations < max_step_iterations &&
           !(psi_t <= 0 && d_phi_t <= -nu * d_phi_0)) {
        if (open_interval) a_t = trial_value_mt(a_l, f_l, g_l, a_u, f_u, g_u, a_t, psi_t, d_psi_t);
        else a_t = trial_value_mt(a_l, f_l, g_l, a_u, f_u, g_u, a_t, phi_t, d_phi_t);
        a_t = std_min(a_t, step_max);
        a_t = std_max(a_t, step_min);
        for (int i = 0; i < 6; i++) x_t[i] = x[i] + step_dir[i] * a_t;
        pose_to_matrix(x_t, r->final_t);
        transform_src(r->final_t, r->src, r->trans, r->n);
        *score = derivative_pass(r->o, r->src, r->trans, r->n, x_t, 1, grad, hess);
        phi_t = -(*score);
        d_phi_t = 0;
        for (int i = 0; i < 6; i++) d_phi_t -= grad[i] * step_dir[i];
        psi_t = psi_mt(a_t, phi_t, phi_0, d_phi_0, mu);
        d_psi_t = dpsi_mt(d_phi_t, d_phi_0, mu);
        if (open_interval && (psi_t <= 0 && d_psi_t >= 0)) {
            open_interval = 0;
            f_l = f_l + phi_0 - mu * d_phi_0 * a_l;
            g_l = g_l + mu * d_phi_0;
            f_u = f_u + phi_0 - mu * d_phi_0 * a_u;
            g_u = g_u + mu * d_phi_0;
        }
        if (open_interval) interval_converged = update_interval_mt(&a_l, &f_l, &g_l, &a_u, &f_u, &g_u, a_t, psi_t, d_psi_t);
        else interval_converged = update_interval_mt(&a_l, &f_l, &g_l, &a_u, &f_u, &g_u, a_t, phi_t, d_phi_t);
        step_iterations++;
    }
    if (step_iterations) derivative_pass(r->o, r->src, r->trans, r->n, x_t, 2, grad, hess);
    return a_t;
}

/* A.7 computeTransformation */
int orc_ndt_align(orc_ndt *o, const void *src, size_t n, size_t stride, int is_dense,
                  const float *guess, const orc_ndt_params *prm, orc_ndt_result *res,
                  void *aligned_out, size_t out_stride)
{
    (void)is_dense;
    double t0 = now_sec();
    gauss_constants(o, prm);
    o->passes = 0;
    ndt_run r;
    r.o = o;
    float *s = copy_xyz(src, n, stride);
    r.src = s;
    r.n = (int)n;
    r.trans = (float *)malloc(12 * (n ? n : 1));
    float I[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    memcpy(r.final_t, I, 64);
    memcpy(r.trans, s, 12 * n);
    if (guess && memcmp(guess, I, 64) != 0) {
        memcpy(r.final_t, guess, 64);
        transform_src(guess, s, r.trans, r.n);
    }
    float er[3];
    euler_xyz(r.final_t, er);
    double p[6] = {r.final_t[12], r.final_t[13], r.final_t[14], er[0], er[1], er[2]};
    double delta_p[6], grad[6], hess[36];
    int nr_iterations = 0, converged = 0;
    double score = derivative_pass(o, s, r.trans, r.n, p, 0, grad, hess);
    while (!converged) {
        double neg_g[6];
        for (int i = 0; i < 6; i++) neg_g[i] = -grad[i];
        orc_svd_solve(hess, 6, neg_g, delta_p);
        double nrm = 0;
        for (int i = 0; i < 6; i++) nrm += delta_p[i] * delta_p[i];
        nrm = sqrt(nrm);
        if (nrm == 0 || nrm != nrm) {
            converged = (nrm == nrm);
            break;
        }
        for (int i = 0; i < 6; i++) delta_p[i] /= nrm;
        nrm = step_length_mt(&r, p, delta_p, nrm, prm->step_size, prm->transformation_epsilon / 2,
                             &score, grad, hess);
        for (int i = 0; i < 6; i++) { delta_p[i] *= nrm; p[i] += delta_p[i]; }
        if (nr_iterations > prm->max_iterations ||
            (nr_iterations && (fabs(nrm) < prm->transformation_epsilon)))
            converged = 1;
        nr_iterations++;
    }
    if (res) {
        memset(res, 0, sizeof(*res));
        memcpy(res->transform, r.final_t, 64);
        res->converged = converged;
        res->iterations = nr_iterations;
        res->score = score;
        res->trans_probability = score / (double)(n ? n : 1);
        res->n_voxels = o->n_leaves;
        res->n_derivative_passes = o->passes;
        res->sec_total = now_sec() - t0;
    }
    if (aligned_out) {
        for (size_t i = 0; i < n; i++) {
            float *dst = (float *)((char *)aligned_out + i * out_stride);
            memcpy(dst, r.trans + 3 * i, 12);
            if (out_stride >= 16) dst[3] = 1.0f;
        }
    }
    free(s);
    free(r.trans);
    return 0;
}
