/*
 * icp_oracle.c — CPU restatement of pcl::IterativeClosestPoint as the reference drives it.
 * TEST INFRASTRUCTURE ONLY (see rsreg_oracle.h: who may use oracle/, and "PARITY UNPINNED").
 *
 * Follows, step by step (PCL 1.9.1, restated from SURVEY.md Appendix A):
 *   A.1  KdTreeFLANN / FLANN KDTreeSingleIndex(leaf 15), L2_Simple<float>, exact k=1 search
 *   A.2  IterativeClosestPoint::computeTransformation (guess, loop, final = T*final)
 *   A.3  Eigen::umeyama without scaling
 *   A.4  DefaultConvergenceCriteria::hasConverged
 *   A.5  ApproximateVoxelGrid::applyFilter
 *   A.7a correspondence gate  !(d2 > max_dist*max_dist)
 *   A.8  transformPointCloud
 * Reference call sites: src/incremental_icp.hpp:46-49,54-64 and the two edge schemes.
 *
 * Build: gcc -O2 -ffp-contract=off -fopenmp (no -ffast-math: float op order is the spec).
 */
#include "rsreg_oracle.h"
#include "orc_linalg.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static double now_sec(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

/* ------------------------------------------------------------------ kd-tree (A.1) */
#define KD_LEAF_MAX 15

typedef struct kd_node {
    int left, right;      /* leaf: [left,right) into vind; inner: child node ids */
    int divfeat;          /* -1 for leaf */
    float divlow, divhigh;
} kd_node;

typedef struct kd_tree {
    const float *pts; /* n x 3 */
    int n;
    int *vind;
    kd_node *nodes;
    int n_nodes, cap_nodes;
    float root_lo[3], root_hi[3];
    float *reordered; /* points in vind order, n x 3 (FLANN reorder_ = true) */
} kd_tree;

static int kd_new_node(kd_tree *t)
{
    if (t->n_nodes == t->cap_nodes) {
        t->cap_nodes = t->cap_nodes ? t->cap_nodes * 2 : 1024;
        t->nodes = (kd_node *)realloc(t->nodes, sizeof(kd_node) * t->cap_nodes);
    }
    return t->n_nodes++;
}

static void kd_minmax(const kd_tree *t, const int *ind, int count, int dim, float *mn, float *mx)
{
    float lo = t->pts[ind[0] * 3 + dim], hi = lo;
    for (int i = 1; i < count; i++) {
        float v = t->pts[ind[i] * 3 + dim];
        if (v < lo) lo = v;
        if (v > hi) hi = v;
    }
    *mn = lo;
    *mx = hi;
}

static void kd_plane_split(const kd_tree *t, int *ind, int count, int cutfeat, float cutval,
                           int *lim1, int *lim2)
{
    int left = 0, right = count - 1;
    for (;;) {
        while (left <= right && t->pts[ind[left] * 3 + cutfeat] < cutval) ++left;
        while (left <= right && t->pts[ind[right] * 3 + cutfeat] >= cutval) --right;
        if (left > right) break;
        int tmp = ind[left]; ind[left] = ind[right]; ind[right] = tmp;
        ++left; --right;
    }
    *lim1 = left;
    right = count - 1;
    for (;;) {
        while (left <= right && t->pts[ind[left] * 3 + cutfeat] <= cutval) ++left;
        while (left <= right && t->pts[ind[right] * 3 + cutfeat] > cutval) --right;
        if (left > right) break;
        int tmp = ind[left]; ind[left] = ind[right]; ind[right] = tmp;
        ++left; --right;
    }
    *lim2 = left;
}

static int kd_divide(kd_tree *t, int left, int right, float *lo, float *hi)
{
    int id = kd_new_node(t);
    int count = right - left;
    if (count <= KD_LEAF_MAX) {
        t->nodes[id].divfeat = -1;
        t->nodes[id].left = left;
        t->nodes[id].right = right;
        for (int d = 0; d < 3; d++) kd_minmax(t, t->vind + left, count, d, &lo[d], &hi[d]);
        return id;
    }
    /* FLANN middleSplit_: widest bbox dimension(s), then widest actual spread */
    int *ind = t->vind + left;
    const float EPS = 0.00001f;
    float max_span = hi[0] - lo[0];
    for (int d = 1; d < 3; d++)
        if (hi[d] - lo[d] > max_span) max_span = hi[d] - lo[d];
    float max_spread = -1;
    int cutfeat = 0;
    for (int d = 0; d < 3; d++) {
        float span = hi[d] - lo[d];
        if (span > (1 - EPS) * max_span) {
            float mn, mx;
            kd_minmax(t, ind, count, d, &mn, &mx);
            if (mx - mn > max_spread) { cutfeat = d; max_spread = mx - mn; }
        }
    }
    float split_val = (lo[cutfeat] + hi[cutfeat]) / 2;
    float mn, mx;
    kd_minmax(t, ind, count, cutfeat, &mn, &mx);
    float cutval = split_val;
    if (split_val < mn) cutval = mn;
    else if (split_val > mx) cutval = mx;
    int lim1, lim2, idx;
    kd_plane_split(t, ind, count, cutfeat, cutval, &lim1, &lim2);
    if (lim1 > count / 2) idx = lim1;
    else if (lim2 < count / 2) idx = lim2;
    else idx = count / 2;

    float llo[3], lhi[3], rlo[3], rhi[3];
    memcpy(llo, lo, sizeof(llo)); memcpy(lhi, hi, sizeof(lhi));
    memcpy(rlo, lo, sizeof(rlo)); memcpy(rhi, hi, sizeof(rhi));
    lhi[cutfeat] = cutval;
    rlo[cutfeat] = cutval;
    int c1 = kd_divide(t, left, left + idx, llo, lhi);
    int c2 = kd_divide(t, left + idx, right, rlo, rhi);
    t->nodes[id].divfeat = cutfeat;
    t->nodes[id].left = c1;
    t->nodes[id].right = c2;
    t->nodes[id].divlow = lhi[cutfeat];
    t->nodes[id].divhigh = rlo[cutfeat];
    for (int d = 0; d < 3; d++) {
        lo[d] = llo[d] < rlo[d] ? llo[d] : rlo[d];
        hi[d] = lhi[d] > rhi[d] ? lhi[d] : rhi[d];
    }
    return id;
}

static kd_tree *kd_build(const float *pts, int n)
{
    kd_tree *t = (kd_tree *)calloc(1, sizeof(kd_tree));
    t->pts = pts;
    t->n = n;
    t->vind = (int *)malloc(sizeof(int) * (n > 0 ? n : 1));
    for (int i = 0; i < n; i++) t->vind[i] = i;
    if (n > 0) {
        for (int d = 0; d < 3; d++) kd_minmax(t, t->vind, n, d, &t->root_lo[d], &t->root_hi[d]);
        float lo[3], hi[3];
        memcpy(lo, t->root_lo, sizeof(lo));
        memcpy(hi, t->root_hi, sizeof(hi));
        kd_divide(t, 0, n, lo, hi);
        memcpy(t->root_lo, lo, sizeof(lo));
        memcpy(t->root_hi, hi, sizeof(hi));
        t->reordered = (float *)malloc(sizeof(float) * 3 * n);
        for (int i = 0; i < n; i++) memcpy(t->reordered + 3 * i, pts + 3 * t->vind[i], 12);
    }
    return t;
}

static void kd_free(kd_tree *t)
{
    if (!t) return;
    free(t->vind); free(t->nodes); free(t->reordered); free(t);
}

/* FLANN L2_Simple<float>: result += diff*diff in x,y,z order, all f32 */
static inline float l2_simple(const float *a, const float *b)
{
    float r = 0.0f;
    float d0 = a[0] - b[0]; r += d0 * d0;
    float d1 = a[1] - b[1]; r += d1 * d1;
    float d2 = a[2] - b[2]; r += d2 * d2;
    return r;
}

typedef struct kd_result { float worst; int index; } kd_result;

static void kd_search_level(const kd_tree *t, kd_result *res, const float *q, int node,
                            float mindistsq, float *dists)
{
    const kd_node *nd = &t->nodes[node];
    if (nd->divfeat < 0) {
        for (int i = nd->left; i < nd->right; i++) {
            float d = l2_simple(q, t->reordered + 3 * i);
            /* FLANN keeps the first point visited among exactly equidistant ones, which
             * depends on its traversal; the canonical rule here (shared with the golden
             * vectors and the HIP path) is: lowest target index wins an exact tie. */
            if (d < res->worst || (d == res->worst && t->vind[i] < res->index)) {
                res->worst = d;
                res->index = t->vind[i];
            }
        }
        return;
    }
    int idx = nd->divfeat;
    float val = q[idx];
    float diff1 = val - nd->divlow, diff2 = val - nd->divhigh;
    int best, other;
    float cut;
    if (diff1 + diff2 < 0) { best = nd->left; other = nd->right; cut = diff2 * diff2; }
    else { best = nd->right; other = nd->left; cut = diff1 * diff1; }
    kd_search_level(t, res, q, best, mindistsq, dists);
    float dst = dists[idx];
    mindistsq = mindistsq + cut - dst;
    dists[idx] = cut;
    if (mindistsq <= res->worst) kd_search_level(t, res, q, other, mindistsq, dists);
    dists[idx] = dst;
}

static int kd_nearest(const kd_tree *t, const float *q, int *index, float *d2)
{
    if (t->n == 0) return 0;
    float dists[3] = {0, 0, 0}, distsq = 0;
    for (int d = 0; d < 3; d++) {
        if (q[d] < t->root_lo[d]) { float x = q[d] - t->root_lo[d]; dists[d] = x * x; distsq += dists[d]; }
        if (q[d] > t->root_hi[d]) { float x = q[d] - t->root_hi[d]; dists[d] = x * x; distsq += dists[d]; }
    }
    kd_result r = {FLT_MAX, -1};
    kd_search_level(t, &r, q, 0, distsq, dists);
    *index = r.index;
    *d2 = r.worst;
    return r.index >= 0;
}

/* ------------------------------------------------------------------ helpers */
static inline int finite3(const float *p) { return isfinite(p[0]) && isfinite(p[1]) && isfinite(p[2]); }

/* column-major 4x4 f32 product, fixed op order, no FMA (shared spec with the product) */
void orc_mat4_mul(const float *a, const float *b, float *c)
{
    float r[16];
    for (int j = 0; j < 4; j++)
        for (int i = 0; i < 4; i++) {
            float s = a[i] * b[j * 4];
            s = s + a[4 + i] * b[j * 4 + 1];
            s = s + a[8 + i] * b[j * 4 + 2];
            s = s + a[12 + i] * b[j * 4 + 3];
            r[j * 4 + i] = s;
        }
    memcpy(c, r, sizeof(r));
}

/* xyz <- M[0:3,0:3] * xyz + M[0:3,3], f32, fixed order ((m0*x + m1*y) + m2*z) + m3 */
static inline void xform_point(const float *M, const float *p, float *o)
{
    float x = p[0], y = p[1], z = p[2];
    float ox = M[0] * x; ox = ox + M[4] * y; ox = ox + M[8] * z; ox = ox + M[12];
    float oy = M[1] * x; oy = oy + M[5] * y; oy = oy + M[9] * z; oy = oy + M[13];
    float oz = M[2] * x; oz = oz + M[6] * y; oz = oz + M[10] * z; oz = oz + M[14];
    o[0] = ox; o[1] = oy; o[2] = oz;
}

static void mat4_identity(float *m)
{
    memset(m, 0, 64);
    m[0] = m[5] = m[10] = m[15] = 1.0f;
}

static int mat4_is_identity(const float *m)
{
    float I[16];
    mat4_identity(I);
    return memcmp(I, m, 64) == 0;
}

/* Umeyama (A.3) from centred second moments: sigma (row-major, q rows x p cols) */
static void rigid_from_sigma(const double *sigma, const double *mu_p, const double *mu_q, float *T)
{
    double U[9], s[3], V[9];
    orc_svd_jacobi(sigma, 3, U, s, V);
    double S[3] = {1, 1, 1};
    if (orc_det3(U) * orc_det3(V) < 0) S[2] = -1;
    double R[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            double v = 0;
            for (int k = 0; k < 3; k++) v += U[i * 3 + k] * S[k] * V[j * 3 + k];
            R[i * 3 + j] = v;
        }
    double t[3];
    for (int i = 0; i < 3; i++)
        t[i] = mu_q[i] - (R[i * 3] * mu_p[0] + R[i * 3 + 1] * mu_p[1] + R[i * 3 + 2] * mu_p[2]);
    mat4_identity(T);
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) T[j * 4 + i] = (float)R[i * 3 + j];
        T[12 + i] = (float)t[i];
    }
}

int orc_umeyama_from_sums(const double *sums, float *T)
{
    double n = sums[0];
    if (!(n >= 1)) return -1;
    double mu_p[3], mu_q[3], sigma[9];
    for (int i = 0; i < 3; i++) { mu_p[i] = sums[1 + i] / n; mu_q[i] = sums[4 + i] / n; }
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) sigma[i * 3 + j] = sums[7 + i * 3 + j] / n - mu_q[i] * mu_p[j];
    rigid_from_sigma(sigma, mu_p, mu_q, T);
    return 0;
}

/* ------------------------------------------------------------------ ICP object */
struct orc_icp {
    /* target */
    float *txyz;   /* finite (and de-duplicated) target points, nt x 3 */
    int *tmap;     /* -> index in the caller's array */
    int nt;
    kd_tree *tree;
    int num_threads_build;
    double sec_build;
    /* source */
    float *sxyz, *cur;
    uint8_t *svalid;
    int ns;
    /* loop state (A.2) */
    orc_icp_params prm;
    float final_t[16], t_inc[16];
    int iterations, state, converged, active;
    int similar;
    double prev_mse, cur_mse;
    int32_t *cidx; /* index into txyz (-1 none) */
    float *cd2;
    uint64_t ncorr;
    double sums_last[ORC_NUM_SUMS];
    double sec_search, t_begin;
};

void orc_icp_params_default(orc_icp_params *p)
{
    memset(p, 0, sizeof(*p));
    p->max_iterations = 10;
    p->criteria_mode = ORC_CRITERIA_PCL;
    p->accum_mode = ORC_ACCUM_F64;
    p->nn_mode = ORC_NN_KDTREE;
    p->dedup_target = 0;
    p->num_threads = 1;
    p->max_correspondence_distance = sqrt(DBL_MAX);
    p->transformation_epsilon = 0.0;
    p->transformation_rotation_epsilon = 0.0;
    p->euclidean_fitness_epsilon = -DBL_MAX;
}

/* src/incremental_icp.hpp:46-49 */
void orc_icp_params_reference(orc_icp_params *p)
{
    orc_icp_params_default(p);
    p->max_iterations = 100;
    p->max_correspondence_distance = 0.01;
    p->transformation_epsilon = 1;
    p->euclidean_fitness_epsilon = 1000;
}

orc_icp *orc_icp_create(void) { return (orc_icp *)calloc(1, sizeof(orc_icp)); }

static void free_target(orc_icp *o)
{
    kd_free(o->tree); o->tree = NULL;
    free(o->txyz); o->txyz = NULL;
    free(o->tmap); o->tmap = NULL;
    o->nt = 0;
}

void orc_icp_destroy(orc_icp *o)
{
    if (!o) return;
    free_target(o);
    free(o->sxyz); free(o->cur); free(o->svalid); free(o->cidx); free(o->cd2);
    free(o);
}

typedef struct dd_rec { uint32_t b[3]; int idx; } dd_rec;
static int dd_cmp(const void *a, const void *b)
{
    const dd_rec *x = (const dd_rec *)a, *y = (const dd_rec *)b;
    for (int k = 0; k < 3; k++)
        if (x->b[k] != y->b[k]) return x->b[k] < y->b[k] ? -1 : 1;
    return x->idx < y->idx ? -1 : (x->idx > y->idx);
}
static int int_cmp(const void *a, const void *b) { return *(const int *)a - *(const int *)b; }

/* icp.setInputTarget + Registration::initCompute's tree build (A.1): non-finite points are
 * left out.  dedup: additionally keep only the lowest-index copy of bit-identical xyz (the
 * NN distance and matched coordinates are unchanged; see DESIGN.md "duplicate points"). */
int orc_icp_set_target(orc_icp *o, const void *pts, size_t n, size_t stride, int is_dense,
                       int dedup, int num_threads)
{
    (void)is_dense; (void)num_threads;
    free_target(o);
    double t0 = now_sec();
    const char *base = (const char *)pts;
    int *keep = (int *)malloc(sizeof(int) * (n ? n : 1));
    int nk = 0;
    for (size_t i = 0; i < n; i++) {
        const float *p = (const float *)(base + i * stride);
        if (finite3(p)) keep[nk++] = (int)i;
    }
    if (dedup && nk > 1) {
        dd_rec *r = (dd_rec *)malloc(sizeof(dd_rec) * nk);
        for (int k = 0; k < nk; k++) {
            const float *p = (const float *)(base + (size_t)keep[k] * stride);
            float q[3] = {p[0] + 0.0f, p[1] + 0.0f, p[2] + 0.0f}; /* -0 -> +0 */
            memcpy(r[k].b, q, 12);
            r[k].idx = keep[k];
        }
        qsort(r, nk, sizeof(dd_rec), dd_cmp);
        int m = 0;
        for (int k = 0; k < nk; k++)
            if (k == 0 || memcmp(r[k].b, r[k - 1].b, 12) != 0) keep[m++] = r[k].idx;
        nk = m;
        qsort(keep, nk, sizeof(int), int_cmp);
        free(r);
    }
    o->nt = nk;
    o->txyz = (float *)malloc(sizeof(float) * 3 * (nk ? nk : 1));
    o->tmap = keep;
    for (int k = 0; k < nk; k++) memcpy(o->txyz + 3 * k, base + (size_t)keep[k] * stride, 12);
    o->tree = kd_build(o->txyz, nk);
    o->sec_build = now_sec() - t0;
    return 0;
}

int orc_icp_set_source(orc_icp *o, const void *pts, size_t n, size_t stride, int is_dense)
{
    (void)is_dense;
    free(o->sxyz); free(o->cur); free(o->svalid); free(o->cidx); free(o->cd2);
    o->ns = (int)n;
    size_t m = n ? n : 1;
    o->sxyz = (float *)malloc(12 * m);
    o->cur = (float *)malloc(12 * m);
    o->svalid = (uint8_t *)malloc(m);
    o->cidx = (int32_t *)malloc(4 * m);
    o->cd2 = (float *)malloc(4 * m);
    const char *base = (const char *)pts;
    for (size_t i = 0; i < n; i++) {
        memcpy(o->sxyz + 3 * i, base + i * stride, 12);
        o->svalid[i] = (uint8_t)finite3(o->sxyz + 3 * i);
    }
    return 0;
}

/* A.2 prologue: final = guess; input_transformed = guess * input (skipped for identity) */
int orc_icp_begin(orc_icp *o, const float *guess, const orc_icp_params *params)
{
    if (!o->tree) return -5;
    if (!o->sxyz) return -8;
    o->prm = *params;
    if (guess) memcpy(o->final_t, guess, 64); else mat4_identity(o->final_t);
    mat4_identity(o->t_inc);
    int ident = mat4_is_identity(o->final_t);
    for (int i = 0; i < o->ns; i++) {
        if (!ident && o->svalid[i]) xform_point(o->final_t, o->sxyz + 3 * i, o->cur + 3 * i);
        else memcpy(o->cur + 3 * i, o->sxyz + 3 * i, 12);
    }
    o->iterations = 0;
    o->state = ORC_CONV_NOT_CONVERGED;
    o->converged = 0;
    o->similar = 0;
    o->prev_mse = DBL_MAX;
    o->cur_mse = 0;
    o->ncorr = 0;
    o->sec_search = 0;
    o->active = 1;
    o->t_begin = now_sec();
    memset(o->sums_last, 0, sizeof(o->sums_last));
    return 0;
}

/* CorrespondenceEstimation::determineReciprocalCorrespondences [PCL: registration/impl/correspondence_estimation.hpp,
 * recalled]: a pair (i, t) survives only if i is the nearest source point of t (k-d tree over the CURRENT source,
 * nearestKSearch (tgt[t], 1); here the lowest index among equidistant source points, like every search of this oracle) */
static void recip_filter(orc_icp *o)
{
    int nv = 0;
    float *pts = (float *)malloc((size_t)(o->ns ? o->ns : 1) * 12);
    int *map = (int *)malloc((size_t)(o->ns ? o->ns : 1) * sizeof(int));
    for (int i = 0; i < o->ns; i++)
        if (o->svalid[i]) {
            memcpy(pts + 3 * nv, o->cur + 3 * i, 12);
            map[nv++] = i;
        }
    kd_tree *st = nv ? kd_build(pts, nv) : NULL;
    for (int i = 0; i < o->ns; i++) {
        if (o->cidx[i] < 0) continue;
        int back = -1;
        float d2 = 0.0f;
        if (!st || !kd_nearest(st, o->txyz + 3 * o->cidx[i], &back, &d2) || map[back] != i) o->cidx[i] = -1;
    }
    if (st) kd_free(st);
    free(pts);
    free(map);
}

/* CorrespondenceRejectorTrimmed::getRemainingCorrespondences [PCL: registration/src/correspondence_rejection_trimmed.cpp,
 * recalled]: number_valid = int (floor (overlap_ratio * float (size))); if smaller than size, std::nth_element by
 * distance and resize: the number_valid closest pairs remain (which of several equidistant pairs nth_element keeps is
 * unspecified; here: lowest source index first) */
typedef struct trim_rec { float d2; int i; } trim_rec;
static int trim_cmp(const void *a, const void *b)
{
    const trim_rec *x = (const trim_rec *)a, *y = (const trim_rec *)b;
    if (x->d2 != y->d2) return x->d2 < y->d2 ? -1 : 1;
    return x->i - y->i;
}
static void trim_filter(orc_icp *o)
{
    int n = 0;
    trim_rec *r = (trim_rec *)malloc((size_t)(o->ns ? o->ns : 1) * sizeof(trim_rec));
    for (int i = 0; i < o->ns; i++)
        if (o->cidx[i] >= 0) {
            r[n].d2 = o->cd2[i];
            r[n].i = i;
            n++;
        }
    const unsigned int keep = (unsigned int)(int)floorf((float)o->prm.trim_overlap_ratio * (float)n);
    if (keep < (unsigned int)n) {
        qsort(r, (size_t)n, sizeof(trim_rec), trim_cmp);
        for (int k = (int)keep; k < n; k++) o->cidx[r[k].i] = -1;
    }
    free(r);
}

/* CorrespondenceEstimation::determineCorrespondences (A.1 + A.7a) */
int orc_icp_search(orc_icp *o, int32_t *index_out, float *sqr_dist_out)
{
    if (!o->active) return -9;
    double t0 = now_sec();
    const double gate = o->prm.max_correspondence_distance * o->prm.max_correspondence_distance;
    const int brute = o->prm.nn_mode == ORC_NN_BRUTE;
    int nthreads = o->prm.num_threads > 0 ? o->prm.num_threads : 1;
    (void)nthreads;
#pragma omp parallel for schedule(dynamic, 1024) num_threads(nthreads) if (nthreads > 1)
    for (int i = 0; i < o->ns; i++) {
        int idx = -1;
        float d2 = 0.0f;
        int ok = 0;
        if (o->svalid[i]) {
            const float *q = o->cur + 3 * i;
            if (brute) {
                float best = FLT_MAX;
                for (int j = 0; j < o->nt; j++) {
                    float d = l2_simple(q, o->txyz + 3 * j);
                    if (d < best) { best = d; idx = j; }
                }
                d2 = best;
                ok = idx >= 0;
            } else {
                ok = kd_nearest(o->tree, q, &idx, &d2);
            }
        }
        if (ok && !((double)d2 > gate)) {
            o->cidx[i] = idx;
            o->cd2[i] = d2;
        } else {
            o->cidx[i] = -1;
            o->cd2[i] = ok ? d2 : 0.0f;
        }
    }
    if (o->prm.use_reciprocal) recip_filter(o);
    if (o->prm.trim_overlap_ratio > 0.0 && o->prm.trim_overlap_ratio < 1.0) trim_filter(o);
    uint64_t nc = 0;
    for (int i = 0; i < o->ns; i++) nc += o->cidx[i] >= 0;
    o->ncorr = nc;
    for (int i = 0; i < o->ns; i++) {
        if (index_out) index_out[i] = o->cidx[i] >= 0 ? o->tmap[o->cidx[i]] : -1;
        if (sqr_dist_out) sqr_dist_out[i] = o->cd2[i];
    }
    o->sec_search += now_sec() - t0;
    return 0;
}

int orc_icp_sums(orc_icp *o, double *sums)
{
    double s[ORC_NUM_SUMS];
    memset(s, 0, sizeof(s));
    for (int i = 0; i < o->ns; i++) {
        int j = o->cidx[i];
        if (j < 0) continue;
        const float *p = o->cur + 3 * i, *q = o->txyz + 3 * j;
        s[0] += 1.0;
        for (int a = 0; a < 3; a++) { s[1 + a] += p[a]; s[4 + a] += q[a]; }
        for (int a = 0; a < 3; a++)
            for (int b = 0; b < 3; b++) s[7 + a * 3 + b] += (double)q[a] * (double)p[b];
        s[16] += o->cd2[i];
    }
    memcpy(sums, s, sizeof(s));
    memcpy(o->sums_last, s, sizeof(s));
    return 0;
}

/* PCL-like float accumulation (A.3 as Eigen evaluates it in f32; summation order differs
 * from Eigen's vectorised reduction, the noise level is the same) */
static void umeyama_f32(orc_icp *o, float *T)
{
    float n = 0, mp[3] = {0, 0, 0}, mq[3] = {0, 0, 0};
    for (int i = 0; i < o->ns; i++) {
        int j = o->cidx[i];
        if (j < 0) continue;
        n += 1.0f;
        for (int a = 0; a < 3; a++) { mp[a] += o->cur[3 * i + a]; mq[a] += o->txyz[3 * j + a]; }
    }
    float inv = 1.0f / n;
    for (int a = 0; a < 3; a++) { mp[a] *= inv; mq[a] *= inv; }
    float sg[9] = {0};
    for (int i = 0; i < o->ns; i++) {
        int j = o->cidx[i];
        if (j < 0) continue;
        float dp[3], dq[3];
        for (int a = 0; a < 3; a++) { dp[a] = o->cur[3 * i + a] - mp[a]; dq[a] = o->txyz[3 * j + a] - mq[a]; }
        for (int a = 0; a < 3; a++)
            for (int b = 0; b < 3; b++) sg[a * 3 + b] += dq[a] * dp[b];
    }
    double sigma[9], mup[3], muq[3];
    for (int k = 0; k < 9; k++) sigma[k] = (double)(sg[k] * inv);
    for (int a = 0; a < 3; a++) { mup[a] = mp[a]; muq[a] = mq[a]; }
    rigid_from_sigma(sigma, mup, muq, T);
}

/* A.4 */
static int criteria_has_converged(orc_icp *o)
{
    const orc_icp_params *p = &o->prm;
    if (o->state != ORC_CONV_NOT_CONVERGED) { o->similar = 0; o->state = ORC_CONV_NOT_CONVERGED; }
    int is_similar = 0;
    if (o->iterations >= p->max_iterations) { o->state = ORC_CONV_ITERATIONS; return 1; }
    if (p->criteria_mode == ORC_CRITERIA_FIXED) return 0;
    double rot_thr = p->transformation_rotation_epsilon > 0 ? p->transformation_rotation_epsilon
                                                             : 1.0 - p->transformation_epsilon;
    double trans_thr = p->transformation_epsilon;
    const float *T = o->t_inc;
    double cos_angle = 0.5 * ((double)T[0] + (double)T[5] + (double)T[10] - 1.0);
    double tsq = (double)T[12] * T[12] + (double)T[13] * T[13] + (double)T[14] * T[14];
    if (cos_angle >= rot_thr && tsq <= trans_thr) {
        if (o->similar >= 0 /* max_iterations_similar_transforms_ = 0 */) { o->state = ORC_CONV_TRANSFORM; return 1; }
        is_similar = 1;
    }
    if (fabs(o->cur_mse - o->prev_mse) < 1e-12) {
        o->state = ORC_CONV_ABS_MSE; return 1;
    }
    if (fabs(o->cur_mse - o->prev_mse) / o->prev_mse < p->euclidean_fitness_epsilon) {
        o->state = ORC_CONV_REL_MSE; return 1;
    }
    if (is_similar) o->similar++; else o->similar = 0;
    o->prev_mse = o->cur_mse;
    return 0;
}

int orc_icp_update(orc_icp *o, const double *sums_in, float *t_inc_out, int *done)
{
    if (!o->active) return -9;
    double sums[ORC_NUM_SUMS];
    if (sums_in) memcpy(sums, sums_in, sizeof(sums)); else orc_icp_sums(o, sums);
    memcpy(o->sums_last, sums, sizeof(sums));
    uint64_t n = (uint64_t)(sums[0] + 0.5);
    o->ncorr = n;
    if (n < 3) {   /* min_number_correspondences_ */
        o->state = ORC_CONV_NO_CORRESPONDENCES;
        o->converged = 0;
        if (done) *done = 1;
        return 0;
    }
    if (o->prm.accum_mode == ORC_ACCUM_F32 && !sums_in) umeyama_f32(o, o->t_inc);
    else orc_umeyama_from_sums(sums, o->t_inc);
    for (int i = 0; i < o->ns; i++)
        if (o->svalid[i]) xform_point(o->t_inc, o->cur + 3 * i, o->cur + 3 * i);
    orc_mat4_mul(o->t_inc, o->final_t, o->final_t);
    o->iterations++;
    o->cur_mse = sums[16] / sums[0];
    o->converged = criteria_has_converged(o);
    if (t_inc_out) memcpy(t_inc_out, o->t_inc, 64);
    if (done) *done = o->converged;
    return 0;
}

int orc_icp_end(orc_icp *o, orc_icp_result *r, void *aligned_out, size_t out_stride)
{
    if (r) {
        memset(r, 0, sizeof(*r));
        memcpy(r->transform, o->final_t, 64);
        r->converged = o->converged;
        r->state = o->state;
        r->iterations = o->iterations;
        r->n_correspondences = o->ncorr;
        r->mse = o->cur_mse;
        memcpy(r->sums_last, o->sums_last, sizeof(r->sums_last));
        r->sec_build = o->sec_build;
        r->sec_search = o->sec_search;
        r->sec_total = now_sec() - o->t_begin;
    }
    if (aligned_out) {   /* output = final * input (A.2 epilogue) */
        char *ob = (char *)aligned_out;
        for (int i = 0; i < o->ns; i++) {
            float *dst = (float *)(ob + (size_t)i * out_stride);
            if (o->svalid[i]) xform_point(o->final_t, o->sxyz + 3 * i, dst);
            else memcpy(dst, o->sxyz + 3 * i, 12);
            if (out_stride >= 16) dst[3] = 1.0f;
        }
    }
    o->active = 0;
    return 0;
}

int orc_icp_align(orc_icp *o, const float *guess, const orc_icp_params *params,
                  orc_icp_result *result, void *aligned_out, size_t out_stride)
{
    int rc = orc_icp_begin(o, guess, params);
    if (rc) return rc;
    int done = 0;
    while (!done) {
        orc_icp_search(o, NULL, NULL);
        orc_icp_update(o, NULL, NULL, &done);
    }
    return orc_icp_end(o, result, aligned_out, out_stride);
}

int orc_icp_get_current(orc_icp *o, float *xyz_out)
{
    memcpy(xyz_out, o->cur, 12 * (size_t)o->ns);
    return 0;
}

/* ------------------------------------------------------------------ A.8 */
int orc_transform_cloud(const void *in, void *out, size_t n, size_t stride, int is_dense,
                        const float *T)
{
    const char *ib = (const char *)in;
    char *ob = (char *)out;
    for (size_t i = 0; i < n; i++) {
        float p[3], q[3];
        memcpy(p, ib + i * stride, 12);
        if (ib != ob) memmove(ob + i * stride, ib + i * stride, stride);
        if (!is_dense && !finite3(p)) continue;
        xform_point(T, p, q);
        memcpy(ob + i * stride, q, 12);
    }
    return 0;
}

/* ------------------------------------------------------------------ A.5 */
typedef struct avg_he { int ix, iy, iz, count; float c[7]; } avg_he;

static size_t avg_flush(char *ob, size_t op, size_t stride, avg_he *h)
{
    float inv = (float)h->count;
    float c[7];
    for (int k = 0; k < 7; k++) c[k] = h->c[k] / inv;
    char *rec = ob + op * stride;
    /* default-constructed PointXYZRGB: xyz 0, data[3] 1, rgba 0xff000000, rest 0 */
    memset(rec, 0, stride);
    float one = 1.0f;
    memcpy(rec, c, 12);
    memcpy(rec + 12, &one, 4);
    int rgb = ((int)c[4]) << 16 | ((int)c[5]) << 8 | ((int)c[6]);
    memcpy(rec + 16, &rgb, 4);
    return op + 1;
}

int orc_approx_voxel_grid(const void *in, size_t n, size_t stride, const float *leaf, void *out,
                          size_t *n_out)
{
    if (stride < 20) return -1;
    enum { HIST = 512 };
    avg_he *hist = (avg_he *)calloc(HIST, sizeof(avg_he));
    float inv_leaf[3] = {1.0f / leaf[0], 1.0f / leaf[1], 1.0f / leaf[2]};
    const char *ib = (const char *)in;
    char *tmp = (char *)malloc((n ? n : 1) * stride);  /* in == out is handled by a temporary */
    size_t op = 0;
    for (size_t cp = 0; cp < n; cp++) {
        float p[4];
        unsigned char rgba[4];
        memcpy(p, ib + cp * stride, 12);
        memcpy(&p[3], ib + cp * stride + 16, 4); /* the 'rgb' field read as a float */
        memcpy(rgba, ib + cp * stride + 16, 4);  /* b g r a */
        if (!finite3(p)) continue; /* PCL: undefined (float->int of NaN); oracle skips them */
        int ix = (int)floorf(p[0] * inv_leaf[0]);
        int iy = (int)floorf(p[1] * inv_leaf[1]);
        int iz = (int)floorf(p[2] * inv_leaf[2]);
        unsigned int hash = (unsigned int)((ix * 7171 + iy * 3079 + iz * 4231) & (HIST - 1));
        avg_he *h = &hist[hash];
        if (h->count && (ix != h->ix || iy != h->iy || iz != h->iz)) {
            op = avg_flush(tmp, op, stride, h);
            h->count = 0;
            memset(h->c, 0, sizeof(h->c));
        }
        h->ix = ix; h->iy = iy; h->iz = iz;
        h->count++;
        float scratch[7] = {p[0], p[1], p[2], p[3], (float)rgba[2], (float)rgba[1], (float)rgba[0]};
        for (int k = 0; k < 7; k++) h->c[k] += scratch[k];
    }
    for (int i = 0; i < HIST; i++)
        if (hist[i].count) op = avg_flush(tmp, op, stride, &hist[i]);
    memcpy(out, tmp, op * stride);
    *n_out = op;
    free(tmp);
    free(hist);
    return 0;
}
