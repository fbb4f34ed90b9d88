/*
 * orc_linalg.h — small dense linear algebra for the CPU oracle (TEST INFRASTRUCTURE ONLY).
 *
 * The oracle is the checker, never the product: only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may use anything under oracle/.
 *
 * Stands in for the Eigen pieces PCL calls on the hot path (none of Eigen is available in
 * this image):
 *   - Eigen::JacobiSVD<Matrix3f/Matrix<double,6,6>>(ComputeFullU|ComputeFullV) and .solve()
 *       [PCL: common/eigen.h umeyama(); registration/impl/ndt.hpp computeTransformation]
 *   - Eigen::SelfAdjointEigenSolver<Matrix3d>  [PCL: filters/impl/voxel_grid_covariance.hpp]
 * Algorithms are textbook Jacobi iterations in f64 (published: Golub & Van Loan §8.5/§8.6).
 */
#ifndef ORC_LINALG_H_
#define ORC_LINALG_H_

#include <math.h>
#include <string.h>

/* One-sided (Hestenes) Jacobi SVD of an n x n row-major matrix, n <= 6.
 * A = U diag(s) V^T, s descending, U and V full orthogonal. */
static void orc_svd_jacobi(const double *A, int n, double *U, double *s, double *V)
{
    double W[36], Vm[36];
    memcpy(W, A, sizeof(double) * n * n);
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) Vm[i * n + j] = (i == j) ? 1.0 : 0.0;

    for (int sweep = 0; sweep < 60; sweep++) {
        double off = 0.0;
        for (int p = 0; p < n - 1; p++) {
            for (int q = p + 1; q < n; q++) {
                double alpha = 0, beta = 0, gamma = 0;
                for (int i = 0; i < n; i++) {
                    alpha += W[i * n + p] * W[i * n + p];
                    beta += W[i * n + q] * W[i * n + q];
                    gamma += W[i * n + p] * W[i * n + q];
                }
                if (gamma == 0.0) continue;
                double lim = 1e-300 + 1e-32 * alpha * beta;
                if (gamma * gamma <= lim) continue;
                off += gamma * gamma / (alpha * beta + 1e-300);
                double zeta = (beta - alpha) / (2.0 * gamma);
                double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                double c = 1.0 / sqrt(1.0 + t * t), sn = c * t;
                for (int i = 0; i < n; i++) {
                    double wp = W[i * n + p], wq = W[i * n + q];
                    W[i * n + p] = c * wp - sn * wq;
                    W[i * n + q] = sn * wp + c * wq;
                    double vp = Vm[i * n + p], vq = Vm[i * n + q];
                    Vm[i * n + p] = c * vp - sn * vq;
                    Vm[i * n + q] = sn * vp + c * vq;
                }
            }
        }
        if (off < 1e-30) break;
    }
    /* singular values = column norms; sort descending */
    double sv[6];
    int order[6];
    for (int j = 0; j < n; j++) {
        double nn = 0;
        for (int i = 0; i < n; i++) nn += W[i * n + j] * W[i * n + j];
        sv[j] = sqrt(nn);
        order[j] = j;
    }
    for (int a = 0; a < n - 1; a++)
        for (int b = a + 1; b < n; b++)
            if (sv[order[b]] > sv[order[a]]) { int t = order[a]; order[a] = order[b]; order[b] = t; }
    double smax = sv[order[0]];
    int rank = 0;
    for (int k = 0; k < n; k++) {
        int j = order[k];
        s[k] = sv[j];
        for (int i = 0; i < n; i++) V[i * n + k] = Vm[i * n + j];
        if (sv[j] > 1e-13 * smax && sv[j] > 0) {
            for (int i = 0; i < n; i++) U[i * n + k] = W[i * n + j] / sv[j];
            rank = k + 1;
        } else {
            for (int i = 0; i < n; i++) U[i * n + k] = 0.0;
        }
    }
    /* complete U to a full orthonormal basis (Gram-Schmidt against unit vectors) */
    for (int k = rank; k < n; k++) {
        double best[6];
        double bestn = -1;
        for (int e = 0; e < n; e++) {
            double v[6];
            for (int i = 0; i < n; i++) v[i] = (i == e) ? 1.0 : 0.0;
            for (int pass = 0; pass < 2; pass++)
                for (int m = 0; m < k; m++) {
                    double d = 0;
                    for (int i = 0; i < n; i++) d += v[i] * U[i * n + m];
                    for (int i = 0; i < n; i++) v[i] -= d * U[i * n + m];
                }
            double nn = 0;
            for (int i = 0; i < n; i++) nn += v[i] * v[i];
            if (nn > bestn) { bestn = nn; memcpy(best, v, sizeof(double) * n); }
        }
        double inv = 1.0 / sqrt(bestn);
        for (int i = 0; i < n; i++) U[i * n + k] = best[i] * inv;
    }
}

static double orc_det3(const double *M)
{
    return M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) +
           M[2] * (M[3] * M[7] - M[4] * M[6]);
}

/* JacobiSVD::solve: x = V diag(1/s_i, i < rank) U^T b, rank by Eigen's default threshold
 * (s_i > max(n) * eps * s_max). */
static void orc_svd_solve(const double *A, int n, const double *b, double *x)
{
    double U[36], s[6], V[36], y[6];
    orc_svd_jacobi(A, n, U, s, V);
    double thr = (double)n * 2.220446049250313e-16 * s[0];
    for (int k = 0; k < n; k++) {
        double d = 0;
        for (int i = 0; i < n; i++) d += U[i * n + k] * b[i];
        y[k] = (s[k] > thr && s[k] > 0) ? d / s[k] : 0.0;
    }
    for (int i = 0; i < n; i++) {
        double v = 0;
        for (int k = 0; k < n; k++) v += V[i * n + k] * y[k];
        x[i] = v;
    }
}

/* Symmetric 3x3 eigen-decomposition (cyclic Jacobi); evals ascending, evecs in columns. */
static void orc_eig_sym3(const double *Ain, double *evals, double *evecs)
{
    double A[9], V[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    memcpy(A, Ain, sizeof(A));
    for (int sweep = 0; sweep < 64; sweep++) {
        double off = A[1] * A[1] + A[2] * A[2] + A[5] * A[5];
        double diag = A[0] * A[0] + A[4] * A[4] + A[8] * A[8];
        if (off <= 1e-34 * (diag + 1e-300)) break;
        for (int p = 0; p < 2; p++)
            for (int q = p + 1; q < 3; q++) {
                double apq = A[p * 3 + q];
                if (apq == 0.0) continue;
                double theta = (A[q * 3 + q] - A[p * 3 + p]) / (2.0 * apq);
                double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < 3; k++) {
                    double akp = A[k * 3 + p], akq = A[k * 3 + q];
                    A[k * 3 + p] = c * akp - s * akq;
                    A[k * 3 + q] = s * akp + c * akq;
                }
                for (int k = 0; k < 3; k++) {
                    double apk = A[p * 3 + k], aqk = A[q * 3 + k];
                    A[p * 3 + k] = c * apk - s * aqk;
                    A[q * 3 + k] = s * apk + c * aqk;
                }
                for (int k = 0; k < 3; k++) {
                    double vkp = V[k * 3 + p], vkq = V[k * 3 + q];
                    V[k * 3 + p] = c * vkp - s * vkq;
                    V[k * 3 + q] = s * vkp + c * vkq;
                }
            }
    }
    int o[3] = {0, 1, 2};
    double d[3] = {A[0], A[4], A[8]};
    for (int a = 0; a < 2; a++)
        for (int b = a + 1; b < 3; b++)
            if (d[o[b]] < d[o[a]]) { int t = o[a]; o[a] = o[b]; o[b] = t; }
    for (int k = 0; k < 3; k++) {
        evals[k] = d[o[k]];
        for (int i = 0; i < 3; i++) evecs[i * 3 + k] = V[i * 3 + o[k]];
    }
}

/* inverse of a 3x3 (row-major); returns 0 if singular / non-finite */
static int orc_inv3(const double *M, double *Inv)
{
    double det = orc_det3(M);
    if (!(fabs(det) > 0.0) || !isfinite(det)) return 0;
    double id = 1.0 / det;
    Inv[0] = (M[4] * M[8] - M[5] * M[7]) * id;
    Inv[1] = (M[2] * M[7] - M[1] * M[8]) * id;
    Inv[2] = (M[1] * M[5] - M[2] * M[4]) * id;
    Inv[3] = (M[5] * M[6] - M[3] * M[8]) * id;
    Inv[4] = (M[0] * M[8] - M[2] * M[6]) * id;
    Inv[5] = (M[2] * M[3] - M[0] * M[5]) * id;
    Inv[6] = (M[3] * M[7] - M[4] * M[6]) * id;
    Inv[7] = (M[1] * M[6] - M[0] * M[7]) * id;
    Inv[8] = (M[0] * M[4] - M[1] * M[3]) * id;
    for (int i = 0; i < 9; i++)
        if (!isfinite(Inv[i])) return 0;
    return 1;
}

#endif /* ORC_LINALG_H_ */
