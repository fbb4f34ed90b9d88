// host_linalg.hpp — host-side dense algebra of the registration loop (product code).
//
// Replaces the Eigen pieces PCL runs on the host between kernels:
//   Eigen::umeyama (JacobiSVD of the 3x3 cross-covariance)  [PCL TransformationEstimationSVD,
//     called from pcl::IterativeClosestPoint::align — reference src/incremental_icp.hpp:59]
//   JacobiSVD<Matrix6d>::solve for the NDT Newton step       [reference ndt_edge...hpp:83,92]
//   SelfAdjointEigenSolver<Matrix3d> for the NDT voxel covariances
// Everything is f64; results are cast to f32 only when they become a 4x4 transform.
#pragma once

#include <cmath>
#include <cstring>

// the 3x3 pieces also run on the device (the device-resident ICP loop, icp.hip k_icp_solve):
// same source, same operation order, f64 +,-,*,/,sqrt only, so both sides give the same bits
#if defined(__HIPCC__)
#define RSREG_HD __host__ __device__
#else
#define RSREG_HD
#endif

namespace rsreg {

// Column-major 4x4 float transform (memcpy-compatible with Eigen::Matrix4f).
struct Mat4f {
    float m[16];
    RSREG_HD static Mat4f identity()
    {
        Mat4f r;
        for (int i = 0; i < 16; ++i) r.m[i] = (i % 5 == 0) ? 1.0f : 0.0f;
        return r;
    }
    bool is_identity() const
    {
        Mat4f i = identity();
        return std::memcmp(i.m, m, sizeof(m)) == 0;
    }
    RSREG_HD float &operator()(int r, int c) { return m[c * 4 + r]; }
    RSREG_HD float operator()(int r, int c) const { return m[c * 4 + r]; }
};

// c = a * b in f32 with the fixed evaluation order ((a0 b0 + a1 b1) + a2 b2) + a3 b3, no FMA.
// The same order is part of the parity spec (oracle/icp_oracle.c orc_mat4_mul).
RSREG_HD inline Mat4f mul(const Mat4f &a, const Mat4f &b)
{
#pragma clang fp contract(off)
    Mat4f r;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float s = a.m[i] * b.m[j * 4];
            s = s + a.m[4 + i] * b.m[j * 4 + 1];
            s = s + a.m[8 + i] * b.m[j * 4 + 2];
            s = s + a.m[12 + i] * b.m[j * 4 + 3];
            r.m[j * 4 + i] = s;
        }
    return r;
}

// Small fixed-capacity square matrix helpers, row-major, n <= 6.
template <int N> struct SvdResult {
    double U[N * N], s[N], V[N * N];
};

// One-sided Jacobi: rotate column pairs of W = A*V until mutually orthogonal.
template <int N> RSREG_HD inline void jacobi_svd(const double *A, SvdResult<N> &out)
{
    double W[N * N], V[N * N];
    for (int i = 0; i < N * N; ++i) W[i] = A[i];
    for (int i = 0; i < N; ++i)
        for (int j = 0; j < N; ++j) V[i * N + j] = i == j ? 1.0 : 0.0;

    for (int sweep = 0; sweep < 64; ++sweep) {
        bool rotated = false;
        for (int p = 0; p + 1 < N; ++p)
            for (int q = p + 1; q < N; ++q) {
                double app = 0, aqq = 0, apq = 0;
                for (int k = 0; k < N; ++k) {
                    app += W[k * N + p] * W[k * N + p];
                    aqq += W[k * N + q] * W[k * N + q];
                    apq += W[k * N + p] * W[k * N + q];
                }
                if (apq == 0.0 || apq * apq <= 1e-32 * app * aqq) continue;
                rotated = true;
                const double tau = (aqq - app) / (2.0 * apq);
                const double t = (tau >= 0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
                const double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
                for (int k = 0; k < N; ++k) {
                    const double wp = W[k * N + p], wq = W[k * N + q];
                    W[k * N + p] = c * wp - s * wq;
                    W[k * N + q] = s * wp + c * wq;
                    const double vp = V[k * N + p], vq = V[k * N + q];
                    V[k * N + p] = c * vp - s * vq;
                    V[k * N + q] = s * vp + c * vq;
                }
            }
        if (!rotated) break;
    }
    double norm[N];
    int ord[N];
    for (int j = 0; j < N; ++j) {
        double nn = 0;
        for (int k = 0; k < N; ++k) nn += W[k * N + j] * W[k * N + j];
        norm[j] = sqrt(nn);
        ord[j] = j;
    }
    for (int a = 0; a + 1 < N; ++a)
        for (int b = a + 1; b < N; ++b)
            if (norm[ord[b]] > norm[ord[a]]) { int t = ord[a]; ord[a] = ord[b]; ord[b] = t; }
    const double smax = norm[ord[0]];
    int rank = 0;
    for (int k = 0; k < N; ++k) {
        const int j = ord[k];
        out.s[k] = norm[j];
        for (int i = 0; i < N; ++i) out.V[i * N + k] = V[i * N + j];
        if (norm[j] > 0 && norm[j] > 1e-13 * smax) {
            for (int i = 0; i < N; ++i) out.U[i * N + k] = W[i * N + j] / norm[j];
            rank = k + 1;
        } else {
            for (int i = 0; i < N; ++i) out.U[i * N + k] = 0.0;
        }
    }
    // complete U with unit vectors orthogonalised against the columns found so far
    for (int k = rank; k < N; ++k) {
        double best[N], bestn = -1;
        for (int e = 0; e < N; ++e) {
            double v[N];
            for (int i = 0; i < N; ++i) v[i] = i == e ? 1.0 : 0.0;
            for (int pass = 0; pass < 2; ++pass)
                for (int m = 0; m < k; ++m) {
                    double d = 0;
                    for (int i = 0; i < N; ++i) d += v[i] * out.U[i * N + m];
                    for (int i = 0; i < N; ++i) v[i] -= d * out.U[i * N + m];
                }
            double nn = 0;
            for (int i = 0; i < N; ++i) nn += v[i] * v[i];
            if (nn > bestn) { bestn = nn; for (int i = 0; i < N; ++i) best[i] = v[i]; }
        }
        const double inv = 1.0 / sqrt(bestn);
        for (int i = 0; i < N; ++i) out.U[i * N + k] = best[i] * inv;
    }
}

// jacobi_svd<3> written so that every array index is a compile-time constant (the loops over
// p, q, k unroll; orderings go through pick3): on the device all of it then lives in registers.
// Same operations in the same order as the generic template above, except that a column of U is scaled by the
// reciprocal of its norm (one division per column instead of three).
RSREG_HD inline double pick3(double a0, double a1, double a2, int i) { return i == 0 ? a0 : (i == 1 ? a1 : a2); }

// the columns of U (row-major 3 x 3) from `rank` on: unit vectors orthogonalised against the columns found so far (a
// cross-covariance of rank < 3: coplanar or collinear matches).  Shared by jacobi_svd3 and its lane-parallel twin on the
// device (icp_kernels.hpp: umeyama_wave).
RSREG_HD inline void complete_u3(double *U, int rank)
{
#pragma clang fp contract(off)
    constexpr int N = 3;
#pragma unroll
    for (int k = 0; k < N; ++k) {
        if (k < rank) continue;
        double best[3] = {0, 0, 0}, bestn = -1;
#pragma unroll
        for (int e = 0; e < N; ++e) {
            double v[3];
#pragma unroll
            for (int i = 0; i < N; ++i) v[i] = i == e ? 1.0 : 0.0;
#pragma unroll
            for (int pass = 0; pass < 2; ++pass)
#pragma unroll
                for (int m = 0; m < N; ++m) {
                    if (m >= k) continue;
                    double d = 0;
#pragma unroll
                    for (int i = 0; i < N; ++i) d += v[i] * U[i * N + m];
#pragma unroll
                    for (int i = 0; i < N; ++i) v[i] -= d * U[i * N + m];
                }
            double nn = 0;
#pragma unroll
            for (int i = 0; i < N; ++i) nn += v[i] * v[i];
            if (nn > bestn) {
                bestn = nn;
#pragma unroll
                for (int i = 0; i < N; ++i) best[i] = v[i];
            }
        }
        const double inv = 1.0 / sqrt(bestn);
#pragma unroll
        for (int i = 0; i < N; ++i) U[i * N + k] = best[i] * inv;
    }
}

// `v0` (optional): an orthogonal matrix to start from, e.g. the V of a nearby matrix (the
// cross-covariances of consecutive ICP iterations differ little: two or three sweeps then do
// what takes six from the identity).  Starting from the identity, W = A exactly as before.
RSREG_HD inline void jacobi_svd3(const double *A, SvdResult<3> &out, const double *v0 = nullptr)
{
#pragma clang fp contract(off)
    constexpr int N = 3;
    double W[9], V[9];
    if (v0) {
#pragma unroll
        for (int i = 0; i < 9; ++i) V[i] = v0[i];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) W[i * 3 + j] = (A[i * 3] * V[j] + A[i * 3 + 1] * V[3 + j]) + A[i * 3 + 2] * V[6 + j];
    } else {
#pragma unroll
        for (int i = 0; i < 9; ++i) W[i] = A[i];
#pragma unroll
        for (int i = 0; i < 9; ++i) V[i] = (i % 4 == 0) ? 1.0 : 0.0;
    }
    for (int sweep = 0; sweep < 64; ++sweep) {
        bool rotated = false;
#pragma unroll
        for (int p = 0; p + 1 < N; ++p)
#pragma unroll
            for (int q = p + 1; q < N; ++q) {
                double app = 0, aqq = 0, apq = 0;
#pragma unroll
                for (int k = 0; k < N; ++k) {
                    app += W[k * N + p] * W[k * N + p];
                    aqq += W[k * N + q] * W[k * N + q];
                    apq += W[k * N + p] * W[k * N + q];
                }
                if (apq == 0.0 || apq * apq <= 1e-32 * app * aqq) continue;
                rotated = true;
                const double tau = (aqq - app) / (2.0 * apq);
                const double t = (tau >= 0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
                const double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
#pragma unroll
                for (int k = 0; k < N; ++k) {
                    const double wp = W[k * N + p], wq = W[k * N + q];
                    W[k * N + p] = c * wp - s * wq;
                    W[k * N + q] = s * wp + c * wq;
                    const double vp = V[k * N + p], vq = V[k * N + q];
                    V[k * N + p] = c * vp - s * vq;
                    V[k * N + q] = s * vp + c * vq;
                }
            }
        if (!rotated) break;
    }
    double norm[3];
#pragma unroll
    for (int j = 0; j < N; ++j) {
        double nn = 0;
#pragma unroll
        for (int k = 0; k < N; ++k) nn += W[k * N + j] * W[k * N + j];
        norm[j] = sqrt(nn);
    }
    // the generic exchange sort on ord[], descending, with its three compare-exchanges spelled out
    int o0 = 0, o1 = 1, o2 = 2;
    if (pick3(norm[0], norm[1], norm[2], o1) > pick3(norm[0], norm[1], norm[2], o0)) { const int t = o0; o0 = o1; o1 = t; }
    if (pick3(norm[0], norm[1], norm[2], o2) > pick3(norm[0], norm[1], norm[2], o0)) { const int t = o0; o0 = o2; o2 = t; }
    if (pick3(norm[0], norm[1], norm[2], o2) > pick3(norm[0], norm[1], norm[2], o1)) { const int t = o1; o1 = o2; o2 = t; }
    const double smax = pick3(norm[0], norm[1], norm[2], o0);
    int rank = 0;
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const int j = k == 0 ? o0 : (k == 1 ? o1 : o2);
        const double nj = pick3(norm[0], norm[1], norm[2], j);
        out.s[k] = nj;
#pragma unroll
        for (int i = 0; i < N; ++i) out.V[i * N + k] = pick3(V[i * N], V[i * N + 1], V[i * N + 2], j);
        if (nj > 0 && nj > 1e-13 * smax) {
            const double inv = 1.0 / nj;
#pragma unroll
            for (int i = 0; i < N; ++i) out.U[i * N + k] = pick3(W[i * N], W[i * N + 1], W[i * N + 2], j) * inv;
            rank = k + 1;
        } else {
#pragma unroll
            for (int i = 0; i < N; ++i) out.U[i * N + k] = 0.0;
        }
    }
    complete_u3(out.U, rank);
}

RSREG_HD inline double det3(const double *M)
{
    return M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) +
           M[2] * (M[3] * M[7] - M[4] * M[6]);
}

// x = pinv(A) b with Eigen's JacobiSVD::solve rank rule (s_i > N * eps * s_max).
template <int N> inline void svd_solve(const double *A, const double *b, double *x)
{
    SvdResult<N> r;
    jacobi_svd<N>(A, r);
    const double thr = double(N) * 2.220446049250313e-16 * r.s[0];
    double y[N];
    for (int k = 0; k < N; ++k) {
        double d = 0;
        for (int i = 0; i < N; ++i) d += r.U[i * N + k] * b[i];
        y[k] = (r.s[k] > thr && r.s[k] > 0) ? d / r.s[k] : 0.0;
    }
    for (int i = 0; i < N; ++i) {
        double v = 0;
        for (int k = 0; k < N; ++k) v += r.V[i * N + k] * y[k];
        x[i] = v;
    }
}

// Eigen::umeyama(src, dst, with_scaling = false) from the 17 sums of an ICP iteration
// (layout: include/rsreg.h RSREG_NUM_SUMS).  Returns false when n < 1.
// `v_warm` (optional, 9 doubles, in/out): the V of the previous solve of the same alignment
RSREG_HD inline bool umeyama_from_sums(const double *sums, Mat4f &T, double *v_warm = nullptr)
{
#pragma clang fp contract(off)
    const double n = sums[0];
    if (!(n >= 1.0)) return false;
    double mu_p[3], mu_q[3], sigma[9];
    // one division, fifteen products: on the device this runs on a single lane between two search kernels, where an
    // f64 division is a chain of a dozen dependent instructions (host and device share this source: same bits)
    const double inv_n = 1.0 / n;
    for (int i = 0; i < 3; ++i) { mu_p[i] = sums[1 + i] * inv_n; mu_q[i] = sums[4 + i] * inv_n; }
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) sigma[i * 3 + j] = sums[7 + i * 3 + j] * inv_n - mu_q[i] * mu_p[j];
    SvdResult<3> r;
    jacobi_svd3(sigma, r, v_warm);
    if (v_warm) {
#pragma unroll
        for (int i = 0; i < 9; ++i) v_warm[i] = r.V[i];
    }
    double S[3] = {1, 1, 1};
    if (det3(r.U) * det3(r.V) < 0) S[2] = -1;
    double R[9];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            double v = 0;
#pragma unroll
            for (int k = 0; k < 3; ++k) v += r.U[i * 3 + k] * S[k] * r.V[j * 3 + k];
            R[i * 3 + j] = v;
        }
    T = Mat4f::identity();
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int j = 0; j < 3; ++j) T(i, j) = float(R[i * 3 + j]);
        T(i, 3) = float(mu_q[i] - (R[i * 3] * mu_p[0] + R[i * 3 + 1] * mu_p[1] + R[i * 3 + 2] * mu_p[2]));
    }
    return true;
}

// Symmetric 3x3 eigen-decomposition by cyclic Jacobi; eigenvalues ascending, vectors in columns.
inline void eig_sym3(const double *Ain, double *evals, double *evecs)
{
    double A[9], V[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    for (int i = 0; i < 9; ++i) A[i] = Ain[i];
    for (int sweep = 0; sweep < 64; ++sweep) {
        const double off = A[1] * A[1] + A[2] * A[2] + A[5] * A[5];
        const double dia = A[0] * A[0] + A[4] * A[4] + A[8] * A[8];
        if (off <= 1e-34 * (dia + 1e-300)) break;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                const double apq = A[p * 3 + q];
                if (apq == 0.0) continue;
                const double th = (A[q * 3 + q] - A[p * 3 + p]) / (2.0 * apq);
                const double t = (th >= 0 ? 1.0 : -1.0) / (std::fabs(th) + std::sqrt(th * th + 1.0));
                const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < 3; ++k) {
                    const double a = A[k * 3 + p], b = A[k * 3 + q];
                    A[k * 3 + p] = c * a - s * b;
                    A[k * 3 + q] = s * a + c * b;
                }
                for (int k = 0; k < 3; ++k) {
                    const double a = A[p * 3 + k], b = A[q * 3 + k];
                    A[p * 3 + k] = c * a - s * b;
                    A[q * 3 + k] = s * a + c * b;
                }
                for (int k = 0; k < 3; ++k) {
                    const double a = V[k * 3 + p], b = V[k * 3 + q];
                    V[k * 3 + p] = c * a - s * b;
                    V[k * 3 + q] = s * a + c * b;
                }
            }
    }
    int o[3] = {0, 1, 2};
    const double d[3] = {A[0], A[4], A[8]};
    for (int a = 0; a < 2; ++a)
        for (int b = a + 1; b < 3; ++b)
            if (d[o[b]] < d[o[a]]) { int t = o[a]; o[a] = o[b]; o[b] = t; }
    for (int k = 0; k < 3; ++k) {
        evals[k] = d[o[k]];
        for (int i = 0; i < 3; ++i) evecs[i * 3 + k] = V[i * 3 + o[k]];
    }
}

inline bool inv3(const double *M, double *Inv)
{
    const double det = det3(M);
    if (!(std::fabs(det) > 0.0) || !std::isfinite(det)) return false;
    const double id = 1.0 / det;
    Inv[0] = (M[4] * M[8] - M[5] * M[7]) * id;
    Inv[1] = (M[2] * M[7] - M[1] * M[8]) * id;
    Inv[2] = (M[1] * M[5] - M[2] * M[4]) * id;
    Inv[3] = (M[5] * M[6] - M[3] * M[8]) * id;
    Inv[4] = (M[0] * M[8] - M[2] * M[6]) * id;
    Inv[5] = (M[2] * M[3] - M[0] * M[5]) * id;
    Inv[6] = (M[3] * M[7] - M[4] * M[6]) * id;
    Inv[7] = (M[1] * M[6] - M[0] * M[7]) * id;
    Inv[8] = (M[0] * M[4] - M[1] * M[3]) * id;
    for (int i = 0; i < 9; ++i)
        if (!std::isfinite(Inv[i])) return false;
    return true;
}

}  // namespace rsreg
