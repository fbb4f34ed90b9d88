// ndt_math.hpp — the scalar side of pcl::NormalDistributionsTransform::computeTransformation's line search, written once
// for the host and the device (ndt_edge_based_registration.hpp:38-43,86-92: setStepSize / setTransformationEpsilon, align).
//
// PCL's computeStepLengthMT (More-Thuente) evaluates the score and its gradient at a sequence of trial steps, each a
// derivative pass over the source cloud, each depending on the one before.  With the control on the host every pass
// costs a round trip (launch, completion seen through pinned memory, a few microseconds of arithmetic, launch): 17 us on
// top of 19 us of kernels.  The line search is therefore a small state machine (NdtLs) that one thread advances -- on the
// host, or on the device at the end of a pass's final reduce, so that the trials of one line search are queued back to
// back and the host looks in once per Newton iteration (the 6 x 6 SVD of the Newton step stays on the host: a Jacobi SVD
// of that size is tens of microseconds of dependent f64 on one GPU lane).  Host and device run this very source with
// -ffp-contract=off: f64 + - * / sqrt, comparisons, and the sine / cosine below instead of the platforms' libm (whose
// last bits differ) -- the two give the same bits.
#pragma once

#include <cmath>

#include "host_linalg.hpp"

namespace rsreg {

constexpr int kNdtSums = 28;   // score, 6 gradient, 21 upper-triangle Hessian entries (= kNdtAcc of ndt_kernels.hpp)

// ---- sine and cosine: Cody-Waite reduction by pi/2 in two parts (exact for the |x| < ~10^5 a pose angle can take) and
// the classic minimax polynomials on [-pi/4, pi/4] (fdlibm's k_sin / k_cos coefficients); below one ulp off the
// correctly rounded value, and the same on every platform.
RSREG_HD inline double ndt_ksin(double x)
{
#pragma clang fp contract(off)
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
                 S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double z = x * x, v = z * x;
    const double r = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
    return x + v * (S1 + z * r);
}
RSREG_HD inline double ndt_kcos(double x)
{
#pragma clang fp contract(off)
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
                 C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    const double z = x * x;
    const double r = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
    return 1.0 - (0.5 * z - z * r);
}
RSREG_HD inline void ndt_sincos(double x, double &s, double &c)
{
#pragma clang fp contract(off)
    const double kInvPio2 = 6.36619772367581382433e-01, kPio2Hi = 1.57079632673412561417e+00, kPio2Lo = 6.07710050650619224932e-11;
    const double fn = floor(x * kInvPio2 + 0.5);
    const double r = (x - fn * kPio2Hi) - fn * kPio2Lo;
    const int q = (int)(long long)fn & 3;
    const double sr = ndt_ksin(r), cr = ndt_kcos(r);
    s = q == 0 ? sr : (q == 1 ? cr : (q == 2 ? -sr : -cr));
    c = q == 0 ? cr : (q == 1 ? -sr : (q == 2 ? -cr : sr));
}

// Translation(p0..2) * Rx(p3) * Ry(p4) * Rz(p5) assembled in f32 (PCL builds it with
// Eigen::Translation<float> * AngleAxis<float> products)
RSREG_HD inline Mat4f ndt_pose_matrix(const double *p)
{
    const float ax = (float)p[3], ay = (float)p[4], az = (float)p[5];
    double sd, cd;
    ndt_sincos((double)ax, sd, cd);
    const float cx = (float)cd, sx = (float)sd;
    ndt_sincos((double)ay, sd, cd);
    const float cy = (float)cd, sy = (float)sd;
    ndt_sincos((double)az, sd, cd);
    const float cz = (float)cd, sz = (float)sd;
    Mat4f Rx = Mat4f::identity(), Ry = Mat4f::identity(), Rz = Mat4f::identity(), Tr = Mat4f::identity();
    Rx(1, 1) = cx; Rx(1, 2) = -sx; Rx(2, 1) = sx; Rx(2, 2) = cx;
    Ry(0, 0) = cy; Ry(0, 2) = sy; Ry(2, 0) = -sy; Ry(2, 2) = cy;
    Rz(0, 0) = cz; Rz(0, 1) = -sz; Rz(1, 0) = sz; Rz(1, 1) = cz;
    Tr(0, 3) = (float)p[0]; Tr(1, 3) = (float)p[1]; Tr(2, 3) = (float)p[2];
    return mul(mul(mul(Tr, Rx), Ry), Rz);
}

// computeAngleDerivatives (Magnusson 2009 eq. 6.19 / 6.21), with PCL's small-angle snap: jang[8][3], hang[15][3]
RSREG_HD inline void ndt_angle_terms(const double *p, double (*jang)[3], double (*hang)[3])
{
#pragma clang fp contract(off)
    double cx, cy, cz, sx, sy, sz;
    if (fabs(p[3]) < 10e-5) { cx = 1.0; sx = 0.0; } else ndt_sincos(p[3], sx, cx);
    if (fabs(p[4]) < 10e-5) { cy = 1.0; sy = 0.0; } else ndt_sincos(p[4], sy, cy);
    if (fabs(p[5]) < 10e-5) { cz = 1.0; sz = 0.0; } else ndt_sincos(p[5], sz, cz);
#define RSREG_SET3(v, a, b, c) do { (v)[0] = (a); (v)[1] = (b); (v)[2] = (c); } while (0)
    RSREG_SET3(jang[0], -sx * sz + cx * sy * cz, -sx * cz - cx * sy * sz, -cx * cy);
    RSREG_SET3(jang[1], cx * sz + sx * sy * cz, cx * cz - sx * sy * sz, -sx * cy);
    RSREG_SET3(jang[2], -sy * cz, sy * sz, cy);
    RSREG_SET3(jang[3], sx * cy * cz, -sx * cy * sz, sx * sy);
    RSREG_SET3(jang[4], -cx * cy * cz, cx * cy * sz, -cx * sy);
    RSREG_SET3(jang[5], -cy * sz, -cy * cz, 0);
    RSREG_SET3(jang[6], cx * cz - sx * sy * sz, -cx * sz - sx * sy * cz, 0);
    RSREG_SET3(jang[7], sx * cz + cx * sy * sz, cx * sy * cz - sx * sz, 0);
    RSREG_SET3(hang[0], -cx * sz - sx * sy * cz, -cx * cz + sx * sy * sz, sx * cy);
    RSREG_SET3(hang[1], -sx * sz + cx * sy * cz, -cx * sy * sz - sx * cz, -cx * cy);
    RSREG_SET3(hang[2], cx * cy * cz, -cx * cy * sz, cx * sy);
    RSREG_SET3(hang[3], sx * cy * cz, -sx * cy * sz, sx * sy);
    RSREG_SET3(hang[4], -sx * cz - cx * sy * sz, sx * sz - cx * sy * cz, 0);
    RSREG_SET3(hang[5], cx * cz - sx * sy * sz, -sx * sy * cz - cx * sz, 0);
    RSREG_SET3(hang[6], -cy * cz, cy * sz, sy);
    RSREG_SET3(hang[7], -sx * sy * cz, sx * sy * sz, sx * cy);
    RSREG_SET3(hang[8], cx * sy * cz, -cx * sy * sz, -cx * cy);
    RSREG_SET3(hang[9], sy * sz, sy * cz, 0);
    RSREG_SET3(hang[10], -sx * cy * sz, -sx * cy * cz, 0);
    RSREG_SET3(hang[11], cx * cy * sz, cx * cy * cz, 0);
    RSREG_SET3(hang[12], -cy * cz, cy * sz, 0);
    RSREG_SET3(hang[13], -cx * sz - sx * sy * cz, -cx * cz + sx * sy * sz, 0);
    RSREG_SET3(hang[14], -sx * sz + cx * sy * cz, -cx * sy * sz - sx * cz, 0);
#undef RSREG_SET3
}

// ---- More-Thuente helpers (ndt.hpp auxilaryFunction_PsiMT / dPsiMT, updateIntervalMT, trialValueSelectionMT)
RSREG_HD inline double ndt_psi_mt(double a, double f_a, double f_0, double g_0, double mu) { return f_a - f_0 - mu * g_0 * a; }
RSREG_HD inline double ndt_dpsi_mt(double g_a, double g_0, double mu) { return g_a - mu * g_0; }

RSREG_HD inline bool ndt_update_interval(double &a_l, double &f_l, double &g_l, double &a_u, double &f_u, double &g_u, double a_t, double f_t,
                                         double g_t)
{
#pragma clang fp contract(off)
    if (f_t > f_l) {
        a_u = a_t; f_u = f_t; g_u = g_t;
        return false;
    }
    if (g_t * (a_l - a_t) > 0) {
        a_l = a_t; f_l = f_t; g_l = g_t;
        return false;
    }
    if (g_t * (a_l - a_t) < 0) {
        a_u = a_l; f_u = f_l; g_u = g_l;
        a_l = a_t; f_l = f_t; g_l = g_t;
        return false;
    }
    return true;
}

RSREG_HD inline double ndt_cubic_min(double a_1, double f_1, double g_1, double a_t, double f_t, double g_t)
{
#pragma clang fp contract(off)
    const double z = 3 * (f_t - f_1) / (a_t - a_1) - g_t - g_1;
    const double w = sqrt(z * z - g_t * g_1);
    return a_1 + (a_t - a_1) * (w - g_1 - z) / (g_t - g_1 + 2 * w);
}

RSREG_HD inline double ndt_trial_value(double a_l, double f_l, double g_l, double a_u, double f_u, double g_u, double a_t, double f_t, double g_t)
{
#pragma clang fp contract(off)
    if (f_t > f_l) {  // case 1
        const double a_c = ndt_cubic_min(a_l, f_l, g_l, a_t, f_t, g_t);
        const double a_q = a_l - 0.5 * (a_l - a_t) * g_l / (g_l - (f_l - f_t) / (a_l - a_t));
        return fabs(a_c - a_l) < fabs(a_q - a_l) ? a_c : 0.5 * (a_q + a_c);
    }
    if (g_t * g_l < 0) {  // case 2
        const double a_c = ndt_cubic_min(a_l, f_l, g_l, a_t, f_t, g_t);
        const double a_s = a_l - (a_l - a_t) / (g_l - g_t) * g_l;
        return fabs(a_c - a_t) >= fabs(a_s - a_t) ? a_c : a_s;
    }
    if (fabs(g_t) <= fabs(g_l)) {  // case 3
        const double a_c = ndt_cubic_min(a_l, f_l, g_l, a_t, f_t, g_t);
        const double a_s = a_l - (a_l - a_t) / (g_l - g_t) * g_l;
        const double a_next = fabs(a_c - a_t) < fabs(a_s - a_t) ? a_c : a_s;
        const double lim = a_t + 0.66 * (a_u - a_t);
        // std::min(lim, a_next) / std::max(lim, a_next) in their operand order (NaN operands then select what PCL selects)
        return a_t > a_l ? ((a_next < lim) ? a_next : lim) : ((lim < a_next) ? a_next : lim);
    }
    return ndt_cubic_min(a_u, f_u, g_u, a_t, f_t, g_t);  // case 4
}

// ---- computeStepLengthMT as a state machine: ndt_ls_begin, then one derivative pass (mode `next_mode` at pose `x_t`)
// per ndt_ls_consume until phase == kNdtLsDone.  The sums of a pass: [0] score, [1..6] gradient, [7..27] Hessian upper
// triangle, row-major.
enum { kNdtLsFirst = 0, kNdtLsTrial = 1, kNdtLsHessian = 2, kNdtLsDone = 3 };

struct NdtLs {
    double x[6], dir[6], x_t[6];
    double step_max, step_min;
    double phi_0, d_phi_0;
    double a_l, f_l, g_l, a_u, f_u, g_u;
    double a_t, phi_t, d_phi_t, psi_t, d_psi_t;
    double score, grad[6], hess[36];   // of the passes so far, as PCL's variables hold them (a pass zeroes the Hessian first)
    int open_interval, interval_converged, step_iterations;
    int phase, next_mode, passes;
};

RSREG_HD inline void ndt_ls_set_trial(NdtLs &s)
{
#pragma clang fp contract(off)
    for (int i = 0; i < 6; ++i) s.x_t[i] = s.x[i] + s.dir[i] * s.a_t;
}

// score / grad: of the pose x the search starts from; dir: the normalised Newton direction (flipped here if it does not
// descend, like PCL does: the caller's copy is updated)
RSREG_HD inline void ndt_ls_begin(NdtLs &s, const double *x, double *dir, double step_init, double step_max, double step_min, double score,
                                  const double *grad, const double *hess)
{
#pragma clang fp contract(off)
    s.passes = 0;
    s.score = score;
    for (int i = 0; i < 6; ++i) { s.x[i] = x[i]; s.grad[i] = grad[i]; s.x_t[i] = x[i]; }
    for (int k = 0; k < 36; ++k) s.hess[k] = hess[k];
    s.step_max = step_max;
    s.step_min = step_min;
    s.phi_0 = -score;
    double d_phi_0 = 0;
    for (int i = 0; i < 6; ++i) d_phi_0 -= grad[i] * dir[i];
    s.a_t = 0;
    s.step_iterations = 0;
    if (d_phi_0 >= 0) {
        if (d_phi_0 == 0) {   // (PCL returns a step of 0 without another pass)
            for (int i = 0; i < 6; ++i) s.dir[i] = dir[i];
            s.d_phi_0 = 0;
            s.phase = kNdtLsDone;
            return;
        }
        d_phi_0 = -d_phi_0;
        for (int i = 0; i < 6; ++i) dir[i] = -dir[i];
    }
    for (int i = 0; i < 6; ++i) s.dir[i] = dir[i];
    s.d_phi_0 = d_phi_0;
    const double mu = 1.e-4;
    s.a_l = 0; s.a_u = 0;
    s.f_l = ndt_psi_mt(s.a_l, s.phi_0, s.phi_0, d_phi_0, mu); s.g_l = ndt_dpsi_mt(d_phi_0, d_phi_0, mu);
    s.f_u = ndt_psi_mt(s.a_u, s.phi_0, s.phi_0, d_phi_0, mu); s.g_u = ndt_dpsi_mt(d_phi_0, d_phi_0, mu);
    s.interval_converged = (step_max - step_min) < 0 ? 1 : 0;
    s.open_interval = 1;
    const double a0 = (step_max < step_init) ? step_max : step_init;   // std::min(step_init, step_max)
    s.a_t = (a0 < step_min) ? step_min : a0;                           // std::max(a_t, step_min)
    ndt_ls_set_trial(s);
    s.phase = kNdtLsFirst;
    s.next_mode = 0;
}

RSREG_HD inline void ndt_ls_consume(NdtLs &s, const double *sums)
{
#pragma clang fp contract(off)
    const double mu = 1.e-4, nu = 0.9;
    const int max_step_iterations = 10;
    const int mode = s.next_mode;
    s.passes++;
    if (mode != 2) {
        s.score = sums[0];
        for (int i = 0; i < 6; ++i) s.grad[i] = sums[1 + i];
    }
    for (int k = 0; k < 36; ++k) s.hess[k] = 0.0;   // PCL zeroes the Hessian at the start of every pass
    if (mode != 1) {
        int k = 7;
        for (int a = 0; a < 6; ++a)
            for (int b = a; b < 6; ++b) {
                s.hess[a * 6 + b] = sums[k];
                s.hess[b * 6 + a] = sums[k];
                ++k;
            }
    }
    if (s.phase == kNdtLsHessian) {
        s.phase = kNdtLsDone;
        return;
    }
    // the pass of a trial step: phi, its slope, psi
    s.phi_t = -s.score;
    double d_phi_t = 0;
    for (int i = 0; i < 6; ++i) d_phi_t -= s.grad[i] * s.dir[i];
    s.d_phi_t = d_phi_t;
    s.psi_t = ndt_psi_mt(s.a_t, s.phi_t, s.phi_0, s.d_phi_0, mu);
    s.d_psi_t = ndt_dpsi_mt(s.d_phi_t, s.d_phi_0, mu);
    if (s.phase == kNdtLsTrial) {   // the tail of PCL's loop body, after the pass of the new trial
        if (s.open_interval && (s.psi_t <= 0 && s.d_psi_t >= 0)) {
            s.open_interval = 0;
            s.f_l = s.f_l + s.phi_0 - mu * s.d_phi_0 * s.a_l;
            s.g_l = s.g_l + mu * s.d_phi_0;
            s.f_u = s.f_u + s.phi_0 - mu * s.d_phi_0 * s.a_u;
            s.g_u = s.g_u + mu * s.d_phi_0;
        }
        s.interval_converged = (s.open_interval ? ndt_update_interval(s.a_l, s.f_l, s.g_l, s.a_u, s.f_u, s.g_u, s.a_t, s.psi_t, s.d_psi_t)
                                                : ndt_update_interval(s.a_l, s.f_l, s.g_l, s.a_u, s.f_u, s.g_u, s.a_t, s.phi_t, s.d_phi_t))
                                   ? 1 : 0;
        ++s.step_iterations;
    }
    // PCL's loop condition
    if (!s.interval_converged && s.step_iterations < max_step_iterations && !(s.psi_t <= 0 && s.d_phi_t <= -nu * s.d_phi_0)) {
        double a_t = s.open_interval ? ndt_trial_value(s.a_l, s.f_l, s.g_l, s.a_u, s.f_u, s.g_u, s.a_t, s.psi_t, s.d_psi_t)
                                     : ndt_trial_value(s.a_l, s.f_l, s.g_l, s.a_u, s.f_u, s.g_u, s.a_t, s.phi_t, s.d_phi_t);
        a_t = (s.step_max < a_t) ? s.step_max : a_t;   // std::min(a_t, step_max)
        a_t = (a_t < s.step_min) ? s.step_min : a_t;   // std::max(a_t, step_min)
        s.a_t = a_t;
        ndt_ls_set_trial(s);
        s.phase = kNdtLsTrial;
        s.next_mode = 1;
        return;
    }
    if (s.step_iterations) {   // the Hessian of the last trial (its pass was score + gradient only)
        s.phase = kNdtLsHessian;
        s.next_mode = 2;
        return;
    }
    s.phase = kNdtLsDone;
}

}  // namespace rsreg
