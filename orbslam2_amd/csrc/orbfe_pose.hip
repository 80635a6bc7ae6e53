// orbfe_pose.hip -- Optimizer::PoseOptimization (src/Optimizer.cc:283-495) on the device: the motion-only bundle
// adjustment that follows every Tracking matcher call (src/Tracking.cc:875,998,1040,1475,1555,1580).
//
// One workgroup per frame (problem).  The reference drives g2o's Levenberg solver over one 6-dof vertex and <= N unary
// edges; everything per-edge (projection, Huber weight, J^T W J, J^T W e, robust chi2) is data parallel, everything per
// trial step (6x6 LDLT with diagonal pivoting, SE3 exponential, acceptance test) is a few hundred flops that every
// lane repeats redundantly so that no broadcast is needed.  One evaluation pass produces the robust chi2 AND the
// normal equations at the trial pose: when the trial is accepted they are exactly what the next iteration's
// computeActiveErrors + buildSystem would produce (same pose, same edges), so an iteration costs one pass and one
// barrier.  All arithmetic is FP64 like g2o's; the block reduction has a fixed order (bit-reproducible run to run),
// which differs from the reference's edge-sequential sums, so the result agrees with the CPU to rounding
// (tests/test_pose.py states the tolerance).
//
// g2o call map:  eval_pass            = SparseOptimizer::computeActiveErrors + activeRobustChi2
//                                       (core/sparse_optimizer.cpp:61-114) + BlockSolver::buildSystem
//                                       (core/block_solver.hpp:502-560) with BaseUnaryEdge::constructQuadraticForm
//                                       (core/base_unary_edge.hpp:45-72) and the analytic Jacobians
//                                       (types/types_six_dof_expmap.cpp:266-288,335-364)
//                solve_ldlt6          = LinearSolverDense::solve (solvers/linear_solver_dense.h:65-112, Eigen::LDLT)
//                se3_exp / se3_mul    = SE3Quat::exp, operator* (types/se3quat.h:103-109,218-252)
//                the do/while         = OptimizationAlgorithmLevenberg::solve
//                                       (core/optimization_algorithm_levenberg.cpp:59-157)
#include <hip/hip_runtime.h>
#include <cfloat>
#include <cstdint>
#include <cstring>

#include "orbfe_device.h"
#include "orbfe_host.h"

namespace {

constexpr int PO_THREADS = 256;
constexpr int PO_WAVES = PO_THREADS / 64;
constexpr int PO_NV = 29; // 21 (upper triangle of H) + 6 (b) + chi + active count

struct Se3 { double x, y, z, w, t[3]; };
struct CamD { double fx, fy, cx, cy, bf; };

__device__ inline void quat_from_matrix(const double m[3][3], Se3 &q)
{
    double t = m[0][0] + m[1][1] + m[2][2];
    if (t > 0.0) {
        t = sqrt(t + 1.0);
        q.w = 0.5 * t;
        t = 0.5 / t;
        q.x = (m[2][1] - m[1][2]) * t;
        q.y = (m[0][2] - m[2][0]) * t;
        q.z = (m[1][0] - m[0][1]) * t;
    } else if (m[0][0] >= m[1][1] && m[0][0] >= m[2][2]) { // i = 0 (the reference's strict '>' tests keep the lower index on ties)
        t = sqrt(m[0][0] - m[1][1] - m[2][2] + 1.0);
        q.x = 0.5 * t;
        t = 0.5 / t;
        q.w = (m[2][1] - m[1][2]) * t;
        q.y = (m[1][0] + m[0][1]) * t;
        q.z = (m[2][0] + m[0][2]) * t;
    } else if (m[1][1] >= m[2][2]) { // i = 1
        t = sqrt(m[1][1] - m[2][2] - m[0][0] + 1.0);
        q.y = 0.5 * t;
        t = 0.5 / t;
        q.w = (m[0][2] - m[2][0]) * t;
        q.z = (m[2][1] + m[1][2]) * t;
        q.x = (m[0][1] + m[1][0]) * t;
    } else { // i = 2
        t = sqrt(m[2][2] - m[0][0] - m[1][1] + 1.0);
        q.z = 0.5 * t;
        t = 0.5 / t;
        q.w = (m[1][0] - m[0][1]) * t;
        q.x = (m[0][2] + m[2][0]) * t;
        q.y = (m[1][2] + m[2][1]) * t;
    }
}

__device__ inline void normalize_rotation(Se3 &q)
{
    if (q.w < 0) { q.x = -q.x; q.y = -q.y; q.z = -q.z; q.w = -q.w; }
    const double n = sqrt(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
    q.x /= n; q.y /= n; q.z /= n; q.w /= n;
}

__device__ inline void quat_rotate(const Se3 &q, const double v[3], double out[3])
{
    double uv[3] = {q.y * v[2] - q.z * v[1], q.z * v[0] - q.x * v[2], q.x * v[1] - q.y * v[0]};
    uv[0] += uv[0]; uv[1] += uv[1]; uv[2] += uv[2];
    const double c[3] = {q.y * uv[2] - q.z * uv[1], q.z * uv[0] - q.x * uv[2], q.x * uv[1] - q.y * uv[0]};
#pragma unroll
    for (int i = 0; i < 3; i++) out[i] = v[i] + q.w * uv[i] + c[i];
}

__device__ inline void se3_map(const Se3 &q, const double p[3], double out[3])
{
    double r[3];
    quat_rotate(q, p, r);
#pragma unroll
    for (int i = 0; i < 3; i++) out[i] = r[i] + q.t[i];
}

__device__ inline Se3 se3_mul(const Se3 &a, const Se3 &b)
{
    Se3 r = a;
    double rt[3];
    quat_rotate(a, b.t, rt);
#pragma unroll
    for (int i = 0; i < 3; i++) r.t[i] += rt[i];
    r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
    r.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
    r.y = a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z;
    r.z = a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x;
    normalize_rotation(r);
    return r;
}

__device__ inline Se3 se3_exp(const double u[6])
{
    const double om[3] = {u[0], u[1], u[2]};
    const double theta = sqrt(om[0] * om[0] + om[1] * om[1] + om[2] * om[2]);
    const double O[3][3] = {{0, -om[2], om[1]}, {om[2], 0, -om[0]}, {-om[1], om[0], 0}};
    double O2[3][3], R[3][3], V[3][3];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) {
            double s = 0;
#pragma unroll
            for (int k = 0; k < 3; k++) s += O[i][k] * O[k][j];
            O2[i][j] = s;
        }
    double a = 1.0, b = 1.0, c = 1.0; // theta < 1e-5: R = V = I + Omega + Omega^2 (se3quat.h:232-238)
    bool same = true;
    if (!(theta < 0.00001)) {
        a = sin(theta) / theta;
        b = (1 - cos(theta)) / (theta * theta);
        c = (theta - sin(theta)) / pow(theta, 3.0);
        same = false;
    }
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const double id = (i == j) ? 1.0 : 0.0;
            if (same) { R[i][j] = id + O[i][j] + O2[i][j]; V[i][j] = R[i][j]; }
            else { R[i][j] = id + a * O[i][j] + b * O2[i][j]; V[i][j] = id + b * O[i][j] + c * O2[i][j]; }
        }
    Se3 q;
    quat_from_matrix(R, q);
#pragma unroll
    for (int i = 0; i < 3; i++) q.t[i] = V[i][0] * u[3] + V[i][1] * u[4] + V[i][2] * u[5];
    normalize_rotation(q);
    return q;
}

__device__ inline Se3 se3_from_cv(const float *T) // Converter::toSE3Quat, src/Converter.cc:26-36
{
    double R[3][3];
    Se3 q;
#pragma unroll
    for (int i = 0; i < 3; i++) {
#pragma unroll
        for (int j = 0; j < 3; j++) R[i][j] = (double)T[i * 4 + j];
        q.t[i] = (double)T[i * 4 + 3];
    }
    quat_from_matrix(R, q);
    normalize_rotation(q);
    return q;
}

__device__ inline void se3_to_cv(const Se3 &q, float *T) // Converter::toCvMat(SE3Quat), src/Converter.cc:38-60
{
    const double tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z;
    const double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
    const double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x, tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
    T[0] = (float)(1 - (tyy + tzz)); T[1] = (float)(txy - twz); T[2] = (float)(txz + twy); T[3] = (float)q.t[0];
    T[4] = (float)(txy + twz); T[5] = (float)(1 - (txx + tzz)); T[6] = (float)(tyz - twx); T[7] = (float)q.t[1];
    T[8] = (float)(txz - twy); T[9] = (float)(tyz + twx); T[10] = (float)(1 - (txx + tyy)); T[11] = (float)q.t[2];
    T[12] = 0.f; T[13] = 0.f; T[14] = 0.f; T[15] = 1.f;
}

// LDLT with diagonal pivoting; returns isPositive().  x is left untouched when the factor is not positive.
__device__ inline bool solve_ldlt6(const double *Hu /*21, upper triangle row-major*/, double lambda, const double *b, double *x)
{
    double A[6][6];
    {
        int k = 0;
#pragma unroll
        for (int i = 0; i < 6; i++)
#pragma unroll
            for (int j = i; j < 6; j++) { A[i][j] = Hu[k]; A[j][i] = Hu[k]; k++; }
#pragma unroll
        for (int i = 0; i < 6; i++) A[i][i] += lambda;
    }
    int perm[6];
    bool positive = true;
    for (int k = 0; k < 6; k++) {
        int p = k;
        double big = fabs(A[k][k]);
        for (int i = k + 1; i < 6; i++)
            if (fabs(A[i][i]) > big) { big = fabs(A[i][i]); p = i; }
        perm[k] = p;
        if (p != k) {
            for (int j = 0; j < 6; j++) { const double t = A[k][j]; A[k][j] = A[p][j]; A[p][j] = t; }
            for (int i = 0; i < 6; i++) { const double t = A[i][k]; A[i][k] = A[i][p]; A[i][p] = t; }
        }
        double d = A[k][k];
        for (int j = 0; j < k; j++) d -= A[k][j] * A[k][j] * A[j][j];
        A[k][k] = d;
        if (d < 0) positive = false;
        for (int i = k + 1; i < 6; i++) {
            double s = A[i][k];
            for (int j = 0; j < k; j++) s -= A[i][j] * A[k][j] * A[j][j];
            A[i][k] = (fabs(d) > DBL_MIN) ? s / d : 0.0;
        }
    }
    if (!positive) return false;
    double y[6];
    for (int i = 0; i < 6; i++) y[i] = b[i];
    for (int k = 0; k < 6; k++) if (perm[k] != k) { const double t = y[k]; y[k] = y[perm[k]]; y[perm[k]] = t; }
    for (int i = 0; i < 6; i++)
        for (int j = 0; j < i; j++) y[i] -= A[i][j] * y[j];
    for (int i = 0; i < 6; i++) y[i] = (fabs(A[i][i]) > DBL_MIN) ? y[i] / A[i][i] : 0.0;
    for (int i = 5; i >= 0; i--)
        for (int j = i + 1; j < 6; j++) y[i] -= A[j][i] * y[j];
    for (int k = 5; k >= 0; k--) if (perm[k] != k) { const double t = y[k]; y[k] = y[perm[k]]; y[perm[k]] = t; }
    for (int i = 0; i < 6; i++) x[i] = y[i];
    return true;
}

struct Problem {
    const KeyPointPOD *keys;
    const float *u_right;
    const uint8_t *has_point;
    const float *Xw;
    uint8_t *outlier;
    int n;
};

// error vector of one edge at pose q (computeError of the two OnlyPose edges); returns chi2 (BaseEdge::chi2)
__device__ inline double edge_error(const Se3 &q, const CamD &c, const double Xw[3], const double obs[3], bool stereo, double info,
                                    double err[3], double p[3])
{
    se3_map(q, Xw, p);
    if (!stereo) {
        const double u = p[0] / p[2], v = p[1] / p[2];
        err[0] = obs[0] - (u * c.fx + c.cx);
        err[1] = obs[1] - (v * c.fy + c.cy);
        err[2] = 0.0;
        return err[0] * (info * err[0]) + err[1] * (info * err[1]);
    }
    const double invz = (double)(float)(1.0 / p[2]); // `const float invz = 1.0f/trans_xyz[2]`, types_six_dof_expmap.cpp:300
    const double r0 = p[0] * invz * c.fx + c.cx, r1 = p[1] * invz * c.fy + c.cy, r2 = r0 - c.bf * invz;
    err[0] = obs[0] - r0;
    err[1] = obs[1] - r1;
    err[2] = obs[2] - r2;
    return err[0] * (info * err[0]) + err[1] * (info * err[1]) + err[2] * (info * err[2]);
}

// One pass over the active edges at pose q: sums[0..20] = H (upper), [21..26] = b, [27] = robust chi2, [28] = #active.
// Every lane returns the same totals.
__device__ void eval_pass(const Problem &P, const Se3 &q, const CamD &c, const float *inv_sigma2, bool robust,
                          double delta_mono, double delta_stereo, double (*part)[PO_NV], double *sums)
{
    double acc[PO_NV];
#pragma unroll
    for (int k = 0; k < PO_NV; k++) acc[k] = 0.0;
    for (int i = threadIdx.x; i < P.n; i += PO_THREADS) {
        if (!P.has_point[i] || P.outlier[i]) continue;
        const KeyPointPOD kp = P.keys[i];
        const float ur = P.u_right[i];
        const bool stereo = !(ur < 0);
        const double info = (double)inv_sigma2[kp.octave];
        const double Xw[3] = {(double)P.Xw[3 * (size_t)i], (double)P.Xw[3 * (size_t)i + 1], (double)P.Xw[3 * (size_t)i + 2]};
        const double obs[3] = {(double)kp.x, (double)kp.y, stereo ? (double)ur : 0.0};
        double err[3], p[3];
        const double chi = edge_error(q, c, Xw, obs, stereo, info, err, p);
        double w = 1.0, rho0 = chi;
        if (robust) { // RobustKernelHuber::robustify, core/robust_kernel_impl.cpp:78-91
            const double delta = stereo ? delta_stereo : delta_mono, dsqr = delta * delta;
            if (chi > dsqr) {
                const double s = sqrt(chi);
                rho0 = 2 * s * delta - dsqr;
                w = delta / s;
            }
        }
        acc[27] += rho0;
        acc[28] += 1.0;
        const double x = p[0], y = p[1], invz = 1.0 / p[2], invz_2 = invz * invz;
        double J[3][6];
        J[0][0] = x * y * invz_2 * c.fx;
        J[0][1] = -(1 + (x * x * invz_2)) * c.fx;
        J[0][2] = y * invz * c.fx;
        J[0][3] = -invz * c.fx;
        J[0][4] = 0;
        J[0][5] = x * invz_2 * c.fx;
        J[1][0] = (1 + y * y * invz_2) * c.fy;
        J[1][1] = -x * y * invz_2 * c.fy;
        J[1][2] = -x * invz * c.fy;
        J[1][3] = 0;
        J[1][4] = -invz * c.fy;
        J[1][5] = y * invz_2 * c.fy;
        J[2][0] = stereo ? J[0][0] - c.bf * y * invz_2 : 0.0;
        J[2][1] = stereo ? J[0][1] + c.bf * x * invz_2 : 0.0;
        J[2][2] = stereo ? J[0][2] : 0.0;
        J[2][3] = stereo ? J[0][3] : 0.0;
        J[2][4] = 0.0;
        J[2][5] = stereo ? J[0][5] - c.bf * invz_2 : 0.0;
        const double wi = w * info; // robustInformation: rho[1] * _information (core/base_edge.h:96-102)
        int k = 0;
#pragma unroll
        for (int a = 0; a < 6; a++) {
            acc[21 + a] -= (w * J[0][a]) * info * err[0] + (w * J[1][a]) * info * err[1] + (w * J[2][a]) * info * err[2];
#pragma unroll
            for (int b = a; b < 6; b++) {
                acc[k] += (J[0][a] * wi) * J[0][b] + (J[1][a] * wi) * J[1][b] + (J[2][a] * wi) * J[2][b];
                k++;
            }
        }
    }
    // wave reduction (fixed butterfly), then the four wave partials are added in wave order by every lane
#pragma unroll
    for (int k = 0; k < PO_NV; k++) {
        double v = acc[k];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
        acc[k] = v;
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane < PO_NV) {
        double v = 0.0;
#pragma unroll
        for (int k = 0; k < PO_NV; k++) if (lane == k) v = acc[k];
        part[wave][lane] = v;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < PO_NV; k++) {
        double v = part[0][k];
#pragma unroll
        for (int wv = 1; wv < PO_WAVES; wv++) v += part[wv][k];
        sums[k] = v;
    }
}

__global__ __launch_bounds__(PO_THREADS) void pose_opt_kernel(const int32_t *__restrict__ offsets, const KeyPointPOD *__restrict__ keys,
                                                               const float *__restrict__ u_right, const uint8_t *__restrict__ has_point,
                                                               const float *__restrict__ Xw, float *__restrict__ Tcw,
                                                               uint8_t *__restrict__ outlier, int32_t *__restrict__ n_inliers,
                                                               const float *__restrict__ inv_sigma2, float fx, float fy, float cx,
                                                               float cy, float bf)
{
    __shared__ double part[2][PO_WAVES][PO_NV]; // double-buffered: one barrier per pass
    __shared__ int s_cnt[2];
    const int prob = blockIdx.x;
    const int o0 = offsets[prob];
    Problem P;
    P.n = offsets[prob + 1] - o0;
    P.keys = keys + o0; P.u_right = u_right + o0; P.has_point = has_point + o0; P.Xw = Xw + (size_t)3 * o0; P.outlier = outlier + o0;
    float *T = Tcw + (size_t)16 * prob;
    const CamD cam = {(double)fx, (double)fy, (double)cx, (double)cy, (double)bf};
    const double delta_mono = (double)(float)sqrt(5.991), delta_stereo = (double)(float)sqrt(7.815); // src/Optimizer.cc:317-318
    const float chi2_mono = 5.991f, chi2_stereo = 7.815f;                                               // :408-409

    // edges start as inliers (:331,362)
    for (int i = threadIdx.x; i < P.n; i += PO_THREADS)
        if (P.has_point[i]) P.outlier[i] = 0;
    // (each lane only ever reads the flags it wrote: the index -> lane mapping is fixed)

    float Tin[16];
#pragma unroll
    for (int k = 0; k < 16; k++) Tin[k] = T[k];

    int buf = 0;
    double S[PO_NV];
    double x[6] = {0, 0, 0, 0, 0, 0};
    Se3 est = se3_from_cv(Tin), last_eval = est;
    int ne = 0, n_bad = 0;
    bool robust = true;
    for (int round = 0; round < 4; round++) {
        est = se3_from_cv(Tin); // :398
        eval_pass(P, est, cam, inv_sigma2, robust, delta_mono, delta_stereo, part[buf], S);
        buf ^= 1;
        if (round == 0) {
            ne = (int)S[28];
            if (ne < 3) { // :404-405
                if (threadIdx.x == 0) n_inliers[prob] = 0;
                return;
            }
        }
        if (S[28] > 0.0) { // otherwise optimize() returns before doing anything (no active vertex)
            last_eval = est;
            double H[21], b[6], current_chi = S[27];
#pragma unroll
            for (int k = 0; k < 21; k++) H[k] = S[k];
#pragma unroll
            for (int k = 0; k < 6; k++) b[k] = S[21 + k];
            double lambda = -1.0, ni = 2.0;
            int lm_bad = 0;
            for (int it = 0; it < 10; it++) {
                last_eval = est; // computeActiveErrors at the current estimate
                const double ini_chi = current_chi;
                if (it == 0) {
                    const double dg[6] = {H[0], H[6], H[11], H[15], H[18], H[20]};
                    double mx = 0.0;
#pragma unroll
                    for (int j = 0; j < 6; j++) mx = fmax(fabs(dg[j]), mx);
                    lambda = 1e-5 * mx;
                    ni = 2.0;
                    lm_bad = 0;
                }
                double rho = 0.0;
                int qmax = 0;
                do {
                    const bool ok2 = solve_ldlt6(H, lambda, b, x);
                    const Se3 trial = se3_mul(se3_exp(x), est);
                    eval_pass(P, trial, cam, inv_sigma2, robust, delta_mono, delta_stereo, part[buf], S);
                    buf ^= 1;
                    last_eval = trial;
                    const double temp_chi = ok2 ? S[27] : DBL_MAX;
                    rho = current_chi - temp_chi;
                    double scale = 0.0;
#pragma unroll
                    for (int j = 0; j < 6; j++) scale += x[j] * (lambda * x[j] + b[j]);
                    scale += 1e-3;
                    rho /= scale;
                    if (rho > 0 && isfinite(temp_chi)) {
                        double alpha = 1.0 - pow(2 * rho - 1, 3.0);
                        alpha = fmin(alpha, 2.0 / 3.0);
                        lambda *= fmax(1.0 / 3.0, alpha);
                        ni = 2;
                        current_chi = temp_chi;
                        est = trial;
#pragma unroll
                        for (int k = 0; k < 21; k++) H[k] = S[k];
#pragma unroll
                        for (int k = 0; k < 6; k++) b[k] = S[21 + k];
                    } else {
                        lambda *= ni;
                        ni *= 2;
                    }
                    qmax++;
                } while (rho < 0 && qmax < 10);
                if (qmax == 10 || rho == 0) break;
                if ((ini_chi - current_chi) * 1e3 < ini_chi) lm_bad++;
                else lm_bad = 0;
                if (lm_bad >= 3) break;
            }
        }
        // classification (:401-455); inliers keep the error of the last evaluated pose (Q11), outliers are recomputed
        int bad = 0;
        for (int i = threadIdx.x; i < P.n; i += PO_THREADS) {
            if (!P.has_point[i]) continue;
            const KeyPointPOD kp = P.keys[i];
            const float ur = P.u_right[i];
            const bool stereo = !(ur < 0);
            const double info = (double)inv_sigma2[kp.octave];
            const double Xw3[3] = {(double)P.Xw[3 * (size_t)i], (double)P.Xw[3 * (size_t)i + 1], (double)P.Xw[3 * (size_t)i + 2]};
            const double obs[3] = {(double)kp.x, (double)kp.y, stereo ? (double)ur : 0.0};
            double err[3], p[3];
            const float chi2 = (float)edge_error(P.outlier[i] ? est : last_eval, cam, Xw3, obs, stereo, info, err, p);
            const bool out = chi2 > (stereo ? chi2_stereo : chi2_mono);
            P.outlier[i] = out ? 1 : 0;
            bad += out;
        }
        if (threadIdx.x == 0) s_cnt[round & 1] = 0;
        __syncthreads();
        for (int off = 32; off >= 1; off >>= 1) bad += __shfl_xor(bad, off, 64);
        if ((threadIdx.x & 63) == 0) atomicAdd(&s_cnt[round & 1], bad);
        __syncthreads();
        n_bad = s_cnt[round & 1];
        if (round == 2) robust = false; // :429-430
        if (ne < 10) break;             // :457-458
    }
    if (threadIdx.x == 0) {
        float Tout[16];
        se3_to_cv(est, Tout);
#pragma unroll
        for (int k = 0; k < 16; k++) T[k] = Tout[k];
        n_inliers[prob] = ne - n_bad;
    }
}

struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    int ensure(size_t need)
    {
        if (need <= bytes) return 0;
        if (p) (void)hipFree(p);
        p = nullptr; bytes = 0;
        if (hipMalloc(&p, need) != hipSuccess) return -1;
        bytes = need;
        return 0;
    }
    ~DevBuf() { if (p) (void)hipFree(p); }
};

} // namespace

struct orbfe_pose_state {
    DevBuf off, keys, ur, has, xw, T, out, ninl, sig;
};

orbfe_pose_state *orbfe_pose_state_create() { return new orbfe_pose_state(); }
void orbfe_pose_state_destroy(orbfe_pose_state *s) { delete s; }

#define PTRY(ctx, expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return orbfe_fail(ctx, ORBFE_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); } while (0)

extern "C" int orbfe_enqueue_pose_optimization(orbfe_context *ctx, int n_problems, const int32_t *d_offsets,
                                               const orbfe_keypoint *d_keys_un, const float *d_u_right, const uint8_t *d_has_point,
                                               const float *d_Xw, float *d_Tcw, uint8_t *d_outlier, int32_t *d_n_inliers, void *stream)
{
    if (!ctx || n_problems < 0) return orbfe_fail(ctx, ORBFE_ERR_INVALID, "bad argument");
    if (n_problems == 0) return ORBFE_OK;
    if (!d_offsets || !d_keys_un || !d_u_right || !d_has_point || !d_Xw || !d_Tcw || !d_outlier || !d_n_inliers)
        return orbfe_fail(ctx, ORBFE_ERR_INVALID, "null device pointer");
    orbfe_pose_state *st = orbfe_ctx_pose_state(ctx);
    hipStream_t s = stream ? (hipStream_t)stream : orbfe_ctx_stream(ctx);
    const orbfe_params *p = orbfe_ctx_params(ctx);
    if (!st->sig.p) { // mvInvLevelSigma2 of the context's pyramid
        if (st->sig.ensure(sizeof(float) * ORBFE_MAX_LEVELS)) return orbfe_fail(ctx, ORBFE_ERR_HIP, "pose scratch allocation failed");
        PTRY(ctx, hipMemcpy(st->sig.p, orbfe_ctx_inv_sigma2(ctx), sizeof(float) * p->nlevels, hipMemcpyHostToDevice));
    }
    hipLaunchKernelGGL(pose_opt_kernel, dim3(n_problems), dim3(PO_THREADS), 0, s, d_offsets, (const KeyPointPOD *)d_keys_un, d_u_right,
                       d_has_point, d_Xw, d_Tcw, d_outlier, d_n_inliers, (const float *)st->sig.p, p->fx, p->fy, p->cx, p->cy, p->bf);
    PTRY(ctx, hipGetLastError());
    return ORBFE_OK;
}

extern "C" int orbfe_pose_optimization_batch(orbfe_context *ctx, int n_problems, const int32_t *offsets, float *Tcw,
                                             const orbfe_keypoint *keys_un, const float *u_right, const uint8_t *has_point,
                                             const float *Xw, uint8_t *outlier, int32_t *n_inliers)
{
    if (!ctx || n_problems < 0) return orbfe_fail(ctx, ORBFE_ERR_INVALID, "bad argument");
    if (n_problems == 0) return ORBFE_OK;
    if (!offsets || !Tcw || !outlier || !n_inliers) return orbfe_fail(ctx, ORBFE_ERR_INVALID, "null argument");
    const int total = offsets[n_problems];
    const orbfe_params *p = orbfe_ctx_params(ctx);
    for (int k = 0; k < n_problems; k++)
        if (offsets[k + 1] < offsets[k] || offsets[0] != 0) return orbfe_fail(ctx, ORBFE_ERR_INVALID, "offsets must start at 0 and not decrease");
    if (total > 0 && (!keys_un || !u_right || !has_point || !Xw)) return orbfe_fail(ctx, ORBFE_ERR_INVALID, "null argument");
    for (int i = 0; i < total; i++)
        if (has_point[i] && (keys_un[i].octave < 0 || keys_un[i].octave >= p->nlevels))
            return orbfe_fail(ctx, ORBFE_ERR_INVALID, "keypoint %d has octave %d outside the context's %d levels", i, keys_un[i].octave, p->nlevels);
    orbfe_pose_state *st = orbfe_ctx_pose_state(ctx);
    hipStream_t s = orbfe_ctx_stream(ctx);
    PTRY(ctx, hipSetDevice(orbfe_ctx_device(ctx)));
    const size_t tn = (size_t)(total > 0 ? total : 1);
    if (st->off.ensure(sizeof(int32_t) * (n_problems + 1)) || st->keys.ensure(sizeof(KeyPointPOD) * tn) || st->ur.ensure(sizeof(float) * tn) ||
        st->has.ensure(tn) || st->xw.ensure(sizeof(float) * 3 * tn) || st->T.ensure(sizeof(float) * 16 * n_problems) || st->out.ensure(tn) ||
        st->ninl.ensure(sizeof(int32_t) * n_problems))
        return orbfe_fail(ctx, ORBFE_ERR_HIP, "pose scratch allocation failed");
    PTRY(ctx, hipMemcpyAsync(st->off.p, offsets, sizeof(int32_t) * (n_problems + 1), hipMemcpyHostToDevice, s));
    PTRY(ctx, hipMemcpyAsync(st->T.p, Tcw, sizeof(float) * 16 * n_problems, hipMemcpyHostToDevice, s));
    if (total > 0) {
        PTRY(ctx, hipMemcpyAsync(st->keys.p, keys_un, sizeof(KeyPointPOD) * tn, hipMemcpyHostToDevice, s));
        PTRY(ctx, hipMemcpyAsync(st->ur.p, u_right, sizeof(float) * tn, hipMemcpyHostToDevice, s));
        PTRY(ctx, hipMemcpyAsync(st->has.p, has_point, tn, hipMemcpyHostToDevice, s));
        PTRY(ctx, hipMemcpyAsync(st->xw.p, Xw, sizeof(float) * 3 * tn, hipMemcpyHostToDevice, s));
        PTRY(ctx, hipMemcpyAsync(st->out.p, outlier, tn, hipMemcpyHostToDevice, s)); // entries without a point keep the caller's value
    }
    int rc = orbfe_enqueue_pose_optimization(ctx, n_problems, (const int32_t *)st->off.p, (const orbfe_keypoint *)st->keys.p, (const float *)st->ur.p,
                                             (const uint8_t *)st->has.p, (const float *)st->xw.p, (float *)st->T.p, (uint8_t *)st->out.p,
                                             (int32_t *)st->ninl.p, nullptr);
    if (rc != ORBFE_OK) return rc;
    // problems with fewer than 3 correspondences leave their pose untouched on the device (the reference returns
    // before SetPose, src/Optimizer.cc:404-405)
    PTRY(ctx, hipMemcpyAsync(Tcw, st->T.p, sizeof(float) * 16 * n_problems, hipMemcpyDeviceToHost, s));
    PTRY(ctx, hipMemcpyAsync(n_inliers, st->ninl.p, sizeof(int32_t) * n_problems, hipMemcpyDeviceToHost, s));
    if (total > 0) PTRY(ctx, hipMemcpyAsync(outlier, st->out.p, tn, hipMemcpyDeviceToHost, s));
    PTRY(ctx, hipStreamSynchronize(s));
    return ORBFE_OK;
}

extern "C" int orbfe_pose_optimization(orbfe_context *ctx, float *Tcw, int n, const orbfe_keypoint *keys_un, const float *u_right,
                                       const uint8_t *has_point, const float *Xw, uint8_t *outlier, int *n_inliers)
{
    if (!n_inliers || n < 0) return orbfe_fail(ctx, ORBFE_ERR_INVALID, "bad argument");
    const int32_t off[2] = {0, n};
    int32_t ninl = 0;
    const int rc = orbfe_pose_optimization_batch(ctx, 1, off, Tcw, keys_un, u_right, has_point, Xw, outlier, &ninl);
    *n_inliers = ninl;
    return rc;
}
