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
#include <algorithm>
#include <cfloat>
#include <cstdint>
#include <cstring>

#include "orbfe_device.h"
#include "orbfe_host.h"

// Contract Q4 (no FMA) governs the bit-exact integer / float stages; this stage is FP64 and compared with a tolerance, so
// fused multiply-adds are allowed here: half the instructions in the per-edge products, one rounding less each.
#pragma clang fp contract(fast)

namespace {

constexpr int PO_THREADS = 256; // one wave per SIMD (512 threads measured slower: every wave repeats the per-trial serial part)

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

// Symmetric exchange of rows / columns K and C of a register-resident 6x6 matrix (compile-time indices only).
template <int K, int C>
__device__ __forceinline__ void sym_swap(double (&A)[6][6], double (&y)[6])
{
#pragma unroll
    for (int j = 0; j < 6; j++) { const double t = A[K][j]; A[K][j] = A[C][j]; A[C][j] = t; }
#pragma unroll
    for (int i = 0; i < 6; i++) { const double t = A[i][K]; A[i][K] = A[i][C]; A[i][C] = t; }
    const double t = y[K]; y[K] = y[C]; y[C] = t;
}

// (Eigen divides the sub-column and the solve's D^-1 step by the pivot; here the pivot's reciprocal is formed once and
// multiplied in -- 6 divisions per solve instead of 21, a last-bit difference that the tolerance of this stage covers.)
template <int K>
__device__ __forceinline__ void ldlt_step(double (&A)[6][6], double (&y)[6], double (&dinv)[6], int (&perm)[6], bool &positive)
{
    // largest remaining diagonal entry, first one wins (Eigen: maxCoeff over the tail of the diagonal)
    int p = K;
    double big = fabs(A[K][K]);
#pragma unroll
    for (int i = K + 1; i < 6; i++)
        if (fabs(A[i][i]) > big) { big = fabs(A[i][i]); p = i; }
    p = __builtin_amdgcn_readfirstlane(p); // the matrix is the same in every lane: a scalar branch, no dynamic register indexing
    perm[K] = p;
    // the right-hand side is permuted along (P b), which is what the forward substitution consumes
    if constexpr (K < 5) {
        switch (p) {
        case 1: if constexpr (K < 1) sym_swap<K, 1>(A, y); break;
        case 2: if constexpr (K < 2) sym_swap<K, 2>(A, y); break;
        case 3: if constexpr (K < 3) sym_swap<K, 3>(A, y); break;
        case 4: if constexpr (K < 4) sym_swap<K, 4>(A, y); break;
        case 5: sym_swap<K, 5>(A, y); break;
        default: break;
        }
    }
    double d = A[K][K];
#pragma unroll
    for (int j = 0; j < K; j++) d -= A[K][j] * A[K][j] * A[j][j];
    A[K][K] = d;
    if (d < 0) positive = false;
    const double di = (fabs(d) > DBL_MIN) ? 1.0 / d : 0.0;
    dinv[K] = di;
#pragma unroll
    for (int i = K + 1; i < 6; i++) {
        double sacc = A[i][K];
#pragma unroll
        for (int j = 0; j < K; j++) sacc -= A[i][j] * A[K][j] * A[j][j];
        A[i][K] = sacc * di;
    }
}

template <int K>
__device__ __forceinline__ void unpermute_step(double (&y)[6], const int (&perm)[6])
{
    if constexpr (K < 5) {
        double t;
        switch (perm[K]) { // scalar (readfirstlane'd above)
        case 1: if constexpr (K < 1) { t = y[K]; y[K] = y[1]; y[1] = t; } break;
        case 2: if constexpr (K < 2) { t = y[K]; y[K] = y[2]; y[2] = t; } break;
        case 3: if constexpr (K < 3) { t = y[K]; y[K] = y[3]; y[3] = t; } break;
        case 4: if constexpr (K < 4) { t = y[K]; y[K] = y[4]; y[4] = t; } break;
        case 5: t = y[K]; y[K] = y[5]; y[5] = t; break;
        default: break;
        }
    }
}

// LDLT with diagonal pivoting (Eigen::LDLT as LinearSolverDense uses it); returns isPositive().  x is left untouched
// when the factor is not positive.  Everything is indexed at compile time so the matrix stays in registers.
__device__ inline bool solve_ldlt6(const double *Hu /*21, upper triangle row-major, then b[6]*/, double lambda, double *x)
{
    double A[6][6], y[6];
    {
        int k = 0;
#pragma unroll
        for (int i = 0; i < 6; i++)
#pragma unroll
            for (int j = i; j < 6; j++) { A[i][j] = Hu[k]; A[j][i] = Hu[k]; k++; }
#pragma unroll
        for (int i = 0; i < 6; i++) { A[i][i] += lambda; y[i] = Hu[21 + i]; }
    }
    int perm[6] = {0, 1, 2, 3, 4, 5};
    double dinv[6];
    bool positive = true;
    ldlt_step<0>(A, y, dinv, perm, positive);
    ldlt_step<1>(A, y, dinv, perm, positive);
    ldlt_step<2>(A, y, dinv, perm, positive);
    ldlt_step<3>(A, y, dinv, perm, positive);
    ldlt_step<4>(A, y, dinv, perm, positive);
    ldlt_step<5>(A, y, dinv, perm, positive);
    if (!positive) return false;
#pragma unroll
    for (int i = 0; i < 6; i++)
#pragma unroll
        for (int j = 0; j < i; j++) y[i] -= A[i][j] * y[j];
#pragma unroll
    for (int i = 0; i < 6; i++) y[i] *= dinv[i];
#pragma unroll
    for (int i = 5; i >= 0; i--)
#pragma unroll
        for (int j = i + 1; j < 6; j++) y[i] -= A[j][i] * y[j];
    unpermute_step<4>(y, perm);
    unpermute_step<3>(y, perm);
    unpermute_step<2>(y, perm);
    unpermute_step<1>(y, perm);
    unpermute_step<0>(y, perm);
#pragma unroll
    for (int i = 0; i < 6; i++) x[i] = y[i];
    return true;
}

// Per-problem edge table.  EDGES_IN_LDS: seven floats and a state byte per keypoint slot staged once (Xw, observation,
// information); otherwise the caller's arrays are re-read on every pass (frames with more slots than the LDS holds).
// state: 0 = no map point, 1 = inlier (level 0), 2 = outlier (level 1).
struct EdgeTable {
    const KeyPointPOD *keys;
    const float *u_right;
    const uint8_t *has_point;
    const float *Xw;
    uint8_t *outlier;
    const float *inv_sigma2;
    float *l_f;      // [7][cap]
    uint8_t *l_st;   // [cap]
    int n, cap;
};

template <bool IN_LDS>
__device__ __forceinline__ int edge_state(const EdgeTable &E, int i)
{
    if constexpr (IN_LDS) return E.l_st[i];
    else return E.has_point[i] ? (E.outlier[i] ? 2 : 1) : 0;
}

template <bool IN_LDS>
__device__ __forceinline__ void edge_set_outlier(const EdgeTable &E, int i, bool out)
{
    if constexpr (IN_LDS) E.l_st[i] = out ? 2 : 1;
    else E.outlier[i] = out ? 1 : 0;
}

template <bool IN_LDS>
__device__ __forceinline__ void edge_load(const EdgeTable &E, int i, double Xw[3], double obs[3], bool &stereo, double &info)
{
    float f[7];
    if constexpr (IN_LDS) {
#pragma unroll
        for (int k = 0; k < 7; k++) f[k] = E.l_f[k * E.cap + i];
    } else {
        const KeyPointPOD kp = E.keys[i];
        f[0] = E.Xw[3 * (size_t)i]; f[1] = E.Xw[3 * (size_t)i + 1]; f[2] = E.Xw[3 * (size_t)i + 2];
        f[3] = kp.x; f[4] = kp.y; f[5] = E.u_right[i]; f[6] = E.inv_sigma2[kp.octave];
    }
    stereo = !(f[5] < 0);
    Xw[0] = (double)f[0]; Xw[1] = (double)f[1]; Xw[2] = (double)f[2];
    obs[0] = (double)f[3]; obs[1] = (double)f[4]; obs[2] = stereo ? (double)f[5] : 0.0;
    info = (double)f[6];
}

// error vector of one edge at pose q (computeError of the two OnlyPose edges); returns chi2 (BaseEdge::chi2)
__device__ __forceinline__ double edge_error(const Se3 &q, const CamD &c, const double Xw[3], const double obs[3], bool stereo, double info,
                                             double err[3], double p[3], double *invz_out = nullptr)
{
    se3_map(q, Xw, p);
    // mono: project2d then fx, cx (types_six_dof_expmap.cpp:290-296); stereo: `const float invz = 1.0f/trans_xyz[2]` (:299-306).
    // One division per edge: the mono path multiplies by 1/z where the reference divides by z (last-bit difference).
    const double invz = 1.0 / p[2];
    const double iz = stereo ? (double)(float)invz : invz;
    const double r0 = p[0] * iz * c.fx + c.cx;
    const double r1 = p[1] * iz * c.fy + c.cy;
    const double invz_s = iz;
    err[0] = obs[0] - r0;
    err[1] = obs[1] - r1;
    err[2] = stereo ? obs[2] - (r0 - c.bf * invz_s) : 0.0;
    double chi = err[0] * (info * err[0]) + err[1] * (info * err[1]);
    if (stereo) chi += err[2] * (info * err[2]);
    if (invz_out) *invz_out = invz;
    return chi;
}

// lane i of each 16-lane row exchanges with lane i ^ m (m = 8: row_mirror then half mirror ... see below)
__device__ __forceinline__ double dpp_f64(double v, const int ctrl_sel)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    switch (ctrl_sel) {
    case 0: lo = __builtin_amdgcn_update_dpp(0, lo, 0x140, 0xf, 0xf, true); hi = __builtin_amdgcn_update_dpp(0, hi, 0x140, 0xf, 0xf, true); break; // row_mirror: i <-> 15 - i
    case 1: lo = __builtin_amdgcn_update_dpp(0, lo, 0x141, 0xf, 0xf, true); hi = __builtin_amdgcn_update_dpp(0, hi, 0x141, 0xf, 0xf, true); break; // row_half_mirror: i <-> 7 - i
    case 2: lo = __builtin_amdgcn_update_dpp(0, lo, 0x4E, 0xf, 0xf, true); hi = __builtin_amdgcn_update_dpp(0, hi, 0x4E, 0xf, 0xf, true); break;   // quad_perm [2,3,0,1]
    default: lo = __builtin_amdgcn_update_dpp(0, lo, 0xB1, 0xf, 0xf, true); hi = __builtin_amdgcn_update_dpp(0, hi, 0xB1, 0xf, 0xf, true); break;  // quad_perm [1,0,3,2]
    }
    return __hiloint2double(hi, lo);
}

// Transposing reduction inside each 16-lane row: at every step a lane keeps one value of each pair (chosen by one bit of
// its lane id), sends the other to its partner and adds what it receives.  32 values in, 2 out per lane, each the sum of
// that value over the 16 lanes: 16 + 8 + 4 + 2 exchanges instead of 4 per value.  Fixed order: bit-reproducible.
template <int NIN, int STEP>
__device__ __forceinline__ void row_transpose_step(double *v, bool upper)
{
#pragma unroll
    for (int k = 0; k < NIN / 2; k++) {
        const double keep = upper ? v[2 * k + 1] : v[2 * k], give = upper ? v[2 * k] : v[2 * k + 1];
        v[k] = keep + dpp_f64(give, STEP);
    }
}

// One pass over the active edges at pose q: s_tot[0..20] = H (upper), [21..26] = b, [27] = robust chi2, [28] = #active,
// left in LDS (the 6x6 solve reads them from there: they are the same for every lane and would cost 58 registers);
// chi2 and the count are returned.  Costs two barriers.
// sum over the three residual rows of (J^T W)[a][d] * J[d][b], skipping the rows whose entry in column a or b is
// structurally zero (row 0 and row 2: column 4; row 1: column 3)
template <int A, int B>
__device__ __forceinline__ double h_term(const double (&jw)[3][6], const double (&J)[3][6])
{
    constexpr bool r02 = (A != 4 && B != 4), r1 = (A != 3 && B != 3);
    if constexpr (r02 && r1) return jw[0][A] * J[0][B] + jw[1][A] * J[1][B] + jw[2][A] * J[2][B];
    else if constexpr (r02) return jw[0][A] * J[0][B] + jw[2][A] * J[2][B];
    else if constexpr (r1) return jw[1][A] * J[1][B];
    else return 0.0;
}

#ifdef ORBFE_POSE_TIMING
__device__ long long g_pose_cycles[8]; // [0] solve [1] exp+mul [2] edge loop [3] reduction [4] passes
#define PO_T(var) const long long var = clock64()
#define PO_ACC(slot, a, b) do { if (threadIdx.x == 0 && blockIdx.x == 0) g_pose_cycles[slot] += (b) - (a); } while (0)
#else
#define PO_T(var)
#define PO_ACC(slot, a, b)
#endif

template <int THREADS, bool IN_LDS>
__device__ void eval_pass(const EdgeTable &E, const Se3 &q, const CamD &c, bool robust, double delta_mono, double delta_stereo,
                          double *s_rows /*[THREADS/16][32]*/, double *s_tot /*[32]*/, double &chi_out, double &cnt_out)
{
    double acc[32];
    PO_T(t_e0);
#pragma unroll
    for (int k = 0; k < 32; k++) acc[k] = 0.0;
    for (int i = threadIdx.x; i < E.n; i += THREADS) {
        if (edge_state<IN_LDS>(E, i) != 1) continue;
        double Xw[3], obs[3], info, err[3], p[3];
        bool stereo;
        edge_load<IN_LDS>(E, i, Xw, obs, stereo, info);
        double invz;
        const double chi = edge_error(q, c, Xw, obs, stereo, info, err, p, &invz);
        double w = 1.0, rho0 = chi;
        if (robust) { // RobustKernelHuber::robustify, core/robust_kernel_impl.cpp:78-91
            const double delta = stereo ? delta_stereo : delta_mono, dsqr = delta * delta;
            if (chi > dsqr) {
                // 1/sqrt(chi): hardware estimate + two Newton steps (relative error below 1e-15) instead of an IEEE sqrt
                // followed by an IEEE division
                double rs = __builtin_amdgcn_rsq(chi);
                rs = rs * (1.5 - 0.5 * chi * rs * rs);
                rs = rs * (1.5 - 0.5 * chi * rs * rs);
                rho0 = 2 * (chi * rs) * delta - dsqr;
                w = delta * rs;
            }
        }
        acc[27] += rho0;
        acc[28] += 1.0;
        const double x = p[0], y = p[1], invz_2 = invz * invz;
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
        const double we[3] = {wi * err[0], wi * err[1], wi * err[2]};
        // J[0][4], J[1][3] and J[2][4] are structurally zero: their products are left out (they would add exact zeros)
        double jw[3][6];
#pragma unroll
        for (int a = 0; a < 6; a++) { jw[0][a] = J[0][a] * wi; jw[1][a] = J[1][a] * wi; jw[2][a] = J[2][a] * wi; }
#define PO_H(k, a, b) acc[k] += h_term<a, b>(jw, J)
        PO_H(0, 0, 0); PO_H(1, 0, 1); PO_H(2, 0, 2); PO_H(3, 0, 3); PO_H(4, 0, 4); PO_H(5, 0, 5);
        PO_H(6, 1, 1); PO_H(7, 1, 2); PO_H(8, 1, 3); PO_H(9, 1, 4); PO_H(10, 1, 5);
        PO_H(11, 2, 2); PO_H(12, 2, 3); PO_H(13, 2, 4); PO_H(14, 2, 5);
        PO_H(15, 3, 3); PO_H(16, 3, 4); PO_H(17, 3, 5);
        PO_H(18, 4, 4); PO_H(19, 4, 5);
        PO_H(20, 5, 5);
#undef PO_H
        acc[21] -= J[0][0] * we[0] + J[1][0] * we[1] + J[2][0] * we[2];
        acc[22] -= J[0][1] * we[0] + J[1][1] * we[1] + J[2][1] * we[2];
        acc[23] -= J[0][2] * we[0] + J[1][2] * we[1] + J[2][2] * we[2];
        acc[24] -= J[0][3] * we[0] + J[2][3] * we[2];
        acc[25] -= J[1][4] * we[1];
        acc[26] -= J[0][5] * we[0] + J[1][5] * we[1] + J[2][5] * we[2];
    }
    PO_T(t_e1);
    PO_ACC(2, t_e0, t_e1);
    const int lane16 = threadIdx.x & 15;
    row_transpose_step<32, 0>(acc, (lane16 & 8) != 0);
    row_transpose_step<16, 1>(acc, (lane16 & 4) != 0);
    row_transpose_step<8, 2>(acc, (lane16 & 2) != 0);
    row_transpose_step<4, 3>(acc, (lane16 & 1) != 0);
    // acc[0], acc[1] now hold the row sums of values v0 and v0 + 1 with v0 = 2 * bitreverse4(lane16)... computed below
    {
        const int b3 = (lane16 >> 3) & 1, b2 = (lane16 >> 2) & 1, b1 = (lane16 >> 1) & 1, b0 = lane16 & 1;
        // step 0 kept index 2k+b3 of 32 -> k; step 1 kept 2k+b2 of 16; step 2 kept 2k+b1 of 8; step 3 kept 2k+b0 of 4 -> 2 left (k = 0, 1)
        // original index of the value now at position k: (((k * 2 + b0) * 2 + b1) * 2 + b2) * 2 + b3
        const int row = threadIdx.x >> 4;
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const int orig = (((k * 2 + b0) * 2 + b1) * 2 + b2) * 2 + b3;
            s_rows[row * 32 + orig] = acc[k];
        }
    }
    __syncthreads();
    if (threadIdx.x < 32) {
        double t = 0.0;
#pragma unroll 8
        for (int r = 0; r < THREADS / 16; r++) t += s_rows[r * 32 + threadIdx.x];
        s_tot[threadIdx.x] = t;
    }
    __syncthreads();
    chi_out = s_tot[27];
    cnt_out = s_tot[28];
    PO_T(t_e2);
    PO_ACC(3, t_e1, t_e2);
    PO_ACC(4, 0, 1);
}

template <int THREADS, bool IN_LDS>
__global__ __launch_bounds__(THREADS) void pose_opt_kernel(const int32_t *__restrict__ offsets, const KeyPointPOD *__restrict__ keys,
                                                            const float *__restrict__ u_right, const uint8_t *__restrict__ has_point,
                                                            const float *__restrict__ Xw, float *__restrict__ Tcw,
                                                            uint8_t *__restrict__ outlier, int32_t *__restrict__ n_inliers,
                                                            const float *__restrict__ inv_sigma2, float fx, float fy, float cx,
                                                            float cy, float bf, int lds_cap)
{
    extern __shared__ double s_dyn[];
    double *s_rows = s_dyn;                         // [2][THREADS/16][32]  (double-buffered across passes)
    double *s_tot = s_rows + 2 * (THREADS / 16) * 32; // [2][32]
    __shared__ int s_cnt[2];
    const int prob = blockIdx.x;
    const int o0 = offsets[prob];
    EdgeTable E;
    E.n = offsets[prob + 1] - o0;
    E.keys = keys + o0; E.u_right = u_right + o0; E.has_point = has_point + o0; E.Xw = Xw + (size_t)3 * o0; E.outlier = outlier + o0;
    E.inv_sigma2 = inv_sigma2;
    E.cap = lds_cap;
    E.l_f = (float *)(s_tot + 2 * 32);
    E.l_st = (uint8_t *)(E.l_f + 7 * (size_t)lds_cap);
    float *T = Tcw + (size_t)16 * prob;
    const CamD cam = {(double)fx, (double)fy, (double)cx, (double)cy, (double)bf};
    const double delta_mono = (double)(float)sqrt(5.991), delta_stereo = (double)(float)sqrt(7.815); // src/Optimizer.cc:317-318
    const float chi2_mono = 5.991f, chi2_stereo = 7.815f;                                               // :408-409

    // edges start as inliers (:331,362)
    for (int i = threadIdx.x; i < E.n; i += THREADS) {
        const bool has = E.has_point[i] != 0;
        if constexpr (IN_LDS) {
            const KeyPointPOD kp = E.keys[i];
            E.l_f[0 * E.cap + i] = E.Xw[3 * (size_t)i];
            E.l_f[1 * E.cap + i] = E.Xw[3 * (size_t)i + 1];
            E.l_f[2 * E.cap + i] = E.Xw[3 * (size_t)i + 2];
            E.l_f[3 * E.cap + i] = kp.x;
            E.l_f[4 * E.cap + i] = kp.y;
            E.l_f[5 * E.cap + i] = E.u_right[i];
            E.l_f[6 * E.cap + i] = has ? inv_sigma2[kp.octave] : 0.f;
            E.l_st[i] = has ? 1 : 0;
        } else if (has) E.outlier[i] = 0;
    }
    // (each lane only ever touches the slots i = lane + k * THREADS: no barrier needed for the table itself)

    float Tin[16];
#pragma unroll
    for (int k = 0; k < 16; k++) Tin[k] = T[k];

    // s_tot[cur] holds the normal equations of the accepted estimate, s_tot[cur ^ 1] receives the trial's; s_rows toggles
    // every pass so that a fast wave's next pass cannot overwrite partials a slow wave still sums
    int rbuf = 0, cur = 0;
    double chi_s, cnt_s;
    double x[6] = {0, 0, 0, 0, 0, 0};
    Se3 est = se3_from_cv(Tin), last_eval = est;
    int ne = 0, n_bad = 0;
    bool robust = true;
#define PO_EVAL(pose, which)                                                                                                               \
    do {                                                                                                                                   \
        eval_pass<THREADS, IN_LDS>(E, pose, cam, robust, delta_mono, delta_stereo, s_rows + rbuf * (THREADS / 16) * 32, s_tot + (which) * 32, \
                                   chi_s, cnt_s);                                                                                          \
        rbuf ^= 1;                                                                                                                         \
    } while (0)
    for (int round = 0; round < 4; round++) {
        est = se3_from_cv(Tin); // :398
        PO_EVAL(est, cur);
        if (round == 0) {
            ne = (int)cnt_s;
            if (ne < 3) break; // :404-405
        }
        if (cnt_s > 0.0) { // otherwise optimize() returns before doing anything (no active vertex)
            last_eval = est;
            double current_chi = chi_s;
            double lambda = -1.0, ni = 2.0;
            int lm_bad = 0;
            for (int it = 0; it < 10; it++) {
                last_eval = est; // computeActiveErrors at the current estimate
                const double ini_chi = current_chi;
                const double *Hb = s_tot + cur * 32;
                if (it == 0) {
                    const double dg[6] = {Hb[0], Hb[6], Hb[11], Hb[15], Hb[18], Hb[20]};
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
                    PO_T(t_s0);
                    const bool ok2 = solve_ldlt6(Hb, lambda, x);
                    PO_T(t_s1);
                    PO_ACC(0, t_s0, t_s1);
                    double scale = 0.0; // computeScale(), with the b of the system that produced x
#pragma unroll
                    for (int j = 0; j < 6; j++) scale += x[j] * (lambda * x[j] + Hb[21 + j]);
                    scale += 1e-3;
                    const Se3 trial = se3_mul(se3_exp(x), est);
                    PO_T(t_s2);
                    PO_ACC(1, t_s1, t_s2);
                    PO_EVAL(trial, cur ^ 1);
                    last_eval = trial;
                    const double temp_chi = ok2 ? chi_s : DBL_MAX;
                    rho = (current_chi - temp_chi) / scale;
                    if (rho > 0 && isfinite(temp_chi)) {
                        const double r21 = 2 * rho - 1;
                        double alpha = 1.0 - r21 * r21 * r21; // pow(2*rho-1, 3)
                        alpha = fmin(alpha, 2.0 / 3.0);
                        lambda *= fmax(1.0 / 3.0, alpha);
                        ni = 2;
                        current_chi = temp_chi;
                        est = trial;
                        cur ^= 1;
                        Hb = s_tot + cur * 32;
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
        for (int i = threadIdx.x; i < E.n; i += THREADS) {
            const int st = edge_state<IN_LDS>(E, i);
            if (st == 0) continue;
            double Xw3[3], obs[3], info, err[3], p[3];
            bool stereo;
            edge_load<IN_LDS>(E, i, Xw3, obs, stereo, info);
            const float chi2 = (float)edge_error(st == 2 ? est : last_eval, cam, Xw3, obs, stereo, info, err, p);
            const bool out = chi2 > (stereo ? chi2_stereo : chi2_mono);
            edge_set_outlier<IN_LDS>(E, i, out);
            bad += out;
        }
        if (threadIdx.x == 0) s_cnt[round & 1] = 0;
        __syncthreads();
        for (int off = 32; off >= 1; off >>= 1) bad += __shfl_xor(bad, off, 64);
        if ((threadIdx.x & 63) == 0 && bad) atomicAdd(&s_cnt[round & 1], bad);
        __syncthreads();
        n_bad = s_cnt[round & 1];
        if (round == 2) robust = false; // :429-430
        if (ne < 10) break;             // :457-458
    }
#undef PO_EVAL
    if constexpr (IN_LDS) { // pFrame->mvbOutlier: written where a map point exists (cleared at edge creation even when ne < 3)
        for (int i = threadIdx.x; i < E.n; i += THREADS) {
            const int st = E.l_st[i];
            if (st) E.outlier[i] = st == 2 ? 1 : 0;
        }
    }
    if (threadIdx.x == 0) {
        if (ne >= 3) {
            float Tout[16];
            se3_to_cv(est, Tout);
#pragma unroll
            for (int k = 0; k < 16; k++) T[k] = Tout[k];
            n_inliers[prob] = ne - n_bad;
        } else n_inliers[prob] = 0;
    }
}

constexpr int PO_LDS_CAP_MAX = 4096; // keypoint slots per problem the LDS edge table holds (29 B each)

template <int THREADS>
size_t pose_lds_bytes(int cap) { return sizeof(double) * (2 * (THREADS / 16) * 32 + 2 * 32) + (size_t)cap * (7 * sizeof(float) + 1) + 16; }

} // namespace

struct orbfe_pose_state {
    DevBuf blk, sig;          // one device block for all arrays of a host call
    uint8_t *h_blk = nullptr; // its pinned host image: one copy up, one copy down
    size_t h_bytes = 0;
};

orbfe_pose_state *orbfe_pose_state_create() { return new orbfe_pose_state(); }
void orbfe_pose_state_destroy(orbfe_pose_state *s)
{
    if (s && s->h_blk) (void)hipHostFree(s->h_blk);
    delete s;
}

#define PTRY(ctx, expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return orbfe_fail(ctx, ORBFE_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); } while (0)

extern "C" int orbfe_enqueue_pose_optimization(orbfe_context *ctx, int n_problems, const int32_t *d_offsets,
                                               const orbfe_keypoint *d_keys_un, const float *d_u_right, const uint8_t *d_has_point,
                                               const float *d_Xw, float *d_Tcw, uint8_t *d_outlier, int32_t *d_n_inliers,
                                               int max_keypoints, void *stream)
try {
    ORBFE_ENTRY(ctx);
    if (!ctx || n_problems < 0) return orbfe_fail(ctx, ORBFE_ERR_INVALID, "bad argument");
    if (n_problems == 0) return ORBFE_OK;
    if (!d_offsets || !d_keys_un || !d_u_right || !d_has_point || !d_Xw || !d_Tcw || !d_outlier || !d_n_inliers)
        return orbfe_fail(ctx, ORBFE_ERR_INVALID, "null device pointer");
    orbfe_pose_state *st = orbfe_ctx_pose_state(ctx);
    hipStream_t s = stream ? (hipStream_t)stream : orbfe_ctx_stream(ctx);
    const orbfe_params *p = orbfe_ctx_params(ctx);
    PTRY(ctx, hipSetDevice(orbfe_ctx_device(ctx)));
    if (!st->sig.p) { // mvInvLevelSigma2 of the context's pyramid
        if (st->sig.ensure(sizeof(float) * ORBFE_MAX_LEVELS)) return orbfe_fail(ctx, ORBFE_ERR_HIP, "pose scratch allocation failed");
        PTRY(ctx, hipMemcpy(st->sig.p, orbfe_ctx_inv_sigma2(ctx), sizeof(float) * p->nlevels, hipMemcpyHostToDevice));
    }
    constexpr int TH = PO_THREADS;
    if (max_keypoints <= PO_LDS_CAP_MAX) {
        const int cap = (std::max(max_keypoints, 1) + 3) & ~3;
        const size_t lds = pose_lds_bytes<TH>(cap);
        static bool attr_set[64] = {}; // the attribute is per device
        const int dev = orbfe_ctx_device(ctx);
        if (dev >= 0 && dev < 64 && !attr_set[dev]) {
            PTRY(ctx, hipFuncSetAttribute((const void *)pose_opt_kernel<TH, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pose_lds_bytes<TH>(PO_LDS_CAP_MAX)));
            attr_set[dev] = true;
        }
        hipLaunchKernelGGL((pose_opt_kernel<TH, true>), dim3(n_problems), dim3(TH), lds, s, d_offsets, (const KeyPointPOD *)d_keys_un, d_u_right,
                           d_has_point, d_Xw, d_Tcw, d_outlier, d_n_inliers, (const float *)st->sig.p, p->fx, p->fy, p->cx, p->cy, p->bf, cap);
    } else {
        hipLaunchKernelGGL((pose_opt_kernel<TH, false>), dim3(n_problems), dim3(TH), pose_lds_bytes<TH>(0), s, d_offsets, (const KeyPointPOD *)d_keys_un,
                           d_u_right, d_has_point, d_Xw, d_Tcw, d_outlier, d_n_inliers, (const float *)st->sig.p, p->fx, p->fy, p->cx, p->cy, p->bf, 0);
    }
    PTRY(ctx, hipGetLastError());
    return ORBFE_OK;
} ORBFE_CATCH(ctx)

#ifdef ORBFE_POSE_TIMING
extern "C" int orbfe_pose_debug_cycles(long long *dst, int reset)
try {
    if (hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_pose_cycles), sizeof(long long) * 8) != hipSuccess) return -1;
    if (reset) { long long z[8] = {}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_pose_cycles), z, sizeof(z)) != hipSuccess) return -1; }
    return 0;
} ORBFE_CATCH(nullptr)
#endif

extern "C" int orbfe_pose_optimization_batch(orbfe_context *ctx, int n_problems, const int32_t *offsets, float *Tcw,
                                             const orbfe_keypoint *keys_un, const float *u_right, const uint8_t *has_point,
                                             const float *Xw, uint8_t *outlier, int32_t *n_inliers)
try {
    ORBFE_ENTRY(ctx);
    if (!ctx || n_problems < 0) return orbfe_fail(ctx, ORBFE_ERR_INVALID, "bad argument");
    if (n_problems == 0) return ORBFE_OK;
    if (!offsets || !Tcw || !outlier || !n_inliers) return orbfe_fail(ctx, ORBFE_ERR_INVALID, "null argument");
    const int total = offsets[n_problems];
    const orbfe_params *p = orbfe_ctx_params(ctx);
    int max_n = 0;
    for (int k = 0; k < n_problems; k++) {
        if (offsets[k + 1] < offsets[k] || offsets[0] != 0) return orbfe_fail(ctx, ORBFE_ERR_INVALID, "offsets must start at 0 and not decrease");
        max_n = std::max(max_n, offsets[k + 1] - offsets[k]);
    }
    if (total > 0 && (!keys_un || !u_right || !has_point || !Xw)) return orbfe_fail(ctx, ORBFE_ERR_INVALID, "null argument");
    for (int i = 0; i < total; i++)
        if (has_point[i] && (keys_un[i].octave < 0 || keys_un[i].octave >= p->nlevels))
            return orbfe_fail(ctx, ORBFE_ERR_INVALID, "keypoint %d has octave %d outside the context's %d levels", i, keys_un[i].octave, p->nlevels);
    orbfe_pose_state *st = orbfe_ctx_pose_state(ctx);
    hipStream_t s = orbfe_ctx_stream(ctx);
    PTRY(ctx, hipSetDevice(orbfe_ctx_device(ctx)));
    const size_t tn = (size_t)(total > 0 ? total : 1);
    // block layout: results first ([Tcw | n_inliers | outlier], copied back in one piece), then the inputs
    auto up16 = [](size_t v) { return (v + 15) & ~(size_t)15; };
    const size_t o_T = 0, o_n = up16(o_T + sizeof(float) * 16 * n_problems), o_out = up16(o_n + sizeof(int32_t) * n_problems);
    const size_t o_off = up16(o_out + tn), o_keys = up16(o_off + sizeof(int32_t) * (n_problems + 1));
    const size_t o_ur = up16(o_keys + sizeof(KeyPointPOD) * tn), o_has = up16(o_ur + sizeof(float) * tn), o_xw = up16(o_has + tn);
    const size_t bytes = up16(o_xw + sizeof(float) * 3 * tn), down = o_off;
    if (st->blk.ensure(bytes)) return orbfe_fail(ctx, ORBFE_ERR_HIP, "pose scratch allocation failed");
    if (st->h_bytes < bytes) {
        if (st->h_blk) (void)hipHostFree(st->h_blk);
        st->h_blk = nullptr; st->h_bytes = 0;
        PTRY(ctx, hipHostMalloc((void **)&st->h_blk, bytes, hipHostMallocDefault));
        st->h_bytes = bytes;
    }
    uint8_t *hb = st->h_blk, *db = (uint8_t *)st->blk.p;
    memcpy(hb + o_T, Tcw, sizeof(float) * 16 * n_problems);
    memcpy(hb + o_off, offsets, sizeof(int32_t) * (n_problems + 1));
    if (total > 0) {
        memcpy(hb + o_out, outlier, tn); // entries without a point keep the caller's value
        memcpy(hb + o_keys, keys_un, sizeof(KeyPointPOD) * tn);
        memcpy(hb + o_ur, u_right, sizeof(float) * tn);
        memcpy(hb + o_has, has_point, tn);
        memcpy(hb + o_xw, Xw, sizeof(float) * 3 * tn);
    }
    PTRY(ctx, hipMemcpyAsync(db, hb, bytes, hipMemcpyHostToDevice, s));
    int rc = orbfe_enqueue_pose_optimization(ctx, n_problems, (const int32_t *)(db + o_off), (const orbfe_keypoint *)(db + o_keys), (const float *)(db + o_ur),
                                             (const uint8_t *)(db + o_has), (const float *)(db + o_xw), (float *)(db + o_T), db + o_out,
                                             (int32_t *)(db + o_n), max_n, nullptr);
    if (rc != ORBFE_OK) return rc;
    PTRY(ctx, hipMemcpyAsync(hb, db, down, hipMemcpyDeviceToHost, s));
    PTRY(ctx, hipStreamSynchronize(s));
    // problems with fewer than 3 correspondences leave their pose untouched on the device (the reference returns
    // before SetPose, src/Optimizer.cc:404-405)
    memcpy(Tcw, hb + o_T, sizeof(float) * 16 * n_problems);
    memcpy(n_inliers, hb + o_n, sizeof(int32_t) * n_problems);
    if (total > 0) memcpy(outlier, hb + o_out, tn);
    return ORBFE_OK;
} ORBFE_CATCH(ctx)

extern "C" int orbfe_pose_optimization(orbfe_context *ctx, float *Tcw, int n, const orbfe_keypoint *keys_un, const float *u_right,
                                       const uint8_t *has_point, const float *Xw, uint8_t *outlier, int *n_inliers)
try {
    ORBFE_ENTRY(ctx);
    if (!n_inliers || n < 0) return orbfe_fail(ctx, ORBFE_ERR_INVALID, "bad argument");
    const int32_t off[2] = {0, n};
    int32_t ninl = 0;
    const int rc = orbfe_pose_optimization_batch(ctx, 1, off, Tcw, keys_un, u_right, has_point, Xw, outlier, &ninl);
    *n_inliers = ninl;
    return rc;
} ORBFE_CATCH(ctx)
