/*
 * orb_oracle_pose.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Restates Optimizer::PoseOptimization (src/Optimizer.cc:283-495) together with the parts of the vendored g2o it
 * drives: OptimizationAlgorithmLevenberg::solve (Thirdparty/g2o/g2o/core/optimization_algorithm_levenberg.cpp:59-157),
 * SparseOptimizer::optimize / computeActiveErrors / activeRobustChi2 (core/sparse_optimizer.cpp:61-114,354-419),
 * BlockSolver::buildSystem / setLambda (core/block_solver.hpp:502-589), BaseUnaryEdge::constructQuadraticForm
 * (core/base_unary_edge.hpp:45-72), RobustKernelHuber (core/robust_kernel_impl.cpp:65-91), the two OnlyPose edges
 * (types/types_six_dof_expmap.{h,cpp}:143-206,266-364) and SE3Quat (types/se3quat.h).
 *
 * Third-party arithmetic that is NOT in /root/reference: Eigen 3 (system dependency, CMakeLists.txt find_package(Eigen3
 * 3.1.0)).  Restated from its published algorithms: Quaternion(Matrix3) (Shepperd branch selection),
 * Quaternion::toRotationMatrix, quaternion * vector (two cross products), LDLT with diagonal pivoting
 * (LinearSolverDense, solvers/linear_solver_dense.h:104-111).  The reference binary is built with -O3 -march=native, so
 * Eigen's products are vectorised and contracted there: the double-precision sums here (edge order, no FMA) agree with
 * it to rounding only.  PARITY UNPINNED, as for the rest of oracle/.
 *
 * Quirk Q11 (kept): after a round, inlier edges are classified with the error vector the LAST Levenberg trial left in
 * them (src/Optimizer.cc:407-414 recomputes the error only for edges that were outliers); when that trial was rejected
 * the errors belong to the rejected pose, not to the pose that is kept.
 */
#include "orb_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct { double x, y, z, w; double t[3]; } se3q;

static void quat_from_matrix(const double m[3][3], se3q *q) /* Eigen quaternionbase_assign_impl<Matrix3> */
{
    double t = m[0][0] + m[1][1] + m[2][2];
    double c[4]; /* x y z w */
    if (t > 0.0) {
        t = sqrt(t + 1.0);
        c[3] = 0.5 * t;
        t = 0.5 / t;
        c[0] = (m[2][1] - m[1][2]) * t;
        c[1] = (m[0][2] - m[2][0]) * t;
        c[2] = (m[1][0] - m[0][1]) * t;
    } else {
        int i = 0;
        if (m[1][1] > m[0][0]) i = 1;
        if (m[2][2] > m[i][i]) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        t = sqrt(m[i][i] - m[j][j] - m[k][k] + 1.0);
        c[i] = 0.5 * t;
        t = 0.5 / t;
        c[3] = (m[k][j] - m[j][k]) * t;
        c[j] = (m[j][i] + m[i][j]) * t;
        c[k] = (m[k][i] + m[i][k]) * t;
    }
    q->x = c[0]; q->y = c[1]; q->z = c[2]; q->w = c[3];
}

static void normalize_rotation(se3q *q) /* se3quat.h:281-286 */
{
    if (q->w < 0) { q->x = -q->x; q->y = -q->y; q->z = -q->z; q->w = -q->w; }
    const double n = sqrt(q->x * q->x + q->y * q->y + q->z * q->z + q->w * q->w);
    q->x /= n; q->y /= n; q->z /= n; q->w /= n;
}

static void quat_rotate(const se3q *q, const double v[3], double out[3]) /* Eigen QuaternionBase::_transformVector */
{
    double uv[3] = {q->y * v[2] - q->z * v[1], q->z * v[0] - q->x * v[2], q->x * v[1] - q->y * v[0]};
    uv[0] += uv[0]; uv[1] += uv[1]; uv[2] += uv[2];
    const double c[3] = {q->y * uv[2] - q->z * uv[1], q->z * uv[0] - q->x * uv[2], q->x * uv[1] - q->y * uv[0]};
    for (int i = 0; i < 3; i++) out[i] = v[i] + q->w * uv[i] + c[i];
}

static void se3_map(const se3q *q, const double p[3], double out[3]) /* SE3Quat::map, se3quat.h:212-215 */
{
    double r[3];
    quat_rotate(q, p, r);
    for (int i = 0; i < 3; i++) out[i] = r[i] + q->t[i];
}

static void se3_mul(const se3q *a, const se3q *b, se3q *out) /* SE3Quat::operator*, se3quat.h:103-109 */
{
    se3q r = *a;
    double rt[3];
    quat_rotate(a, b->t, rt);
    for (int i = 0; i < 3; i++) r.t[i] += rt[i];
    r.w = a->w * b->w - a->x * b->x - a->y * b->y - a->z * b->z;
    r.x = a->w * b->x + a->x * b->w + a->y * b->z - a->z * b->y;
    r.y = a->w * b->y + a->y * b->w + a->z * b->x - a->x * b->z;
    r.z = a->w * b->z + a->z * b->w + a->x * b->y - a->y * b->x;
    normalize_rotation(&r);
    *out = r;
}

static void mat3_mul(const double a[3][3], const double b[3][3], double c[3][3])
{
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            double s = 0;
            for (int k = 0; k < 3; k++) s += a[i][k] * b[k][j];
            c[i][j] = s;
        }
}

static void se3_exp(const double u[6], se3q *out) /* SE3Quat::exp, se3quat.h:218-252 */
{
    const double om[3] = {u[0], u[1], u[2]}, up[3] = {u[3], u[4], u[5]};
    const double theta = sqrt(om[0] * om[0] + om[1] * om[1] + om[2] * om[2]);
    const double O[3][3] = {{0, -om[2], om[1]}, {om[2], 0, -om[0]}, {-om[1], om[0], 0}};
    double O2[3][3], R[3][3], V[3][3];
    mat3_mul(O, O, O2);
    if (theta < 0.00001) {
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) { R[i][j] = ((i == j) ? 1.0 : 0.0) + O[i][j] + O2[i][j]; V[i][j] = R[i][j]; }
    } else {
        const double a = sin(theta) / theta, b = (1 - cos(theta)) / (theta * theta), c = (theta - sin(theta)) / pow(theta, 3);
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) {
                const double id = (i == j) ? 1.0 : 0.0;
                R[i][j] = id + a * O[i][j] + b * O2[i][j];
                V[i][j] = id + b * O[i][j] + c * O2[i][j];
            }
    }
    quat_from_matrix(R, out);
    for (int i = 0; i < 3; i++) out->t[i] = V[i][0] * up[0] + V[i][1] * up[1] + V[i][2] * up[2];
    normalize_rotation(out); /* SE3Quat(const Quaterniond&, const Vector3d&) */
}

static void se3_from_cv(const float *T, se3q *q) /* Converter::toSE3Quat, src/Converter.cc:26-36 */
{
    double R[3][3];
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) R[i][j] = T[i * 4 + j];
        q->t[i] = T[i * 4 + 3];
    }
    quat_from_matrix(R, q);
    normalize_rotation(q);
}

static void se3_to_cv(const se3q *q, float *T) /* Converter::toCvMat(SE3Quat), src/Converter.cc:38-60 */
{
    const double tx = 2 * q->x, ty = 2 * q->y, tz = 2 * q->z;
    const double twx = tx * q->w, twy = ty * q->w, twz = tz * q->w;
    const double txx = tx * q->x, txy = ty * q->x, txz = tz * q->x, tyy = ty * q->y, tyz = tz * q->y, tzz = tz * q->z;
    const double R[3][3] = {{1 - (tyy + tzz), txy - twz, txz + twy}, {txy + twz, 1 - (txx + tzz), tyz - twx}, {txz - twy, tyz + twx, 1 - (txx + tyy)}};
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) T[i * 4 + j] = (float)R[i][j];
        T[i * 4 + 3] = (float)q->t[i];
    }
    T[12] = T[13] = T[14] = 0.0f;
    T[15] = 1.0f;
}

/* Eigen::LDLT (diagonal pivoting) of the 6x6 system, then solve.  Returns isPositive(). */
static int ldlt_solve6(const double Hin[6][6], const double *b, double *x)
{
    double A[6][6];
    int perm[6];
    memcpy(A, Hin, sizeof(A));
    int positive = 1;
    for (int k = 0; k < 6; k++) {
        int p = k;
        double big = fabs(A[k][k]);
        for (int i = k + 1; i < 6; i++)
            if (fabs(A[i][i]) > big) { big = fabs(A[i][i]); p = i; }
        perm[k] = p;
        if (p != k) { /* symmetric row / column exchange */
            for (int j = 0; j < 6; j++) { const double t = A[k][j]; A[k][j] = A[p][j]; A[p][j] = t; }
            for (int i = 0; i < 6; i++) { const double t = A[i][k]; A[i][k] = A[i][p]; A[i][p] = t; }
        }
        /* A[k][k] -= sum_j L[k][j]^2 d_j ; column below likewise (lower triangle holds L, diagonal holds D) */
        double d = A[k][k];
        for (int j = 0; j < k; j++) d -= A[k][j] * A[k][j] * A[j][j];
        A[k][k] = d;
        if (d < 0) positive = 0;
        for (int i = k + 1; i < 6; i++) {
            double s = A[i][k];
            for (int j = 0; j < k; j++) s -= A[i][j] * A[k][j] * A[j][j];
            A[i][k] = (fabs(d) > DBL_MIN) ? s / d : 0.0;
        }
    }
    if (!positive) return 0;
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
    return 1;
}

typedef struct {
    int idx;        /* keypoint index (vnIndexEdgeMono / vnIndexEdgeStereo) */
    int stereo;
    double obs[3], Xw[3], info;
    double err[3];  /* _error as the last computeError left it */
    int level;      /* 0 = optimised, 1 = outlier */
    int robust;
    double delta, dsqr;
} pose_edge;

typedef struct { double fx, fy, cx, cy, bf; } cam_d;

static void edge_compute_error(pose_edge *e, const se3q *est, const cam_d *c) /* computeError + cam_project */
{
    double p[3];
    se3_map(est, e->Xw, p);
    if (!e->stereo) { /* types_six_dof_expmap.cpp:290-296: project2d then fx, cx */
        const double u = p[0] / p[2], v = p[1] / p[2];
        e->err[0] = e->obs[0] - (u * c->fx + c->cx);
        e->err[1] = e->obs[1] - (v * c->fy + c->cy);
        e->err[2] = 0;
    } else { /* :299-306 -- invz is a float there */
        const float invz = (float)(1.0 / p[2]);
        const double r0 = p[0] * invz * c->fx + c->cx, r1 = p[1] * invz * c->fy + c->cy, r2 = r0 - c->bf * invz;
        e->err[0] = e->obs[0] - r0;
        e->err[1] = e->obs[1] - r1;
        e->err[2] = e->obs[2] - r2;
    }
}

static double edge_chi2(const pose_edge *e) /* BaseEdge::chi2, core/base_edge.h:58-61 */
{
    double s = e->err[0] * (e->info * e->err[0]) + e->err[1] * (e->info * e->err[1]);
    if (e->stereo) s += e->err[2] * (e->info * e->err[2]);
    return s;
}

static void huber(const pose_edge *e, double chi, double rho[3])
{
    if (chi <= e->dsqr) { rho[0] = chi; rho[1] = 1.0; rho[2] = 0.0; }
    else {
        const double s = sqrt(chi);
        rho[0] = 2 * s * e->delta - e->dsqr;
        rho[1] = e->delta / s;
        rho[2] = -0.5 * rho[1] / chi;
    }
}

static int g_eval_count; /* diagnostic: computeActiveErrors calls of the last orc_pose_optimization */
int orc_pose_eval_count(void) { return g_eval_count; }

static double active_errors_and_chi(pose_edge *E, int ne, const se3q *est, const cam_d *c)
{
    double chi = 0.0;
    g_eval_count++;
    for (int k = 0; k < ne; k++) {
        if (E[k].level != 0) continue;
        edge_compute_error(&E[k], est, c);
    }
    for (int k = 0; k < ne; k++) { /* activeRobustChi2 */
        if (E[k].level != 0) continue;
        const double e2 = edge_chi2(&E[k]);
        if (E[k].robust) { double rho[3]; huber(&E[k], e2, rho); chi += rho[0]; }
        else chi += e2;
    }
    return chi;
}

static void build_system(const pose_edge *E, int ne, const se3q *est, const cam_d *c, double H[6][6], double b[6])
{
    memset(H, 0, sizeof(double) * 36);
    memset(b, 0, sizeof(double) * 6);
    for (int k = 0; k < ne; k++) {
        const pose_edge *e = &E[k];
        if (e->level != 0) continue;
        double p[3], J[3][6];
        se3_map(est, e->Xw, p);
        const double x = p[0], y = p[1], invz = 1.0 / p[2], invz_2 = invz * invz;
        J[0][0] = x * y * invz_2 * c->fx;
        J[0][1] = -(1 + (x * x * invz_2)) * c->fx;
        J[0][2] = y * invz * c->fx;
        J[0][3] = -invz * c->fx;
        J[0][4] = 0;
        J[0][5] = x * invz_2 * c->fx;
        J[1][0] = (1 + y * y * invz_2) * c->fy;
        J[1][1] = -x * y * invz_2 * c->fy;
        J[1][2] = -x * invz * c->fy;
        J[1][3] = 0;
        J[1][4] = -invz * c->fy;
        J[1][5] = y * invz_2 * c->fy;
        const int D = e->stereo ? 3 : 2;
        if (e->stereo) {
            J[2][0] = J[0][0] - c->bf * y * invz_2;
            J[2][1] = J[0][1] + c->bf * x * invz_2;
            J[2][2] = J[0][2];
            J[2][3] = J[0][3];
            J[2][4] = 0;
            J[2][5] = J[0][5] - c->bf * invz_2;
        }
        double w = 1.0; /* rho[1] */
        if (e->robust) { double rho[3]; huber(e, edge_chi2(e), rho); w = rho[1]; }
        const double wi = w * e->info; /* robustInformation: rho[1] * _information */
        for (int i = 0; i < 6; i++) {
            double bi = 0;
            for (int d = 0; d < D; d++) bi += (w * J[d][i]) * e->info * e->err[d];
            b[i] -= bi;
            for (int j = 0; j < 6; j++) {
                double s = 0;
                for (int d = 0; d < D; d++) s += (J[d][i] * wi) * J[d][j];
                H[i][j] += s;
            }
        }
    }
}

/* One optimizer.optimize(10) call (core/sparse_optimizer.cpp:354-419 driving the Levenberg solver). */
static void optimize_round(pose_edge *E, int ne, se3q *est, const cam_d *c, double *x /* solver's _x, persists */)
{
    int nactive = 0;
    for (int k = 0; k < ne; k++) nactive += E[k].level == 0;
    if (nactive == 0) return; /* no active vertex: optimize() returns -1 before touching anything */
    double lambda = -1.0, ni = 2.0;
    int n_bad = 0;
    for (int it = 0; it < 10; it++) {
        double H[6][6], b[6];
        double current_chi = active_errors_and_chi(E, ne, est, c);
        const double ini_chi = current_chi;
        build_system(E, ne, est, c, H, b);
        if (it == 0) {
            double mx = 0.0;
            for (int j = 0; j < 6; j++) mx = fmax(fabs(H[j][j]), mx);
            lambda = 1e-5 * mx;
            ni = 2.0;
            n_bad = 0;
        }
        double rho = 0.0;
        int qmax = 0;
        do {
            const se3q backup = *est;
            double Hl[6][6];
            memcpy(Hl, H, sizeof(Hl));
            for (int j = 0; j < 6; j++) Hl[j][j] += lambda;
            const int ok2 = ldlt_solve6(Hl, b, x);
            se3q upd, next;
            se3_exp(x, &upd);
            se3_mul(&upd, est, &next); /* VertexSE3Expmap::oplusImpl */
            *est = next;
            double temp_chi = active_errors_and_chi(E, ne, est, c);
            if (!ok2) temp_chi = DBL_MAX;
            rho = current_chi - temp_chi;
            double scale = 0.0;
            for (int j = 0; j < 6; j++) scale += x[j] * (lambda * x[j] + b[j]);
            scale += 1e-3;
            rho /= scale;
            if (rho > 0 && isfinite(temp_chi)) {
                double alpha = 1.0 - pow(2 * rho - 1, 3);
                alpha = fmin(alpha, 2.0 / 3.0);
                const double sf = fmax(1.0 / 3.0, alpha);
                lambda *= sf;
                ni = 2;
                current_chi = temp_chi;
            } else {
                lambda *= ni;
                ni *= 2;
                *est = backup;
            }
            qmax++;
        } while (rho < 0 && qmax < 10);
        if (qmax == 10 || rho == 0) return;
        if ((ini_chi - current_chi) * 1e3 < ini_chi) n_bad++;
        else n_bad = 0;
        if (n_bad >= 3) return;
    }
}

/* Optimizer::PoseOptimization (src/Optimizer.cc:283-495).  Tcw: 4x4 row-major float, in/out (pFrame->mTcw / SetPose).
 * has_point[i] != 0 <=> pFrame->mvpMapPoints[i] != NULL, Xw its world position.  outlier[i] (pFrame->mvbOutlier) is
 * written for those entries only.  Returns nInitialCorrespondences - nBad (0 and nothing written to Tcw when fewer
 * than 3 correspondences, :404-405). */
int orc_pose_optimization(float *Tcw, int N, const orc_keypoint *keys_un, const float *u_right, const uint8_t *has_point,
                          const float *Xw, const float *inv_level_sigma2, float fx, float fy, float cx, float cy, float bf,
                          uint8_t *outlier)
{
    pose_edge *E = (pose_edge *)calloc((size_t)(N > 0 ? N : 1), sizeof(pose_edge));
    g_eval_count = 0;
    const float delta_mono = (float)sqrt(5.991), delta_stereo = (float)sqrt(7.815);
    int ne = 0;
    for (int i = 0; i < N; i++) {
        if (!has_point[i]) continue;
        pose_edge *e = &E[ne++];
        e->idx = i;
        e->stereo = !(u_right[i] < 0);
        outlier[i] = 0;
        e->obs[0] = keys_un[i].x; e->obs[1] = keys_un[i].y; e->obs[2] = e->stereo ? u_right[i] : 0.0;
        e->info = inv_level_sigma2[keys_un[i].octave];
        e->robust = 1;
        e->delta = e->stereo ? delta_stereo : delta_mono;
        e->dsqr = e->delta * e->delta;
        for (int k = 0; k < 3; k++) e->Xw[k] = Xw[(size_t)i * 3 + k];
        e->level = 0;
    }
    if (ne < 3) { free(E); return 0; }
    const cam_d cam = {fx, fy, cx, cy, bf};
    const float chi2_mono = 5.991f, chi2_stereo = 7.815f;
    se3q est;
    double x[6] = {0, 0, 0, 0, 0, 0};
    int n_bad = 0;
    for (int it = 0; it < 4; it++) {
        se3_from_cv(Tcw, &est);
        optimize_round(E, ne, &est, &cam, x);
        n_bad = 0;
        /* the reference walks the mono edges, then the stereo edges; the two walks touch disjoint state */
        for (int k = 0; k < ne; k++) {
            pose_edge *e = &E[k];
            if (outlier[e->idx]) edge_compute_error(e, &est, &cam);
            const float chi2 = (float)edge_chi2(e);
            if (chi2 > (e->stereo ? chi2_stereo : chi2_mono)) { outlier[e->idx] = 1; e->level = 1; n_bad++; }
            else { outlier[e->idx] = 0; e->level = 0; }
            if (it == 2) e->robust = 0;
        }
        if (ne < 10) break;
    }
    se3_to_cv(&est, Tcw);
    free(E);
    return ne - n_bad;
}


/* ---- test hooks (tests/test_pose.py): the Eigen / g2o building blocks restated above, one at a time ---- */
void orc_test_quat_roundtrip(const double *R9, double *q4 /* x y z w, normalised as SE3Quat does */, double *Rout9)
{
    double m[3][3];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) m[i][j] = R9[i * 3 + j];
    se3q q;
    q.t[0] = q.t[1] = q.t[2] = 0;
    quat_from_matrix(m, &q);
    normalize_rotation(&q);
    q4[0] = q.x; q4[1] = q.y; q4[2] = q.z; q4[3] = q.w;
    float T[16];
    se3_to_cv(&q, T); /* float output of Converter::toCvMat; enough for a 1e-6 round trip */
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Rout9[i * 3 + j] = T[i * 4 + j];
}

void orc_test_se3_exp(const double *u6, double *T12 /* 3x4 row major */)
{
    se3q q;
    se3_exp(u6, &q);
    const double tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z;
    const double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
    const double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x, tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
    const double R[3][3] = {{1 - (tyy + tzz), txy - twz, txz + twy}, {txy + twz, 1 - (txx + tzz), tyz - twx}, {txz - twy, tyz + twx, 1 - (txx + tyy)}};
    for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) T12[i * 4 + j] = R[i][j]; T12[i * 4 + 3] = q.t[i]; }
}

int orc_test_ldlt6(const double *H36, const double *b6, double *x6)
{
    double H[6][6];
    for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) H[i][j] = H36[i * 6 + j];
    return ldlt_solve6(H, b6, x6);
}
