/*
 * oracle_posit.c — CPU restatement of CSolverStereoPosit::getTransformationWORLDtoLEFT.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under svi_mapper_amd/ may include, link or call this file;
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the checker.
 *
 * PARITY UNPINNED: no fixtures in the reference, Eigen absent.  Follows
 *   src/optimization/CSolverStereoPosit.cpp:8-170   (the iteratively re-weighted Gauss-Newton loop, its checks)
 *   src/optimization/CSolverStereoPosit.h:89-98     (constants)
 *   src/vision/CMiniVisionToolbox.cpp:341-377       (getSkew, getTransformationFromVector)
 * Eigen pieces restated from their published algorithms: Quaterniond::toRotationMatrix, Isometry3d::inverse
 * (R', -R't), LDLT (Cholesky with diagonal pivoting: at step k the largest remaining |diagonal| is moved to k).
 * Measurements are accumulated in input order, as the reference's range-for does.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

typedef struct orc_posit_params {
    double P_left[12], P_right[12];
    int    min_points, min_inliers, max_iterations;                       /* 25, 15, 1000    :89-91 */
    double max_error_inlier_l2, max_error_average_l2, max_risk;           /* 10, 9, 2        :92-94 */
    double convergence_delta, min_translation_l2;                          /* 1e-5, 1e-3      :95,98 */
} orc_posit_params;

typedef struct orc_posit_result {
    double  T[12];           /* WORLD -> LEFT, R row-major then t */
    double  error_average, risk;
    int32_t status;          /* 0 ok, 1 insufficient points, 2 not converged, 3 insufficient accuracy, 4 high risk */
    int32_t iterations, inliers, n;
} orc_posit_result;

static void quat_R(double w, double x, double y, double z, double* R)
{
    const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
    const double twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y, tyz = tz * y, tzz = tz * z;
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
    R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}

/* Eigen::LDLT (lower, diagonal pivoting) + solve; rhs overwritten by the solution */
static void ldlt_solve6(double* A, double* x)
{
    int perm[6];
    for (int k = 0; k < 6; ++k) {
        int p = k;
        double big = fabs(A[7 * k]);
        for (int i = k + 1; i < 6; ++i) if (fabs(A[7 * i]) > big) { big = fabs(A[7 * i]); p = i; }
        perm[k] = p;
        if (p != k) { /* symmetric swap of rows/columns k and p (full storage) */
            for (int j = 0; j < 6; ++j) { double t = A[6 * k + j]; A[6 * k + j] = A[6 * p + j]; A[6 * p + j] = t; }
            for (int j = 0; j < 6; ++j) { double t = A[6 * j + k]; A[6 * j + k] = A[6 * j + p]; A[6 * j + p] = t; }
        }
        const double d = A[7 * k];
        if (d == 0.0) continue;
        for (int i = k + 1; i < 6; ++i) A[6 * i + k] /= d;
        for (int i = k + 1; i < 6; ++i)
            for (int j = k + 1; j <= i; ++j) { A[6 * i + j] -= A[6 * i + k] * d * A[6 * j + k]; A[6 * j + i] = A[6 * i + j]; }
    }
    for (int k = 0; k < 6; ++k) if (perm[k] != k) { double t = x[k]; x[k] = x[perm[k]]; x[perm[k]] = t; }
    for (int i = 0; i < 6; ++i) for (int j = 0; j < i; ++j) x[i] -= A[6 * i + j] * x[j];
    for (int i = 0; i < 6; ++i) x[i] = (A[7 * i] == 0.0) ? 0.0 : x[i] / A[7 * i];
    for (int i = 5; i >= 0; --i) for (int j = i + 1; j < 6; ++j) x[i] -= A[6 * j + i] * x[j];
    for (int k = 5; k >= 0; --k) if (perm[k] != k) { double t = x[k]; x[k] = x[perm[k]]; x[perm[k]] = t; }
}

static void inverse_t(const double* T, double* t_inv)
{
    for (int r = 0; r < 3; ++r) t_inv[r] = -(T[r] * T[9] + T[3 + r] * T[10] + T[6 + r] * T[11]);
}

void orc_stereo_posit(const orc_posit_params* prm, const double* T_last, const double* t_imu, const double* T_estimate, const double* xyz_world,
                      const float* uv_left, const float* uv_right, const uint8_t* active, int n, orc_posit_result* out)
{
    int m = 0;
    for (int i = 0; i < n; ++i) m += (!active || active[i]) ? 1 : 0;
    memset(out, 0, sizeof(*out));
    memcpy(out->T, T_estimate, 12 * sizeof(double));
    out->n = m;
    if (!((uint32_t)prm->min_points < (uint32_t)m)) { out->status = 1; return; }                   /* :19 */
    double T[12];
    memcpy(T, T_estimate, sizeof(T));
    double prev = 0.0;
    const double* PL = prm->P_left;
    const double* PR = prm->P_right;
    for (int it = 0; it < prm->max_iterations; ++it) {
        double total = 0.0, H[36], b[6];
        int inliers = 0;
        memset(H, 0, sizeof(H));
        memset(b, 0, sizeof(b));
        for (int i = 0; i < n; ++i) {
            if (active && !active[i]) continue;
            const double* x = xyz_world + 3 * i;
            double p[3];
            for (int r = 0; r < 3; ++r) p[r] = T[3 * r] * x[0] + T[3 * r + 1] * x[1] + T[3 * r + 2] * x[2] + T[9 + r];
            if (!(0.0 < p[2])) continue;                                                         /* :41 */
            double aL[3], aR[3];
            for (int r = 0; r < 3; ++r) {
                aL[r] = PL[4 * r] * p[0] + PL[4 * r + 1] * p[1] + PL[4 * r + 2] * p[2] + PL[4 * r + 3];
                aR[r] = PR[4 * r] * p[0] + PR[4 * r + 1] * p[1] + PR[4 * r + 2] * p[2] + PR[4 * r + 3];
            }
            const double e[4] = {aL[0] / aL[2] - uv_left[2 * i], aL[1] / aL[2] - uv_left[2 * i + 1],
                                 aR[0] / aR[2] - uv_right[2 * i], aR[1] / aR[2] - uv_right[2 * i + 1]};   /* :55-58 */
            const double e2 = e[0] * e[0] + e[1] * e[1] + e[2] * e[2] + e[3] * e[3];
            double w = 1.0;
            if (prm->max_error_inlier_l2 < e2) w = prm->max_error_inlier_l2 / e2; else ++inliers;  /* :67-74 */
            total += w * e2;
            /* J = [ Jdiv P(:, :3) | Jdiv P(:, :3) (-2 [p]x) ]   (:78-97; the 4th row of the transform Jacobian is zero) */
            const double S[9] = {0, 2 * p[2], -2 * p[1], -2 * p[2], 0, 2 * p[0], 2 * p[1], -2 * p[0], 0};  /* -2 skew(p) */
            double J[4][6];
            for (int cam = 0; cam < 2; ++cam) {
                const double* P = cam ? PR : PL;
                const double* a = cam ? aR : aL;
                const double c = a[2];
                const double D[2][3] = {{1 / c, 0, -a[0] / (c * c)}, {0, 1 / c, -a[1] / (c * c)}};
                double A[2][3];
                for (int r = 0; r < 2; ++r)
                    for (int k = 0; k < 3; ++k) A[r][k] = D[r][0] * P[k] + D[r][1] * P[4 + k] + D[r][2] * P[8 + k];
                for (int r = 0; r < 2; ++r)
                    for (int k = 0; k < 3; ++k) {
                        J[2 * cam + r][k] = A[r][k];
                        J[2 * cam + r][3 + k] = A[r][0] * S[k] + A[r][1] * S[3 + k] + A[r][2] * S[6 + k];
                    }
            }
            for (int r = 0; r < 6; ++r) {
                for (int k = 0; k < 6; ++k) {
                    double s = 0.0;
                    for (int q = 0; q < 4; ++q) s += J[q][r] * J[q][k];
                    H[6 * r + k] += w * s;                                                        /* :103 */
                }
                double s = 0.0;
                for (int q = 0; q < 4; ++q) s += J[q][r] * e[q];
                b[r] += w * s;                                                                    /* :104 */
            }
        }
        double dx[6];
        for (int k = 0; k < 6; ++k) dx[k] = -b[k];
        ldlt_solve6(H, dx);                                                                        /* :109 */
        double dR[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        const double w2 = dx[3] * dx[3] + dx[4] * dx[4] + dx[5] * dx[5];
        if (1.0 > w2) quat_R(sqrt(1.0 - w2), dx[3], dx[4], dx[5], dR);
        double Tn[12];
        for (int r = 0; r < 3; ++r) {
            for (int k = 0; k < 3; ++k) Tn[3 * r + k] = dR[3 * r] * T[k] + dR[3 * r + 1] * T[3 + k] + dR[3 * r + 2] * T[6 + k];
            Tn[9 + r] = dR[3 * r] * T[9] + dR[3 * r + 1] * T[10] + dR[3 * r + 2] * T[11] + dx[r];
        }
        /* enforce rotation symmetry (:112-115): R -= 0.5 R (R'R - I) */
        double G[9], C[9];
        for (int r = 0; r < 3; ++r)
            for (int k = 0; k < 3; ++k) G[3 * r + k] = Tn[r] * Tn[k] + Tn[3 + r] * Tn[3 + k] + Tn[6 + r] * Tn[6 + k];
        G[0] -= 1.0; G[4] -= 1.0; G[8] -= 1.0;
        for (int r = 0; r < 3; ++r)
            for (int k = 0; k < 3; ++k) C[3 * r + k] = Tn[3 * r] * G[k] + Tn[3 * r + 1] * G[3 + k] + Tn[3 * r + 2] * G[6 + k];
        for (int k = 0; k < 9; ++k) Tn[k] -= 0.5 * C[k];
        memcpy(T, Tn, sizeof(T));
        out->iterations = it + 1;
        out->inliers = inliers;
        if (prm->convergence_delta > fabs(prev - total)) {                                         /* :118 */
            out->error_average = total / m;                                                        /* :124 */
            memcpy(out->T, T, sizeof(T));
            if (prm->max_error_average_l2 < out->error_average && (uint32_t)prm->min_inliers > (uint32_t)inliers) { out->status = 3; return; }
            const double d[3] = {T[9] - T_last[9], T[10] - T_last[10], T[11] - T_last[11]};
            if (prm->min_translation_l2 > d[0] * d[0] + d[1] * d[1] + d[2] * d[2]) { T[9] = T_last[9]; T[10] = T_last[10]; T[11] = T_last[11]; }  /* :137-141 */
            double ti[3], te[3];
            inverse_t(T, ti);
            inverse_t(T_estimate, te);
            double risk = 0.0;
            for (int k = 0; k < 3; ++k) { const double v = ti[k] - te[k] - t_imu[k]; risk += v * v; }   /* :145 */
            out->risk = risk;
            memcpy(out->T, T, sizeof(T));
            if (prm->max_risk < risk) { out->status = 4; return; }                                /* :148 */
            out->status = 0;
            return;
        }
        prev = total;
    }
    memcpy(out->T, T, sizeof(T));
    out->status = 2;                                                                               /* :165 */
}
