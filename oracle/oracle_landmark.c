/*
 * oracle_landmark.c — CPU restatement of CLandmark::optimize / _getOptimizedLandmarkSTEREOUV.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under svi_mapper_amd/ may include, link or call this file;
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the checker.
 *
 * PARITY UNPINNED: no fixtures in the reference, Eigen absent.  Follows
 *   src/types/CLandmark.cpp:281-296   optimize(): only landmarks with MORE than 5 measurements are refined
 *   src/types/CLandmark.cpp:447-581   the re-weighted Gauss-Newton on the stereo reprojection error in WORLD coordinates
 *   src/types/CLandmark.h:90-98       constants
 * Eigen's HouseholderQR (makeHouseholder / applyHouseholderOnTheLeft / triangular solve) is restated from its
 * published algorithm for the 4x3 system  H(:, 0:3) dx = -b  (the homogeneous coordinate is held fixed).
 * Compiled with -ffp-contract=off; the GPU kernel follows the same operation order and is compared bit for bit.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

typedef struct orc_landmark_params {
    int    min_measurements, cap_iterations;                  /* 5, 1000 */
    double convergence_delta, kernel_max_error_l2, min_inlier_ratio, max_error_average_l2;   /* 1e-5, 10, 0.5, 9 */
} orc_landmark_params;

enum { LMO_SKIPPED = 0, LMO_OPTIMAL = 1, LMO_CONVERGED = 2, LMO_REJECTED = 3, LMO_NOT_CONVERGED = 4 };

/* least-squares solution of the 4x3 system A x = r by Householder QR (A is 4x3 row-major, destroyed; r destroyed) */
static void qr_solve_4x3(double A[4][3], double r[4], double x[3])
{
    for (int k = 0; k < 3; ++k) {
        double tail = 0.0;
        for (int i = k + 1; i < 4; ++i) tail += A[i][k] * A[i][k];
        const double c0 = A[k][k];
        double tau, beta, ess[3] = {0, 0, 0};
        if (tail <= 2.2250738585072014e-308) { tau = 0.0; beta = c0; }         /* makeHouseholder: nothing to annihilate */
        else {
            beta = sqrt(c0 * c0 + tail);
            if (c0 >= 0.0) beta = -beta;
            for (int i = k + 1; i < 4; ++i) ess[i - k - 1] = A[i][k] / (c0 - beta);
            tau = (beta - c0) / beta;
        }
        A[k][k] = beta;
        /* apply H = I - tau v v' (v = [1; ess]) to the remaining columns and to the right-hand side */
        for (int j = k + 1; j < 3; ++j) {
            double s = A[k][j];
            for (int i = k + 1; i < 4; ++i) s += ess[i - k - 1] * A[i][j];
            s *= tau;
            A[k][j] -= s;
            for (int i = k + 1; i < 4; ++i) A[i][j] -= ess[i - k - 1] * s;
        }
        double s = r[k];
        for (int i = k + 1; i < 4; ++i) s += ess[i - k - 1] * r[i];
        s *= tau;
        r[k] -= s;
        for (int i = k + 1; i < 4; ++i) r[i] -= ess[i - k - 1] * s;
    }
    for (int i = 2; i >= 0; --i) {
        double s = r[i];
        for (int j = i + 1; j < 3; ++j) s -= A[i][j] * x[j];
        x[i] = s / A[i][i];
    }
}

void orc_landmarks_optimize(const orc_landmark_params* prm, const double* frame_P_left, const double* frame_P_right, int n_frames,
                            const int32_t* seg, const int32_t* meas_frame, const float* uvl, const float* uvr, const double* xyz_in, int n,
                            double* xyz_out, int32_t* status, double* error_avg, int32_t* iterations)
{
    for (int l = 0; l < n; ++l) {
        const int m0 = seg[l], m = seg[l + 1] - seg[l];
        double X[4] = {xyz_in[3 * l], xyz_in[3 * l + 1], xyz_in[3 * l + 2], 1.0};
        memcpy(xyz_out + 3 * l, xyz_in + 3 * l, 3 * sizeof(double));
        error_avg[l] = 0.0; iterations[l] = 0;
        if (!((uint32_t)prm->min_measurements < (uint32_t)m)) { status[l] = LMO_SKIPPED; continue; }   /* :287-295 */
        double prev = 0.0;
        status[l] = LMO_NOT_CONVERGED;
        for (int it = 0; it < prm->cap_iterations; ++it) {
            double total = 0.0, H[4][4], b[4];
            uint32_t inliers = 0;
            memset(H, 0, sizeof(H)); memset(b, 0, sizeof(b));
            for (int q = 0; q < m; ++q) {
                const int f = meas_frame[m0 + q];
                const double* PL = frame_P_left + 12 * (size_t)f;
                const double* PR = frame_P_right + 12 * (size_t)f;
                double aL[3], aR[3];
                for (int r = 0; r < 3; ++r) {
                    aL[r] = PL[4 * r] * X[0] + PL[4 * r + 1] * X[1] + PL[4 * r + 2] * X[2] + PL[4 * r + 3] * X[3];
                    aR[r] = PR[4 * r] * X[0] + PR[4 * r + 1] * X[1] + PR[4 * r + 2] * X[2] + PR[4 * r + 3] * X[3];
                }
                const double cL = aL[2], cR = aR[2];
                const double e[4] = {aL[0] / cL - uvl[2 * (m0 + q)], aL[1] / cL - uvl[2 * (m0 + q) + 1],
                                     aR[0] / cR - uvr[2 * (m0 + q)], aR[1] / cR - uvr[2 * (m0 + q) + 1]};        /* :478-481 */
                const double e2 = e[0] * e[0] + e[1] * e[1] + e[2] * e[2] + e[3] * e[3];
                double w = 1.0;
                if (prm->kernel_max_error_l2 < e2) w = prm->kernel_max_error_l2 / e2; else ++inliers;            /* :491-499 */
                total += w * e2;
                double J[4][4];
                const double dL[2][3] = {{1 / cL, 0, -aL[0] / (cL * cL)}, {0, 1 / cL, -aL[1] / (cL * cL)}};
                const double dR[2][3] = {{1 / cR, 0, -aR[0] / (cR * cR)}, {0, 1 / cR, -aR[1] / (cR * cR)}};
                for (int r = 0; r < 2; ++r)
                    for (int k = 0; k < 4; ++k) {
                        J[r][k] = dL[r][0] * PL[k] + dL[r][1] * PL[4 + k] + dL[r][2] * PL[8 + k];                 /* :512 */
                        J[2 + r][k] = dR[r][0] * PR[k] + dR[r][1] * PR[4 + k] + dR[r][2] * PR[8 + k];             /* :513 */
                    }
                for (int r = 0; r < 4; ++r) {
                    for (int k = 0; k < 4; ++k) {
                        double s = J[0][r] * J[0][k];
                        s += J[1][r] * J[1][k];
                        s += J[2][r] * J[2][k];
                        s += J[3][r] * J[3][k];
                        H[r][k] += w * s;                                                                        /* :519 */
                    }
                    double s = J[0][r] * e[0];
                    s += J[1][r] * e[1];
                    s += J[2][r] * e[2];
                    s += J[3][r] * e[3];
                    b[r] += w * s;                                                                               /* :520 */
                }
            }
            double A[4][3], r4[4], dx[3];
            for (int r = 0; r < 4; ++r) { A[r][0] = H[r][0]; A[r][1] = H[r][1]; A[r][2] = H[r][2]; r4[r] = -b[r]; }
            qr_solve_4x3(A, r4, dx);                                                                             /* :524 */
            X[0] += dx[0]; X[1] += dx[1]; X[2] += dx[2];
            iterations[l] = it + 1;
            if (prm->convergence_delta > fabs(prev - total)) {                                                   /* :531 */
                const double avg = total / (double)m;
                error_avg[l] = avg;
                if (prm->min_inlier_ratio < (double)inliers / (double)m) {                                       /* :537 */
                    status[l] = (prm->max_error_average_l2 > avg) ? LMO_OPTIMAL : LMO_CONVERGED;                 /* :546 */
                    xyz_out[3 * l] = X[0]; xyz_out[3 * l + 1] = X[1]; xyz_out[3 * l + 2] = X[2];
                } else status[l] = LMO_REJECTED;                                                                 /* :559: keeps the initial guess */
                break;
            }
            prev = total;
        }
    }
}
