// posit.hip — CSolverStereoPosit::getTransformationWORLDtoLEFT on the MI355X.
//
// The reference refines the frame pose from the stage-1/2 matches with an iteratively re-weighted Gauss-Newton
// loop (src/optimization/CSolverStereoPosit.cpp:8-170): per measurement a 4x6 Jacobian of the stereo reprojection,
// H += w J'J, b += w J'e, then a 6x6 LDLT solve, the pose update X <- fromVector(dx) * X, a first-order
// re-orthogonalisation of R, and a convergence test on the total weighted error; a few hundred measurements,
// typically 5-15 iterations, at most 1000.
//
// One launch does the WHOLE solve: a single workgroup of 256 threads keeps the pose in LDS, every thread
// accumulates its measurements (stride 256, ascending) into 29 registers (21 H + 6 b + error + inliers), a fixed
// shuffle/LDS tree reduces them (deterministic for a given n), thread 0 factors the 6x6 system (LDLT with
// diagonal pivoting, as Eigen) and applies the update; the loop condition lives in LDS so every wave leaves the
// loop in the same iteration.  Nothing returns to the host between iterations: the problem is latency bound
// (n x 40 B of input, re-read from L2 each iteration), not bandwidth or FLOP bound.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>

#include "common.h"
#include "matcher_handle.h"

namespace {

constexpr int kThreads = 256;
constexpr int kAcc = 29; // 21 (upper H) + 6 (b) + total error + inliers

struct PositArgs {
    double PL[12], PR[12];
    double T_est[12], T_last[12], t_imu[3];
    int    min_points, min_inliers, max_iterations;
    double max_err_inlier, max_err_avg, max_risk, conv_delta, min_trans;
    const double* xyz;
    const float2* uvL;
    const float2* uvR;
    const uint8_t* active;
    int n;
    svi_posit_result* out; // device
};

__device__ void quat_R(double w, double x, double y, double z, double* R)
{
    const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
    const double twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y, tyz = tz * y, tzz = tz * z;
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
    R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}

// Eigen::LDLT: lower Cholesky with diagonal pivoting; x holds the right-hand side, then the solution.
// Everything lives in REGISTERS: all loops are unrolled and the one dynamic index - the pivot row p - is resolved by predicated
// swaps over the candidates, so no element is ever addressed through LDS (the version that kept the 6x6 system in LDS paid an LDS
// round trip for each of its ~300 dependent accesses: ~8 of the 15 us of an iteration).  Operation order as before: same bits.
__device__ __forceinline__ void ldlt_solve6(double (&A)[36], double (&x)[6])
{
    int perm[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        int p = k;
        double big = fabs(A[7 * k]);
#pragma unroll
        for (int i = k + 1; i < 6; ++i) if (fabs(A[7 * i]) > big) { big = fabs(A[7 * i]); p = i; }
        perm[k] = p;
#pragma unroll
        for (int i = k + 1; i < 6; ++i)
            if (p == i) { // swap rows, then columns, k <-> i (static indices under a uniform predicate)
#pragma unroll
                for (int j = 0; j < 6; ++j) { const double t = A[6 * k + j]; A[6 * k + j] = A[6 * i + j]; A[6 * i + j] = t; }
#pragma unroll
                for (int j = 0; j < 6; ++j) { const double t = A[6 * j + k]; A[6 * j + k] = A[6 * j + i]; A[6 * j + i] = t; }
            }
        const double d = A[7 * k];
        if (d != 0.0) {
#pragma unroll
            for (int i = k + 1; i < 6; ++i) A[6 * i + k] /= d;
#pragma unroll
            for (int i = k + 1; i < 6; ++i)
#pragma unroll
                for (int j = k + 1; j <= i; ++j) { A[6 * i + j] -= A[6 * i + k] * d * A[6 * j + k]; A[6 * j + i] = A[6 * i + j]; }
        }
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) {
#pragma unroll
        for (int i = k + 1; i < 6; ++i) if (perm[k] == i) { const double t = x[k]; x[k] = x[i]; x[i] = t; }
    }
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < i; ++j) x[i] -= A[6 * i + j] * x[j];
#pragma unroll
    for (int i = 0; i < 6; ++i) x[i] = (A[7 * i] == 0.0) ? 0.0 : x[i] / A[7 * i];
#pragma unroll
    for (int i = 5; i >= 0; --i)
#pragma unroll
        for (int j = i + 1; j < 6; ++j) x[i] -= A[6 * j + i] * x[j];
#pragma unroll
    for (int k = 5; k >= 0; --k) {
#pragma unroll
        for (int i = k + 1; i < 6; ++i) if (perm[k] == i) { const double t = x[k]; x[k] = x[i]; x[i] = t; }
    }
}

__device__ void inverse_t(const double* T, double* ti)
{
    for (int r = 0; r < 3; ++r) ti[r] = -(T[r] * T[9] + T[3 + r] * T[10] + T[6 + r] * T[11]);
}

__global__ __launch_bounds__(kThreads) void k_stereo_posit(PositArgs a)
{
    __shared__ double s_T[12];
    __shared__ double s_red[kThreads / 64][kAcc];
    __shared__ int    s_m, s_stop;
    __shared__ double s_prev;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    // number of measurements (:16)
    int mine = 0;
    for (int i = tid; i < a.n; i += kThreads) mine += (!a.active || a.active[i]) ? 1 : 0;
    for (int off = 32; off > 0; off >>= 1) mine += __shfl_xor(mine, off);
    if (tid == 0) { s_m = 0; s_stop = 0; s_prev = 0.0; }
    if (tid < 12) s_T[tid] = a.T_est[tid];
    __syncthreads();
    if (lane == 0) atomicAdd(&s_m, mine);
    __syncthreads();
    const int m = s_m;
    if (!(static_cast<uint32_t>(a.min_points) < static_cast<uint32_t>(m))) {                      // :19
        if (tid == 0) {
            svi_posit_result r{};
            for (int k = 0; k < 12; ++k) r.T_world_to_left[k] = a.T_est[k];
            r.status = SVI_POSIT_FEW_POINTS; r.n = m;
            *a.out = r;
        }
        return;
    }

    for (int it = 0; it < a.max_iterations; ++it) {
        double T[12];
#pragma unroll
        for (int k = 0; k < 12; ++k) T[k] = s_T[k];
        double acc[kAcc];
#pragma unroll
        for (int k = 0; k < kAcc; ++k) acc[k] = 0.0;
        for (int i = tid; i < a.n; i += kThreads) {
            if (a.active && !a.active[i]) continue;
            const double x0 = a.xyz[3 * i], x1 = a.xyz[3 * i + 1], x2 = a.xyz[3 * i + 2];
            double p[3];
#pragma unroll
            for (int r = 0; r < 3; ++r) p[r] = T[3 * r] * x0 + T[3 * r + 1] * x1 + T[3 * r + 2] * x2 + T[9 + r];
            if (!(0.0 < p[2])) continue;                                                         // :41
            const float2 mL = a.uvL[i], mR = a.uvR[i];
            double J[4][6], e[4];
#pragma unroll
            for (int cam = 0; cam < 2; ++cam) {
                const double* P = cam ? a.PR : a.PL;
                double h[3];
#pragma unroll
                for (int r = 0; r < 3; ++r) h[r] = P[4 * r] * p[0] + P[4 * r + 1] * p[1] + P[4 * r + 2] * p[2] + P[4 * r + 3];
                const double ic = 1.0 / h[2];
                e[2 * cam] = h[0] / h[2] - static_cast<double>(cam ? mR.x : mL.x);               // :55-58
                e[2 * cam + 1] = h[1] / h[2] - static_cast<double>(cam ? mR.y : mL.y);
                const double d0 = -h[0] / (h[2] * h[2]), d1 = -h[1] / (h[2] * h[2]);
                double A[2][3];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    A[0][k] = ic * P[k] + d0 * P[8 + k];                                        // Jdiv * P(:, :3)  :85-96
                    A[1][k] = ic * P[4 + k] + d1 * P[8 + k];
                }
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    J[2 * cam + r][0] = A[r][0]; J[2 * cam + r][1] = A[r][1]; J[2 * cam + r][2] = A[r][2];
                    // A * (-2 [p]x)                                                              :81
                    J[2 * cam + r][3] = 2 * (A[r][2] * p[1] - A[r][1] * p[2]);
                    J[2 * cam + r][4] = 2 * (A[r][0] * p[2] - A[r][2] * p[0]);
                    J[2 * cam + r][5] = 2 * (A[r][1] * p[0] - A[r][0] * p[1]);
                }
            }
            const double e2 = e[0] * e[0] + e[1] * e[1] + e[2] * e[2] + e[3] * e[3];
            double w = 1.0;
            if (a.max_err_inlier < e2) w = a.max_err_inlier / e2; else acc[28] += 1.0;          // :67-74
            acc[27] += w * e2;
            int q = 0;
#pragma unroll
            for (int r = 0; r < 6; ++r) {
#pragma unroll
                for (int c = r; c < 6; ++c) {
                    acc[q++] += w * (J[0][r] * J[0][c] + J[1][r] * J[1][c] + J[2][r] * J[2][c] + J[3][r] * J[3][c]);   // :103
                }
            }
#pragma unroll
            for (int r = 0; r < 6; ++r) acc[21 + r] += w * (J[0][r] * e[0] + J[1][r] * e[1] + J[2][r] * e[2] + J[3][r] * e[3]);   // :104
        }
        // fixed-shape reduction: butterflies inside the wave, then waves in ascending order
#pragma unroll
        for (int k = 0; k < kAcc; ++k) {
            double v = acc[k];
            for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
            if (lane == 0) s_red[wave][k] = v;
        }
        __syncthreads();
        if (tid == 0) {
            double H[36], dx[6], sum[kAcc]; // wave sums added in ascending wave order; the 6x6 system in registers
#pragma unroll
            for (int k = 0; k < kAcc; ++k) { double v = s_red[0][k]; for (int w = 1; w < kThreads / 64; ++w) v += s_red[w][k]; sum[k] = v; }
            {
                int q = 0;
#pragma unroll
                for (int r = 0; r < 6; ++r)
#pragma unroll
                    for (int c = r; c < 6; ++c) { H[6 * r + c] = sum[q]; H[6 * c + r] = sum[q]; ++q; }
            }
#pragma unroll
            for (int k = 0; k < 6; ++k) dx[k] = -sum[21 + k];
            const double total = sum[27];
            const int inliers = static_cast<int>(sum[28]);
            ldlt_solve6(H, dx);                                                                   // :109
            double dR[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
            const double w2 = dx[3] * dx[3] + dx[4] * dx[4] + dx[5] * dx[5];
            if (1.0 > w2) quat_R(sqrt(1.0 - w2), dx[3], dx[4], dx[5], dR);
            double Tn[12];
            for (int r = 0; r < 3; ++r) {
                for (int k = 0; k < 3; ++k) Tn[3 * r + k] = dR[3 * r] * T[k] + dR[3 * r + 1] * T[3 + k] + dR[3 * r + 2] * T[6 + k];
                Tn[9 + r] = dR[3 * r] * T[9] + dR[3 * r + 1] * T[10] + dR[3 * r + 2] * T[11] + dx[r];
            }
            double G[9], Cm[9];
            for (int r = 0; r < 3; ++r)
                for (int k = 0; k < 3; ++k) G[3 * r + k] = Tn[r] * Tn[k] + Tn[3 + r] * Tn[3 + k] + Tn[6 + r] * Tn[6 + k];
            G[0] -= 1.0; G[4] -= 1.0; G[8] -= 1.0;                                                // :112-115
            for (int r = 0; r < 3; ++r)
                for (int k = 0; k < 3; ++k) Cm[3 * r + k] = Tn[3 * r] * G[k] + Tn[3 * r + 1] * G[3 + k] + Tn[3 * r + 2] * G[6 + k];
            for (int k = 0; k < 9; ++k) Tn[k] -= 0.5 * Cm[k];
            for (int k = 0; k < 12; ++k) s_T[k] = Tn[k];
            if (a.conv_delta > fabs(s_prev - total)) {                                            // :118
                svi_posit_result r{};
                r.n = m; r.iterations = it + 1; r.inliers = inliers;
                r.error_average = total / m;                                                      // :124
                r.status = SVI_POSIT_OK;
                if (a.max_err_avg < r.error_average && static_cast<uint32_t>(a.min_inliers) > static_cast<uint32_t>(inliers)) r.status = SVI_POSIT_INACCURATE;
                else {
                    const double d0 = Tn[9] - a.T_last[9], d1 = Tn[10] - a.T_last[10], d2 = Tn[11] - a.T_last[11];
                    if (a.min_trans > d0 * d0 + d1 * d1 + d2 * d2) { Tn[9] = a.T_last[9]; Tn[10] = a.T_last[10]; Tn[11] = a.T_last[11]; }   // :137-141
                    double ti[3], te[3];
                    inverse_t(Tn, ti);
                    inverse_t(a.T_est, te);
                    double risk = 0.0;
                    for (int k = 0; k < 3; ++k) { const double v = ti[k] - te[k] - a.t_imu[k]; risk += v * v; }   // :145
                    r.risk = risk;
                    if (a.max_risk < risk) r.status = SVI_POSIT_HIGH_RISK;                        // :148
                }
                for (int k = 0; k < 12; ++k) r.T_world_to_left[k] = Tn[k];
                *a.out = r;
                s_stop = 1;
            } else {
                s_prev = total;
                if (it + 1 == a.max_iterations) {                                                 // :165
                    svi_posit_result r{};
                    r.n = m; r.iterations = it + 1; r.inliers = inliers; r.status = SVI_POSIT_NOT_CONVERGED;
                    for (int k = 0; k < 12; ++k) r.T_world_to_left[k] = Tn[k];
                    *a.out = r;
                }
            }
        }
        __syncthreads();
        if (s_stop) break; // uniform: every wave reads the same LDS word after the barrier
    }
}

} // namespace

extern "C" {

void svi_posit_params_default(svi_posit_params* p)
{
    if (!p) return;
    memset(p, 0, sizeof(*p));
    p->min_points = 25; p->min_inliers = 15; p->max_iterations = 1000;         // CSolverStereoPosit.h:89-91
    p->max_error_inlier_l2 = 10.0; p->max_error_average_l2 = 9.0; p->max_risk = 2.0;   // :92-94
    p->convergence_delta = 1e-5; p->min_translation_l2 = 0.001;                // :95, :98
}

int svi_stereo_posit_dev(svi_matcher* m, const svi_posit_params* prm, const double* T_world_to_left_last, const double* t_imu,
                         const double* T_world_to_left_estimate, const double* xyz_world, const float* uv_left, const float* uv_right,
                         const uint8_t* active, int n, svi_posit_result* result)
{
    if (!m || !prm || !T_world_to_left_last || !t_imu || !T_world_to_left_estimate || !result)
        return svi::fail(SVI_ERR_INVALID, "svi_stereo_posit_dev: null handle / parameter / pose / result");
    if (n < 0 || prm->max_iterations < 1) return svi::fail(SVI_ERR_INVALID, "svi_stereo_posit_dev: bad n / max_iterations");
    if (n > 0 && (!xyz_world || !uv_left || !uv_right)) return svi::fail(SVI_ERR_INVALID, "svi_stereo_posit_dev: null measurement array");
    SVI_HIP(svi::enter_device(m->device));
    if (m->track.cap < sizeof(svi_posit_result)) SVI_HIP(hipStreamSynchronize(m->stream));
    if (int rc = m->track.reserve(sizeof(svi_posit_result) + 64)) return rc;
    PositArgs a{};
    for (int k = 0; k < 12; ++k) {
        a.PL[k] = prm->P_left[k]; a.PR[k] = prm->P_right[k];
        a.T_est[k] = T_world_to_left_estimate[k]; a.T_last[k] = T_world_to_left_last[k];
    }
    for (int k = 0; k < 3; ++k) a.t_imu[k] = t_imu[k];
    a.min_points = prm->min_points; a.min_inliers = prm->min_inliers; a.max_iterations = prm->max_iterations;
    a.max_err_inlier = prm->max_error_inlier_l2; a.max_err_avg = prm->max_error_average_l2; a.max_risk = prm->max_risk;
    a.conv_delta = prm->convergence_delta; a.min_trans = prm->min_translation_l2;
    a.xyz = xyz_world;
    a.uvL = reinterpret_cast<const float2*>(uv_left);
    a.uvR = reinterpret_cast<const float2*>(uv_right);
    a.active = active; a.n = n;
    // the plan staging area doubles as the result slot: wait for a plan upload still in flight
    if (m->track_ev) SVI_HIP(hipEventSynchronize(m->track_ev));
    a.out = m->track.as<svi_posit_result>();
    hipLaunchKernelGGL(k_stereo_posit, dim3(1), dim3(kThreads), 0, m->stream, a);
    SVI_HIP(hipGetLastError());
    SVI_HIP(hipMemcpyAsync(result, a.out, sizeof(svi_posit_result), hipMemcpyDeviceToHost, m->stream));
    SVI_HIP(hipStreamSynchronize(m->stream));
    return SVI_OK;
}

} // extern "C"
