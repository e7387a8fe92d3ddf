// landmark.hip — CLandmark::optimize / _getOptimizedLandmarkSTEREOUV for all active landmarks at once.
//
// The reference refines every active landmark every frame (src/core/CTrackerGT.cpp:197) with a re-weighted
// Gauss-Newton on the stereo reprojection error over all of the landmark's measurements
// (src/types/CLandmark.cpp:281-296, 447-581): per measurement a 4x4 Jacobian, H += w J'J, b += w J'e, then the
// 4x3 least-squares system H(:,0:3) dx = -b by Householder QR; at most 1000 iterations, usually a handful.
// The landmarks are independent: a group of 16 lanes per landmark linearises 16 measurements at a time and adds
// their contributions in measurement order (the summation order of the reference), the per-frame projection matrices
// P*T_world_to_camera (24 doubles per frame) are shared through L2.  Compiled with -ffp-contract=off: the results are
// bit-identical to the CPU restatement (oracle/oracle_landmark.c).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>

#include "common.h"
#include "matcher_handle.h"

namespace {

struct LmArgs {
    int    min_measurements, cap_iterations;
    double conv_delta, kernel_max, min_ratio, max_avg;
    const double* PL;
    const double* PR;
    const int32_t* seg;
    const int32_t* frame;
    const float2* uvl;
    const float2* uvr;
    const double* xyz_in;
    int n;
    double* xyz_out;
    int32_t* status;
    double* error_avg;
    int32_t* iterations;
};

// Eigen::HouseholderQR on the 4x3 system (makeHouseholder, applyHouseholderOnTheLeft, back substitution)
__device__ void qr_solve_4x3(double A[4][3], double r[4], double x[3])
{
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        double tail = 0.0;
#pragma unroll
        for (int i = k + 1; i < 4; ++i) tail += A[i][k] * A[i][k];
        const double c0 = A[k][k];
        double tau, beta, ess[3] = {0.0, 0.0, 0.0};
        if (tail <= 2.2250738585072014e-308) { tau = 0.0; beta = c0; }
        else {
            beta = sqrt(c0 * c0 + tail);
            if (c0 >= 0.0) beta = -beta;
#pragma unroll
            for (int i = k + 1; i < 4; ++i) ess[i - k - 1] = A[i][k] / (c0 - beta);
            tau = (beta - c0) / beta;
        }
        A[k][k] = beta;
#pragma unroll
        for (int j = k + 1; j < 3; ++j) {
            double s = A[k][j];
#pragma unroll
            for (int i = k + 1; i < 4; ++i) s += ess[i - k - 1] * A[i][j];
            s *= tau;
            A[k][j] -= s;
#pragma unroll
            for (int i = k + 1; i < 4; ++i) A[i][j] -= ess[i - k - 1] * s;
        }
        double s = r[k];
#pragma unroll
        for (int i = k + 1; i < 4; ++i) s += ess[i - k - 1] * r[i];
        s *= tau;
        r[k] -= s;
#pragma unroll
        for (int i = k + 1; i < 4; ++i) r[i] -= ess[i - k - 1] * s;
    }
#pragma unroll
    for (int i = 2; i >= 0; --i) {
        double s = r[i];
#pragma unroll
        for (int j = i + 1; j < 3; ++j) s -= A[i][j] * x[j];
        x[i] = s / A[i][i];
    }
}

// A landmark is refined by a GROUP of 16 lanes (4 landmarks per wavefront): every lane linearises one measurement of
// the current chunk of 16 and parks its 15 contributions (10 unique H entries, 4 b entries, weighted error) in LDS; lane c
// of the group then adds component c over the chunk IN MEASUREMENT ORDER - the rounding sequence of the reference's
// sequential loop (H += w J'J ...), so the result stays bit-identical to the CPU restatement; lane 0 solves the 4x3
// system (Householder QR) and publishes the new position and the verdict.
constexpr int kGroup = 16, kComp = 15;

__global__ __launch_bounds__(64) void k_landmarks_optimize(LmArgs a)
{
    __shared__ double s_c[4][kGroup][kComp + 1]; // [group][measurement of the chunk][component], +1: inlier flag
    __shared__ double s_sum[4][kComp + 1];
    __shared__ double s_X[4][4];                 // position, then the control word
    const int grp = threadIdx.x >> 4, gl = threadIdx.x & 15;
    const int l = blockIdx.x * 4 + grp;
    const bool live = l < a.n;
    const int m0 = live ? a.seg[l] : 0, m = live ? a.seg[l + 1] - m0 : 0;
    double X[4] = {0.0, 0.0, 0.0, 1.0};
    if (live) { X[0] = a.xyz_in[3 * l]; X[1] = a.xyz_in[3 * l + 1]; X[2] = a.xyz_in[3 * l + 2]; }
    const double X0[3] = {X[0], X[1], X[2]};
    int32_t st = SVI_LM_OPT_SKIPPED, iters = 0;
    double avg_out = 0.0;
    bool keep = false, running = live && static_cast<uint32_t>(a.min_measurements) < static_cast<uint32_t>(m);   // CLandmark.cpp:287
    if (running) st = SVI_LM_OPT_NOT_CONVERGED;
    double prev = 0.0;
    for (int it = 0; it < a.cap_iterations; ++it) {
        if (__ballot(running) == 0ull) break; // every group of the wavefront is done
        double acc = 0.0;        // component gl of H / b / total (lanes 0..14), inlier count in lane 15
        for (int q0 = 0; q0 < m; q0 += kGroup) {
            const int q = q0 + gl;
            double c[kComp + 1];
#pragma unroll
            for (int k = 0; k <= kComp; ++k) c[k] = 0.0;
            if (running && q < m) {
                const int f = a.frame[m0 + q];
                const double* PL = a.PL + 12 * static_cast<size_t>(f);
                const double* PR = a.PR + 12 * static_cast<size_t>(f);
                double pl[12], pr[12];
#pragma unroll
                for (int k = 0; k < 12; ++k) { pl[k] = PL[k]; pr[k] = PR[k]; }
                double aL[3], aR[3];
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    aL[r] = pl[4 * r] * X[0] + pl[4 * r + 1] * X[1] + pl[4 * r + 2] * X[2] + pl[4 * r + 3] * X[3];
                    aR[r] = pr[4 * r] * X[0] + pr[4 * r + 1] * X[1] + pr[4 * r + 2] * X[2] + pr[4 * r + 3] * X[3];
                }
                const double cL = aL[2], cR = aR[2];
                const float2 mL = a.uvl[m0 + q], mR = a.uvr[m0 + q];
                const double e[4] = {aL[0] / cL - mL.x, aL[1] / cL - mL.y, aR[0] / cR - mR.x, aR[1] / cR - mR.y};   // :478-481
                const double e2 = e[0] * e[0] + e[1] * e[1] + e[2] * e[2] + e[3] * e[3];
                double w = 1.0;
                if (a.kernel_max < e2) w = a.kernel_max / e2; else c[kComp] = 1.0;                  // :491-499
                c[14] = w * e2;
                double J[4][4];
                const double dL[2][3] = {{1 / cL, 0, -aL[0] / (cL * cL)}, {0, 1 / cL, -aL[1] / (cL * cL)}};
                const double dR[2][3] = {{1 / cR, 0, -aR[0] / (cR * cR)}, {0, 1 / cR, -aR[1] / (cR * cR)}};
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        J[r][k] = dL[r][0] * pl[k] + dL[r][1] * pl[4 + k] + dL[r][2] * pl[8 + k];   // :512
                        J[2 + r][k] = dR[r][0] * pr[k] + dR[r][1] * pr[4 + k] + dR[r][2] * pr[8 + k];
                    }
                int u = 0;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
#pragma unroll
                    for (int k = r; k < 4; ++k) {   // H is bitwise symmetric (products commute): 10 unique entries
                        double s = J[0][r] * J[0][k];
                        s += J[1][r] * J[1][k];
                        s += J[2][r] * J[2][k];
                        s += J[3][r] * J[3][k];
                        c[u++] = w * s;                                                             // :519
                    }
                    double s = J[0][r] * e[0];
                    s += J[1][r] * e[1];
                    s += J[2][r] * e[2];
                    s += J[3][r] * e[3];
                    c[10 + r] = w * s;                                                              // :520
                }
            }
#pragma unroll
            for (int k = 0; k <= kComp; ++k) s_c[grp][gl][k] = c[k];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const int cnt = min(kGroup, m - q0);
            for (int t = 0; t < cnt; ++t) acc += s_c[grp][t][gl];   // measurement order, one component per lane
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        s_sum[grp][gl] = acc;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (gl == 0 && running) {
            const double* S = s_sum[grp];
            // unique entries in (r,k>=r) order: 00 01 02 03 11 12 13 22 23 33
            const double H[4][4] = {{S[0], S[1], S[2], S[3]}, {S[1], S[4], S[5], S[6]}, {S[2], S[5], S[7], S[8]}, {S[3], S[6], S[8], S[9]}};
            const double total = S[14];
            const uint32_t inliers = static_cast<uint32_t>(S[15]);
            double A[4][3], r4[4], dx[3];
#pragma unroll
            for (int r = 0; r < 4; ++r) { A[r][0] = H[r][0]; A[r][1] = H[r][1]; A[r][2] = H[r][2]; r4[r] = -S[10 + r]; }
            qr_solve_4x3(A, r4, dx);                                                                // :524
            double ctl = 0.0; // 0: go on, 1: converged and accepted, 2: converged and rejected
            s_X[grp][0] = X[0] + dx[0]; s_X[grp][1] = X[1] + dx[1]; s_X[grp][2] = X[2] + dx[2];
            if (a.conv_delta > fabs(prev - total)) {                                                // :531
                const double avg = total / static_cast<double>(m);
                avg_out = avg;
                if (a.min_ratio < static_cast<double>(inliers) / static_cast<double>(m)) {          // :537
                    st = (a.max_avg > avg) ? SVI_LM_OPT_OPTIMAL : SVI_LM_OPT_CONVERGED;             // :546
                    keep = true;
                    ctl = 1.0;
                } else { st = SVI_LM_OPT_REJECTED; ctl = 2.0; }
            }
            prev = total;
            s_X[grp][3] = ctl;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (running) {
            X[0] = s_X[grp][0]; X[1] = s_X[grp][1]; X[2] = s_X[grp][2];
            iters = it + 1;
            if (s_X[grp][3] != 0.0) running = false;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    if (live && gl == 0) {
        a.xyz_out[3 * l] = keep ? X[0] : X0[0];
        a.xyz_out[3 * l + 1] = keep ? X[1] : X0[1];
        a.xyz_out[3 * l + 2] = keep ? X[2] : X0[2];
        a.status[l] = st;
        a.error_avg[l] = avg_out;
        a.iterations[l] = iters;
    }
}

} // namespace

extern "C" {

void svi_landmark_params_default(svi_landmark_params* p)
{
    if (!p) return;
    p->min_measurements = 5; p->cap_iterations = 1000;                                    // CLandmark.h:98, :90
    p->convergence_delta = 1e-5; p->kernel_max_error_l2 = 10.0;                           // :93, :95
    p->min_inlier_ratio = 0.5; p->max_error_average_l2 = 9.0;                             // :94, :96
}

int svi_landmarks_optimize_dev(svi_matcher* m, const svi_landmark_params* prm, const double* frame_P_left, const double* frame_P_right,
                               int n_frames, const int32_t* meas_seg, const int32_t* meas_frame, const float* meas_uv_left,
                               const float* meas_uv_right, const double* xyz_in, int n, double* xyz_out, int32_t* out_status,
                               double* out_error_average, int32_t* out_iterations)
{
    if (!m || !prm) return svi::fail(SVI_ERR_INVALID, "svi_landmarks_optimize_dev: null handle / params");
    if (n < 0 || n_frames < 0 || prm->cap_iterations < 1) return svi::fail(SVI_ERR_INVALID, "svi_landmarks_optimize_dev: bad sizes");
    if (n == 0) return SVI_OK;
    if (!meas_seg || !xyz_in || !xyz_out || !out_status || !out_error_average || !out_iterations)
        return svi::fail(SVI_ERR_INVALID, "svi_landmarks_optimize_dev: null array");
    SVI_HIP(svi::enter_device(m->device));
    LmArgs a{};
    a.min_measurements = prm->min_measurements; a.cap_iterations = prm->cap_iterations;
    a.conv_delta = prm->convergence_delta; a.kernel_max = prm->kernel_max_error_l2; a.min_ratio = prm->min_inlier_ratio;
    a.max_avg = prm->max_error_average_l2;
    a.PL = frame_P_left; a.PR = frame_P_right; a.seg = meas_seg; a.frame = meas_frame;
    a.uvl = reinterpret_cast<const float2*>(meas_uv_left); a.uvr = reinterpret_cast<const float2*>(meas_uv_right);
    a.xyz_in = xyz_in; a.n = n; a.xyz_out = xyz_out; a.status = out_status; a.error_avg = out_error_average; a.iterations = out_iterations;
    hipLaunchKernelGGL(k_landmarks_optimize, dim3((n + 3) / 4), dim3(64), 0, m->stream, a);
    SVI_HIP(hipGetLastError());
    return SVI_OK;
}

} // extern "C"
