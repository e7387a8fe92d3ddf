// ba_chol.hip — tile-sparse Cholesky of the reduced camera system S dx = g on gfx950.
//
// Stands in for what CHOLMOD does inside g2o::LinearSolverCholmod (configured at
// src/optimization/Cg2oOptimizer.cpp:83) on the pose part of the system once the landmarks have
// been eliminated (their elimination is the Schur reduction of ba_kernels.hip).  S is stored as
// TS x TS tiles of its lower block triangle, only tiles that can be non-zero after fill-in; the
// factorisation is right-looking over tile columns:
//     potrf:  L_kk = chol(S_kk + lambda I) and L_kk^-1            one workgroup, LDS resident
//     trsm :  L_ik = S_ik L_kk^-T  as a GEMM with L_kk^-1           one workgroup per tile, FP64 MFMA
//     gemm :  S_ij -= L_ik L_jk'                                    one workgroup per tile, FP64 MFMA
// (v_mfma_f64_16x16x4_f64; this dense block is the only MFMA-shaped work on the whole path).
// A non-positive pivot sets *status = k+1: the caller treats the LM trial as failed, like g2o does
// when CHOLMOD reports "not positive definite".
#include <hip/hip_runtime.h>

#include <cstdint>

#include "ba_device.h"

namespace svi {
namespace {

constexpr int kBlock = 256;
typedef double v4f64 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int ldp(int TS) { return TS + 2; } // LDS row stride: (TS+2) % 32 == 2 -> conflict-free b64 reads

// copy a TS x TS row-major tile global -> LDS (padded rows)
__device__ __forceinline__ void tile_to_lds(const double* __restrict__ g, double* s, int TS)
{
    const int LD = ldp(TS);
    for (int i = threadIdx.x; i < TS * TS / 2; i += kBlock) {
        const int r = (2 * i) / TS, c = (2 * i) % TS;
        const double2 v = *reinterpret_cast<const double2*>(g + 2 * i);
        s[r * LD + c] = v.x;
        s[r * LD + c + 1] = v.y;
    }
}

// ---------------------------------------------------------------------------------------------
// potrf + triangular inverse of one diagonal tile
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_potrf_inv(double* tiles, double* Linv, int tile_id, int k, int TS, int n,
                                                      double lambda, int* status)
{
    extern __shared__ __align__(16) double sm[];
    const int LD = ldp(TS);
    double* sA = sm;           // [TS][LD]
    double* sX = sm + TS * LD; // [TS][LD]
    __shared__ int s_fail;
    const int tid = threadIdx.x;
    if (*status != 0) return;
    double* A = tiles + (size_t)tile_id * TS * TS;
    tile_to_lds(A, sA, TS);
    if (tid == 0) s_fail = 0;
    __syncthreads();
    if (tid < TS && k * TS + tid < n) sA[tid * LD + tid] += lambda; // g2o setLambda: H_jj += lambda on real rows
    __syncthreads();

    // Right-looking LDL-style sweep with ONE barrier per column: at step j every thread applies
    // A[r][c] -= A[r][j] A[c][j] / A[j][j] to its elements of the trailing lower triangle; column j
    // itself is left unscaled and scaled by 1/sqrt(A[j][j]) at the end.
    for (int j = 0; j < TS; ++j) {
        const double djj = sA[j * LD + j];
        if (!(djj > 0.0)) { if (tid == 0) s_fail = 1; break; } // uniform: every thread reads the same value
        const double rinv = 1.0 / djj;
        const int m = TS - j - 1; // trailing size
        // elements (r,c), j < c <= r < TS, enumerated row-major over the m x m lower triangle
        const int cnt = m * (m + 1) / 2;
        for (int q = tid; q < cnt; q += kBlock) {
            int rr = (int)((sqrt(8.0 * q + 1.0) - 1.0) * 0.5);
            while ((rr + 1) * (rr + 2) / 2 <= q) ++rr;
            while (rr * (rr + 1) / 2 > q) --rr;
            const int cc = q - rr * (rr + 1) / 2;
            const int r = j + 1 + rr, c = j + 1 + cc;
            sA[r * LD + c] -= sA[r * LD + j] * sA[c * LD + j] * rinv;
        }
        __syncthreads();
    }
    __syncthreads();
    if (s_fail) { if (tid == 0) *status = k + 1; return; }
    // scale columns: L[r][j] = A[r][j] / sqrt(A[j][j])
    for (int q = tid; q < TS * TS; q += kBlock) {
        const int r = q / TS, j = q % TS;
        if (j <= r) {
            const double s = 1.0 / sqrt(sA[j * LD + j]);
            sX[r * LD + j] = sA[r * LD + j] * s; // stage in sX to avoid racing on the diagonal
        }
    }
    __syncthreads();
    for (int q = tid; q < TS * TS; q += kBlock) {
        const int r = q / TS, j = q % TS;
        const double v = (j <= r) ? sX[r * LD + j] : 0.0;
        sA[r * LD + j] = v;
        A[q] = v; // L_kk with an explicit zero upper triangle
    }
    __syncthreads();
    // X = L^-1, one column per thread (forward substitution on e_j)
    if (tid < TS) {
        const int j = tid;
        for (int i = 0; i < j; ++i) sX[i * LD + j] = 0.0;
        sX[j * LD + j] = 1.0 / sA[j * LD + j];
        for (int i = j + 1; i < TS; ++i) {
            double acc = 0.0;
            for (int m = j; m < i; ++m) acc += sA[i * LD + m] * sX[m * LD + j];
            sX[i * LD + j] = -acc / sA[i * LD + i];
        }
    }
    __syncthreads();
    double* X = Linv + (size_t)k * TS * TS;
    for (int q = tid; q < TS * TS; q += kBlock) X[q] = sX[(q / TS) * LD + (q % TS)];
}

// one 16x16 sub-tile of C = A B' with A, B in LDS (row-major, stride LD), K = TS
__device__ __forceinline__ v4f64 mfma_subtile(const double* sA, const double* sB, int r0, int c0, int TS, int LD)
{
    const int lane = threadIdx.x & 63;
    const double* pa = sA + (r0 + (lane & 15)) * LD + (lane >> 4);
    const double* pb = sB + (c0 + (lane & 15)) * LD + (lane >> 4);
    v4f64 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
    for (int kk = 0; kk < TS; kk += 4) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[kk], pb[kk], acc, 0, 0, 0);
    return acc;
}

// trsm: tile (i,k) <- tile(i,k) * Linv_kk'
__global__ __launch_bounds__(kBlock) void k_trsm(double* tiles, const double* __restrict__ Linv, const int* __restrict__ list,
                                                 int k, int TS, const int* status)
{
    extern __shared__ __align__(16) double sm[];
    if (*status != 0) return;
    const int LD = ldp(TS);
    double* sA = sm;
    double* sB = sm + TS * LD;
    double* A = tiles + (size_t)list[blockIdx.x] * TS * TS;
    tile_to_lds(A, sA, TS);
    tile_to_lds(Linv + (size_t)k * TS * TS, sB, TS);
    __syncthreads();
    const int nsub = TS / 16, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int st = wave; st < nsub * nsub; st += 4) {
        const int r0 = (st / nsub) * 16, c0 = (st % nsub) * 16;
        const v4f64 acc = mfma_subtile(sA, sB, r0, c0, TS, LD);
#pragma unroll
        for (int q = 0; q < 4; ++q) A[(size_t)(r0 + (lane >> 4) + 4 * q) * TS + c0 + (lane & 15)] = acc[q];
    }
}

// update: tile c -= tile a * tile b'
__global__ __launch_bounds__(kBlock) void k_gemm_upd(double* tiles, const int* __restrict__ ua, const int* __restrict__ ub,
                                                     const int* __restrict__ uc, int TS, const int* status)
{
    extern __shared__ __align__(16) double sm[];
    if (*status != 0) return;
    const int LD = ldp(TS);
    double* sA = sm;
    double* sB = sm + TS * LD;
    const int ia = ua[blockIdx.x], ib = ub[blockIdx.x];
    tile_to_lds(tiles + (size_t)ia * TS * TS, sA, TS);
    if (ib != ia) tile_to_lds(tiles + (size_t)ib * TS * TS, sB, TS);
    else sB = sA;
    __syncthreads();
    double* C = tiles + (size_t)uc[blockIdx.x] * TS * TS;
    const int nsub = TS / 16, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int st = wave; st < nsub * nsub; st += 4) {
        const int r0 = (st / nsub) * 16, c0 = (st % nsub) * 16;
        const v4f64 acc = mfma_subtile(sA, sB, r0, c0, TS, LD);
#pragma unroll
        for (int q = 0; q < 4; ++q) C[(size_t)(r0 + (lane >> 4) + 4 * q) * TS + c0 + (lane & 15)] -= acc[q];
    }
}

// ---------------------------------------------------------------------------------------------
// triangular solves L y = g, L' x = y over the tile structure (one workgroup; vectors in global)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_chol_solve(const double* __restrict__ tiles, const double* __restrict__ Linv,
                                                       const double* __restrict__ g, double* x, CholPlan p, const int* status)
{
    __shared__ double s_acc[kMaxTile];
    __shared__ double s_part[4][kMaxTile];
    const int TS = p.TS, NT = p.NT, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (*status != 0) return;
    // forward: y_k = Linv_kk (g_k - sum_{j<k} L_kj y_j); y stored in x
    for (int k = 0; k < NT; ++k) {
        if (tid < TS) s_acc[tid] = g[k * TS + tid];
        __syncthreads();
        for (int q = p.row_ptr[k]; q < p.row_ptr[k + 1]; ++q) {
            const double* L = tiles + (size_t)p.row_tile[q] * TS * TS;
            const double* y = x + p.row_col[q] * TS;
            for (int r = wave; r < TS; r += 4) {
                double s = 0.0;
                for (int c = lane; c < TS; c += 64) s += L[r * TS + c] * y[c];
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
                if (lane == 0) s_acc[r] -= s;
            }
            __syncthreads();
        }
        const double* X = Linv + (size_t)k * TS * TS;
        for (int r = wave; r < TS; r += 4) {
            double s = 0.0;
            for (int c = lane; c <= r; c += 64) s += X[r * TS + c] * s_acc[c];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
            if (lane == 0) x[k * TS + r] = s;
        }
        __syncthreads();
        __threadfence_block();
    }
    // backward: x_k = Linv_kk' (y_k - sum_{i>k} L_ik' x_i)
    for (int k = NT - 1; k >= 0; --k) {
        if (tid < TS) s_acc[tid] = x[k * TS + tid];
        __syncthreads();
        for (int q = p.col_ptr[k]; q < p.col_ptr[k + 1]; ++q) {
            const double* L = tiles + (size_t)p.trsm_tile[q] * TS * TS;
            const double* xi = x + p.trsm_row[q] * TS;
            for (int c = lane; c < TS; c += 64) {
                double s = 0.0;
                for (int r = wave; r < TS; r += 4) s += L[r * TS + c] * xi[r];
                s_part[wave][c] = s;
            }
            __syncthreads();
            if (tid < TS) s_acc[tid] -= (s_part[0][tid] + s_part[1][tid]) + (s_part[2][tid] + s_part[3][tid]);
            __syncthreads();
        }
        const double* X = Linv + (size_t)k * TS * TS;
        for (int c = lane; c < TS; c += 64) {
            double s = 0.0;
            for (int r = wave; r < TS; r += 4)
                if (r >= c) s += X[r * TS + c] * s_acc[r];
            s_part[wave][c] = s;
        }
        __syncthreads();
        if (tid < TS) x[k * TS + tid] = (s_part[0][tid] + s_part[1][tid]) + (s_part[2][tid] + s_part[3][tid]);
        __syncthreads();
        __threadfence_block();
    }
}

bool g_attr_done = false;

} // namespace

void chol_factor_solve(const CholPlan& p, double* tiles, double* Linv, const double* g, double* x, double lambda, int n,
                       int* status, void* st)
{
    hipStream_t s = static_cast<hipStream_t>(st);
    const int TS = p.TS, LD = TS + 2;
    const size_t lds2 = sizeof(double) * 2 * (size_t)TS * LD;
    if (!g_attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_potrf_inv), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_trsm), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm_upd), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        g_attr_done = true;
    }
    for (int k = 0; k < p.NT; ++k) {
        hipLaunchKernelGGL(k_potrf_inv, dim3(1), dim3(kBlock), lds2, s, tiles, Linv, p.h_diag_tile[k], k, TS, n, lambda, status);
        const int nt = p.h_col_ptr[k + 1] - p.h_col_ptr[k];
        if (nt > 0) hipLaunchKernelGGL(k_trsm, dim3(nt), dim3(kBlock), lds2, s, tiles, Linv, p.trsm_tile + p.h_col_ptr[k], k, TS, status);
        const int nu = p.h_upd_ptr[k + 1] - p.h_upd_ptr[k];
        if (nu > 0)
            hipLaunchKernelGGL(k_gemm_upd, dim3(nu), dim3(kBlock), lds2, s, tiles, p.upd_a + p.h_upd_ptr[k], p.upd_b + p.h_upd_ptr[k],
                               p.upd_c + p.h_upd_ptr[k], TS, status);
    }
    hipLaunchKernelGGL(k_chol_solve, dim3(1), dim3(kBlock), 0, s, tiles, Linv, g, x, p, status);
}

} // namespace svi
