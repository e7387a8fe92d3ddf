// snode_potrf.hip — prototype: dense Cholesky of one SUPERNODE (a chain of w tile columns = N16 blocks of 16) by ONE workgroup,
// the whole lower triangle resident in MFMA accumulator registers (16 waves, blocks dealt column-major).  Right-looking over
// block columns of 16:  (a) the wave that owns block (j,j) factorises it alone (four quads: 4x4 pivot chain per lane, one
// v_mfma_f64_16x16x4 per quad) and inverts it (4 -> 8 -> 16 on the matrix cores); (b) barrier; (c) every wave multiplies its
// blocks of column j by X_jj' (L_ij = A_ij L_jj^-T) and publishes them; (d) barrier; (e) every wave subtracts L_ij L_i'j' from
// its trailing blocks.  Two barriers per 16 columns, nothing leaves the CU between the tile columns of the chain.
// Usage: snode_potrf <N16> [reps]   -> checks L against a host Cholesky, prints the kernel time
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double v4f64 __attribute__((ext_vector_type(4)));
constexpr int kThreads = 1024, NW = kThreads / 64, LD = 18;

__device__ __forceinline__ double fast_rcp(double x) { double r = __builtin_amdgcn_rcp(x); r = fma(r, fma(-x, r, 1.0), r); r = fma(r, fma(-x, r, 1.0), r); return r; }
__device__ __forceinline__ double fast_rsqrt(double x) { double r = __builtin_amdgcn_rsq(x); r = fma(r * 0.5, fma(-x * r, r, 1.0), r); r = fma(r * 0.5, fma(-x * r, r, 1.0), r); return r; }
#define WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)

// C = A B' for 16x16 LDS images (row stride LD), K = 16
__device__ __forceinline__ v4f64 mm16(const double* sA, const double* sB, v4f64 acc, bool neg)
{
    const int lane = threadIdx.x & 63;
    const double* pa = sA + (lane & 15) * LD + (lane >> 4);
    const double* pb = sB + (lane & 15) * LD + (lane >> 4);
    double a[4], b[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { a[q] = pa[4 * q]; b[q] = pb[4 * q]; }
    if (neg) {
#pragma unroll
        for (int q = 0; q < 4; ++q) a[q] = -a[q];
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q], b[q], acc, 0, 0, 0);
    return acc;
}
__device__ __forceinline__ void blk_to_lds(const v4f64& acc, double* s)
{
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int q = 0; q < 4; ++q) s[((lane >> 4) + 4 * q) * LD + (lane & 15)] = acc[q];
}

// Cholesky of the 16x16 block `acc` (accumulator layout: row (lane>>4)+4q, column lane&15) by ONE wave: on return acc = L (lower
// triangle valid), sXo = L^-1 (row-major image, zero above the diagonal).  s_w: wave-private scratch of >= 3 * 16 * LD doubles.
__device__ __forceinline__ bool potrf16(v4f64& acc, double* s_w, double* sXo)
{
    const int lane = threadIdx.x & 63, ln = lane & 15, lk = lane >> 4;
    double (*col)[4] = reinterpret_cast<double (*)[4]>(s_w);          // [16][4]
    double* s_d = s_w + 64;                                            // [16] pivots, then 1/sqrt
    double* s_cf = s_w + 80;                                           // [4][6]
    double* sL = s_w + 16 * LD;                                        // [16][LD] L image
    bool fail = false;
#pragma unroll
    for (int jq = 0; jq < 4; ++jq) {
        const int jx = 4 * jq;
        if ((ln >> 2) == jq) {
#pragma unroll
            for (int q = 0; q < 4; ++q) col[lk + 4 * q][ln & 3] = acc[q];
        }
        WAVE_SYNC();
        double w[4][4], l[4][4], rinv[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
#pragma unroll
            for (int q = 0; q <= m; ++q) {
                double v = col[jx + m][q];
#pragma unroll
                for (int t = 0; t < q; ++t) v = fma(-w[m][t], l[q][t], v);
                w[m][q] = v;
                if (q < m) l[m][q] = v * rinv[q];
            }
            const double dm = w[m][m];
            if (!(dm > 0.0)) fail = true;
            rinv[m] = fast_rcp(dm);
        }
        const double c10 = -l[1][0], c21 = -l[2][1], c32 = -l[3][2];
        const double c20 = fma(-l[2][1], c10, -l[2][0]), c31 = fma(-l[3][2], c21, -l[3][1]);
        const double c30 = fma(-l[3][2], c20, fma(-l[3][1], c10, -l[3][0]));
        const double cl0 = lk == 0 ? 1.0 : lk == 1 ? c10 : lk == 2 ? c20 : c30;
        const double cl1 = lk == 0 ? 0.0 : lk == 1 ? 1.0 : lk == 2 ? c21 : c31;
        const double cl2 = lk < 2 ? 0.0 : lk == 2 ? 1.0 : c32;
        const double cl3 = lk == 3 ? 1.0 : 0.0;
        const double rk = -(lk == 0 ? rinv[0] : lk == 1 ? rinv[1] : lk == 2 ? rinv[2] : rinv[3]);
        const double2 a01 = *reinterpret_cast<const double2*>(&col[ln][0]), a23 = *reinterpret_cast<const double2*>(&col[ln][2]);
        const double wb = fma(cl1, a01.y, cl0 * a01.x) + fma(cl3, a23.y, cl2 * a23.x); //  w[lk] at row ln
        const double wa = wb * rk;                                                        // -w[lk] / d[lk]
        const bool keep = ln > jx + lk; // rows / columns up to the pivot stay as they are
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(keep ? wa : 0.0, keep ? wb : 0.0, acc, 0, 0, 0);
        if (lane == 0) {
#pragma unroll
            for (int m = 0; m < 4; ++m) s_d[jx + m] = w[m][m];
            double* cf = s_cf + 6 * jq;
            cf[0] = c10; cf[1] = c20; cf[2] = c21; cf[3] = c30; cf[4] = c31; cf[5] = c32;
        }
        WAVE_SYNC(); // (col is rewritten by the next quad)
    }
    if (lane < 16) s_d[lane] = fast_rsqrt(s_d[lane]);
    for (int i = lane; i < 16 * LD; i += 64) sXo[i] = 0.0;
    WAVE_SYNC();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int r = lk + 4 * q;
        const double v = acc[q] * s_d[ln];
        acc[q] = r >= ln ? v : 0.0;
        sL[r * LD + ln] = acc[q];
    }
    // X = L^-1: 4x4 diagonal blocks from the quads' unit factors (L_qq = Lu D^1/2, so L_qq^-1 = D^-1/2 Lu^-1)
    if (lane < 16) {
        const int o = (lane >> 2) * 4, c = lane & 3;
        const double* cf = s_cf + 6 * (lane >> 2);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const double v = r < c ? 0.0 : r == c ? 1.0 : cf[r * (r - 1) / 2 + c];
            sXo[(o + r) * LD + o + c] = v * s_d[o + r];
        }
    }
    WAVE_SYNC();
    {   // 4 -> 8: X21 = -X22 (L21 X11) of the two blocks of eight, v_mfma_f64_4x4x4_4b (four 4x4x4 products per instruction)
        const int blk = (lane >> 2) & 3, hi = lane >> 4, lo = lane & 3;
        const bool live = blk < 2;
        const int o = 8 * (live ? blk : 0);
        const double a1 = sL[(o + 4 + lo) * LD + o + hi], b1 = sXo[(o + hi) * LD + o + lo];
        const double a2 = sXo[(o + 4 + lo) * LD + o + 4 + hi];
        const double t = __builtin_amdgcn_mfma_f64_4x4x4f64(a1, b1, 0.0, 0, 0, 0);
        const double r = __builtin_amdgcn_mfma_f64_4x4x4f64(a2, t, 0.0, 0, 0, 0);
        WAVE_SYNC();
        if (live) sXo[(o + 4 + hi) * LD + o + lo] = -r;
    }
    WAVE_SYNC();
    {   // 8 -> 16: rows 8..15 of (L X) over k < 8, then X22 times that
        const int m = lane & 15, kq = lane >> 4;
        v4f64 t = {0.0, 0.0, 0.0, 0.0}, r = {0.0, 0.0, 0.0, 0.0};
        double a[2], bb[2], a2[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            a[q] = m >= 8 ? sL[m * LD + 4 * q + kq] : 0.0;
            bb[q] = sXo[(4 * q + kq) * LD + m];
            a2[q] = sXo[m * LD + 8 + 4 * q + kq];
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) t = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q], bb[q], t, 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 2; ++q) r = __builtin_amdgcn_mfma_f64_16x16x4f64(a2[q], t[2 + q], r, 0, 0, 0);
        WAVE_SYNC();
        if (m < 8) {
#pragma unroll
            for (int q = 2; q < 4; ++q) sXo[(kq + 4 * q) * LD + m] = -r[q];
        }
    }
    return !fail;
}

template <int N16>
__global__ __launch_bounds__(kThreads) void k_snode(const double* __restrict__ A, double* __restrict__ Lout, double* __restrict__ Xout, int* status, long long* cycles)
{
    constexpr int N = 16 * N16, NBLK = N16 * (N16 + 1) / 2, MAXB = (NBLK + NW - 1) / NW;
    extern __shared__ __align__(16) double sm[];
    double* sX = sm;                                   // [16][LD] inverse of the current diagonal block
    double* sP = sX + 16 * LD;                         // [2][N16][16][LD] published block column
    double* sW = sP + 2 * N16 * 16 * LD;               // [NW][3 * 16 * LD] wave-private scratch
    __shared__ int s_fail;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, ln = lane & 15, lk = lane >> 4;
    double* s_w = sW + wave * 3 * 16 * LD;
    if (tid == 0) s_fail = 0;
    const long long t0 = clock64();
    // blocks of the lower triangle in column-major order go round the waves
    int bi[MAXB], bj[MAXB];
    v4f64 acc[MAXB];
#pragma unroll
    for (int u = 0; u < MAXB; ++u) {
        const int idx = wave + NW * u;
        int j = 0, rem = idx;
        while (j < N16 && rem >= N16 - j) { rem -= N16 - j; ++j; }
        const bool valid = idx < NBLK;
        bj[u] = valid ? j : -1; bi[u] = valid ? j + rem : -1;
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[u][q] = valid ? A[(size_t)(16 * bi[u] + lk + 4 * q) * N + 16 * bj[u] + ln] : 0.0;
    }
    __syncthreads();
    for (int j = 0; j < N16; ++j) {
        double* P = sP + (size_t)(j & 1) * N16 * 16 * LD;
        // (a) the diagonal block, by its owner alone (selected into one register set: potrf16 is instantiated once)
        {
            v4f64 dg = {0.0, 0.0, 0.0, 0.0};
            bool mine = false;
#pragma unroll
            for (int u = 0; u < MAXB; ++u)
                if (bi[u] == j && bj[u] == j) { dg = acc[u]; mine = true; } // wave-uniform
            if (mine) {
                if (!potrf16(dg, s_w, sX) && lane == 0) s_fail = 1;
#pragma unroll
                for (int u = 0; u < MAXB; ++u)
                    if (bi[u] == j && bj[u] == j) acc[u] = dg;
            }
        }
        __syncthreads();
        if (s_fail) break;
        // (c) L_ij = A_ij X_jj' for the blocks of column j, published
#pragma unroll
        for (int u = 0; u < MAXB; ++u)
            if (bj[u] == j && bi[u] > j) {
                blk_to_lds(acc[u], s_w);
                WAVE_SYNC();
                v4f64 z = {0.0, 0.0, 0.0, 0.0};
                acc[u] = mm16(s_w, sX, z, false);
                blk_to_lds(acc[u], P + bi[u] * 16 * LD);
                WAVE_SYNC();
            }
        __syncthreads();
        // (e) trailing update
#pragma unroll
        for (int u = 0; u < MAXB; ++u)
            if (bj[u] > j) acc[u] = mm16(P + bi[u] * 16 * LD, P + bj[u] * 16 * LD, acc[u], true);
    }
    if (tid == 0) { *status = s_fail; cycles[0] = clock64() - t0; }
#pragma unroll
    for (int u = 0; u < MAXB; ++u)
        if (bi[u] >= 0) {
#pragma unroll
            for (int q = 0; q < 4; ++q) Lout[(size_t)(16 * bi[u] + lk + 4 * q) * N + 16 * bj[u] + ln] = acc[u][q];
        }
    (void)Xout;
}

template <int N16> int run(int reps)
{
    constexpr int N = 16 * N16;
    std::vector<double> h((size_t)N * N), L((size_t)N * N, 0.0), ref((size_t)N * N, 0.0);
    srand(1);
    std::vector<double> B((size_t)N * N);
    for (auto& v : B) v = (rand() / (double)RAND_MAX) - 0.5;
    for (int r = 0; r < N; ++r)
        for (int c = 0; c < N; ++c) { double s = 0; for (int k = 0; k < N; ++k) s += B[(size_t)r * N + k] * B[(size_t)c * N + k]; h[(size_t)r * N + c] = s + (r == c ? N * 0.05 : 0.0); }
    for (int j = 0; j < N; ++j) { // host reference
        double d = h[(size_t)j * N + j];
        for (int k = 0; k < j; ++k) d -= ref[(size_t)j * N + k] * ref[(size_t)j * N + k];
        ref[(size_t)j * N + j] = sqrt(d);
        for (int i = j + 1; i < N; ++i) { double s = h[(size_t)i * N + j]; for (int k = 0; k < j; ++k) s -= ref[(size_t)i * N + k] * ref[(size_t)j * N + k]; ref[(size_t)i * N + j] = s / ref[(size_t)j * N + j]; }
    }
    double *dA, *dL, *dX; int* dst; long long* dcy;
    (void)hipMalloc(&dA, sizeof(double) * N * N); (void)hipMalloc(&dL, sizeof(double) * N * N); (void)hipMalloc(&dX, sizeof(double) * N16 * 256);
    (void)hipMalloc(&dst, 4); (void)hipMalloc(&dcy, 8);
    (void)hipMemcpy(dA, h.data(), sizeof(double) * N * N, hipMemcpyHostToDevice);
    (void)hipMemset(dL, 0, sizeof(double) * N * N);
    const size_t lds = sizeof(double) * (16 * LD + 2 * N16 * 16 * LD + NW * 3 * 16 * LD);
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_snode<N16>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) { printf("LDS request %zu refused\n", lds); return 1; }
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k_snode<N16>, dim3(1), dim3(kThreads), lds, 0, dA, dL, dX, dst, dcy);
    (void)hipEventRecord(a, 0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k_snode<N16>, dim3(1), dim3(kThreads), lds, 0, dA, dL, dX, dst, dcy);
    (void)hipEventRecord(b, 0); (void)hipEventSynchronize(b);
    float ms = 0; (void)hipEventElapsedTime(&ms, a, b);
    int st = 0; long long cy = 0;
    (void)hipMemcpy(L.data(), dL, sizeof(double) * N * N, hipMemcpyDeviceToHost);
    (void)hipMemcpy(&st, dst, 4, hipMemcpyDeviceToHost); (void)hipMemcpy(&cy, dcy, 8, hipMemcpyDeviceToHost);
    double err = 0, mx = 0;
    for (int r = 0; r < N; ++r) for (int c = 0; c <= r; ++c) { err = fmax(err, fabs(L[(size_t)r * N + c] - ref[(size_t)r * N + c])); mx = fmax(mx, fabs(ref[(size_t)r * N + c])); }
    printf("N16 %d (n = %d): status %d, max |L - ref| %.3e (max |L| %.3f), kernel %.2f us per launch, %lld shader cycles in the factorisation, LDS %zu B\n", N16, N, st, err, mx,
           1e3 * ms / reps, cy, lds);
    return 0;
}
int main(int argc, char** argv)
{
    const int n16 = argc > 1 ? atoi(argv[1]) : 12, reps = argc > 2 ? atoi(argv[2]) : 200;
    if (n16 == 3) return run<3>(reps);
    if (n16 == 6) return run<6>(reps);
    if (n16 == 12) return run<12>(reps);
    if (n16 == 15) return run<15>(reps);
    if (n16 == 18) return run<18>(reps);
    printf("N16 must be 3, 6, 12, 15 or 18\n");
    return 1;
}
