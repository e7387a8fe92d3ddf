// ba_kernels.hip — bundle-adjustment sweeps for gfx950 (MI355X): linearisation, Schur reduction,
// back-substitution, update and chi2.  Replaces what g2o does inside
//   SparseOptimizer::computeActiveErrors / BlockSolver::buildSystem / SparseOptimizer::update
// when driven by Cg2oOptimizer::_optimizeUnLimited (src/optimization/Cg2oOptimizer.cpp:954-980)
// with the solver stack configured at Cg2oOptimizer.cpp:83-89.  Edge semantics follow the
// reference's factories (Cg2oOptimizer.cpp:982-1073) and the g2o slam3d types as restated in
// SURVEY.md Appendix B.
//
// All of these kernels are HBM-streaming FP64 code: SoA edge arrays read with one coalesced 8 B
// load per lane and plane, outputs written as planes, vertex data (poses: 96 B, landmarks: 24 B)
// served from L1/L2.  Reductions are two-stage and fixed-order (per-workgroup partials, then one
// workgroup), so results do not depend on scheduling; the only atomics are LDS adds inside the
// Schur workgroups.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "ba_device.h"
#include "ba_math.h"

namespace svi {
namespace {

constexpr int kBlock = 256;

// ---------------------------------------------------------------------------------------------
// block-wide sum of NV doubles held by every thread; result valid in thread 0. Fixed order.
template <int NV, int NW = 4>
__device__ __forceinline__ void block_sum(double (&v)[NV], double* s_red /* [NV][NW] */);

template <int NW = 4>
__device__ __forceinline__ double block_max(double x, double* s_red /* [NW] */)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x = fmax(x, __shfl_down(x, off, 64));
    if (lane == 0) s_red[wave] = x;
    __syncthreads();
    double r = s_red[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) r = fmax(r, s_red[w]);
    __syncthreads();
    return r;
}

// the per-wave partials of the landmark-major kernels ([entry][4] doubles): a thread takes entries tid, tid + NT, ...,
// one 32-byte load each, kRedBatch of them in flight together
constexpr int kRedThreads = 1024, kRedBatch = 14;
struct alignas(32) Part4 { double v[4]; };
template <typename F>
__device__ __forceinline__ void for_each_part(const double* __restrict__ block_part, int n_part, F&& f)
{
    const Part4* __restrict__ src = reinterpret_cast<const Part4*>(block_part);
    for (int base = 0; base < n_part; base += kRedBatch * kRedThreads) {
        Part4 e[kRedBatch];
#pragma unroll
        for (int i = 0; i < kRedBatch; ++i) {
            const int b = base + (int)threadIdx.x + i * kRedThreads;
            e[i] = src[b < n_part ? b : 0];
        }
#pragma unroll
        for (int i = 0; i < kRedBatch; ++i)
            if (base + (int)threadIdx.x + i * kRedThreads < n_part) f(e[i]);
    }
}

// sum over the 16 lanes of a DPP row (every lane of the row gets the same bits): rotate-and-add butterfly
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double x)
{
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double row16_sum(double x)
{
    x += dpp_mov<0x128>(x); // row_ror:8
    x += dpp_mov<0x124>(x); // row_ror:4
    x += dpp_mov<0x122>(x); // row_ror:2
    x += dpp_mov<0x121>(x); // row_ror:1
    return x;
}

__device__ __forceinline__ double readlane_f64(double x, int lane)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), lane), __builtin_amdgcn_readlane(__double2loint(x), lane));
}
// whole-wave sum / max in a fixed order, result in every lane (all 64 lanes must be active)
__device__ __forceinline__ double wave_sum(double x)
{
    x = row16_sum(x);
    return (readlane_f64(x, 0) + readlane_f64(x, 16)) + (readlane_f64(x, 32) + readlane_f64(x, 48));
}
__device__ __forceinline__ double wave_max(double x)
{
    x = fmax(x, dpp_mov<0x128>(x));
    x = fmax(x, dpp_mov<0x124>(x));
    x = fmax(x, dpp_mov<0x122>(x));
    x = fmax(x, dpp_mov<0x121>(x));
    return fmax(fmax(readlane_f64(x, 0), readlane_f64(x, 16)), fmax(readlane_f64(x, 32), readlane_f64(x, 48)));
}

template <int NV, int NW>
__device__ __forceinline__ void block_sum(double (&v)[NV], double* s_red /* [NV][NW] */)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (NW == 4) { // the 256-thread kernels: shuffle tree per wave, the four wave sums added pairwise by thread 0
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            double x = v[k];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
            if (lane == 0) s_red[k * 4 + wave] = x;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
#pragma unroll
            for (int k = 0; k < NV; ++k) v[k] = (s_red[k * 4] + s_red[k * 4 + 1]) + (s_red[k * 4 + 2] + s_red[k * 4 + 3]);
        }
        __syncthreads();
    } else { // the 1024-thread reducers: DPP inside the wave, then the 16 wave sums by one DPP row of wave 0
        static_assert(NW == 4 || NW == 16, "block_sum: 4 or 16 waves");
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const double x = wave_sum(v[k]);
            if (lane == 0) s_red[k * NW + wave] = x;
        }
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int k = 0; k < NV; ++k) v[k] = row16_sum(s_red[k * NW + (lane & 15)]); // valid in every lane of wave 0
        }
        __syncthreads();
    }
}

struct EdgeIn {
    double z[3];
    double info[6]; // upper triangle 00 01 02 11 12 22
    int    type;
    bool   robust;
};

// DIAG: every information matrix of the graph is diagonal (the reference only sets diagonals,
// Cg2oOptimizer.cpp:1014,1038,1066): 3 planes are stored and the off-diagonal terms fold away at compile time
// plane access as (uniform base) + (32-bit byte offset of the lane): selects the SGPR-base addressing mode of the
// global memory instructions, i.e. no 64-bit address arithmetic on the vector ALU per plane
__device__ __forceinline__ double ldp(const double* __restrict__ base, unsigned byte_off)
{
    return *reinterpret_cast<const double*>(reinterpret_cast<const char*>(base) + (size_t)byte_off);
}
__device__ __forceinline__ void stp(double* __restrict__ base, unsigned byte_off, double v)
{
    *reinterpret_cast<double*>(reinterpret_cast<char*>(base) + (size_t)byte_off) = v;
}
// The per-edge operands N (9) and 2Z (3) live in SIX planes of double2: plane p holds values 2p, 2p+1 of edge e at
// NZ2[p E + e].  Writers and readers stay fully coalesced (16 B per lane) and need half the memory instructions of
// twelve 8-byte planes.
__device__ __forceinline__ double2 ldp2(const double* __restrict__ nz, size_t E, int p, unsigned e)
{
    return *reinterpret_cast<const double2*>(reinterpret_cast<const char*>(nz + 2 * (size_t)p * E) + (size_t)(e * 16u));
}
__device__ __forceinline__ void stp2(double* __restrict__ nz, size_t E, int p, unsigned e, double a, double b)
{
    *reinterpret_cast<double2*>(reinterpret_cast<char*>(nz + 2 * (size_t)p * E) + (size_t)(e * 16u)) = make_double2(a, b);
}

// measurement z (3) and information (diagonal: 3, else the 6 of the upper triangle) of an edge, packed in double2 planes
// like N|2Z: [z0 z1][z2 i0][i1 i2]( [i3 i4][i5 -] )
template <bool DIAG>
__device__ __forceinline__ void load_edge(const double* __restrict__ zi, const uint8_t* __restrict__ flags, int E, int e, EdgeIn& in)
{
    const double2 x0 = ldp2(zi, (size_t)E, 0, (unsigned)e), x1 = ldp2(zi, (size_t)E, 1, (unsigned)e), x2 = ldp2(zi, (size_t)E, 2, (unsigned)e);
    in.z[0] = x0.x; in.z[1] = x0.y; in.z[2] = x1.x;
    if (DIAG) {
        in.info[0] = x1.y; in.info[3] = x2.x; in.info[5] = x2.y;
        in.info[1] = 0.0; in.info[2] = 0.0; in.info[4] = 0.0;
    } else {
        const double2 x3 = ldp2(zi, (size_t)E, 3, (unsigned)e), x4 = ldp2(zi, (size_t)E, 4, (unsigned)e);
        in.info[0] = x1.y; in.info[1] = x2.x; in.info[2] = x2.y; in.info[3] = x3.x; in.info[4] = x3.y; in.info[5] = x4.x;
    }
    const unsigned f = flags[e];
    in.type = f & kFlagTypeMask;
    in.robust = (f & kFlagRobust) != 0;
}

// chi2 = e' Omega e ; weight rho1 and rho0 of g2o's RobustKernelCauchy
__device__ __forceinline__ double edge_chi2(const EdgeIn& in, const double* e)
{
    return e[0] * (in.info[0] * e[0] + 2.0 * (in.info[1] * e[1] + in.info[2] * e[2])) +
           e[1] * (in.info[3] * e[1] + 2.0 * in.info[4] * e[2]) + e[2] * in.info[5] * e[2];
}

// ---------------------------------------------------------------------------------------------
// K2: landmark-major Jacobian sweep.  One lane per edge, one block = a run of whole landmarks (<= 256 edges).
// Uses the structured form of ba_math.h (J = A [-I | 2[Z]x | R']): per edge it stores N = A'(rho1 Omega)A R' and
// Z (12 planes, fully coalesced) instead of the 6x3 block H_pl, and sums H_ll = sum R N, b_l = -sum R u per
// landmark in a fixed order through LDS.
// (Measured and dropped: a persistent variant that walks several blocks per workgroup and requests the operands of
// the next block before the tail of the current one - its register budget halves the occupancy, 38 us instead of 29.)
// ---------------------------------------------------------------------------------------------
template <bool DIAG>
struct LmFetch {
    int l0, nl, e0, e1;   // block: landmarks [l0, l0+nl), edges [e0, e1)
    EdgeIn in;            // this lane's edge (clamped into the block)
    int s, l;
    int ta0[2], ta1[2];   // edge ranges (relative to e0) of the landmarks of this lane's first two sum tasks
    bool tfix[2];
};

template <bool DIAG>
__device__ __forceinline__ void lm_fetch(const BaDev& d, int b, LmFetch<DIAG>& f)
{
    const int tid = threadIdx.x;
    const int4 rec = reinterpret_cast<const int4*>(d.lb_rec)[b];
    f.l0 = rec.x; f.nl = rec.y; f.e0 = rec.z; f.e1 = rec.w;
    const int e = min(f.e0 + tid, f.e1 - 1);
    load_edge<DIAG>(d.e_zi, d.e_flags, d.E, e, f.in);
    f.s = d.e_pose[e]; f.l = d.e_lm[e];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int task = tid + q * kBlock, l = f.l0 + min(task / 9, f.nl - 1);
        f.ta0[q] = d.lm_ptr[l] - f.e0; f.ta1[q] = d.lm_ptr[l + 1] - f.e0; f.tfix[q] = d.lm_fixed[l] != 0;
    }
}

#ifndef K2_WAVES
#define K2_WAVES 5 // (96 VGPRs, a few spilled values outside the hot path: one more wave per SIMD hides more of the gathers' latency, 27.9 -> 27.1 us)
#endif
template <bool DIAG>
__global__ __launch_bounds__(kBlock, K2_WAVES) void k_linearize_lm(BaDev d, int cur)
{
    constexpr int LDA = kLmBlockEdges + 1; // odd row stride: the nine rows of a landmark fall into different banks
    __shared__ double s_acc[9][LDA];
    const int tid = threadIdx.x;
    const double* __restrict__ pose = d.pose[cur];
    const double* __restrict__ lm = d.lm[cur];
    const int E = d.E, Ll = d.Ll;

    LmFetch<DIAG> f;
    const int b = blockIdx.x;
    lm_fetch<DIAG>(d, b, f);
    {
        const int l0 = f.l0, nl = f.nl, e0 = f.e0, e1 = f.e1;
        const int e = e0 + tid;
        double part[2] = {0.0, 0.0}; // robust chi2, plain chi2
        double acc9[9];
        if (e < e1) {
            const EdgeIn& in = f.in;
            const int s = f.s, l = f.l;
            double R[9], t[3], p[3];
            {   // the pose record (96 bytes, 16-byte aligned) as six 16-byte gathers instead of twelve 8-byte ones
                const double2* __restrict__ pr = reinterpret_cast<const double2*>(pose + 12 * (size_t)s);
                const double2 q0 = pr[0], q1 = pr[1], q2 = pr[2], q3 = pr[3], q4 = pr[4], q5 = pr[5];
                R[0] = q0.x; R[1] = q0.y; R[2] = q1.x; R[3] = q1.y; R[4] = q2.x; R[5] = q2.y; R[6] = q3.x; R[7] = q3.y; R[8] = q4.x;
                t[0] = q4.y; t[1] = q5.x; t[2] = q5.y;
            }
#pragma unroll
            for (int k = 0; k < 3; ++k) p[k] = lm[3 * l + k];
            const bool lfix = d.lm_fixed[l] != 0;
            double err[3], Z[3], A5[5];
            proj_core(in.type, R, t, p, in.z, d.fx, d.fy, d.cx, d.cy, err, Z, A5);
            const double c2 = edge_chi2(in, err);
            double w = 1.0, r0 = c2;
            if (in.robust) cauchy(d.cauchy_delta, c2, r0, w);
            part[0] = r0; part[1] = c2;
            double O[6], C[6], u[3];
#pragma unroll
            for (int k = 0; k < 6; ++k) O[k] = w * in.info[k];
            proj_cu(A5, O, err, C, u);
            // N = C R'  (C symmetric: c00 c01 c02 c11 c12 c22)
            const double Cf[9] = {C[0], C[1], C[2], C[1], C[3], C[4], C[2], C[4], C[5]};
            double N[9];
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int c = 0; c < 3; ++c) N[3 * r + c] = Cf[3 * r] * R[3 * c] + Cf[3 * r + 1] * R[3 * c + 1] + Cf[3 * r + 2] * R[3 * c + 2];
            double nz[12]; // N (zero for a fixed landmark), then 2Z: every consumer needs K = 2[Z]x
#pragma unroll
            for (int k = 0; k < 9; ++k) nz[k] = lfix ? 0.0 : N[k];
#pragma unroll
            for (int k = 0; k < 3; ++k) nz[9 + k] = 2.0 * Z[k];
#pragma unroll
            for (int pp = 0; pp < 6; ++pp) stp2(d.NZ, (size_t)E, pp, (unsigned)e, nz[2 * pp], nz[2 * pp + 1]);
            // H_ll contribution R N (symmetric, upper 00 01 02 11 12 22) and b_l = -R u
            int k = 0;
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int c = a; c < 3; ++c, ++k) acc9[k] = R[3 * a] * N[c] + R[3 * a + 1] * N[3 + c] + R[3 * a + 2] * N[6 + c];
#pragma unroll
            for (int a = 0; a < 3; ++a) acc9[6 + a] = -(R[3 * a] * u[0] + R[3 * a + 1] * u[1] + R[3 * a + 2] * u[2]);
        }
        if (e < e1) {
#pragma unroll
            for (int k = 0; k < 9; ++k) s_acc[k][tid] = acc9[k];
        }
        const int ta0[2] = {f.ta0[0], f.ta0[1]}, ta1[2] = {f.ta1[0], f.ta1[1]};
        const bool tfix[2] = {f.tfix[0], f.tfix[1]};
        __syncthreads();
        // per-landmark sums, one (landmark, value) pair per lane: fixed order over the landmark's edges
        double mx = 0.0;
        auto lm_store = [&](int task, double sum, bool fixed) {
            const int li = task / 9, k = task - 9 * li, l = l0 + li;
            if (fixed) sum = 0.0;
            if (k < 6) d.Hll[(size_t)k * Ll + l] = sum; else d.bl[(size_t)(k - 6) * Ll + l] = sum;
            if (k == 0 || k == 3 || k == 5) mx = fmax(mx, fabs(sum));
        };
        {
            // two tasks side by side, four LDS reads each in flight (the additions keep the edge order of the landmark)
            int a[2], end[2];
            const double* row[2];
            double sum[2] = {0.0, 0.0};
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int task = tid + q * kBlock;
                const bool on = task < nl * 9;
                a[q] = on ? ta0[q] : 0; end[q] = on ? ta1[q] : 0;
                row[q] = s_acc[task % 9];
            }
            while (a[0] < end[0] || a[1] < end[1]) {
                double v[2][4];
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[q][i] = row[q][min(a[q] + i, kLmBlockEdges - 1)];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (a[q] + i < end[q]) sum[q] += v[q][i];
                    a[q] += 4;
                }
            }
#pragma unroll
            for (int q = 0; q < 2; ++q)
                if (tid + q * kBlock < nl * 9) lm_store(tid + q * kBlock, sum[q], tfix[q]);
        }
        for (int task = tid + 2 * kBlock; task < nl * 9; task += kBlock) {
            const int li = task / 9, k = task - 9 * li, l = l0 + li;
            double sum = 0.0;
            const int a1 = d.lm_ptr[l + 1] - e0;
            for (int a = d.lm_ptr[l] - e0; a < a1; ++a) sum += s_acc[k][a];
            lm_store(task, sum, d.lm_fixed[l] != 0);
        }
        // landmark-closure priors (EdgePointXYZ with a fixed partner, Cg2oOptimizer.cpp:448-458): rare, one lane per landmark
        if (d.n_lmlm > 0) {
            __syncthreads();
            const int l = l0 + tid;
            if (tid < nl && d.lm_ll_ptr[l + 1] > d.lm_ll_ptr[l] && !d.lm_fixed[l]) {
                double acc[9];
#pragma unroll
                for (int k = 0; k < 6; ++k) acc[k] = d.Hll[(size_t)k * Ll + l];
#pragma unroll
                for (int k = 0; k < 3; ++k) acc[6 + k] = d.bl[(size_t)k * Ll + l];
                for (int q = d.lm_ll_ptr[l]; q < d.lm_ll_ptr[l + 1]; ++q) {
                    double ee[3], O[6];
#pragma unroll
                    for (int k = 0; k < 3; ++k) ee[k] = lm[3 * l + k] - d.ll_ref[3 * q + k] - d.ll_z[3 * q + k];
#pragma unroll
                    for (int k = 0; k < 6; ++k) O[k] = d.ll_info[6 * q + k];
                    const double c2 = ee[0] * (O[0] * ee[0] + 2.0 * (O[1] * ee[1] + O[2] * ee[2])) +
                                      ee[1] * (O[3] * ee[1] + 2.0 * O[4] * ee[2]) + ee[2] * O[5] * ee[2];
                    double w = 1.0, r0 = c2;
                    if (d.ll_robust[q]) cauchy(d.cauchy_delta, c2, r0, w);
                    part[0] += r0; part[1] += c2;
#pragma unroll
                    for (int k = 0; k < 6; ++k) acc[k] += w * O[k];
                    acc[6] -= w * (O[0] * ee[0] + O[1] * ee[1] + O[2] * ee[2]);
                    acc[7] -= w * (O[1] * ee[0] + O[3] * ee[1] + O[4] * ee[2]);
                    acc[8] -= w * (O[2] * ee[0] + O[4] * ee[1] + O[5] * ee[2]);
                }
#pragma unroll
                for (int k = 0; k < 6; ++k) d.Hll[(size_t)k * Ll + l] = acc[k];
#pragma unroll
                for (int k = 0; k < 3; ++k) d.bl[(size_t)k * Ll + l] = acc[6 + k];
                mx = fmax(mx, fmax(fabs(acc[0]), fmax(fabs(acc[3]), fabs(acc[5]))));
            }
        }
        // two sums and one max per WAVE, in registers (DPP): no barrier, no serial tail; lin_part holds one entry per wave
        {
            const double x0 = wave_sum(part[0]), x1 = wave_sum(part[1]), x2 = wave_max(mx);
            if ((tid & 63) == 0) {
                double* out = d.lin_part + 4 * (size_t)(4 * b + (tid >> 6));
                out[0] = x0; out[1] = x1; out[2] = x2;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// K3: pose-major Jacobian sweep: one workgroup per chunk of <= kPoseChunk edges of ONE pose.
// H_pp = sum [ C , -C K ; K C , -K C K ],  b_p = sum [ u ; K u ],  K = 2[Z]x  (21 + 6 values), summed
// in a fixed order through a small LDS buffer, nine values per pass.
// ---------------------------------------------------------------------------------------------
template <bool DIAG>
__device__ __forceinline__ void sweep_pose_chunk(const BaDev& d, int cur, int c, double* smem)
{
    double (*s_part)[kBlock + 1] = reinterpret_cast<double (*)[kBlock + 1]>(smem);
    const int tid = threadIdx.x;
    const int s = d.chunk_pose[c];
    const int e0 = d.chunk_begin[c], e1 = d.chunk_begin[c + 1];
    const double* __restrict__ pose = d.pose[cur];
    const double* __restrict__ lm = d.lm[cur];
    const int E = d.E;
    double R[9], t[3];
#pragma unroll
    for (int k = 0; k < 9; ++k) R[k] = pose[12 * s + k];
#pragma unroll
    for (int k = 0; k < 3; ++k) t[k] = pose[12 * s + 9 + k];

    double acc[27];
#pragma unroll
    for (int k = 0; k < 27; ++k) acc[k] = 0.0;
    // software pipeline over the (at most U) edges of a lane: the landmark indices of all of them first, then the
    // operands of edge u+1 travel while edge u is computed.  Prefetch addresses are clamped into the chunk, so
    // the loads need no branch; the accumulation order per lane is unchanged.
    constexpr int U = kPoseChunk / kBlock;
    int li[U];
#pragma unroll
    for (int it = 0; it < U; ++it) li[it] = d.pm_lm[min(e0 + tid + it * kBlock, e1 - 1)];
    EdgeIn nxt;
    double pn[3];
    load_edge<DIAG>(d.pm_zi, d.pm_flags, E, min(e0 + tid, e1 - 1), nxt);
#pragma unroll
    for (int k = 0; k < 3; ++k) pn[k] = lm[3 * li[0] + k];
#pragma unroll
    for (int it = 0; it < U; ++it) {
        const EdgeIn in = nxt;
        const double p[3] = {pn[0], pn[1], pn[2]};
        if (it + 1 < U) {
            load_edge<DIAG>(d.pm_zi, d.pm_flags, E, min(e0 + tid + (it + 1) * kBlock, e1 - 1), nxt);
#pragma unroll
            for (int k = 0; k < 3; ++k) pn[k] = lm[3 * li[it + 1] + k];
        }
        if (e0 + tid + it * kBlock >= e1) continue;
        double err[3], Z[3], A5[5];
        proj_core(in.type, R, t, p, in.z, d.fx, d.fy, d.cx, d.cy, err, Z, A5);
        double w = 1.0, r0;
        if (in.robust) cauchy(d.cauchy_delta, edge_chi2(in, err), r0, w);
        double O[6], C[6], u[3];
#pragma unroll
        for (int k = 0; k < 6; ++k) O[k] = w * in.info[k];
        proj_cu(A5, O, err, C, u);
        const double Cf[9] = {C[0], C[1], C[2], C[1], C[3], C[4], C[2], C[4], C[5]};
        const double z0 = 2.0 * Z[0], z1 = 2.0 * Z[1], z2 = 2.0 * Z[2];
        // G = C K (3x3), K = [[0,-z2,z1],[z2,0,-z0],[-z1,z0,0]]
        double G[9];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            G[3 * r]     = Cf[3 * r + 1] * z2 - Cf[3 * r + 2] * z1;
            G[3 * r + 1] = Cf[3 * r + 2] * z0 - Cf[3 * r] * z2;
            G[3 * r + 2] = Cf[3 * r] * z1 - Cf[3 * r + 1] * z0;
        }
        // Q = K' G = -K G (symmetric 3x3): rows of K times G
        const double q00 = -(-z2 * G[3] + z1 * G[6]), q01 = -(-z2 * G[4] + z1 * G[7]), q02 = -(-z2 * G[5] + z1 * G[8]);
        const double q11 = -(z2 * G[1] - z0 * G[7]), q12 = -(z2 * G[2] - z0 * G[8]);
        const double q22 = -(-z1 * G[2] + z0 * G[5]);
        // upper triangle of the 6x6, row-major: rows 0-2 = [C | -G], rows 3-5 = [. | Q]
        acc[0] += C[0];  acc[1] += C[1];  acc[2] += C[2];  acc[3] -= G[0];  acc[4] -= G[1];  acc[5] -= G[2];
        acc[6] += C[3];  acc[7] += C[4];  acc[8] -= G[3];  acc[9] -= G[4];  acc[10] -= G[5];
        acc[11] += C[5]; acc[12] -= G[6]; acc[13] -= G[7]; acc[14] -= G[8];
        acc[15] += q00;  acc[16] += q01;  acc[17] += q02;
        acc[18] += q11;  acc[19] += q12;
        acc[20] += q22;
        // b_p = [ u ; K u ]
        acc[21] += u[0]; acc[22] += u[1]; acc[23] += u[2];
        acc[24] += -z2 * u[1] + z1 * u[2];
        acc[25] += z2 * u[0] - z0 * u[2];
        acc[26] += -z1 * u[0] + z0 * u[1];
    }
    // 27 sums over the 256 lanes, nine per pass: value k is summed by the 16 lanes of one DPP row (lane q takes
    // the entries q, q+16, ...: consecutive lanes, consecutive banks), the row total by a rotate-and-add butterfly
#pragma unroll
    for (int pass = 0; pass < 3; ++pass) {
        if (pass) __syncthreads();
#pragma unroll
        for (int k = 0; k < 9; ++k) s_part[k][tid] = acc[9 * pass + k];
        __syncthreads();
        if (tid < 9 * 16) {
            const int k = tid >> 4, q = tid & 15;
            double v[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = s_part[k][16 * i + q];
            double sum = v[0];
#pragma unroll
            for (int i = 1; i < 16; ++i) sum += v[i];
            sum = row16_sum(sum);
            if (q == 0) d.chunk_out[(size_t)27 * c + 9 * pass + k] = sum;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// The Jacobian sweep = two launches on the same stream: landmark-major blocks (K2) and pose-major chunks (K3).
// (One fused launch was measured: the two kinds do not fill each other's gaps, and the fused kernel inherits the
// register budget of the pipelined K3 loop, which halves the occupancy of the K2 blocks.)
// ---------------------------------------------------------------------------------------------
#ifndef K3_WAVES
#define K3_WAVES 1
#endif
template <bool DIAG>
__global__ __launch_bounds__(kBlock, K3_WAVES) void k_linearize_pose(BaDev d, int cur)
{
    __shared__ double smem[9 * (kBlock + 1)];
    sweep_pose_chunk<DIAG>(d, cur, blockIdx.x, smem);
}

// ---------------------------------------------------------------------------------------------
// pose-only edges (odometry EdgeSE3, gravity EdgeSE3LinearAcceleration): a single workgroup,
// LINEARIZE: also the quadratic forms.  Writes aux chi2 (robust, plain) to d.scal[6], d.scal[7].
// ---------------------------------------------------------------------------------------------
// EdgeSE3LinearAcceleration: e = R (R_off a) - (0,0,-1) with v = R_off a folded on the host (edge_se3_linear_acceleration.cpp:
// 106-116; the cache's n2w() is the pose times the offset parameter).  J (3x6, nullable): d e / d dt = 0, d e / d dq = -2 R [v]x
// - analytic where g2o differentiates numerically (BaseUnaryEdge::linearizeOplus, central differences with 1e-9).
__device__ __forceinline__ void accel_edge_eval(const double* R, const double* v, double* e, double* J)
{
    e[0] = R[0] * v[0] + R[1] * v[1] + R[2] * v[2];
    e[1] = R[3] * v[0] + R[4] * v[1] + R[5] * v[2];
    e[2] = R[6] * v[0] + R[7] * v[1] + R[8] * v[2] + 1.0;
    if (!J) return;
    const double S[9] = {0, -v[2], v[1], v[2], 0, -v[0], -v[1], v[0], 0};
    for (int r = 0; r < 3; ++r) {
        J[6 * r] = J[6 * r + 1] = J[6 * r + 2] = 0.0;
        for (int c = 0; c < 3; ++c) J[6 * r + 3 + c] = -2.0 * (R[3 * r] * S[c] + R[3 * r + 1] * S[3 + c] + R[3 * r + 2] * S[6 + c]);
    }
}

constexpr int kAuxThreads = 64; // one edge per lane, one wavefront per workgroup: the edges spread over many CUs

template <bool LINEARIZE>
__global__ __launch_bounds__(kAuxThreads) void k_aux_edges(BaDev d, int which)
{
    const double* __restrict__ pose = d.pose[which];
    double part[2] = {0.0, 0.0};
    const int gid = blockIdx.x * kAuxThreads + threadIdx.x;
    for (int k = gid; k < d.n_se3; k += gridDim.x * kAuxThreads) {
        const int si = d.se3_i[k], sj = d.se3_j[k];
        double e[6], Ji[36], Jj[36], O[36];
        se3_edge_eval(pose + 12 * si, pose + 12 * sj, d.se3_Z + 12 * k, e, LINEARIZE ? Ji : nullptr, Jj);
        int q = 0;
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
            for (int c = r; c < 6; ++c, ++q) O[6 * r + c] = O[6 * c + r] = d.se3_info[21 * k + q];
        double c2 = 0.0;
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
            for (int c = 0; c < 6; ++c) c2 += e[r] * O[6 * r + c] * e[c];
        double w = 1.0, r0 = c2;
        if (d.se3_robust[k]) cauchy(d.cauchy_delta, c2, r0, w);
        part[0] += r0; part[1] += c2;
        if (LINEARIZE) {
            double* out = d.se3_out + (size_t)120 * k;
            double Oe[6], OJi[36], OJj[36];
#pragma unroll
            for (int r = 0; r < 6; ++r) {
                double sacc = 0.0;
#pragma unroll
                for (int c = 0; c < 6; ++c) sacc += w * O[6 * r + c] * e[c];
                Oe[r] = sacc;
#pragma unroll
                for (int c = 0; c < 6; ++c) {
                    double a = 0.0, b = 0.0;
#pragma unroll
                    for (int m = 0; m < 6; ++m) { a += w * O[6 * r + m] * Ji[6 * m + c]; b += w * O[6 * r + m] * Jj[6 * m + c]; }
                    OJi[6 * r + c] = a; OJj[6 * r + c] = b;
                }
            }
#pragma unroll
            for (int r = 0; r < 6; ++r)
#pragma unroll
                for (int c = 0; c < 6; ++c) {
                    double hii = 0.0, hjj = 0.0, hij = 0.0;
#pragma unroll
                    for (int m = 0; m < 6; ++m) {
                        hii += Ji[6 * m + r] * OJi[6 * m + c];
                        hjj += Jj[6 * m + r] * OJj[6 * m + c];
                        hij += Ji[6 * m + r] * OJj[6 * m + c];
                    }
                    out[6 * r + c] = hii; out[36 + 6 * r + c] = hjj; out[72 + 6 * r + c] = hij;
                }
#pragma unroll
            for (int r = 0; r < 6; ++r) {
                double bi = 0.0, bj = 0.0;
#pragma unroll
                for (int m = 0; m < 6; ++m) { bi += Ji[6 * m + r] * Oe[m]; bj += Jj[6 * m + r] * Oe[m]; }
                out[108 + r] = -bi; out[114 + r] = -bj;
            }
        }
    }
    for (int k = gid; k < d.n_accel; k += gridDim.x * kAuxThreads) {
        const double* R = pose + 12 * d.acc_pose[k];
        const double v[3] = {d.acc_a[3 * k], d.acc_a[3 * k + 1], d.acc_a[3 * k + 2]};
        double e[3], J[18];
        accel_edge_eval(R, v, e, LINEARIZE ? J : nullptr);
        double O[6];
#pragma unroll
        for (int q = 0; q < 6; ++q) O[q] = d.acc_info[6 * k + q];
        const double c2 = e[0] * (O[0] * e[0] + 2.0 * (O[1] * e[1] + O[2] * e[2])) + e[1] * (O[3] * e[1] + 2.0 * O[4] * e[2]) +
                          e[2] * O[5] * e[2];
        part[0] += c2; part[1] += c2;
        if (LINEARIZE) {
            const double Of[9] = {O[0], O[1], O[2], O[1], O[3], O[4], O[2], O[4], O[5]};
            double* out = d.acc_out + (size_t)42 * k;
#pragma unroll
            for (int r = 0; r < 6; ++r) {
#pragma unroll
                for (int c = 0; c < 6; ++c) {
                    double h = 0.0;
#pragma unroll
                    for (int m = 0; m < 3; ++m)
#pragma unroll
                        for (int n = 0; n < 3; ++n) h += J[6 * m + r] * Of[3 * m + n] * J[6 * n + c];
                    out[6 * r + c] = h;
                }
                double bb = 0.0;
#pragma unroll
                for (int m = 0; m < 3; ++m)
#pragma unroll
                    for (int n = 0; n < 3; ++n) bb += J[6 * m + r] * Of[3 * m + n] * e[n];
                out[36 + r] = -bb;
            }
        }
    }
    // per-workgroup partials, then the LAST workgroup to arrive adds them in workgroup order (deterministic)
#pragma unroll
    for (int q = 0; q < 2; ++q)
        for (int off = 32; off > 0; off >>= 1) part[q] += __shfl_xor(part[q], off);
    __shared__ int s_last;
    if (threadIdx.x == 0) {
        d.aux_part[2 * blockIdx.x] = part[0];
        d.aux_part[2 * blockIdx.x + 1] = part[1];
        __threadfence();
        s_last = (atomicAdd(d.aux_count, 1) == (int)gridDim.x - 1) ? 1 : 0;
    }
    __syncthreads();
    if (s_last && threadIdx.x == 0) {
        __threadfence();
        double r0 = 0.0, r1 = 0.0;
        for (unsigned b = 0; b < gridDim.x; ++b) {
            r0 += __builtin_nontemporal_load(d.aux_part + 2 * b);
            r1 += __builtin_nontemporal_load(d.aux_part + 2 * b + 1);
        }
        // the sums of the linearisation point have slots of their own: they are read after the trial has written 6 and 7
        d.scal[LINEARIZE ? 12 : 6] = r0; d.scal[LINEARIZE ? 13 : 7] = r1;
        *d.aux_count = 0;
    }
}

// chi2 (robust, plain) of the pose-only edges at `pose`, edges k = first, first + stride, ... (the trial's closing reduction
// evaluates them itself: a launch of its own costs more than the thousand edges)
__device__ __forceinline__ void aux_edges_chi2(const BaDev& d, const double* __restrict__ pose, int first, int stride, double (&part)[2])
{
    for (int k = first; k < d.n_se3; k += stride) {
        const int si = d.se3_i[k], sj = d.se3_j[k];
        double e[6], O[36];
        se3_edge_eval(pose + 12 * si, pose + 12 * sj, d.se3_Z + 12 * k, e, nullptr, nullptr);
        int q = 0;
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
            for (int c = r; c < 6; ++c, ++q) O[6 * r + c] = O[6 * c + r] = d.se3_info[21 * k + q];
        double c2 = 0.0;
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
            for (int c = 0; c < 6; ++c) c2 += e[r] * O[6 * r + c] * e[c];
        double w = 1.0, r0 = c2;
        if (d.se3_robust[k]) cauchy(d.cauchy_delta, c2, r0, w);
        part[0] += r0; part[1] += c2;
    }
    for (int k = first; k < d.n_accel; k += stride) {
        const double* R = pose + 12 * d.acc_pose[k];
        const double v[3] = {d.acc_a[3 * k], d.acc_a[3 * k + 1], d.acc_a[3 * k + 2]};
        double e[3];
        accel_edge_eval(R, v, e, nullptr);
        double O[6];
#pragma unroll
        for (int q = 0; q < 6; ++q) O[q] = d.acc_info[6 * k + q];
        const double c2 = e[0] * (O[0] * e[0] + 2.0 * (O[1] * e[1] + O[2] * e[2])) + e[1] * (O[3] * e[1] + 2.0 * O[4] * e[2]) +
                          e[2] * O[5] * e[2];
        part[0] += c2; part[1] += c2;
    }
}

// H_pp / b_p of every free pose = sum of its chunk partials (fixed order) + its pose-only edges.
__device__ __forceinline__ void pose_finalize_block(const BaDev& d, const int* __restrict__ red_slot, int block)
{
    const int idx = block * kBlock + threadIdx.x;
    if (idx >= d.Pf * 27) return;
    const int r = idx / 27, v = idx % 27;
    const int s = red_slot[r];
    // position of upper-triangle entry v in a full 6x6
    int rr = 0, cc = 0;
    if (v < 21) {
        int k = v;
        rr = 0;
        while (k >= 6 - rr) { k -= 6 - rr; ++rr; }
        cc = rr + k;
    }
    // Dependent round trips are what this launch costs: the four list bounds travel together, then the first two chunk sums
    // and the first three pose-only edge references (a pose of a trajectory has a chunk or two, two odometry edges and a
    // gravity edge), then those edges' values - four trips where the plain loops made seven.  Sums in list order as before.
    const int c0 = d.pose_chunk_ptr[s], c1 = d.pose_chunk_ptr[s + 1], q0 = d.pose_aux_ptr[s], q1 = d.pose_aux_ptr[s + 1];
    constexpr int PC = 2, PA = 3;
    double cv[PC];
    int ref[PA];
#pragma unroll
    for (int i = 0; i < PC; ++i) cv[i] = d.chunk_out[(size_t)27 * (c0 + i < c1 ? c0 + i : c0) + v];
#pragma unroll
    for (int i = 0; i < PA; ++i) ref[i] = d.pose_aux_ref[q0 + i < q1 ? q0 + i : q0];
    auto aux_ptr = [&](int rf) -> const double* { // (selects, no branches: the loads below must not sit in divergent regions)
        const int k = rf >> 2, role = rf & 3;
        const size_t oa = (size_t)42 * k + ((v < 21) ? 6 * rr + cc : 36 + (v - 21));
        const size_t os = (size_t)120 * k + ((v < 21) ? 36 * role + 6 * rr + cc : 108 + 6 * role + (v - 21));
        const double* pa = d.acc_out + oa;
        const double* ps = d.se3_out + os;
        return role == 2 ? pa : ps;
    };
    double av[PA];
#pragma unroll
    for (int i = 0; i < PA; ++i) { const double* pp = aux_ptr(ref[i]); av[i] = *((q0 + i < q1) ? pp : d.scal); }
    double sum = 0.0;
#pragma unroll
    for (int i = 0; i < PC; ++i) if (c0 + i < c1) sum += cv[i];
    for (int c = c0 + PC; c < c1; ++c) sum += d.chunk_out[(size_t)27 * c + v];
#pragma unroll
    for (int i = 0; i < PA; ++i) if (q0 + i < q1) sum += av[i];
    for (int q = q0 + PA; q < q1; ++q) sum += *aux_ptr(d.pose_aux_ref[q]);
    if (v < 21) d.Hpp[(size_t)21 * r + v] = sum;
    else d.bp[(size_t)6 * r + (v - 21)] = sum;
}
__global__ __launch_bounds__(kBlock) void k_pose_finalize(BaDev d, const int* __restrict__ red_slot)
{
    pose_finalize_block(d, red_slot, blockIdx.x);
}

// chi2 (robust, plain) of the linearisation point and this rank's max |diag H_ll|.
__global__ __launch_bounds__(kRedThreads) void k_reduce_lin_scalars(BaDev d, int rank, int n_ranks)
{
    constexpr int NW = kRedThreads / 64;
    __shared__ double s_red[2 * NW];
    double part[2] = {0.0, 0.0};
    double mx = 0.0;
    for_each_part(d.lin_part, 4 * d.n_lm_blocks, [&](const Part4& e) { part[0] += e.v[0]; part[1] += e.v[1]; mx = fmax(mx, e.v[2]); });
    block_sum<2, NW>(part, s_red);
    mx = block_max<NW>(mx, s_red);
    if (threadIdx.x == 0) {
        d.lin_scal[0] = part[0] + d.scal[12];
        d.lin_scal[1] = part[1] + d.scal[13];
        for (int k = 0; k < n_ranks; ++k) d.lin_scal[2 + k] = (k == rank) ? mx : 0.0;
    }
}

// max |H_jj| over all free vertices (g2o computeLambdaInit) -> scal[5]; also copies chi2 to scal[0..1].
// SINGLE (one rank, nothing to exchange in between): the sums of k_reduce_lin_scalars are done here as well.
template <bool SINGLE>
__global__ __launch_bounds__(kRedThreads) void k_lin_post(BaDev d, int n_ranks)
{
    constexpr int NW = kRedThreads / 64;
    __shared__ double s_red[2 * NW];
    double mx = 0.0;
    double part[2] = {0.0, 0.0};
    // the sums of the pose-only edges (or, with several ranks, the exchanged totals): fetched first, used last
    const double t0 = SINGLE ? d.scal[12] : d.lin_scal[0], t1 = SINGLE ? d.scal[13] : d.lin_scal[1];
    if (SINGLE) for_each_part(d.lin_part, 4 * d.n_lm_blocks, [&](const Part4& e) { part[0] += e.v[0]; part[1] += e.v[1]; mx = fmax(mx, e.v[2]); });
    for (int i = threadIdx.x; i < d.Pf * 6; i += kRedThreads) {
        const int r = i / 6, a = i % 6;
        // diagonal entry a of the upper-triangle packing: index = a*6 - a(a-1)/2
        mx = fmax(mx, fabs(d.Hpp[(size_t)21 * r + (a * 6 - a * (a - 1) / 2)]));
    }
    if (SINGLE) block_sum<2, NW>(part, s_red);
    else for (int k = threadIdx.x; k < n_ranks; k += kRedThreads) mx = fmax(mx, d.lin_scal[2 + k]);
    mx = block_max<NW>(mx, s_red);
    // slots 8..10 carry the same numbers and survive the trial kernels (which rewrite 0..3): the host may pick them up
    // together with the trial results instead of waiting here
    if (threadIdx.x == 0) {
        const double c0 = SINGLE ? part[0] + t0 : t0, c1 = SINGLE ? part[1] + t1 : t1;
        d.scal[5] = mx; d.scal[0] = c0; d.scal[1] = c1;
        d.scal[8] = c0; d.scal[9] = c1; d.scal[10] = mx;
    }
}

// ---------------------------------------------------------------------------------------------
// K4a: (H_ll + lambda I)^-1 per landmark (upper triangle). A non positive definite block marks the
// trial as failed, like the factorisation of the full system would.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void invert_landmarks_block(const BaDev& d, double lambda, int block, double (*s_rec)[9])
{
    const int l0 = block * kBlock, tid = threadIdx.x, l = l0 + tid;
    const size_t Ll = d.Ll;
    if (l < d.Ll) {
        double inv[6] = {0, 0, 0, 0, 0, 0};
        if (!d.lm_fixed[l]) {
            const double a = d.Hll[l] + lambda, b = d.Hll[Ll + l], c = d.Hll[2 * Ll + l];
            const double e = d.Hll[3 * Ll + l] + lambda, f = d.Hll[4 * Ll + l], i = d.Hll[5 * Ll + l] + lambda;
            const double c00 = e * i - f * f, c01 = c * f - b * i, c02 = b * f - c * e;
            const double det = a * c00 + b * c01 + c * c02;
            const double m1 = a * e - b * b;
            if (!(a > 0.0) || !(m1 > 0.0) || !(det > 0.0)) { *d.chol_status = -1; }
            else {
                const double id = 1.0 / det;
                inv[0] = c00 * id; inv[1] = c01 * id; inv[2] = c02 * id;
                inv[3] = (a * i - c * c) * id; inv[4] = (b * c - a * f) * id; inv[5] = m1 * id;
            }
        }
#pragma unroll
        for (int k = 0; k < 6; ++k) { d.Hinv[(size_t)k * Ll + l] = inv[k]; s_rec[tid][k] = inv[k]; }
#pragma unroll
        for (int k = 0; k < 3; ++k) s_rec[tid][6 + k] = d.bl[(size_t)k * Ll + l];
    }
    __syncthreads();
    // one contiguous record per landmark for the Schur staging (a single cache line instead of nine planes),
    // written out coalesced: the records of this workgroup's landmarks are one contiguous range
    const int nv = min(kBlock, d.Ll - l0);
    double* out = d.HinvB + (size_t)12 * l0;
    for (int i = tid; i < 12 * nv; i += kBlock) {
        const int r = i / 12, k = i - 12 * r;
        if (k < 9) out[i] = s_rec[r][k];
    }
}
__global__ __launch_bounds__(kBlock) void k_invert_landmarks(BaDev d, double lambda)
{
    __shared__ double s_rec[kBlock][9]; // odd row stride: conflict-free both ways
    invert_landmarks_block(d, lambda, blockIdx.x, s_rec);
}
// The first trial of an iteration whose lambda is already known (every iteration but the first of a block): the landmark
// blocks are inverted by extra workgroups of the launch that sums the pose blocks - the two do not depend on each other, and a
// launch of its own costs more than either.
__global__ __launch_bounds__(kBlock) void k_finalize_invert(BaDev d, const int* __restrict__ red_slot, int fin_blocks, double lambda)
{
    __shared__ double s_rec[kBlock][9];
    if ((int)blockIdx.x < fin_blocks) pose_finalize_block(d, red_slot, blockIdx.x);
    else invert_landmarks_block(d, lambda, (int)blockIdx.x - fin_blocks, s_rec);
}

// ---------------------------------------------------------------------------------------------
// K4: Schur reduction on CELLS of 4 x 4 poses (24 x 24 entries of S).  One WAVEFRONT per job = four quarter jobs
// (runs of items of one cell each); lane = (quarter, i, j) = (lane>>4, (lane>>2)&3, lane&3) owns the 6x6 block of
// pose pair (4 cX + i, 4 cY + j) of its quarter's cell and accumulates it in registers: for every item (landmark)
// of the quarter job whose masks contain both poses,
//     block += (W_a Hinv_l) W_b'        (a, b = the landmark's edges to those two poses)
// No atomics, fixed summation order.  Diagonal cells keep the lower blocks only and carry the right-hand side
// g_i += (W_a Hinv_l) b_l.  The kernel is FP64-instruction bound: four-pose cells give 61 % of the
// lanes work at KITTI-like co-visibility where eight-pose cells gave 36 %.
// ---------------------------------------------------------------------------------------------
#ifndef SCHUR_ABL
#define SCHUR_ABL 0
#endif
constexpr int kSchurBatch = 2;        // passes (one item per quarter) staged per wave and buffer
constexpr int kSchurSlot  = 115;      // doubles per staged item: 8 edge slots x (N 9, Z 3), Hinv(6), b_l(3), pad; odd
                                      // multiple chosen so that the 16 (quarter, pose) operand rows sit in 16 different bank pairs

struct SchurStage { double v[kSchurBatch][6]; double hv[kSchurBatch]; int m[kSchurBatch]; };

// item records of the passes [base, base + kSchurBatch) of this lane's quarter job (zero mask beyond its end)
__device__ __forceinline__ void schur_fetch_items(const int4* __restrict__ items, int it0, int it1, int base, int4 (&pk)[kSchurBatch])
{
#pragma unroll
    for (int t = 0; t < kSchurBatch; ++t) {
        const int it = it0 + base + t;
        pk[t] = items[it < it1 ? it : it0];
        if (it >= it1) pk[t].w = 0;
    }
}

// issue the operand loads of the passes whose item records are `pk`.  Staging role of a lane inside its quarter:
// edge slot es (0..3 row segment, 4..7 column segment) and planes 6 pg .. 6 pg + 5 of the 12 (N, Z) planes.
__device__ __forceinline__ void schur_fetch(const BaDev& d, const int4 (&pk)[kSchurBatch], bool diag, int lane, SchurStage& st)
{
    const int es = lane & 7, pg = (lane >> 3) & 1, l16 = lane & 15;
    const size_t E = d.E;
    const double* __restrict__ nzp = d.NZ;
#pragma unroll
    for (int t = 0; t < kSchurBatch; ++t) {
        st.m[t] = pk[t].w;
        const unsigned mI = (unsigned)pk[t].w & 0xFu, mJ = ((unsigned)pk[t].w >> 8) & 0xFu;
        const int nI = __popc(mI), nJ = diag ? 0 : __popc(mJ);
        // unconditional loads from a clamped edge (edge 0 for an empty slot): no divergent regions around the loads,
        // and no masking either - the multiply only ever reads the slots its masks name
        const bool row = es < 4;
        const bool on = row ? (es < nI) : (es - 4 < nJ);
        const unsigned e = on ? (unsigned)((row ? pk[t].y : pk[t].z - 4) + es) : 0u;
#pragma unroll
        for (int q = 0; q < 3; ++q) { // the lane's six values are three double2 planes
#if SCHUR_ABL == 3 || SCHUR_ABL == 4
            // (ablation: the same bytes addressed as 96-byte records per edge - what an array-of-records layout would cost to stage)
            const double2 x = *reinterpret_cast<const double2*>(nzp + (size_t)e * 12 + 6 * pg + 2 * q);
#else
            const double2 x = ldp2(nzp, E, 3 * pg + q, e);
#endif
            st.v[t][2 * q] = x.x; st.v[t][2 * q + 1] = x.y;
        }
        st.hv[t] = d.HinvB[(size_t)12 * (unsigned)pk[t].x + (l16 < 9 ? l16 : 0)];
    }
}

__device__ __forceinline__ void schur_stash(const SchurStage& st, int lane, double (*slots)[4][kSchurSlot])
{
    const int es = lane & 7, pg = (lane >> 3) & 1, l16 = lane & 15, qt = lane >> 4;
#pragma unroll
    for (int t = 0; t < kSchurBatch; ++t) {
#pragma unroll
        for (int q = 0; q < 6; ++q) slots[t][qt][es * 12 + 6 * pg + q] = st.v[t][q];
        if (l16 < 9) slots[t][qt][96 + l16] = st.hv[t];
    }
}

// a slab value leaves the XCD's L2 at once (write-through): the kernel that sums the slabs of a STAGE starts, on another stream,
// while this kernel is still running - no kernel boundary writes the L2 back for it (MI355X_MICROARCH.md, publish-large:
// write-through stores + a drained counter beat plain stores + a release fence)
__device__ __forceinline__ void slab_store(double* p, double v, bool through)
{
    if (through) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}

// agent-scope load (bypasses the CU's L1): data another workgroup of the SAME launch has stored write-through
__device__ __forceinline__ double ld_agent(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// One cell (24 x 24 entries: 4 x 4 poses) of a stored sub-tile, by ONE wave: pose terms minus the slabs of the cell's quarter jobs
// in list order (a fixed order: the result does not depend on which wave gets here), identity padding, and - for a diagonal cell -
// its 24 rows of g.  Called by the wave whose slab was the last of the cell to arrive (or, for a cell without slabs, by the wave
// the cell is dealt to); the slabs of the other waves were stored write-through and are read past the L1.
// A function of its own (inlined into k_schur it costs that kernel its second wave per SIMD: 256 VGPRs).  It reads the layout from
// the copy of BaDev that lives in device memory (d.self): the kernel's own argument handed over by reference would be copied to
// scratch memory and every argument of the kernel indexed there.  The three switches the host sets per trial travel as `flags`.
__device__ __noinline__ void assemble_cell(const BaDev* __restrict__ dp, int flags, int cell, int lane, bool through)
{
    const BaDev& d = *dp;
    const bool add_pose_terms = (flags & 1) != 0, add_aux_blocks = (flags & 2) != 0;
    constexpr int NE = 9; // 36 x 16 values of the cell's slabs / 64 lanes
    const int sub = cell >> 2, u = (cell >> 1) & 1, vv = cell & 1;
    const int cx = d.sub_cx[sub], cy = d.sub_cy[sub], n = 6 * d.Pf, TS = d.TS;
    const int x0 = d.cell_qj_ptr[cell], x1 = d.cell_qj_ptr[cell + 1];
    double v[NE];
#pragma unroll
    for (int k = 0; k < NE; ++k) v[k] = 0.0;
    // the list is walked eight slabs at a time (72 loads in flight per lane: a slab that another XCD stored write-through comes
    // from memory, ~2 us away - one slab after the other the hot diagonal cells, forty pieces, cost the whole kernel 100 us),
    // subtracted in list order
#ifndef SCHUR_ASM_NB
#define SCHUR_ASM_NB 2
#endif
    constexpr int NB = SCHUR_ASM_NB;
    for (int x = x0; x < x1; x += NB) {
        double w[NB][NE];
#pragma unroll
        for (int t = 0; t < NB; ++t) {
            const int qa = d.cell_qj[min(x + t, x1 - 1)];
            const double* sa = d.slab + (size_t)(qa >> 2) * 36 * 64 + (qa & 3) * 16;
#pragma unroll
            for (int k = 0; k < NE; ++k) { const int e = lane + 64 * k; w[t][k] = ld_agent(sa + (e >> 4) * 64 + (e & 15)); }
        }
#pragma unroll
        for (int t = 0; t < NB; ++t)
            if (x + t < x1) {
#pragma unroll
                for (int k = 0; k < NE; ++k) v[k] -= w[t][k];
            }
    }
    const bool diag_cell = cx == cy && u == vv;
    const int R0 = cx * 48, C0 = cy * 48;
    double* out = d.S + (size_t)d.sub_tile[sub] * TS * TS + (size_t)(R0 % TS) * TS + (C0 % TS);
    const int a0 = add_aux_blocks ? d.sub_aux_ptr[sub] : 0, a1 = add_aux_blocks ? d.sub_aux_ptr[sub + 1] : 0;
    // pose terms of the diagonal blocks (requested together, in front of the odometry blocks' index chains)
    double hp[NE];
#pragma unroll
    for (int k = 0; k < NE; ++k) {
        const int e = lane + 64 * k, q = e >> 4, l16 = e & 15, bi = l16 >> 2, bj = l16 & 3, rr = q / 6, cc = q % 6;
        const int r = cx * 8 + 4 * u + bi;
        const int a = rr < cc ? rr : cc, b = rr < cc ? cc : rr;
        hp[k] = (add_pose_terms && diag_cell && bi == bj && r < d.Pf) ? d.Hpp[(size_t)21 * r + (a * 6 - a * (a - 1) / 2) + (b - a)] : 0.0;
    }
    for (int x = a0; x < a1; ++x) { // odometry blocks of this sub-tile (a handful): each one lands on the lanes that hold its block
        const int ref = d.sub_aux_ref[x], ke = ref >> 1, tr = ref & 1;
        const int ri = d.pose_red[d.se3_i[ke]], rj = d.pose_red[d.se3_j[ke]];
        const int rhi = tr ? rj : ri, rlo = tr ? ri : rj;
        const int l16 = lane & 15;
        if ((rhi % 8) != 4 * u + (l16 >> 2) || (rlo % 8) != 4 * vv + (l16 & 3)) continue; // (the block of a lane is the same for all its nine values)
#pragma unroll
        for (int k = 0; k < NE; ++k) {
            const int q = (lane + 64 * k) >> 4, rr = q / 6, cc = q % 6;
            hp[k] += tr ? d.se3_out[(size_t)120 * ke + 72 + 6 * cc + rr] : d.se3_out[(size_t)120 * ke + 72 + 6 * rr + cc];
        }
    }
#pragma unroll
    for (int k = 0; k < NE; ++k) {
        const int e = lane + 64 * k, q = e >> 4, l16 = e & 15, bi = l16 >> 2, bj = l16 & 3, rr = q / 6, cc = q % 6;
        const int r = (4 * u + bi) * 6 + rr, c = (4 * vv + bj) * 6 + cc;
        double val = v[k] + hp[k];
        if (R0 + r >= n || C0 + c >= n) val = (R0 + r == C0 + c) ? 1.0 : 0.0; // identity padding
        slab_store(out + (size_t)r * TS + c, val, through);
    }
    if (diag_cell && lane < 24) { // right-hand side of the cell's four poses
        const int bi = lane / 6, comp = lane % 6, row = R0 + (4 * u + bi) * 6 + comp;
        double g = 0.0;
        if (row < n) {
            if (add_pose_terms) g = d.bp[row];
            for (int x = x0; x < x1; x += 16) {
                double g4[16];
#pragma unroll
                for (int t = 0; t < 16; ++t) g4[t] = ld_agent(d.gslab + ((size_t)d.cell_qj[min(x + t, x1 - 1)] * 6 + comp) * 4 + bi);
#pragma unroll
                for (int t = 0; t < 16; ++t) if (x + t < x1) g -= g4[t];
            }
        }
        slab_store(d.g + row, g, through);
    }
}

// reserve_per_se > 0 (staged launches beside a running factorisation): workgroups that land on compute units 2 .. 2 + reserve_per_se - 1
// of a shader engine leave at once - those CUs stay EMPTY for the factorisation's workgroups (which fit neither beside two Schur
// workgroups nor, by registers, beside one) and its pivot chain shares no SIMD with a Schur wave.  The jobs are therefore not tied to
// block indices in a staged launch: a workgroup draws the next group of sixteen quarter jobs of ITS XCD (HW_REG_XCC_ID, not
// blockIdx % 8) from a ticket counter per stage and XCD until none is left.
__global__ __launch_bounds__(kBlock) void k_schur(BaDev d, int stage0, int stage1, StageSignals sg, int reserve_per_se)
{
    // The operands of kSchurBatch passes are staged in a wave-private LDS region, double buffered: the
    // global loads of batch k+1 are in flight while batch k is being multiplied out of LDS.
    // Block of pose pair (a,b) for one landmark, with M = N_a Hinv N_b' (3x3), Ka = 2[Z_a]x, Kb = 2[Z_b]x:
    //     H_pl,a Hinv H_pl,b' = [ M , -M Kb ; Ka M , -Ka M Kb ]
    // Stages: a wave walks its job of stage0, then of stage0 + 1, ...; leaving a stage it drains its slab stores and arrives at
    // the stage's counter, the last arrival publishes the stage (sg) - the consumers of a stage never wait for a later one.
    __shared__ double s_stage[kBlock / 64][2][kSchurBatch][4][kSchurSlot];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int qt = lane >> 4, i = (lane >> 2) & 3, j = lane & 3;
    const int4* __restrict__ items = reinterpret_cast<const int4*>(d.it_pack);
    const unsigned below_i = (1u << i) - 1u, below_j = (1u << j) - 1u;
    const bool through = stage1 - stage0 > 1 || d.asm_in_schur != 0;
    const bool asm_here = d.asm_in_schur == 1; // (2: ablation - write-through slabs, but the k_assemble launch sums them)
    // several ranks, pose sums not exchanged on their own: this rank's chi2 of the linearisation rides in front of g
    // (rewritten per trial: the all-reduce leaves the total there)
    if (asm_here && d.lin_from_red && blockIdx.x == 0 && threadIdx.x == 0) { d.red_base[0] = d.lin_scal[0]; d.red_base[1] = d.lin_scal[1]; }
    const bool ticketed = stage1 - stage0 > 1;
    int xcc = (int)blockIdx.x & 7;
    if (ticketed) {
        unsigned hw = 0, xc = 0;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xc));
        xcc = (int)(xc & 7u);
        const int cu = (int)((hw >> 8) & 0xFu);
        if (reserve_per_se > 0 && cu >= 2 && cu < 2 + reserve_per_se) return; // (the whole workgroup: it lives on one CU)
    }
    const int groups = d.n_jobs / 32; // groups of sixteen quarter jobs per XCD and stage
    int* const tickets = d.ticket + (size_t)(sg.seq & 1ull) * kMaxStages * 8;
    __shared__ int s_grp;
    for (int stage = stage0; stage < stage1; ++stage) {
    for (int round = 0;; ++round) {
    int grp = (int)blockIdx.x >> 3;
    if (ticketed) {
        __syncthreads(); // (the readers of the last ticket are done with it)
        if (threadIdx.x == 0) s_grp = __hip_atomic_fetch_add(tickets + stage * 8 + xcc, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        grp = s_grp;
        if (grp >= groups) break;
    } else if (round > 0) break;
    const int slot = (8 * grp + xcc) * (kBlock / 64) + wave; // job inside the stage
    const int job = stage * d.n_jobs + slot;
    int mycell = -1;   // (lanes 0, 16, 32, 48) the cell this quarter's slab belongs to; last: its arrival completed the cell
    bool last = false;
    const int it0 = d.qj_begin[4 * job + qt], it1 = d.qj_end[4 * job + qt];
    const bool diag = d.qj_diag[4 * job + qt] != 0;
    const int n_pass = d.job_len[job];
    if (n_pass > 0) { // (wave-uniform; an empty slot of a short list leaves no slab: no cell lists it)
    double acc[36], gacc[6];
#pragma unroll
    for (int q = 0; q < 36; ++q) acc[q] = 0.0;
#pragma unroll
    for (int q = 0; q < 6; ++q) gacc[q] = 0.0;
    // three stages in flight: item records of batch k+2, operands of batch k+1 (registers), batch k (LDS) being multiplied
    SchurStage st;
    int4 pk[kSchurBatch];
    int buf = 0;
    schur_fetch_items(items, it0, it1, 0, pk);
    schur_fetch(d, pk, diag, lane, st);
    schur_fetch_items(items, it0, it1, kSchurBatch, pk);
    // The compiler's scheduler otherwise undoes the software pipeline (it moves the loads of the next batch behind
    // the arithmetic of this one): 267 us instead of 130 us.  The barriers pin the order stash -> issue -> multiply.
#define SCHUR_PIN __builtin_amdgcn_sched_barrier(0);
    for (int base = 0; base < n_pass; base += kSchurBatch, buf ^= 1) {
        schur_stash(st, lane, s_stage[wave][buf]);
        SCHUR_PIN
        int mk[kSchurBatch];
#pragma unroll
        for (int t = 0; t < kSchurBatch; ++t) mk[t] = st.m[t];
        if (base + kSchurBatch < n_pass) {
#if SCHUR_ABL != 2
            schur_fetch(d, pk, diag, lane, st);
#endif
            schur_fetch_items(items, it0, it1, base + 2 * kSchurBatch, pk);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        SCHUR_PIN
#pragma unroll
        for (int t = 0; t < kSchurBatch; ++t) {
            const unsigned mI = (unsigned)mk[t] & 0xFu, mJ = ((unsigned)mk[t] >> 8) & 0xFu;
            const bool active = ((mI >> i) & 1u) && ((mJ >> j) & 1u) && (!diag || i >= j);
            if (!active) continue;
            const double* slot = s_stage[wave][buf][t][qt];
#if SCHUR_ABL == 1 || SCHUR_ABL == 3
            acc[0] += slot[12 * __popc(mI & below_i)] + slot[96]; continue;
#endif
            const double* na = slot + 12 * __popc(mI & below_i);
            const double* nbp = slot + 12 * (diag ? __popc(mJ & below_j) : 4 + __popc(mJ & below_j));
            const double* Hi = slot + 96;
            double T[9]; // N_a Hinv
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const double w0 = na[3 * r], w1 = na[3 * r + 1], w2 = na[3 * r + 2];
                T[3 * r]     = w0 * Hi[0] + w1 * Hi[1] + w2 * Hi[2];
                T[3 * r + 1] = w0 * Hi[1] + w1 * Hi[3] + w2 * Hi[4];
                T[3 * r + 2] = w0 * Hi[2] + w1 * Hi[4] + w2 * Hi[5];
            }
            double M[9]; // T N_b'
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const double w0 = nbp[3 * c], w1 = nbp[3 * c + 1], w2 = nbp[3 * c + 2];
#pragma unroll
                for (int r = 0; r < 3; ++r) M[3 * r + c] = T[3 * r] * w0 + T[3 * r + 1] * w1 + T[3 * r + 2] * w2;
            }
            const double a0 = na[9], a1 = na[10], a2 = na[11];   // 2 Z_a (stored doubled by the sweep)
            const double b0 = nbp[9], b1 = nbp[10], b2 = nbp[11]; // 2 Z_b
            // KM = Ka M
            double KM[9];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                KM[c]     = -a2 * M[3 + c] + a1 * M[6 + c];
                KM[3 + c] = a2 * M[c] - a0 * M[6 + c];
                KM[6 + c] = -a1 * M[c] + a0 * M[3 + c];
            }
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                // top rows: [ M | -M Kb ]   (each cross-product term is two fused multiply-adds into the accumulator)
                acc[6 * r]     += M[3 * r];
                acc[6 * r + 1] += M[3 * r + 1];
                acc[6 * r + 2] += M[3 * r + 2];
                acc[6 * r + 3] = fma(M[3 * r + 2], b1, fma(-M[3 * r + 1], b2, acc[6 * r + 3]));
                acc[6 * r + 4] = fma(M[3 * r], b2, fma(-M[3 * r + 2], b0, acc[6 * r + 4]));
                acc[6 * r + 5] = fma(M[3 * r + 1], b0, fma(-M[3 * r], b1, acc[6 * r + 5]));
                // bottom rows: [ Ka M | -Ka M Kb ]
                acc[6 * (3 + r)]     += KM[3 * r];
                acc[6 * (3 + r) + 1] += KM[3 * r + 1];
                acc[6 * (3 + r) + 2] += KM[3 * r + 2];
                acc[6 * (3 + r) + 3] = fma(KM[3 * r + 2], b1, fma(-KM[3 * r + 1], b2, acc[6 * (3 + r) + 3]));
                acc[6 * (3 + r) + 4] = fma(KM[3 * r], b2, fma(-KM[3 * r + 2], b0, acc[6 * (3 + r) + 4]));
                acc[6 * (3 + r) + 5] = fma(KM[3 * r + 1], b0, fma(-KM[3 * r], b1, acc[6 * (3 + r) + 5]));
            }
            if (diag && i == j) { // g_a += H_pl,a Hinv b_l = [ -w ; -Ka w ],  w = T b_l
                const double w0 = T[0] * Hi[6] + T[1] * Hi[7] + T[2] * Hi[8];
                const double w1 = T[3] * Hi[6] + T[4] * Hi[7] + T[5] * Hi[8];
                const double w2 = T[6] * Hi[6] + T[7] * Hi[7] + T[8] * Hi[8];
                gacc[0] -= w0; gacc[1] -= w1; gacc[2] -= w2;
                gacc[3] -= -a2 * w1 + a1 * w2;
                gacc[4] -= a2 * w0 - a0 * w2;
                gacc[5] -= -a1 * w0 + a0 * w1;
            }
        }
        SCHUR_PIN
        // the buffer written two iterations from now is this one: every lane is past its reads by then
        // (wave-synchronous execution, in-order LDS), the barrier below keeps the compiler honest
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        SCHUR_PIN
    }
    const bool merged = d.job_merged[job] != 0;
    if (merged) { // (wave-uniform) four pieces of one cell: (q0 + q1) + (q2 + q3) in every lane, read from quarter 0
#pragma unroll
        for (int q = 0; q < 36; ++q) { double v = acc[q]; v += __shfl_xor(v, 16); v += __shfl_xor(v, 32); acc[q] = v; }
#pragma unroll
        for (int q = 0; q < 6; ++q) { double v = gacc[q]; v += __shfl_xor(v, 16); v += __shfl_xor(v, 32); gacc[q] = v; }
    }
    double* out = d.slab + (size_t)job * 36 * 64;
    if (!merged || qt == 0) { // (a merged wave: only its first quarter is ever read)
#pragma unroll
        for (int q = 0; q < 36; ++q) slab_store(out + q * 64 + lane, acc[q], through);
    }
    if (diag && i == j && (!merged || qt == 0)) {
#pragma unroll
        for (int q = 0; q < 6; ++q) slab_store(d.gslab + ((size_t)(4 * job + qt) * 6 + q) * 4 + i, gacc[q], through);
    }
    if (asm_here) {
        // The slabs of this wave have left for memory; each of its (up to four) pieces is counted at its cell, and where the count
        // is complete this wave sums the cell: no k_assemble launch, no kernel boundary between reduction and factorisation.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if ((lane & 15) == 0) {
            mycell = d.qj_cell[4 * job + qt];
            if (mycell >= 0) {
                const int np = d.cell_qj_ptr[mycell + 1] - d.cell_qj_ptr[mycell];
                const int k = __hip_atomic_fetch_add(d.cell_count + mycell, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                last = k == np - 1;
                if (last) __hip_atomic_store(d.cell_count + mycell, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    }
    if (asm_here) {
        // cells to sum: those this wave's slabs completed, then its share of the stage's cells that no slab reaches (dealt round
        // the waves) - one loop, so that assemble_cell is inlined once
        unsigned long long todo = __ballot(last);
        int o = d.orphan_ptr[stage] + slot;
        const int o1 = d.orphan_ptr[stage + 1];
        for (;;) {
            int cell;
            if (todo) { const int src = __ffsll((long long)todo) - 1; todo &= todo - 1; cell = __shfl(mycell, src); }
            else if (o < o1) { cell = d.orphan_cell[o]; o += d.n_jobs; }
            else break;
            assemble_cell(d.self, (d.add_pose_terms ? 1 : 0) | (d.add_aux_blocks ? 2 : 0), cell, lane, through);
        }
    }
    if (ticketed) {
        // every store of this wave has left for memory before its job is counted; the last job of a stage (the counter goes back
        // to zero for the next launch, and so do the OTHER parity's tickets of this stage, which the next launch draws from) tells
        // whoever waits on the stage's value
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) {
            const int k = __hip_atomic_fetch_add(d.stage_count + stage, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (k == d.n_jobs - 1) {
                __hip_atomic_store(d.stage_count + stage, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                int* const other = d.ticket + (size_t)((sg.seq & 1ull) ^ 1ull) * kMaxStages * 8 + stage * 8;
#pragma unroll
                for (int x = 0; x < 8; ++x) __hip_atomic_store(other + x, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (sg.sig[stage]) __hip_atomic_store(sg.sig[stage], sg.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
    }
    }
}

// ---------------------------------------------------------------------------------------------
// K4b: one workgroup per stored 48 x 48 sub-tile of S: pose terms (H_pp diagonal blocks, odometry
// blocks) minus the quarter-job slabs of its four cells, summed in a fixed order, written into its tile.
// Diagonal sub-tiles also assemble their part of g and the identity padding of the last rows.
// ---------------------------------------------------------------------------------------------
constexpr int kAsmThreads = 512; // eight waves: the quarter-job slabs of a cell are shared by all of them

// sub0: first sub-tile of this launch (a stage's range, or 0).  accumulate: the tile and g were zeroed at the start of the trial
// and may already carry updates of the factorisation's earlier levels (a later stage is assembled while those run): add, do not store.
__global__ __launch_bounds__(kAsmThreads) void k_assemble(BaDev d, int sub0, int accumulate)
{
    __shared__ double s_t[48][49];
    constexpr int NWA = kAsmThreads / 64;
    __shared__ double s_part[NWA][36 * 16];
    const int sub = sub0 + (int)blockIdx.x, tid = threadIdx.x;
    const int cx = d.sub_cx[sub], cy = d.sub_cy[sub], n = 6 * d.Pf, TS = d.TS;
    // several ranks, pose sums not exchanged on their own: this rank's chi2 of the linearisation rides in front of g
    // (rewritten per trial: the all-reduce leaves the total there)
    if (d.lin_from_red && sub == 0 && tid == 0) { d.red_base[0] = d.lin_scal[0]; d.red_base[1] = d.lin_scal[1]; }
    // slabs: wave w of this workgroup sums cell (u, v) = (w >> 1, w & 1) of the sub-tile.  A quarter job qj left its
    // [36][16] values at slab[(qj >> 2)][q][(qj & 3) * 16 + l16]: block (i, j) = (l16 >> 2, l16 & 3), entry (q / 6, q % 6).
    {
        constexpr int NE = 36 * 16 / 64; // 9 elements per lane, all in flight for every quarter job
        const int w = tid >> 6, lane = tid & 63, u = w >> 1, vv = w & 1;
        // A hot cell can have forty quarter jobs and its neighbours none: the eight waves share the jobs of EVERY cell
        // (wave w takes the jobs w, w+8, ... of the cell's list), the partials meet in LDS and are added in wave
        // order by the wave that owns the cell - a fixed order, so the result stays reproducible.
        double v[NE];
#pragma unroll
        for (int k = 0; k < NE; ++k) v[k] = 0.0;
        // the list bounds of the four cells and this wave's first two job indices of each travel together, in front of the cell
        // loop: index -> slabs was two dependent round trips per cell, four cells one after the other
        int cp[5], qa0[4], qb0[4];
#pragma unroll
        for (int c = 0; c < 5; ++c) cp[c] = d.cell_qj_ptr[4 * sub + c];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int x = cp[c] + w, xe = cp[c + 1];
            qa0[c] = d.cell_qj[x < xe ? x : (xe > cp[c] ? xe - 1 : 0)];
            qb0[c] = d.cell_qj[x + NWA < xe ? x + NWA : (x < xe ? x : (xe > cp[c] ? xe - 1 : 0))];
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            double pv[NE];
#pragma unroll
            for (int k = 0; k < NE; ++k) pv[k] = 0.0;
            const int xe = cp[c + 1];
            for (int x = cp[c] + w; x < xe; x += 2 * NWA) { // two of this wave's jobs per step: 18 loads in flight
                const bool two = x + NWA < xe, first = x == cp[c] + w;
                const int qa = first ? qa0[c] : d.cell_qj[x], qb = first ? qb0[c] : d.cell_qj[two ? x + NWA : x];
                const double* sa = d.slab + (size_t)(qa >> 2) * 36 * 64 + (qa & 3) * 16;
                const double* sb = d.slab + (size_t)(qb >> 2) * 36 * 64 + (qb & 3) * 16;
                double wa[NE], wb[NE];
#pragma unroll
                for (int k = 0; k < NE; ++k) { const int e = lane + 64 * k; wa[k] = sa[(e >> 4) * 64 + (e & 15)]; wb[k] = sb[(e >> 4) * 64 + (e & 15)]; }
#pragma unroll
                for (int k = 0; k < NE; ++k) { pv[k] -= wa[k]; if (two) pv[k] -= wb[k]; }
            }
            if (cp[c + 1] - cp[c] > 0) { // block-uniform
#pragma unroll
                for (int k = 0; k < NE; ++k) s_part[w][lane + 64 * k] = pv[k];
                __syncthreads();
                if (w == c) {
#pragma unroll
                    for (int k = 0; k < NE; ++k) {
                        double t = s_part[0][lane + 64 * k];
#pragma unroll
                        for (int ww = 1; ww < NWA; ++ww) t += s_part[ww][lane + 64 * k];
                        v[k] = t;
                    }
                }
                __syncthreads();
            }
        }
        if (w < 4) { // the waves that own a cell (u, v) = (w >> 1, w & 1)
#pragma unroll
            for (int k = 0; k < NE; ++k) {
                const int e = lane + 64 * k, q = e >> 4, l16 = e & 15;
                s_t[(4 * u + (l16 >> 2)) * 6 + q / 6][(4 * vv + (l16 & 3)) * 6 + q % 6] = v[k];
            }
        }
    }
    __syncthreads();
    if (d.add_pose_terms) {
        if (cx == cy) {
            for (int e = tid; e < 8 * 36; e += kAsmThreads) {
                const int pl = e / 36, rr = (e % 36) / 6, cc = e % 6;
                const int r = cx * 8 + pl;
                if (r < d.Pf) {
                    const int a = rr < cc ? rr : cc, b = rr < cc ? cc : rr;
                    s_t[pl * 6 + rr][pl * 6 + cc] += d.Hpp[(size_t)21 * r + (a * 6 - a * (a - 1) / 2) + (b - a)];
                }
            }
        }
        __syncthreads();
    }
    if (d.add_aux_blocks) {
        for (int e = d.sub_aux_ptr[sub] * 36 + tid; e < d.sub_aux_ptr[sub + 1] * 36; e += kAsmThreads) {
            const int ref = d.sub_aux_ref[e / 36], k = ref >> 1, tr = ref & 1;
            const int rr = (e % 36) / 6, cc = e % 6;
            const int ri = d.pose_red[d.se3_i[k]], rj = d.pose_red[d.se3_j[k]];
            const int rhi = tr ? rj : ri, rlo = tr ? ri : rj; // row pose, column pose of the lower block
            const double v = tr ? d.se3_out[(size_t)120 * k + 72 + 6 * cc + rr] : d.se3_out[(size_t)120 * k + 72 + 6 * rr + cc];
            atomicAdd(&s_t[(rhi % 8) * 6 + rr][(rlo % 8) * 6 + cc], v);
        }
        __syncthreads();
    }
    const int R0 = cx * 48, C0 = cy * 48;
    double* out = d.S + (size_t)d.sub_tile[sub] * TS * TS + (size_t)(R0 % TS) * TS + (C0 % TS);
    for (int e = tid; e < 48 * 48; e += kAsmThreads) {
        const int r = e / 48, c = e % 48;
        double v = s_t[r][c];
        if (R0 + r >= n || C0 + c >= n) v = (R0 + r == C0 + c) ? 1.0 : 0.0; // identity padding
        if (accumulate) out[(size_t)r * TS + c] += v; else out[(size_t)r * TS + c] = v;
    }
    if (cx == cy) {
        // right-hand side: row r of the sub-tile collects the g slabs of its diagonal cell.  Five threads per row share
        // the list (thread part takes entries part, part+5, ..., four in flight), their partials meet in LDS and are
        // added in part order - a hot cell has forty quarter jobs, one thread per row walked them for 8 us.
        constexpr int GP = 5;
        static_assert(GP * 48 <= kAsmThreads && GP * 48 <= NWA * 36 * 16, "g partial buffer");
        double* s_gp = &s_part[0][0]; // [GP][48], the slab partials are long done
        __syncthreads();
        if (tid < GP * 48) {
            const int part = tid / 48, r = tid % 48, row = R0 + r;
            double v = 0.0;
            if (row < n) {
                const int pl = r / 6, cell = 4 * sub + 3 * (pl >> 2); // diagonal cell (u, u) of the row's pose
                const int x1 = d.cell_qj_ptr[cell + 1];
                for (int x = d.cell_qj_ptr[cell] + part; x < x1; x += 4 * GP) {
                    int q4[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) q4[i] = d.cell_qj[min(x + GP * i, x1 - 1)];
                    double g4[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) g4[i] = d.gslab[((size_t)q4[i] * 6 + r % 6) * 4 + (pl & 3)];
#pragma unroll
                    for (int i = 0; i < 4; ++i) if (x + GP * i < x1) v -= g4[i];
                }
            }
            s_gp[part * 48 + r] = v;
        }
        __syncthreads();
        for (int r = tid; r < 48; r += kAsmThreads) {
            const int row = R0 + r;
            double v = 0.0;
            if (row < n) {
                if (d.add_pose_terms) v = d.bp[row];
#pragma unroll
                for (int part = 0; part < GP; ++part) v += s_gp[part * 48 + r];
            }
            if (accumulate) d.g[row] += v; else d.g[row] = v;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// K7 (poses): trial pose = pose [+] dx (g2o VertexSE3::oplusImpl); pose part of g2o's computeScale.
// ---------------------------------------------------------------------------------------------
// Several ranks (scale_mode): the sum below must come out as ONE global number after the all-reduce of the trial scalars.
//   0  one rank: everything                       1  bp holds the exchanged totals: rank 0 reports the sum, the others 0
//   2  bp holds this rank's partial sums (linearisation kept local): every rank reports dx.bp_rank, rank 0 adds lambda |dx|^2
__global__ __launch_bounds__(kRedThreads) void k_update_poses(BaDev d, int cur, double lambda, int scale_mode, int rank)
{
    // one workgroup (the step scale is a single sum), but wide: with 1024 threads a trajectory of up to 1024 poses is one
    // pass of "index, then operands" round trips instead of several
    constexpr int NW = kRedThreads / 64;
    __shared__ double s_red[NW];
    const double* __restrict__ src = d.pose[cur];
    double* __restrict__ dst = d.pose[cur ^ 1];
    double part[1] = {0.0};
    for (int s = threadIdx.x; s < d.Pn; s += kRedThreads) {
        // the pose travels with its index (six 16-byte loads, nothing waits on them yet): a fixed pose used to be copied word
        // by word, twelve dependent round trips in ONE lane (the ISA showed load - wait - store twelve times) - which was
        // this kernel's critical path; a free pose fetched its operands only after the index had arrived
        double T[12], Tn[12];
        {
            const double2* sp = reinterpret_cast<const double2*>(src + 12 * s);
#pragma unroll
            for (int k = 0; k < 6; ++k) { const double2 v = sp[k]; T[2 * k] = v.x; T[2 * k + 1] = v.y; }
        }
        const int r = d.pose_red[s];
        if (r < 0) {
#pragma unroll
            for (int k = 0; k < 12; ++k) Tn[k] = T[k];
        } else {
            double dl[6], bl[6];
            const double wl = (scale_mode == 0 || rank == 0) ? lambda : 0.0, wb = (scale_mode == 1 && rank != 0) ? 0.0 : 1.0;
#pragma unroll
            for (int k = 0; k < 6; ++k) { dl[k] = d.dx[6 * r + k]; bl[k] = d.bp[6 * r + k]; }
#pragma unroll
            for (int k = 0; k < 6; ++k) part[0] += dl[k] * (wl * dl[k] + wb * bl[k]);
            pose_oplus(T, dl, Tn);
        }
        double2* dp = reinterpret_cast<double2*>(dst + 12 * s);
#pragma unroll
        for (int k = 0; k < 6; ++k) dp[k] = make_double2(Tn[2 * k], Tn[2 * k + 1]);
    }
    block_sum<1, NW>(part, s_red);
    if (threadIdx.x == 0) {
        d.scal[3] = part[0];
        // several ranks: the chi2 of the linearisation came in with the reduced system (k_assemble put this rank's share there)
        if (d.lin_from_red) { d.scal[8] = d.red_base[0]; d.scal[9] = d.red_base[1]; }
    }
}

// ---------------------------------------------------------------------------------------------
// K6 + K7 (landmarks) + K8: back-substitution dl = Hinv (b_l - sum_a W_a' dx_a), trial landmark,
// landmark part of computeScale, then the error sweep of the trial state.  Same workgroup shape as K2.
// BACKSUB = false: plain chi2 sweep of state `cur`.
// ---------------------------------------------------------------------------------------------
template <bool BACKSUB, bool DIAG>
__global__ __launch_bounds__(kBlock) void k_backsub_chi2(BaDev d, int cur, double lambda)
{
    __shared__ double s_v[3][kLmBlockEdges];
    __shared__ double s_lm[3][kLmBlockEdges];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int4 rec = reinterpret_cast<const int4*>(d.lb_rec)[b];
    const int l0 = rec.x, l1 = rec.x + rec.y, e0 = rec.z, e1 = rec.w;
    const size_t E = d.E, Ll = d.Ll;
    const int e = e0 + tid;
    const int trial = BACKSUB ? (cur ^ 1) : cur;
    const double* __restrict__ pose_t = d.pose[trial];
    double part[3] = {0.0, 0.0, 0.0}; // robust chi2, plain chi2, scale
    int s = 0;
    if (e < e1) s = d.e_pose[e];
    if (BACKSUB) {
        if (e < e1) {
            const int r = d.pose_red[s];
            double v[3] = {0.0, 0.0, 0.0};
            if (r >= 0) { // H_pl' dx = N' ( -dt + 2 Z x dq )
                double nz[12]; // N (9), 2Z (3): six double2 planes
#pragma unroll
                for (int pp = 0; pp < 6; ++pp) { const double2 x = ldp2(d.NZ, E, pp, (unsigned)e); nz[2 * pp] = x.x; nz[2 * pp + 1] = x.y; }
                const double z0 = nz[9], z1 = nz[10], z2 = nz[11];
                const double* dxp = d.dx + 6 * r;
                const double u0 = -dxp[0] + (z1 * dxp[5] - z2 * dxp[4]);
                const double u1 = -dxp[1] + (z2 * dxp[3] - z0 * dxp[5]);
                const double u2 = -dxp[2] + (z0 * dxp[4] - z1 * dxp[3]);
#pragma unroll
                for (int c = 0; c < 3; ++c) v[c] = nz[c] * u0 + nz[3 + c] * u1 + nz[6 + c] * u2;
            }
            s_v[0][tid] = v[0]; s_v[1][tid] = v[1]; s_v[2][tid] = v[2];
        }
        __syncthreads();
    }
    const int l = l0 + tid;
    if (l < l1) {
        double pn[3];
        const double* __restrict__ lm = d.lm[cur];
        if (BACKSUB) {
            double rhs[3] = {d.bl[l], d.bl[Ll + l], d.bl[2 * Ll + l]};
            const double bb[3] = {rhs[0], rhs[1], rhs[2]};
            for (int a = d.lm_ptr[l] - e0; a < d.lm_ptr[l + 1] - e0; ++a) { rhs[0] -= s_v[0][a]; rhs[1] -= s_v[1][a]; rhs[2] -= s_v[2][a]; }
            double Hi[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) Hi[k] = d.Hinv[(size_t)k * Ll + l];
            const double dl[3] = {Hi[0] * rhs[0] + Hi[1] * rhs[1] + Hi[2] * rhs[2], Hi[1] * rhs[0] + Hi[3] * rhs[1] + Hi[4] * rhs[2],
                                  Hi[2] * rhs[0] + Hi[4] * rhs[1] + Hi[5] * rhs[2]};
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                pn[k] = lm[3 * l + k] + dl[k];
                d.lm[cur ^ 1][3 * l + k] = pn[k];
                part[2] += dl[k] * (lambda * dl[k] + bb[k]);
            }
        } else {
#pragma unroll
            for (int k = 0; k < 3; ++k) pn[k] = lm[3 * l + k];
        }
        s_lm[0][tid] = pn[0]; s_lm[1][tid] = pn[1]; s_lm[2][tid] = pn[2];
        for (int q = d.lm_ll_ptr[l]; q < d.lm_ll_ptr[l + 1]; ++q) {
            double ee[3], O[6];
#pragma unroll
            for (int k = 0; k < 3; ++k) ee[k] = pn[k] - d.ll_ref[3 * q + k] - d.ll_z[3 * q + k];
#pragma unroll
            for (int k = 0; k < 6; ++k) O[k] = d.ll_info[6 * q + k];
            const double c2 = ee[0] * (O[0] * ee[0] + 2.0 * (O[1] * ee[1] + O[2] * ee[2])) +
                              ee[1] * (O[3] * ee[1] + 2.0 * O[4] * ee[2]) + ee[2] * O[5] * ee[2];
            double w = 1.0, r0 = c2;
            if (d.ll_robust[q]) cauchy(d.cauchy_delta, c2, r0, w);
            part[0] += r0; part[1] += c2;
        }
    }
    __syncthreads();
    if (e < e1) {
        EdgeIn in;
        load_edge<DIAG>(d.e_zi, d.e_flags, (int)E, e, in);
        const int ll = d.e_lm[e] - l0;
        double R[9], t[3];
#pragma unroll
        for (int k = 0; k < 9; ++k) R[k] = pose_t[12 * s + k];
#pragma unroll
        for (int k = 0; k < 3; ++k) t[k] = pose_t[12 * s + 9 + k];
        const double p[3] = {s_lm[0][ll], s_lm[1][ll], s_lm[2][ll]};
        double err[3];
        proj_error(in.type, R, t, p, in.z, d.fx, d.fy, d.cx, d.cy, err);
        const double c2 = edge_chi2(in, err);
        double w = 1.0, r0 = c2;
        if (in.robust) cauchy(d.cauchy_delta, c2, r0, w);
        part[0] += r0; part[1] += c2;
    }
    {
        const double x0 = wave_sum(part[0]), x1 = wave_sum(part[1]), x2 = wave_sum(part[2]);
        if ((tid & 63) == 0) {
            double* out = d.block_part + 4 * (size_t)(4 * b + (tid >> 6));
            out[0] = x0; out[1] = x1; out[2] = x2;
        }
    }
}

// The few numbers of an LM decision go straight into pinned host memory, the sequence number last (system-scope
// release); the host spins on the sequence number (ba_host.cpp read_scalars).  The status word is handed over and
// cleared for the next trial.  ONE thread does all of it: its own stores are ordered by the release, so no
// workgroup-wide system fence and barrier are needed (those cost 5 us of the 11 of the kernel that ends a trial).
// `fresh` (may be null): values for the first n_fresh slots that the caller holds in registers.
__device__ __forceinline__ void publish_scalars(const double* scal, int n, int* status, double* h_scal, int* h_status, int seq,
                                                const double* fresh = nullptr, int n_fresh = 0)
{
    if (threadIdx.x != 0) return;
    double v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = (i < n && i >= n_fresh) ? scal[i] : 0.0;
    const int st = *status;
#pragma unroll
    for (int i = 0; i < 16; ++i)
        if (i < n) __hip_atomic_store(h_scal + i, i < n_fresh ? fresh[i] : v[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(h_status, st, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    *status = 0;
    __hip_atomic_store(h_status + 1, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// scal[0] robust chi2, scal[1] plain chi2, scal[2] landmark part of the step scale (trial state `which`);
// n_pub > 0 (one rank): publishes scal[0..n_pub) to the host in the same launch.
// AUX: the pose-only edges of the trial state are evaluated here (no k_aux_edges<false> launch in front).
// LIN: the sums of the LINEARISATION this trial started from (k_lin_post's work: chi2 of the linearisation point, max |H_jj|)
//      are taken here as well - in all but the first iteration of a block only the host reads them, together with the trial.
// Several workgroups: one CU reading the 410 KB of per-wave partials of config 4 needs 4 us for that alone, and the thousand
// pose-only edges are a long dependent computation per lane.  Roles by workgroup: [0, G) sum a slice of the trial's partials,
// [G, 2G) (LIN) a slice of the linearisation's (the first of them also max |diag H_pp|), the last one (AUX) evaluates the
// pose-only edges.  Every workgroup leaves one record; the LAST to arrive adds the records in workgroup order (fixed order:
// the result does not depend on which one that is) and publishes.
constexpr int kRedGroups = 16;
template <bool AUX, bool LIN>
__global__ __launch_bounds__(kRedThreads) void k_reduce_trial(BaDev d, int which, int n_pub, double* h_scal, int* h_status, int seq)
{
    constexpr int NW = kRedThreads / 64, G = kRedGroups;
    __shared__ double s_red[3 * NW];
    __shared__ int s_last;
    const int b = blockIdx.x, n_part = 4 * d.n_lm_blocks;
    double out[3] = {0.0, 0.0, 0.0};
    if (b < (LIN ? 2 * G : G)) {
        const bool lin = LIN && b >= G;
        const int g = lin ? b - G : b;
        const Part4* __restrict__ src = reinterpret_cast<const Part4*>(lin ? d.lin_part : d.block_part);
        const int chunk = (n_part + G - 1) / G, i0 = g * chunk, i1 = min(n_part, i0 + chunk);
        double mx = 0.0;
        for (int i = i0 + (int)threadIdx.x; i < i1; i += kRedThreads) {
            const Part4 e = src[i];
            out[0] += e.v[0]; out[1] += e.v[1];
            if (lin) mx = fmax(mx, e.v[2]); else out[2] += e.v[2];
        }
        if (lin && g == 0)
            for (int i = threadIdx.x; i < d.Pf * 6; i += kRedThreads) {
                const int r = i / 6, a = i % 6; // diagonal entry a of the upper-triangle packing: index = a*6 - a(a-1)/2
                mx = fmax(mx, fabs(d.Hpp[(size_t)21 * r + (a * 6 - a * (a - 1) / 2)]));
            }
        block_sum<3, NW>(out, s_red);
        if (lin) out[2] = block_max<NW>(mx, s_red);
    } else if (AUX) {
        double ap[2] = {0.0, 0.0};
        aux_edges_chi2(d, d.pose[which], (int)threadIdx.x, kRedThreads, ap);
        out[0] = ap[0]; out[1] = ap[1];
        block_sum<3, NW>(out, s_red);
    }
    if (threadIdx.x == 0) {
        double* rec = d.tr_part + 4 * (size_t)b;
        rec[0] = out[0]; rec[1] = out[1]; rec[2] = out[2];
        __threadfence();
        s_last = (atomicAdd(d.tr_count, 1) == (int)gridDim.x - 1) ? 1 : 0;
    }
    __syncthreads();
    if (!s_last || threadIdx.x != 0) return;
    __threadfence();
    double pre[14];
#pragma unroll
    for (int i = 0; i < 14; ++i) pre[i] = __builtin_nontemporal_load(d.scal + i);
    const int st = *d.chol_status;
    double t[3] = {0.0, 0.0, 0.0}, l[2] = {0.0, 0.0}, mx = 0.0;
    for (int g = 0; g < G; ++g) {
        const double* rec = d.tr_part + 4 * (size_t)g;
        t[0] += __builtin_nontemporal_load(rec); t[1] += __builtin_nontemporal_load(rec + 1); t[2] += __builtin_nontemporal_load(rec + 2);
    }
    if (LIN)
        for (int g = G; g < 2 * G; ++g) {
            const double* rec = d.tr_part + 4 * (size_t)g;
            l[0] += __builtin_nontemporal_load(rec); l[1] += __builtin_nontemporal_load(rec + 1); mx = fmax(mx, __builtin_nontemporal_load(rec + 2));
        }
    const double* arec = d.tr_part + 4 * (size_t)(LIN ? 2 * G : G);
    const double a0 = AUX ? __builtin_nontemporal_load(arec) : pre[6], a1 = AUX ? __builtin_nontemporal_load(arec + 1) : pre[7];
    double fresh[3] = {t[0] + a0, t[1] + a1, t[2]};
    // several ranks: a rank whose own landmark blocks or factorisation failed must fail the trial everywhere -
    // its chi2 goes out as +inf, which survives the sum of the all-reduce (ba_host.cpp trial())
    if (n_pub == 0 && st != 0) fresh[0] = __builtin_inf();
    d.scal[0] = fresh[0]; d.scal[1] = fresh[1]; d.scal[2] = fresh[2];
    if (AUX) { d.scal[6] = a0; d.scal[7] = a1; }
    if (LIN) { // (slots 12, 13: the pose-only edges at the linearisation point, k_aux_edges<true>)
        pre[8] = l[0] + pre[12]; pre[9] = l[1] + pre[13]; pre[10] = mx; pre[5] = mx;
        d.scal[5] = mx; d.scal[8] = pre[8]; d.scal[9] = pre[9]; d.scal[10] = mx;
    }
    *d.tr_count = 0;
    if (n_pub > 0) {
#pragma unroll
        for (int i = 0; i < 12; ++i)
            if (i < n_pub) __hip_atomic_store(h_scal + i, i < 3 ? fresh[i] : pre[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(h_status, st, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        *d.chol_status = 0;
        __hip_atomic_store(h_status + 1, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

__global__ __launch_bounds__(kBlock) void k_debug_jacobians(BaDev d, int cur, const int* __restrict__ e_orig, double* err,
                                                            double* Jp, double* Jl)
{
    const int e = blockIdx.x * kBlock + threadIdx.x;
    if (e >= d.E) return;
    EdgeIn in;
    if (d.info_planes == 3) load_edge<true>(d.e_zi, d.e_flags, d.E, e, in);
    else load_edge<false>(d.e_zi, d.e_flags, d.E, e, in);
    const int s = d.e_pose[e], l = d.e_lm[e];
    const double* pose = d.pose[cur];
    const double* lm = d.lm[cur];
    double ee[3], J[27];
    proj_eval(in.type, pose + 12 * s, pose + 12 * s + 9, lm + 3 * l, in.z, d.fx, d.fy, d.cx, d.cy, ee, J);
    const size_t o = e_orig[e];
    for (int r = 0; r < 3; ++r) {
        err[3 * o + r] = ee[r];
        for (int c = 0; c < 6; ++c) Jp[18 * o + 6 * r + c] = J[9 * r + c];
        for (int c = 0; c < 3; ++c) Jl[9 * o + 3 * r + c] = J[9 * r + 6 + c];
    }
}

// parity tap: errors and Jacobians of the pose-only edges through the same device functions as k_aux_edges
__global__ __launch_bounds__(kAuxThreads) void k_debug_aux_jacobians(BaDev d, int cur, double* se3_err, double* se3_Ji, double* se3_Jj,
                                                                     double* acc_err, double* acc_J)
{
    const double* __restrict__ pose = d.pose[cur];
    const int gid = blockIdx.x * kAuxThreads + threadIdx.x;
    for (int k = gid; k < d.n_se3; k += gridDim.x * kAuxThreads) {
        double e[6], Ji[36], Jj[36];
        se3_edge_eval(pose + 12 * d.se3_i[k], pose + 12 * d.se3_j[k], d.se3_Z + 12 * k, e, Ji, Jj);
        for (int q = 0; q < 6; ++q) se3_err[6 * k + q] = e[q];
        for (int q = 0; q < 36; ++q) { se3_Ji[36 * k + q] = Ji[q]; se3_Jj[36 * k + q] = Jj[q]; }
    }
    for (int k = gid; k < d.n_accel; k += gridDim.x * kAuxThreads) {
        const double v[3] = {d.acc_a[3 * k], d.acc_a[3 * k + 1], d.acc_a[3 * k + 2]};
        double e[3], J[18];
        accel_edge_eval(pose + 12 * d.acc_pose[k], v, e, J);
        for (int q = 0; q < 3; ++q) acc_err[3 * k + q] = e[q];
        for (int q = 0; q < 18; ++q) acc_J[18 * k + q] = J[q];
    }
}

} // namespace

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
static inline hipStream_t S_(void* st) { return static_cast<hipStream_t>(st); }

void ba_linearize_lm(const BaDev& d, int cur, void* st)
{
    if (d.n_lm_blocks <= 0) return;
    if (d.info_planes == 3) hipLaunchKernelGGL(k_linearize_lm<true>, dim3(d.n_lm_blocks), dim3(kBlock), 0, S_(st), d, cur);
    else hipLaunchKernelGGL(k_linearize_lm<false>, dim3(d.n_lm_blocks), dim3(kBlock), 0, S_(st), d, cur);
}
void ba_linearize_pose(const BaDev& d, int cur, void* st)
{
    if (d.n_chunks <= 0) return;
    if (d.info_planes == 3) hipLaunchKernelGGL(k_linearize_pose<true>, dim3(d.n_chunks), dim3(kBlock), 0, S_(st), d, cur);
    else hipLaunchKernelGGL(k_linearize_pose<false>, dim3(d.n_chunks), dim3(kBlock), 0, S_(st), d, cur);
}
// Results for the host without a copy engine round trip: the scalars go straight into pinned host memory, the
// sequence number last (system-scope release); the host spins on the sequence number (ba_host.cpp read_scalars).
__global__ __launch_bounds__(64) void k_publish(const double* __restrict__ scal, int n, int* status, double* h_scal, int* h_status, int seq)
{
    publish_scalars(scal, n, status, h_scal, h_status, seq);
}

void ba_publish(const BaDev& d, int n, double* h_scal, int* h_status, int seq, void* st)
{
    hipLaunchKernelGGL(k_publish, dim3(1), dim3(64), 0, S_(st), d.scal, n, d.chol_status, h_scal, h_status, seq);
}

void ba_linearize_aux(const BaDev& d, int cur, int, void* st)
{
    hipLaunchKernelGGL((k_aux_edges<true>), dim3(d.aux_blocks), dim3(kAuxThreads), 0, S_(st), d, cur);
}
void ba_chi2_aux(const BaDev& d, int which, int, void* st)
{
    hipLaunchKernelGGL((k_aux_edges<false>), dim3(d.aux_blocks), dim3(kAuxThreads), 0, S_(st), d, which);
}
// with_invert: the landmark blocks are inverted for `lambda` in the same launch (ba_invert_landmarks is then not needed)
void ba_pose_finalize(const BaDev& d, const int* red_slot, int rank, int n_ranks, int with_invert, double lambda, void* st)
{
    const int fb = (d.Pf * 27 + kBlock - 1) / kBlock, ib = with_invert ? (d.Ll + kBlock - 1) / kBlock : 0;
    if (ib > 0) hipLaunchKernelGGL(k_finalize_invert, dim3(fb + ib), dim3(kBlock), 0, S_(st), d, red_slot, fb, lambda);
    else if (d.Pf > 0) hipLaunchKernelGGL(k_pose_finalize, dim3(fb), dim3(kBlock), 0, S_(st), d, red_slot);
    if (n_ranks > 1) hipLaunchKernelGGL(k_reduce_lin_scalars, dim3(1), dim3(kRedThreads), 0, S_(st), d, rank, n_ranks);
}
void ba_lin_post(const BaDev& d, int n_ranks, void* st)
{
    if (n_ranks > 1) hipLaunchKernelGGL(k_lin_post<false>, dim3(1), dim3(kRedThreads), 0, S_(st), d, n_ranks);
    else hipLaunchKernelGGL(k_lin_post<true>, dim3(1), dim3(kRedThreads), 0, S_(st), d, n_ranks);
}
void ba_invert_landmarks(const BaDev& d, double lambda, void* st)
{
    if (d.Ll > 0) hipLaunchKernelGGL(k_invert_landmarks, dim3((d.Ll + kBlock - 1) / kBlock), dim3(kBlock), 0, S_(st), d, lambda);
}
// all stages in one launch; sg (optional): the values a waiting stream is released by, stage by stage
// seq: the trial's number among the staged launches of the handle (its parity picks the ticket set); reserve_per_se: see k_schur
void ba_schur(const BaDev& d, const StageSignals* sg, unsigned long long seq, int reserve_per_se, int launch_wgs, void* st)
{
    StageSignals s{};
    if (sg) s = *sg;
    s.seq = seq;
    const int wgs = d.n_stages > 1 && launch_wgs > 0 ? launch_wgs : (d.n_jobs + 3) / 4;
    if (d.n_jobs > 0) hipLaunchKernelGGL(k_schur, dim3(wgs), dim3(kBlock), 0, S_(st), d, 0, d.n_stages, s, reserve_per_se);
}
// sub-tiles [sub0, sub1)
void ba_assemble(const BaDev& d, int sub0, int sub1, int accumulate, void* st)
{
    if (sub1 > sub0) hipLaunchKernelGGL(k_assemble, dim3(sub1 - sub0), dim3(kAsmThreads), 0, S_(st), d, sub0, accumulate);
}
void ba_update_poses(const BaDev& d, int cur, double lambda, int scale_mode, int rank, void* st)
{
    hipLaunchKernelGGL(k_update_poses, dim3(1), dim3(kRedThreads), 0, S_(st), d, cur, lambda, scale_mode, rank);
}
void ba_backsub_chi2(const BaDev& d, int cur, double lambda, void* st)
{
    if (d.n_lm_blocks <= 0) return;
    if (d.info_planes == 3) hipLaunchKernelGGL((k_backsub_chi2<true, true>), dim3(d.n_lm_blocks), dim3(kBlock), 0, S_(st), d, cur, lambda);
    else hipLaunchKernelGGL((k_backsub_chi2<true, false>), dim3(d.n_lm_blocks), dim3(kBlock), 0, S_(st), d, cur, lambda);
}
void ba_chi2_only(const BaDev& d, int which, void* st)
{
    if (d.n_lm_blocks <= 0) return;
    if (d.info_planes == 3) hipLaunchKernelGGL((k_backsub_chi2<false, true>), dim3(d.n_lm_blocks), dim3(kBlock), 0, S_(st), d, which, 0.0);
    else hipLaunchKernelGGL((k_backsub_chi2<false, false>), dim3(d.n_lm_blocks), dim3(kBlock), 0, S_(st), d, which, 0.0);
}
// aux_state >= 0: evaluate the pose-only edges of that state here; with_lin: also the sums k_lin_post would have taken
void ba_reduce_trial_scalars(const BaDev& d, int aux_state, int with_lin, int n_pub, double* h_scal, int* h_status, int seq, void* st)
{
    const int w = aux_state >= 0 ? aux_state : 0, aux = aux_state >= 0 ? 1 : 0;
    const dim3 grid(kRedGroups * (with_lin ? 2 : 1) + aux);
    if (aux && with_lin) hipLaunchKernelGGL((k_reduce_trial<true, true>), grid, dim3(kRedThreads), 0, S_(st), d, w, n_pub, h_scal, h_status, seq);
    else if (aux) hipLaunchKernelGGL((k_reduce_trial<true, false>), grid, dim3(kRedThreads), 0, S_(st), d, w, n_pub, h_scal, h_status, seq);
    else if (with_lin) hipLaunchKernelGGL((k_reduce_trial<false, true>), grid, dim3(kRedThreads), 0, S_(st), d, w, n_pub, h_scal, h_status, seq);
    else hipLaunchKernelGGL((k_reduce_trial<false, false>), grid, dim3(kRedThreads), 0, S_(st), d, w, n_pub, h_scal, h_status, seq);
}
void ba_debug_jacobians(const BaDev& d, int cur, const int* e_orig, double* err, double* Jp, double* Jl, void* st)
{
    if (d.E > 0) hipLaunchKernelGGL(k_debug_jacobians, dim3((d.E + kBlock - 1) / kBlock), dim3(kBlock), 0, S_(st), d, cur, e_orig, err, Jp, Jl);
}

void ba_debug_aux_jacobians(const BaDev& d, int cur, double* se3_err, double* se3_Ji, double* se3_Jj, double* acc_err, double* acc_J, void* st)
{
    hipLaunchKernelGGL(k_debug_aux_jacobians, dim3(d.aux_blocks), dim3(kAuxThreads), 0, S_(st), d, cur, se3_err, se3_Ji, se3_Jj, acc_err, acc_J);
}

// Edge values from the device-side log (insertion order, 9 doubles per edge: z, upper triangle of the information) into the
// double2 planes the sweeps read: out[2 * ((v / 2) * stride + k) + (v & 1)], v = 0..2 z, 3.. information (diagonal only when
// planes == 3).  k-th output edge = log row src[via ? via[k] : k]; lm_out (nullable) = lm_in[via[k]] for the pose-major copy.
__global__ __launch_bounds__(256) void k_gather_edges(const double* __restrict__ raw, const uint8_t* __restrict__ raw_flags, const int* __restrict__ src,
                                                      const int* __restrict__ via, int count, int stride, int planes, double* __restrict__ out_zi,
                                                      uint8_t* __restrict__ out_flags, const int* __restrict__ lm_in, int* __restrict__ lm_out)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= count) return;
    const int m = via ? via[k] : k;
    const int s = src[m];
    const double* r = raw + (size_t)9 * s;
    double v[10];
    v[0] = r[0]; v[1] = r[1]; v[2] = r[2];
    if (planes == 3) { v[3] = r[3]; v[4] = r[6]; v[5] = r[8]; v[6] = 0.0; v[7] = 0.0; v[8] = 0.0; v[9] = 0.0; }
    else { for (int q = 0; q < 6; ++q) v[3 + q] = r[3 + q]; v[9] = 0.0; }
    const int np2 = (3 + planes + 1) / 2;
    double2* out = reinterpret_cast<double2*>(out_zi);
    for (int p = 0; p < np2; ++p) out[(size_t)p * stride + k] = make_double2(v[2 * p], v[2 * p + 1]);
    out_flags[k] = raw_flags[s];
    if (lm_out) lm_out[k] = lm_in[m];
}

void ba_gather_edges(const double* raw, const uint8_t* raw_flags, const int* src, const int* via, int count, int stride, int planes, double* out_zi,
                     uint8_t* out_flags, const int* lm_in, int* lm_out, void* st)
{
    if (count > 0)
        hipLaunchKernelGGL(k_gather_edges, dim3((count + 255) / 256), dim3(256), 0, S_(st), raw, raw_flags, src, via, count, stride, planes, out_zi,
                           out_flags, lm_in, lm_out);
}

void ba_configure_kernels(int) {}

} // namespace svi
