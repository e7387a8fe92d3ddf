// ba_chol.hip — tile-sparse Cholesky of the reduced camera system S dx = g on gfx950.
//
// Stands in for what CHOLMOD does inside g2o::LinearSolverCholmod (configured at
// src/optimization/Cg2oOptimizer.cpp:83) on the pose part of the system once the landmarks have
// been eliminated (their elimination is the Schur reduction of ba_kernels.hip).  S is stored as
// TS x TS tiles of its lower block triangle, only tiles that can be non-zero after fill-in; the
// factorisation is right-looking over tile columns k:
//   potrf : L_kk = chol(S_kk + lambda I), L_kk^-1, y_k = L_kk^-1 g_k      one workgroup
//           (register-resident right-looking sweep, one barrier per column; the inverse is built
//            from 16x16 blocks with FP64 MFMA)
//   trsm  : L_ik = S_ik L_kk^-T as a GEMM with L_kk^-1                     one workgroup per 48x48 block, FP64 MFMA
//   gemm  : S_ij -= L_ik L_jk' ;  g_i -= L_ik y_k                          one workgroup per 48x48 block, FP64 MFMA
// so the forward substitution rides along with the factorisation; the backward substitution is one
// workgroup walking the tile columns in reverse.  v_mfma_f64_16x16x4_f64 on this dense block is the
// only MFMA-shaped work on the whole path.  L goes to a separate tile array (trsm is then free of
// read/write races between the workgroups sharing a tile).
// A non-positive pivot sets *status = k+1: the caller treats the LM trial as failed, like g2o does
// when CHOLMOD reports "not positive definite".
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdint>
#include <cstdlib>
#include <vector>

#include "ba_device.h"
#include "ba_math.h"

namespace svi {
extern std::atomic<int> g_backsolve_spin_limit; // ba_host.cpp
namespace {

constexpr int kBlock = 256;
constexpr int kOB = 48; // output block edge of the trsm / gemm workgroups
typedef double v4f64 __attribute__((ext_vector_type(4)));

// -DPOTRF_TS: shader-clock stamps of the first chain workgroup of every level (phase boundaries of k_potrf_inv), printed by
// run() at its 100th call - how the per-phase figures in DESIGN.md were measured.  Compiled out otherwise.
#ifdef POTRF_TS
__device__ unsigned long long g_ts[64 * 16];
__device__ int g_ts_cnt, g_ts_cur;
#define TS_MARK(i) do { if (threadIdx.x == 0 && blockIdx.x == 0) g_ts[g_ts_cur * 16 + (i)] = clock64(); } while (0)
#else
#define TS_MARK(i) do { } while (0)
#endif
template <int TS> struct Lds { static constexpr int LD = TS + 2; }; // (TS+2) % 32 == 2: conflict-free ds_read_b64 of MFMA operands

// rows [r0, r0+NR) of a TS x TS row-major global tile -> LDS (row stride LD)
template <int TS, int NR, int NTHR = kBlock>
__device__ __forceinline__ void rows_to_lds(const double* __restrict__ g, int r0, double* s)
{
    constexpr int LD = Lds<TS>::LD;
    const double2* src = reinterpret_cast<const double2*>(g + (size_t)r0 * TS);
    constexpr int N = NR * TS / 2, IT = (N + NTHR - 1) / NTHR;
    double2 v[IT];
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int i = it * NTHR + threadIdx.x;
        if (N % NTHR == 0 || i < N) v[it] = src[i];
    }
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int i = it * NTHR + threadIdx.x;
        if (N % NTHR == 0 || i < N) {
            const int r = i / (TS / 2), c = 2 * (i % (TS / 2));
            s[r * LD + c] = v[it].x;
            s[r * LD + c + 1] = v[it].y;
        }
    }
}

// a whole TS x TS tile through registers: issue the global loads early, park them in LDS later (512 threads)
template <int TS> struct TileRegs { static constexpr int N = TS * TS / 2, IT = (N + 512 - 1) / 512; double2 v[IT]; };

template <int TS>
__device__ __forceinline__ void tile_load(const double* __restrict__ g, TileRegs<TS>& t)
{
    const double2* src = reinterpret_cast<const double2*>(g);
#pragma unroll
    for (int it = 0; it < TileRegs<TS>::IT; ++it) {
        const int i = it * 512 + threadIdx.x;
        if (TileRegs<TS>::N % 512 == 0 || i < TileRegs<TS>::N) t.v[it] = src[i];
    }
}

// a += b (the assembled tile and the updates the factorisation has accumulated for it live in two arrays when the reduction is staged)
template <int TS>
__device__ __forceinline__ void tile_add(TileRegs<TS>& a, const TileRegs<TS>& b)
{
#pragma unroll
    for (int it = 0; it < TileRegs<TS>::IT; ++it) { a.v[it].x += b.v[it].x; a.v[it].y += b.v[it].y; }
}

template <int TS>
__device__ __forceinline__ void tile_store(const TileRegs<TS>& t, double* s)
{
    constexpr int LD = Lds<TS>::LD;
#pragma unroll
    for (int it = 0; it < TileRegs<TS>::IT; ++it) {
        const int i = it * 512 + threadIdx.x;
        if (TileRegs<TS>::N % 512 == 0 || i < TileRegs<TS>::N) {
            const int r = i / (TS / 2), c = 2 * (i % (TS / 2));
            s[r * LD + c] = t.v[it].x;
            s[r * LD + c + 1] = t.v[it].y;
        }
    }
}

// 16x16 block of C = A B' (A: rows ra.., B: rows rb.. of LDS images with stride LD), K = KK
template <int KK, int LD>
__device__ __forceinline__ v4f64 mfma_block(const double* sA, int ra, const double* sB, int rb)
{
    const int lane = threadIdx.x & 63;
    const double* pa = sA + (ra + (lane & 15)) * LD + (lane >> 4);
    const double* pb = sB + (rb + (lane & 15)) * LD + (lane >> 4);
    double a[KK / 4], b[KK / 4];
#pragma unroll
    for (int q = 0; q < KK / 4; ++q) { a[q] = pa[4 * q]; b[q] = pb[4 * q]; }
    // every operand read is issued before the first product: left to itself the scheduler recycles two operand registers and
    // puts an LDS round trip between every pair of matrix-core steps (2300 cycles for twelve steps instead of 900, in the ISA)
    __builtin_amdgcn_sched_barrier(0);
    v4f64 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int q = 0; q < KK / 4; ++q) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q], b[q], acc, 0, 0, 0);
    return acc;
}

template <int KK, int LD>
__device__ __forceinline__ v4f64 mfma_block_acc(const double* sA, int ra, const double* sB, int rb, v4f64 acc)
{
    const int lane = threadIdx.x & 63;
    const double* pa = sA + (ra + (lane & 15)) * LD + (lane >> 4);
    const double* pb = sB + (rb + (lane & 15)) * LD + (lane >> 4);
    if constexpr (KK <= 48) {
        double a[KK / 4], b[KK / 4];
#pragma unroll
        for (int q = 0; q < KK / 4; ++q) { a[q] = pa[4 * q]; b[q] = pb[4 * q]; }
        __builtin_amdgcn_sched_barrier(0); // (as above: all reads first)
#pragma unroll
        for (int q = 0; q < KK / 4; ++q) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q], b[q], acc, 0, 0, 0);
    } else {
#pragma unroll 6
        for (int k0 = 0; k0 < KK; k0 += 4) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[k0], pb[k0], acc, 0, 0, 0);
    }
    return acc;
}

// 1/x to full double precision from the hardware seed (two Newton steps), no IEEE division sequence
__device__ __forceinline__ double fast_rcp(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    r = fma(r, fma(-x, r, 1.0), r);
    r = fma(r, fma(-x, r, 1.0), r);
    return r;
}
// 1/sqrt(x) to full double precision
__device__ __forceinline__ double fast_rsqrt(double x)
{
    double r = __builtin_amdgcn_rsq(x);
    // r <- r + r*(1 - x r^2)/2, twice
    r = fma(r * 0.5, fma(-x * r, r, 1.0), r);
    r = fma(r * 0.5, fma(-x * r, r, 1.0), r);
    return r;
}

constexpr int kPotrfThreads = 512;

// row r of y = L^-1 g from an LDS image of L^-1 (row stride LD, zero above the diagonal): four partial sums over a quarter of the
// columns each, four interleaved chains inside a quarter - term for term what the factorising workgroup computes when it finishes
// y_k itself (k_potrf_inv), so that a consumer that forms y_q on the spot (StepArgs::defer_y) gets the same bits
template <int TS, int LD>
__device__ __forceinline__ double y_row(const double* sXinv, int r, const double* gv)
{
    constexpr int YC = TS / 4;
    double p[4];
#pragma unroll
    for (int part = 0; part < 4; ++part) {
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
        for (int c = part * YC; c < part * YC + YC; c += 4) {
            a0 = fma(sXinv[r * LD + c], gv[c], a0);
            a1 = fma(sXinv[r * LD + c + 1], gv[c + 1], a1);
            a2 = fma(sXinv[r * LD + c + 2], gv[c + 2], a2);
            a3 = fma(sXinv[r * LD + c + 3], gv[c + 3], a3);
        }
        p[part] = (a0 + a1) + (a2 + a3);
    }
    return (p[0] + p[1]) + (p[2] + p[3]);
}

// "not computed yet" in the solution vector of the one-launch backward substitution: a NaN with a payload no computation produces
__device__ __forceinline__ double solve_pending() { return __longlong_as_double(0x7FF8DEADBEEF0001LL); }
__device__ __forceinline__ bool is_solve_pending(double v) { return __double_as_longlong(v) == 0x7FF8DEADBEEF0001LL; }

// Tile edge 48: the triangular solves of a level (L_ik = S_ik L_kk^-T) are not a launch of their own - every consumer of
// L_ik in the NEXT level's launch (the chain workgroup that applies it to its diagonal tile, the grouped updates) forms it
// on the spot from S_ik and L_kk^-1 (a 48^3 product: 27 matrix-core steps), and the one that owns the diagonal target of
// row i also stores it for the backward substitution.  One launch per dependency level instead of two; with 96-wide tiles
// the three LDS images this needs do not fit beside the factorisation's own (k_trsm stays).
template <int TS> struct Fold { static constexpr bool on = TS == 48; };

// 16x16 block (r0, c0) of S X' for a LOWER TRIANGULAR X (an inverse of a diagonal factor): X[c][m] = 0 for m > c, so the inner
// index stops at the block column's last row - 4, 8 or 12 matrix-core steps instead of 12 for the three block columns of a 48 tile
template <int TS, int LD>
__device__ __forceinline__ v4f64 mfma_block_tri(const double* sS, int r0, const double* sXinv, int c0)
{
    static_assert(TS == 48, "written out for three block columns");
    return c0 == 0 ? mfma_block<16, LD>(sS, r0, sXinv, c0) : c0 == 16 ? mfma_block<32, LD>(sS, r0, sXinv, c0) : mfma_block<48, LD>(sS, r0, sXinv, c0);
}

// the 16x16 accumulator block `acc` (row (lane>>4) + 4q, column lane&15) into an LDS image at (r0, c0)
template <int LD> __device__ __forceinline__ void block_to_lds(const v4f64& acc, double* s, int r0, int c0)
{
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int q = 0; q < 4; ++q) s[(r0 + (lane >> 4) + 4 * q) * LD + c0 + (lane & 15)] = acc[q];
}
// ... and into a row-major global tile
template <int TS> __device__ __forceinline__ void block_to_global(const v4f64& acc, double* __restrict__ g, int r0, int c0)
{
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int q = 0; q < 4; ++q) g[(size_t)(r0 + (lane >> 4) + 4 * q) * TS + c0 + (lane & 15)] = acc[q];
}

// Block ownership and work tables of the factorisation workgroup, packed into 32-bit literals (a nibble per wave, value + 1):
// indexed as arrays they are loads from the constant section, i.e. a memory round trip in front of the operand loads whose
// addresses depend on them (seen in the ISA: 0.4 us per level).
constexpr signed char kOwn6T[8][3][2] = {{{5, 5}, {4, 3}, {3, 1}}, {{5, 4}, {3, 3}, {2, 1}}, {{4, 4}, {5, 2}, {1, 1}},
                                         {{5, 3}, {4, 2}, {5, 0}}, {{3, 2}, {4, 0}, {0, 0}}, {{2, 2}, {3, 0}, {-1, -1}},
                                         {{5, 1}, {2, 0}, {-1, -1}}, {{4, 1}, {1, 0}, {-1, -1}}};
constexpr signed char kOwn3T[8][2] = {{2, 2}, {2, 1}, {1, 1}, {2, 0}, {1, 0}, {0, 0}, {-1, -1}, {-1, -1}};
constexpr signed char kTriT[8][2] = {{2, -1}, {5, -1}, {8, -1}, {1, -1}, {4, -1}, {7, -1}, {0, 3}, {6, -1}};
constexpr unsigned pack_own6(int u, int ab) { unsigned v = 0; for (int w = 0; w < 8; ++w) v |= (unsigned)(kOwn6T[w][u][ab] + 1) << (4 * w); return v; }
constexpr unsigned pack_own3(int ab) { unsigned v = 0; for (int w = 0; w < 8; ++w) v |= (unsigned)(kOwn3T[w][ab] + 1) << (4 * w); return v; }
constexpr unsigned pack_tri(int u) { unsigned v = 0; for (int w = 0; w < 8; ++w) v |= (unsigned)(kTriT[w][u] + 1) << (4 * w); return v; }
__device__ __forceinline__ int unpack_nibble(unsigned packed, int wave) { return (int)((packed >> (4 * wave)) & 0xFu) - 1; }

// ---------------------------------------------------------------------------------------------
// potrf + inverse + y_k of one diagonal tile, 512 threads.  The right-looking sweep eliminates FOUR columns per
// barrier (the 4x4 pivot block is factorised redundantly by every lane) and keeps the tile in MFMA accumulator
// layout: the 16x16 blocks of the lower triangle go round the eight waves (block bi = wave + 8 u), a lane holds rows
// (lane>>4) + 4q, column lane&15 of its blocks.  The
// rank-4 trailing update of a quad is then ONE v_mfma_f64_16x16x4 per block (operands: the eliminated panel entries
// of the block's rows and columns, solved per lane from the published columns), the pending updates of the level
// below accumulate into the same registers, and nothing is computed 16 times over.  Columns <= the pivot are kept
// out of an update by zeroing the operand, so an accumulator column always ends as the unscaled factor column.
// ---------------------------------------------------------------------------------------------
// SPLIT: the reduced system comes in two arrays - S (assembled by the Schur kernel, never written here) and Su (what the levels
// below have subtracted so far, zero at the start of a trial): every tile is read as S + Su, every update goes to Su.  The Schur
// reduction of a later stage may then still be writing ITS tiles of S while earlier levels already update them (ba_host.cpp).
template <int TS, bool SPLIT>
__device__ __forceinline__ int potrf_sweep_mfma(const double* __restrict__ A, const double* __restrict__ Au, const double* __restrict__ Su_all, double* __restrict__ Lg, double* sL, double* sX,
                                                 double (*s_col)[TS][4], double* s_rs, int k, int n, double lambda, int stop_after,
                                                 double* __restrict__ y, double* __restrict__ Lt, const int* __restrict__ pre_tile_g,
                                                 const int* __restrict__ pre_col_g, int npre, double* s_g, const double* __restrict__ S_all,
                                                 const double* __restrict__ Linv_all, double* sT, const int* __restrict__ status,
                                                 int it0, int it1, int ic0, int ic1, double gk, double* s_cf, int defer_y, double* s_y)
{
    // the status word goes first, the operand tiles right behind it: the test waits for its own load only
    const int failed_before = *status;
    // The first two pending sources arrive as values (kernel arguments, or read by the caller with the column record): a
    // conditional index load in front of the tile loads is a join point where the compiler has to wait for EVERY load in
    // flight (the counter cannot tell them apart) - the operand tiles then arrived one round trip after the other.
    static_assert(kInlinePre == 2, "the inline record is selected without indexing");
    auto pre_tile = [=](int w) { return w == 0 ? it0 : w == 1 ? it1 : pre_tile_g[w]; };
    auto pre_col = [=](int w) { return w == 0 ? ic0 : w == 1 ? ic1 : pre_col_g[w]; };
    constexpr int kOk = 0, kNotPositive = 1, kAborted = 2;
    constexpr int NB = TS / 16, LD = Lds<TS>::LD, KB = 4;
    constexpr int NBLK = NB * (NB + 1) / 2, NWV = kPotrfThreads / 64, PER = (NBLK + NWV - 1) / NWV;
    static_assert(kPotrfThreads == 512, "TileRegs assumes 512 threads");
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, ln = lane & 15, lk = lane >> 4;
    // g_k and its pending updates live on the LAST wave (no block of the folded products is its): on wave 0 the 48-long dot
    // products (96 LDS reads per lane) ran behind that wave's own matrix-core block, 0.5 us per level
    const int gt = tid - (kPotrfThreads - 64 * ((TS + 63) / 64));
    const bool g_thr = gt >= 0 && gt < TS;
    int ba[PER], bb[PER];
    bool own[PER];
    v4f64 acc[PER];
    // Which blocks a wave owns.  The pivot chain runs redundantly on all eight waves, but the four that share a SIMD
    // with an older wave issue behind it (measured: 480 vs 795 cycles per quad), and a block is live only while its
    // block column has not been passed: the tables give the long-lived blocks to the fast waves and the slow waves
    // one block less, from a greedy search over per-quad cost = chain(wave) + 250 x live blocks (tools note in DESIGN).
    static_assert((NB == 6 && PER == 3) || (NB == 3 && PER == 1), "block ownership tables exist for 96 and 48 wide tiles");
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const int ta = unpack_nibble(NB == 6 ? pack_own6(u < 3 ? u : 0, 0) : pack_own3(0), wave);
        const int tb = unpack_nibble(NB == 6 ? pack_own6(u < 3 ? u : 0, 1) : pack_own3(1), wave);
        own[u] = ta >= 0;
        ba[u] = own[u] ? ta : 0;
        bb[u] = own[u] ? tb : 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = 16 * ba[u] + lk + 4 * q, c = 16 * bb[u] + ln;
            // (lambda goes on below: nothing may consume this load before the others are issued)
            if constexpr (SPLIT) acc[u][q] = own[u] ? A[r * TS + c] + Au[r * TS + c] : 0.0;
            else acc[u][q] = own[u] ? A[r * TS + c] : 0.0;
        }
    }
    // pending updates of this tile from the columns of the level just below: A -= L(k,q) L(k,q)' and the forward
    // substitution g_k -= L(k,q) y_q, done here so the critical path is one launch per level.  L(k,q) is staged in
    // the (still unused) L image, y_q in s_rs.
    v4f64 accU[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) accU[u] = {0.0, 0.0, 0.0, 0.0};
    if constexpr (Fold<TS>::on) {
        // L(k,q) = S(k,q) L_qq^-T is formed here (no k_trsm launch): S(k,q) -> sX, L_qq^-1 -> sL, the product -> sT and, for
        // the backward substitution, to its tile of Lt
        TileRegs<TS> ps, px, pu;
        double ypre = 0.0;
        if (npre > 0) {
            tile_load<TS>(S_all + (size_t)pre_tile(0) * TS * TS, ps);
            if constexpr (SPLIT) tile_load<TS>(Su_all + (size_t)pre_tile(0) * TS * TS, pu);
            tile_load<TS>(Linv_all + (size_t)pre_col(0) * TS * TS, px);
            if (tid < TS) ypre = y[pre_col(0) * TS + tid];
        }
        if (failed_before != 0) return kAborted; // an earlier column failed: nothing more to do in this trial
        for (int w = 0; w < npre; ++w) {
            if constexpr (SPLIT) tile_add<TS>(ps, pu);
            tile_store<TS>(ps, sX);
            tile_store<TS>(px, sL);
            if (tid < TS) s_rs[tid] = ypre;
            __syncthreads();
            TS_MARK(1); // operands in LDS
            double* Lout = Lt + (size_t)pre_tile(w) * TS * TS;
            if (w + 1 < npre) { // the next source travels while this one is multiplied
                tile_load<TS>(S_all + (size_t)pre_tile(w + 1) * TS * TS, ps);
                if constexpr (SPLIT) tile_load<TS>(Su_all + (size_t)pre_tile(w + 1) * TS * TS, pu);
                tile_load<TS>(Linv_all + (size_t)pre_col(w + 1) * TS * TS, px);
                if (tid < TS) ypre = y[pre_col(w + 1) * TS + tid];
            }
            // nine blocks on eight waves, by cost (block column c needs 4 (c + 1) matrix-core steps): the three of the last
            // column on waves 0-2, the middle column on waves 3-5, two of the first on wave 6 and one on wave 7 - 12 steps at most
            {
                static_assert(NB == 3 && NWV == 8, "block assignment of the folded triangular product (kTriT)");
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int st = unpack_nibble(pack_tri(u), wave);
                    if (st < 0) continue;
                    const int r0 = (st / NB) * 16, c0 = (st % NB) * 16;
                    const v4f64 lb = mfma_block_tri<TS, LD>(sX, r0, sL, c0);
                    if (u == 0) TS_MARK(11);
                    block_to_lds<LD>(lb, sT, r0, c0);
                    block_to_global<TS>(lb, Lout, r0, c0);
                }
            }
            // y_q for the forward substitution below: as it came (s_rs), or - defer_y: s_rs holds the final g_q - formed here from the
            // image of L_qq^-1, on the wave whose block of the product above is the cheapest
            if (g_thr) s_y[gt] = defer_y ? y_row<TS, LD>(sL, gt, s_rs) : s_rs[gt];
            __syncthreads();
            TS_MARK(2); // L(k,q) formed
#pragma unroll
            for (int u = 0; u < PER; ++u)
                if (own[u]) accU[u] = mfma_block_acc<TS, LD>(sT, 16 * ba[u], sT, 16 * bb[u], accU[u]);
            if (g_thr) { // (on the last wave: it owns no block of the product above)
                double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll 6
                for (int m = 0; m < TS; m += 4) {
                    a0 = fma(sT[gt * LD + m], s_y[m], a0);
                    a1 = fma(sT[gt * LD + m + 1], s_y[m + 1], a1);
                    a2 = fma(sT[gt * LD + m + 2], s_y[m + 2], a2);
                    a3 = fma(sT[gt * LD + m + 3], s_y[m + 3], a3);
                }
                gk -= (a0 + a1) + (a2 + a3);
            }
            __syncthreads();
            TS_MARK(3); // pending update applied
        }
    } else {
    TileRegs<TS> pre;
    double ypre = 0.0;
    if (npre > 0) { tile_load<TS>(Lt + (size_t)pre_tile(0) * TS * TS, pre); if (tid < TS) ypre = y[pre_col(0) * TS + tid]; }
    if (failed_before != 0) return kAborted;
    for (int w = 0; w < npre; ++w) {
        tile_store<TS>(pre, sL);
        if (tid < TS) s_rs[tid] = ypre;
        __syncthreads();
        // the next source tile travels while this one is applied
        if (w + 1 < npre) { tile_load<TS>(Lt + (size_t)pre_tile(w + 1) * TS * TS, pre); if (tid < TS) ypre = y[pre_col(w + 1) * TS + tid]; }
#pragma unroll
        for (int u = 0; u < PER; ++u)
            if (own[u]) accU[u] = mfma_block_acc<TS, LD>(sL, 16 * ba[u], sL, 16 * bb[u], accU[u]);
        if (g_thr) { // four partial sums: a single chain of TS dependent FP64 FMAs (36 cycles each) would cost 1.4 us
            double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll 6
            for (int m = 0; m < TS; m += 4) {
                a0 = fma(sL[gt * LD + m], s_rs[m], a0);
                a1 = fma(sL[gt * LD + m + 1], s_rs[m + 1], a1);
                a2 = fma(sL[gt * LD + m + 2], s_rs[m + 2], a2);
                a3 = fma(sL[gt * LD + m + 3], s_rs[m + 3], a3);
            }
            gk -= (a0 + a1) + (a2 + a3);
        }
        __syncthreads();
    }
    }
    if (g_thr) {
        s_g[gt] = gk; // g_k minus the pending sources: read again when y_k is formed, many barriers from here
        if (defer_y) y[k * TS + gt] = gk; // (final: every earlier column has subtracted its part; the consumers form y_k from it)
    }
#pragma unroll
    for (int u = 0; u < PER; ++u)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = 16 * ba[u] + lk + 4 * q, c = 16 * bb[u] + ln;
            double v = acc[u][q];
            if (own[u] && r == c && k * TS + r < n) v += lambda; // g2o setLambda: H_jj += lambda on real rows
            acc[u][q] = v - accU[u][q];
        }
    if (stop_after == 5) { if (tid < TS) y[k * TS + tid] = acc[0][0]; return kOk; }
    long long t_clk0 = 0, t_rt0 = 0;
    const bool probe = stop_after >= 6 && stop_after <= 9;
    if (probe) { t_clk0 = clock64(); t_rt0 = wall_clock64(); }
    TS_MARK(4);
    bool fail = false;
    // (A look-ahead variant - next quad's block column updated and published first, the next pivot chain and the other
    // blocks behind the same barrier, the two waves of a SIMD in opposite order - is correct but measured 33 us per
    // tile instead of 25.6: the chain does not hide behind the other wave's work.)
    for (int jb = 0; jb < NB; ++jb) {
        for (int jq = 0; jq < 16 / KB; ++jq) {
            const int jx = KB * jq, j = 16 * jb + jx;
            double (*col)[4] = s_col[jq & 1]; // [row][pivot of the quad]: the four values of a row are one 32-byte read
            // the four columns of the quad, from the blocks of block column jb
#pragma unroll
            for (int u = 0; u < PER; ++u)
                if (own[u] && bb[u] == jb && (ln >> 2) == jq) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) col[16 * ba[u] + lk + 4 * q][ln & 3] = acc[u][q];
                }
            if (jb == 1 && jq == 1) TS_MARK(12); // (one quad in the middle of the sweep: published)
            __syncthreads();
            if (jb == 1 && jq == 1) TS_MARK(13); // past the barrier
            // KB x KB pivot block, LDL' in registers (every lane redundantly)
            double w[KB][KB], l[KB][KB], rinv[KB];
#pragma unroll
            for (int m = 0; m < KB; ++m) {
#pragma unroll
                for (int q = 0; q <= m; ++q) {
                    double v = col[j + m][q];
#pragma unroll
                    for (int t = 0; t < q; ++t) v = fma(-w[m][t], l[q][t], v);
                    w[m][q] = v;
                    if (q < m) l[m][q] = v * rinv[q];
                }
                const double dm = w[m][m];
                if (!(dm > 0.0)) fail = true; // uniform: every lane sees the same pivot block
                rinv[m] = fast_rcp(dm);
            }
            if (fail) break;
            // Panel entries by the INVERSE of the unit-triangular pivot factor: the eliminated entry of pivot lk at a row
            // is then one short dot product of the four published values with this lane's coefficient row (no chain of
            // dependent FP64 operations and no per-solve selects), the 1/d scaling of the row operand folded in.
            if (jb == 1 && jq == 1) TS_MARK(14); // pivot chain done
            const double c10 = -l[1][0], c21 = -l[2][1], c32 = -l[3][2];
            const double c20 = fma(-l[2][1], c10, -l[2][0]), c31 = fma(-l[3][2], c21, -l[3][1]);
            const double c30 = fma(-l[3][2], c20, fma(-l[3][1], c10, -l[3][0]));
            const double cl0 = lk == 0 ? 1.0 : lk == 1 ? c10 : lk == 2 ? c20 : c30;
            const double cl1 = lk == 0 ? 0.0 : lk == 1 ? 1.0 : lk == 2 ? c21 : c31;
            const double cl2 = lk < 2 ? 0.0 : lk == 2 ? 1.0 : c32;
            const double cl3 = lk == 3 ? 1.0 : 0.0;
            const double rk = -(lk == 0 ? rinv[0] : lk == 1 ? rinv[1] : lk == 2 ? rinv[2] : rinv[3]);
            const double ca0 = cl0 * rk, ca1 = cl1 * rk, ca2 = cl2 * rk, ca3 = cl3 * rk;
#pragma unroll
            for (int u = 0; u < PER; ++u)
                if (own[u] && bb[u] >= jb) {
                    const int ra = 16 * ba[u] + ln, rb = 16 * bb[u] + ln;
                    const double2 a01 = *reinterpret_cast<const double2*>(&col[ra][0]), a23 = *reinterpret_cast<const double2*>(&col[ra][2]);
                    const double2 b01 = *reinterpret_cast<const double2*>(&col[rb][0]), b23 = *reinterpret_cast<const double2*>(&col[rb][2]);
                    const double a0 = a01.x, a1 = a01.y, a2 = a23.x, a3 = a23.y, b0 = b01.x, b1 = b01.y, b2 = b23.x, b3 = b23.y;
                    const double wa = fma(ca1, a1, ca0 * a0) + fma(ca3, a3, ca2 * a2); // = -w[lk] / d[lk] at row ra
                    const double wb = fma(cl1, b1, cl0 * b0) + fma(cl3, b3, cl2 * b2); // =  w[lk]         at row rb
                    // rows / columns up to the pivot stay as they are (in block row / column jb only)
                    const double av = (ba[u] == jb && ln <= jx + lk) ? 0.0 : wa;
                    const double bv = (bb[u] == jb && ln <= jx + lk) ? 0.0 : wb;
                    acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[u], 0, 0, 0);
                }
            // the pivots go out LAST: a store inside the chain (or in front of the panel reads) would put its LDS round
            // trip, through the in-order wait for the next reads, on the dependent path
            if (jb == 1 && jq == 1) TS_MARK(15); // panel + matrix-core step issued
            if (tid == kPotrfThreads - 64) { // (a wave without blocks: on wave 0 these stores delayed its next quad, 50 cycles each)
#pragma unroll
                for (int m = 0; m < KB; ++m) s_rs[j + m] = w[m][m];
                // the inverse of the quad's unit factor is the 4x4 diagonal block of L^-1 up to the row scaling: kept for the
                // inversion stage (which then starts at 4 -> 8)
                double2* cf = reinterpret_cast<double2*>(s_cf + 6 * (j / KB));
                cf[0] = make_double2(c10, c20); cf[1] = make_double2(c21, c30); cf[2] = make_double2(c31, c32);
            }
        }
        if (fail) break;
    }
    __syncthreads();
    TS_MARK(5); // pivot sweep done
    if (fail) return kNotPositive;
    if (probe) { // shader cycles and 100 MHz ticks spent in the pivot sweep
        if (tid == 0) { y[0] = (double)(clock64() - t_clk0); y[1] = (double)(wall_clock64() - t_rt0); y[2] = acc[0][0]; }
        return kOk;
    }
    if (stop_after == 1) { if (tid < TS) y[k * TS + tid] = acc[0][0]; return kOk; }
    if (tid < TS) s_rs[tid] = fast_rsqrt(s_rs[tid]);
    for (int i = tid; i < TS * LD; i += kPotrfThreads) sX[i] = 0.0; // X is read whole (y_k, the store): zero above the diagonal
    __syncthreads();
    // L[r][c] = A[r][c] / sqrt(d_c) -> LDS image.  Only its lower triangle is ever read (by the inverse below), and the
    // diagonal factor itself is needed nowhere else - trsm, the updates and the substitutions work with L_kk^-1 - so
    // neither the upper triangle of the image nor a copy in global memory is written.
#pragma unroll
    for (int u = 0; u < PER; ++u)
        if (own[u]) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r = 16 * ba[u] + lk + 4 * q, c = 16 * bb[u] + ln;
                if (c <= r) sL[r * LD + c] = acc[u][q] * s_rs[c];
            }
        }
    // X = L^-1, first stage: the 4x4 diagonal blocks.  L_qq = Lu D^1/2 with the quad's unit factor Lu, so
    // L_qq^-1 = D^-1/2 Lu^-1 - Lu^-1 is what the sweep solved its panels with (c10 .. c32 above).
    if (tid < TS) {
        const int o = (tid >> 2) * 4, c = tid & 3;
        const double* cf = s_cf + 6 * (tid >> 2);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const double v = r < c ? 0.0 : r == c ? 1.0 : cf[r * (r - 1) / 2 + c];
            sX[(o + r) * LD + o + c] = v * s_rs[o + r];
        }
    }
    TS_MARK(6); // L image, 4x4 inverses
    return kOk;
}

// one 48x48 block of S(c) -= L(a) L(b)' by a 512-thread workgroup (defined below)
struct StepArgs {
    int n_chain;                 // workgroups 0..n_chain-1 factorise one column each, the rest run grouped updates
    int last_level;              // nothing is eliminated after these columns: they finish x_k themselves
    int defer_y;                 // the forward vector y_k = L_kk^-1 g_k is NOT formed by the workgroup that factorises column k (1450 cycles at
                                 // the very end of a level's critical path): it leaves the final g_k in the y buffer, and whoever needs y_k -
                                 // the next level's pending updates, the grouped updates' g rows, the backward substitution - multiplies
                                 // by L_kk^-1, which it has staged anyway
    const int* chain_col;        // [n_chain] tile columns of this level
    const int4* chain_desc;      // [n_chain] pairs of int4: {column, diagonal tile, first pre, #pre}, {first sub-diagonal tile, #, 0, 0}
    const int* diag_tile;        // [NT]
    const int* pre_ptr;          // [NT+1]
    const int* pre_tile;
    const int* pre_col;
    const int* tgt_tile;         // grouped updates of this launch (already offset to the level)
    const int* tgt_row;
    const int* tgt_pair_ptr;
    const int* pair_a;
    const int* pair_b;
    const int* pair_src;
    ChainInline inl;             // the chain workgroups' records, if the level fits
};

template <int TS, bool SPLIT>
__device__ void gemm_target_block(double* __restrict__ S, double* __restrict__ Su, double* __restrict__ Lt, const double* __restrict__ Linv, const StepArgs& sa, int work,
                                  double* __restrict__ g, double* __restrict__ gu, const double* __restrict__ y, double* sm);

// One dependency level of the factorisation.  Workgroups 0..n_chain-1 are the critical path: each applies the
// pending updates of its diagonal tile from the level just below, then factorises it (potrf + inverse + y_k).
// The other workgroups carry every remaining update whose source column sits in the level just below, grouped
// by target tile - nothing on the critical path waits for them inside this launch.
template <int TS, bool SPLIT = false>
__global__ __launch_bounds__(kPotrfThreads) void k_potrf_inv(double* __restrict__ S, double* __restrict__ Lt, double* __restrict__ Linv,
                                                             double* __restrict__ g, double* __restrict__ y, int n, double lambda,
                                                             int* status, int stop_after, StepArgs sa, double* xs, double* __restrict__ Su,
                                                             double* __restrict__ gu)
{
    constexpr int NB = TS / 16, LD = Lds<TS>::LD;
    extern __shared__ __align__(16) double sm[];
    double* sL = sm;            // [TS][LD]
    double* sX = sm + TS * LD;  // [TS][LD]
    // scratch of the pivot sweep: the published columns, [parity][row][column of the quad]
    constexpr int kScratch = 2 * 4 * TS;
    __shared__ __align__(16) double s_buf[kScratch];
    __shared__ double s_rs[TS];        // 1/sqrt(d_j) = 1/L_jj
    __shared__ double s_g[TS];
    __shared__ double s_y[TS];         // y_q of the pending source being applied
    __shared__ __align__(16) double s_cf[(TS / 4) * 6]; // Lu^-1 of every quad (below the diagonal)
    static_assert(sizeof(double) * (2 * TS * LD + kScratch + 2 * TS) <= 160 * 1024, "potrf LDS budget (160 KiB per workgroup)");
    double (*s_col)[TS][4] = reinterpret_cast<double (*)[TS][4]>(s_buf);
    const int tid = threadIdx.x;
#ifdef POTRF_TS
    if (tid == 0 && blockIdx.x == 0) g_ts_cur = atomicAdd(&g_ts_cnt, 1) & 63;
#endif
    TS_MARK(0);
    if ((int)blockIdx.x >= sa.n_chain) {
        if (*status != 0) return;
        gemm_target_block<TS, SPLIT>(S, Su, Lt, Linv, sa, (int)blockIdx.x - sa.n_chain, g, gu, y, sm);
        return;
    }
    // the column record: out of the kernel arguments when the level fits there (no index load in front of the tile loads),
    // else one 16-byte record; the status word is tested inside the sweep, behind the operand loads it must not delay
    const bool inl = sa.inl.n > 0;
    int k, tile_id, pre0 = 0, npre;
    ChainRec rec = sa.inl.c[0]; // (selected with constant indices: indexing the argument would move it to scratch memory)
#pragma unroll
    for (int i = 1; i < kInlineCols; ++i) if ((int)blockIdx.x == i) rec = sa.inl.c[i];
    int it0 = rec.pre_tile[0], it1 = rec.pre_tile[1], ic0 = rec.pre_col[0], ic1 = rec.pre_col[1];
    if (inl) { k = rec.k; tile_id = rec.tile; npre = rec.npre; }
    else {
        const int4 ds = sa.chain_desc[2 * (int)blockIdx.x];
        k = __builtin_amdgcn_readfirstlane(ds.x); tile_id = __builtin_amdgcn_readfirstlane(ds.y);
        pre0 = __builtin_amdgcn_readfirstlane(ds.z); npre = __builtin_amdgcn_readfirstlane(ds.w);
        // (read to scalar registers INSIDE this branch: a vector register that may or may not have a load pending at the join
        // makes the compiler wait, at its first use, for everything issued in between - on the inline path as well)
        const int a0 = npre > 0 ? sa.pre_tile[pre0] : 0, b0 = npre > 0 ? sa.pre_col[pre0] : 0;
        const int a1 = npre > 1 ? sa.pre_tile[pre0 + 1] : 0, b1 = npre > 1 ? sa.pre_col[pre0 + 1] : 0;
        it0 = __builtin_amdgcn_readfirstlane(a0); ic0 = __builtin_amdgcn_readfirstlane(b0);
        it1 = __builtin_amdgcn_readfirstlane(a1); ic1 = __builtin_amdgcn_readfirstlane(b1);
    }
    const double* A = S + (size_t)tile_id * TS * TS;
    double* Lg = Lt + (size_t)tile_id * TS * TS;
    // g_k is requested here and parked in LDS by the sweep once the operand tiles are on their way (a store to LDS right here
    // would wait for this load alone: one memory round trip in front of all the others)
    double gk = 0.0;
    {
        const int gt = tid - (kPotrfThreads - 64 * ((TS + 63) / 64)); // (the sweep keeps g_k on the last wave(s))
        if (gt >= 0 && gt < TS) { gk = g[k * TS + gt]; if constexpr (SPLIT) gk += gu[k * TS + gt]; }
    }
    const int rc = potrf_sweep_mfma<TS, SPLIT>(A, SPLIT ? Su + (size_t)tile_id * TS * TS : nullptr, Su, Lg, sL, sX, s_col, s_rs, k, n, lambda, stop_after, y, Lt, sa.pre_tile + pre0, sa.pre_col + pre0, npre, s_g, S,
                                        Linv, sm + 2 * TS * LD, status, it0, it1, ic0, ic1, gk, s_cf, Fold<TS>::on ? sa.defer_y : 0, s_y);
    if (rc == 2) return;                                  // an earlier column of this trial had failed
    if (rc == 1) { if (tid == 0) *status = k + 1; return; } // not positive definite
    if (stop_after == 5 || (stop_after >= 6 && stop_after <= 9) || stop_after == 1) return;
    __syncthreads();
    if (stop_after == 2) return;
    // ---- X = L^-1 ----
    // (1) inverses of the 16x16 diagonal blocks by doubling, 4 -> 8 -> 16: X21 = -X22 (L21 X11) of every pair of neighbours, both
    //     stages on the matrix cores (one thread per entry on the vector ALU - 72 multiply-adds and 136 LDS reads per thread in the
    //     last stage - was 1.3 us of a tile; a forward substitution per column before that 1.35 us).  sX is zero above the diagonal.
    // (the 4x4 diagonal blocks of X were written by the sweep's last phase, from the unit factors it had inverted anyway)
    if (tid < 64 * ((TS / 8 + 3) / 4)) { // 4 -> 8 on the matrix cores: X21 = -X22 (L21 X11) of the TS/8 blocks of eight, four blocks per instruction
        const int lane = tid & 63, blk = (lane >> 2) & 3, hi = lane >> 4, lo = lane & 3;
        const int bq = 4 * (tid >> 6) + blk;
        const bool live = bq < TS / 8;
        const int o = 8 * (live ? bq : 0);
        // v_mfma_f64_4x4x4_4b: A[blk][i][k] in lane 16 k + 4 blk + i, B[blk][k][j] in lane 16 k + 4 blk + j, D[blk][i][j] in lane
        // 16 i + 4 blk + j (tools/mfma_f64_4x4x4_layout.hip) - D of the first product is the B operand of the second as it stands
        const double a1 = sL[(o + 4 + lo) * LD + o + hi], b1 = sX[(o + hi) * LD + o + lo];
        const double a2 = sX[(o + 4 + lo) * LD + o + 4 + hi];
        const double t = __builtin_amdgcn_mfma_f64_4x4x4f64(a1, b1, 0.0, 0, 0, 0);
        const double r = __builtin_amdgcn_mfma_f64_4x4x4f64(a2, t, 0.0, 0, 0, 0);
        if (live) sX[(o + 4 + hi) * LD + o + lo] = -r;
    }
    __syncthreads();
    if (tid < (TS / 16) * 64) { // 8 -> 16, one wave per block of sixteen: rows 8..15 of (L X) over k < 8, then X22 times that
        const int o = (tid >> 6) * 16, lane = tid & 63, m = lane & 15, kq = lane >> 4;
        // rows < 8 of both products are not used (their A rows hold L11 / zeros): only rows 8..15, columns 0..7 are stored
        v4f64 t = {0.0, 0.0, 0.0, 0.0}, r = {0.0, 0.0, 0.0, 0.0};
        double a[2], bb[2], a2[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            a[q] = m >= 8 ? sL[(o + m) * LD + o + 4 * q + kq] : 0.0;
            bb[q] = sX[(o + 4 * q + kq) * LD + o + m];
            a2[q] = sX[(o + m) * LD + o + 8 + 4 * q + kq];
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) t = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q], bb[q], t, 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 2; ++q) r = __builtin_amdgcn_mfma_f64_16x16x4f64(a2[q], t[2 + q], r, 0, 0, 0);
        if (m < 8) {
#pragma unroll
            for (int q = 2; q < 4; ++q) sX[(o + kq + 4 * q) * LD + o + m] = -r[q];
        }
    }
    __syncthreads();
    TS_MARK(7); // 16x16 diagonal blocks of X
    if (stop_after == 3) return;
    // (2) off-diagonal blocks: X_ij = -X_ii * sum_{m=j}^{i-1} L_im X_mj
    if constexpr (NB == 3) {
        // three blocks.  X_10 and X_21 on waves 0 and 1, and wave 2 starts X_20 with the product that needs neither (L_20 X_00);
        // after a barrier it adds L_21 X_10 and multiplies by X_22 - a chain of eight dependent matrix-core steps on either side
        // of the barrier, where one wave walking block column 0 made twenty.
        const int wave = tid >> 6, lane = tid & 63;
        auto lx = [&](int i, int m, int j, v4f64 acc) { // acc += L_im X_mj
            const double* pa = sL + (16 * i + (lane & 15)) * LD + 16 * m + (lane >> 4);
            const double* pb = sX + (16 * m + (lane >> 4)) * LD + 16 * j + (lane & 15);
            double av[4], bv[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) { av[q] = pa[4 * q]; bv[q] = pb[4 * q * LD]; }
#pragma unroll
            for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[q], bv[q], acc, 0, 0, 0);
            return acc;
        };
        auto finish = [&](int i, int j, const v4f64& acc) { // X_ij = -X_ii acc: the accumulator layout IS the B-operand layout
            const double* pd = sX + (16 * i + (lane & 15)) * LD + 16 * i + (lane >> 4);
            double dv[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) dv[q] = pd[4 * q];
            v4f64 r = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int q = 0; q < 4; ++q) r = __builtin_amdgcn_mfma_f64_16x16x4f64(dv[q], acc[q], r, 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 4; ++q) sX[(16 * i + (lane >> 4) + 4 * q) * LD + 16 * j + (lane & 15)] = -r[q];
        };
        v4f64 acc = {0.0, 0.0, 0.0, 0.0};
        if (wave == 0) { acc = lx(1, 0, 0, acc); finish(1, 0, acc); }
        else if (wave == 1) { acc = lx(2, 1, 1, acc); finish(2, 1, acc); }
        else if (wave == 2) acc = lx(2, 0, 0, acc);
        __syncthreads();
        if (wave == 2) { acc = lx(2, 1, 0, acc); finish(2, 0, acc); }
    } else {
        {
            const int wave = tid >> 6, lane = tid & 63;
            for (int j = wave; j < NB - 1; j += kPotrfThreads / 64) {
                for (int i = j + 1; i < NB; ++i) {
                    // X_ii operand first: it depends on nothing of this step
                    const double* pd = sX + (16 * i + (lane & 15)) * LD + 16 * i + (lane >> 4);
                    double dv[4];
    #pragma unroll
                    for (int q = 0; q < 4; ++q) dv[q] = pd[4 * q];
                    v4f64 acc = {0.0, 0.0, 0.0, 0.0};
                    for (int m = j; m < i; ++m) {
                        const double* pa = sL + (16 * i + (lane & 15)) * LD + 16 * m + (lane >> 4);
                        const double* pb = sX + (16 * m + (lane >> 4)) * LD + 16 * j + (lane & 15);
                        double av[4], bv[4];
    #pragma unroll
                        for (int q = 0; q < 4; ++q) { av[q] = pa[4 * q]; bv[q] = pb[4 * q * LD]; }
    #pragma unroll
                        for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[q], bv[q], acc, 0, 0, 0);
                    }
                    // the accumulator layout (row (lane>>4) + 4q, column lane&15) IS the B-operand layout of the next product
                    // (k = (lane>>4) + 4q, n = lane&15): no trip through LDS
                    v4f64 r = {0.0, 0.0, 0.0, 0.0};
    #pragma unroll
                    for (int q = 0; q < 4; ++q) r = __builtin_amdgcn_mfma_f64_16x16x4f64(dv[q], acc[q], r, 0, 0, 0);
    #pragma unroll
                    for (int q = 0; q < 4; ++q) sX[(16 * i + (lane >> 4) + 4 * q) * LD + 16 * j + (lane & 15)] = -r[q];
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                }
            }
        }
    }
    __syncthreads();
    TS_MARK(8); // off-diagonal blocks of X
    if (stop_after == 4) return;
    {
        double* X = Linv + (size_t)k * TS * TS;
#pragma unroll
        for (int it = 0; it < (TS * TS + kPotrfThreads - 1) / kPotrfThreads; ++it) {
            const int q = it * kPotrfThreads + tid;
            if ((TS * TS) % kPotrfThreads == 0 || q < TS * TS) X[q] = sX[(q / TS) * LD + (q % TS)];
        }
    }
    TS_MARK(9); // X stored
    if (Fold<TS>::on && sa.defer_y && !sa.last_level) { // y_k is formed by its consumers; the backward substitution waits on the marker
        if (tid < TS && xs != y) xs[k * TS + tid] = solve_pending();
        TS_MARK(10);
        return;
    }
    // y_k = L_kk^-1 g_k (g_k is final: every earlier column has already subtracted its part).  Four threads per row, a quarter of
    // the columns each, the partial sums added in a fixed order: one thread per row was a chain of TS LDS reads and multiply-adds
    // (0.65 us) at the very end of the level.
    constexpr int YP = 4, YC = TS / YP;
    static_assert(YP * TS <= kPotrfThreads && YP * TS <= kScratch && YC % 4 == 0, "y_k partial sums");
    double* s_yp = s_buf; // [YP][TS]: the sweep's column scratch is free by now
    if (tid < YP * TS) {
        const int part = tid / TS, r = tid - part * TS;
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0; // X is zero above the diagonal
#pragma unroll
        for (int c = part * YC; c < part * YC + YC; c += 4) {
            a0 = fma(sX[r * LD + c], s_g[c], a0);
            a1 = fma(sX[r * LD + c + 1], s_g[c + 1], a1);
            a2 = fma(sX[r * LD + c + 2], s_g[c + 2], a2);
            a3 = fma(sX[r * LD + c + 3], s_g[c + 3], a3);
        }
        s_yp[part * TS + r] = (a0 + a1) + (a2 + a3);
    }
    __syncthreads();
    if (tid < TS) {
        const double yk = (s_yp[tid] + s_yp[TS + tid]) + (s_yp[2 * TS + tid] + s_yp[3 * TS + tid]);
        if (!sa.last_level) {
            y[k * TS + tid] = yk;
            if (xs != y) xs[k * TS + tid] = solve_pending(); // the one-launch backward substitution waits on this value
        } else s_rs[tid] = yk; // (1/L_jj is no longer needed)
    }
    if (sa.last_level) {
        // the columns of the last level have no rows below them: x_k = L_kk^-T y_k needs nothing else, and the backward
        // substitution starts one level (one launch on the critical path) further down
        __syncthreads();
        if (tid < TS) {
            double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll 6
            for (int r = 0; r < TS; r += 4) { // X is zero above the diagonal
                a0 = fma(sX[r * LD + tid], s_rs[r], a0);
                a1 = fma(sX[(r + 1) * LD + tid], s_rs[r + 1], a1);
                a2 = fma(sX[(r + 2) * LD + tid], s_rs[r + 2], a2);
                a3 = fma(sX[(r + 3) * LD + tid], s_rs[r + 3], a3);
            }
            xs[k * TS + tid] = (a0 + a1) + (a2 + a3);
        }
    }
    TS_MARK(10);
}

// ---------------------------------------------------------------------------------------------
// trsm: L(i,k)[block] = S(i,k)[rows] * Linv_kk[cols]'      grid: (#tiles * (TS/48)^2)
// ---------------------------------------------------------------------------------------------
// output block edge of the trsm workgroups: 32 (four 16x16 MFMA blocks, one per wave) where the tile allows it
template <int TS> struct TrsmBlock { static constexpr int TB = (TS % 32 == 0) ? 32 : kOB; };

template <int TS>
__global__ __launch_bounds__(kBlock) void k_trsm(const double* __restrict__ S, double* __restrict__ Lt, const double* __restrict__ Linv,
                                                 const int* __restrict__ list, const int* __restrict__ list_col, const int* status)
{
    constexpr int LD = Lds<TS>::LD, kTB = TrsmBlock<TS>::TB, Q = TS / kTB, NBK = kTB / 16;
    static_assert(TS % kTB == 0 && kBlock == 256, "trsm block shape");
    extern __shared__ __align__(16) double sm[];
    const int failed = *status; // tested after the operand loads are on their way
    double* sA = sm;
    double* sB = sm + kTB * LD;
    const int item = blockIdx.x / (Q * Q);
    const int t = list[item], k = list_col[item], qq = blockIdx.x % (Q * Q), qr = qq / Q, qc = qq % Q;
    rows_to_lds<TS, kTB>(S + (size_t)t * TS * TS, kTB * qr, sA);
    rows_to_lds<TS, kTB>(Linv + (size_t)k * TS * TS, kTB * qc, sB);
    __syncthreads();
    if (failed != 0) return;
    double* out = Lt + (size_t)t * TS * TS;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int st = wave; st < NBK * NBK; st += 4) {
        const int r0 = (st / NBK) * 16, c0 = (st % NBK) * 16;
        const v4f64 acc = mfma_block<TS, LD>(sA, r0, sB, c0);
#pragma unroll
        for (int q = 0; q < 4; ++q) out[(size_t)(kTB * qr + r0 + (lane >> 4) + 4 * q) * TS + kTB * qc + c0 + (lane & 15)] = acc[q];
    }
}

// ---------------------------------------------------------------------------------------------
// grouped update of ONE target block: S(c)[block] -= sum_q L(i,q)[rows] L(j,q)[cols]' over the source columns q of
// the level just below, in ascending q; diagonal targets also carry g_i -= sum_q L(i,q) y_q.
// Runs in the extra workgroups (512 threads) of the level kernel; the 9 16x16 sub-blocks stay in the MFMA
// accumulators across the sources, the tile is read-modified-written once.
// ---------------------------------------------------------------------------------------------

template <int TS, bool SPLIT>
__device__ void gemm_target_block(double* __restrict__ S, double* __restrict__ Su, double* __restrict__ Lt, const double* __restrict__ Linv, const StepArgs& sa, int work,
                                  double* __restrict__ g, double* __restrict__ gu, const double* __restrict__ y, double* sm)
{
    if constexpr (Fold<TS>::on) {
        // tile edge 48 = one output block per target: the operand tiles L(i,q), L(j,q) are formed here from S and L_qq^-1
        // (no k_trsm launch); the diagonal target of row i stores L(i,q) for the backward substitution
        constexpr int LD = Lds<TS>::LD, NB = TS / 16;
        static_assert(TS == kOB, "one 48 x 48 block per target tile");
        double* sSa = sm;
        double* sSb = sm + TS * LD;
        double* sXq = sm + 2 * TS * LD;
        double* sLa = sm + 3 * TS * LD;
        double* sLb = sm + 4 * TS * LD;
        __shared__ double s_yq[TS];
        const int t = work;
        const int p0 = sa.tgt_pair_ptr[t], p1 = sa.tgt_pair_ptr[t + 1];
        const int row = sa.tgt_row[t];
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        constexpr int NWV = kPotrfThreads / 64;
        v4f64 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
        double gacc = 0.0;
        const bool g_thread = row >= 0 && threadIdx.x >= 64 && threadIdx.x < 64 + TS;
        TileRegs<TS> ra, rb, rx, rau, rbu;
        auto fetch = [&](int q) {
            tile_load<TS>(S + (size_t)sa.pair_a[q] * TS * TS, ra);
            if (sa.pair_b[q] != sa.pair_a[q]) tile_load<TS>(S + (size_t)sa.pair_b[q] * TS * TS, rb);
            if constexpr (SPLIT) {
                tile_load<TS>(Su + (size_t)sa.pair_a[q] * TS * TS, rau);
                if (sa.pair_b[q] != sa.pair_a[q]) tile_load<TS>(Su + (size_t)sa.pair_b[q] * TS * TS, rbu);
            }
            tile_load<TS>(Linv + (size_t)sa.pair_src[q] * TS * TS, rx);
        };
        if (p0 < p1) fetch(p0);
        for (int q = p0; q < p1; ++q) {
            const bool same = sa.pair_a[q] == sa.pair_b[q]; // uniform
            if constexpr (SPLIT) { tile_add<TS>(ra, rau); if (!same) tile_add<TS>(rb, rbu); }
            tile_store<TS>(ra, sSa);
            if (!same) tile_store<TS>(rb, sSb);
            tile_store<TS>(rx, sXq);
            __syncthreads();
            double* Lout = Lt + (size_t)sa.pair_a[q] * TS * TS;
            const double* yq = y + sa.pair_src[q] * TS;
            if (q + 1 < p1) fetch(q + 1);
            for (int st = wave; st < (same ? 1 : 2) * NB * NB; st += NWV) {
                const int which = st / (NB * NB), bq = st % (NB * NB), r0 = (bq / NB) * 16, c0 = (bq % NB) * 16;
                const v4f64 lb = mfma_block_tri<TS, LD>(which ? sSb : sSa, r0, sXq, c0);
                block_to_lds<LD>(lb, which ? sLb : sLa, r0, c0);
                if (same) block_to_global<TS>(lb, Lout, r0, c0); // the diagonal target of row i owns L(i,q)
            }
            if (g_thread && sa.defer_y) s_yq[threadIdx.x - 64] = y_row<TS, LD>(sXq, threadIdx.x - 64, yq); // (yq points at the final g_q)
            __syncthreads();
            const double* pa = sLa;
            const double* pb = same ? sLa : sLb;
            {
                const int st0 = wave, st1 = wave + NWV;
                acc0 = mfma_block_acc<TS, LD>(pa, (st0 / 3) * 16, pb, (st0 % 3) * 16, acc0);
                if (st1 < 9) acc1 = mfma_block_acc<TS, LD>(pa, (st1 / 3) * 16, pb, (st1 % 3) * 16, acc1);
            }
            if (g_thread) { // forward substitution rides along: g_i -= L_iq y_q
                const int r = threadIdx.x - 64;
                const double* yv = sa.defer_y ? s_yq : yq; // (defer_y: formed below the first barrier from the staged L_qq^-1)
                double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll 6
                for (int m = 0; m < TS; m += 4) {
                    a0 = fma(sLa[r * LD + m], yv[m], a0);
                    a1 = fma(sLa[r * LD + m + 1], yv[m + 1], a1);
                    a2 = fma(sLa[r * LD + m + 2], yv[m + 2], a2);
                    a3 = fma(sLa[r * LD + m + 3], yv[m + 3], a3);
                }
                gacc += (a0 + a1) + (a2 + a3);
            }
            __syncthreads();
        }
        double* C = (SPLIT ? Su : S) + (size_t)sa.tgt_tile[t] * TS * TS;
        {
            const int st0 = wave, st1 = wave + NWV;
            const int r0 = (st0 / 3) * 16, c0 = (st0 % 3) * 16;
#pragma unroll
            for (int q = 0; q < 4; ++q) C[(size_t)(r0 + (lane >> 4) + 4 * q) * TS + c0 + (lane & 15)] -= acc0[q];
            if (st1 < 9) {
                const int r1 = (st1 / 3) * 16, c1 = (st1 % 3) * 16;
#pragma unroll
                for (int q = 0; q < 4; ++q) C[(size_t)(r1 + (lane >> 4) + 4 * q) * TS + c1 + (lane & 15)] -= acc1[q];
            }
        }
        if (g_thread) (SPLIT ? gu : g)[row * TS + (threadIdx.x - 64)] -= gacc;
        return;
    } else {
    constexpr int LD = Lds<TS>::LD, Q = TS / kOB;
    double* sA = sm;
    double* sB = sm + kOB * LD;
    const int t = work / (Q * Q), qq = work % (Q * Q), qr = qq / Q, qc = qq % Q;
    const int p0 = sa.tgt_pair_ptr[t], p1 = sa.tgt_pair_ptr[t + 1];
    const int row = sa.tgt_row[t];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int st0 = wave, st1 = wave + kPotrfThreads / 64; // sub-blocks of this wave (st1 only for wave 0)
    v4f64 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
    double gacc = 0.0;
    const bool g_thread = row >= 0 && qc == 0 && threadIdx.x >= 64 && threadIdx.x < 64 + kOB;
    // the operand rows of source q+1 travel (in registers) while source q is multiplied out of LDS
    constexpr int N2 = kOB * TS / 2, IT = (N2 + kPotrfThreads - 1) / kPotrfThreads;
    double2 ra[IT], rb[IT];
    auto fetch = [&](int q) {
        const double2* pa = reinterpret_cast<const double2*>(Lt + (size_t)sa.pair_a[q] * TS * TS + (size_t)kOB * qr * TS);
        const double2* pb = reinterpret_cast<const double2*>(Lt + (size_t)sa.pair_b[q] * TS * TS + (size_t)kOB * qc * TS);
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int i = it * kPotrfThreads + (int)threadIdx.x;
            if (N2 % kPotrfThreads == 0 || i < N2) { ra[it] = pa[i]; rb[it] = pb[i]; }
        }
    };
    auto park = [&]() {
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int i = it * kPotrfThreads + (int)threadIdx.x;
            if (N2 % kPotrfThreads == 0 || i < N2) {
                const int r = i / (TS / 2), c = 2 * (i % (TS / 2));
                sA[r * LD + c] = ra[it].x; sA[r * LD + c + 1] = ra[it].y;
                sB[r * LD + c] = rb[it].x; sB[r * LD + c + 1] = rb[it].y;
            }
        }
    };
    if (p0 < p1) fetch(p0);
    for (int q = p0; q < p1; ++q) {
        park();
        __syncthreads();
        if (q + 1 < p1) fetch(q + 1);
        acc0 = mfma_block_acc<TS, LD>(sA, (st0 / 3) * 16, sB, (st0 % 3) * 16, acc0);
        if (st1 < 9) acc1 = mfma_block_acc<TS, LD>(sA, (st1 / 3) * 16, sB, (st1 % 3) * 16, acc1);
        if (g_thread) { // forward substitution rides along: g_i -= L_iq y_q  (four partial sums: the chain is latency-bound)
            const int r = threadIdx.x - 64;
            const double* yq = y + sa.pair_src[q] * TS;
            double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll 6
            for (int m = 0; m < TS; m += 4) {
                a0 = fma(sA[r * LD + m], yq[m], a0);
                a1 = fma(sA[r * LD + m + 1], yq[m + 1], a1);
                a2 = fma(sA[r * LD + m + 2], yq[m + 2], a2);
                a3 = fma(sA[r * LD + m + 3], yq[m + 3], a3);
            }
            gacc += (a0 + a1) + (a2 + a3);
        }
        __syncthreads();
    }
    double* C = S + (size_t)sa.tgt_tile[t] * TS * TS;
    {
        const int r0 = (st0 / 3) * 16, c0 = (st0 % 3) * 16;
#pragma unroll
        for (int q = 0; q < 4; ++q) C[(size_t)(kOB * qr + r0 + (lane >> 4) + 4 * q) * TS + kOB * qc + c0 + (lane & 15)] -= acc0[q];
    }
    if (st1 < 9) {
        const int r0 = (st1 / 3) * 16, c0 = (st1 % 3) * 16;
#pragma unroll
        for (int q = 0; q < 4; ++q) C[(size_t)(kOB * qr + r0 + (lane >> 4) + 4 * q) * TS + kOB * qc + c0 + (lane & 15)] -= acc1[q];
    }
    if (g_thread) g[row * TS + kOB * qr + (threadIdx.x - 64)] -= gacc;
    }
}

// ---------------------------------------------------------------------------------------------
// backward substitution x_k = Linv_kk' (y_k - sum_{i>k} L_ik' x_i), x in place of y: one launch per dependency
// level (highest first), one workgroup per column of the level.  Thread (c, part) owns column c and a slice of
// the rows of each tile: all its loads of a tile are issued together (constant trip count), partial sums meet in LDS.
// ---------------------------------------------------------------------------------------------
template <int TS>
__global__ __launch_bounds__(kPotrfThreads) void k_back_solve(const double* __restrict__ Lt, const double* __restrict__ Linv, double* x, CholPlan p,
                                                              const int4* __restrict__ desc, const int* status)
{
    constexpr int NW = kPotrfThreads / 64;
    __shared__ double s_part[NW][TS];
    __shared__ double s_x[NW][TS];
    __shared__ double s_acc[TS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int failed = *status; // tested once the first loads are on their way
    const int4 d0 = desc[2 * blockIdx.x], d1 = desc[2 * blockIdx.x + 1];
    const int k = d0.x, q0 = d1.x, nq = d1.y;
    const int c0 = lane < TS ? lane : TS - 1, c1 = lane + 64 < TS ? lane + 64 : TS - 1;
    // this wave's rows of Linv_kk for step (2): they depend on nothing, so they travel while step (1) runs
    constexpr int RW = (TS + NW - 1) / NW;
    double xv0[RW], xv1[RW];
    {
        const double* X = Linv + (size_t)k * TS * TS;
#pragma unroll
        for (int rr = 0; rr < RW; ++rr) {
            const int r = wave * RW + rr < TS ? wave * RW + rr : TS - 1;
            xv0[rr] = X[r * TS + c0];
            if (TS > 64) xv1[rr] = X[r * TS + c1];
        }
    }
    if (failed != 0) return;
    // (1) sum_{i>k} L_ik' x_i : the sub-diagonal tiles of the column are spread over the waves, lane = column of the
    //     tile (and column + 64), rows streamed with the x_i entry broadcast from LDS
    double a0 = 0.0, a1 = 0.0, a0b = 0.0, a1b = 0.0;
    constexpr int RB = 24, NRB = TS / RB; // a work unit = RB rows of one tile; units go round the waves
    static_assert(TS % RB == 0 && RB % 8 == 0, "tile edge");
    for (int u = wave; u < nq * NRB; u += NW) {
        const int q = u / NRB, rb = u % NRB;
        const double* L = Lt + (size_t)p.trsm_tile[q0 + q] * TS * TS + (size_t)rb * RB * TS;
        const double* xi = x + (size_t)p.trsm_row[q0 + q] * TS + rb * RB;
        __builtin_amdgcn_wave_barrier();
        if (lane < RB) s_x[wave][lane] = xi[lane];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // unconditional, batched loads (8 rows in flight per lane); lanes beyond the tile read a clamped column
#pragma unroll
        for (int r8 = 0; r8 < RB; r8 += 8) {
            double v0[8], v1[8];
#pragma unroll
            for (int w = 0; w < 8; ++w) { v0[w] = L[(r8 + w) * TS + c0]; if (TS > 64) v1[w] = L[(r8 + w) * TS + c1]; }
#pragma unroll
            for (int w = 0; w < 8; w += 2) { // two chains per column: the FMAs are latency-bound
                const double xr = s_x[wave][r8 + w], xs = s_x[wave][r8 + w + 1];
                a0 = fma(v0[w], xr, a0); a0b = fma(v0[w + 1], xs, a0b);
                if (TS > 64) { a1 = fma(v1[w], xr, a1); a1b = fma(v1[w + 1], xs, a1b); }
            }
        }
    }
    if (lane < TS) s_part[wave][lane] = a0 + a0b;
    if (lane + 64 < TS) s_part[wave][lane + 64] = a1 + a1b;
    __syncthreads();
    if (tid < TS) {
        double sum = 0.0;
#pragma unroll
        for (int w = 0; w < NW; ++w) sum += s_part[w][tid];
        s_acc[tid] = x[k * TS + tid] - sum;
    }
    __syncthreads();
    // (2) x_k = Linv_kk' s_acc (Linv is lower triangular): rows split over the waves
    double b0[2] = {0.0, 0.0}, b1[2] = {0.0, 0.0};
#pragma unroll
    for (int rr = 0; rr < RW; ++rr) {
        const int r = wave * RW + rr;
        const double sr = r < TS ? s_acc[r < TS ? r : 0] : 0.0;
        b0[rr & 1] = fma(r >= c0 ? xv0[rr] : 0.0, sr, b0[rr & 1]);   // Linv is lower triangular
        if (TS > 64) b1[rr & 1] = fma(r >= c1 ? xv1[rr] : 0.0, sr, b1[rr & 1]);
    }
    if (lane < TS) s_part[wave][lane] = b0[0] + b0[1];
    if (lane + 64 < TS) s_part[wave][lane + 64] = b1[0] + b1[1];
    __syncthreads();
    if (tid < TS) {
        double sum = 0.0;
#pragma unroll
        for (int w = 0; w < NW; ++w) sum += s_part[w][tid];
        x[k * TS + tid] = sum;
    }
}

// The same step with the column's lists in the KERNEL ARGUMENTS (levels of at most kInlineCols columns with at most kInlineSub
// sub-diagonal tiles each; `si` is the FIRST parameter, so it sits at offset 0 of the kernel-argument segment and is read from
// there with scalar loads - indexing a by-value parameter would copy it to scratch memory).  Nothing here waits for an index
// list: the rows of L_kk^-1, the x_i segments and every L_ik row this wave will use are requested in one go at the start, one
// memory round trip instead of three dependent ones (record -> tile list -> tiles), 5.6 -> us per level at config 4.
template <int TS>
__global__ __launch_bounds__(kPotrfThreads) void k_back_solve_inl(SolveInline si, const double* __restrict__ Lt, const double* __restrict__ Linv, double* x,
                                                                  const int* status)
{
    static_assert(TS == 48, "written for the 48-wide tile (one column set per lane)");
    constexpr int NW = kPotrfThreads / 64, RB = 24, NRB = TS / RB, MAXU = (kInlineSub * NRB + NW - 1) / NW;
    __shared__ double s_part[NW][TS];
    __shared__ double s_x[NW][RB];
    __shared__ double s_acc[TS];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int failed = *status;
    typedef const SolveInline __attribute__((address_space(4))) KernargSolve;
    KernargSolve* ksi = (KernargSolve*)__builtin_amdgcn_kernarg_segment_ptr();
    const int k = ksi->c[blockIdx.x].k, nq = ksi->c[blockIdx.x].nq;
    const int c0 = lane < TS ? lane : TS - 1;
    constexpr int RW = (TS + NW - 1) / NW;
    double xv0[RW];
    {
        const double* X = Linv + (size_t)k * TS * TS;
#pragma unroll
        for (int rr = 0; rr < RW; ++rr) {
            const int r = wave * RW + rr < TS ? wave * RW + rr : TS - 1;
            xv0[rr] = X[r * TS + c0];
        }
    }
    double xin[MAXU], lv[MAXU][RB];
#pragma unroll
    for (int s = 0; s < MAXU; ++s) {
        const int u = wave + NW * s;
        xin[s] = 0.0;
#pragma unroll
        for (int w = 0; w < RB; ++w) lv[s][w] = 0.0;
        if (u < nq * NRB) { // uniform
            const int q = u / NRB, rb = u % NRB;
            const int tile = ksi->c[blockIdx.x].tile[q], row = ksi->c[blockIdx.x].row[q];
            const double* L = Lt + (size_t)tile * TS * TS + (size_t)rb * RB * TS;
            xin[s] = x[(size_t)row * TS + rb * RB + (lane < RB ? lane : 0)];
#pragma unroll
            for (int w = 0; w < RB; ++w) lv[s][w] = L[w * TS + c0];
        }
    }
    const double yk = tid < TS ? x[k * TS + tid] : 0.0;
    if (failed != 0) return;
    double a0 = 0.0, a0b = 0.0;
#pragma unroll
    for (int s = 0; s < MAXU; ++s) {
        if (wave + NW * s >= nq * NRB) break; // uniform
        __builtin_amdgcn_wave_barrier();
        if (lane < RB) s_x[wave][lane] = xin[s];
        asm volatile("" ::: "memory"); // (ordering for the compiler only: LDS is in order inside a wave, and a release fence would wait for every L row still on its way)
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int w = 0; w < RB; w += 2) { // two chains per column: the FMAs are latency-bound
            a0 = fma(lv[s][w], s_x[wave][w], a0);
            a0b = fma(lv[s][w + 1], s_x[wave][w + 1], a0b);
        }
    }
    if (lane < TS) s_part[wave][lane] = a0 + a0b;
    __syncthreads();
    if (tid < TS) {
        double sum = 0.0;
#pragma unroll
        for (int w = 0; w < NW; ++w) sum += s_part[w][tid];
        s_acc[tid] = yk - sum;
    }
    __syncthreads();
    double b0[2] = {0.0, 0.0};
#pragma unroll
    for (int rr = 0; rr < RW; ++rr) {
        const int r = wave * RW + rr;
        const double sr = r < TS ? s_acc[r < TS ? r : 0] : 0.0;
        b0[rr & 1] = fma(r >= c0 ? xv0[rr] : 0.0, sr, b0[rr & 1]); // Linv is lower triangular
    }
    __syncthreads(); // (s_part is reused)
    if (lane < TS) s_part[wave][lane] = b0[0] + b0[1];
    __syncthreads();
    if (tid < TS) {
        double sum = 0.0;
#pragma unroll
        for (int w = 0; w < NW; ++w) sum += s_part[w][tid];
        x[k * TS + tid] = sum;
    }
}

// The WHOLE backward substitution in one launch: one workgroup per column below the last level, in descending level order.
// Forward progress: a workgroup only ever waits for workgroups with a LOWER block index (higher level), and the host uses this
// launch only while columns + the pose workgroup <= the number of compute units (ba_structure.cpp: upload) - a 512-thread
// workgroup of this kernel always finds a CU, so every workgroup of the grid is resident and the order in which the hardware
// dispatches them does not matter (HIP promises none).  Longer trajectories take the per-level launches.  A timeout (status
// kBackSolveTimeout) is therefore a DEFECT, reported to the caller as SVI_ERR_INTERNAL and never folded into the LM rule.  A workgroup requests the
// rows of L_kk^-1, its y_k and every L_ik row it will use right away - none of that depends on another column - and then
// waits, entry by entry, for the x_i of the rows above: the forward phase left "pending" markers in the solution vector, a
// marker that disappears IS the hand-over (one remote read per dependency level: 16 hops instead of 16 launches).  Every wait
// is bounded: a value that never arrives (which would be a defect) ends the kernel with a status instead of hanging the GPU.
// The trial poses, by one extra workgroup of k_back_solve_all: a lane per pose, waiting for that pose's six dx entries the way
// the column workgroups wait for their x segments (bounded; a timeout or a failed factorisation ends the trial as failed).
__device__ void pose_tail_wg(const PoseTail& pt, const double* x, int* status, int kSpinLimit)
{
    constexpr int NW = kPotrfThreads / 64;
    __shared__ double s_red[NW];
    __shared__ int s_bad;
    if (threadIdx.x == 0) s_bad = 0;
    __syncthreads();
    double part = 0.0;
    for (int s = threadIdx.x; s < pt.Pn; s += kPotrfThreads) {
        double T[12], Tn[12];
        {
            const double2* sp = reinterpret_cast<const double2*>(pt.src + 12 * s);
#pragma unroll
            for (int k = 0; k < 6; ++k) { const double2 v = sp[k]; T[2 * k] = v.x; T[2 * k + 1] = v.y; }
        }
        const int r = pt.pose_red[s];
#pragma unroll
        for (int k = 0; k < 12; ++k) Tn[k] = T[k];
        if (r >= 0) {
            double dl[6], bl[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) bl[k] = pt.bp[6 * r + k];
            bool pending = true;
            for (int spins = 0; pending; ++spins) {
                pending = false;
#pragma unroll
                for (int k = 0; k < 6; ++k) { dl[k] = __hip_atomic_load(x + 6 * r + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); pending = pending || is_solve_pending(dl[k]); }
                if (pending && (spins >= kSpinLimit || ((spins & 255) == 255 && __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0))) break;
            }
            if (pending) s_bad = 1; // (the pose stays where it is)
            else {
#pragma unroll
                for (int k = 0; k < 6; ++k) part += dl[k] * (pt.wl * dl[k] + pt.wb * bl[k]);
                pose_oplus(T, dl, Tn);
            }
        }
        double2* dp = reinterpret_cast<double2*>(pt.dst + 12 * s);
#pragma unroll
        for (int k = 0; k < 6; ++k) dp[k] = make_double2(Tn[2 * k], Tn[2 * k + 1]);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = part;
    __syncthreads();
    if (threadIdx.x == 0) {
        double sum = 0.0;
#pragma unroll
        for (int w = 0; w < NW; ++w) sum += s_red[w];
        pt.scal[3] = sum;
        if (pt.lin_from_red) { pt.scal[8] = pt.red_base[0]; pt.scal[9] = pt.red_base[1]; }
        if (s_bad) atomicCAS(status, 0, -3);
    }
}

template <int TS>
__global__ __launch_bounds__(kPotrfThreads) void k_back_solve_all(const SolveRec* __restrict__ recs, const double* __restrict__ Lt,
                                                                  const double* __restrict__ Linv, const double* __restrict__ y, double* x, int* status,
                                                                  PoseTail pt, int kSpinLimit, int defer_y)
{
    static_assert(TS == 48, "written for the 48-wide tile (one column set per lane)");
    if (pt.src != nullptr && blockIdx.x == gridDim.x - 1) { // the extra workgroup: the trial poses
        if (*status == 0) pose_tail_wg(pt, x, status, kSpinLimit);
        return;
    }
    constexpr int NW = kPotrfThreads / 64, RB = 24, NRB = TS / RB, MAXU = (kInlineSub * NRB + NW - 1) / NW;
    __shared__ double s_part[NW][TS];
    __shared__ double s_x[NW][RB];
    __shared__ double s_acc[TS];
    __shared__ int s_bad;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (*status != 0) return; // (final before this launch: nothing was left pending by a failed factorisation that matters)
    const SolveRec* rec = recs + blockIdx.x;
    const int k = rec->k, nq = rec->nq;
    const int c0 = lane < TS ? lane : TS - 1;
    constexpr int RW = (TS + NW - 1) / NW;
    double xv0[RW];
    {
        const double* X = Linv + (size_t)k * TS * TS;
#pragma unroll
        for (int rr = 0; rr < RW; ++rr) {
            const int r = wave * RW + rr < TS ? wave * RW + rr : TS - 1;
            xv0[rr] = X[r * TS + c0];
        }
    }
    if (tid == 0) s_bad = 0;
    double lv[MAXU][RB];
    const double* xsrc[MAXU];
#pragma unroll
    for (int s = 0; s < MAXU; ++s) {
        const int u = wave + NW * s;
        xsrc[s] = nullptr;
#pragma unroll
        for (int w = 0; w < RB; ++w) lv[s][w] = 0.0;
        if (u < nq * NRB) { // uniform
            // (the record lists the rows by level, highest first: what was solved long ago is polled first, the row of the level just
            // above - the last to arrive - last; polled first it kept the finished ones waiting behind it: 32.9 -> 27.9 us)
            const int q = u / NRB, rb = u % NRB;
            const double* L = Lt + (size_t)rec->tile[q] * TS * TS + (size_t)rb * RB * TS;
            xsrc[s] = x + (size_t)rec->row[q] * TS + rb * RB + (lane < RB ? lane : 0);
#pragma unroll
            for (int w = 0; w < RB; ++w) lv[s][w] = L[w * TS + c0];
        }
    }
    // (defer_y: the y buffer holds the final g_k, y_k = L_kk^-1 g_k is formed here - same sums as the factorisation would have taken)
    double yk = 0.0;
    if (tid < TS) yk = defer_y ? y_row<TS, TS>(Linv + (size_t)k * TS * TS, tid, y + k * TS) : y[k * TS + tid];
    double a0 = 0.0, a0b = 0.0;
    bool bad = false;
#pragma unroll
    for (int s = 0; s < MAXU; ++s) {
        if (wave + NW * s >= nq * NRB) break; // uniform
        double xin = 0.0;
        if (lane < RB) {
            // bounded, and a waiter gives up at once when a column further up the chain has (its failure is in the status word:
            // without this look every dependant would burn its whole budget, seconds, behind a hand-over that never comes)
            int spins = 0;
            do {
                xin = __hip_atomic_load(xsrc[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (!is_solve_pending(xin)) break;
                if ((spins & 255) == 255 && __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
            } while (++spins < kSpinLimit);
            bad = bad || is_solve_pending(xin);
        }
        asm volatile("" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        if (lane < RB) s_x[wave][lane] = xin;
        asm volatile("" ::: "memory");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int w = 0; w < RB; w += 2) { // two chains per column: the FMAs are latency-bound
            a0 = fma(lv[s][w], s_x[wave][w], a0);
            a0b = fma(lv[s][w + 1], s_x[wave][w + 1], a0b);
        }
    }
    if (bad) s_bad = 1;
    if (lane < TS) s_part[wave][lane] = a0 + a0b;
    __syncthreads();
    if (s_bad) { if (tid == 0) atomicCAS(status, 0, -3); return; } // (the pending markers of this column stay: the columns below end the same way)
    if (tid < TS) {
        double sum = 0.0;
#pragma unroll
        for (int w = 0; w < NW; ++w) sum += s_part[w][tid];
        s_acc[tid] = yk - sum;
    }
    __syncthreads();
    double b0[2] = {0.0, 0.0};
#pragma unroll
    for (int rr = 0; rr < RW; ++rr) {
        const int r = wave * RW + rr;
        const double sr = r < TS ? s_acc[r < TS ? r : 0] : 0.0;
        b0[rr & 1] = fma(r >= c0 ? xv0[rr] : 0.0, sr, b0[rr & 1]); // Linv is lower triangular
    }
    __syncthreads(); // (s_part is reused)
    if (lane < TS) s_part[wave][lane] = b0[0] + b0[1];
    __syncthreads();
    if (tid < TS) {
        double sum = 0.0;
#pragma unroll
        for (int w = 0; w < NW; ++w) sum += s_part[w][tid];
        __hip_atomic_store(x + k * TS + tid, sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// levels [st_begin, st_end) of the factorisation, then (solve) the backward substitution
template <int TS>
int run(const CholPlan& p, double* S, double* Lt, double* Linv, double* g, double* x, double lambda, int n, int* status, hipStream_t s,
        const PoseTail* tail, int* tail_done, int st_begin, int st_end, bool solve, double* Su = nullptr, double* gu = nullptr)
{
    const bool split = TS == 48 && Su != nullptr;
    if (tail_done) *tail_done = 0;
    // one-launch backward substitution (tile 48, every column's list short enough): the forward vector then lives in p.ybuf and
    // the solution vector carries "pending" markers until its entries are computed
    const bool one_launch = TS == 48 && p.solve_recs != nullptr && p.n_solve_cols > 0 && p.ybuf != nullptr;
    double* const yv = one_launch ? p.ybuf : x;
    static const bool no_defer = getenv("SVI_NO_DEFER_Y") != nullptr; // (A/B)
    const int defer_y = (one_launch && !no_defer) ? 1 : 0;
    constexpr int LD = Lds<TS>::LD, Q = TS / kOB;
    const size_t lds_p = sizeof(double) * (Fold<TS>::on ? 5 : 2) * (size_t)TS * LD; // folded: the grouped updates stage five images
    constexpr int kTB = TrsmBlock<TS>::TB, QT = TS / kTB;
    const size_t lds_g = sizeof(double) * 2 * (size_t)kTB * LD;
    static bool attr = false;
    if (!attr) {
        // a workgroup asking for more LDS than the CU has faults the queue: refuse instead of launching
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_potrf_inv<TS, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_p) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(k_potrf_inv<TS, TS == 48>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_p) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(k_trsm<TS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_g) != hipSuccess)
            return 1;
        attr = true;
    }
    StepArgs sa{};
    sa.diag_tile = p.diag_tile; sa.pre_ptr = p.pre_ptr; sa.pre_tile = p.pre_tile; sa.pre_col = p.pre_col;
    sa.tgt_pair_ptr = p.tgt_pair_ptr; // indexed through the level's offset below
    sa.pair_a = p.pair_a; sa.pair_b = p.pair_b; sa.pair_src = p.pair_src;
    for (int st = st_begin; st < st_end; ++st) {
        const int c0 = p.h_step_ptr[st], nc = p.h_step_ptr[st + 1] - c0;
        const int t0 = p.h_tgt_ptr[st], ntg = p.h_tgt_ptr[st + 1] - t0;
        sa.defer_y = defer_y;
        sa.n_chain = nc; sa.last_level = st == p.n_steps - 1; sa.chain_col = p.step_col + c0; sa.chain_desc = reinterpret_cast<const int4*>(p.step_desc) + 2 * c0;
        sa.tgt_tile = p.tgt_tile + t0; sa.tgt_row = p.tgt_row + t0; sa.tgt_pair_ptr = p.tgt_pair_ptr + t0;
        if (p.h_chain_inl) sa.inl = p.h_chain_inl[st]; else sa.inl.n = 0;
        if (split) hipLaunchKernelGGL((k_potrf_inv<TS, TS == 48>), dim3(nc + ntg * Q * Q), dim3(kPotrfThreads), lds_p, s, S, Lt, Linv, g, yv, n, lambda, status, 0, sa, x, Su, gu);
        else hipLaunchKernelGGL((k_potrf_inv<TS, false>), dim3(nc + ntg * Q * Q), dim3(kPotrfThreads), lds_p, s, S, Lt, Linv, g, yv, n, lambda, status, 0, sa, x, nullptr, nullptr);
        const int i0 = p.h_trsm_ptr[st], ni = p.h_trsm_ptr[st + 1] - i0;
        if (ni > 0 && !Fold<TS>::on) hipLaunchKernelGGL(k_trsm<TS>, dim3(ni * QT * QT), dim3(kBlock), lds_g, s, S, Lt, Linv, p.st_tile + i0, p.st_col + i0, status);
    }
#ifdef POTRF_TS
    {
        static int calls = 0;
        if (++calls == 100 && TS == 48) {
            unsigned long long h[64 * 16];
            int cnt = 0;
            if (hipStreamSynchronize(s) == hipSuccess && hipMemcpyFromSymbol(h, HIP_SYMBOL(g_ts), sizeof(h)) == hipSuccess &&
                hipMemcpyFromSymbol(&cnt, HIP_SYMBOL(g_ts_cnt), sizeof(int)) == hipSuccess)
                for (int st = 0; st < p.n_steps; ++st) {
                    const int slot = (cnt - p.n_steps + st) & 63;
                    fprintf(stderr, "TS level %2d:", st);
                    for (int i = 1; i <= 15; ++i) fprintf(stderr, " %6lld", (long long)(h[slot * 16 + i] - h[slot * 16]));
                    fprintf(stderr, "\n");
                }
        }
    }
#endif
    if (!solve) return 0;
    if constexpr (TS == 48) {
        if (one_launch) {
            PoseTail pt{};
            if (tail) { pt = *tail; if (tail_done) *tail_done = 1; }
            hipLaunchKernelGGL(k_back_solve_all<TS>, dim3(p.n_solve_cols + (tail ? 1 : 0)), dim3(kPotrfThreads), 0, s, p.solve_recs, Lt, Linv, yv, x, status, pt,
                               g_backsolve_spin_limit.load(std::memory_order_relaxed), defer_y);
            return 0;
        }
    }
    for (int st = p.n_steps - 2; st >= 0; --st) { // (the last level solved its x inside k_potrf_inv)
        const int c0 = p.h_step_ptr[st], nc = p.h_step_ptr[st + 1] - c0;
        if constexpr (TS == 48) {
            if (p.h_solve_inl && p.h_solve_inl[st].n == nc) {
                hipLaunchKernelGGL(k_back_solve_inl<TS>, dim3(nc), dim3(kPotrfThreads), 0, s, p.h_solve_inl[st], Lt, Linv, x, status);
                continue;
            }
        }
        hipLaunchKernelGGL(k_back_solve<TS>, dim3(nc), dim3(kPotrfThreads), 0, s, Lt, Linv, x, p, reinterpret_cast<const int4*>(p.step_desc) + 2 * c0, status);
    }
    return 0;
}

} // namespace

// timing probe for the diagonal-tile kernel: reps launches on one SPD tile, truncated after phase
// `stop_after` (0 = full kernel); returns the mean kernel time in ms
template <int TS>
static int potrf_probe(int reps, int stop_after, double* ms_out)
{
    constexpr int LD = Lds<TS>::LD;
    const size_t lds_p = sizeof(double) * (Fold<TS>::on ? 5 : 2) * (size_t)TS * LD;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_potrf_inv<TS, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_p) != hipSuccess) return 1;
    double *S = nullptr, *L = nullptr, *X = nullptr, *g = nullptr, *y = nullptr;
    int *st = nullptr, *tab = nullptr; // tab: an all-zero column record (column 0, tile 0, nothing pending)
    std::vector<double> h((size_t)TS * TS);
    for (int r = 0; r < TS; ++r)
        for (int c = 0; c < TS; ++c) h[(size_t)r * TS + c] = (r == c ? TS + 1.0 : 0.0) + 1.0 / (1.0 + r + c);
    if (hipMalloc(&S, sizeof(double) * TS * TS) != hipSuccess) return 1;
    (void)hipMalloc(&L, sizeof(double) * TS * TS); (void)hipMalloc(&X, sizeof(double) * TS * TS);
    (void)hipMalloc(&g, sizeof(double) * TS); (void)hipMalloc(&y, sizeof(double) * TS); (void)hipMalloc(&st, sizeof(int));
    (void)hipMalloc(&tab, 8 * sizeof(int)); (void)hipMemset(tab, 0, 8 * sizeof(int));
    StepArgs sa{};
    sa.n_chain = 1; sa.chain_col = tab; sa.diag_tile = tab; sa.pre_ptr = tab; sa.chain_desc = reinterpret_cast<const int4*>(tab);
    (void)hipMemcpy(S, h.data(), sizeof(double) * TS * TS, hipMemcpyHostToDevice);
    (void)hipMemset(g, 0, sizeof(double) * TS); (void)hipMemset(st, 0, sizeof(int));
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((k_potrf_inv<TS, false>), dim3(1), dim3(kPotrfThreads), lds_p, 0, S, L, X, g, y, TS, 0.0, st, stop_after, sa, y, nullptr, nullptr);
    (void)hipEventRecord(a, 0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k_potrf_inv<TS, false>), dim3(1), dim3(kPotrfThreads), lds_p, 0, S, L, X, g, y, TS, 0.0, st, stop_after, sa, y, nullptr, nullptr);
    (void)hipEventRecord(b, 0);
    (void)hipEventSynchronize(b);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, a, b);
    *ms_out = ms / reps;
    if (stop_after >= 6 && stop_after <= 9) {
        double hy[2] = {0, 0};
        (void)hipMemcpy(hy, y, sizeof(hy), hipMemcpyDeviceToHost);
        *ms_out = hy[0] * 1e-6 + hy[1] * 1e3; // packed: cycles*1e-6 + ticks*1e3 (decoded by tools/probe_chol.py)
        ms_out[0] = hy[0]; // cycles
        // ticks go to the second slot if the caller provided room
        ms_out[1] = hy[1];
    }
    (void)hipFree(S); (void)hipFree(L); (void)hipFree(X); (void)hipFree(g); (void)hipFree(y); (void)hipFree(st); (void)hipFree(tab);
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    return 0;
}
int chol_potrf_probe(int tile, int reps, int stop_after, double* ms)
{
    return tile == 48 ? potrf_probe<48>(reps, stop_after, ms) : potrf_probe<96>(reps, stop_after, ms);
}

// S (tiles) is consumed, L goes to Lt; g is consumed; x receives the solution
// tail (optional): the pose update of the trial rides in the one-launch backward substitution; *tail_done says whether it did
int chol_factor_solve(const CholPlan& p, double* S, double* Lt, double* Linv, double* g, double* x, double lambda, int n, int* status,
                      void* st, const PoseTail* tail, int* tail_done)
{
    hipStream_t s = static_cast<hipStream_t>(st);
    return p.TS == 48 ? run<48>(p, S, Lt, Linv, g, x, lambda, n, status, s, tail, tail_done, 0, p.n_steps, true)
                      : run<96>(p, S, Lt, Linv, g, x, lambda, n, status, s, tail, tail_done, 0, p.n_steps, true);
}

// the same in pieces: levels [st_begin, st_end) only; the backward substitution (and the trial poses) behind the last piece
// Su / gu (tile 48, optional): the updates go to these arrays instead of into S / g (see potrf_sweep_mfma)
int chol_factor_range(const CholPlan& p, double* S, double* Lt, double* Linv, double* g, double* x, double lambda, int n, int* status,
                      void* st, const PoseTail* tail, int* tail_done, int st_begin, int st_end, int solve, double* Su, double* gu)
{
    hipStream_t s = static_cast<hipStream_t>(st);
    return p.TS == 48 ? run<48>(p, S, Lt, Linv, g, x, lambda, n, status, s, tail, tail_done, st_begin, st_end, solve != 0, Su, gu)
                      : run<96>(p, S, Lt, Linv, g, x, lambda, n, status, s, tail, tail_done, st_begin, st_end, solve != 0);
}

} // namespace svi
