// matcher.hip — BRIEF-256 Hamming matcher for gfx950 (MI355X).
//
// Replaces, behind the C ABI of include/svi_hot.h, the arithmetic the reference delegates to
//   cv::BFMatcher(cv::NORM_HAMMING)::match   (src/core/CTriangulator.cpp:12,93,156,227,298;
//                                             src/core/CFundamentalMatcher.cpp:540,657,1080,1200,1584,1708,2356)
//   cv::norm(a, b, cv::NORM_HAMMING)         (src/core/CFundamentalMatcher.cpp:404,423,453,473,573,691,2375)
//   CTriangulator::getPointInLEFT            (src/core/CTriangulator.cpp:326-356)
// in the batched form of SURVEY.md Appendix A: NQ queries x NT pool entries per frame pair, an
// epipolar gate predicate per pair, lexicographic (distance, index) minimum, strict cut-off.
//
// Kernel shape (K1): one query per lane (its 256 bits live in 8 VGPRs for the whole kernel), the
// pool is streamed through LDS in tiles of WAVES*64 descriptors with coalesced 16 B/lane global
// loads; wave w scans entries [64w, 64w+64) of each tile with broadcast ds_read_b128 (all lanes
// read the same entry: conflict free), 8 x (v_xor_b32 + v_bcnt_u32_b32 accumulate) per pair.
// The gate is evaluated first; a pool entry that no lane of the wave accepts is skipped by the
// whole wave (scalar branch on the ballot), which is what makes the row-gated case cheap.
// The per-wave minima are merged through LDS as packed (distance<<32 | index) keys, so "lowest
// index wins ties" holds regardless of which wave or block saw the entry.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>

#include "common.h"
#include "matcher_handle.h"

namespace {

constexpr int      kLanes   = 64;
constexpr uint32_t kNoDist  = 257u;
constexpr uint64_t kNoKey   = 0xFFFFFFFFFFFFFFFFull;

struct MatchArgs {
    const uint4*  q;      // [batch][nq][2]
    const uint4*  t;      // [batch][nt][2]
    const float2* q_uv;   // [batch][nq]
    const float2* t_uv;   // [batch][nt]
    const float*  q_umin; // [batch][nq]
    const float*  q_umax; // [batch][nq]
    float         v_tol;
    int           nq, nt;
    const int32_t* t_seg; // cloud mode (null otherwise): pool z is rows [t_seg[z], t_seg[z+1]) of t, the queries are shared
    int           tiles_per_split; // pool tiles scanned by one block
    int           cutoff;          // keep iff cutoff > distance
    unsigned long long* keys;      // [nsplit][batch][nq] when the pool is split over blocks (every split stores its own minimum), else null
    int           nsplit;
    int32_t*      out_idx;
    int32_t*      out_dist;
    // fused triangulation (null out_xyz: off)
    double        finv, cx, cy, dur, min_disp;
    double*       out_xyz;
    uint8_t*      out_ok;
};

__device__ __forceinline__ uint32_t hamming256(const uint4& a0, const uint4& a1, const uint4& b0, const uint4& b1)
{
    uint32_t d = __popc(a0.x ^ b0.x);
    d += __popc(a0.y ^ b0.y);
    d += __popc(a0.z ^ b0.z);
    d += __popc(a0.w ^ b0.w);
    d += __popc(a1.x ^ b1.x);
    d += __popc(a1.y ^ b1.y);
    d += __popc(a1.z ^ b1.z);
    d += __popc(a1.w ^ b1.w);
    return d;
}

// CTriangulator::getPointInLEFT, operand order of CTriangulator.cpp:340-347
__device__ __forceinline__ bool triangulate(float uL, float vL, float uR, double finv, double cx, double cy,
                                            double dur, double min_disp, double* xyz)
{
    const float disparity = uL - uR;
    if (static_cast<double>(disparity) < min_disp) {
        xyz[0] = 0.0; xyz[1] = 0.0; xyz[2] = 0.0;
        return false;
    }
    const double z = dur / static_cast<double>(disparity);
    const double fz = finv * z;
    xyz[0] = fz * (static_cast<double>(uL) - cx);
    xyz[1] = fz * (static_cast<double>(vL) - cy);
    xyz[2] = z;
    return true;
}

__device__ __forceinline__ void write_result(const MatchArgs& a, size_t qglob, size_t tbase, unsigned long long key,
                                             float qu, float qv)
{
    const uint32_t d = static_cast<uint32_t>(key >> 32);
    const bool hit = key != kNoKey && static_cast<uint32_t>(a.cutoff) > d;
    const int32_t idx = hit ? static_cast<int32_t>(key & 0xFFFFFFFFu) : -1;
    a.out_idx[qglob]  = idx;
    a.out_dist[qglob] = hit ? static_cast<int32_t>(d) : static_cast<int32_t>(kNoDist);
    if (a.out_xyz) {
        double xyz[3] = {0.0, 0.0, 0.0};
        bool ok = false;
        if (hit) {
            const float2 tuv = a.t_uv[tbase + idx];
            ok = triangulate(qu, qv, tuv.x, a.finv, a.cx, a.cy, a.dur, a.min_disp, xyz);
        }
        a.out_xyz[3 * qglob + 0] = xyz[0];
        a.out_xyz[3 * qglob + 1] = xyz[1];
        a.out_xyz[3 * qglob + 2] = xyz[2];
        a.out_ok[qglob] = ok ? 1 : 0;
    }
}

// grid: x = query group of 64, y = pool split, z = frame pair
template <int WAVES, bool GATED>
__global__ __launch_bounds__(WAVES * 64) void k_match_hamming256(MatchArgs a)
{
    constexpr int TILE = WAVES * kLanes;
    __shared__ uint4  s_lo[TILE];     // first 16 bytes of each staged pool descriptor
    __shared__ uint4  s_hi[TILE];     // second 16 bytes
    __shared__ float2 s_uv[TILE];
    __shared__ unsigned long long s_key[WAVES][kLanes];

    const int tid  = threadIdx.x;
    const int lane = tid & (kLanes - 1);
    const int wave = tid >> 6;
    const size_t obase = static_cast<size_t>(blockIdx.z) * a.nq;           // rows of the outputs
    const size_t qbase = a.t_seg ? 0 : obase;                               // rows of the queries
    const size_t tbase = a.t_seg ? static_cast<size_t>(a.t_seg[blockIdx.z]) : static_cast<size_t>(blockIdx.z) * a.nt;
    const int    nt    = a.t_seg ? a.t_seg[blockIdx.z + 1] - a.t_seg[blockIdx.z] : a.nt;
    const int  qi     = blockIdx.x * kLanes + lane;
    const bool qvalid = qi < a.nq;
    const size_t qglob = qbase + (qvalid ? qi : 0);
    const size_t oglob = obase + (qvalid ? qi : 0);

    uint4 q0 = make_uint4(0, 0, 0, 0), q1 = q0;
    float qu = 0.f, qv = 0.f, umin = 0.f, umax = 0.f;
    if (qvalid) {
        q0 = a.q[2 * qglob];
        q1 = a.q[2 * qglob + 1];
        if (GATED || a.out_xyz) {
            const float2 uv = a.q_uv[qglob];
            qu = uv.x; qv = uv.y;
        }
        if (GATED) { umin = a.q_umin[qglob]; umax = a.q_umax[qglob]; }
    }

    uint32_t best_d = 0xFFFFFFFFu, best_j = 0xFFFFFFFFu;
    const int tile0 = blockIdx.y * a.tiles_per_split;
    const int ntile = (nt + TILE - 1) / TILE;
    const int tile1 = min(tile0 + a.tiles_per_split, ntile);

    // register staging of the next tile (issue early, write to LDS late)
    uint4 r_lo = make_uint4(0, 0, 0, 0), r_hi = r_lo;
    float2 r_uv = make_float2(0.f, 0.f);
    auto fetch = [&](int tile) {
        const int j = tile * TILE + tid;
        if (j < nt) {
            r_lo = a.t[2 * (tbase + j)];
            r_hi = a.t[2 * (tbase + j) + 1];
            if (GATED) r_uv = a.t_uv[tbase + j];
        }
    };
    if (tile0 < tile1) fetch(tile0);

    for (int tile = tile0; tile < tile1; ++tile) {
        __syncthreads(); // previous tile fully scanned
        s_lo[tid] = r_lo;
        s_hi[tid] = r_hi;
        if (GATED) s_uv[tid] = r_uv;
        __syncthreads();
        if (tile + 1 < tile1) fetch(tile + 1);

        const int jbase = tile * TILE + wave * kLanes;
        const int jn    = min(kLanes, nt - jbase); // may be <= 0 for the ragged last tile
#pragma unroll 4
        for (int jj = 0; jj < jn; ++jj) {
            const int sj = wave * kLanes + jj;
            bool pass = qvalid;
            if (GATED) {
                const float2 tuv = s_uv[sj];
                pass = pass && (fabsf(tuv.y - qv) <= a.v_tol) && (umin <= tuv.x) && (tuv.x < umax);
                if (__ballot(pass) == 0ull) continue; // nobody in this wave wants entry sj
            }
            const uint32_t d = hamming256(q0, q1, s_lo[sj], s_hi[sj]);
            if (pass && d < best_d) { // ascending j, strict '<': lowest index wins inside the slice
                best_d = d;
                best_j = static_cast<uint32_t>(jbase + jj);
            }
        }
    }

    // merge the WAVES slices: lexicographic (distance, index) minimum == min of the packed key
    const unsigned long long key = (best_j == 0xFFFFFFFFu)
        ? kNoKey : ((static_cast<unsigned long long>(best_d) << 32) | best_j);
    s_key[wave][lane] = key;
    __syncthreads();
    if (wave == 0 && qvalid) {
        unsigned long long k = s_key[0][lane];
#pragma unroll
        for (int w = 1; w < WAVES; ++w) k = min(k, s_key[w][lane]);
        if (a.keys) { // this split's minimum; the second pass takes the minimum over the splits (no initialisation, no atomics)
            a.keys[static_cast<size_t>(blockIdx.y) * gridDim.z * a.nq + oglob] = k;
        } else {
            write_result(a, oglob, tbase, k, qu, qv);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Row-bucketed variant for the gated single-frame call (small pool, few candidates per query): the gate keeps
// |t.v - q.v| <= v_tol, so instead of evaluating the predicate for all NQ x NT pairs every workgroup first sorts the
// pool by image row in LDS (counting sort on floor(v); rows outside [0, kRows) share an overflow bucket), stages the
// descriptors, and each query lane then visits only the buckets its row window touches.  Same results as the full
// scan: the exact predicate is re-checked per candidate and the minimum is taken on the packed (distance, index) key.
// ---------------------------------------------------------------------------------------------
constexpr int kRows = 512, kBucketPool = 3072, kBucketThreads = 512; // 3072 x 44 B = 132 KiB of the 160 KiB LDS

__global__ __launch_bounds__(kBucketThreads) void k_match_rowbucket(MatchArgs a)
{
    extern __shared__ __align__(16) unsigned char smem[];
    uint4*  s_desc = reinterpret_cast<uint4*>(smem);                         // [nt][2]
    float2* s_uv   = reinterpret_cast<float2*>(s_desc + 2 * (size_t)a.nt);  // [nt]
    int*    s_list = reinterpret_cast<int*>(s_uv + a.nt);                   // [nt] pool indices grouped by bucket
    __shared__ int s_cnt[kRows + 2], s_start[kRows + 2];
    const int tid = threadIdx.x;
    const size_t tbase = static_cast<size_t>(blockIdx.z) * a.nt, qbase = static_cast<size_t>(blockIdx.z) * a.nq;
    for (int b = tid; b < kRows + 2; b += kBucketThreads) s_cnt[b] = 0;
    __syncthreads();
    auto bucket_of = [](float v) { return (v >= 0.0f && v < static_cast<float>(kRows)) ? static_cast<int>(v) : kRows; };
    for (int j = tid; j < a.nt; j += kBucketThreads) {
        const float2 uv = a.t_uv[tbase + j];
        s_uv[j] = uv;
        s_desc[2 * j] = a.t[2 * (tbase + j)];
        s_desc[2 * j + 1] = a.t[2 * (tbase + j) + 1];
        atomicAdd(&s_cnt[bucket_of(uv.y)], 1);
    }
    __syncthreads();
    if (tid < 64) { // exclusive scan of the 513 counts by one wavefront
        int carry = 0;
        for (int base = 0; base < kRows + 1; base += 64) {
            const int b = base + tid;
            const int v = b < kRows + 1 ? s_cnt[b] : 0;
            int inc = v;
            for (int off = 1; off < 64; off <<= 1) { const int o = __shfl_up(inc, off); if (tid >= off) inc += o; }
            if (b < kRows + 1) s_start[b] = carry + inc - v;
            carry += __shfl(inc, 63);
        }
        if (tid == 0) s_start[kRows + 1] = carry;
    }
    __syncthreads();
    for (int b = tid; b < kRows + 1; b += kBucketThreads) s_cnt[b] = s_start[b]; // running insert positions
    __syncthreads();
    for (int j = tid; j < a.nt; j += kBucketThreads) s_list[atomicAdd(&s_cnt[bucket_of(s_uv[j].y)], 1)] = j;
    __syncthreads();

    for (int qi = blockIdx.x * kBucketThreads + tid; qi < a.nq; qi += gridDim.x * kBucketThreads) {
        const size_t qglob = qbase + qi;
        const uint4 q0 = a.q[2 * qglob], q1 = a.q[2 * qglob + 1];
        const float2 quv = a.q_uv[qglob];
        const float umin = a.q_umin[qglob], umax = a.q_umax[qglob];
        unsigned long long key = kNoKey;
        auto scan = [&](int b) {
            for (int p = s_start[b]; p < s_start[b + 1]; ++p) {
                const int j = s_list[p];
                const float2 tuv = s_uv[j];
                if ((fabsf(tuv.y - quv.y) <= a.v_tol) && (umin <= tuv.x) && (tuv.x < umax)) {
                    const unsigned long long k = (static_cast<unsigned long long>(hamming256(q0, q1, s_desc[2 * j], s_desc[2 * j + 1])) << 32) |
                                                 static_cast<uint32_t>(j);
                    key = k < key ? k : key;
                }
            }
        };
        // buckets floor(qv - tol) .. floor(qv + tol), clamped; a non-finite window falls back to every bucket
        const float lo = quv.y - a.v_tol, hi = quv.y + a.v_tol;
        int b0 = 0, b1 = kRows - 1;
        if (lo >= 0.0f) b0 = lo < static_cast<float>(kRows) ? static_cast<int>(lo) : kRows;
        if (hi < static_cast<float>(kRows)) b1 = hi >= 0.0f ? static_cast<int>(hi) : -1;
        for (int b = b0; b <= b1; ++b) scan(b);
        scan(kRows); // rows outside the table
        write_result(a, qglob, tbase, key, quv.x, quv.y);
    }
}

// second pass when the pool was split across blocks
__global__ __launch_bounds__(256) void k_match_finalize(MatchArgs a, int batch)
{
    const size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    const size_t n = static_cast<size_t>(batch) * a.nq;
    if (i >= n) return;
    const size_t b = i / a.nq;
    float qu = 0.f, qv = 0.f;
    if (a.out_xyz) { const float2 uv = a.q_uv[i]; qu = uv.x; qv = uv.y; }
    unsigned long long k = a.keys[i];
    for (int sp = 1; sp < a.nsplit; ++sp) k = min(k, a.keys[static_cast<size_t>(sp) * n + i]);
    write_result(a, i, a.t_seg ? static_cast<size_t>(a.t_seg[b]) : b * a.nt, k, qu, qv);
}

__global__ __launch_bounds__(256) void k_hamming256_pairs(const uint4* a, const uint4* b, int n, int32_t* dist)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    dist[i] = static_cast<int32_t>(hamming256(a[2 * i], a[2 * i + 1], b[2 * i], b[2 * i + 1]));
}

__global__ __launch_bounds__(256) void k_triangulate(const float2* uvL, const float2* uvR, int n, double finv, double cx,
                                                     double cy, double dur, double min_disp, double* xyz, uint8_t* ok)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double p[3];
    const bool good = triangulate(uvL[i].x, uvL[i].y, uvR[i].x, finv, cx, cy, dur, min_disp, p);
    xyz[3 * i] = p[0]; xyz[3 * i + 1] = p[1]; xyz[3 * i + 2] = p[2];
    ok[i] = good ? 1 : 0;
}

} // namespace


extern "C" {

int svi_matcher_create(int device, void* stream, svi_matcher** out)
{
    if (!out) return svi::fail(SVI_ERR_INVALID, "svi_matcher_create: out is null");
    *out = nullptr;
    if (int rc = svi::use_device(device)) return rc;
    auto* m = new svi_matcher();
    m->device = device;
    if (stream) { m->stream = static_cast<hipStream_t>(stream); }
    else {
        hipError_t e = hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking);
        if (e != hipSuccess) { delete m; return svi::fail(SVI_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(e)); }
        m->own_stream = true;
    }
    if (const int cu = svi::device_compute_units(device)) m->n_cu = cu;
    *out = m;
    return SVI_OK;
}

int svi_matcher_destroy(svi_matcher* m)
{
    if (!m) return SVI_OK;
    (void)hipSetDevice(m->device);
    (void)hipStreamSynchronize(m->stream);
    m->keys.release();
    m->scratch.release();
    m->track.release();
    if (m->track_ev) (void)hipEventDestroy(m->track_ev);
    if (m->own_stream) (void)hipStreamDestroy(m->stream);
    delete m;
    return SVI_OK;
}

int svi_matcher_sync(svi_matcher* m)
{
    if (!m) return svi::fail(SVI_ERR_INVALID, "null matcher");
    SVI_HIP(hipStreamSynchronize(m->stream));
    return SVI_OK;
}

void* svi_matcher_stream(svi_matcher* m) { return m ? static_cast<void*>(m->stream) : nullptr; }

int svi_matcher_set_gate_path(svi_matcher* m, int path)
{
    if (!m || path < 0 || path > 1) return svi::fail(SVI_ERR_INVALID, "svi_matcher_set_gate_path: path must be 0 or 1");
    m->gate_path = path;
    return SVI_OK;
}

static int launch_match(svi_matcher* m, const uint8_t* q, int nq, const uint8_t* t, int nt, int batch,
                        const svi_gate* gate, int cutoff, int32_t* out_idx, int32_t* out_dist,
                        bool fuse, double f, double cx, double cy, double dur, double min_disp, double* out_xyz,
                        uint8_t* out_ok, const int32_t* t_seg = nullptr)
{
    if (!m) return svi::fail(SVI_ERR_INVALID, "null matcher");
    if (nq < 0 || nt < 0 || batch < 0) return svi::fail(SVI_ERR_INVALID, "negative size");
    if (nq == 0 || batch == 0) return SVI_OK;
    if (!q || !out_idx || !out_dist || (nt > 0 && !t)) return svi::fail(SVI_ERR_INVALID, "null descriptor/output pointer");
    if (gate && (!gate->q_uv || !gate->t_uv || !gate->q_umin || !gate->q_umax))
        return svi::fail(SVI_ERR_INVALID, "svi_gate with null member");
    if (fuse && (!gate || !out_xyz || !out_ok)) return svi::fail(SVI_ERR_INVALID, "fused triangulation needs gate coordinates and outputs");
    if ((reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(t)) & 15)
        return svi::fail(SVI_ERR_INVALID, "descriptor arrays must be 16-byte aligned");
    SVI_HIP(svi::enter_device(m->device));

    MatchArgs a{};
    a.q = reinterpret_cast<const uint4*>(q);
    a.t = reinterpret_cast<const uint4*>(t);
    if (gate) {
        a.q_uv = reinterpret_cast<const float2*>(gate->q_uv);
        a.t_uv = reinterpret_cast<const float2*>(gate->t_uv);
        a.q_umin = gate->q_umin; a.q_umax = gate->q_umax; a.v_tol = gate->v_tol;
    }
    a.nq = nq; a.nt = nt; a.cutoff = cutoff; a.t_seg = t_seg;
    a.out_idx = out_idx; a.out_dist = out_dist;
    if (fuse) { a.finv = 1.0 / f; a.cx = cx; a.cy = cy; a.dur = dur; a.min_disp = min_disp; a.out_xyz = out_xyz; a.out_ok = out_ok; }

    // Gated call on a small pool: sort the pool by row in LDS and visit only the rows a query can match.
    if (gate && !t_seg && nt > 0 && nt <= kBucketPool && gate->v_tol >= 0.0f && gate->v_tol <= 8.0f && m->gate_path != 1) {
        const size_t lds = (size_t)nt * (32 + 8 + 4);
        static bool attr = false;
        if (!attr) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_match_rowbucket), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    kBucketPool * (32 + 8 + 4)) != hipSuccess)
                return svi::fail(SVI_ERR_HIP, "row-bucket matcher: LDS request refused");
            attr = true;
        }
        a.keys = nullptr;
        const int gx = std::max(1, std::min((nq + kBucketThreads - 1) / kBucketThreads, 8));
        hipLaunchKernelGGL(k_match_rowbucket, dim3(gx, 1, batch), dim3(kBucketThreads), lds, m->stream, a);
        SVI_HIP(hipGetLastError());
        return SVI_OK;
    }

    // Decomposition. Blocks = query groups x pool splits x frame pairs. Few query groups (a single
    // frame pair): use 16-wave blocks so a CU still holds 4 waves per SIMD, and split the pool over
    // blocks until the chip is covered; the splits meet in a 64-bit atomicMin on the packed key.
    const int qgroups = (nq + kLanes - 1) / kLanes;
    const long long base_blocks = static_cast<long long>(qgroups) * batch;
    // Splitting costs a memset and a finalize launch: only worth it for a sizeable scan (an ungated one, or a gated one of
    // 2^22 pairs and more).  A scan that IS split runs in 4-wave blocks: the kernel is bound by vector instructions, so one wave
    // per SIMD on every CU beats four waves per SIMD on a quarter of them (2048 x 2048 ungated: 17.4 -> 10.3 us per call); a
    // small gated scan stays in one 16-wave block per query group, the pool spread over its waves.
    const long long pairs = static_cast<long long>(nq) * nt;
    const bool may_split = base_blocks < m->n_cu && (!gate || pairs >= (1LL << 22));
    const int waves = (base_blocks >= 4LL * m->n_cu || may_split) ? 4 : 16;
    const int tile  = waves * kLanes;
    const int ntile = (nt + tile - 1) / tile;
    int nsplit = 1;
    if (ntile > 1 && may_split) nsplit = static_cast<int>(std::min<long long>(ntile, (m->n_cu + base_blocks - 1) / base_blocks));
    a.tiles_per_split = nsplit > 0 ? (std::max(ntile, 1) + nsplit - 1) / nsplit : 1;
    nsplit = std::max(1, (std::max(ntile, 1) + a.tiles_per_split - 1) / a.tiles_per_split);
    if (nsplit > 1) {
        const size_t kb = sizeof(unsigned long long) * static_cast<size_t>(nsplit) * batch * nq;
        if (int rc = m->keys.reserve(kb)) return rc;
        a.keys = m->keys.as<unsigned long long>();
        a.nsplit = nsplit;
    }
    const dim3 grid(qgroups, nsplit, batch);
    if (waves == 4) {
        if (gate) hipLaunchKernelGGL((k_match_hamming256<4, true>), grid, dim3(256), 0, m->stream, a);
        else      hipLaunchKernelGGL((k_match_hamming256<4, false>), grid, dim3(256), 0, m->stream, a);
    } else {
        if (gate) hipLaunchKernelGGL((k_match_hamming256<16, true>), grid, dim3(1024), 0, m->stream, a);
        else      hipLaunchKernelGGL((k_match_hamming256<16, false>), grid, dim3(1024), 0, m->stream, a);
    }
    SVI_HIP(hipGetLastError());
    if (nsplit > 1) {
        const size_t n = static_cast<size_t>(batch) * nq;
        hipLaunchKernelGGL(k_match_finalize, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, m->stream, a, batch);
        SVI_HIP(hipGetLastError());
    }
    return SVI_OK;
}

// Shader clock while every CU runs integer VALU work (the matcher's roofline denominator: 64 lane-ops per CU and cycle):
// block 0 brackets a fixed dependent xor / popcount chain with the shader cycle counter and the 100 MHz constant counter.
__global__ __launch_bounds__(256) void k_clock_probe(int iters, unsigned* sink, double* out)
{
    unsigned x = threadIdx.x * 2654435761u + blockIdx.x, acc = 0;
    const long long c0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) { x ^= x << 13; x ^= x >> 17; x ^= x << 5; acc += __popc(x); }
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    if (acc == 0x7fffffffu) sink[0] = acc; // keeps the chain alive
    if (blockIdx.x == 0 && threadIdx.x == 0) { out[0] = (double)(c1 - c0); out[1] = (double)(w1 - w0); }
}

int svi_debug_shader_clock_mhz(svi_matcher* m, double* mhz)
{
    if (!m || !mhz) return svi::fail(SVI_ERR_INVALID, "null argument");
    SVI_HIP(svi::enter_device(m->device));
    if (int rc = m->scratch.reserve(64)) return rc;
    double* out = m->scratch.as<double>();
    unsigned* sink = reinterpret_cast<unsigned*>(out + 4);
    hipLaunchKernelGGL(k_clock_probe, dim3(m->n_cu * 8), dim3(256), 0, m->stream, 20000, sink, out);
    hipLaunchKernelGGL(k_clock_probe, dim3(m->n_cu * 8), dim3(256), 0, m->stream, 20000, sink, out); // the second one runs on ramped clocks
    SVI_HIP(hipGetLastError());
    double h[2] = {0, 0};
    SVI_HIP(hipMemcpyAsync(h, out, sizeof(h), hipMemcpyDeviceToHost, m->stream));
    SVI_HIP(hipStreamSynchronize(m->stream));
    *mhz = h[1] > 0 ? h[0] / (h[1] / 100.0) : 0.0; // wall_clock64 ticks at 100 MHz
    return SVI_OK;
}

int svi_match_hamming256_dev(svi_matcher* m, const uint8_t* q, int nq, const uint8_t* t, int nt, int batch,
                             const svi_gate* gate, int max_dist_exclusive, int32_t* out_idx, int32_t* out_dist)
{
    return launch_match(m, q, nq, t, nt, batch, gate, max_dist_exclusive, out_idx, out_dist, false, 0, 0, 0, 0, 0, nullptr, nullptr);
}

int svi_match_clouds_dev(svi_matcher* m, const uint8_t* q, int nq, const uint8_t* pools, const int32_t* pool_seg, int n_clouds,
                         int max_pool, int max_dist_exclusive, int32_t* out_idx, int32_t* out_dist)
{
    if (n_clouds > 0 && !pool_seg) return svi::fail(SVI_ERR_INVALID, "svi_match_clouds_dev: null pool_seg");
    if (max_pool < 0) return svi::fail(SVI_ERR_INVALID, "svi_match_clouds_dev: max_pool < 0");
    if (n_clouds > 65535) return svi::fail(SVI_ERR_INVALID, "svi_match_clouds_dev: more than 65535 clouds per call");
    return launch_match(m, q, nq, pools, max_pool, n_clouds, nullptr, max_dist_exclusive, out_idx, out_dist, false, 0, 0, 0, 0, 0, nullptr,
                        nullptr, pool_seg);
}

int svi_match_triangulate_dev(svi_matcher* m, const uint8_t* q, int nq, const uint8_t* t, int nt, int batch,
                              const svi_gate* gate, int max_dist_exclusive, double f, double cx, double cy,
                              double duR_flipped, double min_disparity, int32_t* out_idx, int32_t* out_dist,
                              double* out_xyz, uint8_t* ok)
{
    return launch_match(m, q, nq, t, nt, batch, gate, max_dist_exclusive, out_idx, out_dist, true, f, cx, cy,
                        duR_flipped, min_disparity, out_xyz, ok);
}

// Host-pointer convenience form: stage through device scratch, run, copy back, synchronise.
int svi_match_hamming256(svi_matcher* m, const uint8_t* q, int nq, const uint8_t* t, int nt, const svi_gate* gate,
                         int max_dist_exclusive, int32_t* out_idx, int32_t* out_dist)
{
    if (!m) return svi::fail(SVI_ERR_INVALID, "null matcher");
    if (nq < 0 || nt < 0) return svi::fail(SVI_ERR_INVALID, "negative size");
    if (nq == 0) return SVI_OK;
    if (!q || !out_idx || !out_dist || (nt > 0 && !t)) return svi::fail(SVI_ERR_INVALID, "null pointer");
    SVI_HIP(svi::enter_device(m->device));
    auto al = [](size_t x) { return (x + 255) & ~size_t(255); };
    const size_t oq = 0, ot = oq + al(32ull * nq), oquv = ot + al(32ull * nt), otuv = oquv + al(8ull * nq),
                 omin = otuv + al(8ull * nt), omax = omin + al(4ull * nq), oidx = omax + al(4ull * nq),
                 odist = oidx + al(4ull * nq), total = odist + al(4ull * nq);
    if (int rc = m->scratch.reserve(total)) return rc;
    char* d = m->scratch.as<char>();
    SVI_HIP(hipMemcpyAsync(d + oq, q, 32ull * nq, hipMemcpyHostToDevice, m->stream));
    if (nt) SVI_HIP(hipMemcpyAsync(d + ot, t, 32ull * nt, hipMemcpyHostToDevice, m->stream));
    svi_gate g{};
    if (gate) {
        if (!gate->q_uv || !gate->t_uv || !gate->q_umin || !gate->q_umax) return svi::fail(SVI_ERR_INVALID, "svi_gate with null member");
        SVI_HIP(hipMemcpyAsync(d + oquv, gate->q_uv, 8ull * nq, hipMemcpyHostToDevice, m->stream));
        if (nt) SVI_HIP(hipMemcpyAsync(d + otuv, gate->t_uv, 8ull * nt, hipMemcpyHostToDevice, m->stream));
        SVI_HIP(hipMemcpyAsync(d + omin, gate->q_umin, 4ull * nq, hipMemcpyHostToDevice, m->stream));
        SVI_HIP(hipMemcpyAsync(d + omax, gate->q_umax, 4ull * nq, hipMemcpyHostToDevice, m->stream));
        g.q_uv = reinterpret_cast<float*>(d + oquv); g.t_uv = reinterpret_cast<float*>(d + otuv);
        g.q_umin = reinterpret_cast<float*>(d + omin); g.q_umax = reinterpret_cast<float*>(d + omax);
        g.v_tol = gate->v_tol;
    }
    if (int rc = svi_match_hamming256_dev(m, reinterpret_cast<uint8_t*>(d + oq), nq, reinterpret_cast<uint8_t*>(d + ot), nt, 1,
                                          gate ? &g : nullptr, max_dist_exclusive, reinterpret_cast<int32_t*>(d + oidx),
                                          reinterpret_cast<int32_t*>(d + odist)))
        return rc;
    SVI_HIP(hipMemcpyAsync(out_idx, d + oidx, 4ull * nq, hipMemcpyDeviceToHost, m->stream));
    SVI_HIP(hipMemcpyAsync(out_dist, d + odist, 4ull * nq, hipMemcpyDeviceToHost, m->stream));
    SVI_HIP(hipStreamSynchronize(m->stream));
    return SVI_OK;
}

int svi_hamming256_pairs_dev(svi_matcher* m, const uint8_t* a, const uint8_t* b, int n, int32_t* dist)
{
    if (!m) return svi::fail(SVI_ERR_INVALID, "null matcher");
    if (n < 0) return svi::fail(SVI_ERR_INVALID, "negative size");
    if (n == 0) return SVI_OK;
    if (!a || !b || !dist) return svi::fail(SVI_ERR_INVALID, "null pointer");
    SVI_HIP(svi::enter_device(m->device));
    hipLaunchKernelGGL(k_hamming256_pairs, dim3((n + 255) / 256), dim3(256), 0, m->stream,
                       reinterpret_cast<const uint4*>(a), reinterpret_cast<const uint4*>(b), n, dist);
    SVI_HIP(hipGetLastError());
    return SVI_OK;
}

int svi_hamming256_pairs(svi_matcher* m, const uint8_t* a, const uint8_t* b, int n, int32_t* dist)
{
    if (!m) return svi::fail(SVI_ERR_INVALID, "null matcher");
    if (n < 0) return svi::fail(SVI_ERR_INVALID, "negative size");
    if (n == 0) return SVI_OK;
    if (!a || !b || !dist) return svi::fail(SVI_ERR_INVALID, "null pointer");
    SVI_HIP(svi::enter_device(m->device));
    const size_t nb = 32ull * n, off_b = (nb + 255) & ~size_t(255), off_d = 2 * off_b;
    if (int rc = m->scratch.reserve(off_d + 4ull * n)) return rc;
    char* d = m->scratch.as<char>();
    SVI_HIP(hipMemcpyAsync(d, a, nb, hipMemcpyHostToDevice, m->stream));
    SVI_HIP(hipMemcpyAsync(d + off_b, b, nb, hipMemcpyHostToDevice, m->stream));
    if (int rc = svi_hamming256_pairs_dev(m, reinterpret_cast<uint8_t*>(d), reinterpret_cast<uint8_t*>(d + off_b), n,
                                          reinterpret_cast<int32_t*>(d + off_d)))
        return rc;
    SVI_HIP(hipMemcpyAsync(dist, d + off_d, 4ull * n, hipMemcpyDeviceToHost, m->stream));
    SVI_HIP(hipStreamSynchronize(m->stream));
    return SVI_OK;
}

int svi_triangulate_rectified_dev(svi_matcher* m, double f, double cx, double cy, double duR_flipped, double min_disparity,
                                  const float* uvL, const float* uvR, int n, double* xyz, uint8_t* ok)
{
    if (!m) return svi::fail(SVI_ERR_INVALID, "null matcher");
    if (n < 0) return svi::fail(SVI_ERR_INVALID, "negative size");
    if (n == 0) return SVI_OK;
    if (!uvL || !uvR || !xyz || !ok) return svi::fail(SVI_ERR_INVALID, "null pointer");
    if (!(f != 0.0)) return svi::fail(SVI_ERR_INVALID, "focal length must be non-zero");
    SVI_HIP(svi::enter_device(m->device));
    hipLaunchKernelGGL(k_triangulate, dim3((n + 255) / 256), dim3(256), 0, m->stream, reinterpret_cast<const float2*>(uvL),
                       reinterpret_cast<const float2*>(uvR), n, 1.0 / f, cx, cy, duR_flipped, min_disparity, xyz, ok);
    SVI_HIP(hipGetLastError());
    return SVI_OK;
}

int svi_triangulate_rectified(svi_matcher* m, double f, double cx, double cy, double duR_flipped, double min_disparity,
                              const float* uvL, const float* uvR, int n, double* xyz, uint8_t* ok)
{
    if (!m) return svi::fail(SVI_ERR_INVALID, "null matcher");
    if (n < 0) return svi::fail(SVI_ERR_INVALID, "negative size");
    if (n == 0) return SVI_OK;
    if (!uvL || !uvR || !xyz || !ok) return svi::fail(SVI_ERR_INVALID, "null pointer");
    SVI_HIP(svi::enter_device(m->device));
    auto al = [](size_t x) { return (x + 255) & ~size_t(255); };
    const size_t oL = 0, oR = al(8ull * n), oX = oR + al(8ull * n), oK = oX + al(24ull * n), total = oK + al(n);
    if (int rc = m->scratch.reserve(total)) return rc;
    char* d = m->scratch.as<char>();
    SVI_HIP(hipMemcpyAsync(d + oL, uvL, 8ull * n, hipMemcpyHostToDevice, m->stream));
    SVI_HIP(hipMemcpyAsync(d + oR, uvR, 8ull * n, hipMemcpyHostToDevice, m->stream));
    if (int rc = svi_triangulate_rectified_dev(m, f, cx, cy, duR_flipped, min_disparity, reinterpret_cast<float*>(d + oL),
                                               reinterpret_cast<float*>(d + oR), n, reinterpret_cast<double*>(d + oX),
                                               reinterpret_cast<uint8_t*>(d + oK)))
        return rc;
    SVI_HIP(hipMemcpyAsync(xyz, d + oX, 24ull * n, hipMemcpyDeviceToHost, m->stream));
    SVI_HIP(hipMemcpyAsync(ok, d + oK, n, hipMemcpyDeviceToHost, m->stream));
    SVI_HIP(hipStreamSynchronize(m->stream));
    return SVI_OK;
}

} // extern "C"
