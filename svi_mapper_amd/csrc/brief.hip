// brief.hip — BRIEF-256 extraction on the MI355X: the work the reference hands to
// cv::xfeatures2d::BriefDescriptorExtractor::compute( image( roi ), keypoints, descriptors )
//   src/core/CTriangulator.cpp:11 (create, 32 bytes), :84, :147, :218, :289
//   src/core/CFundamentalMatcher.cpp:401, :450, :534, :651, :2345
// for every candidate pool of every landmark.  OpenCV's algorithm (xfeatures2d brief.cpp, restated):
// integral image, drop key points within 28 px of the ROI border, 256 comparisons of 9x9 box sums.
// OpenCV's baked test-pair table is not available offline, so the table is an input of svi_brief_create.
//
//   k_integral_rows / k_integral_cols   int32 integral image of a whole frame, once per image
//   k_brief_count                       wavefront per ROI: key points that survive the border filter
//   k_scan_i32 (tracker.hip twin)       segment starts of the compacted pools
//   k_brief_compact                     wavefront per ROI: stable compaction (ballot + popcount ranks)
//   k_brief_describe                    wavefront per kept key point: its 58x58 window of the integral image staged in LDS
//                                       (coalesced rows), lane = 4 tests (8 box sums, 32 LDS reads), bits gathered with shuffles
// Integer work, latency / cache bound (a frame: ~66 k key points x 2048 integral reads).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>

#include "common.h"
#include "matcher_handle.h"

struct svi_brief {
    svi_matcher* m = nullptr;
    int8_t* pattern = nullptr;      // device [256][4] (y1, x1, y2, x2)
    int32_t* sum[2] = {nullptr, nullptr};
    int w[2] = {0, 0}, h[2] = {0, 0};
    size_t cap[2] = {0, 0};
    svi::DevBuf owner;              // [total] ROI of every kept key point
};

namespace {

__global__ __launch_bounds__(256) void k_integral_rows(const uint8_t* __restrict__ img, int w, int h, int stride, int32_t* __restrict__ sum)
{
    // one wavefront per image row: running prefix in chunks of 64 pixels
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= h) return;
    int32_t carry = 0;
    int32_t* out = sum + (size_t)(row + 1) * (w + 1);
    if (lane == 0) out[0] = 0;
    for (int x0 = 0; x0 < w; x0 += 64) {
        const int x = x0 + lane;
        int32_t v = x < w ? img[(size_t)row * stride + x] : 0;
        for (int off = 1; off < 64; off <<= 1) {
            const int32_t o = __shfl_up(v, off);
            if (lane >= off) v += o;
        }
        if (x < w) out[x + 1] = carry + v;
        carry += __shfl(v, 63);
    }
}

constexpr int kColSegs = 8; // row segments of the column pass, one wavefront each

__global__ __launch_bounds__(64 * kColSegs) void k_integral_cols(int w, int h, int32_t* __restrict__ sum)
{
    // workgroup = 64 columns; wavefront s owns the rows of segment s: running sums inside the segment first, then the
    // totals of the segments above are added (two short dependent chains instead of one of length h)
    __shared__ int32_t s_tot[kColSegs][64];
    const int lane = threadIdx.x & 63, seg = threadIdx.x >> 6;
    const int x = blockIdx.x * 64 + lane;
    const int rows = (h + kColSegs - 1) / kColSegs;
    const int y0 = 1 + seg * rows, y1 = min(y0 + rows, h + 1);
    const size_t W = (size_t)w + 1;
    int32_t run = 0;
    if (x <= w) {
        if (seg == 0) sum[x] = 0;
        for (int yb = y0; yb < y1; yb += 16) { // 16 rows in flight: loads first, then the running sums, then the stores
            int32_t v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = (yb + u < y1) ? sum[(size_t)(yb + u) * W + x] : 0;
#pragma unroll
            for (int u = 0; u < 16; ++u) { run += v[u]; v[u] = run; }
#pragma unroll
            for (int u = 0; u < 16; ++u) if (yb + u < y1) sum[(size_t)(yb + u) * W + x] = v[u];
        }
    }
    s_tot[seg][lane] = run;
    __syncthreads();
    int32_t off = 0;
    for (int q = 0; q < seg; ++q) off += s_tot[q][lane];
    if (x <= w && off != 0)
        for (int yb = y0; yb < y1; yb += 16) {
            int32_t v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = (yb + u < y1) ? sum[(size_t)(yb + u) * W + x] : 0;
#pragma unroll
            for (int u = 0; u < 16; ++u) if (yb + u < y1) sum[(size_t)(yb + u) * W + x] = v[u] + off;
        }
}

__device__ __forceinline__ bool roi_ok(const int4 r, int w, int h)
{
    return r.z > 56 && r.w > 56 && r.x >= 0 && r.y >= 0 && r.x + r.z <= w && r.y + r.w <= h;
}

__device__ __forceinline__ bool keep_point(const float2 p, const int4 r)
{
    if (!(fabsf(p.x) < 1.0e8f) || !(fabsf(p.y) < 1.0e8f)) return false;
    const int qx = __float2int_rn(p.x), qy = __float2int_rn(p.y); // cvRound: round half to even
    return qx >= 28 && qx < r.z - 28 && qy >= 28 && qy < r.w - 28;
}

__global__ __launch_bounds__(256) void k_brief_count(const int4* __restrict__ roi, const int32_t* __restrict__ seg, const float2* __restrict__ kp,
                                                     int n, int w, int h, int32_t* __restrict__ seg_out)
{
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= n) return;
    const int4 r = roi[i];
    int cnt = 0;
    if (roi_ok(r, w, h))
        for (int k = seg[i] + lane; k < seg[i + 1]; k += 64) cnt += keep_point(kp[k], r) ? 1 : 0;
    for (int off = 32; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off);
    if (lane == 0) seg_out[i] = cnt;
}

__global__ __launch_bounds__(1024) void k_brief_scan(int32_t* __restrict__ seg, int n)
{
    __shared__ int32_t s_wave[16];
    __shared__ int32_t s_carry;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_carry = 0;
    __syncthreads();
    for (int base = 0; base < n; base += 1024) {
        const int i = base + tid;
        const int32_t v = i < n ? seg[i] : 0;
        int32_t inc = v;
        for (int off = 1; off < 64; off <<= 1) {
            const int32_t o = __shfl_up(inc, off);
            if (lane >= off) inc += o;
        }
        if (lane == 63) s_wave[wave] = inc;
        __syncthreads();
        int32_t wpre = 0;
        for (int q = 0; q < wave; ++q) wpre += s_wave[q];
        const int32_t carry = s_carry;
        if (i < n) seg[i] = carry + wpre + inc - v;
        __syncthreads();
        if (tid == 1023) s_carry = carry + wpre + inc;
        __syncthreads();
    }
    if (tid == 0) seg[n] = s_carry;
}

__global__ __launch_bounds__(256) void k_brief_compact(const int4* __restrict__ roi, const int32_t* __restrict__ seg, const float2* __restrict__ kp,
                                                       int n, int w, int h, const int32_t* __restrict__ seg_out, float2* __restrict__ kp_out,
                                                       int32_t* __restrict__ owner)
{
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= n) return;
    const int4 r = roi[i];
    if (!roi_ok(r, w, h)) return;
    int base = seg_out[i];
    for (int k0 = seg[i]; k0 < seg[i + 1]; k0 += 64) {
        const int k = k0 + lane;
        const bool in = k < seg[i + 1];
        const float2 p = in ? kp[k] : make_float2(0.f, 0.f);
        const bool keep = in && keep_point(p, r);
        const unsigned long long mask = __ballot(keep);
        if (keep) {
            const int pos = base + __popcll(mask & ((1ull << lane) - 1ull));
            kp_out[pos] = p;
            owner[pos] = i;
        }
        base += __popcll(mask);
    }
}

constexpr int kPatch = 58; // integral samples around a key point: offsets -28 .. +29 (24 + 4 and 24 + 5)

__global__ __launch_bounds__(256) void k_brief_describe(const int32_t* __restrict__ sum, int w, int h, const int8_t* __restrict__ pattern,
                                                        const int4* __restrict__ roi, const float2* __restrict__ kp_out,
                                                        const int32_t* __restrict__ owner, const int32_t* __restrict__ total,
                                                        uint32_t* __restrict__ desc)
{
    // one wavefront per key point: its 58 x 58 window of the integral image is staged in LDS with coalesced row loads
    // (coordinates clamped to the frame), then every lane evaluates 4 tests = 8 box sums = 32 LDS reads
    __shared__ int32_t s_patch[4][kPatch * kPatch];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int j = blockIdx.x * 4 + wave;
    if (j >= *total) return;
    const int4 r = roi[owner[j]];
    const float2 p = kp_out[j];
    const int cx = r.x + static_cast<int>(static_cast<double>(p.x) + 0.5), cy = r.y + static_cast<int>(static_cast<double>(p.y) + 0.5);
    int32_t* patch = s_patch[wave];
    const size_t W = (size_t)w + 1;
    if (lane < kPatch) {
        const int gx = min(max(cx - 28 + lane, 0), w);
        // all 58 row loads of the lane are issued before the first one is parked (one round trip, not 58)
        int32_t v[kPatch];
#pragma unroll
        for (int yy = 0; yy < kPatch; ++yy) v[yy] = sum[(size_t)min(max(cy - 28 + yy, 0), h) * W + gx];
#pragma unroll
        for (int yy = 0; yy < kPatch; ++yy) patch[yy * kPatch + lane] = v[yy];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // box centred at offset (ox, oy): corners at patch[(28 + oy + {5,-4})][(28 + ox + {5,-4})]
    auto box = [&](int oy, int ox) {
        const int x0 = 24 + ox, x1 = 33 + ox, y0 = (24 + oy) * kPatch, y1 = (33 + oy) * kPatch;
        return patch[y1 + x1] - patch[y1 + x0] - patch[y0 + x1] + patch[y0 + x0];
    };
    uint32_t bits = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const char4 t = reinterpret_cast<const char4*>(pattern)[4 * lane + q];
        const int32_t a = box(t.x, t.y);
        const int32_t b = box(t.z, t.w);
        const int tt = 4 * lane + q;
        if (a < b) bits |= 1u << (7 - (tt & 7));
    }
    // byte b = lanes 2b (high nibble tests) | 2b+1; word = 4 bytes = 8 lanes, byte 0 in the low bits (little endian)
    bits |= __shfl_xor(bits, 1);                  // both lanes of a pair now hold the byte
    uint32_t word = bits << (8 * ((lane >> 1) & 3));
    word |= __shfl_xor(word, 2);
    word |= __shfl_xor(word, 4);
    if ((lane & 7) == 0) desc[8 * (size_t)j + (lane >> 3)] = word;
}

} // namespace

extern "C" {

int svi_brief_create(svi_matcher* m, const int8_t* pattern, svi_brief** out)
{
    if (!m || !pattern || !out) return svi::fail(SVI_ERR_INVALID, "svi_brief_create: null argument");
    for (int t = 0; t < 1024; ++t)
        if (pattern[t] < -24 || pattern[t] > 24) return svi::fail(SVI_ERR_INVALID, "svi_brief_create: test offset %d outside the 48x48 patch", (int)pattern[t]);
    SVI_HIP(svi::enter_device(m->device));
    svi_brief* b = new svi_brief();
    b->m = m;
    if (hipMalloc(reinterpret_cast<void**>(&b->pattern), 1024) != hipSuccess) { delete b; return svi::fail(SVI_ERR_HIP, "hipMalloc failed"); }
    SVI_HIP(hipMemcpy(b->pattern, pattern, 1024, hipMemcpyHostToDevice));
    *out = b;
    return SVI_OK;
}

int svi_brief_destroy(svi_brief* b)
{
    if (!b) return SVI_OK;
    (void)hipSetDevice(b->m->device);
    (void)hipStreamSynchronize(b->m->stream);
    if (b->pattern) (void)hipFree(b->pattern);
    for (int s = 0; s < 2; ++s) if (b->sum[s]) (void)hipFree(b->sum[s]);
    b->owner.release();
    delete b;
    return SVI_OK;
}

int svi_brief_set_image_dev(svi_brief* b, int side, const uint8_t* image, int width, int height, int stride)
{
    if (!b || !image) return svi::fail(SVI_ERR_INVALID, "svi_brief_set_image_dev: null argument");
    if (side < 0 || side > 1 || width <= 0 || height <= 0 || stride < width || width > 16384 || height > 16384)
        return svi::fail(SVI_ERR_INVALID, "svi_brief_set_image_dev: bad side / size");
    hipStream_t st = b->m->stream;
    SVI_HIP(svi::enter_device(b->m->device));
    const size_t need = sizeof(int32_t) * (size_t)(width + 1) * (height + 1);
    if (b->cap[side] < need) {
        SVI_HIP(hipStreamSynchronize(st));
        if (b->sum[side]) (void)hipFree(b->sum[side]);
        b->sum[side] = nullptr; b->cap[side] = 0;
        SVI_HIP(hipMalloc(reinterpret_cast<void**>(&b->sum[side]), need));
        b->cap[side] = need;
    }
    b->w[side] = width; b->h[side] = height;
    hipLaunchKernelGGL(k_integral_rows, dim3((height + 3) / 4), dim3(256), 0, st, image, width, height, stride, b->sum[side]);
    hipLaunchKernelGGL(k_integral_cols, dim3((width + 1 + 63) / 64), dim3(64 * kColSegs), 0, st, width, height, b->sum[side]);
    SVI_HIP(hipGetLastError());
    return SVI_OK;
}

int svi_brief_integral_dev(svi_brief* b, int side, int32_t* out)
{
    if (!b || !out || side < 0 || side > 1 || !b->sum[side]) return svi::fail(SVI_ERR_INVALID, "svi_brief_integral_dev: no image set");
    SVI_HIP(svi::enter_device(b->m->device));
    SVI_HIP(hipMemcpyAsync(out, b->sum[side], sizeof(int32_t) * (size_t)(b->w[side] + 1) * (b->h[side] + 1), hipMemcpyDeviceToDevice, b->m->stream));
    return SVI_OK;
}

int svi_brief_compute_dev(svi_brief* b, int side, const int32_t* roi, const int32_t* seg, const float* kp_uv, int n, int64_t total_in,
                          int32_t* seg_out, float* kp_out, uint8_t* desc_out, int64_t* total_out)
{
    if (!b || side < 0 || side > 1) return svi::fail(SVI_ERR_INVALID, "svi_brief_compute_dev: bad handle / side");
    if (!b->sum[side]) return svi::fail(SVI_ERR_STATE, "svi_brief_compute_dev: svi_brief_set_image_dev has not been called for this side");
    if (n < 0 || total_in < 0 || !seg_out) return svi::fail(SVI_ERR_INVALID, "svi_brief_compute_dev: bad sizes / seg_out");
    if (n > 0 && (!roi || !seg)) return svi::fail(SVI_ERR_INVALID, "svi_brief_compute_dev: null roi / seg");
    if (total_in > 0 && (!kp_uv || !kp_out || !desc_out)) return svi::fail(SVI_ERR_INVALID, "svi_brief_compute_dev: null key point / output array");
    if (reinterpret_cast<uintptr_t>(desc_out) & 3) return svi::fail(SVI_ERR_INVALID, "svi_brief_compute_dev: desc_out must be 4-byte aligned");
    hipStream_t st = b->m->stream;
    SVI_HIP(svi::enter_device(b->m->device));
    if (b->owner.cap < sizeof(int32_t) * (size_t)std::max<int64_t>(total_in, 1)) SVI_HIP(hipStreamSynchronize(st));
    if (int rc = b->owner.reserve(sizeof(int32_t) * (size_t)std::max<int64_t>(total_in, 1))) return rc;
    const int w = b->w[side], h = b->h[side];
    const int4* r4 = reinterpret_cast<const int4*>(roi);
    const float2* kp = reinterpret_cast<const float2*>(kp_uv);
    if (n > 0) hipLaunchKernelGGL(k_brief_count, dim3((n + 3) / 4), dim3(256), 0, st, r4, seg, kp, n, w, h, seg_out);
    hipLaunchKernelGGL(k_brief_scan, dim3(1), dim3(1024), 0, st, seg_out, n);
    if (n > 0 && total_in > 0) {
        hipLaunchKernelGGL(k_brief_compact, dim3((n + 3) / 4), dim3(256), 0, st, r4, seg, kp, n, w, h, seg_out, reinterpret_cast<float2*>(kp_out),
                           b->owner.as<int32_t>());
        // the grid covers the upper bound; workgroups beyond the kept count leave at once
        hipLaunchKernelGGL(k_brief_describe, dim3((unsigned)((total_in + 3) / 4)), dim3(256), 0, st, b->sum[side], w, h, b->pattern, r4,
                           reinterpret_cast<const float2*>(kp_out), b->owner.as<int32_t>(), seg_out + n, reinterpret_cast<uint32_t*>(desc_out));
    }
    SVI_HIP(hipGetLastError());
    if (total_out) {
        int32_t t = 0;
        SVI_HIP(hipMemcpyAsync(&t, seg_out + n, sizeof(int32_t), hipMemcpyDeviceToHost, st));
        SVI_HIP(hipStreamSynchronize(st));
        *total_out = t;
    }
    return SVI_OK;
}

} // extern "C"
