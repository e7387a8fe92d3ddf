// track_cascade.hip — CFundamentalMatcher's cascades behind the C ABI.
//
// The per-pass kernels live in tracker.hip (one launch over all landmarks per step); this file is the C++ host that
// strings them into the reference's entry points
//   getPoseStereoPosit  src/core/CFundamentalMatcher.cpp:340-760   stage 1 (LEFT, RIGHT) -> stage 2 (LEFT, RIGHT) -> pose
//   trackEpipolar       :794-1315                                  stage 3 where the detection point moved, else stage 2
//   trackManual         :1366-2019                                 stage 1 -> stage 2 -> stage 3
//   addNewLandmarks     :83-193                                    stereo partner of fresh key points
// plus the few glue kernels between the passes.  The reference's control flow is try / catch per landmark; here a
// landmark is a ROW of every array and the flow is a u8 mask per pass: a pass runs over all n rows, rows outside its
// mask own empty pool segments and report SVI_TRK_MATCH_SKIPPED.  Nothing is compacted, so nothing has to be counted -
// the host waits only where a ragged pool must be sized before it can be allocated (the row candidates of the stereo
// search, a caller-supplied extractor / detector reporting what it kept).  With the built-in BRIEF extractor the
// kernels read their segment tables from the device and a stage runs with one wait per side.
//
// Bit-exactness: the float expressions of the glue (4s, 8s + 1, kp + 4s, cv::Rect corner rounding) are single IEEE
// operations in the reference's operand order; the file is compiled with -ffp-contract=off like tracker.hip.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <vector>

#include "common.h"
#include "matcher_handle.h"

struct svi_tracker {
    svi_matcher* m = nullptr;
    svi_track_camera cam{};
    svi_brief* brief = nullptr;
    svi_extract_fn ext = nullptr;
    void* ext_user = nullptr;
    svi_detect_fn det = nullptr;
    void* det_user = nullptr;
    int64_t det_cap = 0;
    // frame
    svi_track_landmarks lm{};
    bool planned = false;
    int64_t total_samples = 0;
    int64_t n_no_motion = 0;       // landmarks of the frame whose detection point has not moved (stage 2 stands in for stage 3)
    // workspace: one growable device buffer per role
    enum { kRecords, kS3Seg, kMaskA, kMaskB, kMaskC, kMaskFound, kMaskRun, kUvRef, kTopLeft, kOk, kRoi, kRoiI, kSegIn, kKpIn, kSegE, kKpE, kDescE, kIdx,
           kDist, kStatus, kDescHere, kSegR, kStRange, kRoi2, kPoolUv, kSeg2, kPoolUv2, kPool, kIdx2, kDist2, kStatus2, kUvOther, kXyz, kDescOther,
           kRect, kSegD, kKpD, kPositMask, kCount };
    svi::DevBuf ws[kCount];
    template <class T> int take(int slot, size_t count, T** out)
    {
        // a buffer is only ever re-allocated between two uses that are ordered on the stream and hipFree drains the device
        if (int rc = ws[slot].reserve(sizeof(T) * (count ? count : 1))) return rc;
        *out = ws[slot].as<T>();
        return SVI_OK;
    }
};

namespace {

#define SVI_TRY(x) do { int rc_ = (x); if (rc_ != SVI_OK) return rc_; } while (0)

constexpr int kB = 256;
inline dim3 grid_for(int n) { return dim3((unsigned)((n + kB - 1) / kB)); }

// ---- masks -------------------------------------------------------------------------------------------------------------
enum { kMaskFovBoth = 0, kMaskEpiOk = 1, kMaskNoMotionFov = 2 };
__global__ __launch_bounds__(kB) void k_mask(const svi_track_record* __restrict__ rec, const uint8_t* __restrict__ active, int mode, int n,
                                             uint8_t* __restrict__ out)
{
    const int i = blockIdx.x * kB + threadIdx.x;
    if (i >= n) return;
    const int st = rec[i].status;
    const bool fov = (st & SVI_TRK_FOV_LEFT) && (st & SVI_TRK_FOV_RIGHT);
    bool v = mode == kMaskFovBoth ? fov : mode == kMaskEpiOk ? (st & SVI_TRK_EPI_OK) != 0 : (fov && (st & SVI_TRK_EPI_NO_MOTION));
    if (active && !active[i]) v = false;
    out[i] = v ? 1 : 0;
}

// how many landmarks are inside both fields of view but have no epipolar line (their detection point did not move): counted
// with the plan so that trackEpipolar knows without asking the device again whether its stage-2 branch has anything to do
__global__ __launch_bounds__(kB) void k_count_no_motion(const svi_track_record* __restrict__ rec, int n, const int32_t* __restrict__ seg, int32_t* __restrict__ out)
{
    const int i = blockIdx.x * kB + threadIdx.x;
    int hit = 0;
    if (i < n) {
        const int st = rec[i].status;
        hit = ((st & SVI_TRK_FOV_LEFT) && (st & SVI_TRK_FOV_RIGHT) && (st & SVI_TRK_EPI_NO_MOTION)) ? 1 : 0;
    }
    const unsigned long long m = __ballot(hit);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(out + 1, (int32_t)__popcll(m));
    if (i == 0) out[0] = seg[n];
}

// rows of `run` whose status is not OK: what the next stage of trackManual takes over
__global__ __launch_bounds__(kB) void k_lost(const uint8_t* __restrict__ run, const int32_t* __restrict__ status, int n, uint8_t* __restrict__ out)
{
    const int i = blockIdx.x * kB + threadIdx.x;
    if (i < n) out[i] = (run[i] && status[i] != SVI_TRK_MATCH_OK) ? 1 : 0;
}

__global__ __launch_bounds__(kB) void k_init_result(svi_track_result r, int n)
{
    const int i = blockIdx.x * kB + threadIdx.x;
    if (i >= n) return;
    r.status[i] = SVI_TRK_MATCH_SKIPPED;
    if (r.stage) r.stage[i] = 0;
    r.uv_left[2 * i] = r.uv_left[2 * i + 1] = r.uv_right[2 * i] = r.uv_right[2 * i + 1] = 0.f;
    r.xyz_left[3 * i] = r.xyz_left[3 * i + 1] = r.xyz_left[3 * i + 2] = 0.0;
    uint4* dl = reinterpret_cast<uint4*>(r.desc_left) + 2 * (size_t)i;
    uint4* dr = reinterpret_cast<uint4*>(r.desc_right) + 2 * (size_t)i;
    dl[0] = dl[1] = dr[0] = dr[1] = make_uint4(0, 0, 0, 0);
}

// in-place exclusive scan of counts[0..n) -> seg[0..n], one workgroup (the same scheme as tracker.hip)
__global__ __launch_bounds__(1024) void k_scan(int32_t* __restrict__ seg, int n)
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
        for (int w = 0; w < wave; ++w) wpre += s_wave[w];
        const int32_t carry = s_carry;
        if (i < n) seg[i] = carry + wpre + inc - v;
        __syncthreads();
        if (tid == 1023) s_carry = carry + wpre + inc;
        __syncthreads();
    }
    if (tid == 0) seg[n] = s_carry;
}

// ---- stage 1: one key point at (4s, 4s) of the (8s+1)^2 ROI around the projection (:395-400 / :449-454) ---------------
__global__ __launch_bounds__(kB) void k_s1_counts(const uint8_t* __restrict__ run, int n, int32_t* __restrict__ seg)
{
    const int i = blockIdx.x * kB + threadIdx.x;
    if (i < n) seg[i] = run[i] ? 1 : 0;
}
__global__ __launch_bounds__(kB) void k_s1_inputs(const svi_track_record* __restrict__ rec, const float* __restrict__ kp_size,
                                                  const uint8_t* __restrict__ run, int side, int n, const int32_t* __restrict__ seg,
                                                  float* __restrict__ roi, float2* __restrict__ kp_uv)
{
    const int i = blockIdx.x * kB + threadIdx.x;
    if (i >= n) return;
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
    if (run[i]) {
        const float s = kp_size[i];
        const float* xy = side == 0 ? rec[i].s1_roi_left : rec[i].s1_roi_right;
        const float len = 8 * s + 1;                      // fKeyPointSizePixelsLength :383
        r = make_float4(xy[0], xy[1], len, len);
        kp_uv[seg[i]] = make_float2(4 * s, 4 * s);        // ptOffsetKeyPointHalf :384
    }
    reinterpret_cast<float4*>(roi)[i] = r;
}

// ---- stage 2: search rectangle for the detector, grown rectangle for the extractor (:505-530) ---------------------------
__global__ __launch_bounds__(kB) void k_s2_inputs(const svi_track_record* __restrict__ rec, const uint8_t* __restrict__ run, int side, int n,
                                                  float* __restrict__ rect, float* __restrict__ ext)
{
    const int i = blockIdx.x * kB + threadIdx.x;
    if (i >= n) return;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
    if (run[i]) {
        const float* q = side == 0 ? rec[i].s2_left : rec[i].s2_right;
        const float* e = side == 0 ? rec[i].s2_ext_left : rec[i].s2_ext_right;
        a = make_float4(q[0], q[1], q[2], q[3]);
        // cv::Rect( Point2f, Point2f ): the corners are cvRound()ed (round half to even), then width = lr - ul
        const float c0 = rintf(e[0]), c1 = rintf(e[1]), c2 = rintf(e[2]), c3 = rintf(e[3]);
        b = make_float4(c0, c1, c2 - c0, c3 - c1);
    }
    reinterpret_cast<float4*>(rect)[i] = a;
    reinterpret_cast<float4*>(ext)[i] = b;
}
// detected key points move by (4s, 4s) into the grown rectangle (:533); one wavefront per landmark
__global__ __launch_bounds__(kB) void k_shift_kp(const float* __restrict__ kp_size, const int32_t* __restrict__ seg, int n,
                                                 const float2* __restrict__ kp_in, float2* __restrict__ kp_out)
{
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (w >= n) return;
    const float h = 4 * kp_size[w];
    for (int k = seg[w] + lane; k < seg[w + 1]; k += 64) kp_out[k] = make_float2(kp_in[k].x + h, kp_in[k].y + h);
}

// ---- stage 3: sample counts of the rows that run ------------------------------------------------------------------------
__global__ __launch_bounds__(kB) void k_s3_counts(const svi_track_record* __restrict__ rec, const uint8_t* __restrict__ run, int n,
                                                  int32_t* __restrict__ seg)
{
    const int i = blockIdx.x * kB + threadIdx.x;
    if (i < n) seg[i] = run[i] ? rec[i].s3_count : 0;
}

__global__ __launch_bounds__(kB) void k_roi_trunc(const float* __restrict__ roi, int n4, int32_t* __restrict__ out)
{
    const int i = blockIdx.x * kB + threadIdx.x;
    if (i < n4) out[i] = static_cast<int32_t>(roi[i]); // the implicit float -> int of cv::Rect( x, y, w, h )
}

// ---- after the temporal match: who goes on to the stereo search, and with which descriptor ----------------------------
// ok (nullable, stage 2): the hand-over's range test; a match that fails it becomes SVI_TRK_MATCH_RANGE ("out of tracking range")
__global__ __launch_bounds__(kB) void k_after_match(int32_t* __restrict__ status, const uint8_t* __restrict__ ok, const int32_t* __restrict__ seg,
                                                    const int32_t* __restrict__ idx, const uint4* __restrict__ desc, int n,
                                                    uint8_t* __restrict__ found, uint4* __restrict__ desc_here)
{
    const int i = blockIdx.x * kB + threadIdx.x;
    if (i >= n) return;
    int32_t st = status[i];
    bool f = st == SVI_TRK_MATCH_OK;
    if (f && ok && !ok[i]) { f = false; st = SVI_TRK_MATCH_RANGE; status[i] = st; }
    found[i] = f ? 1 : 0;
    uint4 d0 = make_uint4(0, 0, 0, 0), d1 = d0;
    if (f) { const size_t r = (size_t)seg[i] + idx[i]; d0 = desc[2 * r]; d1 = desc[2 * r + 1]; }
    desc_here[2 * (size_t)i] = d0; desc_here[2 * (size_t)i + 1] = d1;
}

// ---- end of a stage side: final status, measurement, and the mask of what is still to be tried -----------------------
struct FinishArgs {
    const uint8_t* run;        // rows this side worked on
    const uint8_t* found;      // temporal match found (the stereo search ran)
    const int32_t* st_match;   // status of the temporal match (or of the range test)
    const int32_t* st_range;   // status of the stereo range
    const int32_t* st_verify;  // status of the stereo verification
    const int32_t* seg2;       // stereo pool segments
    const int32_t* idx2;
    const uint4*   pool;
    const float2*  uv_here;    // pixel in the image the descriptor was found in (nullptr: the projected pixel of the record)
    const float2*  uv_other;
    const double*  xyz;
    const uint4*   desc_here;
    const svi_track_record* rec;
    int side;                  // 0: found in LEFT (other = RIGHT), 1: found in RIGHT
    int stage;
    uint8_t* next;             // nullable: run && final != OK
};
__global__ __launch_bounds__(kB) void k_finish(FinishArgs a, svi_track_result out, int n)
{
    const int i = blockIdx.x * kB + threadIdx.x;
    if (i >= n) return;
    if (!a.run[i]) { if (a.next) a.next[i] = 0; return; }
    int32_t st = a.st_match[i];
    if (a.found[i]) st = a.st_range[i] != SVI_TRK_MATCH_OK ? a.st_range[i] : a.st_verify[i];
    out.status[i] = st;
    if (a.next) a.next[i] = st != SVI_TRK_MATCH_OK ? 1 : 0;
    if (st != SVI_TRK_MATCH_OK) return;
    float2 here;
    if (a.uv_here) here = a.uv_here[i];
    else here = a.side == 0 ? make_float2(a.rec[i].uv_left[0], a.rec[i].uv_left[1]) : make_float2(a.rec[i].uv_right[0], a.rec[i].uv_right[1]);
    const float2 other = a.uv_other[i];
    float2* uvl = reinterpret_cast<float2*>(out.uv_left);
    float2* uvr = reinterpret_cast<float2*>(out.uv_right);
    uvl[i] = a.side == 0 ? here : other;
    uvr[i] = a.side == 0 ? other : here;
    out.xyz_left[3 * i] = a.xyz[3 * i]; out.xyz_left[3 * i + 1] = a.xyz[3 * i + 1]; out.xyz_left[3 * i + 2] = a.xyz[3 * i + 2];
    const size_t r = (size_t)a.seg2[i] + a.idx2[i];
    const uint4 o0 = a.pool[2 * r], o1 = a.pool[2 * r + 1];
    const uint4 h0 = a.desc_here[2 * (size_t)i], h1 = a.desc_here[2 * (size_t)i + 1];
    uint4* dl = reinterpret_cast<uint4*>(out.desc_left) + 2 * (size_t)i;
    uint4* dr = reinterpret_cast<uint4*>(out.desc_right) + 2 * (size_t)i;
    if (a.side == 0) { dl[0] = h0; dl[1] = h1; dr[0] = o0; dr[1] = o1; }
    else { dl[0] = o0; dl[1] = o1; dr[0] = h0; dr[1] = h1; }
    if (out.stage) out.stage[i] = (int8_t)a.stage;
}

// search_range of the records as a plain array (the LEFT stereo search reads it)
__global__ __launch_bounds__(kB) void k_search_range(const svi_track_record* __restrict__ rec, int n, float* __restrict__ out)
{
    const int i = blockIdx.x * kB + threadIdx.x;
    if (i < n) out[i] = rec[i].search_range;
}

// addNewLandmarks: top-left corner of the RIGHT search, window fMinimumSearchRangePixels = 60 (:119-120)
__global__ __launch_bounds__(kB) void k_new_topleft(const float2* __restrict__ uv, const float* __restrict__ kp_size, int n, float2* __restrict__ tl)
{
    const int i = blockIdx.x * kB + threadIdx.x;
    if (i >= n) return;
    const float half = 4 * kp_size[i];
    tl[i] = make_float2(fmaxf(uv[i].x - 60.0f - half, 0.0f), uv[i].y - half);
}

__global__ __launch_bounds__(kB) void k_ones(uint8_t* __restrict__ m, int n)
{
    const int i = blockIdx.x * kB + threadIdx.x;
    if (i < n) m[i] = 1;
}
__global__ __launch_bounds__(kB) void k_status_ok(const int32_t* __restrict__ status, int n, uint8_t* __restrict__ m)
{
    const int i = blockIdx.x * kB + threadIdx.x;
    if (i < n) m[i] = status[i] == SVI_TRK_MATCH_OK ? 1 : 0;
}

int read_i32(svi_matcher* m, const int32_t* p, int64_t* out)
{
    int32_t v = 0;
    SVI_HIP(hipMemcpyAsync(&v, p, sizeof(int32_t), hipMemcpyDeviceToHost, m->stream));
    SVI_HIP(hipStreamSynchronize(m->stream));
    *out = v;
    return SVI_OK;
}

// the extractor of the frame: built-in BRIEF (no host wait: kernels downstream read seg_out on the device) or the caller's
int extract(svi_tracker* t, int side, const float* roi, const int32_t* seg, const float* kp_uv, int n, int64_t total_in, int32_t* seg_out,
            float* kp_out, uint8_t* desc_out)
{
    svi_matcher* m = t->m;
    if (t->ext) {
        int64_t kept = 0;
        const int rc = t->ext(t->ext_user, side, roi, seg, kp_uv, n, total_in, seg_out, kp_out, desc_out, &kept, m->stream);
        if (rc != 0) return svi::fail(SVI_ERR_INVALID, "the extractor callback returned %d", rc);
        if (kept < 0 || kept > total_in) return svi::fail(SVI_ERR_INVALID, "the extractor callback kept %lld of %lld key points", (long long)kept, (long long)total_in);
        return SVI_OK;
    }
    if (!t->brief) return svi::fail(SVI_ERR_STATE, "no descriptor extractor: svi_tracker_set_brief or svi_tracker_set_extractor first");
    int32_t* roi_i = nullptr;
    SVI_TRY(t->take(svi_tracker::kRoiI, (size_t)4 * n, &roi_i));
    hipLaunchKernelGGL(k_roi_trunc, grid_for(4 * n), dim3(kB), 0, m->stream, roi, 4 * n, roi_i);
    return svi_brief_compute_dev(t->brief, side, roi_i, seg, kp_uv, n, total_in, seg_out, kp_out, desc_out, nullptr);
}

struct StereoOut { const int32_t *st_range, *st_verify, *seg2, *idx2; const uint4* pool; const float2* uv_other; const double* xyz; };

// CTriangulator::getPointTriangulatedInRIGHT / InLEFT for the rows of `found`, the caller's depth gate and descriptor check
int stereo(svi_tracker* t, int in_left, const float* kp_size, const float* search_range, const uint8_t* ref_desc, const uint8_t* last_other,
           const float* uv_ref, const float* topleft, const uint8_t* found, int n, int cutoff_other, int inclusive, bool depth_gate, StereoOut* so)
{
    svi_matcher* m = t->m;
    int32_t *seg_r, *st_range, *seg2, *idx2, *dist2, *st2;
    float *roi2, *pool_uv, *pool_uv2, *uv_other;
    uint8_t* pool;
    double* xyz;
    SVI_TRY(t->take(svi_tracker::kSegR, (size_t)n + 1, &seg_r));
    SVI_TRY(t->take(svi_tracker::kStRange, (size_t)n, &st_range));
    SVI_TRY(t->take(svi_tracker::kRoi2, (size_t)4 * n, &roi2));
    int64_t total = 0;
    SVI_TRY(svi_track_stereo_range_dev(m, t->cam.width, in_left, uv_ref, topleft, kp_size, search_range, found, n, seg_r, st_range, roi2, &total));
    SVI_TRY(t->take(svi_tracker::kPoolUv, (size_t)2 * total, &pool_uv));
    SVI_TRY(svi_track_stereo_candidates_dev(m, in_left, kp_size, n, seg_r, pool_uv));
    SVI_TRY(t->take(svi_tracker::kSeg2, (size_t)n + 1, &seg2));
    SVI_TRY(t->take(svi_tracker::kPoolUv2, (size_t)2 * total, &pool_uv2));
    SVI_TRY(t->take(svi_tracker::kPool, (size_t)32 * total, &pool));
    SVI_TRY(extract(t, in_left ? 0 : 1, roi2, seg_r, pool_uv, n, total, seg2, pool_uv2, pool));
    SVI_TRY(t->take(svi_tracker::kIdx2, (size_t)n, &idx2));
    SVI_TRY(t->take(svi_tracker::kDist2, (size_t)n, &dist2));
    SVI_TRY(t->take(svi_tracker::kStatus2, (size_t)n, &st2));
    SVI_TRY(t->take(svi_tracker::kUvOther, (size_t)2 * n, &uv_other));
    SVI_TRY(t->take(svi_tracker::kXyz, (size_t)3 * n, &xyz));
    svi_track_stereo_params prm{};
    const double dur = -t->cam.P_right[3];
    prm.f = t->cam.P_left[0]; prm.cx = t->cam.P_left[2]; prm.cy = t->cam.P_left[6]; // m_dFx, m_dPu, m_dPv (CTriangulator.cpp:14-17)
    prm.duR_flipped = dur; prm.min_disparity = 0.01;                                   // CTriangulator.h:21
    prm.depth_min = depth_gate ? dur / t->cam.width : -1.0e300;                        // CTriangulator.cpp:20-21
    prm.depth_max = depth_gate ? dur / 0.01 : 1.0e300;
    prm.cutoff_match = 100; prm.cutoff_other = cutoff_other; prm.other_inclusive = inclusive; prm.search_in_left = in_left;
    // the verification runs where the range was fine: rows of `found` with a range failure keep SVI_TRK_MATCH_RANGE
    uint8_t* run = nullptr;
    SVI_TRY(t->take(svi_tracker::kMaskRun, (size_t)n, &run));
    hipLaunchKernelGGL(k_status_ok, grid_for(n), dim3(kB), 0, m->stream, st_range, n, run);
    SVI_TRY(svi_track_stereo_verify_dev(m, &prm, ref_desc, last_other, run, uv_ref, topleft, n, seg2, pool, pool_uv2, idx2, dist2, st2, uv_other, xyz));
    so->st_range = st_range; so->st_verify = st2; so->seg2 = seg2; so->idx2 = idx2; so->pool = reinterpret_cast<const uint4*>(pool);
    so->uv_other = reinterpret_cast<const float2*>(uv_other); so->xyz = xyz;
    return SVI_OK;
}

int check_result(const svi_track_result* out)
{
    if (!out || !out->status || !out->uv_left || !out->uv_right || !out->xyz_left || !out->desc_left || !out->desc_right)
        return svi::fail(SVI_ERR_INVALID, "svi_track_result with a null array");
    if ((reinterpret_cast<uintptr_t>(out->desc_left) | reinterpret_cast<uintptr_t>(out->desc_right)) & 15)
        return svi::fail(SVI_ERR_INVALID, "svi_track_result: descriptor arrays must be 16-byte aligned");
    return SVI_OK;
}

int check_frame(svi_tracker* t, bool need_ref)
{
    if (!t) return svi::fail(SVI_ERR_INVALID, "null tracker");
    if (!t->planned) return svi::fail(SVI_ERR_STATE, "svi_tracker_plan has not been called for this frame");
    if (t->lm.n > 0 && (!t->lm.last_desc_left || !t->lm.last_desc_right)) return svi::fail(SVI_ERR_INVALID, "svi_track_landmarks without last descriptors");
    if (need_ref && t->lm.n > 0 && !t->lm.ref_desc_left) return svi::fail(SVI_ERR_INVALID, "svi_track_landmarks without reference descriptors");
    if ((reinterpret_cast<uintptr_t>(t->lm.last_desc_left) | reinterpret_cast<uintptr_t>(t->lm.last_desc_right) | reinterpret_cast<uintptr_t>(t->lm.ref_desc_left)) & 15)
        return svi::fail(SVI_ERR_INVALID, "svi_track_landmarks: descriptor arrays must be 16-byte aligned");
    return SVI_OK;
}

const svi_track_record* records_of(svi_tracker* t) { return t->ws[svi_tracker::kRecords].as<svi_track_record>(); }

void init_result(svi_tracker* t, const svi_track_result* out, int n)
{
    if (n > 0) hipLaunchKernelGGL(k_init_result, grid_for(n), dim3(kB), 0, t->m->stream, *out, n);
}

// ---- stage 1 (:391-486) over the rows of `run0`; `lost` (nullable) = rows of run0 that found nothing --------------------
int run_stage1(svi_tracker* t, const uint8_t* run0, const svi_track_result* out, uint8_t* lost)
{
    svi_matcher* m = t->m;
    const int n = t->lm.n;
    if (n == 0) return SVI_OK;
    const svi_track_record* rec = records_of(t);
    uint8_t *cur, *nxt, *found, *ok;
    float *uv_ref, *topleft, *roi, *kp_in, *kp_e, *rng;
    int32_t *seg_in, *seg_e, *idx, *dist, *status;
    uint8_t *desc_e, *desc_here;
    SVI_TRY(t->take(svi_tracker::kMaskA, (size_t)n, &cur));
    SVI_TRY(t->take(svi_tracker::kMaskB, (size_t)n, &nxt));
    SVI_TRY(t->take(svi_tracker::kMaskFound, (size_t)n, &found));
    SVI_TRY(t->take(svi_tracker::kOk, (size_t)n, &ok));
    SVI_TRY(t->take(svi_tracker::kUvRef, (size_t)2 * n, &uv_ref));
    SVI_TRY(t->take(svi_tracker::kTopLeft, (size_t)2 * n, &topleft));
    SVI_TRY(t->take(svi_tracker::kRoi, (size_t)4 * n, &roi));
    SVI_TRY(t->take(svi_tracker::kSegIn, (size_t)n + 1, &seg_in));
    SVI_TRY(t->take(svi_tracker::kKpIn, (size_t)2 * n, &kp_in));
    SVI_TRY(t->take(svi_tracker::kSegE, (size_t)n + 1, &seg_e));
    SVI_TRY(t->take(svi_tracker::kKpE, (size_t)2 * n, &kp_e));
    SVI_TRY(t->take(svi_tracker::kDescE, (size_t)32 * n, &desc_e));
    SVI_TRY(t->take(svi_tracker::kIdx, (size_t)n, &idx));
    SVI_TRY(t->take(svi_tracker::kDist, (size_t)n, &dist));
    SVI_TRY(t->take(svi_tracker::kStatus, (size_t)n, &status));
    SVI_TRY(t->take(svi_tracker::kDescHere, (size_t)32 * n, &desc_here));
    SVI_TRY(t->take(svi_tracker::kRect, (size_t)n, &rng));
    hipLaunchKernelGGL(k_search_range, grid_for(n), dim3(kB), 0, m->stream, rec, n, rng);
    SVI_HIP(hipMemcpyAsync(cur, run0, (size_t)n, hipMemcpyDeviceToDevice, m->stream));
    for (int side = 0; side < 2; ++side) {
        const uint8_t* here = side == 0 ? t->lm.last_desc_left : t->lm.last_desc_right;
        const uint8_t* there = side == 0 ? t->lm.last_desc_right : t->lm.last_desc_left;
        SVI_TRY(svi_track_handover_dev(m, side, rec, t->lm.kp_size, nullptr, n, nullptr, nullptr, nullptr, nullptr, uv_ref, topleft, ok));
        hipLaunchKernelGGL(k_s1_counts, grid_for(n), dim3(kB), 0, m->stream, cur, n, seg_in);
        hipLaunchKernelGGL(k_scan, dim3(1), dim3(1024), 0, m->stream, seg_in, n);
        hipLaunchKernelGGL(k_s1_inputs, grid_for(n), dim3(kB), 0, m->stream, rec, t->lm.kp_size, cur, side, n, seg_in, roi,
                           reinterpret_cast<float2*>(kp_in));
        SVI_TRY(extract(t, side, roi, seg_in, kp_in, n, n, seg_e, kp_e, desc_e));
        SVI_TRY(svi_match_ragged_dev(m, here, nullptr, cur, n, seg_e, desc_e, 25, 257, idx, dist, status));            // :404 / :453
        hipLaunchKernelGGL(k_after_match, grid_for(n), dim3(kB), 0, m->stream, status, nullptr, seg_e, idx, reinterpret_cast<const uint4*>(desc_e), n,
                           found, reinterpret_cast<uint4*>(desc_here));
        StereoOut so{};
        SVI_TRY(stereo(t, side, t->lm.kp_size, rng, desc_here, there, uv_ref, topleft, found, n, 25, 1, true, &so));    // :423 / :473
        // the measurement keeps the PROJECTED pixel of the image the descriptor was found in (:427 / :477)
        FinishArgs fa{cur, found, status, so.st_range, so.st_verify, so.seg2, so.idx2, so.pool, nullptr, so.uv_other, so.xyz,
                      reinterpret_cast<const uint4*>(desc_here), rec, side, 1, side == 0 ? nxt : lost};
        hipLaunchKernelGGL(k_finish, grid_for(n), dim3(kB), 0, m->stream, fa, *out, n);
        SVI_HIP(hipGetLastError());
        std::swap(cur, nxt);
    }
    return SVI_OK;
}

// ---- stage 2 (:489-709, :1042-1290) -----------------------------------------------------------------------------------------
int run_stage2(svi_tracker* t, const uint8_t* run0, const svi_track_result* out, uint8_t* lost)
{
    svi_matcher* m = t->m;
    const int n = t->lm.n;
    if (n == 0) return SVI_OK;
    if (!t->det) return svi::fail(SVI_ERR_STATE, "stage 2 needs a detector: svi_tracker_set_detector first");
    const svi_track_record* rec = records_of(t);
    uint8_t *cur, *nxt, *found, *ok;
    float *uv_ref, *topleft, *rect, *ext, *kp_d, *kp_in, *kp_e, *rng;
    int32_t *seg_d, *seg_e, *idx, *dist, *status;
    uint8_t *desc_e, *desc_here;
    const int64_t cap = t->det_cap;
    SVI_TRY(t->take(svi_tracker::kMaskA, (size_t)n, &cur));
    SVI_TRY(t->take(svi_tracker::kMaskB, (size_t)n, &nxt));
    SVI_TRY(t->take(svi_tracker::kMaskFound, (size_t)n, &found));
    SVI_TRY(t->take(svi_tracker::kOk, (size_t)n, &ok));
    SVI_TRY(t->take(svi_tracker::kUvRef, (size_t)2 * n, &uv_ref));
    SVI_TRY(t->take(svi_tracker::kTopLeft, (size_t)2 * n, &topleft));
    SVI_TRY(t->take(svi_tracker::kRect, (size_t)5 * n, &rect));   // [rect 4n | search range n]
    rng = rect + (size_t)4 * n;
    SVI_TRY(t->take(svi_tracker::kRoi, (size_t)4 * n, &ext));
    SVI_TRY(t->take(svi_tracker::kSegD, (size_t)n + 1, &seg_d));
    SVI_TRY(t->take(svi_tracker::kKpD, (size_t)2 * cap, &kp_d));
    SVI_TRY(t->take(svi_tracker::kKpIn, (size_t)2 * cap, &kp_in));
    SVI_TRY(t->take(svi_tracker::kSegE, (size_t)n + 1, &seg_e));
    SVI_TRY(t->take(svi_tracker::kKpE, (size_t)2 * cap, &kp_e));
    SVI_TRY(t->take(svi_tracker::kDescE, (size_t)32 * cap, &desc_e));
    SVI_TRY(t->take(svi_tracker::kIdx, (size_t)n, &idx));
    SVI_TRY(t->take(svi_tracker::kDist, (size_t)n, &dist));
    SVI_TRY(t->take(svi_tracker::kStatus, (size_t)n, &status));
    SVI_TRY(t->take(svi_tracker::kDescHere, (size_t)32 * n, &desc_here));
    hipLaunchKernelGGL(k_search_range, grid_for(n), dim3(kB), 0, m->stream, rec, n, rng);
    SVI_HIP(hipMemcpyAsync(cur, run0, (size_t)n, hipMemcpyDeviceToDevice, m->stream));
    for (int side = 0; side < 2; ++side) {
        const uint8_t* here = side == 0 ? t->lm.last_desc_left : t->lm.last_desc_right;
        const uint8_t* there = side == 0 ? t->lm.last_desc_right : t->lm.last_desc_left;
        hipLaunchKernelGGL(k_s2_inputs, grid_for(n), dim3(kB), 0, m->stream, rec, cur, side, n, rect, ext);
        SVI_HIP(hipGetLastError());
        int64_t total_d = 0;
        const int rc = t->det(t->det_user, side, rect, cur, n, cap, seg_d, kp_d, &total_d, m->stream);
        if (rc != 0) return svi::fail(SVI_ERR_INVALID, "the detector callback returned %d", rc);
        if (total_d < 0 || total_d > cap) return svi::fail(SVI_ERR_INVALID, "the detector callback reported %lld key points (capacity %lld)", (long long)total_d, (long long)cap);
        hipLaunchKernelGGL(k_shift_kp, dim3((n + 3) / 4), dim3(kB), 0, m->stream, t->lm.kp_size, seg_d, n, reinterpret_cast<const float2*>(kp_d),
                           reinterpret_cast<float2*>(kp_in));                                                              // :533
        SVI_TRY(extract(t, side, ext, seg_d, kp_in, n, total_d, seg_e, kp_e, desc_e));
        SVI_TRY(svi_match_ragged_dev(m, here, nullptr, cur, n, seg_e, desc_e, 50, 257, idx, dist, status));               // :540-545
        SVI_TRY(svi_track_handover_dev(m, 2 + side, rec, t->lm.kp_size, nullptr, n, seg_e, kp_e, idx, nullptr, uv_ref, topleft, ok));
        hipLaunchKernelGGL(k_after_match, grid_for(n), dim3(kB), 0, m->stream, status, ok, seg_e, idx, reinterpret_cast<const uint4*>(desc_e), n, found,
                           reinterpret_cast<uint4*>(desc_here));
        StereoOut so{};
        SVI_TRY(stereo(t, side, t->lm.kp_size, rng, desc_here, there, uv_ref, topleft, found, n, 50, 0, true, &so));       // :573 / :691
        FinishArgs fa{cur, found, status, so.st_range, so.st_verify, so.seg2, so.idx2, so.pool, reinterpret_cast<const float2*>(uv_ref), so.uv_other,
                      so.xyz, reinterpret_cast<const uint4*>(desc_here), rec, side, 2, side == 0 ? nxt : lost};
        hipLaunchKernelGGL(k_finish, grid_for(n), dim3(kB), 0, m->stream, fa, *out, n);
        SVI_HIP(hipGetLastError());
        std::swap(cur, nxt);
    }
    return SVI_OK;
}

// ---- stage 3 (:847-1030): sampling depths 0 and 2, _getMatch, _addMeasurementToLandmarkLEFT -----------------------------------
int run_stage3(svi_tracker* t, const uint8_t* run0, const svi_track_result* out)
{
    svi_matcher* m = t->m;
    const int n = t->lm.n;
    if (n == 0) return SVI_OK;
    const svi_track_record* rec = records_of(t);
    uint8_t *cur, *nxt, *found, *ok;
    float *uv_ref, *topleft, *roi, *rng;
    int32_t *seg_in, *seg_e, *idx, *dist, *status;
    uint8_t* desc_here;
    SVI_TRY(t->take(svi_tracker::kMaskA, (size_t)n, &cur));
    SVI_TRY(t->take(svi_tracker::kMaskB, (size_t)n, &nxt));
    SVI_TRY(t->take(svi_tracker::kMaskFound, (size_t)n, &found));
    SVI_TRY(t->take(svi_tracker::kOk, (size_t)n, &ok));
    SVI_TRY(t->take(svi_tracker::kUvRef, (size_t)2 * n, &uv_ref));
    SVI_TRY(t->take(svi_tracker::kTopLeft, (size_t)2 * n, &topleft));
    SVI_TRY(t->take(svi_tracker::kRoi, (size_t)4 * n, &roi));
    SVI_TRY(t->take(svi_tracker::kSegIn, (size_t)n + 1, &seg_in));
    SVI_TRY(t->take(svi_tracker::kSegE, (size_t)n + 1, &seg_e));
    SVI_TRY(t->take(svi_tracker::kIdx, (size_t)n, &idx));
    SVI_TRY(t->take(svi_tracker::kDist, (size_t)n, &dist));
    SVI_TRY(t->take(svi_tracker::kStatus, (size_t)n, &status));
    SVI_TRY(t->take(svi_tracker::kDescHere, (size_t)32 * n, &desc_here));
    SVI_TRY(t->take(svi_tracker::kRect, (size_t)n, &rng));
    hipLaunchKernelGGL(k_search_range, grid_for(n), dim3(kB), 0, m->stream, rec, n, rng);
    SVI_HIP(hipMemcpyAsync(cur, run0, (size_t)n, hipMemcpyDeviceToDevice, m->stream));
    for (int depth = 0; depth <= 2; depth += 2) {               // m_uRecursionLimitEpipolarLines = 2, step 2 (CFundamentalMatcher.h:84-85)
        hipLaunchKernelGGL(k_s3_counts, grid_for(n), dim3(kB), 0, m->stream, rec, cur, n, seg_in);
        hipLaunchKernelGGL(k_scan, dim3(1), dim3(1024), 0, m->stream, seg_in, n);
        // the samples of the rows that run are at most the samples of the plan: sized without asking the device
        const int64_t total = t->total_samples;
        if (total == 0) break;
        float *samples, *kp_e;
        uint8_t* desc_e;
        SVI_TRY(t->take(svi_tracker::kKpIn, (size_t)2 * total, &samples));
        SVI_TRY(t->take(svi_tracker::kKpE, (size_t)2 * total, &kp_e));
        SVI_TRY(t->take(svi_tracker::kDescE, (size_t)32 * total, &desc_e));
        SVI_TRY(svi_track_epipolar_samples_dev(m, &t->cam, rec, t->lm.kp_size, nullptr, n, seg_in, depth, samples, roi));
        SVI_TRY(extract(t, 0, roi, seg_in, samples, n, total, seg_e, kp_e, desc_e));
        SVI_TRY(svi_match_ragged_dev(m, t->lm.last_desc_left, t->lm.ref_desc_left, cur, n, seg_e, desc_e, 50, 100, idx, dist, status));   // _getMatch
        SVI_TRY(svi_track_handover_dev(m, 4, rec, t->lm.kp_size, nullptr, n, seg_e, kp_e, idx, roi, uv_ref, topleft, ok));
        hipLaunchKernelGGL(k_after_match, grid_for(n), dim3(kB), 0, m->stream, status, nullptr, seg_e, idx, reinterpret_cast<const uint4*>(desc_e), n,
                           found, reinterpret_cast<uint4*>(desc_here));
        StereoOut so{};
        SVI_TRY(stereo(t, 0, t->lm.kp_size, rng, desc_here, nullptr, uv_ref, topleft, found, n, -1, 0, true, &so));
        // only an internal "no match" recurses (:2221-2233); a stereo failure after a match is final: next = run && !found
        FinishArgs fa{cur, found, status, so.st_range, so.st_verify, so.seg2, so.idx2, so.pool, reinterpret_cast<const float2*>(uv_ref), so.uv_other,
                      so.xyz, reinterpret_cast<const uint4*>(desc_here), rec, 0, 3, nullptr};
        hipLaunchKernelGGL(k_finish, grid_for(n), dim3(kB), 0, m->stream, fa, *out, n);
        hipLaunchKernelGGL(k_lost, grid_for(n), dim3(kB), 0, m->stream, cur, status, n, nxt);   // status here = the temporal match's
        SVI_HIP(hipGetLastError());
        std::swap(cur, nxt);
    }
    return SVI_OK;
}

int make_mask(svi_tracker* t, const uint8_t* active, int mode, int slot, uint8_t** out)
{
    const int n = t->lm.n;
    SVI_TRY(t->take(slot, (size_t)n, out));
    if (n > 0) hipLaunchKernelGGL(k_mask, grid_for(n), dim3(kB), 0, t->m->stream, records_of(t), active, mode, n, *out);
    SVI_HIP(hipGetLastError());
    return SVI_OK;
}

} // namespace

extern "C" {

int svi_tracker_create(svi_matcher* m, const svi_track_camera* cam, svi_tracker** out)
{
    if (!m || !cam || !out) return svi::fail(SVI_ERR_INVALID, "svi_tracker_create: null argument");
    if (!(cam->width > 0.0) || !(cam->height > 0.0) || !(cam->P_left[0] > 0.0)) return svi::fail(SVI_ERR_INVALID, "svi_tracker_create: bad camera");
    svi_tracker* t = new svi_tracker();
    t->m = m;
    t->cam = *cam;
    *out = t;
    return SVI_OK;
}

int svi_tracker_destroy(svi_tracker* t)
{
    if (!t) return SVI_OK;
    (void)hipSetDevice(t->m->device);
    (void)hipStreamSynchronize(t->m->stream);
    for (auto& b : t->ws) b.release();
    delete t;
    return SVI_OK;
}

int svi_tracker_set_brief(svi_tracker* t, svi_brief* b)
{
    if (!t) return svi::fail(SVI_ERR_INVALID, "null tracker");
    t->brief = b;
    t->ext = nullptr; t->ext_user = nullptr;
    return SVI_OK;
}

int svi_tracker_set_extractor(svi_tracker* t, svi_extract_fn fn, void* user)
{
    if (!t) return svi::fail(SVI_ERR_INVALID, "null tracker");
    t->ext = fn; t->ext_user = user;
    return SVI_OK;
}

int svi_tracker_set_detector(svi_tracker* t, svi_detect_fn fn, void* user, int64_t capacity)
{
    if (!t) return svi::fail(SVI_ERR_INVALID, "null tracker");
    if (fn && capacity < 1) return svi::fail(SVI_ERR_INVALID, "svi_tracker_set_detector: capacity must be positive");
    t->det = fn; t->det_user = user; t->det_cap = capacity;
    return SVI_OK;
}

int svi_tracker_plan(svi_tracker* t, const double* T_world_to_left, const double* dp_T_left_to_world, int n_dp, double motion_scaling,
                     const svi_track_landmarks* lm)
{
    if (!t || !lm) return svi::fail(SVI_ERR_INVALID, "svi_tracker_plan: null argument");
    if (lm->n < 0) return svi::fail(SVI_ERR_INVALID, "svi_tracker_plan: negative landmark count");
    t->planned = false;
    svi_track_record* rec = nullptr;
    int32_t* seg = nullptr;
    SVI_HIP(svi::enter_device(t->m->device));
    SVI_TRY(t->take(svi_tracker::kRecords, (size_t)lm->n, &rec));
    SVI_TRY(t->take(svi_tracker::kS3Seg, (size_t)lm->n + 1, &seg));
    SVI_TRY(svi_track_plan_dev(t->m, &t->cam, T_world_to_left, dp_T_left_to_world, n_dp, motion_scaling, lm->xyz_world, lm->kp_size, lm->last_disparity,
                               lm->uv_reference, lm->dp_index, lm->n, rec, seg, nullptr));
    // one wait for both numbers the host needs: the samples of the frame and the landmarks without an epipolar line
    int32_t* cnt = nullptr;
    SVI_TRY(t->take(svi_tracker::kIdx2, 2, &cnt));
    SVI_HIP(hipMemsetAsync(cnt, 0, 2 * sizeof(int32_t), t->m->stream));
    hipLaunchKernelGGL(k_count_no_motion, grid_for(std::max(lm->n, 1)), dim3(kB), 0, t->m->stream, rec, lm->n, seg, cnt);
    int32_t h[2] = {0, 0};
    SVI_HIP(hipMemcpyAsync(h, cnt, sizeof(h), hipMemcpyDeviceToHost, t->m->stream));
    SVI_HIP(hipStreamSynchronize(t->m->stream));
    t->lm = *lm;
    t->total_samples = h[0];
    t->n_no_motion = h[1];
    t->planned = true;
    return SVI_OK;
}

// the descriptors of the frame's landmarks may be (re)bound after the plan: CLandmark::getLastDescriptorLEFT / RIGHT and
// matDescriptorReferenceLEFT change with every measurement, the geometry of the plan does not
int svi_tracker_set_descriptors(svi_tracker* t, const uint8_t* last_desc_left, const uint8_t* last_desc_right, const uint8_t* ref_desc_left)
{
    if (!t) return svi::fail(SVI_ERR_INVALID, "null tracker");
    t->lm.last_desc_left = last_desc_left; t->lm.last_desc_right = last_desc_right; t->lm.ref_desc_left = ref_desc_left;
    return SVI_OK;
}

int svi_tracker_records(svi_tracker* t, const svi_track_record** records, const int32_t** s3_seg, int64_t* total_samples)
{
    if (!t) return svi::fail(SVI_ERR_INVALID, "null tracker");
    if (!t->planned) return svi::fail(SVI_ERR_STATE, "svi_tracker_plan has not been called for this frame");
    if (records) *records = records_of(t);
    if (s3_seg) *s3_seg = t->ws[svi_tracker::kS3Seg].as<int32_t>();
    if (total_samples) *total_samples = t->total_samples;
    return SVI_OK;
}

int svi_track_stage1(svi_tracker* t, const uint8_t* active, const svi_track_result* out)
{
    SVI_TRY(check_frame(t, false));
    SVI_TRY(check_result(out));
    SVI_HIP(svi::enter_device(t->m->device));
    init_result(t, out, t->lm.n);
    uint8_t* run = nullptr;
    SVI_TRY(make_mask(t, active, kMaskFovBoth, svi_tracker::kMaskC, &run));   // :389
    return run_stage1(t, run, out, nullptr);
}

int svi_track_stage2(svi_tracker* t, const uint8_t* active, const svi_track_result* out)
{
    SVI_TRY(check_frame(t, false));
    SVI_TRY(check_result(out));
    SVI_HIP(svi::enter_device(t->m->device));
    init_result(t, out, t->lm.n);
    uint8_t* run = nullptr;
    SVI_TRY(make_mask(t, active, kMaskFovBoth, svi_tracker::kMaskC, &run));
    return run_stage2(t, run, out, nullptr);
}

int svi_track_epipolar(svi_tracker* t, const uint8_t* active, const svi_track_result* out)
{
    SVI_TRY(check_frame(t, true));
    SVI_TRY(check_result(out));
    SVI_HIP(svi::enter_device(t->m->device));
    init_result(t, out, t->lm.n);
    uint8_t* run = nullptr;
    SVI_TRY(make_mask(t, active, kMaskEpiOk, svi_tracker::kMaskC, &run));
    SVI_TRY(run_stage3(t, run, out));
    // a detection point that has not moved has no epipolar line (:847): its landmarks are searched by stage 2 (:1026-1290)
    if (t->det && t->n_no_motion > 0) {
        SVI_TRY(make_mask(t, active, kMaskNoMotionFov, svi_tracker::kMaskC, &run));
        SVI_TRY(run_stage2(t, run, out, nullptr));
    }
    return SVI_OK;
}

int svi_track_manual(svi_tracker* t, const uint8_t* active, const svi_track_result* out)
{
    SVI_TRY(check_frame(t, true));
    SVI_TRY(check_result(out));
    SVI_HIP(svi::enter_device(t->m->device));
    const int n = t->lm.n;
    init_result(t, out, n);
    uint8_t *run = nullptr, *lost = nullptr, *run3 = nullptr;
    SVI_TRY(make_mask(t, active, kMaskFovBoth, svi_tracker::kMaskC, &run));   // outside either field of view: not tracked at all (:1415)
    SVI_TRY(t->take(svi_tracker::kPositMask, (size_t)n, &lost));
    SVI_TRY(run_stage1(t, run, out, lost));
    // stage 2 for what stage 1 lost; its own `lost` output may alias its input: the input is copied first
    SVI_TRY(run_stage2(t, lost, out, lost));
    SVI_TRY(make_mask(t, lost, kMaskEpiOk, svi_tracker::kMaskC, &run3));
    return run_stage3(t, run3, out);
}

int svi_track_pose_stereo_posit(svi_tracker* t, const uint8_t* active, const svi_posit_params* prm, const double* T_world_to_left_last,
                                const double* t_imu, const double* T_world_to_left_estimate, const svi_track_result* out, svi_posit_result* pose)
{
    SVI_TRY(check_frame(t, false));
    SVI_TRY(check_result(out));
    if (!prm || !pose) return svi::fail(SVI_ERR_INVALID, "svi_track_pose_stereo_posit: null parameters / result");
    SVI_HIP(svi::enter_device(t->m->device));
    const int n = t->lm.n;
    init_result(t, out, n);
    uint8_t *run = nullptr, *lost = nullptr;
    SVI_TRY(make_mask(t, active, kMaskFovBoth, svi_tracker::kMaskC, &run));
    SVI_TRY(t->take(svi_tracker::kPositMask, (size_t)n, &lost));
    SVI_TRY(run_stage1(t, run, out, lost));
    if (t->det) SVI_TRY(run_stage2(t, lost, out, nullptr));
    // vecMeasurementsForStereoPosit = what stage 1 / 2 found (:436-441, :486, :595, :711)
    if (n > 0) hipLaunchKernelGGL(k_status_ok, grid_for(n), dim3(kB), 0, t->m->stream, out->status, n, lost);
    SVI_HIP(hipGetLastError());
    return svi_stereo_posit_dev(t->m, prm, T_world_to_left_last, t_imu, T_world_to_left_estimate, t->lm.xyz_world, out->uv_left, out->uv_right, lost, n,
                                pose);
}

int svi_track_add_new_landmarks(svi_tracker* t, const float* uv_left, const float* kp_size, const uint8_t* desc_left, int n,
                                const svi_track_result* out)
{
    if (!t) return svi::fail(SVI_ERR_INVALID, "null tracker");
    if (n < 0) return svi::fail(SVI_ERR_INVALID, "svi_track_add_new_landmarks: n < 0");
    if (n == 0) return SVI_OK;
    SVI_TRY(check_result(out));
    if (!uv_left || !kp_size || !desc_left) return svi::fail(SVI_ERR_INVALID, "svi_track_add_new_landmarks: null array");
    if (reinterpret_cast<uintptr_t>(desc_left) & 15) return svi::fail(SVI_ERR_INVALID, "svi_track_add_new_landmarks: desc_left must be 16-byte aligned");
    svi_matcher* m = t->m;
    SVI_HIP(svi::enter_device(m->device));
    svi_track_result o = *out;
    o.stage = nullptr;
    hipLaunchKernelGGL(k_init_result, grid_for(n), dim3(kB), 0, m->stream, o, n);
    uint8_t* all = nullptr;
    float* topleft = nullptr;
    int32_t* st_ok = nullptr;
    SVI_TRY(t->take(svi_tracker::kMaskA, (size_t)n, &all));
    SVI_TRY(t->take(svi_tracker::kTopLeft, (size_t)2 * n, &topleft));
    SVI_TRY(t->take(svi_tracker::kStatus, (size_t)n, &st_ok));
    hipLaunchKernelGGL(k_ones, grid_for(n), dim3(kB), 0, m->stream, all, n);
    SVI_HIP(hipMemsetAsync(st_ok, 0, sizeof(int32_t) * (size_t)n, m->stream));   // SVI_TRK_MATCH_OK: the "temporal match" is the detection itself
    hipLaunchKernelGGL(k_new_topleft, grid_for(n), dim3(kB), 0, m->stream, reinterpret_cast<const float2*>(uv_left), kp_size, n,
                       reinterpret_cast<float2*>(topleft));
    StereoOut so{};
    SVI_TRY(stereo(t, 0, kp_size, nullptr, desc_left, nullptr, uv_left, topleft, all, n, -1, 0, false, &so));
    const int saved_n = t->lm.n;
    FinishArgs fa{all, all, st_ok, so.st_range, so.st_verify, so.seg2, so.idx2, so.pool, reinterpret_cast<const float2*>(uv_left), so.uv_other, so.xyz,
                  reinterpret_cast<const uint4*>(desc_left), nullptr, 0, 0, nullptr};
    hipLaunchKernelGGL(k_finish, grid_for(n), dim3(kB), 0, m->stream, fa, o, n);
    SVI_HIP(hipGetLastError());
    (void)saved_n;
    return SVI_OK;
}

} // extern "C"
