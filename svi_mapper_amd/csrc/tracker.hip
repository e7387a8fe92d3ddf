// tracker.hip — the temporal tracking schedule of CFundamentalMatcher as batched passes.
//
// The reference walks its active landmarks one by one and, per landmark, projects it into the new
// frame, cuts search rectangles / an epipolar segment, has OpenCV extract BRIEF descriptors there,
// matches ONE descriptor against that small pool and validates the stereo partner
//   src/core/CFundamentalMatcher.cpp:368-733   getPoseStereoPosit (stages 1 and 2)
//   src/core/CFundamentalMatcher.cpp:794-1315  trackEpipolar      (stage 3, stage 2 fallback)
//   src/core/CFundamentalMatcher.cpp:2142-2334 _getMatchSampleRecursiveU/V
//   src/core/CFundamentalMatcher.cpp:2336-2397 _getMatch
//   src/core/CFundamentalMatcher.cpp:2400-2450 _addMeasurementToLandmarkLEFT
//   src/core/CTriangulator.cpp:185-324         getPointTriangulatedInRIGHT / InLEFT
// Here every step is one launch over all landmarks of the frame:
//   k_track_fundamental  one thread per detection point   F = K^-T (R [t]x) K^-1
//   k_track_plan         one thread per landmark          projections, FoV gate, rectangles, clipped line
//   k_scan_i32           one workgroup                    ragged segment starts
//   k_track_samples      one wavefront per landmark       epipolar key points + ROI
//   k_track_handover     one thread per landmark          reference point + top-left corner of the stereo search
//   k_stereo_range / k_stereo_candidates                  row candidates of the stereo search
//   k_match_ragged<VERIFY>  one wavefront per landmark    k=1 Hamming NN inside the landmark's own pool
//                                                         segment + the reference's accept/reject chain
// All of it is latency/launch bound (a few thousand landmarks, < 1 MB of descriptors per frame); the only
// streaming traffic is the candidate pool, read exactly once with 32 B per lane.
//
// Arithmetic follows the reference expression by expression (operand order, float vs double, the
// std::round / truncation points); the file is compiled with -ffp-contract=off so that no multiply-add is
// fused and the CPU restatement (oracle/oracle_track.c) can be matched bit for bit.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>

#include "common.h"
#include "matcher_handle.h"

namespace {

constexpr uint32_t kNoDist = 257u;

struct Cam {
    double PL[12], PR[12], Kinv[9];
    double W, H;
};

// ------------------------------------------------------------------------------------------------
// fundamental matrix per detection point (CFundamentalMatcher.cpp:800-806)
//   T = T_world_to_left * T_dp_left_to_world ;  E = R [t]x ;  F = (K^-T E) K^-1
// out: F[9] row-major, then |t|^2
// ------------------------------------------------------------------------------------------------
__device__ void mat3_mul(const double* a, const double* b, double* c)
{
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) c[3 * i + j] = a[3 * i] * b[j] + a[3 * i + 1] * b[3 + j] + a[3 * i + 2] * b[6 + j];
}

__global__ void k_track_fundamental(Cam cam, const double* __restrict__ Tw2l, const double* __restrict__ dpT, int n_dp,
                                    double* __restrict__ F)
{
    const int d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= n_dp) return;
    const double* A = Tw2l;          // R (9) t (3)
    const double* B = dpT + 12 * d;
    double R[9], t[3];
    mat3_mul(A, B, R);
    for (int i = 0; i < 3; ++i) t[i] = A[3 * i] * B[9] + A[3 * i + 1] * B[10] + A[3 * i + 2] * B[11] + A[9 + i];
    const double S[9] = {0.0, -t[2], t[1], t[2], 0.0, -t[0], -t[1], t[0], 0.0}; // CMiniVisionToolbox::getSkew
    double E[9], KT[9], KE[9];
    mat3_mul(R, S, E);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) KT[3 * i + j] = cam.Kinv[3 * j + i];
    mat3_mul(KT, E, KE);
    mat3_mul(KE, cam.Kinv, F + 10 * d);
    F[10 * d + 9] = t[0] * t[0] + t[1] * t[1] + t[2] * t[2];
}

// ------------------------------------------------------------------------------------------------
// plan
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void project_rounded(const double* P, const double* x, float* uv)
{
    // m_matProjection * (x, y, z, 1) then std::round(static_cast<float>(.)) (CPinholeCamera.h:202-210)
    const double a0 = P[0] * x[0] + P[1] * x[1] + P[2] * x[2] + P[3] * 1.0;
    const double a1 = P[4] * x[0] + P[5] * x[1] + P[6] * x[2] + P[7] * 1.0;
    const double a2 = P[8] * x[0] + P[9] * x[1] + P[10] * x[2] + P[11] * 1.0;
    uv[0] = roundf(static_cast<float>(a0 / a2));
    uv[1] = roundf(static_cast<float>(a1 / a2));
}

__device__ __forceinline__ bool in_fov(const Cam& c, const float* uv)
{
    // cv::Rect(28, 28, w-56, h-56).contains (CPinholeCamera.h:61); non-finite projections are outside
    if (!(fabsf(uv[0]) < 1.0e9f) || !(fabsf(uv[1]) < 1.0e9f)) return false;
    const double u = uv[0], v = uv[1];
    return u >= 28.0 && u < c.W - 28.0 && v >= 28.0 && v < c.H - 28.0;
}

// stage-2 rectangles of one image (CFundamentalMatcher.cpp:499-506, 527-530)
__device__ __forceinline__ void stage2_rect(const float* uv, double cxP, double cyP, double W, double H, double ms, float half,
                                            float* rect, float* ext)
{
    const double wU = sqrt(fabs(static_cast<double>(uv[0]) - cxP)) / 10.0; // getPrincipalWeightU
    const double wV = sqrt(fabs(static_cast<double>(uv[1]) - cyP)) / 10.0;
    const double scaleU = round(wU + ms), scaleV = round(wV + ms);
    const double halfW = round(scaleU * 15.0), halfH = round(scaleV * 15.0); // m_uSearchBlockSizePoseOptimization
    rect[0] = static_cast<float>(fmax(static_cast<double>(uv[0]) - halfW, 0.0));
    rect[1] = static_cast<float>(fmax(static_cast<double>(uv[1]) - halfH, 0.0));
    rect[2] = static_cast<float>(fmin(static_cast<double>(uv[0]) + halfW, W));
    rect[3] = static_cast<float>(fmin(static_cast<double>(uv[1]) + halfH, H));
    const float Wf = static_cast<float>(W), Hf = static_cast<float>(H);
    ext[0] = fmaxf(rect[0] - half, 0.0f);
    ext[1] = fmaxf(rect[1] - half, 0.0f);
    ext[2] = fminf(rect[2] + half, Wf);
    ext[3] = fminf(rect[3] + half, Hf);
}

__device__ __forceinline__ double curve_u(const double* c, double v) { return -(c[1] * v + c[2]) / c[0]; } // _getCurveU :2525
__device__ __forceinline__ double curve_v(const double* c, double u) { return -(c[0] * u + c[2]) / c[1]; } // _getCurveV :2529

__global__ __launch_bounds__(256) void k_track_plan(Cam cam, const double* __restrict__ Tw2l, const double* __restrict__ F, double ms,
                                                    const double* __restrict__ xyz, const float* __restrict__ kp_size,
                                                    const float* __restrict__ last_disp, const double* __restrict__ uv_ref,
                                                    const int32_t* __restrict__ dp_index, int n, int n_dp, svi_track_record* __restrict__ rec,
                                                    int32_t* __restrict__ seg)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    svi_track_record r;
    const double x[3] = {xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]};
    for (int k = 0; k < 3; ++k) r.xyz_left[k] = Tw2l[3 * k] * x[0] + Tw2l[3 * k + 1] * x[1] + Tw2l[3 * k + 2] * x[2] + Tw2l[9 + k];
    project_rounded(cam.PL, r.xyz_left, r.uv_left);
    project_rounded(cam.PR, r.xyz_left, r.uv_right);
    int status = 0;
    const bool fovL = in_fov(cam, r.uv_left), fovR = in_fov(cam, r.uv_right);
    if (fovL) status |= SVI_TRK_FOV_LEFT;
    if (fovR) status |= SVI_TRK_FOV_RIGHT;

    const float s = kp_size[i];
    const float half = 4 * s;                                     // fKeyPointSizePixelsHalf :382
    const float scale = static_cast<float>(1.0 + ms);             // fTriangulationScale :365
    r.search_range = scale * last_disp[i];                        // :386
    r.s1_roi_left[0] = r.uv_left[0] - half;  r.s1_roi_left[1] = r.uv_left[1] - half;    // :395
    r.s1_roi_right[0] = r.uv_right[0] - half; r.s1_roi_right[1] = r.uv_right[1] - half; // :449
    stage2_rect(r.uv_left, cam.PL[2], cam.PL[6], cam.W, cam.H, ms, half, r.s2_left, r.s2_ext_left);
    stage2_rect(r.uv_right, cam.PR[2], cam.PR[6], cam.W, cam.H, ms, half, r.s2_right, r.s2_ext_right);

    // ---- stage 3: clipped epipolar segment (:847-991) ----
    r.line[0] = r.line[1] = r.line[2] = 0.0;
    r.s3_start = 0.0; r.s3_count = 0; r.s3_axis = 0;
    const int d = dp_index[i];
    const bool dp_ok = d >= 0 && d < n_dp;
    if (!dp_ok || !(0.0 < F[10 * d + 9])) {
        status |= SVI_TRK_EPI_NO_MOTION;                         // :847
    } else if (fovL) {                                           // :855 "projection out of sight" otherwise
        const double* Fd = F + 10 * d;
        const double ur = uv_ref[2 * i], vr = uv_ref[2 * i + 1];
        double c[3];
        for (int k = 0; k < 3; ++k) c[k] = Fd[3 * k] * ur + Fd[3 * k + 1] * vr + Fd[3 * k + 2] * 1.0; // :861
        r.line[0] = c[0]; r.line[1] = c[1]; r.line[2] = c[2];
        const double pu = r.uv_left[0], pv = r.uv_left[1];
        const double hl = ms * 10;                                                       // dHalfLineLength :779
        const double halfU = 15.0 + (sqrt(fabs(pu - cam.PL[2])) / 10.0) * hl;            // :858
        const double halfV = 15.0 + (sqrt(fabs(pv - cam.PL[6])) / 10.0) * hl;            // :859
        const double uMinRaw = fmax(pu - halfU, 0.0), uMaxRaw = fmin(pu + halfU, cam.W); // :865-866
        const double vMinRaw = curve_v(c, uMinRaw), vMaxRaw = curve_v(c, uMaxRaw);       // :867-868
        if ((0.0 > vMinRaw && 0.0 > vMaxRaw) || (cam.H < vMinRaw && cam.H < vMaxRaw)) {
            status |= SVI_TRK_EPI_OUT_OF_SIGHT;                                          // :880-885
        } else {
            const double vLimMin = fmax(pv - halfV, 0.0), vLimMax = fmin(pv + halfV, cam.H); // :894-895
            double uMin = uMinRaw, uMax = uMaxRaw, vMin = -1.0, vMax = -1.0;
            bool bad = false;
            if (vMinRaw < vMaxRaw) {                                                     // :905
                if (vLimMin > vMaxRaw || vLimMax < vMinRaw) bad = true;                  // :908
                else {
                    if (vLimMin > vMinRaw) { vMin = vLimMin; uMin = curve_u(c, vMin); } else vMin = vMinRaw;
                    if (vLimMax < vMaxRaw) { vMax = vLimMax; uMax = curve_u(c, vMax); } else vMax = vMaxRaw;
                }
            } else {
                if (vLimMin > vMinRaw || vLimMax < vMaxRaw) bad = true;                  // :937
                else {
                    if (vLimMin > vMaxRaw) { vMin = vLimMin; uMax = curve_u(c, vMin); } else vMin = vMaxRaw;
                    if (vLimMax < vMinRaw) { vMax = vLimMax; uMin = curve_u(c, vMax); } else vMax = vMinRaw;
                }
            }
            const double du = uMax - uMin, dv = vMax - vMin;
            // negative / non-finite extents are undefined behaviour in the reference (its asserts :957-962 are
            // compiled out); they are reported as a bad projection here
            const double cap = cam.W + cam.H;
            if (bad || !(du >= 0.0) || !(dv >= 0.0) || !(du <= cap) || !(dv <= cap)) {
                status |= SVI_TRK_EPI_BAD_PROJ;
            } else {
                const uint32_t nU = static_cast<uint32_t>(du), nV = static_cast<uint32_t>(dv); // :966-967
                if (nU == 0 && nV == 0) status |= SVI_TRK_EPI_ZERO_LENGTH;               // :970
                else {
                    status |= SVI_TRK_EPI_OK;
                    if (nV < nU) { r.s3_axis = 0; r.s3_start = uMin; r.s3_count = static_cast<int32_t>(nU); } // :981-985
                    else         { r.s3_axis = 1; r.s3_start = vMin; r.s3_count = static_cast<int32_t>(nV); } // :987-991
                }
            }
        }
    }
    r.status = status;
    rec[i] = r;
    if (seg) seg[i] = (status & SVI_TRK_EPI_OK) ? r.s3_count : 0;
}

// ------------------------------------------------------------------------------------------------
// in-place exclusive scan of counts[0..n) -> seg[0..n], one workgroup
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_scan_i32(int32_t* __restrict__ seg, int n)
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

// ------------------------------------------------------------------------------------------------
// epipolar samples (CFundamentalMatcher.cpp:2154-2227 over U, :2246-2318 over V)
// one wavefront per selected landmark
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void epi_point(const svi_track_record& r, int k, double offset, float* pt)
{
    if (r.s3_axis == 0) {
        const double dU = r.s3_start + k;
        const double dV = curve_v(r.line, dU) + offset;
        pt[0] = static_cast<float>(dU); pt[1] = static_cast<float>(dV);
    } else {
        const double dV = r.s3_start + k;
        const double dU = curve_u(r.line, dV) + offset;
        pt[0] = static_cast<float>(dU); pt[1] = static_cast<float>(dV);
    }
}

__global__ __launch_bounds__(256) void k_track_samples(float Wf, float Hf, const svi_track_record* __restrict__ rec,
                                                       const float* __restrict__ kp_size, const int32_t* __restrict__ sel, int n_sel,
                                                       const int32_t* __restrict__ seg, int depth, float* __restrict__ sample_uv,
                                                       float* __restrict__ roi)
{
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (w >= n_sel) return;
    const int i = sel ? sel[w] : w;
    const int32_t s0 = seg[w], cnt = seg[w + 1] - s0;
    if (cnt <= 0) {
        if (lane == 0) { roi[4 * w] = 0.f; roi[4 * w + 1] = 0.f; roi[4 * w + 2] = 0.f; roi[4 * w + 3] = 0.f; }
        return;
    }
    const svi_track_record r = rec[i];
    // even depth samples towards +depth, odd depth towards -depth (:2157, :2177)
    const double offset = (depth % 2 == 0) ? static_cast<double>(depth) : -static_cast<double>(depth);
    float front[2], back[2], center[2];
    epi_point(r, 0, offset, front);
    epi_point(r, cnt - 1, offset, back);
    epi_point(r, cnt / 2, offset, center);
    const float s = kp_size[i];
    const float dU = fabsf(front[0] - back[0]) + 16 * s; // :2199
    const float dV = fabsf(front[1] - back[1]) + 16 * s;
    const float u0 = fmaxf(center[0] - dU / 2, 0.0f);    // :2203
    const float v0 = fmaxf(center[1] - dV / 2, 0.0f);
    if (lane == 0) {
        roi[4 * w] = u0; roi[4 * w + 1] = v0;
        roi[4 * w + 2] = fminf(dU, Wf - u0);             // :2207
        roi[4 * w + 3] = fminf(dV, Hf - v0);
    }
    for (int k = lane; k < cnt; k += 64) {
        float pt[2];
        epi_point(r, k, offset, pt);
        sample_uv[2 * (size_t)(s0 + k)] = pt[0] - u0;    // :2214
        sample_uv[2 * (size_t)(s0 + k) + 1] = pt[1] - v0;
    }
}

// ------------------------------------------------------------------------------------------------
// stereo row candidates (CTriangulator.cpp:194-213 in RIGHT, :262-284 in LEFT)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_stereo_range(float Wf, int in_left, const float2* __restrict__ uv_ref, const float2* __restrict__ topleft,
                                                      const float* __restrict__ kp_size, const float* __restrict__ search_range,
                                                      const uint8_t* __restrict__ active, int n, int32_t* __restrict__ seg,
                                                      int32_t* __restrict__ status, float* __restrict__ roi)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int32_t cnt = 0, st = SVI_TRK_MATCH_OK;
    float rw = 0.f, rh = 0.f;
    const float2 tl = topleft[i];
    if (active && !active[i]) st = SVI_TRK_MATCH_SKIPPED;
    else {
        const float s = kp_size[i];
        const float border = 4 * s, full = 8 * s + 1;
        float c;
        if (!in_left) {
            if (uv_ref[i].x <= tl.x + border) st = SVI_TRK_MATCH_RANGE;              // :197
            c = ceilf(uv_ref[i].x - tl.x - border);                                   // :203
        } else {
            if (0 >= search_range[i]) st = SVI_TRK_MATCH_RANGE;                       // :265
            c = ceilf(fminf(search_range[i], Wf - tl.x)) + 1;                         // :271
        }
        // a pool that cannot exist (negative / non-finite size: undefined behaviour in the reference) is a range failure
        if (st == SVI_TRK_MATCH_OK && !(c >= 1.0f && c <= 65536.0f)) st = SVI_TRK_MATCH_RANGE;
        if (st == SVI_TRK_MATCH_OK) {
            cnt = static_cast<int32_t>(c);
            rw = fminf(static_cast<float>(cnt) + full, Wf - tl.x);                    // :213
            rh = full;
        }
    }
    seg[i] = cnt;
    status[i] = st;
    if (roi) { roi[4 * i] = tl.x; roi[4 * i + 1] = tl.y; roi[4 * i + 2] = rw; roi[4 * i + 3] = rh; }
}

__global__ __launch_bounds__(256) void k_stereo_candidates(int in_left, const float* __restrict__ kp_size, int n, const int32_t* __restrict__ seg,
                                                           float2* __restrict__ pool_uv)
{
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (w >= n) return;
    const int32_t s0 = seg[w], cnt = seg[w + 1] - s0;
    const float border = 4 * kp_size[w];
    for (int k = lane; k < cnt; k += 64) {
        const float kf = static_cast<float>(k);
        pool_uv[s0 + k] = make_float2(in_left ? border + kf + 1 : border + kf, border);  // :209 / :280
    }
}

// ------------------------------------------------------------------------------------------------
// hand-over to the stereo search (call sites listed in include/svi_hot.h)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_track_handover(int mode, const svi_track_record* __restrict__ rec, const float* __restrict__ kp_size,
                                                        const int32_t* __restrict__ sel, int n_sel, const int32_t* __restrict__ seg,
                                                        const float2* __restrict__ pool_uv, const int32_t* __restrict__ idx,
                                                        const float* __restrict__ roi, float2* __restrict__ uv_ref, float2* __restrict__ topleft,
                                                        uint8_t* __restrict__ ok)
{
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= n_sel) return;
    const int i = sel ? sel[w] : w;
    const svi_track_record& r = rec[i];
    const float half = 4 * kp_size[i];
    const float range = r.search_range;
    float2 ref = make_float2(0.f, 0.f), tl = make_float2(0.f, 0.f);
    bool good = true;
    if (mode == 0) {
        ref = make_float2(r.s1_roi_left[0] + half, r.s1_roi_left[1] + half);
        tl = make_float2(fmaxf(0.0f, r.s1_roi_left[0] - range), r.s1_roi_left[1]);
    } else if (mode == 1) {
        ref = make_float2(r.s1_roi_right[0] + half, r.s1_roi_right[1] + half);
        tl = make_float2(r.s1_roi_right[0], r.s1_roi_right[1]);
    } else {
        const int32_t bi = idx[w];
        if (bi < 0) good = false;
        else {
            const float2 kp = pool_uv[seg[w] + bi];
            if (mode == 4) {
                ref = make_float2(kp.x + roi[4 * w], kp.y + roi[4 * w + 1]);
                tl = make_float2(fmaxf(0.0f, ref.x - range - half), ref.y - half);
            } else {
                const float* ul = mode == 2 ? r.s2_left : r.s2_right;
                ref = make_float2(ul[0] + kp.x - half, ul[1] + kp.y - half);
                const float v = ref.y - half;
                if (!(0.0 <= v)) good = false;
                tl = make_float2(mode == 2 ? fmaxf(0.0f, ref.x - range - half) : fmaxf(0.0f, ref.x - half), v);
            }
        }
    }
    uv_ref[w] = ref;
    topleft[w] = tl;
    ok[w] = good ? 1 : 0;
}

// ------------------------------------------------------------------------------------------------
// ragged k=1 matcher: one wavefront per query, lanes stride over the query's own pool segment
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t hamming256(const uint4& a0, const uint4& a1, const uint4& b0, const uint4& b1)
{
    return __popc(a0.x ^ b0.x) + __popc(a0.y ^ b0.y) + __popc(a0.z ^ b0.z) + __popc(a0.w ^ b0.w) + __popc(a1.x ^ b1.x) +
           __popc(a1.y ^ b1.y) + __popc(a1.z ^ b1.z) + __popc(a1.w ^ b1.w);
}

struct VerifyArgs {
    double finv, cx, cy, dur, min_disp, depth_min, depth_max;
    int cutoff_other, other_inclusive, in_left;
    const uint4* last_other;
    const float2* uv_ref;
    const float2* topleft;
    const float2* pool_uv;
    float2* out_uv_other;
    double* out_xyz;
};

template <bool VERIFY>
__global__ __launch_bounds__(256) void k_match_ragged(const uint4* __restrict__ q, const uint4* __restrict__ original, const uint8_t* __restrict__ active,
                                                      int nq, const int32_t* __restrict__ seg, const uint4* __restrict__ pool, int cutoff_rel,
                                                      int cutoff_orig, int32_t* __restrict__ out_idx, int32_t* __restrict__ out_dist,
                                                      int32_t* __restrict__ out_status, VerifyArgs va)
{
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (w >= nq) return;
    int32_t st = SVI_TRK_MATCH_OK, idx = -1;
    uint32_t best_d = kNoDist;
    float2 uvo = make_float2(0.f, 0.f);
    double xyz[3] = {0.0, 0.0, 0.0};
    if (active && !active[w]) st = SVI_TRK_MATCH_SKIPPED;
    else {
        const int32_t s0 = seg[w], cnt = seg[w + 1] - s0;
        const uint4 q0 = q[2 * w], q1 = q[2 * w + 1];
        unsigned long long key = ~0ull;
        for (int k = lane; k < cnt; k += 64) {
            const uint4 t0 = pool[2 * (size_t)(s0 + k)], t1 = pool[2 * (size_t)(s0 + k) + 1];
            const unsigned long long cand = (static_cast<unsigned long long>(hamming256(q0, q1, t0, t1)) << 32) | static_cast<uint32_t>(k);
            key = cand < key ? cand : key;   // k ascending per lane: first minimum wins
        }
        for (int off = 32; off > 0; off >>= 1) {
            const unsigned long long o = __shfl_xor(key, off);
            key = o < key ? o : key;
        }
        if (cnt <= 0) st = SVI_TRK_MATCH_EMPTY_POOL;                    // :2348 / CTriangulator.cpp:216
        else {
            best_d = static_cast<uint32_t>(key >> 32);
            const int32_t bi = static_cast<int32_t>(key & 0xFFFFFFFFu);
            const uint4 w0 = pool[2 * (size_t)(s0 + bi)], w1 = pool[2 * (size_t)(s0 + bi) + 1];
            if (!(static_cast<uint32_t>(cutoff_rel) > best_d)) st = SVI_TRK_MATCH_DISTANCE;   // :2370 / CTriangulator.cpp:234
            else if (!VERIFY) {
                if (original) {
                    const uint32_t d0 = hamming256(original[2 * w], original[2 * w + 1], w0, w1);
                    if (!(static_cast<uint32_t>(cutoff_orig) > d0)) st = SVI_TRK_MATCH_ORIGINAL;  // :2372
                }
            } else {
                const float2 p = va.pool_uv[s0 + bi], tl = va.topleft[w], ref = va.uv_ref[w];
                uvo = make_float2(p.x + tl.x, p.y + tl.y);              // CTriangulator.cpp:237 / :308
                const float uL = va.in_left ? uvo.x : ref.x, vL = va.in_left ? uvo.y : ref.y;
                const float uR = va.in_left ? ref.x : uvo.x;
                const float disparity = uL - uR;                        // getPointInLEFT :329-347
                if (static_cast<double>(disparity) < va.min_disp) st = SVI_TRK_MATCH_DISPARITY;
                else {
                    const double z = va.dur / static_cast<double>(disparity);
                    const double fz = va.finv * z;
                    xyz[0] = fz * (static_cast<double>(uL) - va.cx);
                    xyz[1] = fz * (static_cast<double>(vL) - va.cy);
                    xyz[2] = z;
                    if (va.depth_min > z || va.depth_max < z) st = SVI_TRK_MATCH_DEPTH;       // :416
                    else if (va.cutoff_other >= 0) {
                        const uint32_t d1 = hamming256(va.last_other[2 * w], va.last_other[2 * w + 1], w0, w1);
                        const bool ok = va.other_inclusive ? !(static_cast<uint32_t>(va.cutoff_other) < d1)   // :423
                                                           : static_cast<uint32_t>(va.cutoff_other) > d1;     // :573
                        if (!ok) st = SVI_TRK_MATCH_OTHER_MISMATCH;
                    }
                }
            }
            if (st == SVI_TRK_MATCH_OK) idx = bi;
        }
    }
    if (lane == 0) {
        out_idx[w] = idx;
        out_dist[w] = static_cast<int32_t>(best_d);
        out_status[w] = st;
        if (VERIFY) {
            va.out_uv_other[w] = uvo;
            va.out_xyz[3 * w] = xyz[0]; va.out_xyz[3 * w + 1] = xyz[1]; va.out_xyz[3 * w + 2] = xyz[2];
        }
    }
}

Cam make_cam(const svi_track_camera* c)
{
    Cam k;
    for (int i = 0; i < 12; ++i) { k.PL[i] = c->P_left[i]; k.PR[i] = c->P_right[i]; }
    for (int i = 0; i < 9; ++i) k.Kinv[i] = c->K_inv[i];
    k.W = c->width; k.H = c->height;
    return k;
}

int finish_total(svi_matcher* m, const int32_t* seg, int n, int64_t* total)
{
    if (!total) return SVI_OK;
    int32_t t = 0;
    SVI_HIP(hipMemcpyAsync(&t, seg + n, sizeof(int32_t), hipMemcpyDeviceToHost, m->stream));
    SVI_HIP(hipStreamSynchronize(m->stream));
    *total = t;
    return SVI_OK;
}

} // namespace

extern "C" {

int svi_track_plan_dev(svi_matcher* m, const svi_track_camera* cam, const double* T_world_to_left, const double* dp_T_left_to_world,
                       int n_dp, double motion_scaling, const double* xyz_world, const float* kp_size, const float* last_disparity,
                       const double* uv_reference, const int32_t* dp_index, int n, svi_track_record* records, int32_t* s3_seg,
                       int64_t* total_samples)
{
    if (!m || !cam || !T_world_to_left) return svi::fail(SVI_ERR_INVALID, "svi_track_plan_dev: null handle / camera / transform");
    if (n < 0 || n_dp < 0 || (n_dp > 0 && !dp_T_left_to_world)) return svi::fail(SVI_ERR_INVALID, "svi_track_plan_dev: bad sizes");
    if (n > 0 && (!xyz_world || !kp_size || !last_disparity || !uv_reference || !dp_index || !records))
        return svi::fail(SVI_ERR_INVALID, "svi_track_plan_dev: null array");
    if (total_samples && !s3_seg) return svi::fail(SVI_ERR_INVALID, "svi_track_plan_dev: total_samples needs s3_seg");
    SVI_HIP(svi::enter_device(m->device));
    // device staging: [T (12)] [dp_T (12 n_dp)] [F (10 n_dp)]
    const size_t nd = 12 + 12 * (size_t)n_dp + 10 * (size_t)n_dp;
    // the previous call's kernels may still read the buffer: drain before it can be re-allocated
    if (m->track.cap < nd * sizeof(double)) SVI_HIP(hipStreamSynchronize(m->stream));
    if (int rc = m->track.reserve(nd * sizeof(double))) return rc;
    double* dT = m->track.as<double>();
    double* dDp = dT + 12;
    double* dF = dDp + 12 * (size_t)n_dp;
    // the caller's host arrays are copied into the handle first, so they may be freed on return
    if (!m->track_ev) SVI_HIP(hipEventCreateWithFlags(&m->track_ev, hipEventDisableTiming));
    else SVI_HIP(hipEventSynchronize(m->track_ev));
    m->track_host.assign(T_world_to_left, T_world_to_left + 12);
    if (n_dp > 0) m->track_host.insert(m->track_host.end(), dp_T_left_to_world, dp_T_left_to_world + 12 * (size_t)n_dp);
    SVI_HIP(hipMemcpyAsync(dT, m->track_host.data(), m->track_host.size() * sizeof(double), hipMemcpyHostToDevice, m->stream));
    SVI_HIP(hipEventRecord(m->track_ev, m->stream));
    if (n_dp > 0) hipLaunchKernelGGL(k_track_fundamental, dim3((n_dp + 63) / 64), dim3(64), 0, m->stream, make_cam(cam), dT, dDp, n_dp, dF);
    if (n > 0)
        hipLaunchKernelGGL(k_track_plan, dim3((n + 255) / 256), dim3(256), 0, m->stream, make_cam(cam), dT, dF, motion_scaling, xyz_world,
                           kp_size, last_disparity, uv_reference, dp_index, n, n_dp, records, s3_seg);
    if (s3_seg) hipLaunchKernelGGL(k_scan_i32, dim3(1), dim3(1024), 0, m->stream, s3_seg, n);
    SVI_HIP(hipGetLastError());
    return s3_seg ? finish_total(m, s3_seg, n, total_samples) : SVI_OK;
}

int svi_track_epipolar_samples_dev(svi_matcher* m, const svi_track_camera* cam, const svi_track_record* records, const float* kp_size,
                                   const int32_t* sel, int n_sel, const int32_t* seg, int depth, float* sample_uv, float* roi)
{
    if (!m || !cam) return svi::fail(SVI_ERR_INVALID, "svi_track_epipolar_samples_dev: null handle / camera");
    if (n_sel < 0 || depth < 0 || depth > 127) return svi::fail(SVI_ERR_INVALID, "svi_track_epipolar_samples_dev: bad n_sel / depth");
    if (n_sel == 0) return SVI_OK;
    if (!records || !kp_size || !seg || !sample_uv || !roi) return svi::fail(SVI_ERR_INVALID, "svi_track_epipolar_samples_dev: null array");
    SVI_HIP(svi::enter_device(m->device));
    hipLaunchKernelGGL(k_track_samples, dim3((n_sel + 3) / 4), dim3(256), 0, m->stream, static_cast<float>(cam->width),
                       static_cast<float>(cam->height), records, kp_size, sel, n_sel, seg, depth, sample_uv, roi);
    SVI_HIP(hipGetLastError());
    return SVI_OK;
}

int svi_track_handover_dev(svi_matcher* m, int mode, const svi_track_record* records, const float* kp_size, const int32_t* sel, int n_sel,
                           const int32_t* seg, const float* pool_uv, const int32_t* idx, const float* roi, float* uv_ref, float* topleft,
                           uint8_t* ok)
{
    if (!m) return svi::fail(SVI_ERR_INVALID, "null matcher");
    if (mode < 0 || mode > 4 || n_sel < 0) return svi::fail(SVI_ERR_INVALID, "svi_track_handover_dev: bad mode / n_sel");
    if (n_sel == 0) return SVI_OK;
    if (!records || !kp_size || !uv_ref || !topleft || !ok) return svi::fail(SVI_ERR_INVALID, "svi_track_handover_dev: null array");
    if (mode >= 2 && (!seg || !pool_uv || !idx)) return svi::fail(SVI_ERR_INVALID, "svi_track_handover_dev: mode %d needs seg / pool_uv / idx", mode);
    if (mode == 4 && !roi) return svi::fail(SVI_ERR_INVALID, "svi_track_handover_dev: mode 4 needs roi");
    SVI_HIP(svi::enter_device(m->device));
    hipLaunchKernelGGL(k_track_handover, dim3((n_sel + 255) / 256), dim3(256), 0, m->stream, mode, records, kp_size, sel, n_sel, seg,
                       reinterpret_cast<const float2*>(pool_uv), idx, roi, reinterpret_cast<float2*>(uv_ref), reinterpret_cast<float2*>(topleft), ok);
    SVI_HIP(hipGetLastError());
    return SVI_OK;
}

int svi_track_stereo_range_dev(svi_matcher* m, double width, int search_in_left, const float* uv_ref, const float* topleft,
                               const float* kp_size, const float* search_range, const uint8_t* active, int n, int32_t* seg,
                               int32_t* out_status, float* roi, int64_t* total)
{
    if (!m) return svi::fail(SVI_ERR_INVALID, "null matcher");
    if (n < 0 || !seg) return svi::fail(SVI_ERR_INVALID, "svi_track_stereo_range_dev: bad n / seg");
    if (n > 0 && (!uv_ref || !topleft || !kp_size || !out_status || (search_in_left && !search_range)))
        return svi::fail(SVI_ERR_INVALID, "svi_track_stereo_range_dev: null array");
    SVI_HIP(svi::enter_device(m->device));
    if (n > 0)
        hipLaunchKernelGGL(k_stereo_range, dim3((n + 255) / 256), dim3(256), 0, m->stream, static_cast<float>(width), search_in_left,
                           reinterpret_cast<const float2*>(uv_ref), reinterpret_cast<const float2*>(topleft), kp_size, search_range, active, n,
                           seg, out_status, roi);
    hipLaunchKernelGGL(k_scan_i32, dim3(1), dim3(1024), 0, m->stream, seg, n);
    SVI_HIP(hipGetLastError());
    return finish_total(m, seg, n, total);
}

int svi_track_stereo_candidates_dev(svi_matcher* m, int search_in_left, const float* kp_size, int n, const int32_t* seg, float* pool_uv)
{
    if (!m) return svi::fail(SVI_ERR_INVALID, "null matcher");
    if (n < 0) return svi::fail(SVI_ERR_INVALID, "svi_track_stereo_candidates_dev: n < 0");
    if (n == 0) return SVI_OK;
    if (!kp_size || !seg || !pool_uv) return svi::fail(SVI_ERR_INVALID, "svi_track_stereo_candidates_dev: null array");
    SVI_HIP(svi::enter_device(m->device));
    hipLaunchKernelGGL(k_stereo_candidates, dim3((n + 3) / 4), dim3(256), 0, m->stream, search_in_left, kp_size, n, seg,
                       reinterpret_cast<float2*>(pool_uv));
    SVI_HIP(hipGetLastError());
    return SVI_OK;
}

int svi_match_ragged_dev(svi_matcher* m, const uint8_t* q, const uint8_t* original, const uint8_t* active, int nq, const int32_t* seg,
                         const uint8_t* pool, int cutoff_relative, int cutoff_original, int32_t* out_idx, int32_t* out_dist,
                         int32_t* out_status)
{
    if (!m) return svi::fail(SVI_ERR_INVALID, "null matcher");
    if (nq < 0) return svi::fail(SVI_ERR_INVALID, "svi_match_ragged_dev: nq < 0");
    if (nq == 0) return SVI_OK;
    if (!q || !seg || !out_idx || !out_dist || !out_status) return svi::fail(SVI_ERR_INVALID, "svi_match_ragged_dev: null array");
    if (cutoff_relative < 0 || cutoff_original < 0) return svi::fail(SVI_ERR_INVALID, "svi_match_ragged_dev: negative cut-off");
    SVI_HIP(svi::enter_device(m->device));
    VerifyArgs va{};
    hipLaunchKernelGGL(k_match_ragged<false>, dim3((nq + 3) / 4), dim3(256), 0, m->stream, reinterpret_cast<const uint4*>(q),
                       reinterpret_cast<const uint4*>(original), active, nq, seg, reinterpret_cast<const uint4*>(pool), cutoff_relative,
                       cutoff_original, out_idx, out_dist, out_status, va);
    SVI_HIP(hipGetLastError());
    return SVI_OK;
}

int svi_track_stereo_verify_dev(svi_matcher* m, const svi_track_stereo_params* prm, const uint8_t* ref, const uint8_t* last_other,
                                const uint8_t* active, const float* uv_ref, const float* topleft, int nq, const int32_t* seg,
                                const uint8_t* pool, const float* pool_uv, int32_t* out_idx, int32_t* out_dist, int32_t* out_status,
                                float* out_uv_other, double* out_xyz)
{
    if (!m || !prm) return svi::fail(SVI_ERR_INVALID, "svi_track_stereo_verify_dev: null handle / params");
    if (nq < 0) return svi::fail(SVI_ERR_INVALID, "svi_track_stereo_verify_dev: nq < 0");
    if (nq == 0) return SVI_OK;
    if (!ref || !uv_ref || !topleft || !seg || !pool_uv || !out_idx || !out_dist || !out_status || !out_uv_other || !out_xyz)
        return svi::fail(SVI_ERR_INVALID, "svi_track_stereo_verify_dev: null array");
    if (prm->cutoff_other >= 0 && !last_other) return svi::fail(SVI_ERR_INVALID, "svi_track_stereo_verify_dev: cutoff_other needs last_other");
    if (prm->cutoff_match < 0 || !(prm->f > 0.0)) return svi::fail(SVI_ERR_INVALID, "svi_track_stereo_verify_dev: bad parameters");
    SVI_HIP(svi::enter_device(m->device));
    VerifyArgs va{};
    va.finv = 1.0 / prm->f; va.cx = prm->cx; va.cy = prm->cy; va.dur = prm->duR_flipped; va.min_disp = prm->min_disparity;
    va.depth_min = prm->depth_min; va.depth_max = prm->depth_max;
    va.cutoff_other = prm->cutoff_other; va.other_inclusive = prm->other_inclusive; va.in_left = prm->search_in_left;
    va.last_other = reinterpret_cast<const uint4*>(last_other);
    va.uv_ref = reinterpret_cast<const float2*>(uv_ref);
    va.topleft = reinterpret_cast<const float2*>(topleft);
    va.pool_uv = reinterpret_cast<const float2*>(pool_uv);
    va.out_uv_other = reinterpret_cast<float2*>(out_uv_other);
    va.out_xyz = out_xyz;
    hipLaunchKernelGGL(k_match_ragged<true>, dim3((nq + 3) / 4), dim3(256), 0, m->stream, reinterpret_cast<const uint4*>(ref), nullptr, active, nq,
                       seg, reinterpret_cast<const uint4*>(pool), prm->cutoff_match, 0, out_idx, out_dist, out_status, va);
    SVI_HIP(hipGetLastError());
    return SVI_OK;
}

} // extern "C"
