/*
 * oracle_track.c — CPU restatement of the reference's temporal tracking schedule.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under svi_mapper_amd/ may include, link or call this file;
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the checker.
 *
 * PARITY UNPINNED: the reference ships no tests or golden vectors and cannot be built here
 * (OpenCV / Eigen absent).  What follows restates, one landmark at a time and in the reference's own control
 * flow (its exceptions become early returns), the geometry and the accept/reject decisions of
 *
 *   CFundamentalMatcher::getPoseStereoPosit       src/core/CFundamentalMatcher.cpp:368-733
 *   CFundamentalMatcher::trackEpipolar            src/core/CFundamentalMatcher.cpp:794-1315
 *   CFundamentalMatcher::_getMatchSampleRecursiveU/V   :2142-2334,  _getMatch :2336-2397
 *   CFundamentalMatcher::_addMeasurementToLandmarkLEFT :2400-2450
 *   CTriangulator::getPointTriangulatedInRIGHT / InLEFT / getPointInLEFT   src/core/CTriangulator.cpp:185-356
 *   CPinholeCamera::getProjectionRounded / getPrincipalWeightU/V / m_cFieldOfView   src/vision/CPinholeCamera.h:61,202-227
 *   CMiniVisionToolbox::getSkew                   src/vision/CMiniVisionToolbox.cpp:341-352
 *
 * BRIEF extraction and GFTT detection are OpenCV's and stay outside: pools of candidate descriptors are inputs.
 * Eigen's fixed-size products are restated as plain left-to-right sums; the file is compiled with
 * -ffp-contract=off (oracle/Makefile).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct orc_track_camera {
    double P_left[12], P_right[12], K_inv[9];
    double width, height;
} orc_track_camera;

/* same layout as svi_track_record (include/svi_hot.h); tests check the size */
typedef struct orc_track_record {
    double  xyz_left[3];
    double  line[3];
    double  s3_start;
    float   uv_left[2], uv_right[2];
    float   search_range;
    float   s1_roi_left[2], s1_roi_right[2];
    float   s2_left[4], s2_right[4];
    float   s2_ext_left[4], s2_ext_right[4];
    int32_t s3_count, s3_axis, status;
} orc_track_record;

enum { FOV_LEFT = 1, FOV_RIGHT = 2, EPI_NO_MOTION = 4, EPI_OUT_OF_SIGHT = 8, EPI_BAD_PROJ = 16, EPI_ZERO_LENGTH = 32, EPI_OK = 64 };
enum { M_OK = 0, M_EMPTY_POOL = 1, M_DISTANCE = 2, M_ORIGINAL = 3, M_RANGE = 4, M_DISPARITY = 5, M_DEPTH = 6, M_OTHER = 7, M_SKIPPED = 8 };

int orc_track_record_size(void) { return (int)sizeof(orc_track_record); }

static int popcount256(const uint8_t* a, const uint8_t* b)
{
    int d = 0;
    for (int k = 0; k < 32; ++k) d += __builtin_popcount((unsigned)(a[k] ^ b[k]));
    return d;
}

static void mul33(const double* a, const double* b, double* c)
{
    for (int r = 0; r < 3; ++r)
        for (int k = 0; k < 3; ++k) {
            double s = a[3 * r + 0] * b[0 + k];
            s = s + a[3 * r + 1] * b[3 + k];
            s = s + a[3 * r + 2] * b[6 + k];
            c[3 * r + k] = s;
        }
}

/* :800-806 — F (9 doubles) and |t|^2 per detection point, 10 doubles each */
void orc_track_fundamental(const orc_track_camera* cam, const double* Tw2l, const double* dpT, int n_dp, double* F10)
{
    for (int d = 0; d < n_dp; ++d) {
        const double* B = dpT + 12 * d;
        double R[9], t[3], S[9], E[9], Kt[9], KE[9];
        mul33(Tw2l, B, R);                                   /* Isometry product: linear part ... */
        for (int r = 0; r < 3; ++r) {                        /* ... and R1 t2 + t1 */
            double s = Tw2l[3 * r] * B[9];
            s = s + Tw2l[3 * r + 1] * B[10];
            s = s + Tw2l[3 * r + 2] * B[11];
            t[r] = s + Tw2l[9 + r];
        }
        S[0] = 0.0;   S[1] = -t[2]; S[2] = t[1];             /* getSkew */
        S[3] = t[2];  S[4] = 0.0;   S[5] = -t[0];
        S[6] = -t[1]; S[7] = t[0];  S[8] = 0.0;
        mul33(R, S, E);                                      /* matEssential = R * skew(t)      :804 */
        for (int r = 0; r < 3; ++r)
            for (int k = 0; k < 3; ++k) Kt[3 * r + k] = cam->K_inv[3 * k + r];
        mul33(Kt, E, KE);                                    /* K^-T * E * K^-1, left to right  :805 */
        mul33(KE, cam->K_inv, F10 + 10 * d);
        F10[10 * d + 9] = t[0] * t[0] + t[1] * t[1] + t[2] * t[2]; /* squaredNorm :847 */
    }
}

static void projection_rounded(const double* P, const double* p, float* uv)
{
    double h[3];
    for (int r = 0; r < 3; ++r) {
        double s = P[4 * r] * p[0];
        s = s + P[4 * r + 1] * p[1];
        s = s + P[4 * r + 2] * p[2];
        h[r] = s + P[4 * r + 3] * 1.0;
    }
    uv[0] = roundf((float)(h[0] / h[2]));
    uv[1] = roundf((float)(h[1] / h[2]));
}

static int fov_contains(const orc_track_camera* cam, const float* uv)
{
    if (!isfinite(uv[0]) || !isfinite(uv[1]) || fabsf(uv[0]) >= 1.0e9f || fabsf(uv[1]) >= 1.0e9f) return 0;
    const long u = lrintf(uv[0]), v = lrintf(uv[1]);       /* Point2f -> Point2i, then Rect_<int>::contains */
    const long w = (long)cam->width, h = (long)cam->height;
    return 28 <= u && u < 28 + (w - 56) && 28 <= v && v < 28 + (h - 56);
}

static double weight(double x, double c) { return sqrt(fabs(x - c)) / 10.0; }
static double curveU(const double* c, double v) { return -(c[1] * v + c[2]) / c[0]; }
static double curveV(const double* c, double u) { return -(c[0] * u + c[2]) / c[1]; }

static void stage2(const orc_track_camera* cam, const double* P, const float* uv, double ms, float half, float* rect, float* ext)
{
    const double dScaleU = round(weight(uv[0], P[2]) + ms);                 /* :499 */
    const double dScaleV = round(weight(uv[1], P[6]) + ms);                 /* :500 */
    const double dHalfW = round(dScaleU * 15);                              /* :501 */
    const double dHalfH = round(dScaleV * 15);                              /* :502 */
    rect[0] = (float)fmax(uv[0] - dHalfW, 0.0);                             /* :505 */
    rect[1] = (float)fmax(uv[1] - dHalfH, 0.0);
    rect[2] = (float)fmin(uv[0] + dHalfW, cam->width);                      /* :506 */
    rect[3] = (float)fmin(uv[1] + dHalfH, cam->height);
    ext[0] = fmaxf(rect[0] - half, 0.0f);                                   /* :527-530 */
    ext[1] = fmaxf(rect[1] - half, 0.0f);
    ext[2] = fminf(rect[2] + half, (float)cam->width);
    ext[3] = fminf(rect[3] + half, (float)cam->height);
}

/* the epipolar branch of trackEpipolar for one landmark; returns the status bits it adds */
static int epipolar(const orc_track_camera* cam, const double* F, double ms, const double* uvref, orc_track_record* r)
{
    if (!(0.0 < F[9])) return EPI_NO_MOTION;                                /* :847 */
    if (!(r->status & FOV_LEFT)) return 0;                                  /* :855 */
    double* c = r->line;
    for (int k = 0; k < 3; ++k) {
        double s = F[3 * k] * uvref[0];
        s = s + F[3 * k + 1] * uvref[1];
        c[k] = s + F[3 * k + 2] * 1.0;                                      /* :861 */
    }
    const double pu = r->uv_left[0], pv = r->uv_left[1];
    const double dHalfLineLength = ms * 10;                                 /* :779 */
    const double hU = 15.0 + weight(pu, cam->P_left[2]) * dHalfLineLength;  /* :858 */
    const double hV = 15.0 + weight(pv, cam->P_left[6]) * dHalfLineLength;  /* :859 */
    const double uMinRaw = fmax(pu - hU, 0.0);
    const double uMaxRaw = fmin(pu + hU, cam->width);
    const double vMinRaw = curveV(c, uMinRaw);
    const double vMaxRaw = curveV(c, uMaxRaw);
    if ((0.0 > vMinRaw && 0.0 > vMaxRaw) || (cam->height < vMinRaw && cam->height < vMaxRaw)) return EPI_OUT_OF_SIGHT;
    double uMin = uMinRaw, uMax = uMaxRaw, vMin = -1.0, vMax = -1.0;
    const double vLimMin = fmax(pv - hV, 0.0);
    const double vLimMax = fmin(pv + hV, cam->height);
    if (vMinRaw < vMaxRaw) {
        if (vLimMin > vMaxRaw || vLimMax < vMinRaw) return EPI_BAD_PROJ;    /* :908 */
        if (vLimMin > vMinRaw) { vMin = vLimMin; uMin = curveU(c, vMin); } else { vMin = vMinRaw; }
        if (vLimMax < vMaxRaw) { vMax = vLimMax; uMax = curveU(c, vMax); } else { vMax = vMaxRaw; }
    } else {
        if (vLimMin > vMinRaw || vLimMax < vMaxRaw) return EPI_BAD_PROJ;    /* :937 */
        if (vLimMin > vMaxRaw) { vMin = vLimMin; uMax = curveU(c, vMin); } else { vMin = vMaxRaw; }
        if (vLimMax < vMinRaw) { vMax = vLimMax; uMin = curveU(c, vMax); } else { vMax = vMinRaw; }
    }
    const double du = uMax - uMin, dv = vMax - vMin;
    /* the reference converts these to uint32_t unchecked (asserts compiled out): negative / NaN extents are
     * undefined there and classified as a bad projection here */
    if (!(du >= 0.0 && dv >= 0.0 && du <= cam->width + cam->height && dv <= cam->width + cam->height)) return EPI_BAD_PROJ;
    const uint32_t uDeltaU = (uint32_t)du, uDeltaV = (uint32_t)dv;          /* :966-967 */
    if (0 == uDeltaU && 0 == uDeltaV) return EPI_ZERO_LENGTH;               /* :970 */
    if (uDeltaV < uDeltaU) { r->s3_axis = 0; r->s3_start = uMin; r->s3_count = (int32_t)uDeltaU; }
    else                   { r->s3_axis = 1; r->s3_start = vMin; r->s3_count = (int32_t)uDeltaV; }
    return EPI_OK;
}

void orc_track_plan(const orc_track_camera* cam, const double* Tw2l, const double* dpT, int n_dp, double ms, const double* xyz,
                    const float* kp_size, const float* last_disp, const double* uv_ref, const int32_t* dp_index, int n,
                    orc_track_record* rec, int32_t* seg)
{
    double* F10 = NULL;
    double Fbuf[10 * 64];
    double* Fheap = NULL;
    if (n_dp > 64) { Fheap = (double*)malloc(sizeof(double) * 10 * (size_t)n_dp); F10 = Fheap; }
    else F10 = Fbuf;
    orc_track_fundamental(cam, Tw2l, dpT, n_dp, F10);
    int32_t run = 0;
    for (int i = 0; i < n; ++i) {
        orc_track_record* r = &rec[i];
        memset(r, 0, sizeof(*r));
        for (int k = 0; k < 3; ++k) {                                       /* Isometry3d * Vector3d :377 */
            double s = Tw2l[3 * k] * xyz[3 * i];
            s = s + Tw2l[3 * k + 1] * xyz[3 * i + 1];
            s = s + Tw2l[3 * k + 2] * xyz[3 * i + 2];
            r->xyz_left[k] = s + Tw2l[9 + k];
        }
        projection_rounded(cam->P_left, r->xyz_left, r->uv_left);           /* :378 */
        projection_rounded(cam->P_right, r->xyz_left, r->uv_right);         /* :379 */
        if (fov_contains(cam, r->uv_left)) r->status |= FOV_LEFT;
        if (fov_contains(cam, r->uv_right)) r->status |= FOV_RIGHT;
        const float fSize = kp_size[i];
        const float fHalf = 4 * fSize;                                      /* :382 */
        const float fScale = (float)(1.0 + ms);                             /* :365 */
        r->search_range = fScale * last_disp[i];                            /* :386 */
        r->s1_roi_left[0] = r->uv_left[0] - fHalf;   r->s1_roi_left[1] = r->uv_left[1] - fHalf;   /* :395 */
        r->s1_roi_right[0] = r->uv_right[0] - fHalf; r->s1_roi_right[1] = r->uv_right[1] - fHalf; /* :449 */
        stage2(cam, cam->P_left, r->uv_left, ms, fHalf, r->s2_left, r->s2_ext_left);
        stage2(cam, cam->P_right, r->uv_right, ms, fHalf, r->s2_right, r->s2_ext_right);
        const int d = dp_index[i];
        if (d < 0 || d >= n_dp) r->status |= EPI_NO_MOTION;
        else r->status |= epipolar(cam, F10 + 10 * d, ms, uv_ref + 2 * i, r);
        if (seg) { seg[i] = run; run += (r->status & EPI_OK) ? r->s3_count : 0; }
    }
    if (seg) seg[n] = run;
    free(Fheap);
}

/* _getMatchSampleRecursiveU / V up to the _getMatch call: key points (ROI coordinates) and the ROI */
void orc_track_epipolar_samples(const orc_track_camera* cam, const orc_track_record* rec, const float* kp_size, const int32_t* sel,
                                int n_sel, const int32_t* seg, int depth, float* sample_uv, float* roi)
{
    for (int w = 0; w < n_sel; ++w) {
        const int i = sel ? sel[w] : w;
        const int cnt = seg[w + 1] - seg[w];
        float* out = sample_uv + 2 * (size_t)seg[w];
        roi[4 * w] = roi[4 * w + 1] = roi[4 * w + 2] = roi[4 * w + 3] = 0.0f;
        if (cnt <= 0) continue;
        const orc_track_record* r = &rec[i];
        const int8_t iSamplingOffset = (0 == depth % 2) ? (int8_t)depth : (int8_t)(-depth);   /* :2157 / :2177 */
        for (int k = 0; k < cnt; ++k) {
            double dU, dV;
            if (r->s3_axis == 0) { dU = r->s3_start + k; dV = curveV(r->line, dU) + iSamplingOffset; }  /* :2163-2164 */
            else                 { dV = r->s3_start + k; dU = curveU(r->line, dV) + iSamplingOffset; }  /* :2255-2256 */
            out[2 * k] = (float)dU;                                        /* cv::KeyPoint( dU, dV, size ) */
            out[2 * k + 1] = (float)dV;
        }
        const float* front = out;
        const float* back = out + 2 * (cnt - 1);
        const float cx = out[2 * (cnt / 2)], cy = out[2 * (cnt / 2) + 1];   /* :2196 */
        const float fDeltaU = fabsf(front[0] - back[0]) + 16 * kp_size[i];  /* :2199 */
        const float fDeltaV = fabsf(front[1] - back[1]) + 16 * kp_size[i];
        const float fUTopLeft = fmaxf(cx - fDeltaU / 2, 0.0f);              /* :2203 */
        const float fVTopLeft = fmaxf(cy - fDeltaV / 2, 0.0f);
        roi[4 * w] = fUTopLeft;
        roi[4 * w + 1] = fVTopLeft;
        roi[4 * w + 2] = fminf(fDeltaU, (float)cam->width - fUTopLeft);     /* :2207 */
        roi[4 * w + 3] = fminf(fDeltaV, (float)cam->height - fVTopLeft);
        for (int k = 0; k < cnt; ++k) { out[2 * k] -= fUTopLeft; out[2 * k + 1] -= fVTopLeft; } /* :2214 */
    }
}

/* pool sizes of getPointTriangulatedInRIGHT (:194-213) / InLEFT (:262-284) */
void orc_track_stereo_range(double width, int in_left, const float* uv_ref, const float* topleft, const float* kp_size,
                            const float* search_range, const uint8_t* active, int n, int32_t* seg, int32_t* status, float* roi)
{
    int32_t run = 0;
    const float Wf = (float)width;
    for (int i = 0; i < n; ++i) {
        int32_t cnt = 0, st = M_OK;
        float rw = 0.0f, rh = 0.0f;
        const float fUTopLeft = topleft[2 * i];
        if (active && !active[i]) st = M_SKIPPED;
        else {
            const float fBorderCenter = 4 * kp_size[i];
            const float fFullHeight = 8 * kp_size[i] + 1;
            float c = 0.0f;
            if (!in_left) {
                if (uv_ref[2 * i] <= fUTopLeft + fBorderCenter) st = M_RANGE;                  /* :197 */
                else c = ceilf(uv_ref[2 * i] - fUTopLeft - fBorderCenter);                     /* :203 */
            } else {
                if (0 >= search_range[i]) st = M_RANGE;                                        /* :265 */
                else c = ceilf(fminf(search_range[i], Wf - fUTopLeft)) + 1;                    /* :271 */
            }
            if (st == M_OK && !(c >= 1.0f && c <= 65536.0f)) st = M_RANGE;  /* size_t conversion undefined in the reference */
            if (st == M_OK) {
                cnt = (int32_t)c;
                rw = fminf((float)cnt + fFullHeight, Wf - fUTopLeft);                          /* :213 */
                rh = fFullHeight;
            }
        }
        seg[i] = run;
        run += cnt;
        status[i] = st;
        if (roi) { roi[4 * i] = fUTopLeft; roi[4 * i + 1] = topleft[2 * i + 1]; roi[4 * i + 2] = rw; roi[4 * i + 3] = rh; }
    }
    seg[n] = run;
}

void orc_track_stereo_candidates(int in_left, const float* kp_size, int n, const int32_t* seg, float* pool_uv)
{
    for (int i = 0; i < n; ++i) {
        const float fBorderCenter = 4 * kp_size[i];
        for (int32_t k = 0; k < seg[i + 1] - seg[i]; ++k) {
            float* p = pool_uv + 2 * (size_t)(seg[i] + k);
            p[0] = in_left ? fBorderCenter + (float)k + 1 : fBorderCenter + (float)k;          /* :280 / :209 */
            p[1] = fBorderCenter;
        }
    }
}

/* cv::BFMatcher::match of ONE query against its pool: first minimum wins */
static int best_in_pool(const uint8_t* q, const uint8_t* pool, int cnt, int* dist)
{
    int bi = -1, bd = 1 << 30;
    for (int k = 0; k < cnt; ++k) {
        const int d = popcount256(q, pool + 32 * (size_t)k);
        if (d < bd) { bd = d; bi = k; }
    }
    *dist = bd;
    return bi;
}

/* _getMatch :2336-2397 */
void orc_match_ragged(const uint8_t* q, const uint8_t* original, const uint8_t* active, int nq, const int32_t* seg, const uint8_t* pool,
                      int cutoff_relative, int cutoff_original, int32_t* out_idx, int32_t* out_dist, int32_t* out_status)
{
    for (int i = 0; i < nq; ++i) {
        out_idx[i] = -1; out_dist[i] = 257;
        if (active && !active[i]) { out_status[i] = M_SKIPPED; continue; }
        const int cnt = seg[i + 1] - seg[i];
        if (cnt <= 0) { out_status[i] = M_EMPTY_POOL; continue; }                 /* :2348 */
        const uint8_t* p = pool + 32 * (size_t)seg[i];
        int d;
        const int bi = best_in_pool(q + 32 * (size_t)i, p, cnt, &d);
        out_dist[i] = d;
        if (!((double)cutoff_relative > (double)d)) { out_status[i] = M_DISTANCE; continue; }      /* :2370 */
        if (original) {
            const double dToOriginal = popcount256(original + 32 * (size_t)i, p + 32 * (size_t)bi);
            if (!((double)cutoff_original > dToOriginal)) { out_status[i] = M_ORIGINAL; continue; } /* :2372 */
        }
        out_idx[i] = bi;
        out_status[i] = M_OK;
    }
}

typedef struct orc_track_stereo_params {
    double f, cx, cy, duR_flipped, min_disparity, depth_min, depth_max;
    int cutoff_match, cutoff_other, other_inclusive, search_in_left;
} orc_track_stereo_params;

/* getPointTriangulatedInRIGHT / InLEFT after the extractor, then the caller's depth and descriptor checks */
void orc_track_stereo_verify(const orc_track_stereo_params* prm, const uint8_t* ref, const uint8_t* last_other, const uint8_t* active,
                             const float* uv_ref, const float* topleft, int nq, const int32_t* seg, const uint8_t* pool, const float* pool_uv,
                             int32_t* out_idx, int32_t* out_dist, int32_t* out_status, float* out_uv_other, double* out_xyz)
{
    const double dFInverse = 1.0 / prm->f;
    for (int i = 0; i < nq; ++i) {
        out_idx[i] = -1; out_dist[i] = 257;
        out_uv_other[2 * i] = out_uv_other[2 * i + 1] = 0.0f;
        out_xyz[3 * i] = out_xyz[3 * i + 1] = out_xyz[3 * i + 2] = 0.0;
        if (active && !active[i]) { out_status[i] = M_SKIPPED; continue; }
        const int cnt = seg[i + 1] - seg[i];
        if (cnt <= 0) { out_status[i] = M_EMPTY_POOL; continue; }                 /* CTriangulator.cpp:216 */
        const uint8_t* p = pool + 32 * (size_t)seg[i];
        int d;
        const int bi = best_in_pool(ref + 32 * (size_t)i, p, cnt, &d);
        out_dist[i] = d;
        if (!((float)prm->cutoff_match > (float)d)) { out_status[i] = M_DISTANCE; continue; }      /* :234 */
        const float* kp = pool_uv + 2 * (size_t)(seg[i] + bi);
        const float uo = kp[0] + topleft[2 * i], vo = kp[1] + topleft[2 * i + 1];                  /* :237 */
        out_uv_other[2 * i] = uo; out_uv_other[2 * i + 1] = vo;
        float uL, vL, uR;
        if (prm->search_in_left) { uL = uo; vL = vo; uR = uv_ref[2 * i]; }
        else                     { uL = uv_ref[2 * i]; vL = uv_ref[2 * i + 1]; uR = uo; }
        if (uL - uR < prm->min_disparity) { out_status[i] = M_DISPARITY; continue; }               /* :329 */
        const double dZ = prm->duR_flipped / (uL - uR);                                            /* :340 */
        out_xyz[3 * i] = dFInverse * dZ * (uL - prm->cx);                                          /* :346 */
        out_xyz[3 * i + 1] = dFInverse * dZ * (vL - prm->cy);
        out_xyz[3 * i + 2] = dZ;
        if (prm->depth_min > dZ || prm->depth_max < dZ) { out_status[i] = M_DEPTH; continue; }      /* CFundamentalMatcher.cpp:416 */
        if (prm->cutoff_other >= 0) {
            const double dn = popcount256(last_other + 32 * (size_t)i, p + 32 * (size_t)bi);
            if (prm->other_inclusive) { if ((double)prm->cutoff_other < dn) { out_status[i] = M_OTHER; continue; } }     /* :423 */
            else                      { if (!((double)prm->cutoff_other > dn)) { out_status[i] = M_OTHER; continue; } }  /* :573 */
        }
        out_idx[i] = bi;
        out_status[i] = M_OK;
    }
}

/* what each call site hands to getPointTriangulatedInRIGHT / InLEFT (modes as in include/svi_hot.h) */
void orc_track_handover(int mode, const orc_track_record* rec, const float* kp_size, const int32_t* sel, int n_sel, const int32_t* seg,
                        const float* pool_uv, const int32_t* idx, const float* roi, float* uv_ref, float* topleft, uint8_t* ok)
{
    for (int w = 0; w < n_sel; ++w) {
        const int i = sel ? sel[w] : w;
        const orc_track_record* r = &rec[i];
        const float fKeyPointSizePixelsHalf = 4 * kp_size[i];
        const float fSearchRange = r->search_range;
        float* ref = uv_ref + 2 * w;
        float* tl = topleft + 2 * w;
        ref[0] = ref[1] = tl[0] = tl[1] = 0.0f;
        ok[w] = 1;
        if (mode == 0) {                                                              /* :407-412 */
            ref[0] = r->s1_roi_left[0] + fKeyPointSizePixelsHalf;
            ref[1] = r->s1_roi_left[1] + fKeyPointSizePixelsHalf;
            tl[0] = fmaxf(0.0f, r->s1_roi_left[0] - fSearchRange);
            tl[1] = r->s1_roi_left[1];
            continue;
        }
        if (mode == 1) {                                                              /* :460-466 */
            ref[0] = r->s1_roi_right[0] + fKeyPointSizePixelsHalf;
            ref[1] = r->s1_roi_right[1] + fKeyPointSizePixelsHalf;
            tl[0] = r->s1_roi_right[0];
            tl[1] = r->s1_roi_right[1];
            continue;
        }
        if (idx[w] < 0) { ok[w] = 0; continue; }
        const float* kp = pool_uv + 2 * (size_t)(seg[w] + idx[w]);
        if (mode == 4) {                                                              /* :2382-2383, :2413-2424 */
            ref[0] = kp[0] + roi[4 * w];
            ref[1] = kp[1] + roi[4 * w + 1];
            tl[0] = fmaxf(0.0f, ref[0] - fSearchRange - fKeyPointSizePixelsHalf);
            tl[1] = ref[1] - fKeyPointSizePixelsHalf;
            continue;
        }
        const float* ul = (mode == 2) ? r->s2_left : r->s2_right;                     /* :546-547 / :663-664 */
        ref[0] = ul[0] + kp[0] - fKeyPointSizePixelsHalf;
        ref[1] = ul[1] + kp[1] - fKeyPointSizePixelsHalf;
        const float fVReference = ref[1] - fKeyPointSizePixelsHalf;                   /* :551 */
        if (!(0.0 <= fVReference)) ok[w] = 0;                                         /* :554 */
        tl[0] = (mode == 2) ? fmaxf(0.0f, ref[0] - fSearchRange - fKeyPointSizePixelsHalf)    /* :557 */
                            : fmaxf(0.0f, ref[0] - fKeyPointSizePixelsHalf);                  /* :677 */
        tl[1] = fVReference;
    }
}
