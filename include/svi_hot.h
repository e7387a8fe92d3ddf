/*
 * svi_hot.h — C ABI of the MI355X-native hot path of svi_mapper.
 *
 * Two things live behind this boundary:
 *
 *   1. the BRIEF-256 Hamming matcher that the reference reaches through
 *      std::shared_ptr<cv::DescriptorMatcher>  (src/core/CTriangulator.cpp:12,
 *      shared with CFundamentalMatcher at src/core/CFundamentalMatcher.cpp:20), and
 *   2. the Levenberg-Marquardt bundle adjustment that the reference reaches through
 *      g2o::SparseOptimizer + BlockSolverX + LinearSolverCholmod
 *      (src/optimization/Cg2oOptimizer.cpp:83-89).
 *
 * Conventions
 *   - every entry point returns an svi_status (0 == SVI_OK); no exception crosses the boundary;
 *     the reference's "no match" exceptions (src/exceptions/CExceptionNoMatchFound.h) become
 *     idx == -1 in the output arrays, never an error code.
 *   - plain pointers and sizes only. "host" pointers are caller-owned host memory; entry points
 *     with the _dev suffix take device (HBM) pointers, enqueue on the handle's HIP stream and do
 *     NOT synchronise (call svi_*_sync).
 *   - handles are opaque, not thread-safe; distinct handles may be used from distinct threads.
 *   - the library never falls back to a CPU implementation: without a gfx950 device every
 *     compute entry point returns SVI_ERR_NO_DEVICE.
 */
#ifndef SVI_HOT_H
#define SVI_HOT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SVI_HOT_VERSION 100 /* 0.1.0 */

typedef enum svi_status {
    SVI_OK              = 0,
    SVI_ERR_INVALID     = 1, /* bad argument (null pointer, negative size, unknown id ...) */
    SVI_ERR_NO_DEVICE   = 2, /* no HIP device / not a gfx950 */
    SVI_ERR_HIP         = 3, /* a HIP runtime call failed; see svi_last_error() */
    SVI_ERR_STATE       = 4, /* call out of order (e.g. optimize before initialize) */
    SVI_ERR_UNSUPPORTED = 5, /* graph shape outside what the path supports */
    SVI_ERR_NOT_FOUND   = 6, /* vertex id not in graph */
    SVI_ERR_IO          = 7, /* .g2o file could not be read / written */
    SVI_ERR_COMM        = 8, /* the all-reduce hook reported failure */
    SVI_ERR_INTERNAL    = 9  /* a defect of the library itself was detected (e.g. a hand-over inside a launch timed out): the call failed, nothing was applied */
} svi_status;

const char* svi_status_string(int status);
/* Message of the last failure on the calling thread ("" if none). */
const char* svi_last_error(void);
int         svi_version(void);
/* Number of visible HIP devices; 0 if none (never fails). */
int         svi_device_count(void);

/* ------------------------------------------------------------------------------------------
 * Matcher  — replaces cv::BFMatcher(NORM_HAMMING)::match (k = 1) and cv::norm(a,b,NORM_HAMMING)
 * as used at CTriangulator.cpp:93,156,227,298 and CFundamentalMatcher.cpp:540,657,1080,1200,
 * 1584,1708,2356 (match) / :404,:423,:453,:473,:573,:691,:2375 (norm), plus the rectified
 * triangulation CTriangulator::getPointInLEFT (CTriangulator.cpp:326-356).
 * ---------------------------------------------------------------------------------------- */

typedef struct svi_matcher svi_matcher;

/* Epipolar gate of the batched form. Pair (i,j) is a candidate iff
 *     |t_uv[j].v - q_uv[i].v| <= v_tol   &&   q_umin[i] <= t_uv[j].u < q_umax[i]
 * The reference enumerates integer pixels of ONE row in ascending u
 * (CTriangulator.cpp:67-77, 201-211, 272-282): v_tol = 0, q_umax = u_L. */
typedef struct svi_gate {
    const float* q_uv;   /* nq x 2 (u,v) */
    const float* t_uv;   /* nt x 2 (u,v) */
    const float* q_umin; /* nq */
    const float* q_umax; /* nq */
    float        v_tol;
} svi_gate;

/* stream == NULL: the handle creates and owns a HIP stream; otherwise it borrows the caller's
 * hipStream_t (so the caller can bracket launches with its own events). */
int svi_matcher_create(int device, void* stream, svi_matcher** out);
int svi_matcher_destroy(svi_matcher* m);
int svi_matcher_sync(svi_matcher* m);
/* the hipStream_t the handle launches on */
void* svi_matcher_stream(svi_matcher* m);

/* How a gated call on a small pool runs: 0 (default) the pool is bucketed by image row in LDS and a query visits only the
 * rows its window touches; 1 the gate is evaluated as a predicate on every (query, pool) pair inside the brute-force scan
 * (what SURVEY.md 8d-ii quotes pairs/s on).  Results are identical. */
int svi_matcher_set_gate_path(svi_matcher* m, int path);

/* measured shader clock (MHz) while all CUs run integer VALU work: the denominator of the matcher's lane-op roofline
 * (n_cu x 64 lane-ops per cycle) */
int svi_debug_shader_clock_mhz(svi_matcher* m, double* mhz);

/* k=1 brute force Hamming NN over 256-bit descriptors (32 B rows, contiguous).
 *   out_idx[i]  = smallest j attaining min_j popcount(q_i ^ t_j) over gated j,
 *                 or -1 if there is no candidate or min >= max_dist_exclusive
 *                 (reference cut-offs: 100 CTriangulator.cpp:13,107; 50 CFundamentalMatcher.cpp:545;
 *                  25 :404);  pass max_dist_exclusive = 257 to disable the cut-off
 *   out_dist[i] = that minimum, or 257 when out_idx[i] == -1
 * gate may be NULL (plain NQ x NT). Host-pointer form: copies in, runs, copies out, synchronises. */
int svi_match_hamming256(svi_matcher* m, const uint8_t* q, int nq, const uint8_t* t, int nt,
                         const svi_gate* gate, int max_dist_exclusive,
                         int32_t* out_idx, int32_t* out_dist);

/* Device-resident, batched form: `batch` independent frame pairs laid out back to back
 * (q: batch*nq*32 B, t: batch*nt*32 B, gate arrays likewise, outputs batch*nq). All pointers,
 * including those inside *gate (the struct itself is read on the host), are device pointers.
 * Asynchronous on the handle's stream. */
int svi_match_hamming256_dev(svi_matcher* m, const uint8_t* q, int nq, const uint8_t* t, int nt,
                             int batch, const svi_gate* gate, int max_dist_exclusive,
                             int32_t* out_idx, int32_t* out_dist);

/* Loop-closure candidate search, the USING_BF variant: the query key frame's descriptor pool
 * against the pool of EVERY past key frame, k = 1 per (key frame, query), kept iff MAXIMUM_DISTANCE_HAMMING (25,
 * CKeyFrame.h:12) > distance  (CTrackerSVI.cpp:1221-1259, CTrackerSV.cpp:725-763, CTrackerGT.cpp:425-452).
 *   pools: the key frames' pools back to back (32 B rows), pool_seg: n_clouds+1 row offsets (device), max_pool:
 *   the largest pool (host; only sizes the launch);  out_idx / out_dist: n_clouds x nq, row c = key frame c,
 *   out_idx = row INSIDE that key frame's pool (cv::DMatch::trainIdx) or -1, out_dist as svi_match_hamming256. */
int svi_match_clouds_dev(svi_matcher* m, const uint8_t* q, int nq, const uint8_t* pools, const int32_t* pool_seg,
                         int n_clouds, int max_pool, int max_dist_exclusive, int32_t* out_idx, int32_t* out_dist);

/* dist[i] = popcount(a_i ^ b_i), i < n  (cv::norm(a,b,NORM_HAMMING) batched). */
int svi_hamming256_pairs(svi_matcher* m, const uint8_t* a, const uint8_t* b, int n, int32_t* dist);
int svi_hamming256_pairs_dev(svi_matcher* m, const uint8_t* a, const uint8_t* b, int n, int32_t* dist);

/* CTriangulator::getPointInLEFT batched: f32 pixel pairs -> f64 camera points.
 *   ok[i] = 0 and xyz untouched(0) iff uL - uR < min_disparity (0.01, CTriangulator.h:21)
 *   z = duR_flipped/(uL-uR), x = z (uL-cx)/f, y = z (vL-cy)/f   with duR_flipped = -P_R(0,3). */
int svi_triangulate_rectified(svi_matcher* m, double f, double cx, double cy, double duR_flipped,
                              double min_disparity, const float* uvL, const float* uvR, int n,
                              double* xyz, uint8_t* ok);
int svi_triangulate_rectified_dev(svi_matcher* m, double f, double cx, double cy, double duR_flipped,
                                  double min_disparity, const float* uvL, const float* uvR, int n,
                                  double* xyz, uint8_t* ok);

/* Fused stereo step (a-1 + a-2 + a-3 of SURVEY.md §8a): gated match with the match's right pixel
 * fed straight into the triangulation; out_xyz[i] valid iff out_idx[i] >= 0 && ok[i]. */
int svi_match_triangulate_dev(svi_matcher* m, const uint8_t* q, int nq, const uint8_t* t, int nt,
                              int batch, const svi_gate* gate, int max_dist_exclusive,
                              double f, double cx, double cy, double duR_flipped, double min_disparity,
                              int32_t* out_idx, int32_t* out_dist, double* out_xyz, uint8_t* ok);

/* ------------------------------------------------------------------------------------------
 * Temporal tracking schedule — the per-landmark geometry and decisions that
 * CFundamentalMatcher::getPoseStereoPosit (:368-733), trackEpipolar (:794-1315), trackManual
 * (:1366-2019), _getMatchSampleRecursiveU/V (:2142-2334), _getMatch (:2336-2397) and
 * _addMeasurementToLandmarkLEFT (:2400-2450) run one landmark at a time around the matcher,
 * restated as batched device passes over ALL active landmarks of a frame.
 * What stays with the caller: BRIEF extraction and GFTT detection (OpenCV) - the
 * passes below say WHERE descriptors are needed (plan, samples) and decide on the descriptors the
 * caller's extractor returns (ragged pools: landmark i owns pool rows [seg[i], seg[i+1])).
 * All pointers are device pointers unless a comment says host; launches are asynchronous on the
 * matcher's stream.
 * ---------------------------------------------------------------------------------------- */

typedef struct svi_track_camera {
    double P_left[12];   /* CPinholeCamera::m_matProjection LEFT, 3x4 row-major  (CPinholeCamera.h:28-29) */
    double P_right[12];  /* ... RIGHT */
    double K_inv[9];     /* m_matIntrinsicPInverse of the LEFT camera, row-major (CPinholeCamera.h:31)   */
    double width, height;/* m_dWidthPixels / m_dHeightPixels; the FoV gate is rect(28,28,w-56,h-56) (:61) */
} svi_track_camera;

/* status bits of svi_track_record.status */
enum {
    SVI_TRK_FOV_LEFT        = 1 << 0,  /* rounded LEFT projection inside the FoV rect                */
    SVI_TRK_FOV_RIGHT       = 1 << 1,  /* rounded RIGHT projection inside the FoV rect               */
    SVI_TRK_EPI_NO_MOTION   = 1 << 2,  /* |t|^2 == 0 for the detection point: no epipolar line (:847) */
    SVI_TRK_EPI_OUT_OF_SIGHT= 1 << 3,  /* "vertical out of sight" (:880-885)                          */
    SVI_TRK_EPI_BAD_PROJ    = 1 << 4,  /* "caught bad projection ..." (:908, :937)                    */
    SVI_TRK_EPI_ZERO_LENGTH = 1 << 5,  /* "zero line length" (:975)                                   */
    SVI_TRK_EPI_OK          = 1 << 6   /* s3_* fields describe a sampling run                         */
};

/* One landmark's schedule for one frame (AoS, 168 bytes, written by svi_track_plan_dev). */
typedef struct svi_track_record {
    double  xyz_left[3];     /* T_world_to_left * vecPointXYZOptimized                                 (:377) */
    double  line[3];         /* epipolar coefficients F * vecUVReferenceLEFT                            (:855) */
    double  s3_start;        /* dUMinimum (s3_axis 0) or dVForUMinimum (s3_axis 1)                 (:984-991) */
    float   uv_left[2];      /* getProjectionRounded LEFT                                     (:378, :850)    */
    float   uv_right[2];     /* getProjectionRounded RIGHT                                             (:379) */
    float   search_range;    /* (1 + motion_scaling) * last disparity                             (:365,:386) */
    float   s1_roi_left[2];  /* uv_left - 4*size  : top-left of the (8*size+1)^2 stage-1 ROI           (:395) */
    float   s1_roi_right[2]; /* uv_right - 4*size                                                      (:449) */
    float   s2_left[4];      /* stage-2 search rectangle LEFT : upper-left (u,v), lower-right (u,v) (:505-506) */
    float   s2_right[4];     /* ... RIGHT                                                          (:622-623) */
    float   s2_ext_left[4];  /* rectangle the descriptors are computed in (grown by 4*size, clamped) (:527-530) */
    float   s2_ext_right[4]; /*                                                                     (:644-647) */
    int32_t s3_count;        /* uDeltaU or uDeltaV: samples per recursion depth                   (:966-967)  */
    int32_t s3_axis;         /* 0: sample over U (uDeltaV < uDeltaU), 1: over V                        (:981) */
    int32_t status;          /* SVI_TRK_* bits                                                                */
} svi_track_record;

/* Plan: projection, FoV gate, stage-1/2 rectangles and the clipped epipolar segment of every landmark.
 *   T_world_to_left : host, 12 doubles (R row-major, t)
 *   dp_T_left_to_world : host, n_dp x 12 doubles - CDetectionPoint::matTransformationLEFTtoWORLD; the
 *                     fundamental matrix of a detection point is K^-T (R [t]x) K^-1 with (R,t) =
 *                     T_world_to_left * dp_T (:800-806)
 *   xyz_world n x 3 f64, kp_size n f32, last_disparity n f32, uv_reference n x 2 f64 (vecUVReferenceLEFT),
 *   dp_index n i32 (which detection point owns the landmark)
 *   records  n svi_track_record
 *   s3_seg   n+1 i32 (nullable): exclusive scan of s3_count over landmarks with SVI_TRK_EPI_OK
 *   total_samples : host out (nullable); when given the call synchronises the stream. */
int svi_track_plan_dev(svi_matcher* m, const svi_track_camera* cam, const double* T_world_to_left,
                       const double* dp_T_left_to_world, int n_dp, double motion_scaling,
                       const double* xyz_world, const float* kp_size, const float* last_disparity,
                       const double* uv_reference, const int32_t* dp_index, int n,
                       svi_track_record* records, int32_t* s3_seg, int64_t* total_samples);

/* Epipolar sampling at recursion depth `depth` (0, then 2 on failure: CFundamentalMatcher.h:84-85):
 * key points along the clipped line, the ROI the reference cuts around them and the key points in ROI
 * coordinates (:2154-2227 / :2246-2318).
 *   sel      : n_sel i32 landmark indices (nullable: identity over all n_sel = n landmarks)
 *   seg      : n_sel+1 i32 segment starts inside sample_uv (for sel == NULL this is s3_seg of the plan;
 *              landmarks without SVI_TRK_EPI_OK own empty segments)
 *   sample_uv: total x 2 f32, key point positions in ROI coordinates (what the extractor is given)
 *   roi      : n_sel x 4 f32 (fUTopLeft, fVTopLeft, fWidth, fHeigth) */
int svi_track_epipolar_samples_dev(svi_matcher* m, const svi_track_camera* cam, const svi_track_record* records,
                                   const float* kp_size, const int32_t* sel, int n_sel, const int32_t* seg,
                                   int depth, float* sample_uv, float* roi);

/* status of a ragged match / stereo verification */
enum {
    SVI_TRK_MATCH_OK             = 0,
    SVI_TRK_MATCH_EMPTY_POOL     = 1, /* "empty key point pool" / "could not compute descriptors"       */
    SVI_TRK_MATCH_DISTANCE       = 2, /* best distance >= cut-off                                       */
    SVI_TRK_MATCH_ORIGINAL       = 3, /* "ORIGINAL matching distance too big" (:2375-2391)              */
    SVI_TRK_MATCH_RANGE          = 4, /* "insufficient search range" (CTriangulator.cpp:199, :268)      */
    SVI_TRK_MATCH_DISPARITY      = 5, /* "zero disparity" (CTriangulator.cpp:329)                       */
    SVI_TRK_MATCH_DEPTH          = 6, /* "invalid depth" (:417, :2426)                                  */
    SVI_TRK_MATCH_OTHER_MISMATCH = 7, /* "triangulation descriptor mismatch" (:423, :573)               */
    SVI_TRK_MATCH_SKIPPED        = 8  /* active[i] == 0                                                  */
};

/* _getMatch batched (:2336-2397): k=1 Hamming NN of q_i inside its own pool segment, first minimum wins,
 * accepted iff  cutoff_relative > distance  and (original != NULL)  cutoff_original > popcount(original_i ^ winner).
 *   out_idx : index INSIDE the segment (cv::DMatch::trainIdx) or -1;  out_dist : best distance or 257 for an
 *   empty pool (reported even when rejected);  out_status : SVI_TRK_MATCH_*;  active nullable (u8). */
int svi_match_ragged_dev(svi_matcher* m, const uint8_t* q, const uint8_t* original, const uint8_t* active, int nq,
                         const int32_t* seg, const uint8_t* pool, int cutoff_relative, int cutoff_original,
                         int32_t* out_idx, int32_t* out_dist, int32_t* out_status);

typedef struct svi_track_stereo_params {
    double f, cx, cy, duR_flipped, min_disparity; /* CTriangulator::getPointInLEFT (CTriangulator.cpp:326-356) */
    double depth_min, depth_max;                  /* CTriangulator.cpp:20-21                                    */
    int    cutoff_match;                          /* 100: CTriangulator.cpp:13; accepted iff cutoff > distance  */
    int    cutoff_other;                          /* 25 (stage 1) / 50 (stage 2); <0: no check (stage 3)        */
    int    other_inclusive;                       /* 1: accepted iff distance <= cutoff_other (:423: throws on
                                                     cutoff < d);  0: accepted iff cutoff_other > distance (:573) */
    int    search_in_left;                        /* 0: getPointTriangulatedInRIGHT, 1: ...InLEFT               */
} svi_track_stereo_params;

/* Hand-over from a temporal match to the stereo search: the reference point and the top-left corner every call
 * site passes to getPointTriangulatedInRIGHT / InLEFT.
 *   mode 0  stage 1, found in LEFT  (:407-412)  uv_ref = s1_roi_left + (4s,4s);  topleft = (max(0, s1_roi_left.u - range), s1_roi_left.v)
 *   mode 1  stage 1, found in RIGHT (:460-466)  uv_ref = s1_roi_right + (4s,4s); topleft = s1_roi_right
 *   mode 2  stage 2, found in LEFT  (:546-562)  p = s2_left.ul + kp - (4s,4s);   v = p.v - 4s;  ok iff 0 <= v;
 *                                               uv_ref = p;  topleft = (max(0, p.u - range - 4s), v)
 *   mode 3  stage 2, found in RIGHT (:663-681)  p = s2_right.ul + kp - (4s,4s);  topleft = (max(0, p.u - 4s), v)
 *   mode 4  stage 3 (:2382-2383, :2413-2424)    uv_ref = kp + roi.(u,v);         topleft = (max(0, uv_ref.u - range - 4s), uv_ref.v - 4s)
 * kp = pool_uv[seg[w] + idx[w]] is the key point of the winning candidate (modes 2-4), range = record.search_range,
 * s = kp_size.  sel (nullable) maps row w of seg/idx/roi/outputs to its landmark; ok[w] = 0 when idx[w] < 0 or the
 * mode's range test fails. */
int svi_track_handover_dev(svi_matcher* m, int mode, const svi_track_record* records, const float* kp_size,
                           const int32_t* sel, int n_sel, const int32_t* seg, const float* pool_uv, const int32_t* idx,
                           const float* roi, float* uv_ref, float* topleft, uint8_t* ok);

/* Candidate key points of CTriangulator::getPointTriangulatedInRIGHT (CTriangulator.cpp:194-211) /
 * getPointTriangulatedInLEFT (:262-282): every integer pixel column of ONE row.
 *   range:  seg (n+1 i32) = exclusive scan of uSearchRangeComplete, out_status = SVI_TRK_MATCH_OK / _RANGE /
 *           _SKIPPED, roi n x 4 f32 = the cv::Rect handed to the extractor (u, v, width, height; :213/:284),
 *           total (host, nullable; synchronises)
 *   candidates: pool_uv total x 2 f32 in ROI coordinates: (4s + k, 4s) in RIGHT, (4s + k + 1, 4s) in LEFT
 *   search_range is read only when search_in_left != 0 (p_fSearchRange). */
int svi_track_stereo_range_dev(svi_matcher* m, double width, int search_in_left, const float* uv_ref,
                               const float* topleft, const float* kp_size, const float* search_range,
                               const uint8_t* active, int n, int32_t* seg, int32_t* out_status, float* roi,
                               int64_t* total);
int svi_track_stereo_candidates_dev(svi_matcher* m, int search_in_left, const float* kp_size, int n,
                                    const int32_t* seg, float* pool_uv);

/* The stereo half of every stage: CTriangulator::getPointTriangulatedInRIGHT/InLEFT (:185-324) on a ragged
 * pool of row candidates, triangulation, depth gate (:416-419) and the check of the found descriptor against
 * the landmark's last descriptor of that image (:423 / :573).
 *   ref       nq x 32  descriptor found in the first image (the match query)
 *   last_other nq x 32 landmark's last descriptor in the searched image (nullable iff cutoff_other < 0)
 *   uv_ref    nq x 2 f32 pixel of `ref` in its image
 *   topleft   nq x 2 f32 (p_fUTopLeft, p_fVTopLeft) of the searched ROI
 *   seg/pool/pool_uv : ragged candidates; pool_uv total x 2 f32 in ROI coordinates (cv::KeyPoint::pt)
 *   out_uv_other nq x 2 f32 = pool_uv[idx] + topleft (:110, :244);  out_xyz nq x 3 f64 LEFT camera frame */
int svi_track_stereo_verify_dev(svi_matcher* m, const svi_track_stereo_params* prm, const uint8_t* ref,
                                const uint8_t* last_other, const uint8_t* active, const float* uv_ref,
                                const float* topleft, int nq, const int32_t* seg, const uint8_t* pool,
                                const float* pool_uv, int32_t* out_idx, int32_t* out_dist, int32_t* out_status,
                                float* out_uv_other, double* out_xyz);

/* ------------------------------------------------------------------------------------------
 * BRIEF-256 extraction — replaces cv::xfeatures2d::BriefDescriptorExtractor::compute( image( roi ),
 * keypoints, descriptors ) as called at CTriangulator.cpp:84,147,218,289 and CFundamentalMatcher.cpp:401,450,534,651,2345:
 * integral image, key points within 28 px of the ROI border are dropped (KeyPointsFilter::runByImageBorder), 256
 * comparisons of 9x9 box sums, test t -> byte t/8, bit 7 - t%8.  OpenCV's baked test pairs are not part of the reference
 * tree: `pattern` is the caller's table, 256 x (y1, x1, y2, x2) in int8, each within [-24, 24] (generated_32.i order).
 * ---------------------------------------------------------------------------------------- */
typedef struct svi_brief svi_brief;
int svi_brief_create(svi_matcher* m, const int8_t* pattern /* host, 1024 */, svi_brief** out);
int svi_brief_destroy(svi_brief* b);
/* integral image of one frame (side 0 = LEFT, 1 = RIGHT); image: device u8, `stride` bytes per row */
int svi_brief_set_image_dev(svi_brief* b, int side, const uint8_t* image, int width, int height, int stride);
/* debug / test tap: the (height+1) x (width+1) int32 integral image of a side, device to device */
int svi_brief_integral_dev(svi_brief* b, int side, int32_t* out);
/* roi: n x 4 i32 (x, y, w, h) = the cv::Rect of every pool; key points kp_uv[seg[i] .. seg[i+1]) in ROI coordinates.
 * Outputs are compacted per ROI in input order: seg_out n+1, kp_out / desc_out with room for total_in rows (desc_out
 * 4-byte aligned); *total_out (host, nullable; synchronises) = key points kept.  A ROI that does not lie inside the
 * frame (cv::Mat::operator() would assert) or is not larger than 56 px keeps nothing. */
int svi_brief_compute_dev(svi_brief* b, int side, const int32_t* roi, const int32_t* seg, const float* kp_uv, int n,
                          int64_t total_in, int32_t* seg_out, float* kp_out, uint8_t* desc_out, int64_t* total_out);

/* ------------------------------------------------------------------------------------------
 * The cascades themselves - CFundamentalMatcher's entry points (src/core/CFundamentalMatcher.h:112-170) behind the
 * boundary: the C++ host composes the passes above (masks instead of the reference's exceptions, its own HIP kernels
 * for the glue; the host waits only where a ragged pool has to be sized).  One svi_tracker per camera pair, bound to a
 * matcher (stream, device).  BRIEF extraction is either the built-in svi_brief or the caller's extractor; GFTT
 * detection (stage 2) is the caller's detector.
 * ---------------------------------------------------------------------------------------- */
typedef struct svi_tracker svi_tracker;

/* cv::DescriptorExtractor::compute( image( roi ), keypoints, descriptors ) for n regions of image `side` (0 LEFT,
 * 1 RIGHT) at once.  Device pointers, work ordered on `stream` (the matcher's hipStream_t; the callback may
 * synchronise it).  roi n x 4 f32 (x, y, w, h as handed to cv::Rect), seg n+1 i32, kp_uv total_in x 2 f32 in ROI
 * coordinates; total_in = rows behind kp_uv and room in the outputs (an upper bound: seg[n] is the number of key points).
 * Outputs: the key points kept (OpenCV drops those near the ROI border) per region in input order, their 32-byte
 * descriptors, seg_out n+1, *total_out = rows kept.  Return 0 on success. */
typedef int (*svi_extract_fn)(void* user, int side, const float* roi, const int32_t* seg, const float* kp_uv, int n,
                              int64_t total_in, int32_t* seg_out, float* kp_out, uint8_t* desc_out, int64_t* total_out,
                              void* stream);
/* cv::FeatureDetector::detect( image( rect ), keypoints ) for n search rectangles (rect n x 4 f32: upper-left and
 * lower-right corner, as CFundamentalMatcher.cpp:505-509 builds cv::Rect from them); rows with active[i] == 0 must get
 * an empty segment.  kp_out (room for `cap` rows) in rectangle coordinates, seg_out n+1, *total_out <= cap. */
typedef int (*svi_detect_fn)(void* user, int side, const float* rect, const uint8_t* active, int n, int64_t cap,
                             int32_t* seg_out, float* kp_out, int64_t* total_out, void* stream);

int svi_tracker_create(svi_matcher* m, const svi_track_camera* cam, svi_tracker** out);
int svi_tracker_destroy(svi_tracker* t);
/* the built-in BRIEF extractor (svi_brief_set_image_dev must have been called for the frame) ... */
int svi_tracker_set_brief(svi_tracker* t, svi_brief* b);
/* ... or the caller's (replaces the built-in one; fn == NULL removes it) */
int svi_tracker_set_extractor(svi_tracker* t, svi_extract_fn fn, void* user);
/* key points the detector may return per call (`capacity` rows of kp_out are provided) */
int svi_tracker_set_detector(svi_tracker* t, svi_detect_fn fn, void* user, int64_t capacity);

/* the active landmarks of a frame, SoA device arrays of n rows (CLandmark: vecPointXYZOptimized, dKeyPointSize,
 * getLastDisparity(), vecUVReferenceLEFT, the detection point it belongs to, getLastDescriptorLEFT / RIGHT,
 * matDescriptorReferenceLEFT).  The arrays must stay valid until the cascades of the frame have run. */
typedef struct svi_track_landmarks {
    int            n;
    const double*  xyz_world;       /* n x 3 */
    const float*   kp_size;         /* n     */
    const float*   last_disparity;  /* n     */
    const double*  uv_reference;    /* n x 2 */
    const int32_t* dp_index;        /* n     */
    const uint8_t* last_desc_left;  /* n x 32, 16-byte aligned */
    const uint8_t* last_desc_right; /* n x 32 */
    const uint8_t* ref_desc_left;   /* n x 32 (stage 3 only) */
} svi_track_landmarks;

/* outcome per landmark (device arrays of n rows, caller-owned): status SVI_TRK_MATCH_* (SKIPPED where no stage ran),
 * stage (nullable) 1 / 2 / 3 = the stage that produced the measurement, 0 none; the measurement (CMatchTracking /
 * CSolverStereoPosit::CMatch: ptUVLEFT, ptUVRIGHT, vecPointXYZLEFT) and the descriptors found are valid where
 * status == SVI_TRK_MATCH_OK */
typedef struct svi_track_result {
    int32_t* status;
    int8_t*  stage;
    float*   uv_left;    /* n x 2 */
    float*   uv_right;   /* n x 2 */
    double*  xyz_left;   /* n x 3 */
    uint8_t* desc_left;  /* n x 32 */
    uint8_t* desc_right; /* n x 32 */
} svi_track_result;

/* svi_track_plan_dev for the frame, kept inside the tracker (host transforms as there) */
int svi_tracker_plan(svi_tracker* t, const double* T_world_to_left, const double* dp_T_left_to_world, int n_dp,
                     double motion_scaling, const svi_track_landmarks* lm);
/* (re)binds the descriptor arrays of the frame's landmarks (n x 32 each, 16-byte aligned) without planning again */
int svi_tracker_set_descriptors(svi_tracker* t, const uint8_t* last_desc_left, const uint8_t* last_desc_right,
                                const uint8_t* ref_desc_left);
/* the plan of the frame: device records (n) and stage-3 segment table (n+1), total samples */
int svi_tracker_records(svi_tracker* t, const svi_track_record** records, const int32_t** s3_seg, int64_t* total_samples);

/* active: u8 per landmark, nullable (= all).  Every call below initialises *out first (SKIPPED / 0). */
/* stage 1 of getPoseStereoPosit / trackManual (:391-486): descriptor at the projection, LEFT then RIGHT, stereo partner */
int svi_track_stage1(svi_tracker* t, const uint8_t* active, const svi_track_result* out);
/* stage 2 (:489-709): detector inside the search rectangle, LEFT then RIGHT */
int svi_track_stage2(svi_tracker* t, const uint8_t* active, const svi_track_result* out);
/* trackEpipolar (:794-1315): landmarks whose detection point has moved are searched along the clipped epipolar line
 * (recursion depths 0 and 2, _getMatch, _addMeasurementToLandmarkLEFT); the others by stage 2 (:1026-1290) */
int svi_track_epipolar(svi_tracker* t, const uint8_t* active, const svi_track_result* out);
/* trackManual (:1366-2019): stage 1 -> stage 2 -> epipolar, each for what the previous one lost */
int svi_track_manual(svi_tracker* t, const uint8_t* active, const svi_track_result* out);
/* addNewLandmarks (:83-193): stereo partner + triangulation of n freshly detected LEFT key points
 * (getPointTriangulatedInRIGHTFull: search window fMinimumSearchRangePixels = 60, no depth gate, no second descriptor
 * check); needs no plan.  out->stage is not written. */
int svi_track_add_new_landmarks(svi_tracker* t, const float* uv_left, const float* kp_size, const uint8_t* desc_left,
                                int n, const svi_track_result* out);

/* ------------------------------------------------------------------------------------------
 * Frame pose from the stage-1/2 matches — replaces CSolverStereoPosit::getTransformationWORLDtoLEFT
 * (src/optimization/CSolverStereoPosit.cpp:8-170): iteratively re-weighted Gauss-Newton on the
 * stereo reprojection error, the whole loop in ONE launch (no host round trip per iteration).
 * ---------------------------------------------------------------------------------------- */
typedef struct svi_posit_params {
    double P_left[12], P_right[12];     /* m_matProjectionLEFT / RIGHT, 3x4 row-major                        */
    int    min_points;                  /* 25   CSolverStereoPosit.h:89: solved iff min_points < n            */
    int    min_inliers;                 /* 15   :90                                                           */
    int    max_iterations;              /* 1000 :91                                                           */
    double max_error_inlier_l2;         /* 10   :92  squared pixel error above which a point is down-weighted */
    double max_error_average_l2;        /* 9    :93                                                           */
    double max_risk;                    /* 2    :94                                                           */
    double convergence_delta;           /* 1e-5 :95                                                           */
    double min_translation_l2;          /* 1e-3 :98                                                           */
} svi_posit_params;

enum {
    SVI_POSIT_OK            = 0,
    SVI_POSIT_FEW_POINTS    = 1, /* "insufficient number of points"            (:168) */
    SVI_POSIT_NOT_CONVERGED = 2, /* "system did not converge"                  (:165) */
    SVI_POSIT_INACCURATE    = 3, /* "insufficient accuracy"                    (:127-130) */
    SVI_POSIT_HIGH_RISK     = 4  /* "inconsistent with prior (HIGH RISK...)"   (:148-151) */
};

typedef struct svi_posit_result {
    double  T_world_to_left[12];  /* R row-major then t; the estimate the loop ended with (also on failure)  */
    double  error_average;        /* dErrorTotalPixelsL2 / n                                                 */
    double  risk;                 /* dOptimizationRISK                                                       */
    int32_t status;               /* SVI_POSIT_*  (the reference throws CExceptionPoseOptimization for 1..4) */
    int32_t iterations;
    int32_t inliers;
    int32_t n;                    /* measurements used (active ones)                                         */
} svi_posit_result;

void svi_posit_params_default(svi_posit_params* p);
/* poses and t_imu are host arrays (12 / 3 doubles); xyz_world n x 3 f64 (CMatch::vecPointXYZWORLD), uv_left / uv_right
 * n x 2 f32 (CMatch::ptUVLEFT / ptUVRIGHT) and active (u8, nullable: e.g. status == OK of the tracking stages) are
 * device arrays; result is a host struct - the call synchronises the matcher's stream. */
int svi_stereo_posit_dev(svi_matcher* m, const svi_posit_params* prm, const double* T_world_to_left_last,
                         const double* t_imu, const double* T_world_to_left_estimate, const double* xyz_world,
                         const float* uv_left, const float* uv_right, const uint8_t* active, int n,
                         svi_posit_result* result);

/* getPoseStereoPosit (:340-760): stage 1 -> stage 2 over the landmarks with active != 0 (bIsOptimal), then
 * CSolverStereoPosit::getTransformationWORLDtoLEFT on what they found; *pose as svi_stereo_posit_dev (synchronises) */
int svi_track_pose_stereo_posit(svi_tracker* t, const uint8_t* active, const svi_posit_params* prm,
                                const double* T_world_to_left_last, const double* t_imu,
                                const double* T_world_to_left_estimate, const svi_track_result* out,
                                svi_posit_result* pose);

/* ------------------------------------------------------------------------------------------
 * Per-landmark refinement — replaces CLandmark::optimize / _getOptimizedLandmarkSTEREOUV (src/types/CLandmark.cpp:281-296,
 * 447-581), run by the reference for every active landmark of every frame (CTrackerGT.cpp:197).
 * ---------------------------------------------------------------------------------------- */
typedef struct svi_landmark_params {
    int    min_measurements;      /* 5    CLandmark.h:98: refined iff min_measurements < count           */
    int    cap_iterations;        /* 1000 :90                                                             */
    double convergence_delta;     /* 1e-5 :93                                                             */
    double kernel_max_error_l2;   /* 10   :95                                                             */
    double min_inlier_ratio;      /* 0.5  :94                                                             */
    double max_error_average_l2;  /* 9    :96                                                             */
} svi_landmark_params;

enum {
    SVI_LM_OPT_SKIPPED       = 0, /* too few measurements: position kept, bIsOptimal = true  (:293-295)                  */
    SVI_LM_OPT_OPTIMAL       = 1, /* converged, inlier ratio fine, average error below the limit: bIsOptimal = true     */
    SVI_LM_OPT_CONVERGED     = 2, /* converged and accepted (uOptimizationsSuccessful++) but not optimal                */
    SVI_LM_OPT_REJECTED      = 3, /* converged with too few inliers: initial guess kept, uOptimizationsFailed++ (:556)  */
    SVI_LM_OPT_NOT_CONVERGED = 4  /* iteration cap reached: initial guess kept, uOptimizationsFailed++ (:573)           */
};

void svi_landmark_params_default(svi_landmark_params* p);
/* All arrays on the device.  frame_P_left / frame_P_right: n_frames x 12 f64, the WORLD->image projection of every
 * frame that holds a measurement (CMeasurementLandmark::matProjectionWORLDtoLEFT / RIGHT, Types.h:12-54);
 * landmark l owns measurements [meas_seg[l], meas_seg[l+1]) in the order they were added: meas_frame (i32 index into
 * the frame arrays), meas_uv_left / meas_uv_right (f32 pixels);  xyz_in / xyz_out n x 3 f64 (vecPointXYZOptimized);
 * out_status SVI_LM_OPT_*, out_error_average = dCurrentAverageSquaredError, out_iterations.  Asynchronous. */
int svi_landmarks_optimize_dev(svi_matcher* m, const svi_landmark_params* prm, const double* frame_P_left,
                               const double* frame_P_right, int n_frames, const int32_t* meas_seg,
                               const int32_t* meas_frame, const float* meas_uv_left, const float* meas_uv_right,
                               const double* xyz_in, int n, double* xyz_out, int32_t* out_status,
                               double* out_error_average, int32_t* out_iterations);

/* ------------------------------------------------------------------------------------------
 * Bundle adjustment — replaces the g2o::SparseOptimizer m_cOptimizerSparse of Cg2oOptimizer
 * (Cg2oOptimizer.h:80) together with its solver stack (Cg2oOptimizer.cpp:83-89).
 * Vertex ids follow the reference: landmark id = uID, pose id = uID + 1e6 (Cg2oOptimizer.h:83);
 * variables are ordered by ascending id like g2o's index mapping.
 * Poses are LEFT->WORLD isometries (Cg2oOptimizer.cpp:1232-1237) given as 12 doubles:
 * R row-major (9) then t (3).
 * ---------------------------------------------------------------------------------------- */

typedef struct svi_ba svi_ba;

typedef struct svi_ba_options {
    /* g2o::ParameterCamera Kcam (Cg2oOptimizer.cpp:106) */
    double fx, fy, cx, cy;
    /* CStereoCamera::m_dBaselineMeters (CStereoCamera.h:28); used by the disparity measurement */
    double baseline_m;
    /* RobustKernelCauchy delta (g2o default 1.0; Cg2oOptimizer.cpp:1018,1042,1070) */
    double cauchy_delta;
    /* g2o::OptimizationAlgorithmLevenberg defaults */
    double lm_tau;              /* 1e-5 */
    double lm_good_step_lower;  /* 1/3  */
    double lm_good_step_upper;  /* 2/3  */
    int    lm_max_trials;       /* 10   */
    /* _setLandmarkMeasurementsWORLD thresholds on squared norms (Cg2oOptimizer.h:92-94) */
    double max_depth_xyz_l2;        /* 10    */
    double max_depth_uvdepth_l2;    /* 50    */
    double max_depth_uvdisp_l2;     /* 10000 */
    double sane_position_l2;        /* 1e12 (Cg2oOptimizer.h:95) */
    /* placement */
    int    device;
    void*  stream;              /* NULL: own stream */
    /* landmark sharding: this handle owns the landmarks whose slot falls in the
     * rank-th of n_ranks contiguous, edge-balanced ranges; pose-only edges belong to rank 0. */
    int    rank;
    int    n_ranks;
    /* record hipEvents around every phase (svi_ba_get_phase_times) */
    int    profile;
    /* reduced camera system tile edge (48 or 96; 0 = default 48: shorter pivot chains per dependency level - measured faster on
     * configs 3, 4 and 5 although the levels double) */
    int    chol_tile;
    /* elimination order of the reduced camera system: 0 = nested dissection of the key-frame sequence (independent
     * chains factorised side by side), 1 = natural (ascending id: one chain) */
    int    chol_order;
    /* record a HIP event pair around the Jacobian sweep (K2 + K3) of every linearisation inside the ordinary,
     * non-synchronised LM loop (svi_ba_get_sweep_time); unlike `profile` nothing else is timed and the host still
     * waits on the published trial scalars only */
    int    sweep_events;
} svi_ba_options;

void svi_ba_options_default(svi_ba_options* o);

int svi_ba_create(const svi_ba_options* o, svi_ba** out);
int svi_ba_destroy(svi_ba* ba);

/* --- graph construction (host side, before svi_ba_initialize) ----------------------------- */
int svi_ba_add_pose(svi_ba* ba, int64_t id, const double T[12], int fixed);
int svi_ba_add_landmark(svi_ba* ba, int64_t id, const double p[3], int fixed);
/* EdgeSE3PointXYZ / EdgeSE3PointXYZDepth / EdgeSE3PointXYZDisparity with identity offset
 * (factories at Cg2oOptimizer.cpp:999-1073). info_upper = upper triangle row-major
 * (00 01 02 11 12 22); the reference only ever sets the diagonal. robust != 0: Cauchy. */
int svi_ba_add_edge_xyz(svi_ba* ba, int64_t pose_id, int64_t lm_id, const double z[3],
                        const double info_upper[6], int robust);
int svi_ba_add_edge_depth(svi_ba* ba, int64_t pose_id, int64_t lm_id, const double z[3],
                          const double info_upper[6], int robust);
int svi_ba_add_edge_disparity(svi_ba* ba, int64_t pose_id, int64_t lm_id, const double z[3],
                              const double info_upper[6], int robust);
/* bulk form of the three above: type[i] in {0 xyz, 1 depth, 2 disparity} */
int svi_ba_add_edges_bulk(svi_ba* ba, int64_t n, const int32_t* type, const int64_t* pose_id,
                          const int64_t* lm_id, const double* z /*n x 3*/,
                          const double* info_upper /*n x 6*/, const int32_t* robust);
/* EdgeSE3 odometry (Cg2oOptimizer.cpp:1248-1266); Z = measured X_i^-1 X_j (12 doubles),
 * info_upper = 21 doubles (upper triangle row-major of the 6x6). */
int svi_ba_add_edge_se3(svi_ba* ba, int64_t id_i, int64_t id_j, const double Z[12],
                        const double info_upper[21], int robust);
/* EdgeSE3LinearAcceleration (edge_se3_linear_acceleration.cpp:106-116); off = IMU->LEFT offset
 * (12 doubles, NULL = identity). */
int svi_ba_add_edge_accel(svi_ba* ba, int64_t pose_id, const double a[3], const double off[12],
                          const double info_upper[6]);
/* EdgePointXYZ landmark closure (Cg2oOptimizer.cpp:448-458): e = p_j - p_i - z. One of the two
 * landmarks must be fixed (the reference fixes the reference landmark, :445). */
int svi_ba_add_edge_lm_lm(svi_ba* ba, int64_t id_i, int64_t id_j, const double z[3],
                          const double info_upper[6], int robust);

/* --- reference-shaped construction (the rules of Cg2oOptimizer) --------------------------- */
/* _setAndgetPose (Cg2oOptimizer.cpp:1229-1290) + _getEdgeLinearAcceleration (:982-997):
 * adds the pose vertex (translation shifted by `shift`, NULL = 0), the odometry edge from
 * `from_id` with measurement = current relative estimate and information
 * 1e5*diag(s,s,s,1,1,1), s = 1/(1+|t_ij|^2), and the gravity edge carrying accel (NULL = 0). */
int svi_ba_add_keyframe(svi_ba* ba, int64_t id, int64_t from_id, const double T_left_to_world[12],
                        const double shift[3], const double accel[3]);
/* g2o::ParameterSE3Offset eOFFSET_IMUtoLEFT (Cg2oOptimizer.cpp:209-213; CPinholeCameraIMU::m_matTransformationIMUtoCAMERA of
 * the LEFT camera): the offset every gravity edge created by svi_ba_add_keyframe carries from now on (12 doubles, R row-major
 * then t; identity until set - the stereo-only cameras).  The edge error is R_pose R_off a - (0,0,-1). */
int svi_ba_set_imu_offset(svi_ba* ba, const double off[12]);
/* _setLandmarkMeasurementsWORLD (Cg2oOptimizer.cpp:1383-1466): for each measurement of a
 * landmark already in the graph apply the consistency gate 0.75 < |X^-1 l|^2/|p_m|^2 < 1.25 and
 * choose XYZ / UV-depth / UV-disparity by |p_m|^2 (10 / 50 / 10000), information from w = 1/z_m.
 * uv_left/uv_right: n x 2 float (cv::Point2f), xyz_left: n x 3 double.
 * stored[3] (nullable) receives the number of edges of each kind that were added. */
int svi_ba_add_measurements(svi_ba* ba, int64_t pose_id, int64_t n, const int64_t* lm_id,
                            const float* uv_left, const float* uv_right, const double* xyz_left,
                            int64_t stored[3]);

/* --- optimisation -------------------------------------------------------------------------- */
/* g2o initializeOptimization(): freeze the graph, build the device structures, upload. */
int svi_ba_initialize(svi_ba* ba);
/* one g2o SparseOptimizer::optimize(iterations) block: structure + lambda re-initialised,
 * returns the number of LM iterations executed in *performed (nullable). */
int svi_ba_optimize(svi_ba* ba, int iterations, int* performed);
/* Cg2oOptimizer::_optimizeUnLimited (Cg2oOptimizer.cpp:954-980):
 *   optimize(first); prev = 1.1*chi2; while (chi2/prev < ratio) { prev = chi2; optimize(block); }
 * nominal = first + block*k (the reference's counter), executed = LM iterations really run. */
int svi_ba_optimize_until(svi_ba* ba, double ratio, int first, int block,
                          uint64_t* nominal, uint64_t* executed);
/* The estimates stay in HBM across optimize() calls; the host copy that the getters, the write-back,
 * the .g2o writer, graph edits and a re-initialisation read is refreshed by the first such call after an
 * optimize() (g2o keeps its estimates in host memory: Cg2oOptimizer.cpp:1468-1540 reads them there).
 * With n_ranks > 1 that refresh gathers the landmark shards through the all-reduce hook, so the first
 * reading call after an optimize() must be made by every rank.  svi_ba_sync_host does it explicitly. */
int svi_ba_sync_host(svi_ba* ba);
/* g2o OptimizableGraph::chi2(): plain sum e' Omega e with the most recently evaluated errors;
 * robust: g2o activeRobustChi2(). Either pointer may be NULL. */
int svi_ba_chi2(svi_ba* ba, double* plain, double* robust);
/* current LM damping (after the last optimize) */
int svi_ba_lambda(svi_ba* ba, double* lambda);

/* --- results -------------------------------------------------------------------------------- */
int svi_ba_get_pose(svi_ba* ba, int64_t id, double T[12]);
int svi_ba_get_landmark(svi_ba* ba, int64_t id, double p[3]);
/* all vertices in ascending-id order; ids nullable */
int svi_ba_num_poses(svi_ba* ba, int64_t* n);
int svi_ba_num_landmarks(svi_ba* ba, int64_t* n);
int svi_ba_num_edges(svi_ba* ba, int64_t* n);
int svi_ba_get_poses(svi_ba* ba, int64_t* ids, double* T /*n x 12*/);
int svi_ba_get_landmarks(svi_ba* ba, int64_t* ids, double* p /*n x 3*/);
/* _applyOptimizationToLandmarks rule (Cg2oOptimizer.cpp:1468-1512): remove every landmark with
 * |p|^2 >= sane_position_l2 together with its edges; *removed (nullable) = how many. The graph
 * must be re-initialised afterwards. */
int svi_ba_prune_diverged(svi_ba* ba, int64_t* removed);
/* The whole write-back of Cg2oOptimizer::optimize (:1468-1540) in one call, vertices in ascending-id order:
 *   landmark: kept[k] = 1 and xyz[k] = estimate - shift when |estimate|^2 < sane_position_l2 (:1486-1489), else
 *             kept[k] = 0, xyz[k] = 0 and the vertex leaves the graph with its edges (:1493-1503);
 *   key frame: T[k] = estimate with translation - shift (:1527-1528).
 * Array sizes are those of svi_ba_num_landmarks / svi_ba_num_poses BEFORE the call; any output may be NULL;
 * shift NULL = 0; *erased (nullable) = landmarks removed (the graph must then be re-initialised). */
int svi_ba_apply_optimization(svi_ba* ba, const double shift[3], int64_t* lm_ids, double* lm_xyz, uint8_t* lm_kept,
                              int64_t* kf_ids, double* kf_T, int64_t* erased);

/* --- .g2o text interchange (Cg2oOptimizer.cpp:495-497, 514) ------------------------------- */
int svi_ba_load_g2o(svi_ba* ba, const char* path);
int svi_ba_save_g2o(svi_ba* ba, const char* path);

/* --- multi-GPU hook ------------------------------------------------------------------------ */
/* Sum-all-reduce of `count` doubles living at device pointer `buf`, in place, ordered on
 * `stream` (a hipStream_t). Return 0 on success. The harness implements it with
 * torch.distributed / RCCL (svi_mapper_amd/dist.py); with n_ranks == 1 it is never called.
 * Every rank issues its calls in the same order: per LM trial the reduced system and four
 * scalars (chi2 robust / plain, landmark and pose part of the step scale), plus the pose sums in front of the first trial of an optimize() block. */
typedef int (*svi_allreduce_fn)(void* user, void* buf, size_t count, void* stream);
int svi_ba_set_allreduce(svi_ba* ba, svi_allreduce_fn fn, void* user);

/* A native hook: RCCL (ncclAllReduce, ncclDouble, ncclSum) on the handle's stream, one process per GPU, no Python in
 * between.  librccl is resolved from the process like the HIP runtime is (csrc/rccl_hook.cpp).
 *   rank 0:      svi_rccl_unique_id(id)  and hands the 128 bytes to the other ranks by whatever means the host has
 *   every rank:  svi_rccl_create(id, rank, n_ranks, device, &c);  svi_ba_set_allreduce(ba, svi_rccl_allreduce, c) */
typedef struct svi_rccl svi_rccl;
int svi_rccl_available(void); /* 1: librccl resolved in this process (no collective): let the ranks agree on it BEFORE svi_rccl_create */
int svi_rccl_unique_id(void* id_out /* 128 bytes */);
int svi_rccl_create(const void* unique_id, int rank, int n_ranks, int device, svi_rccl** out);
int svi_rccl_destroy(svi_rccl* c);
int svi_rccl_allreduce(void* user /* svi_rccl* */, void* buf, size_t count, void* stream);

/* --- instrumentation ----------------------------------------------------------------------- */
enum svi_ba_phase {
    SVI_PH_LINEARIZE_LM   = 0, /* K2: landmark-major Jacobian sweep -> N,Z per edge, H_ll, b_l, chi2   */
    SVI_PH_LINEARIZE_POSE = 1, /* K3: pose-major sweep -> per-pose H_pp / b_p                 */
    SVI_PH_POSE_EDGES     = 2, /* odometry / gravity / landmark-closure edges                 */
    SVI_PH_SCHUR          = 3, /* K4: per-landmark inverse + windowed S, g contributions      */
    SVI_PH_ASSEMBLE       = 4, /* K4b: ordered reduction of the windows into the tile store  */
    SVI_PH_ALLREDUCE      = 5, /* RCCL hook                                                   */
    SVI_PH_CHOLESKY       = 6, /* K5: tile-sparse LL' + triangular solves                     */
    SVI_PH_BACKSUB_UPDATE = 7, /* K6+K7: landmark back-substitution, oplus                    */
    SVI_PH_CHI2           = 8, /* K8: error sweep                                             */
    SVI_PH_COUNT          = 9
};
/* accumulated device milliseconds and launch counts per phase since the last reset
 * (only when options.profile != 0) */
int svi_ba_get_phase_times(svi_ba* ba, double ms[SVI_PH_COUNT], int64_t calls[SVI_PH_COUNT]);
int svi_ba_reset_phase_times(svi_ba* ba);

/* accumulated device milliseconds and launch count of the Jacobian sweep inside the LM loop since the last reset
 * (options.sweep_events != 0 or options.profile != 0) */
int svi_ba_get_sweep_time(svi_ba* ba, double* ms_total, int64_t* calls);

/* sizes of the device structures, for roofline accounting */
typedef struct svi_ba_stats {
    int64_t n_poses, n_poses_free, n_landmarks, n_landmarks_local;
    int64_t n_edges_proj, n_edges_proj_local, n_edges_se3, n_edges_accel, n_edges_lmlm;
    int64_t n_schur_tiles, n_window_blocks;    /* K4 decomposition */
    int64_t chol_n, chol_tile, chol_tiles_nnz; /* reduced system */
    int64_t chol_steps;                        /* dependency levels of its tile columns = launches on the critical path */
    int64_t reduce_doubles;                    /* payload of one all-reduce */
    double  chol_flops;                        /* of one factorisation on the tile structure */
    uint64_t lm_iterations, lm_trials, chol_failures;
    uint64_t backsolve_timeouts;               /* trials ended by SVI_ERR_INTERNAL (never counted as chol_failures) */
} svi_ba_stats;
int svi_ba_get_stats(svi_ba* ba, svi_ba_stats* s);

/* --- debug / parity taps (device state -> host; small graphs) ------------------------------ */
/* per projection edge, in insertion order: error (3) and Jacobians at the current estimate */
int svi_ba_debug_edge_jacobians(svi_ba* ba, double* err /*E x 3*/, double* J_pose /*E x 18*/,
                                double* J_lm /*E x 9*/);
/* pose-only edges at the current estimate, in insertion order among their kind (rank 0 holds them): EdgeSE3 error (6) and
 * its two 6x6 Jacobians (row-major, w.r.t. the increments of pose i / pose j); gravity edge error (3) and its 3x6 Jacobian */
int svi_ba_debug_aux_jacobians(svi_ba* ba, double* se3_err /*n_se3 x 6*/, double* se3_Ji /*n_se3 x 36*/, double* se3_Jj,
                               double* acc_err /*n_accel x 3*/, double* acc_J /*n_accel x 18*/);
/* dense reduced system of the last linearisation with damping `lambda`:
 * S (n x n row-major, full symmetric) and g (n), n = 6 * free poses; *n_out receives n */
int svi_ba_debug_reduced_system(svi_ba* ba, double lambda, double* S, double* g, int64_t cap,
                                int64_t* n_out);

/* mean duration (ms) of the Jacobian sweep kernel over `reps` back-to-back launches on the handle's stream,
 * bracketed by two HIP events (the state is not modified: the sweep only writes linearisation outputs) */
int svi_ba_debug_time_sweep(svi_ba* ba, int reps, double* ms_avg);
/* the same for one of the two kernels of the sweep: which = 1 landmark-major (K2), 2 pose-major (K3) */
int svi_ba_debug_time_sweep_part(svi_ba* ba, int reps, int which, double* ms_avg);

/* the sweep with cold caches: before each of the `reps` sweeps a scratch buffer of `evict_bytes` (>= 2 x the 256 MiB
 * Infinity Cache to be sure) is overwritten, every sweep sits between its own event pair; mean ms per sweep */
int svi_ba_debug_time_sweep_cold(svi_ba* ba, int reps, size_t evict_bytes, double* ms_avg);

/* timing probe of the diagonal-tile Cholesky kernel (tile 48 or 96): mean ms per launch over `reps`
 * launches, truncated after phase `stop_after` (0 full, 1 pivot sweep, 2 +scale/store, 3 +diagonal
 * block inverses, 4 +off-diagonal inverse blocks, 5 load only; 6: ms[0] = shader cycles and ms[1] = 100 MHz
 * ticks spent in the pivot sweep, ms must have room for 2 doubles) */
int svi_debug_chol_probe(int device, int tile, int reps, int stop_after, double* ms);

/* test knob: how many polls a workgroup of the one-launch backward substitution grants a pending entry (default 2^22).  With a
 * tiny budget the hand-overs time out, and svi_ba_optimize must return SVI_ERR_INTERNAL instead of treating the trial as
 * "not positive definite" (process-wide; restore the default afterwards) */
int svi_debug_set_backsolve_spin_limit(int polls);

#ifdef __cplusplus
}
#endif
#endif /* SVI_HOT_H */
