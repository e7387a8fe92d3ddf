/*
 * oracle_match.c — CPU restatement of the reference's descriptor-matching arithmetic.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under svi_mapper_amd/ may include, link or call this file;
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the checker.
 *
 * PARITY UNPINNED: the arithmetic lives in OpenCV (cv::BFMatcher / cv::norm), which is not part of
 * /root/reference and not installed here, and the reference ships no tests or golden vectors
 * (SURVEY.md §4, §8c).  The semantics below are restated from the reference's call sites:
 *
 *   cv::BFMatcher(cv::NORM_HAMMING)->match(query 1x32, pool Nx32)     src/core/CTriangulator.cpp:12,93,156,227,298
 *        k = 1; distance = popcount(q ^ t) over 256 bits              (same value as src/types/CBNode.h:622-627)
 *        the first (lowest index) minimum wins                         (OpenCV scans j ascending with strict '<')
 *   caller keeps the match iff  cutoff > distance                      src/core/CTriangulator.cpp:107,170,241,312
 *   empty pool -> "no match"                                           src/core/CTriangulator.cpp:86-98
 *   candidate pool = integer pixels of one image row, ascending u      src/core/CTriangulator.cpp:67-77,201-211,272-282
 *   getPointInLEFT                                                     src/core/CTriangulator.cpp:326-356
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

static inline int hamming256(const uint8_t* a, const uint8_t* b)
{
    uint64_t x[4], y[4];
    memcpy(x, a, 32);
    memcpy(y, b, 32);
    return __builtin_popcountll(x[0] ^ y[0]) + __builtin_popcountll(x[1] ^ y[1]) +
           __builtin_popcountll(x[2] ^ y[2]) + __builtin_popcountll(x[3] ^ y[3]);
}

/* Batched form of SURVEY.md Appendix A. gate arrays may be NULL (has_gate == 0). */
void orc_match_hamming256(const uint8_t* q, int nq, const uint8_t* t, int nt,
                          int has_gate, const float* q_uv, const float* t_uv,
                          const float* q_umin, const float* q_umax, float v_tol,
                          int max_dist_exclusive, int32_t* out_idx, int32_t* out_dist)
{
    for (int i = 0; i < nq; ++i) {
        int best = 257, best_j = -1;
        for (int j = 0; j < nt; ++j) {
            if (has_gate) {
                const float dv = fabsf(t_uv[2 * j + 1] - q_uv[2 * i + 1]);
                const float tu = t_uv[2 * j];
                if (!(dv <= v_tol && q_umin[i] <= tu && tu < q_umax[i])) continue;
            }
            const int d = hamming256(q + 32 * (size_t)i, t + 32 * (size_t)j);
            if (d < best) { best = d; best_j = j; } /* strict '<': lowest index wins ties */
        }
        if (best_j >= 0 && max_dist_exclusive > best) { /* cutoff > distance, CTriangulator.cpp:107 */
            out_idx[i]  = best_j;
            out_dist[i] = best;
        } else {
            out_idx[i]  = -1;
            out_dist[i] = 257;
        }
    }
}

/* cv::norm(a, b, cv::NORM_HAMMING) batched (CFundamentalMatcher.cpp:404,423,...) */
void orc_hamming256_pairs(const uint8_t* a, const uint8_t* b, int n, int32_t* dist)
{
    for (int i = 0; i < n; ++i) dist[i] = hamming256(a + 32 * (size_t)i, b + 32 * (size_t)i);
}

/* CTriangulator::getPointInLEFT (CTriangulator.cpp:326-356). Pixel coordinates are cv::Point2f;
 * the disparity is formed in float, everything else in double, with the reference's operand
 * order  m_dFInverse*dZ*( u - m_dPu ). */
void orc_triangulate_rectified(double f, double cx, double cy, double duR_flipped, double min_disparity,
                               const float* uvL, const float* uvR, int n, double* xyz, uint8_t* ok)
{
    const double finv = 1.0 / f; /* m_dFInverse, CTriangulator.cpp:15 */
    for (int i = 0; i < n; ++i) {
        const float disparity = uvL[2 * i] - uvR[2 * i];
        if ((double)disparity < min_disparity) { /* :329 */
            ok[i] = 0;
            xyz[3 * i] = xyz[3 * i + 1] = xyz[3 * i + 2] = 0.0;
            continue;
        }
        const double z = duR_flipped / (double)disparity;                 /* :340 */
        xyz[3 * i + 0] = finv * z * ((double)uvL[2 * i] - cx);            /* :346 */
        xyz[3 * i + 1] = finv * z * ((double)uvL[2 * i + 1] - cy);        /* :347 */
        xyz[3 * i + 2] = z;
        ok[i] = 1;
    }
}
