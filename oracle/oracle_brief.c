/*
 * oracle_brief.c — CPU restatement of the BRIEF-256 extraction the reference delegates to OpenCV.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under svi_mapper_amd/ may include, link or call this file;
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the checker.
 *
 * PARITY UNPINNED: cv::xfeatures2d::BriefDescriptorExtractor (created at src/core/CTriangulator.cpp:11 with 32 bytes,
 * called as m_pExtractor->compute( image( roi ), keypoints, descriptors ) at CTriangulator.cpp:84,147,218,289 and
 * CFundamentalMatcher.cpp:401,450,534,651,2345) is not part of /root/reference and not installed.  Restated from the
 * published OpenCV 3.x algorithm (xfeatures2d/src/brief.cpp, features2d/src/keypoint.cpp):
 *   - integral image (CV_32S) of the image handed in - here of the whole frame: every box used lies inside the ROI,
 *     so the ROI-relative integral gives the same box sums;
 *   - KeyPointsFilter::runByImageBorder with PATCH_SIZE/2 + KERNEL_SIZE/2 = 24 + 4 = 28: a key point is kept iff its
 *     cvRound()ed position lies in [28, w-28) x [28, h-28) of the ROI (all dropped when w or h <= 56);
 *   - 256 tests  smoothedSum(y1,x1) < smoothedSum(y2,x2)  on 9x9 box sums centred at ((int)(pt.x+0.5)+x, (int)(pt.y+0.5)+y),
 *     test t -> byte t/8, bit 7 - t%8.
 * OpenCV's 256 baked test pairs (generated_32.i) are NOT available offline: the table is an input (y1,x1,y2,x2 per test).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* (h+1) x (w+1) int32 integral image, row 0 and column 0 zero */
void orc_brief_integral(const uint8_t* img, int w, int h, int stride, int32_t* sum)
{
    memset(sum, 0, sizeof(int32_t) * (size_t)(w + 1));
    for (int y = 0; y < h; ++y) {
        int32_t run = 0;
        int32_t* out = sum + (size_t)(y + 1) * (w + 1);
        const int32_t* up = sum + (size_t)y * (w + 1);
        out[0] = 0;
        for (int x = 0; x < w; ++x) { run += img[(size_t)y * stride + x]; out[x + 1] = up[x + 1] + run; }
    }
}

static int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* 9x9 box sum centred at (cx, cy); integral coordinates are clamped to the frame (only the half-pixel corner case
 * px = w_roi - 28.5 of a ROI touching the frame border can reach it - OpenCV would read past the ROI there) */
static int32_t box9(const int32_t* sum, int w, int h, int cx, int cy)
{
    const size_t W = (size_t)w + 1;
    const int x0 = clampi(cx - 4, 0, w), x1 = clampi(cx + 5, 0, w), y0 = clampi(cy - 4, 0, h), y1 = clampi(cy + 5, 0, h);
    return sum[(size_t)y1 * W + x1] - sum[(size_t)y1 * W + x0] - sum[(size_t)y0 * W + x1] + sum[(size_t)y0 * W + x0];
}

static int round_half_even(float v) { return (int)lrintf(v); } /* cvRound */

/* roi: n x 4 int32 (x, y, w, h) as cv::Rect; key points in ROI coordinates; outputs compacted per ROI.
 * returns the number of key points kept; seg_out has n+1 entries. */
int orc_brief_compute(const int32_t* sum, int w, int h, const int8_t* pattern, const int32_t* roi, const int32_t* seg, const float* kp_uv, int n,
                      int32_t* seg_out, float* kp_out, uint8_t* desc_out)
{
    int kept = 0;
    for (int i = 0; i < n; ++i) {
        seg_out[i] = kept;
        const int rx = roi[4 * i], ry = roi[4 * i + 1], rw = roi[4 * i + 2], rh = roi[4 * i + 3];
        if (rw <= 56 || rh <= 56) continue;                                       /* runByImageBorder: everything goes */
        if (rx < 0 || ry < 0 || rx + rw > w || ry + rh > h) continue;              /* cv::Mat::operator()(Rect) would assert */
        for (int k = seg[i]; k < seg[i + 1]; ++k) {
            const float px = kp_uv[2 * k], py = kp_uv[2 * k + 1];
            if (!(fabsf(px) < 1.0e8f) || !(fabsf(py) < 1.0e8f)) continue;
            const int qx = round_half_even(px), qy = round_half_even(py);
            if (!(qx >= 28 && qx < rw - 28 && qy >= 28 && qy < rh - 28)) continue;
            const int cx = rx + (int)(px + 0.5), cy = ry + (int)(py + 0.5);       /* smoothedSum: (int)(pt.x + 0.5) */
            uint8_t* d = desc_out + 32 * (size_t)kept;
            memset(d, 0, 32);
            for (int t = 0; t < 256; ++t) {
                const int8_t* p = pattern + 4 * t;
                const int32_t a = box9(sum, w, h, cx + p[1], cy + p[0]);
                const int32_t b = box9(sum, w, h, cx + p[3], cy + p[2]);
                if (a < b) d[t >> 3] |= (uint8_t)(1u << (7 - (t & 7)));
            }
            kp_out[2 * kept] = px; kp_out[2 * kept + 1] = py;
            ++kept;
        }
    }
    seg_out[n] = kept;
    return kept;
}
