"""CPU tests of the BRIEF restatement (oracle/oracle_brief.c) against plain numpy.  PARITY UNPINNED (OpenCV absent)."""
import numpy as np

import brief_case


def test_integral_and_descriptor_against_numpy(oracle):
    img = brief_case.image(120, 200, 1)
    S = oracle.brief_integral(img)
    ref = np.zeros((121, 201), np.int64)
    ref[1:, 1:] = img.astype(np.int64).cumsum(0).cumsum(1)
    assert np.array_equal(S, ref)
    pat = brief_case.pattern()
    roi = np.array([[10, 20, 150, 90]], np.float32)
    kp = np.array([[40.2, 33.7], [28, 28], [121.4, 61.0]], np.float32)
    seg_out, kp_out, desc = oracle.brief_compute(S, pat, roi, [0, 3], kp)
    assert list(seg_out) == [0, 3] and np.array_equal(kp_out, kp)
    blur = lambda cy, cx: int(img[cy - 4:cy + 5, cx - 4:cx + 5].astype(np.int64).sum())  # noqa: E731
    for k in range(3):
        cx, cy = 10 + int(kp[k, 0] + 0.5), 20 + int(kp[k, 1] + 0.5)
        bits = [blur(cy + y1, cx + x1) < blur(cy + y2, cx + x2) for y1, x1, y2, x2 in pat.astype(int)]
        want = np.packbits(np.array(bits, np.uint8))           # MSB first: test t -> bit 7 - t % 8
        assert np.array_equal(desc[k], want)


def test_border_filter_and_bad_rois(oracle):
    img = brief_case.image(376, 1241, 2)
    S = oracle.brief_integral(img)
    roi, seg, kp = brief_case.pools(376, 1241, 64, 3)
    seg_out, kp_out, desc = oracle.brief_compute(S, brief_case.pattern(), roi, seg, kp)
    ri = np.trunc(roi).astype(int)
    for i in range(len(roi)):
        x, y, w, h = ri[i]
        pts = kp[seg[i]:seg[i + 1]]
        usable = w > 56 and h > 56 and x >= 0 and y >= 0 and x + w <= 1241 and y + h <= 376
        with np.errstate(invalid="ignore"):
            q = np.rint(pts)
            keep = usable & (q[:, 0] >= 28) & (q[:, 0] < w - 28) & (q[:, 1] >= 28) & (q[:, 1] < h - 28) & np.isfinite(pts).all(1)
        assert seg_out[i + 1] - seg_out[i] == keep.sum(), i
        assert np.array_equal(kp_out[seg_out[i]:seg_out[i + 1]], pts[keep])
    assert seg_out[2] - seg_out[1] <= 2 and seg_out[3] == seg_out[2]     # corner ROI keeps only cvRound == (28,28); width-56 ROI nothing
    assert len(desc) == seg_out[-1] > 100
