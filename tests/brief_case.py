"""Synthetic frames and a stand-in test-pair table for the BRIEF tests (OpenCV's own table is not available offline)."""
import numpy as np


def pattern(seed=1):
    """256 x (y1, x1, y2, x2) int8 in [-24, 24]: isotropic Gaussian pairs like the original BRIEF G II sampling"""
    r = np.random.default_rng(seed)
    p = np.clip(np.rint(r.normal(0, 48 / 5.0, (256, 4))), -24, 24).astype(np.int8)
    p[0] = [-24, -24, 24, 24]      # extreme corners of the patch are legal
    p[1] = [24, 24, -24, -24]
    return p


def image(h, w, seed):
    """smooth random texture + noise, uint8"""
    r = np.random.default_rng(seed)
    img = np.zeros((h, w))
    for s in (32, 16, 8, 4):
        g = r.normal(0, 1, (h // s + 2, w // s + 2))
        img += np.kron(g, np.ones((s, s)))[:h, :w] * s
    img += r.normal(0, 3, (h, w))
    img = (img - img.min()) / (img.max() - img.min()) * 255
    return img.astype(np.uint8)


def pools(h, w, n, seed, max_kp=60):
    """random ROIs (some too small, some outside the frame) with ragged key points incl. border and half-pixel cases"""
    r = np.random.default_rng(seed)
    roi = np.zeros((n, 4), np.float32)
    roi[:, 0] = r.uniform(-10, w - 40, n)
    roi[:, 1] = r.uniform(-10, h - 40, n)
    roi[:, 2] = r.uniform(30, 300, n)
    roi[:, 3] = r.uniform(30, 120, n)
    roi[0] = [0, 0, w, h]                        # the whole frame
    roi[1] = [w - 57, h - 57, 57, 57]            # smallest usable ROI in the corner: only (28,28) survives
    roi[2] = [5.9, 7.2, 56.9, 80]                # truncates to width 56: nothing survives
    cnt = r.integers(0, max_kp, n)
    cnt[3] = 0
    seg = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int32)
    kp = np.zeros((int(seg[-1]), 2), np.float32)
    for i in range(n):
        a, b = seg[i], seg[i + 1]
        kp[a:b, 0] = r.uniform(20, max(roi[i, 2] - 20, 21), b - a)
        kp[a:b, 1] = r.uniform(20, max(roi[i, 3] - 20, 21), b - a)
        if b - a >= 6:
            kp[a] = [28, 28]
            kp[a + 1] = [27.5, 28.5]                                   # cvRound -> (28, 28): kept; centre (28, 29)
            kp[a + 2] = [np.trunc(roi[i, 2]) - 28.5, 30.25]            # the half-pixel corner case at the right border
            kp[a + 3] = [np.trunc(roi[i, 2]) - 28, 40]                 # first column that is dropped
            kp[a + 4] = [27.49, 40]                                    # dropped
            kp[a + 5] = [np.float32(np.nan), 30]
    return roi, seg, kp
