"""CPU tests of the tracking-schedule oracle (oracle/oracle_track.c): golden fixture, an independent numpy
restatement of the geometry, brute-force matching, and the statuses the reference's exceptions map to.
PARITY UNPINNED: the reference has no fixtures for this path; the fixture pins OUR oracle."""
import os

import numpy as np
import pytest

import track_scene as ts

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def cam(oracle):
    return oracle.track_camera(ts.P_LEFT, ts.P_RIGHT, ts.K_INV, ts.W, ts.H)


def plan_of(oracle, cam, sc):
    return oracle.track_plan(cam, sc.T_est_w2l, sc.dp_T, sc.motion_scaling, sc.xyz_world, sc.kp_size, sc.last_disparity, sc.uv_reference,
                             sc.dp_index)


def test_golden_fixture(oracle, cam):
    g = np.load(os.path.join(HERE, "golden", "track_small.npz"))
    sc = ts.Scene(n=96, seed=3)
    rec, seg = plan_of(oracle, cam, sc)
    assert np.array_equal(rec.view(np.uint8).reshape(len(rec), -1), g["records"])
    assert np.array_equal(seg, g["seg"])
    for depth in (0, 2):
        s, roi = oracle.track_epipolar_samples(cam, rec, sc.kp_size, seg, depth)
        assert np.array_equal(s, g["samples%d" % depth]) and np.array_equal(roi, g["roi%d" % depth])
    om = oracle.OracleFundamentalMatcher(cam, sc.stereo_dict())
    res = om.epipolar(rec, sc.kp_size, sc.extract_one, sc.last_left, sc.ref_desc)
    assert np.array_equal(np.array([d["status"] for d in res], np.int32), g["s3_status"])
    assert np.array_equal(np.array([d.get("xyz", (0, 0, 0)) for d in res]), g["s3_xyz"])


def test_plan_against_numpy(oracle, cam):
    """vectorised float64 numpy restatement of the projection / rectangles (independent of the C code)"""
    sc = ts.Scene(n=500, seed=11)
    rec, seg = plan_of(oracle, cam, sc)
    R, t = sc.T_est_w2l[:9].reshape(3, 3), sc.T_est_w2l[9:]
    p = sc.xyz_world @ R.T + t
    assert np.allclose(rec["xyz_left"], p, rtol=1e-13, atol=1e-12)
    with np.errstate(all="ignore"):
        uL = np.round((ts.FX * p[:, 0] / p[:, 2] + ts.CX).astype(np.float32))
        vL = np.round((ts.FX * p[:, 1] / p[:, 2] + ts.CY).astype(np.float32))
        uR = np.round(((ts.FX * p[:, 0] + ts.DUR) / p[:, 2] + ts.CX).astype(np.float32))
    # np.round is half-to-even, roundf half-away: they can differ only on exact .5 values (none with these inputs)
    sane = np.abs(p[:, 2]) > 1e-3
    assert np.array_equal(rec["uv_left"][sane, 0], uL[sane]) and np.array_equal(rec["uv_left"][sane, 1], vL[sane])
    assert np.array_equal(rec["uv_right"][sane, 0], uR[sane])
    fov = (uL >= 28) & (uL < ts.W - 28) & (vL >= 28) & (vL < ts.H - 28)
    assert np.array_equal((rec["status"] & oracle.FOV_LEFT) != 0, fov)
    half = 4 * sc.kp_size
    assert np.array_equal(rec["s1_roi_left"][:, 0], rec["uv_left"][:, 0] - half)
    assert np.array_equal(rec["search_range"], np.float32(1.0 + sc.motion_scaling) * sc.last_disparity)
    # stage-2 rectangle: round(round(sqrt|u - cx|/10 + ms) * 15) around the projection, clamped to the image
    wu = np.round(np.sqrt(np.abs(rec["uv_left"][:, 0].astype(np.float64) - ts.CX)) / 10 + sc.motion_scaling)
    hw = np.round(wu * 15)
    assert np.array_equal(rec["s2_left"][:, 0], np.maximum(rec["uv_left"][:, 0] - hw, 0).astype(np.float32))
    assert np.array_equal(rec["s2_left"][:, 2], np.minimum(rec["uv_left"][:, 0] + hw, ts.W).astype(np.float32))
    # segment starts = exclusive scan of the counts of the landmarks with a sampling run
    cnt = np.where((rec["status"] & oracle.EPI_OK) != 0, rec["s3_count"], 0)
    assert np.array_equal(seg, np.concatenate([[0], np.cumsum(cnt)]).astype(np.int32))


def test_epipolar_line_and_samples(oracle, cam):
    sc = ts.Scene(n=400, seed=5)
    rec, seg = plan_of(oracle, cam, sc)
    ok = (rec["status"] & oracle.EPI_OK) != 0
    degenerate = sc.dp_index == len(sc.dp_T) - 1   # detection point on top of the estimate: |t| ~ 1e-17, the line is rounding noise
    ok_all, ok = ok, ok & ~degenerate
    assert ok.sum() > 50
    # F = K^-T (R [t]x) K^-1 of the landmark's detection point (reference formula, CFundamentalMatcher.cpp:800-806)
    A = np.eye(4); A[:3, :3] = sc.T_est_w2l[:9].reshape(3, 3); A[:3, 3] = sc.T_est_w2l[9:]
    for i in np.nonzero(ok)[0][:40]:
        B = np.eye(4); d = sc.dp_T[sc.dp_index[i]]; B[:3, :3] = d[:9].reshape(3, 3); B[:3, 3] = d[9:]
        T = A @ B
        tx = np.array([[0, -T[2, 3], T[1, 3]], [T[2, 3], 0, -T[0, 3]], [-T[1, 3], T[0, 3], 0]])
        F = ts.K_INV.T @ (T[:3, :3] @ tx) @ ts.K_INV
        c = F @ np.array([sc.uv_reference[i, 0], sc.uv_reference[i, 1], 1.0])
        assert np.allclose(rec["line"][i], c, rtol=1e-9, atol=1e-18)
    s0, roi0 = oracle.track_epipolar_samples(cam, rec, sc.kp_size, seg, 0)
    s2, roi2 = oracle.track_epipolar_samples(cam, rec, sc.kp_size, seg, 2)
    for i in np.nonzero(ok)[0]:
        a, b = seg[i], seg[i + 1]
        assert b - a == rec["s3_count"][i] > 0
        absu = s0[a:b, 0].astype(np.float64) + roi0[i, 0]
        absv = s0[a:b, 1].astype(np.float64) + roi0[i, 1]
        c = rec["line"][i]
        resid = np.abs(c[0] * absu + c[1] * absv + c[2]) / np.hypot(c[0], c[1])
        assert resid.max() < 2e-3, "depth-0 samples lie on the line (float32 pixels)"
        # one pixel apart along the sampling axis, starting at s3_start
        main0 = absu if rec["s3_axis"][i] == 0 else absv
        assert np.allclose(np.diff(main0), 1.0, atol=1e-3)
        assert abs(main0[0] - rec["s3_start"][i]) < 1e-3
        # depth 2 is the same run shifted by +2 on the other axis (CFundamentalMatcher.h:84-85)
        other0 = absv if rec["s3_axis"][i] == 0 else absu
        other2 = (s2[a:b, 1].astype(np.float64) + roi2[i, 1]) if rec["s3_axis"][i] == 0 else (s2[a:b, 0].astype(np.float64) + roi2[i, 0])
        assert np.allclose(other2 - other0, 2.0, atol=2e-3)
        # segment stays inside the clipping window around the projection
        hl = 15.0 + np.sqrt(abs(rec["uv_left"][i, 0] - ts.CX)) / 10 * sc.motion_scaling * 10
        assert absu.min() >= max(rec["uv_left"][i, 0] - hl, 0) - 1e-3 and absu.max() <= min(rec["uv_left"][i, 0] + hl, ts.W) + 1e-3
    for i in np.nonzero(~ok_all)[0]:
        assert seg[i] == seg[i + 1]


def test_zero_motion_and_bad_indices(oracle, cam):
    ident = np.array([1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0], np.float64)
    xyz = np.array([[0.5, 0.1, 10.0], [0.0, 0.0, 5.0], [1.0, 1.0, -3.0]])
    rec, seg = oracle.track_plan(cam, ident, ident[None], 1.0, xyz, np.full(3, 7, np.float32), np.full(3, 30, np.float32),
                                 np.array([[600., 180.], [610., 190.], [0., 0.]]), np.array([0, 5, -1], np.int32))
    assert np.all(rec["status"] & oracle.EPI_NO_MOTION)          # |t|^2 == 0, and out-of-range detection points
    assert np.all((rec["status"] & oracle.EPI_OK) == 0) and seg[-1] == 0
    assert rec["status"][0] & oracle.FOV_LEFT and rec["status"][0] & oracle.FOV_RIGHT
    assert not rec["status"][2] & oracle.FOV_LEFT                  # behind the camera: projects outside
    # empty input
    rec, seg = oracle.track_plan(cam, ident, ident[None], 1.0, np.zeros((0, 3)), np.zeros(0, np.float32), np.zeros(0, np.float32),
                                 np.zeros((0, 2)), np.zeros(0, np.int32))
    assert len(rec) == 0 and list(seg) == [0]


def test_match_ragged_against_bruteforce(oracle):
    r = np.random.default_rng(1)
    nq = 200
    cnt = r.integers(0, 40, nq)
    cnt[:5] = 0
    seg = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int32)
    pool = r.integers(0, 256, (seg[-1], 32), dtype=np.uint8)
    q = r.integers(0, 256, (nq, 32), dtype=np.uint8)
    orig = q.copy()
    for i in range(nq):  # plant near matches, duplicates (ties) and far originals
        if cnt[i] >= 3:
            k = r.integers(0, cnt[i])
            pool[seg[i] + k] = ts.flip_bits(q[i], r.integers(0, 70), i)
            if i % 3 == 0:
                k2 = r.integers(0, cnt[i])
                pool[seg[i] + k2] = pool[seg[i] + k]
            if i % 5 == 0:
                orig[i] = ts.flip_bits(q[i], 120, 900 + i)
    active = (r.random(nq) < 0.9).astype(np.uint8)
    idx, dist, st = oracle.match_ragged(q, orig, seg, pool, 50, 100, active)
    bits = np.unpackbits(pool, axis=1)
    for i in range(nq):
        if not active[i]:
            assert st[i] == oracle.M_SKIPPED and idx[i] == -1
            continue
        if cnt[i] == 0:
            assert st[i] == oracle.M_EMPTY_POOL and idx[i] == -1 and dist[i] == 257
            continue
        d = (bits[seg[i]:seg[i + 1]] != np.unpackbits(q[i])).sum(1)
        b = int(np.argmin(d))  # first minimum
        assert dist[i] == d[b]
        if d[b] >= 50:
            assert st[i] == oracle.M_DISTANCE and idx[i] == -1
        elif (bits[seg[i] + b] != np.unpackbits(orig[i])).sum() >= 100:
            assert st[i] == oracle.M_ORIGINAL and idx[i] == -1
        else:
            assert st[i] == oracle.M_OK and idx[i] == b
    assert set(st) >= {oracle.M_OK, oracle.M_EMPTY_POOL, oracle.M_DISTANCE, oracle.M_ORIGINAL, oracle.M_SKIPPED}


def test_stereo_range_and_candidates(oracle):
    uv = np.array([[300, 100], [50, 80], [20, 60], [1200, 90], [1235, 90]], np.float32)
    tl = np.array([[200, 72], [30, 52], [0, 32], [1100, 62], [1230, 62]], np.float32)
    kp = np.full(5, 7, np.float32)
    seg, st, roi = oracle.track_stereo_range(ts.W, 0, uv, tl, kp)
    # RIGHT search: ceil(uL - uTopLeft - 4s) candidates, "insufficient search range" when uL <= uTopLeft + 4s
    assert list(st) == [0, oracle.M_RANGE, oracle.M_RANGE, 0, oracle.M_RANGE]
    assert list(np.diff(seg)) == [72, 0, 0, 72, 0]
    cand = oracle.track_stereo_candidates(0, kp, seg)
    assert np.array_equal(cand[:3], np.array([[28, 28], [29, 28], [30, 28]], np.float32))
    assert roi[0].tolist() == [200, 72, 72 + 57, 57]
    # LEFT search: ceil(min(range, W - uTopLeft)) + 1 candidates starting one pixel right of the border centre
    rng = np.array([40.5, 0.0, -3.0, 500.0, 10.0], np.float32)
    seg, st, roi = oracle.track_stereo_range(ts.W, 1, uv, tl, kp, rng)
    assert list(st) == [0, oracle.M_RANGE, oracle.M_RANGE, 0, 0]
    assert list(np.diff(seg)) == [42, 0, 0, 142, 11]
    cand = oracle.track_stereo_candidates(1, kp, seg)
    assert np.array_equal(cand[:2], np.array([[29, 28], [30, 28]], np.float32))


def test_stereo_verify_statuses(oracle):
    st = ts.Scene(n=8, seed=1).stereo_dict()
    r = np.random.default_rng(3)
    ref = r.integers(0, 256, (7, 32), dtype=np.uint8)
    last = np.stack([ts.flip_bits(ref[i], 10, i) for i in range(7)])
    seg = np.array([0, 4, 8, 12, 16, 16, 20, 24], np.int32)
    pool = r.integers(0, 256, (24, 32), dtype=np.uint8)
    pool_uv = np.tile(np.array([[28, 28], [29, 28], [30, 28], [31, 28]], np.float32), (6, 1))
    uv_ref = np.array([[400, 100]] * 7, np.float32)
    tl = np.array([[300, 72]] * 7, np.float32)
    pool[1] = ts.flip_bits(ref[0], 5, 50)            # 0: fine
    pool[4 + 2] = ts.flip_bits(ref[1], 99, 51)       # 1: best is 99 < 100 but last_other mismatch
    pool[8 + 0] = ts.flip_bits(ref[2], 100, 52)      # 2: 100 is not < 100 -> matching distance
    pool[12 + 3] = ts.flip_bits(ref[3], 3, 53)       # 3: fine but zero disparity (uv_ref moved onto it)
    uv_ref[3] = [331, 100]
    pool[16 + 1] = ts.flip_bits(ref[5], 4, 54)       # 5: depth out of range (disparity 0.5 px -> z = 772 m... allowed), use depth_max
    pool[20 + 2] = ts.flip_bits(ref[6], 2, 55)       # 6: inclusive cut-off: exactly 25 passes
    last[6] = ts.flip_bits(pool[20 + 2], 25, 56)
    prm = oracle.stereo_params(st["f"], st["cx"], st["cy"], st["duR_flipped"], 0.01, st["depth_min"], 3.0, 100, 25, 1, 0)
    idx, dist, status, uvo, xyz = oracle.track_stereo_verify(prm, ref, last, uv_ref, tl, seg, pool, pool_uv)
    assert status[0] == oracle.M_DEPTH                                 # z = 386.14/71 = 5.4 m > depth_max 3
    prm = oracle.stereo_params(st["f"], st["cx"], st["cy"], st["duR_flipped"], 0.01, st["depth_min"], st["depth_max"], 100, 25, 1, 0)
    idx, dist, status, uvo, xyz = oracle.track_stereo_verify(prm, ref, last, uv_ref, tl, seg, pool, pool_uv)
    assert list(status) == [oracle.M_OK, oracle.M_OTHER, oracle.M_DISTANCE, oracle.M_DISPARITY, oracle.M_EMPTY_POOL, oracle.M_OK, oracle.M_OK]
    assert idx[0] == 1 and np.array_equal(uvo[0], np.array([329, 100], np.float32))
    assert xyz[0, 2] == st["duR_flipped"] / 71.0 and xyz[0, 0] == (1.0 / st["f"]) * xyz[0, 2] * (400.0 - st["cx"])
    # exclusive variant (stage 2): 25 is not < 25
    prm = oracle.stereo_params(st["f"], st["cx"], st["cy"], st["duR_flipped"], 0.01, st["depth_min"], st["depth_max"], 100, 25, 0, 0)
    assert oracle.track_stereo_verify(prm, ref, last, uv_ref, tl, seg, pool, pool_uv)[2][6] == oracle.M_OTHER
    # search in LEFT: the found pixel is the LEFT one
    prm = oracle.stereo_params(st["f"], st["cx"], st["cy"], st["duR_flipped"], 0.01, st["depth_min"], st["depth_max"], 100, -1, 0, 1)
    uv_r = np.array([[250, 100]] * 7, np.float32)
    idx, dist, status, uvo, xyz = oracle.track_stereo_verify(prm, ref, None, uv_r, tl, seg, pool, pool_uv)
    assert status[0] == oracle.M_OK and xyz[0, 2] == st["duR_flipped"] / 79.0


def test_cascades_cover_the_reference_outcomes(oracle, cam):
    sc = ts.Scene(n=400, seed=7)
    rec, seg = plan_of(oracle, cam, sc)
    om = oracle.OracleFundamentalMatcher(cam, sc.stereo_dict())
    s1 = [d["status"] for d in om.stage1(rec, sc.kp_size, sc.extract_one, sc.last_left, sc.last_right)]
    s2 = [d["status"] for d in om.stage2(rec, sc.kp_size, sc.detect_one, sc.extract_one, sc.last_left, sc.last_right)]
    res3 = om.epipolar(rec, sc.kp_size, sc.extract_one, sc.last_left, sc.ref_desc)
    s3 = [d["status"] for d in res3]
    assert s1.count(oracle.M_OK) > 10 and s2.count(oracle.M_OK) > 80 and s3.count(oracle.M_OK) > 80
    assert oracle.M_ORIGINAL in s3 and oracle.M_EMPTY_POOL in s3 and oracle.M_DISTANCE in s3
    # a stage-3 success lands on (or next to) the true pixel of the synthetic image and triangulates its depth
    hit = 0
    for i, d in enumerate(res3):
        if d["status"] == oracle.M_OK:
            hit += abs(d["uv_left"][0] - sc.true_uL[i]) <= 2 and abs(d["uv_left"][1] - sc.true_v[i]) <= 2
    assert hit > 0.9 * s3.count(oracle.M_OK)
