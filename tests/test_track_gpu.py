"""GPU parity of the temporal tracking schedule (svi_mapper_amd/csrc/tracker.hip through the C ABI) against the CPU
oracle (oracle/oracle_track.c): every record byte, sample, candidate, index, status and triangulated point must be
bit-identical.  PARITY UNPINNED with respect to the reference itself."""
import os

import numpy as np
import pytest

import track_scene as ts

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def torch():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("the -m gpu tests need a visible GPU (torch.cuda.is_available() is False)")
    return torch


@pytest.fixture(scope="module")
def fm(svi, torch):
    from svi_mapper_amd import temporal
    return temporal.FundamentalMatcher(temporal.StereoCamera(ts.P_LEFT, ts.P_RIGHT, ts.W, ts.H))


@pytest.fixture(scope="module")
def cam(oracle):
    return oracle.track_camera(ts.P_LEFT, ts.P_RIGHT, ts.K_INV, ts.W, ts.H)


def raw(a):
    """bytes of an array (fields of structured arrays are strided: copy first)"""
    return np.ascontiguousarray(a).view(np.uint8)


def dev(torch, a, dt=None):
    return torch.tensor(np.ascontiguousarray(a if dt is None else np.asarray(a, dt)), device="cuda")


def gpu_plan(torch, fm, sc):
    return fm.plan(sc.T_est_w2l, sc.dp_T, sc.motion_scaling, dev(torch, sc.xyz_world), dev(torch, sc.kp_size), dev(torch, sc.last_disparity),
                   dev(torch, sc.uv_reference), dev(torch, sc.dp_index))


def cpu_plan(oracle, cam, sc):
    return oracle.track_plan(cam, sc.T_est_w2l, sc.dp_T, sc.motion_scaling, sc.xyz_world, sc.kp_size, sc.last_disparity, sc.uv_reference,
                             sc.dp_index)


def test_camera_inverse_matches_fixture(fm):
    assert np.array_equal(fm.camera.K_inv, ts.K_INV)


@pytest.mark.parametrize("n,seed,ms", [(96, 3, 1.0), (1, 4, 1.0), (777, 5, 2.5), (5000, 6, 0.5)])
def test_plan_bit_exact(oracle, cam, torch, fm, n, seed, ms):
    sc = ts.Scene(n=n, seed=seed, motion_scaling=ms)
    rec, seg = cpu_plan(oracle, cam, sc)
    plan = gpu_plan(torch, fm, sc)
    got = plan.host()
    for name in got.dtype.names:
        assert np.array_equal(raw(got[name]), raw(rec[name])), name
    assert np.array_equal(plan.seg.cpu().numpy(), seg) and plan.total == seg[-1]
    if n == 96:
        g = np.load(os.path.join(HERE, "golden", "track_small.npz"))
        assert np.array_equal(plan.records.cpu().numpy(), g["records"])


def test_plan_zero_motion_empty_and_errors(oracle, cam, torch, fm, svi):
    ident = np.array([1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0], np.float64)
    xyz = np.array([[0.5, 0.1, 10.0], [0.0, 0.0, 5.0], [1.0, 1.0, -3.0], [0.0, 0.0, 0.0]])
    kp, dis = np.full(4, 7, np.float32), np.full(4, 30, np.float32)
    uvr, dpi = np.array([[600., 180.], [610., 190.], [0., 0.], [1., 1.]]), np.array([0, 5, -1, 0], np.int32)
    rec, seg = oracle.track_plan(cam, ident, ident[None], 1.0, xyz, kp, dis, uvr, dpi)
    plan = fm.plan(ident, ident[None], 1.0, dev(torch, xyz), dev(torch, kp), dev(torch, dis), dev(torch, uvr), dev(torch, dpi))
    got = plan.host()
    assert np.array_equal(got["status"], rec["status"])      # incl. the NaN projection of the origin: outside the FoV on both sides
    for name in ("uv_left", "uv_right", "s3_count", "line"):  # (NaN sign bits differ between x86 and gfx950: row 3 is left out)
        assert np.array_equal(raw(got[name][:3]), raw(rec[name][:3])), name
    assert np.isnan(got["uv_left"][3]).all() and not got["status"][3] & 3
    assert np.all(got["status"] & 4) and plan.total == 0
    empty = fm.plan(ident, np.zeros((0, 12)), 1.0, dev(torch, np.zeros((0, 3))), dev(torch, np.zeros(0, np.float32)),
                    dev(torch, np.zeros(0, np.float32)), dev(torch, np.zeros((0, 2))), dev(torch, np.zeros(0, np.int32)))
    assert empty.n == 0 and empty.total == 0 and empty.seg.cpu().tolist() == [0]
    with pytest.raises(ValueError):
        fm.plan(ident, ident[None], 1.0, dev(torch, xyz.astype(np.float32)), dev(torch, kp), dev(torch, dis), dev(torch, uvr), dev(torch, dpi))
    lib = svi.load_library()
    assert lib.svi_track_plan_dev(None, None, None, None, 0, 1.0, None, None, None, None, None, 0, None, None, None) == 1
    assert lib.svi_match_ragged_dev(fm._h, None, None, None, 4, None, None, 50, 100, None, None, None) == 1
    assert lib.svi_track_handover_dev(fm._h, 9, None, None, None, 1, None, None, None, None, None, None, None) == 1


@pytest.mark.parametrize("depth", [0, 2, 3])
def test_epipolar_samples_bit_exact(oracle, cam, torch, fm, depth):
    sc = ts.Scene(n=600, seed=8)
    rec, seg = cpu_plan(oracle, cam, sc)
    plan = gpu_plan(torch, fm, sc)
    want, want_roi = oracle.track_epipolar_samples(cam, rec, sc.kp_size, seg, depth)
    g_seg, got, got_roi = fm.epipolar_samples(plan, depth)
    assert np.array_equal(got.cpu().numpy().view(np.uint32), want.view(np.uint32))
    assert np.array_equal(got_roi.cpu().numpy().view(np.uint32), want_roi.view(np.uint32))
    # a subset in arbitrary order (the recursion re-samples only the landmarks that found nothing)
    r = np.random.default_rng(depth)
    sel = r.permutation(len(rec))[:217].astype(np.int32)
    cnt = np.diff(seg)[sel]
    seg_sel = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int32)
    want, want_roi = oracle.track_epipolar_samples(cam, rec, sc.kp_size, seg_sel, depth, sel)
    g_seg, got, got_roi = fm.epipolar_samples(plan, depth, dev(torch, sel))
    assert np.array_equal(g_seg.cpu().numpy(), seg_sel)
    assert np.array_equal(got.cpu().numpy().view(np.uint32), want.view(np.uint32))
    assert np.array_equal(got_roi.cpu().numpy().view(np.uint32), want_roi.view(np.uint32))


def ragged_case(n, seed, max_cnt=40):
    r = np.random.default_rng(seed)
    cnt = r.integers(0, max_cnt, n)
    cnt[r.random(n) < 0.05] = 0
    if n > 3:
        cnt[1] = 300  # longer than a wavefront's single pass
    seg = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int32)
    pool = r.integers(0, 256, (int(seg[-1]), 32), dtype=np.uint8)
    q = r.integers(0, 256, (n, 32), dtype=np.uint8)
    other = q.copy()
    for i in range(n):
        if cnt[i] >= 2:
            k = r.integers(0, cnt[i])
            pool[seg[i] + k] = ts.flip_bits(q[i], r.integers(0, 120), seed * 7919 + i)
            if i % 3 == 0:
                pool[seg[i] + r.integers(0, cnt[i])] = pool[seg[i] + k]      # tie: the lower index must win
            other[i] = ts.flip_bits(pool[seg[i] + k], r.integers(0, 130), seed * 104729 + i)
    active = (r.random(n) < 0.9).astype(np.uint8)
    return seg, pool, q, other, active


@pytest.mark.parametrize("n,seed", [(1, 1), (257, 2), (4096, 3)])
def test_match_ragged_bit_exact(oracle, torch, fm, n, seed):
    seg, pool, q, orig, active = ragged_case(n, seed)
    for cut, cut_o, use_orig, use_active in ((50, 100, True, True), (25, 257, False, False), (257, 60, True, False), (0, 0, True, True)):
        want = oracle.match_ragged(q, orig if use_orig else None, seg, pool, cut, cut_o, active if use_active else None)
        got = fm.get_match(dev(torch, q), dev(torch, orig) if use_orig else None, dev(torch, seg), dev(torch, pool), cut, cut_o,
                           dev(torch, active) if use_active else None)
        for a, b, name in zip(got, want, ("idx", "dist", "status")):
            assert np.array_equal(a.cpu().numpy(), b), (name, cut, cut_o)
    if n >= 257:
        st = oracle.match_ragged(q, orig, seg, pool, 50, 100, active)[2]
        assert set(st) >= {0, 1, 2, 3, 8}


@pytest.mark.parametrize("in_left", [0, 1])
def test_stereo_range_candidates_bit_exact(oracle, torch, fm, in_left):
    r = np.random.default_rng(10 + in_left)
    n = 3000
    kp = r.choice(np.array([7.0, 3.5, 10.0], np.float32), n)
    uv = np.stack([r.uniform(0, ts.W, n), r.integers(28, ts.H - 28, n)], 1).astype(np.float32)
    uv[:200, 0] = np.rint(uv[:200, 0])
    tl = np.stack([np.maximum(uv[:, 0] - r.uniform(-10, 160, n), 0), uv[:, 1] - 4 * kp], 1).astype(np.float32)
    rng = r.uniform(-5, 200, n).astype(np.float32)
    rng[:10] = 0.0
    active = (r.random(n) < 0.9).astype(np.uint8)
    w_seg, w_st, w_roi = oracle.track_stereo_range(ts.W, in_left, uv, tl, kp, rng, active)
    g_seg, g_st, g_roi, total = fm.stereo_range(in_left, dev(torch, uv), dev(torch, tl), dev(torch, kp), dev(torch, rng), dev(torch, active))
    assert np.array_equal(g_seg.cpu().numpy(), w_seg) and total == w_seg[-1]
    assert np.array_equal(g_st.cpu().numpy(), w_st)
    assert np.array_equal(g_roi.cpu().numpy().view(np.uint32), w_roi.view(np.uint32))
    assert set(w_st) == {0, 4, 8}
    want = oracle.track_stereo_candidates(in_left, kp, w_seg)
    got = fm.stereo_candidates(in_left, dev(torch, kp), g_seg, total)
    assert np.array_equal(got.cpu().numpy().view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("in_left,cut_other,incl", [(0, 25, 1), (0, 50, 0), (1, 25, 1), (1, -1, 0)])
def test_stereo_verify_bit_exact(oracle, torch, fm, in_left, cut_other, incl):
    n = 2048
    seg, pool, ref, last, active = ragged_case(n, 40 + in_left, max_cnt=90)
    r = np.random.default_rng(50 + in_left)
    kp = np.full(n, 7, np.float32)
    pool_uv = np.zeros((int(seg[-1]), 2), np.float32)
    for i in range(n):
        k = seg[i + 1] - seg[i]
        pool_uv[seg[i]:seg[i + 1], 0] = 28 + np.arange(k) + (1 if in_left else 0)
        pool_uv[seg[i]:seg[i + 1], 1] = 28
    uv_ref = np.stack([r.uniform(100, 1200, n), r.integers(40, 330, n)], 1).astype(np.float32)
    tl = uv_ref.copy()
    tl[:, 0] = np.maximum(uv_ref[:, 0] - r.uniform(20, 120, n), 0) if not in_left else uv_ref[:, 0] - 28 + r.uniform(-3, 3, n)
    tl[:, 1] -= 28
    tl = tl.astype(np.float32)
    st = ts.Scene(n=4, seed=1).stereo_dict()
    for depth_max in (st["depth_max"], 12.0):
        prm_o = oracle.stereo_params(st["f"], st["cx"], st["cy"], st["duR_flipped"], 0.01, st["depth_min"], depth_max, 100, cut_other, incl, in_left)
        want = oracle.track_stereo_verify(prm_o, ref, last if cut_other >= 0 else None, uv_ref, tl, seg, pool, pool_uv, active)
        prm = fm.stereo_params(in_left, cut_other, incl)
        prm.depth_max = depth_max
        got = fm.stereo_verify(prm, dev(torch, ref), dev(torch, last) if cut_other >= 0 else None, dev(torch, uv_ref), dev(torch, tl),
                               dev(torch, seg), dev(torch, pool), dev(torch, pool_uv), dev(torch, active))
        for a, b, name in zip(got, want, ("idx", "dist", "status", "uv_other", "xyz")):
            a = a.cpu().numpy()
            assert np.array_equal(raw(a), raw(b)), (name, depth_max)
    seen = set(want[2])
    assert {0, 1, 2, 6, 8} <= seen
    if cut_other >= 0:
        assert 7 in seen
    if in_left:
        assert 5 in seen


@pytest.mark.parametrize("mode", [0, 1, 2, 3, 4])
def test_handover_bit_exact(oracle, cam, torch, fm, mode):
    sc = ts.Scene(n=500, seed=20 + mode)
    rec, seg = cpu_plan(oracle, cam, sc)
    plan = gpu_plan(torch, fm, sc)
    r = np.random.default_rng(mode)
    sel = r.permutation(len(rec))[:333].astype(np.int32)
    cnt = r.integers(0, 9, len(sel))
    pseg = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int32)
    pool_uv = r.uniform(0, 120, (int(pseg[-1]), 2)).astype(np.float32)
    idx = np.array([r.integers(0, c) if c and r.random() < 0.8 else -1 for c in cnt], np.int32)
    roi = r.uniform(0, 900, (len(sel), 4)).astype(np.float32)
    want = oracle.track_handover(mode, rec, sc.kp_size, sel, pseg, pool_uv, idx, roi)
    got = fm.handover(mode, plan, dev(torch, sel), dev(torch, pseg), dev(torch, pool_uv), dev(torch, idx), dev(torch, roi))
    for a, b, name in zip(got, want, ("uv_ref", "topleft", "ok")):
        assert np.array_equal(raw(a.cpu().numpy()), raw(b)), name
    if mode >= 2:
        assert 0 in want[2] and 1 in want[2]


def check_stage(res, want, n):
    status = res.status.cpu().numpy()
    assert np.array_equal(status, np.array([d["status"] for d in want], np.int32))
    uvl, uvr, xyz = res.uv_left.cpu().numpy(), res.uv_right.cpu().numpy(), res.xyz_left.cpu().numpy()
    dl, dr = res.desc_left.cpu().numpy(), res.desc_right.cpu().numpy()
    for i, d in enumerate(want):
        if d["status"] == 0:
            assert np.array_equal(uvl[i], d["uv_left"]) and np.array_equal(uvr[i], d["uv_right"]), i
            assert np.array_equal(xyz[i], d["xyz"]), i
            assert np.array_equal(dl[i], d["desc_left"]) and np.array_equal(dr[i], d["desc_right"]), i
    return status


def test_cascades_match_the_per_landmark_replay(oracle, cam, torch, fm):
    """the batched stage 1 / 2 / 3 cascades against the reference's one-landmark-at-a-time try/catch flow"""
    sc = ts.Scene(n=400, seed=7)
    rec, seg = cpu_plan(oracle, cam, sc)
    plan = gpu_plan(torch, fm, sc)
    om = oracle.OracleFundamentalMatcher(cam, sc.stereo_dict())
    ext, det = sc.make_extractor(torch, "cuda"), sc.make_detector(torch, "cuda")
    ll, lr, rf = dev(torch, sc.last_left), dev(torch, sc.last_right), dev(torch, sc.ref_desc)
    s1 = check_stage(fm.track_stage1(plan, ext, ll, lr), om.stage1(rec, sc.kp_size, sc.extract_one, sc.last_left, sc.last_right), sc.n)
    s2 = check_stage(fm.track_stage2(plan, det, ext, ll, lr), om.stage2(rec, sc.kp_size, sc.detect_one, sc.extract_one, sc.last_left, sc.last_right), sc.n)
    s3 = check_stage(fm.track_epipolar(plan, ext, ll, rf), om.epipolar(rec, sc.kp_size, sc.extract_one, sc.last_left, sc.ref_desc), sc.n)
    assert (s1 == 0).sum() > 10 and (s2 == 0).sum() > 80 and (s3 == 0).sum() > 80
    g = np.load(os.path.join(HERE, "golden", "track_small.npz"))
    sc = ts.Scene(n=96, seed=3)
    plan = gpu_plan(torch, fm, sc)
    ext = sc.make_extractor(torch, "cuda")
    res = fm.track_epipolar(plan, ext, dev(torch, sc.last_left), dev(torch, sc.ref_desc))
    assert np.array_equal(res.status.cpu().numpy(), g["s3_status"])
    ok = g["s3_status"] == 0
    assert np.array_equal(res.xyz_left.cpu().numpy()[ok], g["s3_xyz"][ok])
    assert np.array_equal(res.uv_left.cpu().numpy()[ok], g["s3_uv_left"][ok])


def test_track_manual_and_new_landmarks(oracle, cam, torch, fm):
    """trackManual = stage 1 -> 2 -> 3 per landmark; addNewLandmarks = stereo partner of fresh key points"""
    sc = ts.Scene(n=350, seed=17)
    rec, seg = cpu_plan(oracle, cam, sc)
    plan = gpu_plan(torch, fm, sc)
    om = oracle.OracleFundamentalMatcher(cam, sc.stereo_dict())
    ext, det = sc.make_extractor(torch, "cuda"), sc.make_detector(torch, "cuda")
    ll, lr, rf = dev(torch, sc.last_left), dev(torch, sc.last_right), dev(torch, sc.ref_desc)
    got = fm.track_manual(plan, det, ext, ll, lr, rf)
    want = om.manual(rec, sc.kp_size, sc.detect_one, sc.extract_one, sc.last_left, sc.last_right, sc.ref_desc)
    check_stage(got, want, sc.n)
    stage = got.stage.cpu().numpy()
    assert np.array_equal(stage, np.array([d["stage"] for d in want], np.int8))
    assert {1, 2, 3} <= set(stage.tolist())
    # fresh key points: the true pixels of the landmarks (where the synthetic image carries their descriptor) + distractors
    inside = (sc.true_uL >= 80) & (sc.true_uL < ts.W - 30) & (sc.true_v >= 30) & (sc.true_v < ts.H - 30)
    uv = np.stack([sc.true_uL[inside], sc.true_v[inside]], 1).astype(np.float32)
    uv = np.concatenate([uv, np.array([[10, 100], [700.5, 200], [ts.W - 1, 50]], np.float32)])
    size = np.full(len(uv), 7, np.float32)
    desc = np.concatenate([sc.cur_left[inside], sc.describe(0, [10, 700, ts.W - 1], [100, 200, 50])])
    got = fm.add_new_landmarks(ext, dev(torch, uv), dev(torch, size), dev(torch, desc))
    want = om.new_landmarks(sc.extract_one, uv, size, desc)
    st = check_stage(got, want, len(uv))
    assert (st == 0).sum() > 50 and (st != 0).sum() > 0
    z = got.xyz_left.cpu().numpy()[st == 0, 2]
    assert np.all(z > 0)


def test_many_random_frames(oracle, cam, torch, fm):
    """24 further random frames (different motion scalings, pose errors, key point sizes): plan, both sampling depths and the
    stage-3 cascade stay bit-identical"""
    for seed in range(100, 124):
        r = np.random.default_rng(seed)
        sc = ts.Scene(n=int(r.integers(50, 400)), seed=seed, pose_error=float(r.uniform(0, 0.05)),
                      kp_sizes=(7.0, float(r.choice([7.0, 5.5, 9.25]))), motion_scaling=float(r.uniform(0.2, 3.0)))
        rec, seg = cpu_plan(oracle, cam, sc)
        plan = gpu_plan(torch, fm, sc)
        assert np.array_equal(plan.records.cpu().numpy(), rec.view(np.uint8).reshape(len(rec), -1)), seed
        assert np.array_equal(plan.seg.cpu().numpy(), seg), seed
        for depth in (0, 2):
            want, want_roi = oracle.track_epipolar_samples(cam, rec, sc.kp_size, seg, depth)
            _, got, got_roi = fm.epipolar_samples(plan, depth)
            assert np.array_equal(raw(got.cpu().numpy()), raw(want)) and np.array_equal(raw(got_roi.cpu().numpy()), raw(want_roi)), (seed, depth)
        if seed % 4 == 0:
            om = oracle.OracleFundamentalMatcher(cam, sc.stereo_dict())
            ext = sc.make_extractor(torch, "cuda")
            check_stage(fm.track_epipolar(plan, ext, dev(torch, sc.last_left), dev(torch, sc.ref_desc)),
                        om.epipolar(rec, sc.kp_size, sc.extract_one, sc.last_left, sc.ref_desc), sc.n)


def test_nothing_to_track(oracle, cam, torch, fm):
    """every landmark inactive / no landmark at all: the cascades run through with empty pools"""
    sc = ts.Scene(n=64, seed=4)
    plan = gpu_plan(torch, fm, sc)
    ext, det = sc.make_extractor(torch, "cuda"), sc.make_detector(torch, "cuda")
    ll, lr, rf = dev(torch, sc.last_left), dev(torch, sc.last_right), dev(torch, sc.ref_desc)
    off = torch.zeros(sc.n, dtype=torch.uint8, device="cuda")
    for res in (fm.track_stage1(plan, ext, ll, lr, off), fm.track_stage2(plan, det, ext, ll, lr, off), fm.track_epipolar(plan, ext, ll, rf, off),
                fm.track_manual(plan, det, ext, ll, lr, rf, off)):
        assert (res.status == 8).all()
    # landmarks that match nothing (random descriptors): the stereo stage gets no work at all
    junk = dev(torch, np.random.default_rng(0).integers(0, 256, (sc.n, 32), dtype=np.uint8))
    res = fm.track_epipolar(plan, ext, junk, junk)
    assert (res.status != 0).all()
    e = lambda *shape, dt=torch.float32: torch.zeros(shape, dtype=dt, device="cuda")  # noqa: E731
    res = fm.add_new_landmarks(ext, e(0, 2), e(0), e(0, 32, dt=torch.uint8))
    assert res.status.numel() == 0


def test_large_frame_properties(torch, fm):
    """200k landmarks (far beyond a real frame): segment table is a scan of the counts, every sample lies inside its
    ROI, ragged matching of a pool against itself finds itself"""
    sc_n = 200_000
    r = np.random.default_rng(99)
    base = ts.Scene(n=2000, seed=9)
    rep = sc_n // base.n
    xyz = np.tile(base.xyz_world, (rep, 1)) + r.normal(0, 0.01, (sc_n, 3))
    kp, dis = np.tile(base.kp_size, rep), np.tile(base.last_disparity, rep)
    uvr, dpi = np.tile(base.uv_reference, (rep, 1)), np.tile(base.dp_index, rep)
    plan = fm.plan(base.T_est_w2l, base.dp_T, 1.0, dev(torch, xyz), dev(torch, kp), dev(torch, dis), dev(torch, uvr), dev(torch, dpi))
    rec = plan.host()
    cnt = np.where(rec["status"] & 64, rec["s3_count"], 0)
    seg = plan.seg.cpu().numpy()
    assert np.array_equal(seg, np.concatenate([[0], np.cumsum(cnt)]).astype(np.int32)) and plan.total == cnt.sum()
    g_seg, samples, roi = fm.epipolar_samples(plan, 0)
    s, roi = samples.cpu().numpy(), roi.cpu().numpy()
    owner = np.repeat(np.arange(sc_n), cnt)
    assert np.all(s[:, 0] >= -1e-3) and np.all(s[:, 1] >= -1e-3)
    inside = (s[:, 0] <= roi[owner, 2] + 1e-3) | (roi[owner, 0] + roi[owner, 2] >= ts.W - 1e-3)
    assert inside.mean() > 0.999
    # self-match: query i = some row of its own segment -> distance 0 at the first identical row
    nq = 50_000
    c = r.integers(1, 20, nq)
    pseg = np.concatenate([[0], np.cumsum(c)]).astype(np.int32)
    pool = r.integers(0, 256, (int(pseg[-1]), 32), dtype=np.uint8)
    pick = (pseg[:-1] + r.integers(0, c)).astype(np.int64)
    q = pool[pick]
    idx, dist, st = fm.get_match(dev(torch, q), None, dev(torch, pseg), dev(torch, pool), 1)
    idx, dist, st = idx.cpu().numpy(), dist.cpu().numpy(), st.cpu().numpy()
    assert np.all(st == 0) and np.all(dist == 0) and np.all(idx <= pick - pseg[:-1])
    assert np.array_equal(pool[pseg[:-1] + idx], q)


def test_track_epipolar_without_motion_falls_back_to_stage2(oracle, cam, torch, fm):
    """trackEpipolar (:841-1290): a detection point that has not moved since (|t|^2 == 0) has no epipolar line - its landmarks are
    searched by stage 2 instead, everything else along its line"""
    sc = ts.Scene(n=300, seed=29)
    # detection point 0 sits exactly at the current pose: the relative transform is the identity
    sc.dp_T = np.array(sc.dp_T, np.float64).copy()
    sc.dp_T[0] = ts.inv12(sc.T_est_w2l)
    # ... bit for bit: t_rel = ((R0 t0 + R1 t1) + R2 t2) + t_est in the reference's operand order (:800-806) must be exactly 0
    A, td = sc.T_est_w2l.copy(), sc.dp_T[0][9:]
    for i in range(3):
        A[9 + i] = -(((A[3 * i] * td[0]) + (A[3 * i + 1] * td[1])) + (A[3 * i + 2] * td[2]))
    sc.T_est_w2l = A
    rec, seg = cpu_plan(oracle, cam, sc)
    still = (rec["status"] & 4) != 0
    assert still.sum() > 20 and (~still).sum() > 20
    plan = gpu_plan(torch, fm, sc)
    om = oracle.OracleFundamentalMatcher(cam, sc.stereo_dict())
    ext, det = sc.make_extractor(torch, "cuda"), sc.make_detector(torch, "cuda")
    ll, lr, rf = dev(torch, sc.last_left), dev(torch, sc.last_right), dev(torch, sc.ref_desc)
    got = fm.track_epipolar(plan, ext, ll, rf, detector=det, last_desc_right=lr)
    s3 = om.epipolar(rec, sc.kp_size, sc.extract_one, sc.last_left, sc.ref_desc)
    s2 = om.stage2(rec, sc.kp_size, sc.detect_one, sc.extract_one, sc.last_left, sc.last_right)
    want = [b if still[i] else a for i, (a, b) in enumerate(zip(s3, s2))]
    st = check_stage(got, want, sc.n)
    stage = got.stage.cpu().numpy()
    assert set(stage[still & (st == 0)].tolist()) == {2} and set(stage[~still & (st == 0)].tolist()) == {3}
    # without a detector the landmarks of the still detection point are simply not searched
    got = fm.track_epipolar(plan, ext, ll, rf)
    assert (got.status.cpu().numpy()[still] == 8).all()


def test_cascade_entry_points_reject_bad_calls(torch, fm, svi):
    sc = ts.Scene(n=32, seed=2)
    plan_old = gpu_plan(torch, fm, sc)
    plan = gpu_plan(torch, fm, sc)
    ext = sc.make_extractor(torch, "cuda")
    ll, lr = dev(torch, sc.last_left), dev(torch, sc.last_right)
    with pytest.raises(ValueError):
        fm.track_stage1(plan_old, ext, ll, lr)          # not the frame the tracker holds
    with pytest.raises(svi.SviError) as e:
        fm.track_stage2(plan, None, ext, ll, lr)        # stage 2 without a detector
    assert e.value.status == 4
    lib = svi.load_library()
    assert lib.svi_track_manual(None, None, None) == 1
    assert lib.svi_tracker_create(None, None, None) == 1
    # an extractor that fails must fail the call, not crash it
    def broken(side, roi, seg, kp_uv):
        raise RuntimeError("extractor failure (expected in this test)")
    with pytest.raises(svi.SviError):
        fm.track_stage1(plan, broken, ll, lr)
    assert (fm.track_stage1(plan, ext, ll, lr).status != 8).any()   # the handle is still usable
