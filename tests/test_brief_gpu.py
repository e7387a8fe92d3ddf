"""GPU parity of the BRIEF extraction (svi_mapper_amd/csrc/brief.hip) against oracle/oracle_brief.c: integral image, kept
key points and descriptors are bit-identical; then a whole frame is tracked on the device with the GPU extractor plugged
into the cascades and compared with the per-landmark replay using the CPU extractor.  PARITY UNPINNED vs OpenCV."""
import numpy as np
import pytest

import brief_case
import track_scene as ts

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("the -m gpu tests need a visible GPU (torch.cuda.is_available() is False)")
    return torch


@pytest.fixture(scope="module")
def brief(svi, torch):
    from svi_mapper_amd import temporal
    b = temporal.BriefExtractor(brief_case.pattern())
    yield b
    b.close()


@pytest.mark.parametrize("h,w,seed", [(376, 1241, 1), (57, 57, 2), (480, 640, 3), (1, 1, 4), (200, 1000, 5)])
def test_integral_bit_exact(oracle, torch, brief, h, w, seed):
    img = brief_case.image(max(h, 8), max(w, 8), seed)[:h, :w]
    brief.set_image("left", torch.tensor(np.ascontiguousarray(img), device="cuda"))
    assert np.array_equal(brief.integral("left").cpu().numpy(), oracle.brief_integral(img))
    # strided rows (a ROI view of a larger frame)
    big = brief_case.image(h + 16, w + 24, seed + 10)
    t = torch.tensor(big, device="cuda")[5:5 + h, 7:7 + w]
    brief.set_image("right", t)
    assert np.array_equal(brief.integral("right").cpu().numpy(), oracle.brief_integral(big[5:5 + h, 7:7 + w]))


@pytest.mark.parametrize("n,seed,max_kp", [(64, 3, 60), (1, 4, 200), (3000, 5, 40), (7, 6, 700)])
def test_compute_bit_exact(oracle, torch, brief, n, seed, max_kp):
    img = brief_case.image(376, 1241, seed)
    brief.set_image("left", torch.tensor(img, device="cuda"))
    roi, seg, kp = brief_case.pools(376, 1241, max(n, 4), seed, max_kp)
    roi, seg = roi[:n], seg[:n + 1]
    kp = kp[:seg[-1]]
    want = oracle.brief_compute(oracle.brief_integral(img), brief_case.pattern(), roi, seg, kp)
    d = lambda a: torch.tensor(np.ascontiguousarray(a), device="cuda")  # noqa: E731
    got = brief("left", d(roi), d(seg), d(kp))
    assert np.array_equal(got[0].cpu().numpy(), want[0])
    assert np.array_equal(got[1].cpu().numpy().view(np.uint32), want[1].view(np.uint32))
    assert np.array_equal(got[2].cpu().numpy(), want[2])
    if n >= 64:
        assert want[0][-1] > 50 and (np.unpackbits(want[2]).mean() - 0.5) ** 2 < 0.01     # descriptors are not degenerate


def test_errors(brief, svi, torch):
    lib = svi.load_library()
    import ctypes as C
    bad = np.full(1024, 30, np.int8)
    h = C.c_void_p()
    assert lib.svi_brief_create(brief.matcher._h, bad.ctypes.data_as(C.c_void_p), C.byref(h)) == 1
    from svi_mapper_amd import temporal
    fresh = temporal.BriefExtractor(brief_case.pattern(), matcher=brief.matcher)
    z = torch.zeros((1, 4), dtype=torch.float32, device="cuda")
    with pytest.raises(svi.SviError) as e:
        fresh("left", z, torch.zeros(2, dtype=torch.int32, device="cuda"), torch.zeros((0, 2), dtype=torch.float32, device="cuda"))
    assert e.value.status == 4                                        # no image set
    with pytest.raises(ValueError):
        fresh.set_image("left", torch.zeros((4, 4), dtype=torch.float32, device="cuda"))
    fresh.close()


def test_frame_tracked_on_the_device(oracle, torch, brief):
    """stage 1, 2 and 3 with the GPU extractor against the per-landmark replay with the CPU extractor on real (synthetic)
    images: the landmark's previous descriptors are what the extractor sees at its true pixels, a few bits flipped"""
    from svi_mapper_amd import temporal
    sc = ts.Scene(n=500, seed=21, kp_sizes=(7.0,))
    left, right = brief_case.image(ts.H, ts.W, 31), brief_case.image(ts.H, ts.W, 32)
    SL, SR = oracle.brief_integral(left), oracle.brief_integral(right)
    pat = brief_case.pattern()

    def cpu_extract(side, roi, kp_uv):
        seg_out, kp_out, desc = oracle.brief_compute(SL if side == "left" else SR, pat, np.asarray(roi, np.float32)[None], [0, len(kp_uv)], kp_uv)
        return kp_out, desc

    # previous descriptors: extracted at the true pixels of this frame (full-frame ROI), a few bits flipped
    full = np.array([0, 0, ts.W, ts.H], np.float32)
    r = np.random.default_rng(5)
    last_l = r.integers(0, 256, (sc.n, 32), dtype=np.uint8)
    last_r = last_l.copy()
    for i in range(sc.n):
        for side, uu, arr in (("left", sc.true_uL[i], last_l), ("right", sc.true_uR[i], last_r)):
            if 28 <= uu < ts.W - 28 and 28 <= sc.true_v[i] < ts.H - 28:
                k, dsc = cpu_extract(side, full, np.array([[uu, sc.true_v[i]]], np.float32))
                arr[i] = ts.flip_bits(dsc[0], int(r.integers(0, 12)), 100 + i)
    ref_l = np.stack([ts.flip_bits(last_l[i], int(r.integers(0, 10)), 900 + i) for i in range(sc.n)])

    brief.set_image("left", torch.tensor(left, device="cuda"))
    brief.set_image("right", torch.tensor(right, device="cuda"))
    fm = temporal.FundamentalMatcher(temporal.StereoCamera(ts.P_LEFT, ts.P_RIGHT, ts.W, ts.H), matcher=brief.matcher)
    d = lambda a: torch.tensor(np.ascontiguousarray(a), device="cuda")  # noqa: E731
    plan = fm.plan(sc.T_est_w2l, sc.dp_T, sc.motion_scaling, d(sc.xyz_world), d(sc.kp_size), d(sc.last_disparity), d(sc.uv_reference), d(sc.dp_index))
    cam = oracle.track_camera(ts.P_LEFT, ts.P_RIGHT, ts.K_INV, ts.W, ts.H)
    rec, seg = oracle.track_plan(cam, sc.T_est_w2l, sc.dp_T, sc.motion_scaling, sc.xyz_world, sc.kp_size, sc.last_disparity, sc.uv_reference,
                                 sc.dp_index)
    om = oracle.OracleFundamentalMatcher(cam, sc.stereo_dict())
    det = sc.make_detector(torch, "cuda")
    from test_track_gpu import check_stage
    s1 = check_stage(fm.track_stage1(plan, brief, d(last_l), d(last_r)), om.stage1(rec, sc.kp_size, cpu_extract, last_l, last_r), sc.n)
    s2 = check_stage(fm.track_stage2(plan, det, brief, d(last_l), d(last_r)), om.stage2(rec, sc.kp_size, sc.detect_one, cpu_extract, last_l, last_r), sc.n)
    s3 = check_stage(fm.track_epipolar(plan, brief, d(last_l), d(ref_l)), om.epipolar(rec, sc.kp_size, cpu_extract, last_l, ref_l), sc.n)
    assert (s1 == 0).sum() + (s2 == 0).sum() + (s3 == 0).sum() > 30
