"""The C++ facade header compiles against the C ABI with a plain g++ and links to the library."""
import os
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SRC = r'''
#include "svi_g2o_optimizer.hpp"
#include <cstdio>
int main() {
    std::printf("%d %s\n", svi_version(), svi_status_string(0));
    try { svi::BundleAdjusterGPU ba(718.856, 718.856, 607.1928, 185.2157, 0.54); }
    catch (const std::exception& e) { std::printf("no device: %s\n", e.what()); }
    return 0;
}
'''


def test_facade_compiles_and_links(svi):
    from svi_mapper_amd import _capi
    libdir = os.path.dirname(_capi.LIB_PATH)
    hip = "/opt/rocm/lib"
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "t.cpp")
        open(c, "w").write(SRC)
        exe = os.path.join(d, "t")
        subprocess.check_call(["g++", "-std=c++17", "-I", os.path.join(ROOT, "include"), c, "-o", exe, "-L", libdir, "-lsvi_hot",
                               "-L", hip, "-lamdhip64", "-Wl,-rpath," + libdir, "-Wl,-rpath," + hip])
        out = subprocess.check_output([exe]).decode()
    assert out.startswith("100 ok")


REPLAY = r'''
// A maintainer-side program: plain C++, the stock HIP runtime, no Python.  Replays a saved graph through the facade
// (what Cg2oOptimizer::optimize does after the graph is built) and writes the optimised graph back.
#include "svi_g2o_optimizer.hpp"
#include <cstdio>
int main(int argc, char** argv) {
    if (argc < 3) return 2;
    try {
        svi::BundleAdjusterGPU ba(718.856, 718.856, 607.1928, 185.2157, 0.54);
        ba.load(argv[1]);
        uint64_t executed = 0;
        const uint64_t nominal = ba.optimizeUnLimited(&executed);
        ba.save(argv[2]);
        std::printf("%llu %llu %.17g\n", (unsigned long long)nominal, (unsigned long long)executed, ba.chi2());
    } catch (const std::exception& e) { std::printf("error: %s\n", e.what()); return 1; }
    return 0;
}
'''


import pytest  # noqa: E402


@pytest.mark.gpu
def test_cpp_program_replays_a_graph(svi, tmp_path):
    """g++-built program against libsvi_hot.so + /opt/rocm libamdhip64: same result as the Python-driven library"""
    import numpy as np
    from svi_mapper_amd import _capi, synth
    prob = synth.make_ba_problem(12, 300, 2200, seed=7)
    cam = prob["cam"]
    a = svi.BundleAdjuster(cam["fx"], cam["fy"], cam["cx"], cam["cy"], cam["baseline_m"])
    synth.build_ba_graph(a, prob)
    f_in, f_out = tmp_path / "in.g2o", tmp_path / "out.g2o"
    a.save_g2o(f_in)
    libdir = os.path.dirname(_capi.LIB_PATH)
    hip = "/opt/rocm/lib"
    c = tmp_path / "replay.cpp"
    c.write_text(REPLAY)
    exe = tmp_path / "replay"
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"), str(c), "-o", str(exe), "-L", libdir, "-lsvi_hot",
                           "-L", hip, "-lamdhip64", "-Wl,-rpath," + libdir, "-Wl,-rpath," + hip])
    out = subprocess.check_output([str(exe), str(f_in), str(f_out)]).decode().split()
    # the same graph through the Python harness (loaded from the same text file: identical inputs)
    b = svi.BundleAdjuster(1, 1, 0, 0, cam["baseline_m"])
    b.load_g2o(f_in)
    b.initialize()
    nominal, executed = b.optimize_until()
    assert (int(out[0]), int(out[1])) == (nominal, executed)
    assert abs(float(out[2]) - b.chi2()[0]) <= 1e-9 * b.chi2()[0]
    r = svi.BundleAdjuster(1, 1, 0, 0, cam["baseline_m"])
    r.load_g2o(f_out)
    assert np.abs(r.get_landmarks()[1] - b.get_landmarks()[1]).max() < 1e-6


TRACK = r'''
// A maintainer-side program: plain C++ + the stock HIP runtime, no Python.  Tracks ONE frame the way CTrackerGT does
// (CFundamentalMatcher::trackManual: stage 1 -> 2 -> 3 per landmark) through include/svi_fundamental_matcher.hpp, with the
// built-in BRIEF extractor and a detector callback written here (the corners of the frame come from the input file, the
// callback cuts them to the search rectangles like cv::FeatureDetector::detect( image( rect ) ) would).
#include "svi_fundamental_matcher.hpp"
#include <cmath>
#include <cstdio>
#include <cstring>
struct Corners { std::vector<float> pts[2]; };
static int detect(void* user, int side, const float* rect, const uint8_t* active, int n, int64_t cap, int32_t* seg_out, float* kp_out,
                  int64_t* total_out, void* stream) {
    const Corners* c = static_cast<const Corners*>(user);
    hipStream_t st = static_cast<hipStream_t>(stream);
    std::vector<float> r((size_t)4 * n); std::vector<uint8_t> a(n);
    if (hipMemcpyAsync(r.data(), rect, sizeof(float) * 4 * n, hipMemcpyDeviceToHost, st) != hipSuccess) return 1;
    if (hipMemcpyAsync(a.data(), active, n, hipMemcpyDeviceToHost, st) != hipSuccess) return 1;
    if (hipStreamSynchronize(st) != hipSuccess) return 1;
    std::vector<int32_t> seg(n + 1, 0); std::vector<float> kp;
    const std::vector<float>& p = c->pts[side];
    for (int i = 0; i < n; ++i) {
        if (a[i]) {
            const float u0 = std::floor(r[4 * i]), v0 = std::floor(r[4 * i + 1]), u1 = std::floor(r[4 * i + 2]), v1 = std::floor(r[4 * i + 3]);
            for (size_t k = 0; k + 1 < p.size(); k += 2)
                if (p[k] >= u0 && p[k] < u1 && p[k + 1] >= v0 && p[k + 1] < v1) { kp.push_back(p[k] - u0); kp.push_back(p[k + 1] - v0); }
        }
        seg[i + 1] = (int32_t)(kp.size() / 2);
    }
    if ((int64_t)(kp.size() / 2) > cap) return 2;
    if (hipMemcpyAsync(seg_out, seg.data(), sizeof(int32_t) * (n + 1), hipMemcpyHostToDevice, st) != hipSuccess) return 1;
    if (!kp.empty() && hipMemcpyAsync(kp_out, kp.data(), sizeof(float) * kp.size(), hipMemcpyHostToDevice, st) != hipSuccess) return 1;
    if (hipStreamSynchronize(st) != hipSuccess) return 1;
    *total_out = (int64_t)(kp.size() / 2);
    return 0;
}
template <class T> static void rd(FILE* f, std::vector<T>& v, size_t n) { v.resize(n); if (n && fread(v.data(), sizeof(T), n, f) != n) throw std::runtime_error("short input"); }
template <class T> static void wr(FILE* f, const std::vector<T>& v) { if (!v.empty()) fwrite(v.data(), sizeof(T), v.size(), f); }
int main(int argc, char** argv) {
    if (argc < 3) return 2;
    try {
        FILE* f = fopen(argv[1], "rb");
        if (!f) return 2;
        std::vector<int32_t> hd; rd(f, hd, 6);
        const int W = hd[0], H = hd[1], n = hd[2], n_dp = hd[3], nc0 = hd[4], nc1 = hd[5];
        std::vector<double> cam, T, dpT, ms;
        rd(f, cam, 33); rd(f, T, 12); rd(f, dpT, (size_t)12 * n_dp); rd(f, ms, 1);
        svi::FrameLandmarks lm;
        rd(f, lm.xyz_world, (size_t)3 * n); rd(f, lm.uv_reference, (size_t)2 * n); rd(f, lm.kp_size, n); rd(f, lm.last_disparity, n);
        rd(f, lm.dp_index, n); rd(f, lm.last_desc_left, (size_t)32 * n); rd(f, lm.last_desc_right, (size_t)32 * n); rd(f, lm.ref_desc_left, (size_t)32 * n);
        std::vector<int8_t> pattern; rd(f, pattern, 1024);
        std::vector<uint8_t> left, right; rd(f, left, (size_t)W * H); rd(f, right, (size_t)W * H);
        Corners c; rd(f, c.pts[0], (size_t)2 * nc0); rd(f, c.pts[1], (size_t)2 * nc1);
        fclose(f);
        svi_track_camera sc{};
        memcpy(sc.P_left, cam.data(), 96); memcpy(sc.P_right, cam.data() + 12, 96); memcpy(sc.K_inv, cam.data() + 24, 72);
        sc.width = W; sc.height = H;
        svi::FundamentalMatcherGPU fm(sc);
        fm.setImages(left.data(), right.data(), W, H, pattern.data());
        if (svi_tracker_set_detector(fm.handle(), detect, &c, 1 << 20) != SVI_OK) return 1;
        fm.planFrame(T.data(), dpT, ms[0], lm);
        svi::TrackOutcome o = fm.trackManual();
        FILE* g = fopen(argv[2], "wb");
        wr(g, o.status); wr(g, o.stage); wr(g, o.uv_left); wr(g, o.uv_right); wr(g, o.xyz_left); wr(g, o.desc_left); wr(g, o.desc_right);
        // getPoseStereoPosit on the same frame: last pose = estimate, no IMU translation
        const double t0[3] = {0, 0, 0};
        svi_posit_result pr{};
        try { fm.getPoseStereoPosit(T.data(), t0, T.data(), nullptr, nullptr, &pr); } catch (const std::exception&) {}
        fwrite(&pr, sizeof(pr), 1, g);
        fclose(g);
        int ok = 0;
        for (int s : o.status) ok += s == 0;
        std::printf("%d %d\n", n, ok);
    } catch (const std::exception& e) { std::printf("error: %s\n", e.what()); return 1; }
    return 0;
}
'''


@pytest.mark.gpu
def test_cpp_program_tracks_a_frame(svi, oracle, tmp_path):
    """A g++-built program tracks one synthetic frame through include/svi_fundamental_matcher.hpp (trackManual with the built-in
    BRIEF extractor and a detector callback written in C++, then getPoseStereoPosit): bit-identical to the Python-driven
    library on the same inputs, and both equal the per-landmark replay of the reference's cascade on the CPU"""
    import ctypes as C

    import numpy as np
    import torch

    import brief_case
    import track_scene as ts
    from svi_mapper_amd import _capi, temporal
    from test_track_gpu import check_stage
    sc = ts.Scene(n=400, seed=33, kp_sizes=(7.0,))
    left, right = brief_case.image(ts.H, ts.W, 41), brief_case.image(ts.H, ts.W, 42)
    SL, SR = oracle.brief_integral(left), oracle.brief_integral(right)
    pat = brief_case.pattern()

    def cpu_extract(side, roi, kp_uv):
        _, kp_out, desc = oracle.brief_compute(SL if side == "left" else SR, pat, np.asarray(roi, np.float32)[None], [0, len(kp_uv)], kp_uv)
        return kp_out, desc

    full = np.array([0, 0, ts.W, ts.H], np.float32)
    r = np.random.default_rng(6)
    last_l = r.integers(0, 256, (sc.n, 32), dtype=np.uint8)
    last_r = last_l.copy()
    for i in range(sc.n):
        for side, uu, arr in (("left", sc.true_uL[i], last_l), ("right", sc.true_uR[i], last_r)):
            if 28 <= uu < ts.W - 28 and 28 <= sc.true_v[i] < ts.H - 28:
                _, dsc = cpu_extract(side, full, np.array([[uu, sc.true_v[i]]], np.float32))
                arr[i] = ts.flip_bits(dsc[0], int(r.integers(0, 12)), 100 + i)
    ref_l = np.stack([ts.flip_bits(last_l[i], int(r.integers(0, 10)), 900 + i) for i in range(sc.n)])
    f_in, f_out = tmp_path / "frame.bin", tmp_path / "out.bin"
    with open(f_in, "wb") as f:
        np.array([ts.W, ts.H, sc.n, len(sc.dp_T), len(sc.corners[0]), len(sc.corners[1])], np.int32).tofile(f)
        np.concatenate([np.asarray(ts.P_LEFT, np.float64).ravel(), np.asarray(ts.P_RIGHT, np.float64).ravel(), np.asarray(ts.K_INV, np.float64).ravel()]).tofile(f)
        np.asarray(sc.T_est_w2l, np.float64).tofile(f)
        np.asarray(sc.dp_T, np.float64).tofile(f)
        np.array([sc.motion_scaling], np.float64).tofile(f)
        for a, dt in ((sc.xyz_world, np.float64), (sc.uv_reference, np.float64), (sc.kp_size, np.float32), (sc.last_disparity, np.float32),
                      (sc.dp_index, np.int32), (last_l, np.uint8), (last_r, np.uint8), (ref_l, np.uint8), (pat, np.int8), (left, np.uint8),
                      (right, np.uint8), (sc.corners[0], np.float32), (sc.corners[1], np.float32)):
            np.ascontiguousarray(a, dt).tofile(f)
    libdir = os.path.dirname(_capi.LIB_PATH)
    hip = "/opt/rocm/lib"
    c = tmp_path / "track.cpp"
    c.write_text(TRACK)
    exe = tmp_path / "track"
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-D__HIP_PLATFORM_AMD__", "-I", "/opt/rocm/include", "-I", os.path.join(ROOT, "include"),
                           str(c), "-o", str(exe), "-L", libdir, "-lsvi_hot", "-L", hip, "-lamdhip64", "-Wl,-rpath," + libdir, "-Wl,-rpath," + hip])
    out = subprocess.check_output([str(exe), str(f_in), str(f_out)]).decode().split()
    assert int(out[0]) == sc.n and int(out[1]) > 30, out
    n = sc.n
    raw = open(f_out, "rb").read()
    off = 0

    def take(dt, count):
        nonlocal off
        a = np.frombuffer(raw, dt, count, off)
        off += a.nbytes
        return a
    cpp = dict(status=take(np.int32, n), stage=take(np.int8, n), uv_left=take(np.float32, 2 * n).reshape(n, 2), uv_right=take(np.float32, 2 * n).reshape(n, 2),
               xyz=take(np.float64, 3 * n).reshape(n, 3), dl=take(np.uint8, 32 * n).reshape(n, 32), dr=take(np.uint8, 32 * n).reshape(n, 32))
    pose_cpp = _capi.PositResult.from_buffer_copy(raw[off:off + C.sizeof(_capi.PositResult)])
    # the same frame through the Python harness
    d = lambda a: torch.tensor(np.ascontiguousarray(a), device="cuda")  # noqa: E731
    m = svi.HammingMatcher()
    brief = temporal.BriefExtractor(pat, matcher=m)
    brief.set_image("left", d(left))
    brief.set_image("right", d(right))
    fm = temporal.FundamentalMatcher(temporal.StereoCamera(ts.P_LEFT, ts.P_RIGHT, ts.W, ts.H), matcher=m)
    plan = fm.plan(sc.T_est_w2l, sc.dp_T, sc.motion_scaling, d(sc.xyz_world), d(sc.kp_size), d(sc.last_disparity), d(sc.uv_reference), d(sc.dp_index))
    det = sc.make_detector(torch, "cuda")
    got = fm.track_manual(plan, det, brief, d(last_l), d(last_r), d(ref_l))
    st = got.status.cpu().numpy()
    assert np.array_equal(cpp["status"], st) and np.array_equal(cpp["stage"], got.stage.cpu().numpy())
    ok = st == 0
    for k, t in (("uv_left", got.uv_left), ("uv_right", got.uv_right), ("xyz", got.xyz_left), ("dl", got.desc_left), ("dr", got.desc_right)):
        assert np.array_equal(cpp[k][ok], t.cpu().numpy()[ok]), k
    # ... and the per-landmark replay of the reference's cascade with the CPU extractor
    cam = oracle.track_camera(ts.P_LEFT, ts.P_RIGHT, ts.K_INV, ts.W, ts.H)
    rec, _ = oracle.track_plan(cam, sc.T_est_w2l, sc.dp_T, sc.motion_scaling, sc.xyz_world, sc.kp_size, sc.last_disparity, sc.uv_reference, sc.dp_index)
    om = oracle.OracleFundamentalMatcher(cam, sc.stereo_dict())
    want = om.manual(rec, sc.kp_size, sc.detect_one, cpu_extract, last_l, last_r, ref_l)
    check_stage(got, want, n)
    assert np.array_equal(cpp["stage"], np.array([w["stage"] for w in want], np.int8))
    # the pose of the C++ program's getPoseStereoPosit = the harness' one-call variant = oracle posit on the stage-1/2 finds
    solver = temporal.SolverStereoPosit(ts.P_LEFT, ts.P_RIGHT, matcher=m)
    res12, pose = fm.pose_stereo_posit(plan, det, brief, d(last_l), d(last_r), solver, sc.T_est_w2l, np.zeros(3), sc.T_est_w2l)
    assert pose.status == pose_cpp.status and pose.n == pose_cpp.n and pose.iterations == pose_cpp.iterations
    assert np.array_equal(np.array(pose.T_world_to_left[:]), np.array(pose_cpp.T_world_to_left[:]))
    act = (res12.status == 0).to(torch.uint8)
    want_pose = oracle.stereo_posit(oracle.posit_params(ts.P_LEFT, ts.P_RIGHT), sc.T_est_w2l, np.zeros(3), sc.T_est_w2l, sc.xyz_world,
                                    res12.uv_left.cpu().numpy(), res12.uv_right.cpu().numpy(), act.cpu().numpy())
    assert pose.status == want_pose["status"] and pose.n == want_pose["n"] == int(act.sum()) and pose.iterations == want_pose["iterations"]
    assert np.abs(np.array(pose.T_world_to_left[:]) - want_pose["T"]).max() < 1e-9
    # stage 1 -> 2 of the one-call variant = the two stages called one after the other
    s1 = om.stage1(rec, sc.kp_size, cpu_extract, last_l, last_r)
    s2 = om.stage2(rec, sc.kp_size, sc.detect_one, cpu_extract, last_l, last_r)
    want12 = [a if a["status"] in (0, 8) else b for a, b in zip(s1, s2)]
    check_stage(res12, want12, n)
