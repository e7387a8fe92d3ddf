#!/usr/bin/env python3
"""bench.py — headline benchmark of the hot path on MI355X (contract: see the task brief / DESIGN.md).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Primary metric   LM-BA iterations/sec on the BASELINE config-4 graph (500 keyframes x 100 k landmarks x
                 800 k edges, KITTI-00-shaped, synthetic), state resident in HBM; a "step" is one LM
                 iteration (linearise + damped trial(s)) of the reference's schedule.
                 N > 1: the literal config 4 landmark-sharded N ways ("scaling": "strong", value =
                 iterations/sec of that one graph; the reduced camera system is summed with one RCCL
                 all-reduce per trial).  The weak-scaling figure (every rank a config-4-sized shard of a
                 graph with N x 100 k landmarks over the same 500 keyframes, value = N x iterations/sec)
                 is measured in the same run and reported beside it in "other_scaling".
Secondary        stereo desc-pairs/sec on config 2 (2 x 2048 BRIEF-256, epipolar-gated), in "matcher".
roofline         the Jacobian sweep kernel, algorithmic bytes 328 E + 96 P + 24 L over its
                 mean launch duration measured with HIP events on the library's stream.
cpu_baseline     the CPU oracle (oracle/, a restatement of the g2o/CHOLMOD path, 1 thread; all cores as a
                 courtesy figure) on a bounded sample of the same workload, rank 0 at N = 1 only; the
                 matcher's brute-force CPU figure sits in "matcher"."cpu_baseline".
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_MFMA_PEAK_TF = 78.6   # SURVEY.md §8d (FP64 vector/matrix)


def cached_problem(scale):
    """config 4 (x landmark scale), cached under /tmp because the numpy generator takes ~40 s."""
    from svi_mapper_amd import synth
    path = "/tmp/svi_c4_scale%d_v3.npz" % scale
    keys = ("R_true", "t_true", "R_init", "t_init", "lm_true", "lm_init", "obs_kf", "obs_lm", "uvL", "uvR", "xyz")
    if os.path.exists(path):
        try:
            z = np.load(path)
            prob = {k: z[k] for k in keys}
            prob.update(cam=synth.kitti_camera(), n_kf=int(z["n_kf"]), n_lm=int(z["n_lm"]))
            return prob
        except Exception:
            pass
    prob = synth.make_c4(landmark_scale=scale)
    try:
        tmp = path + ".%d.tmp.npz" % os.getpid()
        np.savez(tmp, n_kf=prob["n_kf"], n_lm=prob["n_lm"], **{k: prob[k] for k in keys})
        os.replace(tmp, path)
    except Exception:
        pass
    return prob


LM_BLOCK = 10   # Cg2oOptimizer::_optimizeUnLimited runs optimize(10) blocks (Cg2oOptimizer.cpp:975)


def run_exact(ba, n):
    """exactly n LM iterations in the reference's blocks of 10 (lambda re-initialised per block, Appendix B); a block
    that terminates early is followed by a fresh block, like the reference's while loop would"""
    done = 0
    while done < n:
        r = ba.optimize(min(LM_BLOCK, n - done))
        if r <= 0:
            raise RuntimeError("optimize performed no iteration")
        done += r
    return done


def bench_optimize_call(svi, prob, device, label):
    """What one Cg2oOptimizer::optimize costs behind the boundary once the graph is in the handle (Cg2oOptimizer.cpp:
    503-510): initializeOptimization (structure analysis + upload), _optimizeUnLimited, write-back incl. the read-back of
    the estimates.  Wall-clock on the host."""
    import torch
    from svi_mapper_amd import synth
    cam = prob["cam"]
    ba = svi.BundleAdjuster(cam["fx"], cam["fy"], cam["cx"], cam["cy"], cam["baseline_m"], device=device)
    synth.build_ba_graph(ba, prob)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ba.initialize()
    t1 = time.perf_counter()
    nominal, executed = ba.optimize_until()
    t2 = time.perf_counter()
    ba.apply_optimization()
    t3 = time.perf_counter()
    # a second call on the unchanged graph (only the estimates go back to the device) ...
    ba.initialize()
    t4 = time.perf_counter()
    # ... and one after an edit (a landmark without edges joins): the whole structure analysis again, on warm device buffers -
    # what every later Cg2oOptimizer::optimize of a growing map pays
    ba.add_landmark(987654321, [0.0, 0.0, 5.0])
    t5 = time.perf_counter()
    ba.initialize()
    t6 = time.perf_counter()
    ba.close()
    return {"workload": label, "initialize_ms": 1e3 * (t1 - t0), "optimize_until_ms": 1e3 * (t2 - t1), "write_back_ms": 1e3 * (t3 - t2),
            "total_ms": 1e3 * (t3 - t0), "iterations_nominal": int(nominal), "iterations_executed": int(executed),
            "reinitialize_unchanged_graph_ms": 1e3 * (t4 - t3), "reinitialize_after_edit_ms": 1e3 * (t6 - t5),
            "note": "initialize_ms is the first call on a fresh handle (device buffers allocated); the edge values were sent to the "
                    "device-side log by the add_* calls of the graph construction, which is not in these numbers"}


def bench_config3(svi, device, steps=20, warmup=5):
    """SURVEY 8(d) also asks for config 3 (100 keyframes / 20 k landmarks / 150 k edges) on one GPU: same schedule,
    same timing method as the headline number"""
    import torch
    from svi_mapper_amd import synth
    prob = synth.make_c3()
    cam = prob["cam"]
    ba = svi.BundleAdjuster(cam["fx"], cam["fy"], cam["cx"], cam["cy"], cam["baseline_m"], device=device)
    stored = synth.build_ba_graph(ba, prob)
    ba.initialize()
    run_exact(ba, warmup)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run_exact(ba, steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st = ba.stats()
    out = {"workload": "config 3: %d keyframes x %d landmarks x %d projection edges" % (int(st.n_poses), int(st.n_landmarks), int(st.n_edges_proj)),
           "value": steps / dt, "unit": "iterations/s", "ms_per_step": 1e3 * dt / steps, "steps": steps, "warmup": warmup,
           "chol_levels": int(st.chol_steps), "reduced_tiles": int(st.chol_tiles_nnz), "stored_edges": [int(x) for x in stored]}
    ba.close()
    out["optimize_call"] = bench_optimize_call(svi, prob, device, "config 3")
    return out


def bench_matcher(svi, steps=1000, warmup=50):
    import torch
    from svi_mapper_amd import synth
    dev = torch.device("cuda", torch.cuda.current_device())
    c2 = synth.make_descriptor_pair()
    nq, nt = len(c2["q"]), len(c2["t"])
    st = torch.cuda.Stream()
    m = svi.HammingMatcher(device=dev.index, stream=st.cuda_stream)
    out = {}

    def timed(batch, gated, n, w):
        rep = lambda a: torch.from_numpy(np.tile(a, (batch,) + (1,) * (a.ndim - 1))).to(dev)  # noqa: E731
        q, t = rep(c2["q"]), rep(c2["t"])
        gate = None
        if gated:
            g = c2["gate"]
            gate = dict(q_uv=rep(g["q_uv"]), t_uv=rep(g["t_uv"]), q_umin=rep(g["q_umin"]), q_umax=rep(g["q_umax"]), v_tol=0.0)
        idx = torch.empty(batch * nq, dtype=torch.int32, device=dev)
        dist = torch.empty_like(idx)
        torch.cuda.synchronize()
        for _ in range(w):
            m.match_dev(q, t, nq, nt, batch, idx, dist, gate, c2["cutoff"] if gated else 257)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        st.synchronize()
        e0.record(st)
        for _ in range(n):
            m.match_dev(q, t, nq, nt, batch, idx, dist, gate, c2["cutoff"] if gated else 257)
        e1.record(st)
        e1.synchronize()
        ms = e0.elapsed_time(e1) / n
        return ms, float(batch) * nq * nt / (ms * 1e-3)

    # desc-pairs/s = NQ * NT / kernel time of one call with the gate evaluated per pair as a predicate
    m.set_gate_path(1)
    ms, rate = timed(1, True, steps, warmup)
    out["gated_predicate_single"] = {"ms_per_call": ms, "pairs_per_s": rate}
    ms, rate = timed(64, True, max(steps // 10, 20), 5)
    out["gated_predicate_batch64"] = {"ms_per_call": ms, "pairs_per_s": rate}
    m.set_gate_path(0)
    # the default path of a gated call: the pool bucketed by image row, a query visits only its candidates - far fewer
    # pairs are looked at, so this is reported per call and per candidate, NOT as NQ * NT / t
    g = c2["gate"]
    tv, tu = g["t_uv"][:, 1], g["t_uv"][:, 0]
    cand = 0
    for i in range(nq):
        cand += int(((np.abs(tv - g["q_uv"][i, 1]) <= g.get("v_tol", 0.0)) & (tu >= g["q_umin"][i]) & (tu < g["q_umax"][i])).sum())
    ms, _ = timed(1, True, steps, warmup)
    out["gated_rowbucket_single"] = {"ms_per_call": ms, "gated_candidates": cand, "gated_candidates_per_s": cand / (ms * 1e-3)}
    ms, _ = timed(64, True, max(steps // 10, 20), 5)
    out["gated_rowbucket_batch64"] = {"ms_per_call": ms, "gated_candidates_per_s": 64 * cand / (ms * 1e-3)}
    ms, rate = timed(1, False, steps, warmup)
    out["ungated_single"] = {"ms_per_call": ms, "pairs_per_s": rate}
    ms, rate = timed(64, False, max(steps // 10, 20), 5)
    out["ungated_batch64"] = {"ms_per_call": ms, "pairs_per_s": rate}
    # 180 224 algorithmic bytes per call, 17 integer lane-ops per pair; the binding resource is integer VALU
    # issue: n_cu x 64 lane-ops per cycle at the clock measured while the chip does exactly this kind of work
    alg_bytes = 32 * (nq + nt) + 8 * (nq + nt) + 8 * nq
    mhz = m.shader_clock_mhz()
    n_cu = torch.cuda.get_device_properties(dev).multi_processor_count
    peak = n_cu * 64 * mhz * 1e6
    b = out["ungated_batch64"]
    out["roofline"] = {"bound": "valu-int", "unit": "lane-ops/s", "achieved": 17.0 * b["pairs_per_s"], "peak": peak,
                       "frac": 17.0 * b["pairs_per_s"] / peak if peak > 0 else None, "kernel": "k_match_hamming256 (ungated, batch of 64 pair sets)",
                       "lane_ops_per_pair": 17, "shader_clock_mhz_measured": mhz, "compute_units": n_cu,
                       "frac_single_call_predicate": 17.0 * out["gated_predicate_single"]["pairs_per_s"] / peak if peak > 0 else None,
                       "algorithmic_bytes_per_pair_set": alg_bytes,
                       "hbm_GBps_batch64": 64 * alg_bytes / (b["ms_per_call"] * 1e-3) / 1e9}
    m.close()
    return {"metric": "stereo desc-pairs/sec", "workload": "config 2: 2x2048 BRIEF-256, epipolar gate evaluated per pair, KITTI-00 camera, one call",
            "value": out["gated_predicate_single"]["pairs_per_s"], "unit": "pairs/s", "dtype": "u8", **out}


def bench_frontend(svi, reps=200):
    """Secondary: the per-frame passes around the matcher on frame-sized synthetic inputs.
    Launch/latency bound by nature (a few thousand landmarks); reported as ms per call with the stream drained."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import landmark_case
    import posit_case
    import track_scene as ts
    from svi_mapper_amd import temporal
    dev = torch.device("cuda", torch.cuda.current_device())
    d = lambda a: torch.tensor(np.ascontiguousarray(a), device=dev)  # noqa: E731
    out = {}

    def timed(fn, n=reps):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t0) / n

    sc = ts.Scene(n=2000, seed=7)
    fm = temporal.FundamentalMatcher(temporal.StereoCamera(ts.P_LEFT, ts.P_RIGHT, ts.W, ts.H), device=dev.index)
    args = (sc.T_est_w2l, sc.dp_T, sc.motion_scaling, d(sc.xyz_world), d(sc.kp_size), d(sc.last_disparity), d(sc.uv_reference), d(sc.dp_index))
    plan = fm.plan(*args)
    out["track_plan_2000_landmarks_ms"] = timed(lambda: fm.plan(*args))
    out["epipolar_samples_ms"] = timed(lambda: fm.epipolar_samples(plan, 0))
    seg = plan.seg
    rng = np.random.default_rng(0)
    pool = d(rng.integers(0, 256, (plan.total, 32), dtype=np.uint8))
    ll, rf = d(sc.last_left), d(sc.ref_desc)
    out["ragged_match_ms"] = timed(lambda: fm.get_match(ll, rf, seg, pool, 50, 100))
    out["ragged_match_candidates"] = int(plan.total)
    # BRIEF extraction of the same frame's epipolar candidates (stand-in test-pair table, tests/brief_case.py)
    import brief_case
    br = temporal.BriefExtractor(brief_case.pattern(), matcher=fm.matcher, device=dev.index)
    img = d(brief_case.image(ts.H, ts.W, 1))
    out["brief_integral_ms"] = timed(lambda: br.set_image("left", img))
    g_seg, samples, roi = fm.epipolar_samples(plan, 0)
    kept = br("left", roi, g_seg, samples)
    out["brief_extract_ms"] = timed(lambda: br("left", roi, g_seg, samples), 100)
    out["brief_keypoints_in"] = int(samples.shape[0])
    out["brief_keypoints_kept"] = int(kept[1].shape[0])
    # a whole frame through the C++ cascades (CFundamentalMatcher::trackManual: stage 1 -> 2 -> 3 for every landmark, BRIEF on the
    # device; the stage-2 detector is a stand-in on the device behind the Python callback shim, its cost is inside)
    pts = {"left": d(sc.corners[0]), "right": d(sc.corners[1])}

    def det(side, rect):
        p = pts[side]
        ul, lr = rect[:, :2].floor(), rect[:, 2:].floor()
        inside = (p[None, :, 0] >= ul[:, None, 0]) & (p[None, :, 0] < lr[:, None, 0]) & (p[None, :, 1] >= ul[:, None, 1]) & (p[None, :, 1] < lr[:, None, 1])
        idx = torch.nonzero(inside)
        sg = torch.zeros(rect.shape[0] + 1, dtype=torch.int32, device=dev)
        sg[1:] = torch.cumsum(inside.sum(1), 0)
        return sg, (p[idx[:, 1]] - ul[idx[:, 0]]).contiguous()
    img_r = d(brief_case.image(ts.H, ts.W, 2))
    br.set_image("right", img_r)
    lr_ = d(sc.last_right)
    plan = fm.plan(*args)
    res = fm.track_manual(plan, det, br, ll, lr_, rf)
    out["track_manual_frame_ms"] = timed(lambda: fm.track_manual(plan, det, br, ll, lr_, rf), 30)
    out["track_manual_frame_landmarks"] = int(plan.n)
    out["track_manual_frame_rows_tracked"] = int((res.status != 8).sum())
    out["track_stage1_frame_ms"] = timed(lambda: fm.track_stage1(plan, br, ll, lr_), 30)
    out["track_epipolar_frame_ms"] = timed(lambda: fm.track_epipolar(plan, br, ll, rf), 30)
    # loop closure: 2000 query descriptors against 200 key frames x ~1500 descriptors
    nq, n_clouds = 2000, 200
    sizes = rng.integers(1000, 2000, n_clouds)
    cseg = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    pools = d(rng.integers(0, 256, (int(cseg[-1]), 32), dtype=np.uint8))
    q = d(rng.integers(0, 256, (nq, 32), dtype=np.uint8))
    idx = torch.empty(n_clouds * nq, dtype=torch.int32, device=dev)
    dist = torch.empty_like(idx)
    cs = d(cseg)

    def clouds():
        fm.matcher.match_clouds_dev(q, nq, pools, cs, n_clouds, int(sizes.max()), idx, dist)
        fm.matcher.synchronize()
    ms = timed(clouds, 50)
    out["loop_closure_200_keyframes_ms"] = ms
    out["loop_closure_pairs_per_s"] = float(nq) * float(cseg[-1]) / (ms * 1e-3)
    c = posit_case.make(500, 2)
    solver = temporal.SolverStereoPosit(ts.P_LEFT, ts.P_RIGHT, matcher=fm.matcher, device=dev.index)
    px, pl, pr = d(c["xyz"]), d(c["uvl"]), d(c["uvr"])
    r = solver.solve(c["T_last"], c["t_imu"], c["T_est"], px, pl, pr)
    out["stereo_posit_500_points_ms"] = timed(lambda: solver.solve(c["T_last"], c["t_imu"], c["T_est"], px, pl, pr))
    out["stereo_posit_iterations"] = int(r.iterations)
    lc = landmark_case.make(5000, 5)
    lo = temporal.LandmarkOptimizer(matcher=fm.matcher, device=dev.index)
    la = [d(lc[k]) for k in ("PL", "PR", "seg", "frame", "uvl", "uvr", "xyz0")]
    out["landmark_optimize_5000_landmarks_ms"] = timed(lambda: lo.optimize(*la), 50)
    out["landmark_measurements"] = int(lc["seg"][-1])
    return out


def bench_config5(svi, device, n_frames=420):
    """BASELINE config 5: vi_sensor stereo + IMU stream (svi_mapper_amd/vi_stream.py: rendered frames of a textured ground
    plane, resident in HBM before the clock starts), online tracking + BA every > 20 key frames over the growing graph on one
    GPU, end-to-end frames/s INCLUDING svi_ba_initialize and the write-back.  The loop around the hot path (landmark
    bookkeeping, the stand-in detector of stage 2) is Python / torch and is inside the clock."""
    import torch
    from svi_mapper_amd import vi_stream
    dev = torch.device("cuda", device)
    s = vi_stream.ViStream(n_frames, dev, step=0.08)
    frames = [s.render(t) for t in range(n_frames)]
    trk = vi_stream.OnlineTracker(s, device_index=device)
    trk.start(frames[0])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    vis = []
    for t in range(1, n_frames):
        vis.append(trk.step(t, frames[t]))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st = trk.stats
    et, er = trk.pose_error(n_frames - 1)
    bst = trk.ba.stats()
    return {"metric": "end-to-end frames/sec", "value": (n_frames - 1) / dt, "unit": "frames/s", "frames": n_frames - 1,
            "workload": "config 5: synthetic vi_sensor stereo+IMU stream (752x480, f 450.5, baseline 0.110 m), %d frames, getPoseStereoPosit + "
                        "trackEpipolar per frame (built-in BRIEF), landmark refinement every 10 frames, BA every > 20 key frames" % (n_frames - 1),
            "ms_per_frame": 1e3 * dt / (n_frames - 1), "landmarks_visible_mean": float(np.mean(vis)), "landmarks_created": st["landmarks_created"],
            "tracks_stage1": st["stage1"], "tracks_stage2": st["stage2"], "tracks_stage3": st["stage3"], "posit_failures": st["posit_fail"],
            "key_frames": len(trk.key_frames), "ba_calls": st["ba_calls"], "ba_iterations": st["ba_iterations"], "ba_ms_total": st["ba_ms"],
            "ba_initialize_ms_total": st["ba_initialize_ms"],
            "detector_stand_in": {"ms_total": st["detector_ms"], "calls": st["detector_calls"], "share": st["detector_ms"] / (1e3 * dt),
                                  "frames_per_s_without_it": (n_frames - 1) / max(dt - 1e-3 * st["detector_ms"], 1e-9),
                                  "note": "host wall clock inside the Python stand-in for GFTT (stage 2's detector callback; the caller's in the "
                                          "reference too), incl. the host waits its torch.nonzero forces"},
            "library": {"ms_total": st["library_ms"] + st["ba_ms"], "ms_per_frame": (st["library_ms"] + st["ba_ms"]) / (n_frames - 1),
                        "frames_per_s": (n_frames - 1) / (1e-3 * (st["library_ms"] + st["ba_ms"])),
                        "note": "host wall clock inside the library's entry points alone (plan, the cascades incl. BRIEF and the frame pose, landmark "
                                "refinement, new landmarks, graph construction + initialize + _optimizeUnLimited + write-back), through the ctypes "
                                "shim, the detector callback's time subtracted: what the path behind the C ABI costs per frame; the rest of "
                                "ms_per_frame is the Python / torch loop around it"},
            "ba_graph": {"poses": int(bst.n_poses), "landmarks": int(bst.n_landmarks),
                                                                            "projection_edges": int(bst.n_edges_proj), "gravity_edges": int(bst.n_edges_accel)},
            "final_pose_error_m": et, "final_pose_error_deg": er, "dtype": "u8 (matching) / f64 (pose, BA)"}


def _host_cpu():
    model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except Exception:
        pass
    try:
        avail = len(os.sched_getaffinity(0))
    except Exception:
        avail = os.cpu_count() or 1
    return model, avail


def _oracle_native():
    from oracle import oracle as orc
    try:
        path = orc.build(native=True)
        return orc.load(path), path
    except Exception:
        return orc.load(), os.path.join(ROOT, "oracle", "liboracle.so")


def cpu_baseline(prob, iters, all_cores=True):
    """The CPU oracle (restatement of the g2o/CHOLMOD path), -O3 -march=native, timed on this host: 1 thread (the
    reference is single-threaded: cv::setNumThreads(1), CTrackerGT.cpp:48, no OpenMP anywhere) and, as a courtesy figure
    (BASELINE.md section 3), all cores - the restatement has no parallel solver, so that figure is the THROUGHPUT of one
    independent solve of the same graph per core, child processes that never touch the GPU."""
    import subprocess
    from oracle import oracle as orc
    from svi_mapper_amd import synth
    lib, libpath = _oracle_native()
    cam = prob["cam"]
    o = orc.OracleBA(cam["fx"], cam["fy"], cam["cx"], cam["cy"], cam["baseline_m"], lib=lib)
    synth.build_ba_graph(o, prob)
    o.initialize()
    t0 = time.time()
    done = 0
    while done < iters:
        done += o.optimize(iters - done)
    dt = time.time() - t0
    model, avail = _host_cpu()
    out = {"value": done / dt, "unit": "LM iterations/s", "cores": 1, "kind": "port",
           "sample": "first %d LM iterations of the same config-4 graph (%d edges), CPU restatement of the g2o/CHOLMOD path "
                     "(full-system sparse LL', no Schur), gcc -O3 -march=native, %.1f s" % (done, o.num_edges, dt),
           "host_cpu": model, "host_cores_available": avail}
    del o
    if all_cores and avail > 1:
        n = min(avail, 16)
        path = "/tmp/svi_c4_scale1_v3.npz"
        try:
            if not os.path.exists(path):
                cached_problem(1)
            it_each = max(6, iters // 3)
            ps = [subprocess.Popen([sys.executable, os.path.join(ROOT, "oracle", "cpu_worker.py"), path, str(it_each), libpath],
                                   stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True) for _ in range(n)]
            for p_ in ps:
                if p_.stdout.readline().strip() != "ready":
                    raise RuntimeError("worker did not come up")
            t0 = time.time()
            for p_ in ps:
                p_.stdin.write("go\n")
                p_.stdin.flush()
            res = [json.loads(p_.stdout.readline()) for p_ in ps]
            wall = time.time() - t0
            for p_ in ps:
                p_.wait(timeout=60)
            total = sum(r["iterations"] for r in res)
            out["all_cores"] = {"value": total / wall, "unit": "LM iterations/s (throughput over independent solves)", "cores": n, "kind": "port",
                                "sample": "%d processes, each the first %d LM iterations of the same config-4 graph; %.1f s wall; "
                                          "per-process rate %.2f it/s" % (n, it_each, wall, total / wall / n)}
        except Exception as e:  # noqa: BLE001
            out["all_cores"] = {"error": "%s: %s" % (type(e).__name__, e)}
            for p_ in locals().get("ps", []):
                try:
                    p_.kill()
                except Exception:
                    pass
    return out


def cpu_baseline_matcher(seconds=3.0):
    """cv::BFMatcher(NORM_HAMMING) as the reference uses it (CTriangulator.cpp:12; single-threaded by CTrackerGT.cpp:48),
    restated in oracle/oracle_match.c: brute force over all NQ x NT pairs of config 2 with the gate as a per-pair predicate,
    -O3 -march=native (hardware popcount).  1 thread, and all cores as a courtesy figure (threads over independent calls)."""
    import threading
    from oracle import oracle as orc
    from svi_mapper_amd import synth
    lib, _ = _oracle_native()
    c2 = synth.make_descriptor_pair()
    nq, nt = len(c2["q"]), len(c2["t"])

    def calls(budget, gate):
        n, t0 = 0, time.time()
        while True:
            orc.match_hamming256(c2["q"], c2["t"], gate, c2["cutoff"] if gate is not None else 257, lib=lib)
            n += 1
            dt = time.time() - t0
            if dt >= budget:
                return n, dt

    model, avail = _host_cpu()
    n_g, t_g = calls(seconds / 2, c2["gate"])
    n_u, t_u = calls(seconds / 2, None)
    out = {"value": n_g * float(nq) * nt / t_g, "unit": "pairs/s", "cores": 1, "kind": "port",
           "sample": "%d gated + %d ungated brute-force calls on the config-2 descriptor sets (2 x 2048 BRIEF-256), %.1f s; value = "
                     "NQ x NT / time of a gated call (the gate rejects most pairs before the popcount)" % (n_g, n_u, t_g + t_u),
           "ungated_pairs_per_s": n_u * float(nq) * nt / t_u, "ms_per_call_gated": 1e3 * t_g / n_g, "ms_per_call_ungated": 1e3 * t_u / n_u,
           "host_cpu": model, "host_cores_available": avail}
    n = min(avail, 16)
    if n > 1:
        cnt = [0] * n

        def work(i):   # (ctypes releases the GIL for the duration of the C call)
            cnt[i] = calls(seconds / 2, None)[0]
        th = [threading.Thread(target=work, args=(i,)) for i in range(n)]
        t0 = time.time()
        for t_ in th:
            t_.start()
        for t_ in th:
            t_.join()
        wall = time.time() - t0
        out["all_cores"] = {"value": sum(cnt) * float(nq) * nt / wall, "unit": "pairs/s (ungated, throughput over independent calls)", "cores": n,
                            "kind": "port", "sample": "%d threads x brute-force ungated calls for %.1f s" % (n, wall)}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--scaling", choices=("auto", "weak", "strong"), default="auto",
                    help="N > 1: strong = the literal config 4 (500 key frames / 100 k landmarks / 800 k edges) split N ways - the "
                         "headline (auto); weak = N x config-4 landmarks over the same key frames, value = N x iterations/s. The "
                         "other one is measured too and reported beside the headline (other_scaling)")
    ap.add_argument("--no-other-scaling", action="store_true", help="N > 1: measure the headline scaling mode only")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-matcher", action="store_true")
    ap.add_argument("--no-frontend", action="store_true")
    ap.add_argument("--no-replay", action="store_true", help="skip the back-to-back sweep replays (rocprof runs: the kernel stats "
                    "then hold in-loop launches only)")
    ap.add_argument("--cpu-iters", type=int, default=30)
    ap.add_argument("--chol-tile", type=int, default=48)
    ap.add_argument("--backend", default="nccl")
    ap.add_argument("--hook", choices=("native", "torch"), default="native",
                    help="N > 1: the all-reduce of the reduced system through the library's own RCCL communicator (C++, svi_rccl_*) or "
                         "through torch.distributed (the gloo rehearsal always uses torch)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    import svi_mapper_amd as svi
    from svi_mapper_amd import dist as sdist
    from svi_mapper_amd import synth

    rank, world, local = sdist.init_from_env(args.backend)
    if args.scaling == "auto":
        # BASELINE config 4 IS one 500 / 100 k / 800 k graph sharded over the GPUs of a node: the fixed-size job is the headline
        # (the replicated Cholesky bounds it, DESIGN.md section 7 - the weak figure beside it shows what the sharded part scales to)
        args.scaling = "strong" if int(os.environ.get("WORLD_SIZE", "1")) > 1 else "weak"
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d" % (args.gpus, world, args.gpus))
    local = local % max(torch.cuda.device_count(), 1)   # more ranks than GPUs only happens in the gloo rehearsal on one card
    torch.cuda.set_device(local)
    if world > 1:
        # rank 0 generates (and caches) the synthetic graphs, the others pick the cache up afterwards
        if rank == 0:
            if not (args.no_other_scaling and args.scaling == "strong"):
                cached_problem(world)
            cached_problem(1)
        dist.barrier()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    native = None
    hook_used = "none"
    if world > 1:
        hook_used = "torch.distributed"
        if args.hook == "native" and args.backend == "nccl":
            try:
                native = sdist.NativeRccl(rank, world, local)
                hook_used = "native RCCL (svi_rccl_allreduce)"
            except Exception as e:  # noqa: BLE001
                print("rank %d: native RCCL hook unavailable (%s): using torch.distributed" % (rank, e), file=sys.stderr)
                native = None
        # every rank must take the same path
        flag = torch.tensor([1 if native is not None else 0], device="cpu" if args.backend == "gloo" else "cuda")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            native, hook_used = None, "torch.distributed"

    def attach(ba):
        if world > 1:
            if native is not None:
                ba.set_allreduce_native(native)
            else:
                ba.set_allreduce(sdist.make_allreduce_hook())

    def measure(prob, steps, warmup):
        """one handle per rank over `prob` (landmark-sharded `world` ways), W untimed + K timed LM iterations"""
        cam = prob["cam"]
        ba = svi.BundleAdjuster(cam["fx"], cam["fy"], cam["cx"], cam["cy"], cam["baseline_m"], device=local, rank=rank, n_ranks=world,
                                chol_tile=args.chol_tile)
        stored = synth.build_ba_graph(ba, prob)
        attach(ba)
        ba.initialize()
        run_exact(ba, warmup)
        s0 = ba.stats()
        barrier()
        t0 = time.perf_counter()
        run_exact(ba, steps)
        barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device="cpu" if args.backend == "gloo" else "cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        s1 = ba.stats()
        trials = (s1.lm_trials - s0.lm_trials) / max(1, s1.lm_iterations - s0.lm_iterations)
        chi = ba.chi2()
        # the metric is quoted with the state resident on the device; what one read-back of the optimised
        # poses and landmarks to the host costs (with several ranks: incl. the gather of the shards) is reported beside it
        t1 = time.perf_counter()
        ba.sync_host()
        readback_ms = 1e3 * (time.perf_counter() - t1)
        ba.close()
        return dict(dt=dt, stats=s1, trials=trials, chi=chi, readback_ms=readback_ms, stored=stored)

    # ---- timed run -------------------------------------------------------------------------------
    scale_of = {"weak": world, "strong": 1}
    prob = cached_problem(scale_of[args.scaling])
    cam = prob["cam"]
    head = measure(prob, args.steps, args.warmup)
    dt, stored = head["dt"], head["stored"]
    units = world if args.scaling == "weak" else 1     # weak: shard-iterations per second over the whole job
    other = None
    if world > 1 and not args.no_other_scaling:
        mode = "strong" if args.scaling == "weak" else "weak"
        o = measure(cached_problem(scale_of[mode]), args.steps, args.warmup)
        ou = world if mode == "weak" else 1
        other = {"scaling": mode, "value": ou * args.steps / o["dt"], "unit": "iterations/s", "ms_per_step": 1e3 * o["dt"] / args.steps,
                 "landmarks_total": int(o["stats"].n_landmarks), "edges_total": int(o["stats"].n_edges_proj),
                 "landmarks_per_gpu": int(o["stats"].n_landmarks_local), "allreduce_doubles_per_trial": int(o["stats"].reduce_doubles),
                 "note": "strong = the literal config 4 (500 key frames / 100 k landmarks / 800 k edges) split %d ways, value = iterations/s of "
                         "that one graph; weak = %d x config-4 landmarks over the same key frames, value = %d x iterations/s" % (world, world, world)}

    def make(**kw):
        ba = svi.BundleAdjuster(cam["fx"], cam["fy"], cam["cx"], cam["cy"], cam["baseline_m"], device=local, rank=rank,
                                n_ranks=world, chol_tile=args.chol_tile, **kw)
        synth.build_ba_graph(ba, prob)
        attach(ba)
        ba.initialize()
        return ba

    # ---- the Jacobian sweep where it runs: HIP event pairs around K2 + K3 of every linearisation inside the ordinary LM loop
    # (no other instrumentation, the host waits on the published trial scalars only; between two sweeps the Schur, Cholesky
    # and back-substitution kernels of the trials run and push the sweep's operands around the cache hierarchy) -----------
    bae = make(sweep_events=True)
    run_exact(bae, args.warmup)
    bae.reset_phase_times()
    run_exact(bae, args.steps)
    loop_ms, loop_n = bae.sweep_time()
    sweep_loop_ms = loop_ms / max(loop_n, 1)
    bae.close()

    # ---- profiled run: per-phase device time from HIP events on the library's stream (every phase synchronised) ------
    bap = make(profile=True)
    run_exact(bap, args.warmup)
    bap.reset_phase_times()
    run_exact(bap, args.steps)
    phases = bap.phase_times()
    st = bap.stats()
    sweep_replay_ms = sweep_cold_ms = None
    sweep_parts = (None, None)
    if not args.no_replay:
        bap.time_sweep(1000)                    # untimed: clocks ramped up after the mostly idle profiled run
        sweep_replay_ms = bap.time_sweep(200)   # 200 back-to-back sweeps: operands warm in the 256 MiB Infinity Cache
        sweep_parts = (bap.time_sweep(200, 1), bap.time_sweep(200, 2))
        sweep_cold_ms = bap.time_sweep_cold(20, 640 << 20)   # every sweep behind a 640 MiB fill: nothing of it cached
    bap.close()

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return
    E, P, L = int(st.n_edges_proj_local), int(st.n_poses), int(st.n_landmarks_local)
    sweep_bytes = 328 * E + 96 * P + 24 * L
    gbs = lambda ms: sweep_bytes / (ms * 1e-3) / 1e9 if ms else None  # noqa: E731
    achieved = gbs(sweep_loop_ms) or 0.0
    ms_chol, n_chol = phases["cholesky"]
    chol_ms = ms_chol / max(n_chol, 1)
    pmc = None
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json"))).get("sweep_hbm_bytes_per_launch")
    except Exception:
        pass
    hs = head["stats"]
    line = {
        "metric": "LM-BA iterations/sec", "value": units * args.steps / dt, "unit": "iterations/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True,
        "scaling": args.scaling, "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "config 4: KITTI-00-shaped BA, 500 keyframes x %d landmarks x %d projection edges "
                               "(xyz %d / uv-depth %d / uv-disparity %d) + 499 odometry + 500 gravity edges; "
                               "%d-way landmark-sharded, reference LM schedule (blocks of %d iterations)"
                               % (int(hs.n_landmarks), int(hs.n_edges_proj), stored[0], stored[1], stored[2], world, LM_BLOCK),
                   "keyframes": P, "landmarks_per_gpu": int(hs.n_landmarks_local), "edges_per_gpu": int(hs.n_edges_proj_local),
                   "parallelism": "landmark-shard x%d" % world,
                   "chol_tile": int(hs.chol_tile), "reduced_n": int(hs.chol_n), "reduced_tiles": int(hs.chol_tiles_nnz),
                   "trials_per_iteration": head["trials"], "final_chi2_plain": head["chi"][0], "final_chi2_robust": head["chi"][1],
                   "allreduce_doubles_per_trial": int(hs.reduce_doubles) if world > 1 else 0, "allreduce_hook": hook_used,
                   "state": "resident in HBM during the timed iterations (SURVEY 8d-i)", "readback_ms_once": head["readback_ms"]},
        "roofline": {"bound": "hbm", "kernel": "Jacobian sweep = k_linearize_lm (K2) + k_linearize_pose (K3)",
                     "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "measured": "HIP event pairs around K2+K3 of all %d linearisations inside the non-synchronised LM loop" % loop_n,
                     "algorithmic_bytes": sweep_bytes, "avg_ms": sweep_loop_ms,
                     "frac_replay": (gbs(sweep_replay_ms) or 0.0) / HBM_PEAK_GBS if sweep_replay_ms else None, "avg_ms_replay": sweep_replay_ms,
                     "frac_cold": (gbs(sweep_cold_ms) or 0.0) / HBM_PEAK_GBS if sweep_cold_ms else None, "avg_ms_cold": sweep_cold_ms,
                     "frac_profiled_loop": (gbs((phases["linearize_lm"][0] + phases["linearize_pose"][0]) / max(phases["linearize_lm"][1], 1)) or 0.0) / HBM_PEAK_GBS,
                     "avg_ms_k2_alone_replay": sweep_parts[0], "avg_ms_k3_alone_replay": sweep_parts[1],
                     "traffic": pmc,
                     "traffic_note": "bytes at the L2 <-> fabric boundary (FETCH_SIZE + WRITE_SIZE, corrected per MI355X_MICROARCH.md), "
                                     "Infinity-Cache hits included, from the committed rocprofv3 --pmc passes of this workload "
                                     "(profiles/pmc_traffic.json) - a constant of the profile, not measured in this run"},
        "roofline_cholesky": {"bound": "mfma", "kernel": "tile-sparse LL' (potrf+trsm+gemm+solve)", "achieved": st.chol_flops / (chol_ms * 1e-3) / 1e12 if chol_ms > 0 else 0.0,
                              "peak": FP64_MFMA_PEAK_TF, "unit": "TFLOP/s", "flops": st.chol_flops, "avg_ms": chol_ms},
        "phases_ms_per_call": {k: (v[0] / v[1] if v[1] else 0.0) for k, v in phases.items()},
        "phases_calls": {k: v[1] for k, v in phases.items()},
    }
    line["roofline_cholesky"]["frac"] = line["roofline_cholesky"]["achieved"] / FP64_MFMA_PEAK_TF
    if other is not None:
        line["other_scaling"] = other
    if world == 1:
        line["optimize_call"] = bench_optimize_call(svi, prob, local, "config 4")
    if world == 1 and not args.no_matcher:
        line["config3"] = bench_config3(svi, local)
        line["matcher"] = bench_matcher(svi)
    if world == 1 and not args.no_frontend:
        line["frontend"] = bench_frontend(svi)
        line["config5"] = bench_config5(svi, local)
    if world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(prob, args.cpu_iters)
        line["speedup_vs_cpu_baseline"] = line["value"] / line["cpu_baseline"]["value"]
        if "matcher" in line:
            line["matcher"]["cpu_baseline"] = cpu_baseline_matcher()
    print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
