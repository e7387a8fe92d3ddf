"""BASELINE config 5 end to end on a synthetic vi_sensor stream (svi_mapper_amd/vi_stream.py): rendered stereo frames of a
textured ground plane, per frame getPoseStereoPosit -> trackEpipolar through the C++ cascades with the built-in BRIEF
extractor, landmark refinement every 10 frames, key frames, and Cg2oOptimizer::optimize over the growing graph every > 20
key frames (gravity edges with real accelerometer readings and the IMU offset).  A functional test - the parts are checked
against the oracle one by one elsewhere: the estimated trajectory has to follow the rendered one."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_online_tracking_follows_the_truth(svi):
    import torch
    assert torch.cuda.is_available(), "the -m gpu tests need a visible GPU"
    from svi_mapper_amd import vi_stream
    n_frames = 330
    s = vi_stream.ViStream(n_frames, torch.device("cuda", 0), step=0.08)
    trk = vi_stream.OnlineTracker(s)
    trk.start(s.render(0))
    assert trk.n_used > 300
    visible, worst_t, worst_r = [], 0.0, 0.0
    for t in range(1, n_frames):
        visible.append(trk.step(t, s.render(t)))
        et, er = trk.pose_error(t)
        worst_t, worst_r = max(worst_t, et), max(worst_r, er)
    st = trk.stats
    assert min(visible) > 100, (min(visible), st)
    assert st["posit_fail"] == 0, st
    assert st["stage1"] > 0 and st["stage3"] + st["stage2"] > 0, st
    assert len(trk.key_frames) > 25 and st["ba_calls"] >= 1 and st["ba_iterations"] >= 2, st
    # 26 m of walking on integer-pixel stereo with a 11 cm baseline: the drift stays bounded, through the bundle adjustment too
    assert worst_t < 0.4 and worst_r < 2.5, (worst_t, worst_r, st)
    # the graph carries one gravity edge per key frame with the IMU offset and non-trivial measurements
    bst = trk.ba.stats()
    assert bst.n_edges_accel == bst.n_poses and bst.n_edges_se3 == bst.n_poses - 1 and bst.n_edges_proj > 2000
