"""Times the Jacobian sweep (k_jacobian_sweep) at config 4 on the library's stream and prints the roofline fraction;
optionally a short LM run as a sanity check of the results.  Usage: python tools/time_sweep.py [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import svi_mapper_amd as svi
from svi_mapper_amd import synth

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
prob = bench.cached_problem(1)
cam = synth.kitti_camera()
ba = svi.BundleAdjuster(cam["fx"], cam["fy"], cam["cx"], cam["cy"], cam["baseline_m"])
synth.build_ba_graph(ba, prob)
ba.initialize()
st = ba.stats()
B = 328 * st.n_edges_proj_local + 96 * st.n_poses + 24 * st.n_landmarks_local
for _ in range(3):
    ms = ba.time_sweep(reps)
    print("sweep %.2f us  %.2f TB/s algorithmic  frac %.3f   (K2 alone %.2f us, K3 alone %.2f us)" %
          (ms * 1e3, B / ms / 1e9, B / ms / 1e9 / 8.0, 1e3 * ba.time_sweep(reps, 1), 1e3 * ba.time_sweep(reps, 2)))
n = ba.optimize(3)
print("optimize(3) ->", n, "chi2", ba.chi2(), "lambda", ba.lm_lambda)
