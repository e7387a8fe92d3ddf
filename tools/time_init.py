"""svi_ba_initialize on the GPU box: cold (first call on a fresh handle), warm (second call, graph unchanged), and after an
append of more key frames' worth of edges; SVI_DEBUG_PLAN=1 prints the section times of build_structure."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import svi_mapper_amd as svi  # noqa: E402
from svi_mapper_amd import synth  # noqa: E402

prob = bench.cached_problem(1)
cam = synth.kitti_camera()
ba = svi.BundleAdjuster(cam["fx"], cam["fy"], cam["cx"], cam["cy"], cam["baseline_m"])
t0 = time.perf_counter()
synth.build_ba_graph(ba, prob)
t1 = time.perf_counter()
print("build graph (python adds) %.2f s" % (t1 - t0))
for rep in range(5):
    if rep in (2, 3):   # an edit of the graph (a landmark without edges): the whole structure analysis runs again, on warm buffers
        ba.add_landmark(900000 + rep, [0.0, 0.0, 5.0])
    t1 = time.perf_counter()
    ba.initialize()
    t2 = time.perf_counter()
    r = ba.optimize(2)
    t3 = time.perf_counter()
    print("initialize %.2f ms, optimize(2) -> %d in %.2f ms, chi2 %.6g" % (1e3 * (t2 - t1), r, 1e3 * (t3 - t2), ba.chi2()[1]))
