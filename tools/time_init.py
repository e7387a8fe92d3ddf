import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import svi_mapper_amd as svi
from svi_mapper_amd import synth
prob = bench.cached_problem(1)
cam = synth.kitti_camera()
for rep in range(2):
    t0 = time.perf_counter()
    ba = svi.BundleAdjuster(cam["fx"], cam["fy"], cam["cx"], cam["cy"], cam["baseline_m"])
    synth.build_ba_graph(ba, prob)
    t1 = time.perf_counter()
    ba.initialize()
    t2 = time.perf_counter()
    ba.optimize(1)
    t3 = time.perf_counter()
    print("build graph (python adds) %.2f s, initialize %.3f s, first optimize(1) %.4f s" % (t1 - t0, t2 - t1, t3 - t2))
    ba.close()
