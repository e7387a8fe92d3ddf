import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench, svi_mapper_amd as svi
from svi_mapper_amd import synth
prob = bench.cached_problem(1); cam = synth.kitti_camera()
ba = svi.BundleAdjuster(cam["fx"], cam["fy"], cam["cx"], cam["cy"], cam["baseline_m"])
synth.build_ba_graph(ba, prob); ba.initialize(); print(ba.optimize(2))
