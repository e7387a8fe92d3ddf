import sys, os
sys.path.insert(0, '.')
import bench, svi_mapper_amd as svi
from svi_mapper_amd import synth
prob = bench.cached_problem(1)
cam = prob["cam"]
for tile in (48, 96):
    ba = svi.BundleAdjuster(cam["fx"], cam["fy"], cam["cx"], cam["cy"], cam["baseline_m"], chol_tile=tile)
    synth.build_ba_graph(ba, prob)
    print("tile", tile, flush=True)
    ba.initialize()
    st = ba.stats()
    print("levels", st.chol_steps, "tiles", st.chol_tiles_nnz, "flops %.3g" % st.chol_flops, flush=True)
    ba.close()
