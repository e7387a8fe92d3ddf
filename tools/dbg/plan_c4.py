"""SVI_DEBUG_PLAN=1 python tools/dbg/plan_c4.py : the level / column / tile structure of config 4 as the library plans it (stderr)."""
import os, sys
sys.path.insert(0, '.')
os.environ["SVI_DEBUG_PLAN"] = "1"
import bench, svi_mapper_amd as svi
from svi_mapper_amd import synth
prob = bench.cached_problem(1)
cam = prob["cam"]
ba = svi.BundleAdjuster(cam["fx"], cam["fy"], cam["cx"], cam["cy"], cam["baseline_m"])
synth.build_ba_graph(ba, prob)
ba.initialize()
st = ba.stats()
print("levels", st.chol_steps, "tiles", st.chol_tiles_nnz, "flops %.3g" % st.chol_flops, flush=True)
