"""K2 / K3 / whole-sweep replay times and the in-loop sweep time at config 4 (what bench.py's roofline block reports), quickly."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench, svi_mapper_amd as svi
from svi_mapper_amd import synth
prob = bench.cached_problem(1)
cam = prob["cam"]
def make(**kw):
    ba = svi.BundleAdjuster(cam["fx"], cam["fy"], cam["cx"], cam["cy"], cam["baseline_m"], **kw)
    synth.build_ba_graph(ba, prob)
    ba.initialize()
    return ba
bae = make(sweep_events=True)
bench.run_exact(bae, 10); bae.reset_phase_times(); bench.run_exact(bae, 40)
ms, n = bae.sweep_time()
st = bae.stats()
B = 328 * st.n_edges_proj_local + 96 * st.n_poses + 24 * st.n_landmarks_local
print("in-loop sweep %.2f us  frac %.3f" % (1e3 * ms / n, B / (ms / n * 1e-3) / 8e12), flush=True)
bae.time_sweep(500)
w, k2, k3 = bae.time_sweep(200), bae.time_sweep(200, 1), bae.time_sweep(200, 2)
print("replay: sweep %.2f us (frac %.3f)  K2 %.2f  K3 %.2f" % (1e3 * w, B / (w * 1e-3) / 8e12, 1e3 * k2, 1e3 * k3), flush=True)
print("cold %.2f us" % (1e3 * bae.time_sweep_cold(20, 640 << 20)), flush=True)
