"""A/B of the staged / overlapped trial at config 4: iterations/s with the environment as given (SVI_NO_OVERLAP, SVI_SCHUR_STAGES,
SVI_SCHUR_RESERVE_CUS)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench, svi_mapper_amd as svi
from svi_mapper_amd import synth
prob = bench.cached_problem(1)
cam = prob["cam"]
def make(**kw):
    ba = svi.BundleAdjuster(cam["fx"], cam["fy"], cam["cx"], cam["cy"], cam["baseline_m"], **kw)
    synth.build_ba_graph(ba, prob)
    ba.initialize()
    return ba
ba = make()
print("levels", ba.stats().chol_steps, flush=True)
bench.run_exact(ba, 10)
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter(); bench.run_exact(ba, 40); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("it/s %.1f  ms/it %.4f" % (40 / dt, 1e3 * dt / 40), flush=True)
print("chi2", ba.chi2(), "failures", ba.stats().chol_failures, flush=True)
np.save(ROOT + "/gpurun_out/stage_T_%s.npy" % os.environ.get("TAG", "x"), ba.get_poses()[1])
if "--phases" in sys.argv:
    bp = make(profile=True)
    bench.run_exact(bp, 10); bp.reset_phase_times(); bench.run_exact(bp, 20)
    print({k: round(v[0] / max(v[1], 1) * 1e3, 1) for k, v in bp.phase_times().items()}, flush=True)
